#!/usr/bin/env python3
"""Randomised stress of the HOST planner behind the C ABI (no GPU): artn_contract_query on single steps and
artn_contract2_query on pairs with random label orders, contracted / free bit counts, ragged batch extents, strided
operands, complex64 / complex128 / bf16-operand descriptors.  The planner half of tools/stress_random.py (which runs the
kernels on a GPU); `make asan` runs it under AddressSanitizer + UBSan against the host-only build of the library.

    python tools/stress_planner.py [cases] [seed]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import artensor_amd as A  # noqa: E402
from artensor_amd import contraction as C  # noqa: E402


def main(cases=1000, seed=0):
    rng = np.random.default_rng(seed)
    kernels, fused, declined = {}, 0, 0
    t0 = time.time()
    for case in range(cases):
        dtype = torch.complex128 if rng.random() < 0.3 else torch.complex64
        ra = int(rng.integers(3, 31))
        k1 = int(rng.integers(0, min(ra, 17)))
        n1 = int(rng.integers(0, 13))
        la = [f"a{x}" for x in range(ra)]
        kl1 = [str(x) for x in rng.choice(la, size=k1, replace=False)]
        nl1 = [f"n{x}" for x in range(n1)]
        lb1 = kl1 + nl1
        rng.shuffle(lb1)
        lo1 = [x for x in la if x not in kl1] + nl1
        rng.shuffle(lo1)
        batch = int(rng.choice([0, 0, 0, 3, 4, 7, 100, 619, 10000]))
        if batch:
            la, lb1, lo1 = ["z"] + la, ["z"] + lb1, ["z"] + lo1
        ext = lambda labs: tuple(batch if x == "z" else 2 for x in labs)
        # (a quarter of the cases: the first operand is a view with a padded leading stride or transposed dims)
        a_shape, a_stride = ext(la), None
        if rng.random() < 0.25 and len(a_shape) >= 2:
            dense = list(C._dense_strides(a_shape))
            if rng.random() < 0.5:
                dense[0] *= 2
            else:
                i, j = sorted(rng.choice(len(a_shape), size=2, replace=False))
                dense[i], dense[j] = dense[j], dense[i]
            a_stride = tuple(dense)
        with A.precision("bf16" if dtype == torch.complex64 and rng.random() < 0.2 else "fp32"):
            try:
                info = A.step_info((tuple(la), tuple(lb1), tuple(lo1)), a_shape, ext(lb1), dtype=dtype, a_stride=a_stride)
                kernels[info["kernel"]] = kernels.get(info["kernel"], 0) + 1
                assert info["flops"] >= 0 and info["bytes"] > 0 and info["grid"] >= 0
            except RuntimeError:
                declined += 1   # (a descriptor the ABI refuses: too many labels, extents beyond the envelope)
                continue
            if a_stride is None and rng.random() < 0.5:
                # a second step on the result: the pair query
                k2 = int(rng.integers(1, min(len(lo1), 8) + 1)) if len(lo1) > 1 else 0
                cand = [x for x in lo1 if x != "z"]
                if k2 and len(cand) >= k2:
                    kl2 = [str(x) for x in rng.choice(cand, size=k2, replace=False)]
                    nl2 = [f"m{x}" for x in range(int(rng.integers(0, 9)))]
                    lb2 = kl2 + nl2
                    rng.shuffle(lb2)
                    lo2 = [x for x in lo1 if x not in kl2] + nl2
                    if batch:
                        lo2 = ["z"] + [x for x in lo2 if x != "z"]
                    try:
                        pi = C.pair_info((tuple(la), tuple(lb1), tuple(lo1)), a_shape, ext(lb1), (tuple(lo1), tuple(lb2), tuple(lo2)),
                                         ext(lb2), dtype=dtype)
                        fused += pi is not None
                    except RuntimeError:
                        declined += 1
    print(f"stress_planner: {cases} cases in {time.time() - t0:.1f} s; kernels {dict(sorted(kernels.items()))}, "
          f"{fused} pairs fused, {declined} descriptors refused; OK")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
