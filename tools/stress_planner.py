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
    kernels, fused, declined, triples = {}, 0, 0, 0
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
                    # ... and a third step: the triple query (artn_contract3_query / make_bits3)
                    cand3 = [x for x in lo2 if x != "z"]
                    k3 = int(rng.integers(1, 7))
                    if dtype == torch.complex64 and len(cand3) >= k3:
                        kl3 = [str(x) for x in rng.choice(cand3, size=k3, replace=False)]
                        nl3 = [f"p{x}" for x in range(int(rng.integers(0, 8)))]
                        lb3 = kl3 + nl3
                        rng.shuffle(lb3)
                        lo3 = [x for x in lo2 if x not in kl3] + nl3
                        try:
                            ti = C.triple_info((tuple(la), tuple(lb1), tuple(lo1)), a_shape, ext(lb1), (tuple(lo1), tuple(lb2), tuple(lo2)),
                                               ext(lb2), (tuple(lo2), tuple(lb3), tuple(lo3)), ext(lb3))
                            triples += ti is not None
                        except RuntimeError:
                            declined += 1
    # triples the planner can accept: rank 26..30, rank-preserving steps of 3..5 contracted bits, later steps contracting
    # bits the step before has just produced (consecutive gates on overlapping qubits)
    for case in range(max(1, cases // 10)):
        ra = int(rng.integers(26, 31))
        cur, last_new, eqs, shapes_b = [f"a{x}" for x in range(ra)], [], [], []
        for st in range(3):
            k = int(rng.integers(3, 6))
            pick = [x for x in last_new if rng.random() < 0.7][:k]
            rest = [x for x in cur if x not in pick]
            pick += [str(x) for x in rng.choice(rest, size=k - len(pick), replace=False)]
            new = [f"s{st}n{x}" for x in range(k)]
            lb = pick + new
            rng.shuffle(lb)
            lo = [x for x in cur if x not in pick] + new
            rng.shuffle(lo)
            eqs.append((tuple(cur), tuple(lb), tuple(lo)))
            shapes_b.append((2,) * len(lb))
            cur, last_new = lo, new
        ti = C.triple_info(eqs[0], (2,) * ra, shapes_b[0], eqs[1], shapes_b[1], eqs[2], shapes_b[2])
        if ti is not None:
            triples += 1
            assert ti["tile_in_bits"] == 12 and ti["k3_bits"] in (3, 4, 5) and ti["n_tiles"] >= 1 << 14
    print(f"stress_planner: {cases} cases in {time.time() - t0:.1f} s; kernels {dict(sorted(kernels.items()))}, "
          f"{fused} pairs fused, {triples} triples fused, {declined} descriptors refused; OK")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
