"""Worker of tests/test_gpu_parity.py::test_wide_kernel_pairs: run with ARTN_WIDE=1 ARTN_WIDE_MIN_TILES=1 in the environment (the
planner reads its tuning once per process), checks artn_k_wide -- one 8-wave workgroup per CU on one tile, artn_wide_kernel.h --
on the 13 fusable pairs of the n30 scheme (surrogates of 2^22 elements: 4 tiles per workgroup, so prologue, steady state and
the last tile's drain all run) and on random pairs of every (k1, k2) in 3..6, against the ORACLE run step by step."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from artensor_amd.contraction import contract2, fusion_schedule, pair_info   # noqa: E402
from artensor_amd.fixtures import load_case                                   # noqa: E402
from helpers import GOLDEN, crandn, dense_scheme_shapes, shrink_pair          # noqa: E402
from oracle import oracle                                                     # noqa: E402

TOL = 2e-5
gpu = lambda x: torch.from_numpy(x).cuda()
rel = lambda got, want: float(np.abs(got - want).max() / np.abs(want).max())


def is_wide(info):
    # (the only plans with one workgroup per CU and four 32 KiB regions)
    return info is not None and info["lds_bytes"] >= 4 * 32768 and info["grid"] <= 256


def main():
    case = load_case(os.path.join(GOLDEN, "n30_dense.npz"))
    steps = dense_scheme_shapes(case)
    pairs = [e for e in fusion_schedule(case.scheme) if e[0] == "pair" and np.prod(steps[e[1]][1]) >= 2 ** 22]
    wide = 0
    for _, n, m in pairs:
        eq1, sa, sb1 = steps[n]
        eq2, _, sb2 = steps[m]
        e1, a_s, b1_s, e2, b2_s = shrink_pair(eq1, sa, sb1, eq2, sb2, max_log2=22)
        info = pair_info(e1, a_s, b1_s, e2, b2_s)
        rng = np.random.default_rng(n)
        a, b1, b2 = crandn(rng, a_s), crandn(rng, b1_s), crandn(rng, b2_s)
        got = contract2(e1, gpu(a), gpu(b1), e2, gpu(b2))
        if got is None:
            continue
        want = oracle.einsum_pair(e2, oracle.einsum_pair(e1, a, b1), b2)
        err = rel(got.cpu().numpy(), want)
        assert err < TOL, (n, m, err)
        wide += is_wide(info)
    assert wide >= 8, wide
    # random pairs: every (k1, k2), scattered bit positions
    rng = np.random.default_rng(7)
    seen = set()
    for trial in range(200):
        k1, k2 = int(rng.integers(3, 7)), int(rng.integers(3, 7))
        if (k1, k2) in seen:
            continue
        ra = 22
        la = [chr(65 + x) for x in range(ra)]
        kl1 = list(rng.choice(la[: 12 + k1 // 2], size=k1, replace=False))
        nl1 = [chr(97 + x) for x in range(k1)]
        lb1 = kl1 + nl1
        rng.shuffle(lb1)
        lo1 = [x for x in la if x not in kl1]
        pos = sorted(int(x) for x in rng.choice(len(lo1) + 1, size=k1))
        for i, x in zip(reversed(pos), reversed(nl1)):
            lo1.insert(i, x)
        kl2 = list(rng.choice(lo1[: 14], size=k2, replace=False))
        nl2 = [chr(110 + x) for x in range(k2)]
        lb2 = kl2 + nl2
        rng.shuffle(lb2)
        lo2 = [x for x in lo1 if x not in kl2]
        pos = sorted(int(x) for x in rng.choice(len(lo2) + 1, size=k2))
        for i, x in zip(reversed(pos), reversed(nl2)):
            lo2.insert(i, x)
        e1 = "".join(la) + "," + "".join(lb1) + "->" + "".join(lo1)
        e2 = "".join(lo1) + "," + "".join(lb2) + "->" + "".join(lo2)
        info = pair_info(e1, (2,) * ra, (2,) * (2 * k1), e2, (2,) * (2 * k2))
        if not is_wide(info):
            continue
        a, b1, b2 = crandn(rng, (2,) * ra), crandn(rng, (2,) * (2 * k1)), crandn(rng, (2,) * (2 * k2))
        got = contract2(e1, gpu(a), gpu(b1), e2, gpu(b2))
        assert got is not None
        want = oracle.einsum_pair(e2, oracle.einsum_pair(e1, a, b1), b2)
        err = rel(got.cpu().numpy(), want)
        assert err < TOL, (k1, k2, e1, e2, err)
        seen.add((k1, k2))
    assert len(seen) >= 12, sorted(seen)
    print(f"wide ok: {wide} n30 pairs, random pairs {sorted(seen)}")


if __name__ == "__main__":
    main()
