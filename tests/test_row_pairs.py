"""The distinct-row form of the sparse executor's batched products over index PAIRS (artensor_amd.contraction.contract_row_pairs;
reference contraction.py:149-156, :177-179): host-side pair bookkeeping on the CPU, the numerics on the GPU."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from artensor_amd import contraction as C  # noqa: E402


def test_pair_bookkeeping_and_when_the_form_is_taken():
    n = 4096
    g = torch.Generator().manual_seed(1)
    ra = torch.randint(0, 64, (n,), generator=g)
    rb = torch.randint(0, 16, (n,), generator=g)
    ua, ub, grid = C._row_pairs(ra, rb, 64, 16)
    assert ua.tolist() == sorted(set(ra.tolist())) and ub.tolist() == sorted(set(rb.tolist()))
    # pair p sits at (position of ra[p] in ua, position of rb[p] in ub) of the distinct x distinct grid
    assert torch.equal(ua[grid // len(ub)], ra) and torch.equal(ub[grid % len(ub)], rb)
    assert C._row_pairs(ra, rb, 64, 16) is C._row_pairs(ra, rb, 64, 16)          # cached per pair of index tensors
    # negative indices count from the end (the reference's indexing semantics), out of range raises
    neg = ra - 64
    ua2, _, grid2 = C._row_pairs(neg, rb, 64, 16)
    assert torch.equal(ua2, ua) and torch.equal(grid2, grid)
    with pytest.raises(RuntimeError, match="out of range"):
        C._row_pairs(ra + 1, rb, 64, 16)
    # no re-use of rows (every pair its own row of a), too few pairs, or a grid much bigger than the pairs: not taken
    assert C._row_pairs(torch.arange(n), rb, n, 16) is None
    assert C._row_pairs(ra[:100], rb[:100], 64, 16) is None
    sparse_a, sparse_b = torch.arange(n) % 1000, (torch.arange(n) * 7) % 1000
    assert C._row_pairs(sparse_a, sparse_b, 1000, 1000) is None                    # 10^6 grid cells for 4 096 pairs


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.complex64, 3e-6), (torch.complex128, 1e-12)])
def test_row_pairs_equal_the_gathered_batched_product(dtype, tol):
    """contract_row_pairs against the same step on materialised gathers (the reference's `tensors[i][idx]` + einsum), and
    through the sparse executor's chunk loop (A) and gathered step (B) with ARTN_ROW_PAIRS on and off."""
    import artensor_amd as A
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(5)
    rnd = lambda *s: torch.view_as_complex(torch.randn(s + (2,), device=dev, generator=g, dtype=torch.float64)).to(dtype)
    a, b = rnd(96, 2, 2, 2, 2, 2, 2), rnd(24, 2, 2, 2, 2, 2, 2)          # z + 6 bits each; 4 contracted
    eq = "zabcdef,zcdefgh->zabgh"
    cg = torch.Generator().manual_seed(6)
    n = 3000
    ra, rb = torch.randint(0, 96, (n,), generator=cg), torch.randint(0, 24, (n,), generator=cg)
    got = C.contract_row_pairs(eq, a, ra, b, rb)
    assert got is not None and tuple(got.shape) == (n, 2, 2, 2, 2)
    want = torch.einsum(eq, a[ra.to(dev)], b[rb.to(dev)])
    assert (got - want).abs().max().item() <= tol * want.abs().max().item()
    # the executor: one gathered step (B), then a chunk loop (A) of two chunks on its result
    c = rnd(8, 2, 2)
    r2a = [torch.randint(0, n, (1500,), generator=cg), torch.randint(0, n, (1400,), generator=cg)]
    r2b = [torch.randint(0, 8, (1500,), generator=cg), torch.randint(0, 8, (1400,), generator=cg)]
    scheme = [((0, 1), eq, [[ra], [rb]], None, (n, 2, 2, 2, 2)),
              ((0, 2), "zabgh,zgh->zab", [r2a, r2b], None, (2900, 2, 2))]
    outs = []
    for flag in ("1", "0"):
        os.environ["ARTN_ROW_PAIRS"] = flag
        try:
            outs.append(A.tensor_contraction_sparse({0: a.clone(), 1: b.clone(), 2: c.clone()}, scheme).cpu())
        finally:
            del os.environ["ARTN_ROW_PAIRS"]
    mid = torch.einsum(eq, a[ra.to(dev)], b[rb.to(dev)])
    ref = torch.cat([torch.einsum("zabgh,zgh->zab", mid[r2a[k].to(dev)], c[r2b[k].to(dev)]) for k in range(2)]).cpu()
    for o in outs:
        assert tuple(o.shape) == (2900, 2, 2)
        assert (o - ref).abs().max().item() <= 2 * tol * ref.abs().max().item()
