#!/usr/bin/env python3
"""Diagnostic: label strides (log2) of the big launches of one slice of a fixture: python tools/dump_steps.py <fixture.npz> [sparse]
Per launch: kernel, k, and per class (AB contracted / AC free of A / BC free of B) the log2 strides in A, B and C."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import artensor_amd as A
from artensor_amd import contraction as C
from artensor_amd.fixtures import load_case
case = load_case(os.path.join(ROOT, "tests", "golden", sys.argv[1]))
sparse = len(sys.argv) > 2 and sys.argv[2] == "sparse"
leaves = case.fresh_tensors(device="cuda")
nb = len(case.slicing_indices or {})
orig_q = C._query
seen = []
def q(d):
    info = orig_q(d)
    rows = []
    numel = 1
    for i in range(d.n_labels):
        e, sa, sb, sc = d.extent[i], d.stride_a[i], d.stride_b[i], d.stride_c[i]
        cls = "".join(c for c, st in zip("ABC", (sa, sb, sc)) if st >= 0)
        rows.append((cls, e, sa, sb, sc))
        if sa >= 0: numel *= e
    if numel >= 1 << 26:
        seen.append((info, rows))
    return info
C._query = q
sl = A.apply_slice(leaves, case.slicing_indices, A.slice_assignments(nb, 0)) if nb else dict(leaves)
(A.tensor_contraction_sparse if sparse else A.tensor_contraction)(sl, case.scheme)
torch.cuda.synchronize()
lg = lambda s: "-" if s < 0 else str(s.bit_length() - 1)
for info, rows in seen:
    print(f"kernel={info['kernel']} k={info['k_bits']}+{info['k2_bits']} T={info['tile_in_bits']}/{info['tile_out_bits']} tiles={info['n_tiles']} runs={info['run_in_bits']}/{info['run_out_bits']} GF={info['flops']/1e9:.1f}")
    for cls in ("AB", "AC", "BC", "ABC"):
        r = sorted([x for x in rows if x[0] == cls], key=lambda x: x[2] if x[2] >= 0 else x[3])
        if r:
            print(f"    {cls:3s} ext {[x[1] for x in r]}  A 2^{[lg(x[2]) for x in r]}  B 2^{[lg(x[3]) for x in r]}  C 2^{[lg(x[4]) for x in r]}".replace("'", ""))
