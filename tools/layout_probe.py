#!/usr/bin/env python3
"""Does a fused pair of the n30 scheme run faster when its INPUT tiles are contiguous?  (round 5: tools/probes/tri_probe.hip says
a 2^12-element tile read as 256 scattered runs costs 0.5-0.6 ms per 16 GiB pass against a contiguous one, whatever the run
length.)  For every fused pair of the dense n30 scheme: time it as the scheme has it, then with the first operand's labels
reordered [others][free labels of the tile][contracted labels] -- same contraction, same output order.
    python3 tools/layout_probe.py [first_pair last_pair]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import artensor_amd as A
from artensor_amd import contraction as C
from artensor_amd.fixtures import load_case
case = load_case(os.path.join(ROOT, "tests", "golden", "n30_dense.npz"))
shapes = {k: tuple(t.shape) for k, t in case.tensors.items()}
prog, ops = C._compile_dense(case.scheme, dict(shapes), torch.complex64)
pairs = [op for op in ops if len(op.steps) == 2]
lo_, hi_ = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, len(pairs))
g = torch.Generator(device="cuda").manual_seed(1)
def rnd(shape):
    return torch.view_as_complex(torch.randn(tuple(shape) + (2,), generator=g, device="cuda", dtype=torch.float32))
def timeit(fn, n=4):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
for q, op in enumerate(pairs[lo_:hi_], lo_):
    n, m = op.steps
    eq1, eq2 = case.scheme[n][1], case.scheme[m][1]
    la1, lb1, lo1 = C._labels(eq1)
    la2, lb2, lo2 = C._labels(eq2)
    # (every equation has its own letters: the second equation's first operand IS the first result, position by position)
    to1 = dict(zip(la2, lo1))
    k1 = [x for x in la1 if x in lb1 and x not in lo1]
    k2 = [to1[x] for x in la2 if x in lb2 and x not in lo2 and to1[x] in la1]
    kset = k1 + [x for x in k2 if x not in k1]
    # free labels for the tile: the fastest labels of the OUTPUT that come from A (keeps today's write runs), then A's fastest
    room = 12 - len(kset)
    out_from_a = [to1[x] for x in reversed(lo2) if x in to1 and to1[x] in la1]
    cand = [x for x in out_from_a if x not in kset] + [x for x in reversed(la1) if x not in kset]
    mset = list(dict.fromkeys(cand))[:max(room, 0)]
    # (the free labels fastest: the lanes of an operand read run along them -- contracted labels at the bottom of the tile
    #  put the lanes 2^k elements apart: 8-way bank conflicts)
    new_la = [x for x in la1 if x not in kset and x not in mset] + kset + list(reversed(mset))
    a_shape = tuple(2 for _ in la1)
    b1, b2 = rnd(tuple(2 for _ in lb1)), rnd(tuple(2 for _ in lb2))
    a = rnd(a_shape)
    info0 = C.pair_info(eq1, a_shape, b1.shape, eq2, b2.shape)
    eq1n = "".join(new_la) + "," + "".join(lb1) + "->" + "".join(lo1)
    info1 = C.pair_info(eq1n, a_shape, b1.shape, eq2, b2.shape)
    t0 = timeit(lambda: C.contract2(eq1, a, b1, eq2, b2))
    t1 = timeit(lambda: C.contract2(eq1n, a, b1, eq2, b2)) if info1 and not os.environ.get("ONLY_SCHEME") else float("nan")
    f = lambda i: f"k={i['k_bits']}+{i['k2_bits']} runs {i['run_in_bits']}/{i['run_out_bits']} T {i['tile_in_bits']}/{i['tile_out_bits']} rr {i['a_rereads']}" if i else "declined"
    print(f"pair {q} steps {n}+{m}: scheme layout {t0:6.3f} ms ({f(info0)})   contiguous input tiles {t1:6.3f} ms ({f(info1)})", flush=True)
    del a, b1, b2
