#!/usr/bin/env python3
"""GPU: does the operand layout matter to artn_k_xrow64?  9 -> 9 on 3^16 rows and 27 -> 27 on 3^15 rows with the contracted labels
slowest / inside / fastest in the first operand (answer, round 6: no -- 1.54-1.61 ms whatever the layout; the 10-15 % to the
stand-alone probe are the row-offset tables and the launch, not the strides).   python tools/xrow_layouts.py"""
import sys, time; sys.path.insert(0, '/root/repo')
import torch, artensor_amd as A
def run(name, la, lb, lo, ext):
    sa, sb = tuple(ext[x] for x in la), tuple(ext[x] for x in lb)
    a = torch.view_as_complex(torch.randn(sa + (2,), device='cuda')); b = torch.view_as_complex(torch.randn(sb + (2,), device='cuda'))
    info = A.step_info((la, lb, lo), sa, sb)
    for _ in range(2): out = A.contract((la, lb, lo), a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): out = A.contract((la, lb, lo), a, b)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:50s} m_tile_bits {info['m_tile_bits']} {e0.elapsed_time(e1)/5:.3f} ms")
m = [f"m{i}" for i in range(16)]
ext = {x: 3 for x in m + ["k0", "k1", "n0", "n1"]}
lb = ("n1", "k1", "k0", "n0")
lo = ("n1", "n0") + tuple(m)
run("9->9, contracted labels slowest in the operand", ("k1", "k0") + tuple(m), lb, lo, ext)
run("9->9, contracted labels at positions 6 and 13", tuple(m[:3]) + ("k1",) + tuple(m[3:10]) + ("k0",) + tuple(m[10:]), lb, lo, ext)
run("9->9, contracted labels fastest", tuple(m) + ("k1", "k0"), lb, lo, ext)
m = [f"m{i}" for i in range(15)]
ext = {x: 3 for x in m + ["k0", "k1", "k2", "n0", "n1", "n2"]}
lb = ("n2", "k2", "n1", "k1", "k0", "n0")
lo = ("n2", "n1", "n0") + tuple(m)
run("27->27, contracted labels slowest", ("k2", "k1", "k0") + tuple(m), lb, lo, ext)
run("27->27, contracted labels inside", tuple(m[:4]) + ("k2",) + tuple(m[4:8]) + ("k1", "k0") + tuple(m[8:]), lb, lo, ext)
run("27->27, one contracted label fastest", tuple(m[:7]) + ("k2", "k1") + tuple(m[7:]) + ("k0",), lb, lo, ext)
