#!/usr/bin/env python3
"""An n53 m14 slice with the chain planner's spill factor at 2.5 (shipped) and at 1 (the spilling 6+5 / 5+5 shrink pairs allowed):
python3 tools/spill_pairs.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for f in ("2.5", "1.0"):
    code = f"import sys; sys.path.insert(0, {ROOT!r}); from artensor_amd import contraction as C; C.CHAIN_SPILL_FACTOR = {f}; sys.argv = ['slice_steps.py', 'n53_m14_sliced.npz', '400']; __file__ = {os.path.join(ROOT, 'tools', 'slice_steps.py')!r}; exec(open({os.path.join(ROOT, 'tools', 'slice_steps.py')!r}).read())"
    print("spill factor", f, flush=True)
    subprocess.run([sys.executable, "-c", code])
