"""Shared helpers for the tests (surrogate steps, emulator loading)."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def crandn(rng, shape):
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)


def dense_scheme_shapes(case):
    """Shapes of both operands at every step of a dense scheme (no arithmetic)."""
    shapes = {i: tuple(t.shape) for i, t in case.tensors.items()}
    out = []
    for (i, j), eq in case.scheme:
        lhs, lo = eq.split("->")
        la, lb = lhs.split(",")
        dim = dict(zip(la, shapes[i]))
        dim.update(zip(lb, shapes[j]))
        out.append((eq, shapes[i], shapes[j]))
        shapes[i] = tuple(dim[x] for x in lo)
    return out


def shrink_step(eq, a_shape, b_shape, max_log2=17):
    """Surrogate of a big step: drop free labels of A (present in A and the output only),
    highest A positions first, until A has at most 2**max_log2 elements.  The contracted
    labels, the small operand and the relative order of every surviving label are kept, so
    the low-bit structure (what the kernel's tiling sees) is the step's own."""
    lhs, lo = eq.split("->")
    la, lb = lhs.split(",")
    la, lb, lo = list(la), list(lb), list(lo)
    a_shape = list(a_shape)
    numel = int(np.prod(a_shape))
    pos = 0
    while numel > 2 ** max_log2 and pos < len(la):
        lab = la[pos]
        if lab not in lb and lab in lo:
            numel //= a_shape[pos]
            la.pop(pos)
            a_shape.pop(pos)
            lo.remove(lab)
        else:
            pos += 1
    return "".join(la) + "," + "".join(lb) + "->" + "".join(lo), tuple(a_shape), tuple(b_shape)


_emu = None


def emulator():
    """Build (g++) and load the CPU emulation of the MFMA kernel's index algebra."""
    global _emu
    if _emu is not None:
        return _emu
    build = os.path.join(ROOT, "tests", "_build")
    os.makedirs(build, exist_ok=True)
    so = os.path.join(build, "libplan_emulate.so")
    src = os.path.join(ROOT, "tests", "csrc", "plan_emulate.cpp")
    deps = [src, os.path.join(ROOT, "artensor_amd", "csrc", "artn_plan.h"), os.path.join(ROOT, "include", "artn.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "artensor_amd", "csrc"), src, "-o", so])
    _emu = ctypes.CDLL(so)
    _emu.artn_emulate.restype = ctypes.c_int
    return _emu


def emulate(eq, a, b, force_generic=False):
    """Run one step through the emulator; returns (result, kernel_used)."""
    import torch
    from artensor_amd import contraction as C
    la, lb, lo = C._parse(eq)
    ta, tb = torch.from_numpy(a), torch.from_numpy(b)
    d, out_shape = C._descriptor(la, lb, lo, tuple(ta.shape), tuple(ta.stride()), tuple(tb.shape),
                                 tuple(tb.stride()), torch.complex64)
    out = np.zeros(out_shape, dtype=np.complex64)
    used = ctypes.c_int(-1)
    rc = emulator().artn_emulate(ctypes.byref(d), a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p),
                                 out.ctypes.data_as(ctypes.c_void_p), int(force_generic), ctypes.byref(used))
    assert rc == 0, rc
    return out, used.value
