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
    if os.environ.get("ARTN_EMU_LIB"):   # a prebuilt emulator (make asan: the AddressSanitizer / UBSan build)
        _emu = ctypes.CDLL(os.environ["ARTN_EMU_LIB"])
        _emu.artn_emulate.restype = ctypes.c_int
        return _emu
    build = os.path.join(ROOT, "tests", "_build")
    os.makedirs(build, exist_ok=True)
    so = os.path.join(build, "libplan_emulate.so")
    src = os.path.join(ROOT, "tests", "csrc", "plan_emulate.cpp")
    deps = [src, os.path.join(ROOT, "artensor_amd", "csrc", "artn_plan.h"), os.path.join(ROOT, "include", "artn.h"),
            os.path.join(ROOT, "artensor_amd", "csrc", "artn_gemm_kernel.h"),
            os.path.join(ROOT, "artensor_amd", "csrc", "artn_gemm128_kernel.h"),
            os.path.join(ROOT, "artensor_amd", "csrc", "artn_pgemm_kernel.h"),
            os.path.join(ROOT, "artensor_amd", "csrc", "artn_xgemm_plan.h"),
            os.path.join(ROOT, "artensor_amd", "csrc", "artn_xgemm_kernel.h"),
            os.path.join(ROOT, "artensor_amd", "csrc", "artn_xrow_kernel.h")]
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


def emulate128(eq, a, b):
    """One complex128 step through the emulation of artn_k_gemm128; (result, kernel id) or (None, kernel id) when
    the planner gives the step to the strided kernel."""
    import torch
    from artensor_amd import contraction as C
    la, lb, lo = C._labels(eq)
    ta, tb = torch.from_numpy(a), torch.from_numpy(b)
    d, out_shape = C._descriptor(la, lb, lo, tuple(ta.shape), tuple(ta.stride()), tuple(tb.shape), tuple(tb.stride()),
                                 torch.complex128)
    out = np.zeros(out_shape, dtype=np.complex128)
    used = ctypes.c_int(-1)
    rc = emulator().artn_emulate(ctypes.byref(d), a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p),
                                 out.ctypes.data_as(ctypes.c_void_p), 0, ctypes.byref(used))
    if rc == -2:
        return None, used.value
    assert rc == 0, rc
    return out, used.value


def emulate_gemm(eq, a, b, bf16=False, m3=-1):
    """One step through the emulation of the two-operand GEMM kernel (plan forced); returns
    (result, planner info) or (None, None) when the GEMM planner declines the step."""
    import torch
    from artensor_amd import contraction as C
    from artensor_amd import _native as N
    la, lb, lo = C._labels(eq)
    ta, tb = torch.from_numpy(a), torch.from_numpy(b)
    with C.precision("bf16" if bf16 else None):
        d, out_shape = C._descriptor(la, lb, lo, tuple(ta.shape), tuple(ta.stride()), tuple(tb.shape),
                                     tuple(tb.stride()), torch.complex64)
    out = np.zeros(out_shape, dtype=np.complex64)
    info = N.ArtnStepInfo()
    emu = emulator()
    emu.artn_emulate_gemm.restype = ctypes.c_int
    rc = emu.artn_emulate_gemm(ctypes.byref(d), a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p),
                               out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(info), ctypes.c_int(m3))
    if rc == -2:
        return None, None
    assert rc == 0, rc
    return out, {name: getattr(info, name) for name, _ in N.ArtnStepInfo._fields_}


def emulate_pgemm(eq, a, b, bf16=False):
    """One step through the CPU replay of the packed-operand GEMM (packing passes + artn_k_pgemm / artn_k_pgemm3m from the
    same ArtnPackPlan); (result, planner info) or (None, None) when the planner declines."""
    import torch
    from artensor_amd import contraction as C
    from artensor_amd import _native as N
    la, lb, lo = C._labels(eq)
    ta, tb = torch.from_numpy(a), torch.from_numpy(b)
    with C.precision("bf16" if bf16 else None):
        d, out_shape = C._descriptor(la, lb, lo, tuple(ta.shape), tuple(ta.stride()), tuple(tb.shape),
                                     tuple(tb.stride()), torch.complex64)
    out = np.zeros(out_shape, dtype=np.complex64)
    info = N.ArtnStepInfo()
    emu = emulator()
    emu.artn_emulate_pgemm.restype = ctypes.c_int
    rc = emu.artn_emulate_pgemm(ctypes.byref(d), a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p),
                                out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(info))
    if rc == -2:
        return None, None
    assert rc == 0, rc
    return out, {name: getattr(info, name) for name, _ in N.ArtnStepInfo._fields_}


def shrink_pair(eq1, a_shape, b1_shape, eq2, b2_shape, max_log2=17):
    """Surrogate of two consecutive big steps (eq2's first operand is eq1's result, matched
    by position): drop axes that are free in BOTH steps, highest A positions first."""
    lhs, lo1 = eq1.split("->")
    la1, lb1 = lhs.split(",")
    lhs2, lo2 = eq2.split("->")
    la2, lb2 = lhs2.split(",")
    la1, lo1, la2, lo2 = list(la1), list(lo1), list(la2), list(lo2)
    a_shape = list(a_shape)
    numel = int(np.prod(a_shape))
    pos = 0
    while numel > 2 ** max_log2 and pos < len(la1):
        lab = la1[pos]
        if lab not in lb1 and lab in lo1:
            lab2 = la2[lo1.index(lab)]
            if lab2 not in lb2 and lab2 in lo2:
                numel //= a_shape[pos]
                la1.pop(pos)
                a_shape.pop(pos)
                k = lo1.index(lab)
                lo1.pop(k)
                la2.pop(k)
                lo2.remove(lab2)
                continue
        pos += 1
    return ("".join(la1) + "," + lb1 + "->" + "".join(lo1), tuple(a_shape), tuple(b1_shape),
            "".join(la2) + "," + lb2 + "->" + "".join(lo2), tuple(b2_shape))


def shrink_triple(eq1, a_shape, b1_shape, eq2, b2_shape, eq3, b3_shape, max_log2=17):
    """Surrogate of three consecutive big steps (each later first operand is the result before it, matched by position):
    drop axes that are free in ALL three steps, highest A positions first."""
    def parse(eq):
        lhs, lo = eq.split("->")
        la, lb = lhs.split(",")
        return list(la), lb, list(lo)
    la1, lb1, lo1 = parse(eq1)
    la2, lb2, lo2 = parse(eq2)
    la3, lb3, lo3 = parse(eq3)
    a_shape = list(a_shape)
    numel = int(np.prod(a_shape))
    pos = 0
    while numel > 2 ** max_log2 and pos < len(la1):
        lab = la1[pos]
        if lab not in lb1 and lab in lo1:
            lab2 = la2[lo1.index(lab)]
            if lab2 not in lb2 and lab2 in lo2:
                lab3 = la3[lo2.index(lab2)]
                if lab3 not in lb3 and lab3 in lo3:
                    numel //= a_shape[pos]
                    la1.pop(pos)
                    a_shape.pop(pos)
                    k = lo1.index(lab)
                    lo1.pop(k)
                    la2.pop(k)
                    k2 = lo2.index(lab2)
                    lo2.pop(k2)
                    la3.pop(k2)
                    lo3.remove(lab3)
                    continue
        pos += 1
    return ("".join(la1) + "," + lb1 + "->" + "".join(lo1), tuple(a_shape), tuple(b1_shape),
            "".join(la2) + "," + lb2 + "->" + "".join(lo2), tuple(b2_shape),
            "".join(la3) + "," + lb3 + "->" + "".join(lo3), tuple(b3_shape))


def emulate3(eq1, a, b1, eq2, b2, eq3, b3, run=True):
    """Fused triple through the emulator (artn_k_bits3 replayed stage by stage from make_bits3's plan); returns
    (result or None if the planner declines, planner info).  run=False: plan only (full-size tensors: a is a shape carrier)."""
    import torch
    from artensor_amd import contraction as C
    from artensor_amd import _native as N
    emu = emulator()
    emu.artn_emulate3.restype = ctypes.c_int
    ta, tb1, tb2, tb3 = (torch.from_numpy(x) if isinstance(x, np.ndarray) else x for x in (a, b1, b2, b3))
    d1, d2, d3, out_shape = C._triple_descriptors(eq1, ta, tb1, eq2, tb2, eq3, tb3)
    info = N.ArtnStepInfo()
    if not run:
        rc = emu.artn_emulate3(ctypes.byref(d1), ctypes.byref(d2), ctypes.byref(d3), None, None, None, None, None, ctypes.byref(info))
        return (None if rc else True), {name: getattr(info, name) for name, _ in N.ArtnStepInfo._fields_}
    out = np.zeros(out_shape, dtype=np.complex64)
    rc = emu.artn_emulate3(ctypes.byref(d1), ctypes.byref(d2), ctypes.byref(d3), a.ctypes.data_as(ctypes.c_void_p),
                           b1.ctypes.data_as(ctypes.c_void_p), b2.ctypes.data_as(ctypes.c_void_p), b3.ctypes.data_as(ctypes.c_void_p),
                           out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(info))
    if rc == -2:
        return None, None
    assert rc == 0, rc
    return out, {name: getattr(info, name) for name, _ in N.ArtnStepInfo._fields_}


def emulate2(eq1, a, b1, eq2, b2):
    """Fused pair through the emulator; returns the result or None if the planner declines."""
    import torch
    from artensor_amd import contraction as C
    from artensor_amd import _native as N
    emu = emulator()
    emu.artn_emulate2.restype = ctypes.c_int
    ta, tb1, tb2 = torch.from_numpy(a), torch.from_numpy(b1), torch.from_numpy(b2)
    d1, d2, out_shape = C._pair_descriptors(eq1, ta, tb1, eq2, tb2)
    out = np.zeros(out_shape, dtype=a.dtype)   # (complex128 pairs: the replay of artn_k_bits128)
    info = N.ArtnStepInfo()
    rc = emu.artn_emulate2(ctypes.byref(d1), ctypes.byref(d2), a.ctypes.data_as(ctypes.c_void_p),
                           b1.ctypes.data_as(ctypes.c_void_p), b2.ctypes.data_as(ctypes.c_void_p),
                           out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(info))
    if rc == -2:
        return None, None
    assert rc == 0, rc
    return out, {name: getattr(info, name) for name, _ in N.ArtnStepInfo._fields_}


def emulate_bits(eq, a, b):
    """One step forced onto the state-streaming plan (make_bits; complex64 -> artn_k_bits, complex128 -> artn_k_bits128);
    (result, info) or (None, None) when make_bits declines."""
    import torch
    from artensor_amd import contraction as C
    from artensor_amd import _native as N
    la, lb, lo = C._labels(eq)
    ta, tb = torch.from_numpy(a), torch.from_numpy(b)
    d, out_shape = C._descriptor(la, lb, lo, tuple(ta.shape), tuple(ta.stride()), tuple(tb.shape), tuple(tb.stride()), ta.dtype)
    out = np.zeros(out_shape, dtype=a.dtype)
    info = N.ArtnStepInfo()
    emu = emulator()
    emu.artn_emulate_bits.restype = ctypes.c_int
    rc = emu.artn_emulate_bits(ctypes.byref(d), a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p),
                               out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(info))
    if rc == -2:
        return None, None
    assert rc == 0, rc
    return out, {name: getattr(info, name) for name, _ in N.ArtnStepInfo._fields_}


# ---- small-step program images (artn_program_build), executed on the CPU ------------------------------
PROG_MAX_OUT, PROG_MAX_RED, PROG_TASK_ELEMS = 24, 12, 128
PROG_RED_ENTRIES, PROG_ARENA_BYTES = 2048, 128 * 1024
PROG_STEP_DTYPE = np.dtype([
    ("n_out", "<i4"), ("n_red", "<i4"), ("out_numel", "<i4"), ("red_numel", "<i4"), ("a_numel", "<i4"), ("b_numel", "<i4"),
    ("loc_a", "<i8"), ("loc_b", "<i8"), ("loc_c", "<i8"), ("tab_off", "<i8"),
    ("lds_a", "<i4"), ("lds_b", "<i4"), ("lds_c", "<i4"), ("pre_a", "<i4"), ("pre_b", "<i4"), ("to_ws", "<i4"),
    ("red_base", "<i4"), ("level", "<i4"), ("fast", "<i4"), ("n_mbits", "<i4"), ("n_nbits", "<i4"), ("fast_index", "<i4"),
    ("mbit_sA", "<i4", 14), ("mbit_sC", "<i4", 14), ("nbit_sB", "<i4", 10), ("nbit_sC", "<i4", 10),
    ("out_ext", "<i4", PROG_MAX_OUT), ("out_lg", "<i4", PROG_MAX_OUT), ("out_sA", "<i4", PROG_MAX_OUT), ("out_sB", "<i4", PROG_MAX_OUT), ("out_sC", "<i4", PROG_MAX_OUT),
    ("red_ext", "<i4", PROG_MAX_RED), ("red_lg", "<i4", PROG_MAX_RED), ("red_sA", "<i4", PROG_MAX_RED), ("red_sB", "<i4", PROG_MAX_RED)])


def parse_program_image(image):
    """Header, groups, levels, wave tasks and records of an artn_program_build image (numpy uint8 array)."""
    from artensor_amd import _native as N
    buf = np.asarray(image, dtype=np.uint8)
    hdr = np.frombuffer(buf[:32].tobytes(), dtype="<i4")
    magic, n_groups, n_steps, n_levels, n_wtasks = (int(x) for x in hdr[:5])
    assert magic == 0x41525032
    offs = np.frombuffer(buf[32:64].tobytes(), dtype="<i8")
    assert PROG_STEP_DTYPE.itemsize == N.lib().artn_program_record_bytes()
    take = lambda off, dt, n: np.frombuffer(buf[off:off + np.dtype(dt).itemsize * n].tobytes(), dtype=dt)
    groups = take(int(offs[0]), "<i4", 4 * n_groups).reshape(n_groups, 4)
    levels = take(int(offs[1]), "<i4", 4 * n_levels).reshape(n_levels, 4)[:, :2]
    wtasks = take(int(offs[2]), "<i4", 2 * n_wtasks).reshape(n_wtasks, 2)
    recs = take(int(offs[3]), PROG_STEP_DTYPE, n_steps)
    parse_program_image.raw = buf
    return groups, levels, wtasks, recs


def emulate_program(prog, leaves):
    """Execute a compiled small-step program the way artn_k_program does -- per group: preloads, then level by
    level, wave task by wave task, results written in place into the arena / workspace as soon as they are
    computed -- and return the workspace (viewed in the program's element type: complex64, or complex128 with 16-byte
    elements).  An arena block handed out while its previous tenant was still to be read shows up as a wrong result."""
    import torch
    groups, levels, wtasks, recs = parse_program_image(prog.host_image.numpy())
    c128 = getattr(prog, "dtype", torch.complex64) == torch.complex128
    esz, cdt = (16, np.complex128) if c128 else (8, np.complex64)
    assert not (c128 and (recs["fast"] != 0).any())   # matrix-core steps are complex64 only
    ws = np.zeros(prog.ws_bytes // esz + 2, dtype=cdt)
    ext = [np.ascontiguousarray(leaves[t]).reshape(-1).astype(cdt) for t in prog.ext_ids]
    stats = {"levels": len(levels), "wtasks": len(wtasks), "in_lds": 0, "to_ws": 0, "fast": int((recs["fast"] != 0).sum())}
    for (sb, se, lb, le) in groups:
        arena = np.full(PROG_ARENA_BYTES // esz, np.nan + 0j, dtype=cdt)
        red = {}
        for s in range(sb, se):
            R = recs[s]
            assert R["fast"] or R["red_base"] + R["red_numel"] <= PROG_RED_ENTRIES
            raw = parse_program_image.raw
            from_image = np.frombuffer(raw[int(R["tab_off"]): int(R["tab_off"]) + 8 * int(R["red_numel"])].tobytes(), dtype="<i4").reshape(-1, 2)
            tab = []
            for q in range(int(R["red_numel"])):
                rr, ka, kb = q, 0, 0
                for d in range(int(R["n_red"])):
                    e = int(R["red_ext"][d])
                    ka += (rr % e) * int(R["red_sA"][d]); kb += (rr % e) * int(R["red_sB"][d]); rr //= e
                tab.append((ka, kb))
            assert [tuple(int(v) for v in row) for row in from_image] == tab   # the image's own table (fast steps read it)
            red[s] = tab
            for which in "ab":
                if R["pre_" + which]:
                    src = ext[-(int(R["loc_" + which]) + 1)]
                    n, off = int(R[which + "_numel"]), int(R["lds_" + which])
                    assert off % 16 == 0 and off + esz * n <= PROG_ARENA_BYTES and src.size == n
                    arena[off // esz: off // esz + n] = src
        for L in range(lb, le):
            wb, wc = levels[L]
            for (s, first) in wtasks[wb: wb + wc]:
                R = recs[s]
                assert sb <= s < se
                def operand(which):
                    if R["lds_" + which] >= 0:
                        return arena[int(R["lds_" + which]) // esz:]
                    loc = int(R["loc_" + which])
                    return ws[loc // esz:] if loc >= 0 else ext[-(loc + 1)]
                A, B = operand("a"), operand("b")
                ka = np.array([t[0] for t in red[s]], dtype=np.int64); kb = np.array([t[1] for t in red[s]], dtype=np.int64)
                if R["fast"]:
                    # matrix-core task: block of 32 first-operand rows x 16 second-operand columns, offsets from the
                    # per-bit stride tables
                    mb, nb = int(R["n_mbits"]), int(R["n_nbits"])
                    assert mb >= 5 and len(ka) >= 2
                    task = int(first)
                    msub, ntile = task & ((1 << (mb - 5)) - 1), task >> (mb - 5)
                    bits = lambda v, tab, n: sum(int(tab[b]) for b in range(n) if (v >> b) & 1)
                    m = np.arange(msub * 32, msub * 32 + 32)
                    n = np.array([x for x in range(ntile * 16, ntile * 16 + 16) if x < (1 << nb)])
                    oa = np.array([bits(int(v), R["mbit_sA"], mb) for v in m]); ocm = np.array([bits(int(v), R["mbit_sC"], mb) for v in m])
                    ob = np.array([bits(int(v), R["nbit_sB"], nb) for v in n]); ocn = np.array([bits(int(v), R["nbit_sC"], nb) for v in n])
                    Am = A[oa[:, None] + ka[None, :]].astype(np.complex128)            # [m][k]
                    Bm = B[ob[:, None] + kb[None, :]].astype(np.complex128)            # [n][k]
                    acc = Am @ Bm.T                                                     # [m][n]
                    oc = (ocm[:, None] + ocn[None, :]).reshape(-1)
                    acc = acc.reshape(-1)
                else:
                    idx = np.arange(int(first), min(int(first) + PROG_TASK_ELEMS, int(R["out_numel"])))
                    r, oa, ob, oc = idx.copy(), np.zeros_like(idx), np.zeros_like(idx), np.zeros_like(idx)
                    for d in range(int(R["n_out"])):
                        e = int(R["out_ext"][d])
                        oa += (r % e) * int(R["out_sA"][d]); ob += (r % e) * int(R["out_sB"][d]); oc += (r % e) * int(R["out_sC"][d]); r //= e
                    acc = (A[oa[:, None] + ka[None, :]].astype(np.complex128) * B[ob[:, None] + kb[None, :]].astype(np.complex128)).sum(axis=1)
                assert R["lds_c"] >= 0 or R["to_ws"]
                if R["lds_c"] >= 0:
                    assert int(R["lds_c"]) % 16 == 0 and int(R["lds_c"]) + esz * int(R["out_numel"]) <= PROG_ARENA_BYTES
                    arena[int(R["lds_c"]) // esz + oc] = acc
                if R["to_ws"]:
                    ws[int(R["loc_c"]) // esz + oc] = acc
        for s in range(sb, se):
            stats["in_lds"] += int(recs[s]["lds_c"] >= 0)
            stats["to_ws"] += int(recs[s]["to_ws"] != 0)
    return ws, stats


def emulate_xgemm(eq, a, b, n_cu=None, rows16=False):
    """One step through the CPU replay of the extent-based GEMM (artn_k_xgemm; plan forced); returns (result, planner
    info, modes) -- modes = dict(amode, bmode, trans, swapped, nb, flush_chunks, ...) -- or (None, None, None) when
    make_xgemm declines.  `eq`: an einsum string or a triple of label tuples; a / b may be strided views; `n_cu`: the CU
    count the plan is made for (256; a test-size step is a many-round launch on a one-CU device)."""
    import torch
    from artensor_amd import contraction as C
    from artensor_amd import _native as N
    la, lb, lo = C._labels(eq) if isinstance(eq, str) else (tuple(eq[0]), tuple(eq[1]), tuple(eq[2]))
    ta, tb = torch.from_numpy(a), torch.from_numpy(b)
    # (complex128 operands: the replay of artn_k_xgemm128)
    d, out_shape = C._descriptor(la, lb, lo, tuple(ta.shape), tuple(ta.stride()), tuple(tb.shape), tuple(tb.stride()), ta.dtype)
    out = np.full(out_shape, np.nan + 0j, dtype=a.dtype)
    info = N.ArtnStepInfo()
    modes = (ctypes.c_int32 * 9)()
    modes[8] = int(n_cu or 0)
    modes[7] = 1 if rows16 else 0   # (the 16-row shape of the row-streaming form where the planner would take the 64-row one)
    emu = emulator()
    emu.artn_emulate_xgemm.restype = ctypes.c_int
    # (strided views: the emulator takes the base pointer of the view, like the kernel)
    rc = emu.artn_emulate_xgemm(ctypes.byref(d), ctypes.c_void_p(ta.data_ptr()), ctypes.c_void_p(tb.data_ptr()),
                                out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(info), modes)
    if rc == -2:
        return None, None, None
    assert rc == 0, rc
    names = ("amode", "bmode", "trans", "swapped", "nb", "flush_chunks", "kc", "rowmode", "tail_nb")
    return out, {name: getattr(info, name) for name, _ in N.ArtnStepInfo._fields_}, dict(zip(names, modes))
