"""Host-side mirror of the reference's executor module (artensor/contraction.py).

Same entry points, same argument meaning, same step formats:

    contraction_scheme(ctree)                      -> (scheme, output_bonds)   ref :23-59
    tensor_contraction(tensors, scheme)            -> Tensor                   ref :62-76
    contraction_scheme_sparse(ctree, bitstrings, sc_target)                    ref :208-341
    tensor_contraction_sparse(tensors, scheme, scientific_notation=False)      ref :132-205

but every `torch.einsum`, row gather, `torch.cat` and renormalisation of the reference is
one call through the C ABI of libartn_hip.so (include/artn.h) into hand-written gfx950
kernels.  Tree traversal stays in Python on PyTorch-ROCm tensors.  Errors raise
RuntimeError (the reference prints and sys.exit(1)s, contraction.py:71-74).
"""
import ctypes
import threading
import warnings
from math import ceil

import numpy as np
import torch

from . import _native as N

__all__ = [
    "einsum_eq_convert", "contraction_scheme", "tensor_contraction",
    "contraction_scheme_sparse", "tensor_contraction_sparse", "contract", "step_info",
]

# the reference's einsum alphabet (contraction.py:9-10): A-Y, a-y  (Z/z excluded)
letters = [chr(c) for c in list(range(65, 90)) + list(range(97, 122))]


# ----------------------------------------------------------------------------------------
# one pairwise step through the C ABI
# ----------------------------------------------------------------------------------------
def _parse(eq):
    try:
        lhs, out = eq.split("->")
        a, b = lhs.split(",")
    except ValueError:
        raise RuntimeError(f"not a two-operand einsum equation: {eq!r}")
    for part in (a, b, out):
        if len(set(part)) != len(part):
            raise RuntimeError(f"repeated label inside one operand is not supported: {eq!r}")
    for lab in out:
        if lab not in a and lab not in b:
            raise RuntimeError(f"output label {lab!r} not carried by an operand: {eq!r}")
    return tuple(a), tuple(b), tuple(out)


def _labels(eq):
    """An equation is an einsum string or a triple of label tuples (any hashable labels: no
    50-letter limit, reference contraction.py:9-10)."""
    return _parse(eq) if isinstance(eq, str) else (tuple(eq[0]), tuple(eq[1]), tuple(eq[2]))


_DTYPES = {torch.complex64: N.ARTN_C64, torch.complex128: N.ARTN_C128}


class _Bounded(dict):
    """A dict that forgets everything once it holds `limit` entries: the host-side caches below are
    pure memoisation (descriptors, planner answers, device copies of index tensors), so dropping them
    costs a recomputation, never a wrong answer."""

    def __init__(self, limit):
        super().__init__()
        self.limit = limit

    def __setitem__(self, key, value):
        if len(self) >= self.limit and key not in self:
            self.clear()
        super().__setitem__(key, value)


def _same_steps(snapshot, scheme, _is=__import__("operator").is_):
    """Caches keyed on id(scheme) hold the scheme object (so its id cannot be reused) AND a tuple of its step objects as
    they were when it was compiled: the reference re-reads the list on every call (contraction.py:66), so a scheme list
    mutated in place between two calls -- steps appended, removed, replaced or reordered -- must not replay the stale
    plan.  One pass of pointer comparisons in C (68 steps: ~1 us).  Steps themselves are treated as values: editing the
    inside of a step's own lists in place is not seen."""
    return len(snapshot) == len(scheme) and all(map(_is, snapshot, scheme))


_MISS = object()


class _IdMemo(_Bounded):
    """Memo keyed on the IDENTITY of objects -- scheme lists, descriptors, the index tensors of a scheme -- plus hashable
    extras.  The reference re-reads its inputs on every call; what is cached here is derived from objects the caller keeps
    (a scheme is reused for every slice), so the key is who they are, not what they hold.  The one rule, in one place: an
    entry HOLDS the objects it was made for (their ids cannot be reused while it lives), is returned only when each of them
    `is` the caller's, and -- for scheme lists, which callers may edit in place between two calls -- only when the list
    still holds the same step objects (_same_steps).  (tensor_contraction's own two look-ups, _scheme_ids and _plan_cache,
    stay hand-written: they share one snapshot check per call on the launch-latency path.)"""

    def find(self, objs, extra=(), schemes=()):
        hit = self.get(tuple(map(id, objs)) + tuple(extra))
        if hit is None:
            return _MISS
        held, snaps, value = hit
        for h, o in zip(held, objs):
            if h is not o:
                return _MISS
        for snap, sch in zip(snaps, schemes):
            if not _same_steps(snap, sch):
                return _MISS
        return value

    def keep(self, objs, value, extra=(), schemes=()):
        self[tuple(map(id, objs)) + tuple(extra)] = (tuple(objs), tuple(tuple(sch) for sch in schemes), value)
        return value


_desc_cache = _Bounded(8192)   # key: labels + shapes + strides of one step (values, not identities)

# Optional per-launch timing hook (bench.py / profiling only): an object with
# .record(info_dict, start_event, end_event); events are recorded on the launch stream.
profiler = None
_info_cache = _IdMemo(8192)    # descriptor -> planner answer


def _query(d):
    info = N.ArtnStepInfo()
    N.check(N.lib().artn_contract_query(ctypes.byref(d), ctypes.byref(info)))
    return {name: getattr(info, name) for name, _ in N.ArtnStepInfo._fields_}


def _step_info_cached(d):
    info = _info_cache.find((d,))
    return _info_cache.keep((d,), _query(d)) if info is _MISS else info


# A big step that the bit planner declines runs on the strided kernel (one thread per output
# element): correct, but far from the MFMA kernel's rate.  Say so once per reason.
GENERIC_WARN_NUMEL = 1 << 20
_warned_generic = set()


def _warn_if_generic(d, numel, what):
    if numel < GENERIC_WARN_NUMEL:
        return
    if _info_cache.find((d,)) is not _MISS:
        return                    # already looked at (and warned about, if need be)
    info = _step_info_cached(d)   # leaves the planner's reason in artn_last_plan_note()
    if info["kernel"] != N.KERNEL_GENERIC:
        return
    note = N.lib().artn_last_plan_note().decode()
    if note in _warned_generic:
        return
    _warned_generic.add(note)
    warnings.warn(f"artensor_amd: {what} with a {numel}-element operand runs on the strided fallback kernel "
                  f"(artn_k_generic), not the MFMA kernel: {note}", RuntimeWarning, stacklevel=3)


class precision:
    """Context manager selecting the arithmetic of the big steps of complex64 contractions:

        with artensor_amd.precision("bf16"):
            amps = artensor_amd.tensor_contraction(tensors, scheme)

    "bf16": tensors stay complex64 in memory; the operands of the MFMA kernel are rounded to
    bfloat16 (round to nearest even) and accumulated in fp32 -- the reduced-precision sampling
    mode of BASELINE configs[4].  The reference has no such path: it is defined against this
    package's own complex64 results (tests check state fidelity).  None / "fp32": the default."""
    _state = threading.local()   # per thread: a `with precision(...)` in one thread leaves the others alone

    def __init__(self, mode):
        if mode not in (None, "fp32", "bf16"):
            raise RuntimeError(f"unknown precision {mode!r} (use None, 'fp32' or 'bf16')")
        self.mode = None if mode == "fp32" else mode

    @staticmethod
    def current():
        return getattr(precision._state, "mode", None)

    def __enter__(self):
        self._prev = precision.current()
        precision._state.mode = self.mode
        return self

    def __exit__(self, *exc):
        precision._state.mode = self._prev
        return False


def _descriptor(la, lb, lo, a_shape, a_stride, b_shape, b_stride, dtype):
    reduced = precision.current() == "bf16" and dtype == torch.complex64
    key = (la, lb, lo, a_shape, a_stride, b_shape, b_stride, dtype, reduced)
    hit = _desc_cache.get(key)
    if hit is not None:
        return hit
    if len(la) != len(a_shape) or len(lb) != len(b_shape):
        raise RuntimeError(f"operand rank does not match the equation: {la} {a_shape} / {lb} {b_shape}")
    labels = list(la) + [x for x in lb if x not in la]
    if len(labels) > N.ARTN_MAX_LABELS:
        raise RuntimeError(f"step has {len(labels)} labels; the ABI carries at most {N.ARTN_MAX_LABELS}")
    ext = {}
    for lab, n in zip(la, a_shape):
        ext[lab] = n
    for lab, n in zip(lb, b_shape):
        if ext.setdefault(lab, n) != n:
            raise RuntimeError(f"label {lab!r} has extent {ext[lab]} in one operand and {n} in the other")
    out_shape = tuple(ext[x] for x in lo)
    c_stride = {}
    s = 1
    for lab in reversed(lo):
        c_stride[lab] = s
        s *= ext[lab]
    d = N.ArtnStepDesc()
    d.dtype = N.ARTN_C64_BF16 if reduced else _DTYPES[dtype]
    d.n_labels = len(labels)
    sa, sb = dict(zip(la, a_stride)), dict(zip(lb, b_stride))
    for n, lab in enumerate(labels):
        d.extent[n] = ext[lab]
        d.stride_a[n] = sa.get(lab, -1)
        d.stride_b[n] = sb.get(lab, -1)
        d.stride_c[n] = c_stride.get(lab, -1)
    hit = (d, out_shape)
    _desc_cache[key] = hit
    return hit


def _as_operand(t):
    # degenerate strides (expanded views) are materialised; ordinary views pass through
    if any(st == 0 and n > 1 for st, n in zip(t.stride(), t.shape)):
        return t.contiguous()
    return t


MAX_TILE_K_BITS = 8   # contracted bits one LDS tile of the state-streaming kernel can hold
SPLIT_K_MIN_TILES = int(__import__("os").environ.get("ARTN_SPLIT_K_MIN_TILES", "512"))   # the GEMM kernel loops over any number of
                          # contracted bits inside a workgroup; contracted labels are turned into a batch label only to get this many tiles


SPLIT_K_FEW_TILES = int(__import__("os").environ.get("ARTN_SPLIT_K_FEW_TILES", "4096"))   # ... this many when the unsplit step has fewer tiles than CUs
XGEMM_FEW_TILES = 256     # below this a step cannot even fill the CUs: contracted values are split off down to 32 per tile
XGEMM_SPLIT_TILES = 4096  # the extent GEMM's persistent grid is 512 workgroups: eight rounds keep the last one's idle share under 6 %
_ONE = object()  # operand id of the scalar 1 in a compiled sum-out op
_ones = {}


def _one_scalar(dtype, device):
    key = (dtype, str(device))
    t = _ones.get(key)
    if t is None:
        t = _ones[key] = torch.ones((), dtype=dtype, device=device)
    return t


def _big_k_outer(la, lb, lo, a_shape, a_stride=None, b_shape=None, b_stride=None, dtype=torch.complex64):
    """Contracted labels to turn into a temporary batch label (split-K with the partial results in
    HBM, summed afterwards), or None.

    A step that contracts more bits than one LDS tile of the state-streaming kernel holds runs on the
    two-operand GEMM kernel, which walks all contracted bits inside the workgroup -- no split needed
    unless the result is so small that its tiles cannot fill the chip (closing steps: two 2^30
    tensors contracted to 2^10 amplitudes); then the slowest-varying contracted labels of A are
    split off until SPLIT_K_MIN_TILES workgroups have work.  Where the planner does not give the
    step to the GEMM kernel the old rule applies: split until MAX_TILE_K_BITS remain."""
    numel = 1
    for e in a_shape:
        numel *= e
    if numel < (1 << 16):
        return None
    if a_stride is None:
        a_stride = _dense_strides(tuple(a_shape))
    ka = [(a_stride[n], x) for n, x in enumerate(la) if x in lb and x not in lo]
    bits = 0
    for _, x in ka:
        e = a_shape[la.index(x)]
        if e & (e - 1):
            return _big_k_outer_extents(la, lb, lo, a_shape, a_stride, b_shape, b_stride, dtype, ka)
        bits += e.bit_length() - 1
    if numel < (1 << 20):
        return None
    if bits <= MAX_TILE_K_BITS:
        return None
    keep = MAX_TILE_K_BITS
    if b_shape is not None:
        d, out_shape = _descriptor(tuple(la), tuple(lb), tuple(lo), tuple(a_shape), tuple(a_stride), tuple(b_shape),
                                   tuple(b_stride if b_stride is not None else _dense_strides(tuple(b_shape))),
                                   dtype)
        info = _step_info_cached(d)
        if info["kernel"] == N.KERNEL_PGEMM:
            return None   # packed-operand GEMM: every contracted bit is looped over in the kernel, thousands of tiles
        if info["kernel"] == N.KERNEL_GEMM_MFMA:
            want = 0   # bits to split off for parallelism
            tiles = max(1, info["n_tiles"])
            # (a step that cannot fill the CUs even once is one long chain per workgroup: 64 tiles x 2^16 contracted values of the
            #  D = 4 network ran at 44 TFLOP/s split into 256 x 2^14 -- such steps are split on, up to SPLIT_K_FEW_TILES ...
            while (tiles << want) < SPLIT_K_MIN_TILES and bits - want > 6:
                want += 1
            if tiles < XGEMM_FEW_TILES:
                # ... unless its operands stream from HBM for longer than a tile's chain of chunks takes anyway (1.5 us per chunk of
                # 32 contracted values, measured): the closing steps of n53 m20 -- 1 024 results of 2^25-term sums, 17 GB -- lost
                # 13 % with 2 048 tiles of 2^14 values instead of 512 of 2^16
                stream_us = info.get("bytes", 0.0) / 5.0e12 * 1e6
                while (tiles << want) < SPLIT_K_FEW_TILES and bits - want > 6 and 1.5 * (1 << max(0, bits - want - 5)) > stream_us:
                    want += 1
            if want == 0:
                return None
            keep = bits - want
    ka.sort(reverse=True)  # highest A stride first
    outer = []
    for _, x in ka:
        if bits <= keep:
            break
        outer.append(x)
        bits -= a_shape[la.index(x)].bit_length() - 1
    return outer


def _big_k_outer_extents(la, lb, lo, a_shape, a_stride, b_shape, b_stride, dtype, ka):
    """_big_k_outer for contracted labels whose extents are not powers of two (the extent-based GEMM, artn_k_xgemm, walks
    every contracted value inside one workgroup): when the result has too few 128 x 96 tiles to fill the chip -- the
    closing steps of a bond-dimension-3 network contract 3^11 values into a 3^7 x 3^5 result: 36 tiles -- the
    slowest-varying contracted labels of A become a temporary batch label until SPLIT_K_MIN_TILES workgroups have
    work, as long as a few hundred contracted values stay inside (a few dozen while the step has fewer tiles than the chip CUs)."""
    if b_shape is None or dtype != torch.complex64:
        return None
    d, _ = _descriptor(tuple(la), tuple(lb), tuple(lo), tuple(a_shape), tuple(a_stride), tuple(b_shape),
                       tuple(b_stride if b_stride is not None else _dense_strides(tuple(b_shape))), dtype)
    info = _step_info_cached(d)
    if info["kernel"] != N.KERNEL_XGEMM:
        return None
    tiles = max(1, info["n_tiles"])
    k_total = 1
    for _, x in ka:
        k_total *= a_shape[la.index(x)]
    outer = []
    for _, x in sorted(ka, reverse=True):   # highest A stride first
        e = a_shape[la.index(x)]
        # (fewer tiles than CUs: a chain of chunks at one workgroup per CU, 3 us per chunk of 16 values -- 6 tiles x 1 296 values
        #  took 0.24 ms for 0.1 GFLOP in the bond-dimension-6 network: split down to a few dozen values per tile)
        if tiles >= XGEMM_SPLIT_TILES or k_total // e < (256 if tiles >= XGEMM_FEW_TILES else 32):
            break
        outer.append(x)
        tiles *= e
        k_total //= e
    # The split step carries `outer` as batch labels (in A, B and the result): the planner has its own limits for those
    # (ARTN_XG_MAXH batch labels, 20 labels per side) -- a split the extent GEMM declines would run the whole contraction on
    # the strided kernel instead.  Ask for the split descriptor and give labels back until the answer is the extent GEMM.
    bs = tuple(b_stride if b_stride is not None else _dense_strides(tuple(b_shape)))
    while outer:
        mid = tuple(outer) + tuple(lo)
        ds, _ = _descriptor(tuple(la), tuple(lb), mid, tuple(a_shape), tuple(a_stride), tuple(b_shape), bs, dtype)
        if _step_info_cached(ds)["kernel"] == N.KERNEL_XGEMM:
            break
        outer.pop()
    return outer or None


def sum_leading(part, n_rows, out=None):
    """out[c] = sum_r part.reshape(n_rows, -1)[r, c] through artn_sum_axis_c64, as a two-pass tree
    when there are many rows (the first pass keeps every CU busy, the order of additions is
    fixed).  `part` must be a contiguous complex64 (artn_sum_axis_c64) or complex128 (artn_sum_axis_c128) GPU tensor."""
    n_cols = part.numel() // n_rows
    lib = N.lib()
    sum_axis = lib.artn_sum_axis_c64 if part.dtype == torch.complex64 else lib.artn_sum_axis_c128
    per16 = 2 if part.dtype == torch.complex64 else 1   # elements per 16-byte lane
    dst = out
    if out is None or not out.is_contiguous() or out.data_ptr() % 16 or out.numel() != n_cols:   # (tmp rows of an odd n_cols are 8-byte aligned: fine)
        out = torch.empty(n_cols, dtype=part.dtype, device=part.device)
    with torch.cuda.device(part.device):
        stream = N.current_stream_ptr(part.device)
        groups = 1
        col_tiles = (n_cols // per16 + 63) // 64
        while groups * col_tiles < 2048 and n_rows % (groups * 2) == 0 and n_rows // (groups * 2) >= 16:
            groups *= 2
        if groups > 1:
            tmp = torch.empty(groups * n_cols, dtype=part.dtype, device=part.device)
            N.check(sum_axis(part.data_ptr(), tmp.data_ptr(), groups, n_rows // groups, n_cols, stream))
            N.check(sum_axis(tmp.data_ptr(), out.data_ptr(), 1, groups, n_cols, stream))
        else:
            N.check(sum_axis(part.data_ptr(), out.data_ptr(), 1, n_rows, n_cols, stream))
    if dst is not None and dst is not out:
        dst.copy_(out.reshape(dst.shape))
        return dst
    return out


def _sum_leading_ok(t, n_rows):
    if not (t.is_contiguous() and n_rows > 1 and t.numel() % n_rows == 0 and t.data_ptr() % 16 == 0):
        return False
    return t.dtype in (torch.complex128, torch.complex64)   # (complex64: an odd column count moves one element per lane)


def _split_big_k(la, lb, lo, a, b):
    """More contracted bits than one LDS tile holds (closing steps of the sparse path contract
    15 bonds at once): keep the slowest-varying contracted labels as a temporary batch label,
    run the MFMA kernel per value, then sum that label out -- split-K with the partial results
    in HBM.  Returns None when the step does not need / allow it."""
    if a.dtype not in _DTYPES:
        return None
    outer = _big_k_outer(la, lb, lo, tuple(a.shape), tuple(a.stride()), tuple(b.shape), tuple(b.stride()), a.dtype)
    if not outer:
        return None
    mid = tuple(outer) + tuple(lo)
    part = contract((la, lb, mid), a, b)
    return (mid, (), tuple(lo)), part, _one_scalar(a.dtype, a.device), len(outer)


def _launch_step(d, a, b, out, stream):
    """artn_contract, or artn_contract_ws with a scratch buffer where the planner asks for one (big contractions over
    2^9+ / 2^10+ values pack their operands first: to bfloat16 in the reduced-precision mode, in tile order otherwise).  The library never allocates: the scratch is
    a torch buffer, returned to the caching allocator in stream order.  Returns the status code."""
    lib = N.lib()
    if d.dtype != N.ARTN_C128 and max(a.numel(), b.numel()) >= (1 << 20):   # (only big contractions ever pack their operands;
        #                                 the planner may swap them, so the bigger of the two decides: what is queried is what runs)
        ws_bytes = _step_info_cached(d)["workspace_bytes"]
        if ws_bytes > 0:
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a.device)
            return lib.artn_contract_ws(ctypes.byref(d), a.data_ptr(), b.data_ptr(), out.data_ptr(), ws.data_ptr(), ws_bytes, stream)
    return lib.artn_contract(ctypes.byref(d), a.data_ptr(), b.data_ptr(), out.data_ptr(), stream)


def contract(eq, a, b, out=None):
    """C = einsum(eq, a, b) on the GPU through artn_contract (stands in for torch.einsum at
    reference contraction.py:70,147,156,163,169,179,181,190).  `eq` is an einsum string or
    a triple of label tuples (labels may then be any hashables: no 50-letter limit)."""
    la, lb, lo = _parse(eq) if isinstance(eq, str) else (tuple(eq[0]), tuple(eq[1]), tuple(eq[2]))
    N.require_gpu(a, "contract")
    N.require_gpu(b, "contract")
    split = _split_big_k(la, lb, lo, a, b) if a.is_cuda and b.is_cuda and a.dtype == b.dtype else None
    if split is not None:
        (la, lb, lo), a, b, n_outer = split
        n_rows = 1
        for e in a.shape[:n_outer]:
            n_rows *= e
        if _sum_leading_ok(a, n_rows):   # the temporary label leads: a plain column sum
            if out is None:
                return sum_leading(a, n_rows).reshape(a.shape[n_outer:])
            if out.is_contiguous() and out.dtype == a.dtype and out.device == a.device and tuple(out.shape) == tuple(a.shape[n_outer:]):
                # (with out=: the chunk loop of the sparse executor in complex128 -- the sum went to the strided kernel)
                sum_leading(a, n_rows, out=out.reshape(-1))
                return out
    if a.dtype != b.dtype or a.dtype not in _DTYPES:
        raise RuntimeError(f"operands must both be complex64 or complex128, got {a.dtype} and {b.dtype}")
    if a.device != b.device:
        raise RuntimeError(f"operands live on different devices: {a.device} / {b.device}")
    a, b = _as_operand(a), _as_operand(b)
    d, out_shape = _descriptor(la, lb, lo, tuple(a.shape), tuple(a.stride()), tuple(b.shape),
                               tuple(b.stride()), a.dtype)
    if out is None:
        out = torch.empty(out_shape, dtype=a.dtype, device=a.device)
    else:
        if tuple(out.shape) != out_shape or not out.is_contiguous() or out.dtype != a.dtype or out.device != a.device:
            raise RuntimeError("out= must be a contiguous tensor of the result shape and dtype on the operands' device")
    if out.numel() == 0:
        return out
    _warn_if_generic(d, max(a.numel(), out.numel()), f"contract({eq!r})" if isinstance(eq, str) else "contract()")
    with torch.cuda.device(a.device):
        if profiler is None:
            N.check(_launch_step(d, a, b, out, N.current_stream_ptr(a.device)))
        else:
            info = _step_info_cached(d)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            N.check(_launch_step(d, a, b, out, N.current_stream_ptr(a.device)))
            e1.record()
            profiler.record(info, e0, e1)
    return out


def _dense_strides(shape):
    st, s = [], 1
    for n in reversed(shape):
        st.append(s)
        s *= n
    return tuple(reversed(st))


_pair_cache = _IdMemo(4096)     # (descriptor of step 1, of step 2) -> planner answer or False


def _resolve_reshape(numel, shape):
    known = 1
    for e in shape:
        if e != -1:
            known *= e
    if known == 0 or numel % known:
        raise RuntimeError(f"cannot view {numel} elements as {tuple(shape)}")
    return tuple(numel // known if e == -1 else e for e in shape)


def _pair_descriptors(eq1, a, b1, eq2, b2, mid_view=None):
    """mid_view: the shape the second equation sees the first result in (the free reshape between
    two steps of the sparse executor, reference contraction.py:181); same memory, so only the
    second descriptor's label split changes."""
    la1, lb1, lo1 = _parse(eq1) if isinstance(eq1, str) else tuple(map(tuple, eq1))
    la2, lb2, lo2 = _parse(eq2) if isinstance(eq2, str) else tuple(map(tuple, eq2))
    d1, mid_shape = _descriptor(la1, lb1, lo1, tuple(a.shape), tuple(a.stride()), tuple(b1.shape),
                                tuple(b1.stride()), a.dtype)
    if mid_view is not None:
        numel = 1
        for e in mid_shape:
            numel *= e
        mid_shape = _resolve_reshape(numel, mid_view)
    if len(la2) != len(mid_shape):
        raise RuntimeError("second equation's first operand does not match the first result")
    d2, out_shape = _descriptor(la2, lb2, lo2, mid_shape, _dense_strides(mid_shape), tuple(b2.shape),
                                tuple(b2.stride()), a.dtype)
    return d1, d2, out_shape


def _triple_descriptors(eq1, a, b1, eq2, b2, eq3, b3):
    """Descriptors of three consecutive steps on the same first operand: every later step sees the result before it as
    a dense tensor in that result's label order."""
    la1, lb1, lo1 = _parse(eq1) if isinstance(eq1, str) else tuple(map(tuple, eq1))
    la2, lb2, lo2 = _parse(eq2) if isinstance(eq2, str) else tuple(map(tuple, eq2))
    la3, lb3, lo3 = _parse(eq3) if isinstance(eq3, str) else tuple(map(tuple, eq3))
    d1, s1 = _descriptor(la1, lb1, lo1, tuple(a.shape), tuple(a.stride()), tuple(b1.shape), tuple(b1.stride()), a.dtype)
    if len(la2) != len(s1):
        raise RuntimeError("second equation's first operand does not match the first result")
    d2, s2 = _descriptor(la2, lb2, lo2, s1, _dense_strides(s1), tuple(b2.shape), tuple(b2.stride()), a.dtype)
    if len(la3) != len(s2):
        raise RuntimeError("third equation's first operand does not match the second result")
    d3, s3 = _descriptor(la3, lb3, lo3, s2, _dense_strides(s2), tuple(b3.shape), tuple(b3.stride()), a.dtype)
    return d1, d2, d3, s3


_triple_cache = _IdMemo(4096)


def contract3(eq1, a, b1, eq2, b2, eq3, b3):
    """einsum(eq3, einsum(eq2, einsum(eq1, a, b1), b2), b3) in ONE pass over HBM through artn_contract3: neither
    intermediate leaves LDS (three consecutive steps of reference contraction.py:66-70 on the state tensor).  Returns None
    when the planner declines the triple (the caller falls back to a pair and a single step)."""
    if not N.has("artn_contract3"):
        raise RuntimeError("three-step fusion (artn_contract3) is compiled into development builds only: make dev")
    for t in (a, b1, b2, b3):
        N.require_gpu(t, "contract3")
    if not (a.dtype == b1.dtype == b2.dtype == b3.dtype == torch.complex64) or not a.is_contiguous() or precision.current() not in (None, "fp32"):
        return None
    b1, b2, b3 = _as_operand(b1), _as_operand(b2), _as_operand(b3)
    d1, d2, d3, out_shape = _triple_descriptors(eq1, a, b1, eq2, b2, eq3, b3)
    info = _triple_cache.find((d1, d2, d3))
    if info is _MISS:
        q = N.ArtnStepInfo()
        rc = N.lib().artn_contract3_query(ctypes.byref(d1), ctypes.byref(d2), ctypes.byref(d3), ctypes.byref(q))
        if rc == -2:
            info = False
        else:
            N.check(rc)
            info = {name: getattr(q, name) for name, _ in N.ArtnStepInfo._fields_}
        _triple_cache.keep((d1, d2, d3), info)
    if info is False:
        return None
    out = torch.empty(out_shape, dtype=a.dtype, device=a.device)
    with torch.cuda.device(a.device):
        e0 = e1 = None
        if profiler is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        N.check(N.lib().artn_contract3(ctypes.byref(d1), ctypes.byref(d2), ctypes.byref(d3), a.data_ptr(), b1.data_ptr(),
                                       b2.data_ptr(), b3.data_ptr(), out.data_ptr(), N.current_stream_ptr(a.device)))
        if profiler is not None:
            e1.record()
            profiler.record(info, e0, e1)
    return out


def triple_info(eq1, a_shape, b1_shape, eq2, b2_shape, eq3, b3_shape):
    """Planner decision for fusing three consecutive steps (host only); None if the triple does not fit."""
    class _S:  # shape/stride carrier
        def __init__(self, shape):
            self.shape, self._st, self.dtype = tuple(shape), _dense_strides(tuple(shape)), torch.complex64

        def stride(self):
            return self._st
    if not N.has("artn_contract3_query"):
        return None   # (the product library has no triples: development builds only, make dev)
    d1, d2, d3, out_shape = _triple_descriptors(eq1, _S(a_shape), _S(b1_shape), eq2, _S(b2_shape), eq3, _S(b3_shape))
    info = N.ArtnStepInfo()
    rc = N.lib().artn_contract3_query(ctypes.byref(d1), ctypes.byref(d2), ctypes.byref(d3), ctypes.byref(info))
    if rc == -2:
        return None
    N.check(rc)
    res = {name: getattr(info, name) for name, _ in N.ArtnStepInfo._fields_}
    res["out_shape"] = out_shape
    return res


def pair_info(eq1, a_shape, b1_shape, eq2, b2_shape, dtype=torch.complex64):
    """Planner decision for fusing two consecutive steps (host only); None if not fusable."""
    class _S:  # shape/stride carrier
        def __init__(self, shape):
            self.shape, self._st, self.dtype = tuple(shape), _dense_strides(tuple(shape)), dtype

        def stride(self):
            return self._st
    d1, d2, out_shape = _pair_descriptors(eq1, _S(a_shape), _S(b1_shape), eq2, _S(b2_shape))
    info = N.ArtnStepInfo()
    rc = N.lib().artn_contract2_query(ctypes.byref(d1), ctypes.byref(d2), ctypes.byref(info))
    if rc == -2:
        return None
    N.check(rc)
    res = {name: getattr(info, name) for name, _ in N.ArtnStepInfo._fields_}
    res["out_shape"] = out_shape
    return res


def contract2(eq1, a, b1, eq2, b2, mid_view=None):
    """einsum(eq2, einsum(eq1, a, b1), b2) in ONE pass over HBM through artn_contract2: the
    intermediate never leaves LDS.  Returns None when the planner declines to fuse the
    pair (the caller then runs the two steps one after the other).  mid_view: shape in which
    eq2 addresses the first result (see _pair_descriptors)."""
    for t in (a, b1, b2):
        N.require_gpu(t, "contract2")
    if not (a.dtype == b1.dtype == b2.dtype and a.dtype in _DTYPES) or not a.is_contiguous():
        return None
    b1, b2 = _as_operand(b1), _as_operand(b2)
    d1, d2, out_shape = _pair_descriptors(eq1, a, b1, eq2, b2, mid_view)
    info = _pair_cache.find((d1, d2))
    if info is _MISS:
        q = N.ArtnStepInfo()
        rc = N.lib().artn_contract2_query(ctypes.byref(d1), ctypes.byref(d2), ctypes.byref(q))
        if rc == -2:
            info = False
        else:
            N.check(rc)
            info = {name: getattr(q, name) for name, _ in N.ArtnStepInfo._fields_}
        _pair_cache.keep((d1, d2), info)
    if info is False:
        return None
    out = torch.empty(out_shape, dtype=a.dtype, device=a.device)
    with torch.cuda.device(a.device):
        e0 = e1 = None
        if profiler is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        N.check(N.lib().artn_contract2(ctypes.byref(d1), ctypes.byref(d2), a.data_ptr(), b1.data_ptr(),
                                       b2.data_ptr(), out.data_ptr(), N.current_stream_ptr(a.device)))
        if profiler is not None:
            e1.record()
            profiler.record(info, e0, e1)
    return out


def step_info(eq, a_shape, b_shape, dtype=torch.complex64, a_stride=None, b_stride=None):
    """Planner decision for one step (host only, works without a GPU)."""
    la, lb, lo = _parse(eq) if isinstance(eq, str) else (tuple(eq[0]), tuple(eq[1]), tuple(eq[2]))

    def dense(shape):
        st, s = [], 1
        for n in reversed(shape):
            st.append(s)
            s *= n
        return tuple(reversed(st))

    a_shape, b_shape = tuple(a_shape), tuple(b_shape)
    d, out_shape = _descriptor(la, lb, lo, a_shape, tuple(a_stride or dense(a_shape)), b_shape,
                               tuple(b_stride or dense(b_shape)), dtype)
    res = dict(_query(d))
    res["out_shape"] = out_shape
    res["note"] = N.lib().artn_last_plan_note().decode()
    return res


# ----------------------------------------------------------------------------------------
# dense executor
# ----------------------------------------------------------------------------------------
FUSE_MIN_NUMEL = 1 << 22  # pairs are fused only when the shared operand is at least this big
FUSE_MIN_MID = 4          # ... and the intermediate at least 1/4 of it: a first step that shrinks its tensor 8x or more leaves
                          # little traffic to save, while the fused kernel's two stages per tile cost more than they hide
                          # (n53 m14: the 5+4 pair 2^30 -> 2^27 -> 2^26 took 4.26 ms fused at 2.1 TB/s, 2.4 ms as two launches)


def fusion_schedule(scheme):
    """Execution order with candidate fused pairs.

    The big "state" tensor is operand 0 of every step that touches it (rep tensor of the
    larger child, reference contraction.py:41-46).  Two steps s < s' with the same first
    operand i, such that no step in between reads or writes tensor i, can run as one pass:
    the steps in between only build the small operand of s' from other tensors and are
    moved in front of s (they do not depend on it).  Returns a list of
    ("one", n) / ("pair", n, n') entries covering every step exactly once."""
    order, done = [], set()
    n_steps = len(scheme)
    unpaired = set()
    for n in range(n_steps):
        if n in done:
            continue
        i, j = scheme[n][0]
        partner = None
        for m in range(n + 1, n_steps if n not in unpaired else 0):
            if m in done:
                continue
            i2, j2 = scheme[m][0]
            if i2 == i:
                partner = m
                break
            if j2 == i or i2 == j or j2 == j:
                break
        if partner is not None:
            for m in range(n + 1, partner):
                if m not in done:
                    order.append(("one", m))
                    done.add(m)
            order.append(("pair", n, partner))
            done.update((n, partner))
        else:
            order.append(("one", n))
            done.add(n)
    return order


def chain_schedule(scheme, _steps=None):
    """Execution order with CHAINS: maximal runs s0 < s1 < ... of steps on the same first operand such that no step in
    between reads or writes that tensor or any of the chain's second operands (the steps in between only build later small
    operands from other tensors and are moved in front of the chain: they do not depend on it).  Returns a list of
    ("one", n) / ("chain", [s0, s1, ...]) entries (chains have two or more members) covering every step exactly once.
    A chain is what one, two or three stages per pass over HBM are cut from (_compile_dense, _plan_chain).  The steps
    moved in front of a chain are scheduled the same way among themselves (a circuit scheme's first tensor is touched by
    its first two steps and by its last one: everything in between is "moved" and holds every other chain)."""
    steps = list(range(len(scheme))) if _steps is None else _steps
    order, done = [], set()
    for pos, n in enumerate(steps):
        if n in done:
            continue
        i, j = scheme[n][0]
        chain, js, moved = [n], {j}, []
        last = pos
        while True:
            partner, between = None, []
            for q in range(last + 1, len(steps)):
                m = steps[q]
                if m in done:
                    continue
                i2, j2 = scheme[m][0]
                if i2 == i:
                    partner = q
                    break
                if j2 == i or i2 in js or j2 in js:
                    break
                between.append(m)
            if partner is None:
                break
            moved += between
            done.update(between)
            chain.append(steps[partner])
            js.add(scheme[steps[partner]][0][1])
            last = partner
        if moved:
            order += chain_schedule(scheme, moved)
        if len(chain) == 1:
            order.append(("one", n))
        else:
            order.append(("chain", chain))
        done.update(chain)
    return order


def triple_schedule(scheme, shapes, dtype=torch.complex64):
    """Host-only: how _compile_dense would cut the chains of a dense scheme -- a list of ("one", n) / ("pair", n, n') /
    ("triple", n, n', n'') in execution order (the small-step program is ignored: every step appears)."""
    out = []
    shapes = dict(shapes)
    for entry in chain_schedule(scheme):
        members = [entry[1]] if entry[0] == "one" else entry[1]
        for g in _cut_chain(scheme, members, shapes, dtype, set()):
            out.append((("one", "pair", "triple")[len(g[0]) - 1],) + tuple(g[0]))
    return out


def _cut_chain(scheme, members, shapes, dtype, skip):
    """Cut one chain into groups of one, two or three consecutive steps, minimising the bytes the chain's tensor moves
    through HBM (elements in + elements out per group; dynamic programme over the planner's host-only answers).
    `shapes` is advanced past the chain.  Returns [(steps, planner info or None, descriptors or None, out_shape)];
    members in `skip` (taken by the small-step program) are dropped."""
    members = [n for n in members if n not in skip]
    if not members:
        return []
    i = scheme[members[0]][0][0]
    # shapes of the chain's tensor before / after every member
    seq = [tuple(shapes[i])]
    for n in members:
        (_, j), eq = scheme[n][0], scheme[n][1]
        la, lb, lo = _labels(eq)
        ext = dict(zip(la, seq[-1]))
        ext.update(zip(lb, shapes[j]))
        seq.append(tuple(ext[x] for x in lo))

    def numel(shape):
        r = 1
        for e in shape:
            r *= e
        return r
    fuse_ok = dtype in _DTYPES
    fuse3_ok = dtype == torch.complex64 and precision.current() in (None, "fp32") and N.has("artn_contract3_query")
    L = len(members)
    cand = {}   # (p, g) -> (info, descriptors, out_shape)

    def desc(p, shape_in):
        (_, j), eq = scheme[members[p]][0], scheme[members[p]][1]
        la, lb, lo = _labels(eq)
        if len(la) != len(shape_in):
            return None, None
        return _descriptor(la, lb, lo, shape_in, _dense_strides(shape_in), shapes[j], _dense_strides(shapes[j]), dtype)
    for p in range(L):
        if not fuse_ok or numel(seq[p]) < FUSE_MIN_NUMEL:
            continue
        if p + 1 < L and numel(seq[p + 1]) * FUSE_MIN_MID >= numel(seq[p]):
            d1, mid = desc(p, seq[p])
            d2, out = desc(p + 1, mid) if d1 is not None else (None, None)
            if d2 is not None:
                q = N.ArtnStepInfo()
                rc = N.lib().artn_contract2_query(ctypes.byref(d1), ctypes.byref(d2), ctypes.byref(q))
                if rc == 0:
                    cand[(p, 2)] = ({name: getattr(q, name) for name, _ in N.ArtnStepInfo._fields_}, (d1, d2), out)
                elif rc != -2:
                    N.check(rc)
                if fuse3_ok and p + 2 < L:
                    d3, out3 = desc(p + 2, out)
                    if d3 is not None:
                        q = N.ArtnStepInfo()
                        rc = N.lib().artn_contract3_query(ctypes.byref(d1), ctypes.byref(d2), ctypes.byref(d3), ctypes.byref(q))
                        if rc == 0:
                            cand[(p, 3)] = ({name: getattr(q, name) for name, _ in N.ArtnStepInfo._fields_}, (d1, d2, d3), out3)
                        elif rc != -2:
                            N.check(rc)
    # best[p]: least elements moved for members p..; at equal bytes a pair beats a single step (fewer launches) and a
    # triple must SAVE bytes to be taken (a tiny surcharge: the pair kernels are the measured ones; n30 m14's two fitting
    # triples sit at an odd distance, so they would only trade a pair for a triple and a single step)
    best = [0.0] * (L + 1)
    take = [1] * L
    for p in range(L - 1, -1, -1):
        best[p], take[p] = numel(seq[p]) + numel(seq[p + 1]) + best[p + 1], 1
        for g in (2, 3):
            if (p, g) in cand:
                c = numel(seq[p]) + numel(seq[p + g]) + best[p + g] + (1.0 if g == 3 else 0.0)
                if c <= best[p]:
                    best[p], take[p] = c, g
    # the cut of rounds 1-3 -- pairs from the left, as fusion_schedule forms them -- stays unless the dynamic programme finds a
    # triple that saves bytes: re-pairing alone is not a reason to leave the measured launch lists (it cost the random
    # D = 2 network 8 %: fewer launches, but pairs whose first stage is re-run for every value of an outer result bit)
    p, uses_triple = 0, False
    while p < L:
        uses_triple = uses_triple or take[p] == 3
        p += take[p]
    if not uses_triple:
        p = 0
        while p < L:
            take[p] = 2 if (p, 2) in cand else 1
            p += take[p]
    groups, p = [], 0
    while p < L:
        g = take[p]
        if g == 1:
            groups.append(((members[p],), None, None, seq[p + 1]))
        else:
            info, ds, out = cand[(p, g)]
            groups.append((tuple(members[p:p + g]), info, ds, out))
        p += g
    shapes[i] = seq[-1]
    return groups


_plan_cache = _Bounded(64)
_schedule_cache = _IdMemo(64)    # scheme -> its chain / fusion schedule


class _Op:
    """One launch of a compiled dense scheme: a single step, a fused pair or a fused triple."""
    __slots__ = ("steps", "i", "j", "j2", "j3", "d1", "d2", "d3", "out_shape", "info", "sum_rows", "acc")
    # (acc: False once artn_contract[2]_acc has declined this launch -- see tensor_contraction(accumulate_into=...))



# ----------------------------------------------------------------------------------------
# the launch-latency tail of a dense scheme as ONE launch (artn_program_*)
# ----------------------------------------------------------------------------------------
PROGRAM_MAX_NUMEL = 1 << 14   # operands and result of a step that goes into the small-step program
PROGRAM_MIN_STEPS = 4         # not worth a program below this many steps
KERNEL_PROGRAM = 3            # profiler label of the program launch (not a planner kernel id)


class _Program:
    """Compiled small steps of a dense scheme: device image per device, workspace layout, which tensor
    ids it reads from the caller (`ext_ids`, in kernel-argument order) and which results it leaves
    (`outputs`: id -> (workspace byte offset, shape))."""
    __slots__ = ("host_image", "host_groups", "n_groups", "n_steps", "ext_ids", "ws_bytes", "outputs", "step_out", "dev", "flops", "ext_array",
                 "dtype")

    def device_copy(self, device):
        hit = self.dev.get(device)
        if hit is None:
            hit = self.dev[device] = self.host_image.to(device)
        return hit


def _plan_small_program(scheme, shapes, dtype):
    """Split a dense scheme into the steps that only combine small, leaf-derived tensors -- compiled into
    a one-launch program -- and the rest, in scheme order.  Returns (program or None, indices of the
    remaining steps).  Sequential semantics are kept: a step joins the program only if neither operand
    was produced by a remaining step and its target is not still to be read by an earlier remaining step."""
    every = list(range(len(scheme)))
    if dtype not in _DTYPES or len(scheme) < PROGRAM_MIN_STEPS or __import__("os").environ.get("ARTN_NO_PROGRAM", "0") not in ("", "0"):
        return None, every
    cur = dict(shapes)
    tainted, main_reads, small, main = set(), set(), [], []

    def numel(sh):
        n = 1
        for e in sh:
            n *= e
        return n

    recs = {}
    for n, step in enumerate(scheme):
        (i, j), eq = step[0], step[1]
        la, lb, lo = _labels(eq)
        ok = (_is_plain_step(step) and i not in tainted and j not in tainted and i not in main_reads and i in cur and j in cur
              and i != j and len(la) == len(cur[i]) and len(lb) == len(cur[j]))
        out_shape = None
        if ok:
            ext = dict(zip(la, cur[i]))
            ext.update(zip(lb, cur[j]))
            out_shape = tuple(ext[x] for x in lo)
            ok = 0 < max(numel(cur[i]), numel(cur[j]), numel(out_shape)) <= PROGRAM_MAX_NUMEL and numel(out_shape) > 0
        if ok:
            recs[n] = (la, lb, lo, cur[i], cur[j], out_shape)
            small.append(n)
            cur[i] = out_shape
        else:
            main.append(n)
            tainted.add(i)
            main_reads.add(j)
            cur.pop(i, None)
    if len(small) < PROGRAM_MIN_STEPS:
        return None, every
    needed = set()
    for n in main:
        needed.update(scheme[n][0])
    needed.add(scheme[-1][0][0])
    prog = _build_program(scheme, small, recs, needed, dtype)
    if prog is None:
        return None, every
    return prog, main


def _build_program(scheme, small, recs, needed=None, dtype=torch.complex64):
    """Compile the steps `small` (indices into `scheme`, in execution order; recs[n] = (la, lb, lo, shape_i, shape_j,
    out_shape)) into a one-launch program image.  needed: tensor ids whose LAST version must be in the workspace after
    the launch (everything else may live and die in LDS); None: the result of EVERY step is kept (prog.step_out[n] =
    (workspace byte offset, shape)) -- what the slice loop needs, which keeps each step's result for later slices.
    dtype: complex64 (small matrix-core steps included) or complex128 (artn_k_program<double>: vector ALU only).
    Returns a _Program or None (the caller then runs the steps one by one)."""
    esz = 16 if dtype == torch.complex128 else 8
    def numel(sh):
        n = 1
        for e in sh:
            n *= e
        return n

    # groups = connected components of the small steps (steps of different groups share no tensor)
    parent = {}

    def find(x):
        while parent.setdefault(x, x) != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for n in small:
        i, j = scheme[n][0]
        parent[find(i)] = find(j)
    seen = {}
    for n in small:
        r = find(scheme[n][0][0])
        if r not in seen:
            seen[r] = len(seen)
    small_sorted = sorted(small, key=lambda n: (seen[find(scheme[n][0][0])], n))
    group_start = [0]
    for k in range(1, len(small_sorted)):
        if find(scheme[small_sorted[k]][0][0]) != find(scheme[small_sorted[k - 1]][0][0]):
            group_start.append(k)
    group_start.append(len(small_sorted))
    # locations: tensors the steps read but do not produce are external pointers, results live in the workspace
    loc, ext_ids, ws = {}, [], 0
    la_, lb_, lc_, descs, flops = [], [], [], [], 0.0
    where, step_out = {}, {}
    for n in small_sorted:
        i, j = scheme[n][0]
        la, lb, lo, sa, sb, so = recs[n]
        for t in (i, j):
            if t not in loc:
                loc[t] = -(len(ext_ids) + 1)
                ext_ids.append(t)
        d, _ = _descriptor(la, lb, lo, sa, _dense_strides(sa), sb, _dense_strides(sb), dtype)
        descs.append(d)
        la_.append(loc[i])
        lb_.append(loc[j])
        lc_.append(ws)
        loc[i] = ws
        where[i] = (ws, so)
        step_out[n] = (ws, so)
        ws += (numel(so) * esz + 15) // 16 * 16
        f = 8.0
        for x in dict.fromkeys(la + lb):
            f *= dict(zip(la, sa)).get(x) or dict(zip(lb, sb))[x]
        flops += f
    if len(ext_ids) > N.ARTN_PROGRAM_MAX_EXT:
        return None
    if needed is None:
        keep = [1] * len(small_sorted)
    else:
        # results a remaining step reads, or the scheme's own result: those must be in the workspace after the
        # launch (the last version of the id); everything else may live and die in the workgroup's LDS
        last_writer = {}
        for k, n in enumerate(small_sorted):
            last_writer[scheme[n][0][0]] = k
        keep = [1 if last_writer[scheme[n][0][0]] == k and scheme[n][0][0] in needed else 0 for k, n in enumerate(small_sorted)]
    lib = N.lib()
    arr = (ctypes.POINTER(N.ArtnStepDesc) * len(descs))(*[ctypes.pointer(d) for d in descs])
    i64 = lambda v: (ctypes.c_int64 * len(v))(*v)
    n_groups = len(group_start) - 1
    image_bytes = int(lib.artn_program_image_bytes(len(descs), arr, n_groups))
    if image_bytes < 0:
        return None
    image = torch.zeros(image_bytes, dtype=torch.uint8)
    rc = lib.artn_program_build(len(descs), arr, i64(la_), i64(lb_), i64(lc_), (ctypes.c_uint8 * len(keep))(*keep), n_groups,
                                (ctypes.c_int32 * len(group_start))(*group_start), image.data_ptr(), image_bytes)
    if rc != 0:   # no program: every step runs as its own launch (what a scheme did before programs existed)
        return None
    prog = _Program()
    prog.host_image, prog.host_groups = image, torch.tensor(group_start, dtype=torch.int32)
    prog.n_groups, prog.n_steps, prog.ext_ids, prog.ws_bytes = n_groups, len(small_sorted), ext_ids, max(ws, 16)
    prog.outputs = {t: v for t, v in where.items() if needed is None or t in needed}
    prog.step_out = step_out
    prog.dev, prog.flops, prog.ext_array = {}, flops, None
    prog.dtype = dtype
    return prog


def _run_program(prog, tensors, dtype, device, stream):
    image = prog.device_copy(device)
    ws = torch.empty(prog.ws_bytes, dtype=torch.uint8, device=device)
    ext = (ctypes.c_void_p * len(prog.ext_ids))()   # per call: the plan is shared by every thread and device that hits the cache
    for q, t in enumerate(prog.ext_ids):
        ext[q] = tensors[t].data_ptr()
    if profiler is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    N.check(N.lib().artn_program_run(image.data_ptr(), prog.n_groups, ext, len(prog.ext_ids), ws.data_ptr(), _DTYPES[prog.dtype], stream))
    if profiler is not None:
        e1.record()
        profiler.record({"kernel": KERNEL_PROGRAM, "flops": prog.flops, "bytes": 0.0, "k_bits": 0, "k2_bits": 0, "m_tile_bits": 0,
                         "n_tile_bits": 0, "tile_in_bits": 0, "tile_out_bits": 0, "n_tiles": prog.n_steps, "a_rereads": 1}, e0, e1)
    for t, (off, shape) in prog.outputs.items():
        tensors[t] = _ws_view(ws, off, shape, prog.dtype)
    return ws


def _ws_view(ws, off, shape, dtype=torch.complex64):
    n = 16 if dtype == torch.complex128 else 8
    for e in shape:
        n *= e
    return ws[off:off + n].view(dtype).reshape(shape)


def _is_plain_step(step):
    """A dense 2-tuple, or a sparse 3-tuple that is not chunked (branch D of the sparse executor, reference
    contraction.py:189-191: its index lists are not looked at): one einsum."""
    return len(step) == 2 or (len(step) == 3 and len(step[2][0]) <= 1)


def _compile_dense(scheme, shapes, dtype):
    """Resolve a dense scheme once: the small-step program, execution order of the rest, fused pairs
    (host-only planner queries), descriptors and result shapes.  Everything the per-call loop needs
    except pointers.  Returns (program or None, launch list)."""
    shapes = dict(shapes)
    fuse_ok = dtype in _DTYPES   # (complex128 pairs: artn_k_bits128)
    ops = []
    prog, main_idx = _plan_small_program(scheme, shapes, dtype)
    in_prog = set(range(len(scheme))) - set(main_idx)
    if prog is not None:
        for t, (off, shape) in prog.outputs.items():
            shapes[t] = shape
    scheme = _own_layouts(scheme, main_idx, shapes, dtype)

    def emit(n, i, j, la, lb, lo, sa, sb, warn=True):
        op = _Op()
        op.steps, op.i, op.j, op.j2, op.d2, op.j3, op.d3 = (n,), i, j, None, None, None, None
        op.d1, op.out_shape = _descriptor(la, lb, lo, sa, _dense_strides(sa), sb, _dense_strides(sb), dtype)
        op.info = None
        op.sum_rows = 0
        ops.append(op)
        numel = 1
        for e in (sa if len(sa) >= len(op.out_shape) else op.out_shape):
            numel *= e
        if warn:
            _warn_if_generic(op.d1, numel, f"tensor_contraction step {n}")
        return op.out_shape

    def single(n):
        (i, j), eq = scheme[n][0], scheme[n][1]
        la, lb, lo = _labels(eq)
        outer = _big_k_outer(la, lb, lo, shapes[i], None, shapes[j], None, dtype)
        if outer:
            # more contracted bits than one LDS tile holds (big x big steps of random networks):
            # split-K through a temporary batch label, then sum it out (see _split_big_k)
            mid = tuple(outer) + tuple(lo)
            mid_shape = emit(n, i, j, la, lb, mid, shapes[i], shapes[j])
            shapes[i] = emit(n, i, _ONE, mid, (), lo, mid_shape, (), warn=False)   # runs as artn_sum_axis_c64
            rows = 1
            for e in mid_shape[:len(outer)]:
                rows *= e
            ops[-1].sum_rows = rows   # the temporary label leads `mid`: a column sum (artn_sum_axis_c64)
        else:
            shapes[i] = emit(n, i, j, la, lb, lo, shapes[i], shapes[j])

    def pair(n, m):
        """steps n and m as one fused launch, if the planner takes them; False: nothing emitted"""
        (i, j), eq1 = scheme[n][0], scheme[n][1]
        (_, j2), eq2 = scheme[m][0], scheme[m][1]
        numel = 1
        for e in shapes[i]:
            numel *= e
        info = None
        if fuse_ok and numel >= FUSE_MIN_NUMEL:
            la1, lb1, lo1 = _labels(eq1)
            la2, lb2, lo2 = _labels(eq2)
            d1, mid = _descriptor(la1, lb1, lo1, shapes[i], _dense_strides(shapes[i]), shapes[j],
                                  _dense_strides(shapes[j]), dtype)
            mid_numel = 1
            for e in mid:
                mid_numel *= e
            if len(la2) == len(mid) and mid_numel * FUSE_MIN_MID >= numel:
                d2, out_shape = _descriptor(la2, lb2, lo2, mid, _dense_strides(mid), shapes[j2],
                                            _dense_strides(shapes[j2]), dtype)
                q = N.ArtnStepInfo()
                rc = N.lib().artn_contract2_query(ctypes.byref(d1), ctypes.byref(d2), ctypes.byref(q))
                if rc == 0:
                    info = {name: getattr(q, name) for name, _ in N.ArtnStepInfo._fields_}
                elif rc != -2:
                    N.check(rc)
        if info is None:
            return False
        op = _Op()
        op.steps, op.i, op.j, op.j2, op.d1, op.d2, op.out_shape, op.info = (n, m), i, j, j2, d1, d2, out_shape, info
        op.j3 = op.d3 = None
        op.sum_rows = 0
        shapes[i] = out_shape
        ops.append(op)
        return True

    # Three-step fusion (artn_contract3) changes the launch list only where a triple SAVES bytes (section 4.1c of DESIGN.md: on
    # no committed workload); everywhere else the pairing of rounds 1-3 -- fusion_schedule, pairs from the left -- is kept
    # exactly: re-pairing alone cost the random D = 2 network 8 % (pairs the old schedule never formed).
    use_chains = False
    if (dtype == torch.complex64 and precision.current() in (None, "fp32") and not _os_environ.get("ARTN_NO_FUSE")
            and N.has("artn_contract3_query")):   # (development builds only: the product library has no triples)
        probe = dict(shapes)
        for entry in chain_schedule(scheme):
            if entry[0] == "chain":
                if any(len(g[0]) == 3 for g in _cut_chain(scheme, entry[1], probe, dtype, in_prog)):
                    use_chains = True
                    break
            elif entry[1] not in in_prog:
                (ci, cj), ceq = scheme[entry[1]][0], scheme[entry[1]][1]
                cla, clb, clo = _labels(ceq)
                cext = dict(zip(cla, probe[ci]))
                cext.update(zip(clb, probe[cj]))
                probe[ci] = tuple(cext[x] for x in clo)
    if use_chains:
        # (the chains are found on the whole scheme, as if there were no program: steps the program has taken are simply
        #  skipped -- they ran before everything else)
        for entry in chain_schedule(scheme):
            if entry[0] == "one":
                if entry[1] not in in_prog:
                    single(entry[1])
                continue
            i = scheme[entry[1][0]][0][0]
            before = dict(shapes)
            for steps, info, ds, out_shape in _cut_chain(scheme, entry[1], shapes, dtype, in_prog):
                if len(steps) == 1:
                    shapes[i] = before[i]
                    single(steps[0])
                    before[i] = shapes[i]
                    continue
                op = _Op()
                op.steps, op.i, op.out_shape, op.info, op.sum_rows = steps, i, out_shape, info, 0
                op.j, op.j2 = scheme[steps[0]][0][1], scheme[steps[1]][0][1]
                op.j3 = scheme[steps[2]][0][1] if len(steps) == 3 else None
                op.d1, op.d2 = ds[0], ds[1]
                op.d3 = ds[2] if len(steps) == 3 else None
                before[i] = out_shape
                ops.append(op)
            shapes[i] = before[i]
        return prog, ops
    # (the pairing is decided on the whole scheme, as if there were no program: steps the program has
    #  taken are simply skipped -- they ran before everything else)
    if _chain_plan_on():
        # every chain of steps on one tensor is cut into single steps and fused pairs by the priced dynamic programme of
        # the sparse executor (_cut_sparse_chain); pairs from the left (fusion_schedule, below) with ARTN_CHAIN_PLAN=0
        schedule = []
        for entry in chain_schedule(scheme):
            if entry[0] == "one":
                schedule.append(entry)
                continue
            members = [n for n in entry[1] if n not in in_prog]
            if len(members) < 2:
                schedule += [("one", n) for n in members]
                continue
            schedule.append(("cut", members))
    else:
        schedule = fusion_schedule(scheme)
    for entry in schedule:
        if entry[0] == "one":
            if entry[1] not in in_prog:
                single(entry[1])
            continue
        if entry[0] == "cut":
            members = entry[1]
            ci = scheme[members[0]][0][0]
            groups = _cut_sparse_chain(scheme, members, tuple(shapes[ci]), [tuple(shapes[scheme[n][0][1]]) for n in members], dtype)
            covered = sum(len(g) for g in groups)
            groups += [(n,) for n in members[covered:]]
            for g in groups:
                if len(g) == 1 or not pair(g[0], g[1]):
                    for n in g:
                        single(n)
            continue
        n, m = entry[1], entry[2]
        if n in in_prog or m in in_prog:
            for q in (n, m):
                if q not in in_prog:
                    single(q)
            continue
        if not pair(n, m):
            single(n)
            single(m)
    return prog, ops


def _own_layouts(scheme, main_idx, shapes, dtype):
    """Label order of the INTERMEDIATES of a dense scheme whose labels have extents that are not powers of two.

    The reference's compiler orders every result's labels as `list(set)` left them (contraction.py:49): for the bit
    kernels that is a permutation of address bits inside a tile, but with odd extents nothing is aligned and every
    label that separates two neighbours of the flattened row index costs a factor of its extent in run length.
    tensor_contraction only promises the label order of the scheme's RESULT (fused pairs never materialise their
    intermediate either), so the steps that run as launches get
        result labels = batch labels, then the free labels of the operand with fewer free values (in that operand's
                        order), then the free labels of the other operand (in its order, fastest)
    -- consecutive rows of the extent GEMM's 128-row tiles are then consecutive in the first operand's free labels
    AND in the result (1 KiB contiguous per tile column), whatever the planner's `list(set)` said.  The last step
    keeps the scheme's order, and so does every step none of whose labels has an odd extent (a mixed network's
    power-of-two steps stay what the bit kernels' planner was tuned on).  Returns the scheme itself or a rewritten copy (label-tuple equations)."""
    if dtype not in _DTYPES or not main_idx or _os_environ.get("ARTN_OWN_LAYOUTS", "1") in ("0",):
        return scheme
    if all(e & (e - 1) == 0 for sh in shapes.values() for e in sh):
        return scheme
    perm = {}      # tensor id -> current dim order of the tensor as positions of the scheme's order
    out = list(scheme)
    shapes = dict(shapes)
    last = len(scheme) - 1
    for n in main_idx:
        step = scheme[n]
        if not _is_plain_step(step):
            return scheme
        (i, j), eq = step[0], step[1]
        la, lb, lo = _labels(eq)
        pa, pb = perm.get(i), perm.get(j)
        la_act = tuple(la[p] for p in pa) if pa else tuple(la)
        lb_act = tuple(lb[p] for p in pb) if pb else tuple(lb)
        ext = {}
        if i in shapes:
            ext.update(zip(la_act, shapes[i]))
        if j in shapes:
            ext.update(zip(lb_act, shapes[j]))
        pow2_step = all(e & (e - 1) == 0 for e in ext.values())
        if n == last or len(set(lo)) != len(lo) or pow2_step:
            # (a step whose labels are all powers of two runs on the bit kernels: its pair eligibility and its plan are
            #  tuned to the reference's orders -- only its operands' ACTUAL orders are passed on)
            lo_act = tuple(lo)
        else:
            in_a, in_b = set(la), set(lb)
            h = [x for x in la_act if x in in_b and x in lo]
            m = [x for x in la_act if x not in in_b and x in lo]
            nn = [x for x in lb_act if x not in in_a and x in lo]
            rest = [x for x in lo if x not in h and x not in m and x not in nn]
            size = lambda v: __import__("math").prod(ext.get(x, 1) for x in v)
            small, big = (nn, m) if size(m) >= size(nn) else (m, nn)
            lo_act = tuple(rest + h + small + big)
        if any(x not in ext for x in lo_act):
            return scheme   # (shapes unknown: leave the scheme alone)
        perm[i] = tuple(lo.index(x) for x in lo_act) if lo_act != tuple(lo) else None
        out[n] = ((i, j), (la_act, lb_act, lo_act))
        shapes[i] = tuple(ext[x] for x in lo_act)
    return out


_scheme_ids = _Bounded(256)   # id(scheme) -> (scheme, ids it reads, first-use order)
_os_environ = __import__("os").environ
_NULL_CTX = __import__("contextlib").nullcontext()
_pair_query = []


def _HAS_PAIR_QUERY():
    if not _pair_query:
        _pair_query.append(N.lib().artn_contract2_query is not None)
    return _pair_query[0]


def _axpy(acc, x):
    """acc += x (artn_axpy_c64 / _c128) on acc's device and the current stream"""
    x = x.contiguous()
    axpy = N.lib().artn_axpy_c64 if acc.dtype == torch.complex64 else N.lib().artn_axpy_c128
    with torch.cuda.device(acc.device):
        N.check(axpy(acc.data_ptr(), x.data_ptr(), acc.numel(), N.current_stream_ptr(acc.device)))


def tensor_contraction(tensors, scheme, accumulate_into=None):
    """Run a dense scheme: for each ((i, j), eq): tensors[i] <- contract(eq, tensors[i],
    tensors[j]); returns the last tensors[i] (reference contraction.py:62-76; `tensors` is
    mutated the same way).

    accumulate_into: a contiguous tensor of the result's dtype and number of elements (the slice loop's
    `collect_tensor`, reference simulation.py:114): the result is ADDED to it and it is returned.  When the last launch of
    the scheme is a state-streaming step or pair the add happens in that launch's store phase (artn_contract_acc /
    artn_contract2_acc: the slice's result never exists in memory); otherwise the scheme runs as usual and the result is
    added with artn_axpy_c64.  With accumulate_into the entry of `tensors` that the last step rebinds is NOT the slice's
    result (none exists): it aliases the accumulator -- the running sum -- and must not be read as the reference's
    `tensors[i]`.

    Behind the unchanged entry point the scheme is compiled once (per scheme object and leaf
    shapes) into a launch list; two consecutive steps on the same big tensor execute as ONE
    pass over HBM when their contracted bits fit one LDS tile (artn_contract2).  Results are
    those of the step-by-step order."""
    if len(scheme) == 0:
        raise RuntimeError("empty contraction scheme")
    # (host time matters for the launch-latency workloads -- n12 is ONE 90 us launch: the ids a scheme reads are
    #  resolved once per scheme object, the per-tensor checks are inlined, shapes are hashed as torch.Size)
    su = _scheme_ids.get(id(scheme))
    if su is None or su[0] is not scheme or not _same_steps(su[2], scheme):
        seen = {}
        for step in scheme:
            for k in step[0]:
                seen.setdefault(k, None)
        su = _scheme_ids[id(scheme)] = (scheme, tuple(seen), tuple(scheme))
    is_dict = isinstance(tensors, dict)
    n_list = 0 if is_dict else len(tensors)
    first = None
    dtype = device = None
    shape_key = []
    for k in su[1]:
        if is_dict:
            if k not in tensors:
                continue   # (reported below, by the compiler, as the reference's KeyError would be)
        elif not (isinstance(k, int) and 0 <= k < n_list):
            continue
        t = tensors[k]
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            N.require_gpu(t, "tensor_contraction")
        if first is None:
            first, dtype, device = t, t.dtype, t.device
        elif t.dtype != dtype or t.device != device:
            raise RuntimeError("all tensors of a scheme must share dtype and device")
        if not t.is_contiguous():
            tensors[k] = t = t.contiguous()
        shape_key.append(t.shape)
    if first is None or dtype not in _DTYPES:
        raise RuntimeError("tensor_contraction needs complex64 or complex128 GPU tensors")
    key = (id(scheme), dtype, precision.current(), tuple(shape_key), _HAS_PAIR_QUERY(), bool(_os_environ.get("ARTN_NO_FUSE")), _chain_plan_on(),
           _os_environ.get("ARTN_OWN_LAYOUTS", "1"))
    hit = _plan_cache.get(key)
    if hit is None or hit[0] is not scheme or hit[3] is not su[2]:   # (su[2]: the step snapshot just verified above)
        shapes = {k: tuple(tensors[k].shape) for k in su[1] if (k in tensors if is_dict else isinstance(k, int) and 0 <= k < n_list)}
        try:
            prog, ops = _compile_dense(scheme, shapes, first.dtype)
        except KeyError as e:
            raise RuntimeError(f"scheme refers to tensor id {e} that was not supplied") from e
        hit = _plan_cache[key] = (scheme, ops, prog, su[2])
    ops, prog = hit[1], hit[2]
    lib = N.lib()
    byref = ctypes.byref
    with (_NULL_CTX if torch.cuda.current_device() == device.index else torch.cuda.device(device)):
        stream = N.current_stream_ptr(device)
        if prog is not None:   # every step that only combines small leaf-derived tensors: one launch
            _run_program(prog, tensors, dtype, device, stream)
        last_op = ops[-1] if ops else None
        done_acc = False
        acc_ok = False
        if accumulate_into is not None:
            if not (isinstance(accumulate_into, torch.Tensor) and accumulate_into.is_cuda and accumulate_into.device == device
                    and accumulate_into.is_contiguous() and accumulate_into.dtype == dtype):
                raise RuntimeError("accumulate_into must be a contiguous tensor of the scheme's dtype on its device")
            acc_ok = (last_op is not None and (dtype == torch.complex128 or (dtype == torch.complex64 and precision.current() in (None, "fp32")))
                      and last_op.steps and last_op.steps[-1] == len(scheme) - 1
                      and int(np.prod(last_op.out_shape, dtype=np.int64)) == accumulate_into.numel())
        for op in ops:
            a = tensors[op.i]
            if op.sum_rows and _sum_leading_ok(a, op.sum_rows):
                tensors[op.i] = sum_leading(a, op.sum_rows).reshape(op.out_shape)
                continue
            b = _one_scalar(dtype, device) if op.j is _ONE else tensors[op.j]
            fused_acc = False
            if op is last_op and acc_ok and op.d3 is None and getattr(op, "acc", None) is None:
                # (decided once per compiled op: a last step that would run packed -- it wants a workspace the accumulating
                #  entry point does not take -- or has an empty result keeps the separate add)
                kern = (op.info or _step_info_cached(op.d1))["kernel"] if op.d2 is None else N.KERNEL_BITS_MFMA
                op.acc = kern == N.KERNEL_BITS_MFMA and accumulate_into.numel() > 0
            if op is last_op and acc_ok and op.d3 is None and getattr(op, "acc", None) is True:
                # the slice loop's `collect += result` in the store phase of the last launch
                if profiler is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                if op.d2 is None:
                    rc = lib.artn_contract_acc(byref(op.d1), a.data_ptr(), b.data_ptr(), accumulate_into.data_ptr(), stream)
                else:
                    rc = lib.artn_contract2_acc(byref(op.d1), byref(op.d2), a.data_ptr(), b.data_ptr(), tensors[op.j2].data_ptr(),
                                                accumulate_into.data_ptr(), stream)
                if rc == -2:
                    op.acc = False   # (this plan's store phase cannot add: remembered per compiled op)
                else:
                    fused_acc = True
                    out = accumulate_into
            if fused_acc:
                pass
            else:
                out = torch.empty(op.out_shape, dtype=dtype, device=device)
                if profiler is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
            if fused_acc:
                pass
            elif op.d2 is None:
                rc = _launch_step(op.d1, a, b, out, stream) if out.numel() else 0
            elif op.d3 is None:
                b2 = tensors[op.j2]
                rc = lib.artn_contract2(byref(op.d1), byref(op.d2), a.data_ptr(), b.data_ptr(), b2.data_ptr(),
                                        out.data_ptr(), stream)
            else:
                b2, b3 = tensors[op.j2], tensors[op.j3]
                rc = lib.artn_contract3(byref(op.d1), byref(op.d2), byref(op.d3), a.data_ptr(), b.data_ptr(), b2.data_ptr(),
                                        b3.data_ptr(), out.data_ptr(), stream)
            if rc != 0:
                msg = lib.artn_last_error()
                raise RuntimeError(f"tensor_contraction failed at step(s) {op.steps} "
                                   f"{[scheme[n][1] for n in op.steps]}: {msg.decode() if msg else rc}")
            if profiler is not None:
                e1.record()
                if op.info is None:
                    op.info = _query(op.d1)
                info = op.info
                if fused_acc:   # the launch also reads the accumulator
                    info = dict(info)
                    info["bytes"] = info["bytes"] + float(out.numel() * out.element_size())
                profiler.record(info, e0, e1)
            tensors[op.i] = out
            done_acc = done_acc or fused_acc
    res = tensors[scheme[-1][0][0]]
    if accumulate_into is None:
        return res
    if not done_acc:
        if res.numel() != accumulate_into.numel() or res.dtype != accumulate_into.dtype:
            raise RuntimeError("accumulate_into does not match the result")
        _axpy(accumulate_into, res.reshape(accumulate_into.shape))
    return accumulate_into


# ----------------------------------------------------------------------------------------
# sparse-state executor
# ----------------------------------------------------------------------------------------
_index_cache = _IdMemo(4096)    # (index tensor; device, rows of the operand) -> device copy
# Out-of-range bookkeeping of the gather kernels is per THREAD (like `precision` and the deferred check of the slice
# loop): a thread reads and clears only the flags its own launches may have set.
_flag_state = threading.local()


def _flag_cache():
    """device -> sticky int32 out-of-range flag written by this thread's gather launches"""
    c = getattr(_flag_state, "cache", None)
    if c is None:
        c = _flag_state.cache = {}
    return c


def _flags_used():
    """devices whose flag some launch of this thread since its last check may have set"""
    u = getattr(_flag_state, "used", None)
    if u is None:
        u = _flag_state.used = set()
    return u


def _device_index(idx, device, src_rows=None):
    """int64 row indices of a scheme step (CPU tensors built at reference
    contraction.py:249-283) cached on the device: the scheme is reused for every slice.

    With `src_rows` the indices are validated ONCE on the host against the operand they select
    from, with the reference's semantics (`tensors[i][idx]`, contraction.py:149-150, :177-178,
    :187): an index outside [-src_rows, src_rows) raises (the reference dies with IndexError there,
    contraction.py:192-195), a negative one counts from the end."""
    extra = (str(device), src_rows)
    dev = _index_cache.find((idx,), extra)
    if dev is not _MISS:
        return dev
    host = torch.as_tensor(idx, dtype=torch.int64).reshape(-1).cpu()
    if src_rows is not None and host.numel():
        lo, hi = int(host.min()), int(host.max())
        if lo < -src_rows or hi >= src_rows:
            raise RuntimeError(f"row index out of range: indices span [{lo}, {hi}] but the operand has {src_rows} rows "
                               "(IndexError in the reference, contraction.py:192-195)")
        if lo < 0:
            host = torch.where(host < 0, host + src_rows, host)
    return _index_cache.keep((idx,), host.to(device).contiguous(), extra)


def _flag(device):
    cache = _flag_cache()
    flag = cache.get(device)
    if flag is None:
        flag = cache[device] = torch.zeros(1, dtype=torch.int32, device=device)
    return flag


def check_gather_flag(what="gather"):
    """Read (one sync per device) and clear the sticky out-of-range flag of the gather kernels; raise
    if any launch since the last check saw a row index outside its operand.  Host validation in
    _device_index catches bad schemes before launch; this is the device's own word, read once per
    scheme by the executors rather than after every launch."""
    bad = []
    used, cache = _flags_used(), _flag_cache()
    for device in list(used):
        flag = cache[device]
        if int(flag.item()) != 0:
            flag.zero_()
            bad.append(str(device))
    used.clear()
    if bad:
        raise RuntimeError(f"{what}: a row gather read an index outside its operand on {', '.join(bad)} "
                           "(the reference raises IndexError, contraction.py:192-195)")


_single_row_cache = _IdMemo(4096)   # index tensor of one entry -> that entry


def _single_row(idx, t):
    """the row an index list of ONE entry selects from t (reference semantics: `t[idx]`, negative counts from the end,
    out of range raises as contraction.py:192-195 would die), or None when it selects several rows or t has no rows"""
    return _single_row_of_shape(idx, tuple(t.shape), t.element_size())


def _single_row_of_shape(idx, shape, itemsize):
    """_single_row for a tensor known by its shape only (the chain planner of the sparse executor)"""
    # (a 0-d index would DROP the dimension in the reference's t[idx]: only 1-d index lists of one entry are row selects)
    if not isinstance(idx, torch.Tensor) or idx.dim() != 1 or idx.numel() != 1 or len(shape) == 0:
        return None
    rows = shape[0]
    # the value is read ONCE per index tensor (a device-resident index would cost a host synchronisation per step and per
    # slice here, and int() of it is illegal during HIP-graph capture): cached like _is_identity / _device_index
    v = _single_row_cache.find((idx,))
    if v is _MISS:
        v = _single_row_cache.keep((idx,), int(idx.reshape(-1)[0]))
    if v < -rows or v >= rows:
        raise RuntimeError(f"row index out of range: {v} for {rows} rows")
    # (a row of one element would be an 8-byte view: the tiled kernels want 16-byte aligned operands)
    row_numel = 1
    for e in shape[1:]:
        row_numel *= e
    if rows == 0 or row_numel * itemsize % 16 != 0:
        return None
    return v % rows


_identity_cache = _IdMemo(4096)   # index tensor -> its length if it is 0, 1, ..., n - 1, else -1


def _is_identity(idx, rows):
    """True when `idx` is 0, 1, ..., rows-1: the gather would copy the tensor onto itself (every
    row select of a single-bitstring scheme is of this kind; checked once per index tensor)."""
    ident = _identity_cache.find((idx,))
    if ident is _MISS:
        flat = torch.as_tensor(idx, dtype=torch.int64).reshape(-1).cpu()
        n = flat.numel()
        ident = _identity_cache.keep((idx,), n if bool(torch.equal(flat, torch.arange(n, dtype=torch.int64))) else -1)
    return ident == rows


def contract_gathered(eq, a, rows_a, b, rows_b, out=None, label=None, _validate=True):
    """einsum(eq, a[rows_a], b[rows_b]) without materialising the gathered operands
    (artn_contract_gather): the first label of every operand that has row indices must be the
    first label of the result (the shared batch label of the sparse executor, reference
    contraction.py:149-156, :177-179).  rows_* are the reference's int64 index tensors or None.
    Returns None when the step does not fit the tiled kernel (the caller gathers instead)."""
    la, lb, lo = _parse(eq) if isinstance(eq, str) else (tuple(eq[0]), tuple(eq[1]), tuple(eq[2]))
    if a.dtype != b.dtype or a.dtype not in _DTYPES or not lo:
        return None
    lab = lo[0] if label is None else label
    if (rows_a is not None and (not la or la[0] != lab)) or (rows_b is not None and (not lb or lb[0] != lab)):
        return None
    n = len(rows_a) if rows_a is not None else len(rows_b)
    if rows_a is not None and rows_b is not None and len(rows_b) != n:
        raise RuntimeError("row index lists of the two operands differ in length")
    a, b = _as_operand(a), _as_operand(b)
    a_shape = ((n,) + tuple(a.shape[1:])) if rows_a is not None else tuple(a.shape)
    b_shape = ((n,) + tuple(b.shape[1:])) if rows_b is not None else tuple(b.shape)
    outer = _big_k_outer(la, lb, lo, a_shape, tuple(a.stride())) if label is None else None
    if outer:  # more contracted bits than a tile holds: split-K around the gathered contraction
        part = contract_gathered((la, lb, tuple(outer) + tuple(lo)), a, rows_a, b, rows_b, label=lab, _validate=_validate)
        if part is None:
            return None
        n_rows = 1
        for e in part.shape[:len(outer)]:
            n_rows *= e
        if not _sum_leading_ok(part, n_rows):
            return None
        res = sum_leading(part, n_rows, out=out.reshape(-1) if out is not None and out.is_contiguous() else None)
        if out is not None:
            if res.data_ptr() != out.data_ptr():
                out.copy_(res.reshape(out.shape))
            return out
        return res.reshape(part.shape[len(outer):])
    d, out_shape = _descriptor(la, lb, lo, a_shape, tuple(a.stride()), b_shape, tuple(b.stride()), a.dtype)
    labels = list(la) + [x for x in lb if x not in la]
    if out is None:
        out = torch.empty(out_shape, dtype=a.dtype, device=a.device)
    elif tuple(out.shape) != out_shape or not out.is_contiguous():
        raise RuntimeError("out= must be a contiguous tensor of the result shape")
    if out.numel() == 0:
        return out
    ia = _device_index(rows_a, a.device, a.shape[0] if _validate else None) if rows_a is not None else None
    ib = _device_index(rows_b, b.device, b.shape[0] if _validate else None) if rows_b is not None else None
    flag = _flag(a.device)
    with torch.cuda.device(a.device):
        e0 = e1 = None
        if profiler is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        rc = N.lib().artn_contract_gather(ctypes.byref(d), a.data_ptr(), b.data_ptr(), out.data_ptr(), labels.index(lab),
                                          ia.data_ptr() if ia is not None else None, a.shape[0],
                                          ib.data_ptr() if ib is not None else None, b.shape[0],
                                          flag.data_ptr(), N.current_stream_ptr(a.device))
        if rc == -2:
            return None
        N.check(rc)
        if profiler is not None:
            e1.record()
            profiler.record(_step_info_cached(d), e0, e1)
    _flags_used().add(a.device)
    return out


def gather_rows(t, idx, _validate=True):
    """t[idx] along dim 0 through artn_gather_rows (reference contraction.py:149-150 etc.).
    An index that selects every row in order returns `t` itself (the reference copies).
    Indices are validated once per index tensor on the host (out of range raises, negative wraps:
    the reference's indexing semantics); the kernel additionally zero-fills and flags any row it
    could not read (check_gather_flag)."""
    N.require_gpu(t, "gather_rows")
    if t.dim() > 0 and _is_identity(idx, t.shape[0]):
        return t
    t = t.contiguous()
    dev_idx = _device_index(idx, t.device, t.shape[0] if _validate and t.dim() > 0 else None)
    nrows = dev_idx.numel()
    out = torch.empty((nrows,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    if out.numel() == 0:
        return out
    row_bytes = (t.numel() // t.shape[0]) * t.element_size()
    flag = _flag(t.device)   # sticky out-of-range flag, one per device
    with torch.cuda.device(t.device):
        N.check(N.lib().artn_gather_rows(t.data_ptr(), dev_idx.data_ptr(), out.data_ptr(), nrows, row_bytes,
                                         t.shape[0], flag.data_ptr(), N.current_stream_ptr(t.device)))
    _flags_used().add(t.device)
    return out


_pair_rows_cache = _IdMemo(1024)   # (rows of the first operand, rows of the second; source row counts) -> distinct rows + pair index
ROW_PAIRS_MIN = 256          # index pairs from which the distinct-row form is considered
ROW_PAIRS_REUSE = 4          # ... when each operand's distinct rows are used at least this many times on average
ROW_PAIRS_WASTE = 2.0        # ... and the full (distinct x distinct) product computes at most this many times the pairs asked for


def _row_pairs(rows_a, rows_b, src_rows_a, src_rows_b):
    """(distinct rows of a, distinct rows of b, index of pair p in the distinct x distinct grid) for two index lists of equal
    length, or None when the pairs do not re-use rows enough.  Host side, once per pair of index tensors (cached): the
    reference's indexing semantics as in _device_index (negative counts from the end, out of range raises)."""
    extra = (src_rows_a, src_rows_b)
    got = _pair_rows_cache.find((rows_a, rows_b), extra)
    if got is not _MISS:
        return got
    res = None
    ha = torch.as_tensor(rows_a, dtype=torch.int64).reshape(-1).cpu()
    hb = torch.as_tensor(rows_b, dtype=torch.int64).reshape(-1).cpu()
    n = ha.numel()
    if n == hb.numel() and n >= ROW_PAIRS_MIN:
        for h, rows in ((ha, src_rows_a), (hb, src_rows_b)):
            if int(h.min()) < -rows or int(h.max()) >= rows:
                raise RuntimeError(f"row index out of range: indices span [{int(h.min())}, {int(h.max())}] but the operand has {rows} rows "
                                   "(IndexError in the reference, contraction.py:192-195)")
        ha = torch.where(ha < 0, ha + src_rows_a, ha)
        hb = torch.where(hb < 0, hb + src_rows_b, hb)
        ua, inva = torch.unique(ha, return_inverse=True)
        ub, invb = torch.unique(hb, return_inverse=True)
        if (ua.numel() * ROW_PAIRS_REUSE <= n and ub.numel() * ROW_PAIRS_REUSE <= n
                and ua.numel() * ub.numel() <= ROW_PAIRS_WASTE * n):
            res = (ua, ub, inva * ub.numel() + invb)
    return _pair_rows_cache.keep((rows_a, rows_b), res, extra)


def contract_row_pairs(eq, a, rows_a, b, rows_b, out=None):
    """einsum(eq, a[rows_a], b[rows_b]) -- the batched product over index PAIRS of the sparse executor's chunk loop and gathered
    step (reference contraction.py:149-156, :177-179) -- when the pairs re-use few distinct rows: ONE unbatched contraction of
    the distinct rows of `a` with the distinct rows of `b` (a plain GEMM on the bit kernels) followed by a row gather of the
    pairs asked for.  The n53 m20 big-batch slice (65 536 bitstrings) asks for 15 344 pairs of 512 x 32 distinct rows, over 2^10
    contracted values: as a batched product with a 4 x 16 block per pair it ran on the extent GEMM at 1.5 TFLOP/s (128-row
    tiles with 4 valid rows; 2 x 5.4 ms of a 76 ms slice in the reduced-precision mode), as one 2 048 x 512 x 1 024 GEMM plus a
    gather of 64-element rows it is a fraction of a millisecond.  Same products, same sums (fp32 summation order differs).
    Returns None when the pairs do not re-use rows (the caller gathers inside the kernel or materialises the gathers)."""
    la, lb, lo = _parse(eq) if isinstance(eq, str) else (tuple(eq[0]), tuple(eq[1]), tuple(eq[2]))
    if (rows_a is None or rows_b is None or not la or not lb or not lo or la[0] != lo[0] or lb[0] != lo[0]
            or la[0] in la[1:] or lb[0] in lb[1:] or lo[0] in lo[1:]
            or a.dtype != b.dtype or a.dtype not in _DTYPES or a.dim() != len(la) or b.dim() != len(lb)):
        return None
    pairs = _row_pairs(rows_a, rows_b, a.shape[0], b.shape[0])
    if pairs is None:
        return None
    ua, ub, grid = pairs
    au = gather_rows(a, ua)                 # (the identity returns `a` itself)
    bu = gather_rows(b, ub)
    ra, rb = ("row pairs", 0), ("row pairs", 1)   # two private labels in place of the shared batch label
    full = contract(((ra,) + la[1:], (rb,) + lb[1:], (ra, rb) + lo[1:]), au, bu)
    res = gather_rows(full.reshape((ua.numel() * ub.numel(),) + tuple(full.shape[2:])), grid)
    if out is not None:
        out.copy_(res.reshape(out.shape))
        return out
    return res


def _normalize_inplace(t):
    """t /= t.abs().max(); returns the device scalar abs-max (reference contraction.py:197-199)."""
    if t.dtype not in _DTYPES:
        raise RuntimeError(f"scientific_notation needs complex64 or complex128 tensors, got {t.dtype}")
    c128 = t.dtype == torch.complex128
    amax = torch.empty(1, dtype=torch.float64 if c128 else torch.float32, device=t.device)
    fn = N.lib().artn_absmax_normalize_c128 if c128 else N.lib().artn_absmax_normalize_c64
    with torch.cuda.device(t.device):
        N.check(fn(t.data_ptr(), t.numel(), amax.data_ptr(), N.current_stream_ptr(t.device)))
    return amax


def _out_numel(eq, a, b):
    la, lb, lo = _labels(eq)
    ext = dict(zip(la, a.shape))
    ext.update(zip(lb, b.shape))
    n = 1
    for x in lo:
        n *= ext[x]
    return n


def _fusable_kind(step):
    """Steps the pair fusion understands: plain 3-tuples (branch D) and un-chunked 5-tuples of
    branch (C) -- contraction, free reshape, optional row select (reference contraction.py:180-188)."""
    bi, bj = step[2]
    if len(bi) > 1:
        return False
    if len(step) == 3:
        return True
    return not (len(bi) == 1 and len(bj) == 1)


LAZY_SELECT_MIN_NUMEL = 1 << 24   # a row select of a tensor at least this big is deferred to its consumer (_RowsOf)
_lazy_state = threading.local()    # .on: set by tensor_contraction_sparse around its step loop (never under scientific_notation)
_compose_cache = _IdMemo(1024)   # (select, rows) -> select[rows]


class _RowsOf:
    """`base[idx]` along dim 0, NOT yet materialised: the row select that ends a branch-(C) step (reference
    contraction.py:187, `tensors[i] = tensors[i][batch_i[0]]`) when the tensor is big.  The steps that follow a select in
    the reference's sparse schemes are the chunk loop (A) and the gathered step (B), which index the rows of their operands
    anyway (`tensors[i][batch_i[k]]`, :149-150, :177): `base[idx][rows] == base[idx[rows]]`, so they read the un-selected
    tensor through composed indices inside the contraction kernel and the select's copy of the whole tensor -- 2 x 7.8 GB
    on the n30 x 10 000 scheme -- never happens.  Any other consumer materialises it (`rows_of`)."""
    __slots__ = ("base", "idx")

    def __init__(self, base, idx):
        self.base, self.idx = base, idx

    @property
    def shape(self):
        return (len(self.idx),) + tuple(self.base.shape[1:])


def rows_of(t):
    """The tensor itself, or the materialised rows of a deferred select."""
    return gather_rows(t.base, t.idx) if isinstance(t, _RowsOf) else t


def _composed(sel, rows):
    """sel[rows] as a CPU int64 tensor, cached per (select, rows) pair: the scheme's index tensors live as long as the scheme."""
    got = _compose_cache.find((sel, rows))
    if got is _MISS:
        s_ = torch.as_tensor(sel, dtype=torch.int64).reshape(-1)
        r_ = torch.as_tensor(rows, dtype=torch.int64).reshape(-1)
        if r_.numel() and (int(r_.min()) < -len(s_) or int(r_.max()) >= len(s_)):
            raise RuntimeError(f"row index out of range: indices span [{int(r_.min())}, {int(r_.max())}] but the operand has {len(s_)} rows "
                               "(IndexError in the reference, contraction.py:192-195)")
        got = _compose_cache.keep((sel, rows), s_[r_])
    return got


def _sparse_step(tensors, step):
    """One step of the sparse executor: the four branches of reference contraction.py:140-191."""
    i, j = step[0]
    eq = step[1]
    batch_i, batch_j = step[2]
    if isinstance(tensors[j], _RowsOf):
        tensors[j] = rows_of(tensors[j])
    lazy_i = tensors[i] if isinstance(tensors[i], _RowsOf) else None
    if lazy_i is not None and not (len(batch_i) > 1 or (len(step) > 3 and len(batch_i) == len(batch_j) == 1)):
        tensors[i] = rows_of(lazy_i)   # (a consumer that does not index rows itself)
        lazy_i = None
    if len(batch_i) > 1:
        src_i, src_j = tensors[i], tensors[j]
        if lazy_i is not None:   # rows of a deferred select: compose the indices, read the un-selected tensor
            src_i = lazy_i.base
            rows_i = [_composed(lazy_i.idx, x) for x in batch_i]
        else:
            rows_i = batch_i
        la, lb, lo = _labels(eq)
        rows = [len(x) for x in batch_i]
        first = None
        r0 = 0
        for k in range(len(batch_i)):
            if first is None:
                ext = dict(zip(la, (rows[k],) + tuple(src_i.shape[1:])))
                ext.update(zip(lb, (rows[k],) + tuple(src_j.shape[1:])))
                first = torch.empty((sum(rows),) + tuple(ext[x] for x in lo[1:]), dtype=src_i.dtype, device=src_i.device)
            dst = first[r0:r0 + rows[k]]
            # pairs that re-use few distinct rows: one plain GEMM of the distinct rows + a gather of the pairs; else rows
            # gathered inside the contraction kernel; two gathers + contraction otherwise
            if _row_pairs_on() and contract_row_pairs(eq, src_i, rows_i[k], src_j, batch_j[k], out=dst) is not None:
                pass
            elif contract_gathered(eq, src_i, rows_i[k], src_j, batch_j[k], out=dst) is None:
                contract(eq, gather_rows(src_i, rows_i[k]), gather_rows(src_j, batch_j[k]), out=dst)
            r0 += rows[k]
        if step[3]:
            first = first.reshape((-1,) + tuple(step[3][1:]))
        tensors[j] = []
        tensors[i] = first
    elif len(step) > 3 and len(batch_i) == len(batch_j) == 1:
        fused = None
        rows_i0 = batch_i[0]
        if lazy_i is not None:
            tensors[i], rows_i0 = lazy_i.base, _composed(lazy_i.idx, batch_i[0])
        plain = (isinstance(tensors[i], torch.Tensor) and isinstance(tensors[j], torch.Tensor)
                 and _is_identity(rows_i0, tensors[i].shape[0]) and _is_identity(batch_j[0], tensors[j].shape[0]))
        if not plain and isinstance(tensors[i], torch.Tensor) and isinstance(tensors[j], torch.Tensor):
            # ONE row of each operand (a single-bitstring slice): `t[idx]` is a contiguous row -- views, no gather, and the step
            # runs as a plain one (three-product stages; the row-gather instantiations have none: n53's 2^28 -> 2^29 step
            # 1.98 -> 1.4 ms)
            ri, rj = _single_row(rows_i0, tensors[i]), _single_row(batch_j[0], tensors[j])
            if ri is not None and rj is not None:
                fused = contract(eq, tensors[i][ri:ri + 1], tensors[j][rj:rj + 1])
        if (fused is None and not plain and _row_pairs_on() and isinstance(tensors[i], torch.Tensor)
                and isinstance(tensors[j], torch.Tensor)):
            fused = contract_row_pairs(eq, tensors[i], rows_i0, tensors[j], batch_j[0])
        if fused is None and not plain and isinstance(tensors[i], torch.Tensor) and tensors[i].numel() >= (1 << 20):
            fused = contract_gathered(eq, tensors[i], rows_i0, tensors[j], batch_j[0])
        if fused is None:
            tensors[i] = gather_rows(tensors[i], rows_i0)
            tensors[j] = gather_rows(tensors[j], batch_j[0])
            fused = contract(eq, tensors[i], tensors[j])
        # (the reference leaves the gathered operand in tensors[j]; here it keeps its rows when the
        #  gather ran inside the kernel -- tensors[j] is consumed by this step either way)
        tensors[i] = fused
    elif len(step) > 3:
        tensors[i] = contract(eq, tensors[i], tensors[j]).reshape(step[3])
        if len(batch_i) == 1:
            t = tensors[i]
            if (getattr(_lazy_state, "on", False) and t.numel() >= LAZY_SELECT_MIN_NUMEL and t.dim() > 0
                    and not _is_identity(batch_i[0], t.shape[0])):
                _device_index(batch_i[0], t.device, t.shape[0])   # (validated now, as gather_rows would: out of range raises here)
                tensors[i] = _RowsOf(t, batch_i[0])               # deferred: see _RowsOf
            else:
                tensors[i] = gather_rows(t, batch_i[0])
        tensors[j] = []
    else:
        tensors[i] = contract(eq, tensors[i], tensors[j])
        tensors[j] = []


_sparse_prog_cache = _IdMemo(64)   # scheme -> (shape signature, program or None, hoisted step indices)
_NO_HOIST = frozenset()


def _sparse_program(scheme, tensors):
    """Small-step program of a sparse-state scheme (compiled once per scheme object and leaf shapes): the plain
    steps (branch D) whose operands are small leaf-derived tensors, hoisted in front of everything else with the
    sequential semantics kept (_plan_small_program)."""
    items = tensors.items() if isinstance(tensors, dict) else enumerate(tensors)
    shapes, dtype, on_gpu = {}, None, True
    for k, t in items:
        if isinstance(t, torch.Tensor):
            shapes[k] = tuple(t.shape)
            dtype = dtype or t.dtype
            on_gpu = on_gpu and t.is_cuda and t.is_contiguous()
    if dtype not in _DTYPES or not on_gpu:
        return None, _NO_HOIST
    sig = tuple(shapes.items())
    hit = _sparse_prog_cache.find((scheme,), schemes=(scheme,))
    if hit is _MISS or hit[0] != sig:
        prog, main = _plan_small_program(scheme, shapes, dtype)
        hoisted = frozenset(range(len(scheme))) - frozenset(main) if prog is not None else _NO_HOIST
        hit = _sparse_prog_cache.keep((scheme,), (sig, prog, hoisted), schemes=(scheme,))
    return hit[1], hit[2]


_defer = threading.local()   # .flag_check: the slice loop reads the gather flag once, after its last slice


def tensor_contraction_sparse(tensors, contraction_scheme, scientific_notation=False):
    """Run a sparse-state scheme (reference contraction.py:132-205); the four branches are
    keyed exactly like the reference:

    (A) several index chunks: per chunk gather rows of both operands and run the batched
        contraction straight into its row range of the result (no torch.cat copy)
    (B) 5-tuple with one index list per operand: gather both, batched contraction
    (C) other 5-tuples: contraction with both batch labels in the output, merged by a
        free reshape, then an optional row select
    (D) 3-tuples: plain contraction.
    Consumed operands are released (`tensors[j] = []`) as the reference does.  Two consecutive
    (D) steps on the same big tensor run as one fused pass, like in the dense executor
    (not when scientific_notation renormalises after every step)."""
    if len(contraction_scheme) == 0:
        raise RuntimeError("empty contraction scheme")
    scheme = contraction_scheme
    schedule = _schedule_cache.find((scheme,), (_chain_plan_on(),), (scheme,))
    if schedule is _MISS:
        schedule = _schedule_cache.keep((scheme,), chain_schedule(scheme) if _chain_plan_on() else fusion_schedule(scheme),
                                        (_chain_plan_on(),), (scheme,))
    factor = None
    last = scheme[-1][0][0]

    def one(n):
        try:
            _sparse_step(tensors, scheme[n])
        except Exception as e:
            raise RuntimeError(f"tensor_contraction_sparse failed at step {n} {scheme[n][0]} {scheme[n][1]!r}: {e}") from e

    def normalize(i):
        nonlocal factor
        lg = torch.log10(_normalize_inplace(tensors[i]))
        factor = lg if factor is None else factor + lg

    if scientific_notation:
        for n in range(len(scheme)):
            one(n)
            normalize(scheme[n][0][0])
        if _flags_used() and not getattr(_defer, "flag_check", False):
            check_gather_flag("tensor_contraction_sparse")
        return factor.reshape(()).to(tensors[last].dtype), tensors[last]

    # the plain steps that only combine small leaf-derived tensors: one launch of the small-step program, as in the dense
    # executor (151 of the 180 steps of the n30 sparse-state schemes are of this kind)
    prog, hoisted = _sparse_program(scheme, tensors)
    if prog is not None:
        first = next(t for t in (tensors.values() if isinstance(tensors, dict) else tensors) if isinstance(t, torch.Tensor))
        with torch.cuda.device(first.device):
            _run_program(prog, tensors, first.dtype, first.device, N.current_stream_ptr(first.device))
        for n in hoisted:   # consumed operands are released, as the step-by-step loop does
            j = scheme[n][0][1]
            if j not in prog.outputs:
                tensors[j] = []
    _lazy_state.on = not _os_environ.get("ARTN_NO_LAZY_SELECT")
    try:
        _run_sparse_main(tensors, scheme, schedule, hoisted, one)
    finally:
        _lazy_state.on = False
    if isinstance(tensors[last], _RowsOf):
        tensors[last] = rows_of(tensors[last])
    # abort-on-failure semantics of the reference (contraction.py:192-195) for the one failure the
    # kernels cannot raise themselves: one flag read per scheme (the slice loop defers it to its end)
    if _flags_used() and not getattr(_defer, "flag_check", False):
        check_gather_flag("tensor_contraction_sparse")
    return tensors[last]


def _row_pairs_on():
    return _os_environ.get("ARTN_ROW_PAIRS", "1") not in ("0",)


def _chain_plan_on():
    return _os_environ.get("ARTN_CHAIN_PLAN", "1") not in ("0",)


class _ShapeOnly:
    """shape / stride / dtype carrier for host-only planner queries"""
    __slots__ = ("shape", "_st", "dtype")

    def __init__(self, shape, dtype):
        self.shape, self._st, self.dtype = tuple(shape), _dense_strides(tuple(shape)), dtype

    def stride(self):
        return self._st


def _plain_form(step, a_shape, b_shape, itemsize):
    """How an un-chunked step of the sparse executor runs as ONE plain contraction on (views of) its operands:
    (row of the first operand or None = all of it, row of the second or None, shape the result is viewed in or None,
    row select after it or None, shape of tensors[i] afterwards); None for the chunk loop (A) and for gathers of
    several rows (B with real index lists): those steps are never part of a fused pair."""
    bi, bj = step[2] if len(step) > 2 else ((), ())   # (dense steps are 2-tuples: plain contractions)
    if len(bi) > 1:
        return None
    la, lb, lo = _labels(step[1])
    ra = rb = view = select = None
    if len(step) > 3 and len(bi) == 1 and len(bj) == 1:   # (B): identities and single rows are views
        if len(a_shape) == 0 or len(b_shape) == 0:
            return None
        if not _is_identity(bi[0], a_shape[0]):
            ra = _single_row_of_shape(bi[0], a_shape, itemsize)
            if ra is None:
                return None
            a_shape = (1,) + tuple(a_shape[1:])
        if not _is_identity(bj[0], b_shape[0]):
            rb = _single_row_of_shape(bj[0], b_shape, itemsize)
            if rb is None:
                return None
            b_shape = (1,) + tuple(b_shape[1:])
    if len(la) != len(a_shape) or len(lb) != len(b_shape):
        return None
    ext = dict(zip(la, a_shape))
    for x, e in zip(lb, b_shape):
        if ext.setdefault(x, e) != e:
            return None
    after = tuple(ext[x] for x in lo)
    if len(step) > 3 and not (len(bi) == 1 and len(bj) == 1):   # (C): free reshape, optional row select
        n = 1
        for e in after:
            n *= e
        try:
            after = _resolve_reshape(n, step[3])
        except RuntimeError:
            return None
        view = step[3]
        view_rows = after[0] if len(after) else 0
        if len(bi) == 1:
            select = bi[0]
            after = (len(select),) + tuple(after[1:])
        return ra, rb, view, select, tuple(a_shape), tuple(b_shape), after, view_rows
    return ra, rb, view, select, tuple(a_shape), tuple(b_shape), after, 0


_chain_trace = None             # optional hook: called with the prices and the cut of every planned chain
_chain_cache = _IdMemo(1024)   # (scheme; first member, count, shape of the chain's tensor, ...) -> groups
CHAIN_BW, CHAIN_FLOPS = 5.0e12, 120e12   # what a tile-structured pass / the fp32 matrix pipe sustain: the cost model
CHAIN_PAIR_BYTES = 1.35
CHAIN_SPILL_FACTOR = 2.5
CHAIN_MIN_GAIN = 0.97                    # the pairs-from-the-left cut stays unless another one is estimated 3 % faster


def _plan_chain(tensors, scheme, members):
    """Cut the head of a chain (steps on the same first operand, chain_schedule) into single steps and fused pairs.

    Pairs from the left -- what fusion_schedule forms -- leave a declined pair as two single steps even when the second
    of them fuses with ITS successor, and never look at what a pair saves: an n53 m14 slice ran 2^29 -> 2^30 -> 2^27
    (6 contracted bits, then 5) as two launches, 17 GB through HBM for an intermediate that fits the tile.  Here every
    adjacent pair the planner accepts is a candidate, a launch is priced at max(bytes / 5 TB/s, FLOP / 120 TFLOP/s)
    (pairs: 1.35 x the bytes, re-run first stages paid for, spilling instantiations x 2.5), and a dynamic programme picks the cut; the
    left-to-right cut is kept unless the estimate improves by 3 %.  Returns groups (tuples of one or two members)
    covering a prefix of `members`: planning stops at the first step that is not a plain contraction."""
    first = scheme[members[0]]
    a = tensors[first[0][0]]
    if not isinstance(a, torch.Tensor) or not a.is_cuda or a.dtype not in _DTYPES:
        return [(members[0],)]
    extra = (members[0], len(members), tuple(a.shape), a.dtype, precision.current())
    groups = _chain_cache.find((scheme,), extra, (scheme,))
    if groups is not _MISS:
        return groups
    b_shapes = [tuple(tensors[scheme[n][0][1]].shape) if hasattr(tensors[scheme[n][0][1]], "shape") else None for n in members]
    return _chain_cache.keep((scheme,), _cut_sparse_chain(scheme, members, tuple(a.shape), b_shapes, a.dtype), extra, (scheme,))


_left_cache = _IdMemo(64)


def _left_pairs(scheme):
    """the candidate pairs of fusion_schedule (pairs from the left: the cut of rounds 1-4) as a set of (n, m)"""
    pairs = _left_cache.find((scheme,), schemes=(scheme,))
    if pairs is _MISS:
        pairs = _left_cache.keep((scheme,), frozenset((e[1], e[2]) for e in fusion_schedule(scheme) if e[0] == "pair"), schemes=(scheme,))
    return pairs


def _cut_sparse_chain(scheme, members, a_shape, b_shapes, dtype):
    """_plan_chain on shapes alone (host only): a_shape = the chain's tensor before members[0], b_shapes[q] = the second
    operand of members[q] (None: not a tensor)."""
    itemsize = 8 if dtype == torch.complex64 else 16
    a = _ShapeOnly(a_shape, dtype)
    forms, shape = [], tuple(a_shape)
    for n, sb in zip(members, b_shapes):
        f = _plain_form(scheme[n], shape, sb, itemsize) if sb is not None else None
        if f is None:
            break
        forms.append(f)
        shape = f[6]
    L = len(forms)
    if L == 0:
        return [(members[0],)]

    def numel(sh):
        r = 1
        for e in sh:
            r *= e
        return r

    single, pair, flops1, pair_q = [0.0] * L, {}, [0.0] * L, {}
    for p in range(L):
        ra, rb, view, select, sa, sb, after, view_rows = forms[p]
        la, lb, lo = _labels(scheme[members[p]][1])
        d, _ = _descriptor(la, lb, lo, sa, _dense_strides(sa), sb, _dense_strides(sb), a.dtype)
        info = _step_info_cached(d)
        single[p], flops1[p] = max(info["bytes"] / CHAIN_BW, info["flops"] / CHAIN_FLOPS), info["flops"]
        if p + 1 >= L or numel(sa) < FUSE_MIN_NUMEL:
            continue
        # between the two contractions the first step may only reshape (free) or select every row in order (the
        # identity): anything else needs the intermediate in memory; the second step reads ALL of it
        ra2, sb2 = forms[p + 1][0], forms[p + 1][5]
        if ra2 is not None or (select is not None and not _is_identity(select, view_rows)):
            continue
        out1 = numel(after)
        if out1 * FUSE_MIN_MID < numel(sa):
            continue
        try:
            d1, d2, _ = _pair_descriptors(scheme[members[p]][1], _ShapeOnly(sa, a.dtype), _ShapeOnly(sb, a.dtype),
                                          scheme[members[p + 1]][1], _ShapeOnly(sb2, a.dtype), view)
        except RuntimeError:
            continue
        q = N.ArtnStepInfo()
        rc = N.lib().artn_contract2_query(ctypes.byref(d1), ctypes.byref(d2), ctypes.byref(q))
        if rc == -2:
            continue
        N.check(rc)
        # a pair's pass costs more per byte than a single step's (two stages per tile between its copy phases: the
        # measured pairs of the n53 slices sit at 1.3-1.4 x bytes / 5 TB/s); a first stage that is re-run for every
        # value of an outer result bit re-reads its input and repeats its FLOP; and the four-wave kernel's instantiations
        # for two 5- or 6-bit stages of which the second SHRINKS the tile spill (6+5: 228 bytes of scratch, 2.05 ms for a
        # pass that two launches do in 0.72 on n53 m14; 5+5: 84 registers, 6.45 ms for 412 GFLOP on the random D = 2 network)
        rr = max(1, int(q.stage1_reruns))
        t = max(CHAIN_PAIR_BYTES * (q.bytes + (rr - 1) * float(itemsize) * numel(sa)) / CHAIN_BW,
                (q.flops + (rr - 1) * flops1[p]) / CHAIN_FLOPS)
        if q.tile_out_bits < q.tile_mid_bits and q.k_bits >= 5 and q.k2_bits >= 5 and q.grid > 256:
            t *= CHAIN_SPILL_FACTOR
        pair[p] = t
        pair_q[members[p]] = {name: getattr(q, name) for name, _ in N.ArtnStepInfo._fields_}
    best, take = [0.0] * (L + 2), [1] * L
    for p in range(L - 1, -1, -1):
        best[p], take[p] = single[p] + best[p + 1], 1
        if p in pair and pair[p] + best[p + 2] <= best[p]:
            best[p], take[p] = pair[p] + best[p + 2], 2
    left_pairs = _left_pairs(scheme)
    left, cost_left, p = [], 0.0, 0
    while p < L:
        g = 2 if p in pair and (members[p], members[p + 1]) in left_pairs else 1
        left.append(g)
        cost_left += pair[p] if g == 2 else single[p]
        p += g
    groups, p = [], 0
    use_dp = best[0] < CHAIN_MIN_GAIN * cost_left
    while p < L:
        g = take[p] if use_dp else left[len(groups)]
        groups.append(tuple(members[p:p + g]))
        p += g
    if _chain_trace is not None:   # (tools/chain_plans.py)
        _chain_trace({"members": list(members[:L]), "log2_numel": [numel(f[4]).bit_length() - 1 for f in forms], "single_ms": [x * 1e3 for x in single],
                      "pair_ms": {members[q]: v * 1e3 for q, v in pair.items()}, "cut_ms": best[0] * 1e3, "left_ms": cost_left * 1e3,
                      "groups": groups, "pair_info": pair_q})
    return groups


def _run_sparse_pair(tensors, scheme, n, m):
    """Steps n and m (a group of _plan_chain) as one pass through artn_contract2; False when the pair does not run
    after all (the caller then runs the two steps one after the other)."""
    s1, s2 = scheme[n], scheme[m]
    (i, j1), j2 = s1[0], s2[0][1]
    for t in (j1, j2):   # (second operands are never read through a deferred select)
        if isinstance(tensors[t], _RowsOf):
            tensors[t] = rows_of(tensors[t])
    a, b1, b2 = tensors[i], tensors[j1], tensors[j2]
    if not all(isinstance(t, torch.Tensor) for t in (a, b1, b2)) or not a.is_cuda:
        return False
    f1 = _plain_form(s1, tuple(a.shape), tuple(b1.shape), a.element_size())
    if f1 is None or (f1[3] is not None and not _is_identity(f1[3], f1[7])):
        return False
    f2 = _plain_form(s2, f1[6], tuple(b2.shape), a.element_size())
    if f2 is None or f2[0] is not None:
        return False
    if f1[0] is not None:
        a = a[f1[0]:f1[0] + 1]
    if f1[1] is not None:
        b1 = b1[f1[1]:f1[1] + 1]
    if f2[1] is not None:
        b2 = b2[f2[1]:f2[1] + 1]
    try:
        fused = contract2(s1[1], a, b1, s2[1], b2, mid_view=f1[2])
    except Exception as e:
        raise RuntimeError(f"tensor_contraction_sparse failed at fused steps {n}+{m}: {e}") from e
    if fused is None:
        return False
    if f2[2] is not None:   # branch (C) of the second step: free reshape, optional row select
        fused = fused.reshape(f2[2])
        if f2[3] is not None:
            if (getattr(_lazy_state, "on", False) and fused.numel() >= LAZY_SELECT_MIN_NUMEL
                    and not _is_identity(f2[3], fused.shape[0])):
                _device_index(f2[3], fused.device, fused.shape[0])
                fused = _RowsOf(fused, f2[3])
            else:
                fused = gather_rows(fused, f2[3])
    tensors[i] = fused
    tensors[j1] = []
    tensors[j2] = []
    return True


def _run_sparse_main(tensors, scheme, schedule, hoisted, one):
    """The step loop of tensor_contraction_sparse after the small-step program: single steps and fused pairs."""
    for entry in schedule:
        if entry[0] == "one":
            if entry[1] not in hoisted:
                one(entry[1])
            continue
        if entry[0] == "chain":
            members = [n for n in entry[1] if n not in hoisted]
            p = 0
            while p < len(members):
                for g in _plan_chain(tensors, scheme, members[p:]):
                    if len(g) == 1 or not _run_sparse_pair(tensors, scheme, g[0], g[1]):
                        for n in g:
                            one(n)
                    p += len(g)
            continue
        n, m = entry[1], entry[2]
        if n in hoisted or m in hoisted:
            for q in (n, m):
                if q not in hoisted:
                    one(q)
            continue
        s1, s2 = scheme[n], scheme[m]
        fused = None
        for t in (s1[0][1], s2[0][1]):   # (second operands are never read through a deferred select)
            if isinstance(tensors[t], _RowsOf):
                tensors[t] = rows_of(tensors[t])
        a = tensors[s1[0][0]]
        if (_fusable_kind(s1) and _fusable_kind(s2)
                and isinstance(a, torch.Tensor) and a.is_cuda and a.numel() >= FUSE_MIN_NUMEL
                and _out_numel(s1[1], a, tensors[s1[0][1]]) * FUSE_MIN_MID >= a.numel()):
            # between the two contractions the first step may only reshape (free) or select every
            # row in order (the identity): anything else needs the intermediate in memory
            mid_view = s1[3] if len(s1) > 3 else None
            select1 = s1[2][0][0] if len(s1) > 3 and len(s1[2][0]) == 1 else None
            rows1 = _resolve_reshape(_out_numel(s1[1], a, tensors[s1[0][1]]), mid_view)[0] if select1 is not None else 0
            if select1 is None or _is_identity(select1, rows1):
                try:
                    fused = contract2(s1[1], a, tensors[s1[0][1]], s2[1], tensors[s2[0][1]], mid_view=mid_view)
                except Exception as e:
                    raise RuntimeError(f"tensor_contraction_sparse failed at fused steps {n}+{m}: {e}") from e
        if fused is None:
            one(n)
            one(m)
        else:
            if len(s2) > 3:  # branch (C) of the second step: free reshape, optional row select
                fused = fused.reshape(s2[3])
                if len(s2[2][0]) == 1:
                    if (getattr(_lazy_state, "on", False) and fused.numel() >= LAZY_SELECT_MIN_NUMEL
                            and not _is_identity(s2[2][0][0], fused.shape[0])):
                        _device_index(s2[2][0][0], fused.device, fused.shape[0])
                        fused = _RowsOf(fused, s2[2][0][0])
                    else:
                        fused = gather_rows(fused, s2[2][0][0])
            tensors[s1[0][0]] = fused
            tensors[s1[0][1]] = []
            tensors[s2[0][1]] = []


# ----------------------------------------------------------------------------------------
# scheme compilers (host only; consume the planner's ContractionTree by duck typing)
# ----------------------------------------------------------------------------------------
def einsum_eq_convert(ixs, iy):
    """Bond-label lists -> einsum string over the reference alphabet (contraction.py:13-20).
    The label -> letter assignment follows set iteration order exactly like the reference
    so that equations compare equal string for string under the same PYTHONHASHSEED."""
    unique = list(set(sum(ixs, start=[]) + iy))
    if len(unique) > len(letters):
        raise RuntimeError(f"{len(unique)} labels exceed the {len(letters)}-letter einsum alphabet; "
                           "use contract() with label tuples instead")
    m = {lab: letters[k] for k, lab in enumerate(unique)}
    return ",".join("".join(m[x] for x in ix) for ix in ixs) + "->" + "".join(m[x] for x in iy)


def _equation(ixs, iy, labels):
    """labels="einsum": the reference's equation string (50-letter alphabet, contraction.py:9-10);
    labels="tuples": the bond labels themselves as (labels of operand 0, of operand 1, of the
    result) -- any number of distinct labels per step, understood by every executor here."""
    if labels == "einsum":
        return einsum_eq_convert(ixs, iy)
    if labels == "tuples":
        return (tuple(ixs[0]), tuple(ixs[1]), tuple(iy))
    raise RuntimeError(f"labels must be 'einsum' or 'tuples', got {labels!r}")


def contraction_scheme(ctree, labels="einsum"):
    """Dense scheme of a contraction tree (reference contraction.py:23-59): depth-first from
    the root, larger-space child first, emitted in reverse; each step is
    ((rep, other), equation) where `rep` is the vertex's representative tensor id (always
    the child with the larger sc, contraction_tree.py:305-314) and the output labels are
    list(vertex.contain_bonds).  labels="tuples" emits label tuples instead of einsum strings
    (no 50-symbol limit; not understood by the reference's executor)."""
    ctree.mark_rep_tensor()
    root = ctree.tree[ctree.all_tensors]
    bonds_of = ctree.tn.tensor_bonds
    scheme, output_bonds = [], None
    todo = [root]
    while todo:
        v = todo.pop()
        if not (v.left and v.right):
            continue
        child_labels = []
        for ch in (v.left, v.right):
            child_labels.append(bonds_of[ch.rep_tensor] if ch.is_leaf() else list(ch.contain_bonds))
        if v.rep_tensor == v.left.rep_tensor:
            pair, ixs = (v.left.rep_tensor, v.right.rep_tensor), (child_labels[0], child_labels[1])
        elif v.rep_tensor == v.right.rep_tensor:
            pair, ixs = (v.right.rep_tensor, v.left.rep_tensor), (child_labels[1], child_labels[0])
        else:
            raise ValueError("Incorrect rep tensor mark process.")
        iy = list(v.contain_bonds)
        if v is root:
            output_bonds = iy
        scheme.append((pair, _equation(ixs, iy, labels)))
        todo += [v.left, v.right] if v.left.sc > v.right.sc else [v.right, v.left]
    scheme.reverse()
    return scheme, output_bonds


def _select_bits(bitstrings, inds):
    return ["".join(b[k] for k in inds) for b in bitstrings]


def _merge_bits(bits_i, bits_j, loc_i, loc_j):
    n = len(loc_i) + len(loc_j)
    return "".join(bits_i[loc_i.index(k)] if k in loc_i else bits_j[loc_j.index(k)] for k in range(n))


class _BitTable:
    """The bitstrings as a 0/1 matrix: `packed(cols)` = the integers whose binary digits are the chosen columns, first
    column most significant -- int(''.join(b[k] for k in cols), 2) for every bitstring at once.  (The reference selects
    and compares substrings bitstring by bitstring, contraction.py:249-283: quadratic in the batch; 2^16 bitstrings of
    n53 m20 take hours there and seconds here, with the same tuples.)"""

    def __init__(self, bitstrings):
        self.n = len(bitstrings)
        self.width = len(bitstrings[0]) if self.n else 0
        self.ok = self.n > 0 and self.width <= 62 and all(len(b) == self.width for b in bitstrings)
        if self.ok:
            raw = np.frombuffer("".join(bitstrings).encode("ascii"), dtype=np.uint8).reshape(self.n, self.width)
            self.ok = bool(np.all((raw == 48) | (raw == 49)))
            self.bits = (raw - 48).astype(np.int64)

    def packed(self, cols):
        out = np.zeros(self.n, dtype=np.int64)
        for k in cols:
            out = (out << 1) | self.bits[:, k]
        return out


def _sub_bits(values, width, cols):
    """For integers of `width` binary digits: the integer made of digits `cols` (0 = most significant), in that order."""
    out = np.zeros(len(values), dtype=np.int64)
    for k in cols:
        out = (out << 1) | ((values >> (width - 1 - k)) & 1)
    return out


def _first_index(haystack, needles):
    """np.argwhere(haystack == v)[0][0] for every v of `needles` (IndexError when one is absent, like the reference)."""
    order = np.argsort(haystack, kind="stable")
    srt = haystack[order]
    pos = np.searchsorted(srt, needles, side="left")
    if len(needles) and (pos.max(initial=0) >= len(srt) or np.any(srt[np.minimum(pos, len(srt) - 1)] != needles)):
        raise IndexError("index 0 is out of bounds for axis 0 with size 0")
    return order[pos]


def contraction_scheme_sparse(ctree, bitstrings=None, sc_target=31, labels="einsum"):
    """Sparse-state scheme (reference contraction.py:208-341).

    Walks the tree's DFS order keeping, per live tensor, which final qubits it already
    fixes and the list of partial bitstrings (one batch row each).  Emits
        (edge, eq, batch_seq)                                   3-tuple, or
        (edge, eq, batch_seq, rshape_or_None, next_shape)       5-tuple
    with batch labels -1 (left rows), -2 (right rows), -3 (shared rows of a chunked
    gather), exactly the reference's step format.  Returns (scheme, bonds of the final
    tensor, bitstrings in output-row order)."""
    order = ctree.tree_order_dfs()
    tensor_bonds = ctree.tn.tensor_bonds
    final_qubits = ctree.tn.final_qubits
    if isinstance(final_qubits, (set, frozenset)):
        final_qubits = sorted(final_qubits)
    final_qubits = list(final_qubits)
    # per tensor: (sorted final-qubit positions it fixes, int-encoded partial bitstrings per row)
    info = {}
    for tid in tensor_bonds.keys():
        if tid in final_qubits:
            info[tid] = ([final_qubits.index(tid)], np.array([0, 1]))
        else:
            info[tid] = ([], np.array([-1]))
    scheme = []
    out_bits = None
    table = None   # _BitTable of the bitstrings, built when the first step with rows on both sides needs it
    for edge in order:
        i, j = edge
        bond_i, bond_j = tensor_bonds[i], tensor_bonds[j]
        shared = sorted(frozenset(bond_i) & frozenset(bond_j))
        # a shared bond that a third live tensor still carries is a hyper-edge: keep it
        keep = []
        for b in shared:
            for x in tensor_bonds.keys():
                if x == i or x == j or len(tensor_bonds[x]) == 0:
                    continue
                if b in tensor_bonds[x]:
                    keep.append(b)
                    break
        contracted = [b for b in shared if b not in keep]
        pos_i = [bond_i.index(b) for b in contracted]
        pos_j = [bond_j.index(b) for b in contracted]
        new_i = [bond_i[m] for m in range(len(bond_i)) if m not in pos_i]
        new_i += [bond_j[n] for n in range(len(bond_j)) if n not in pos_j and bond_j[n] not in new_i]
        tensor_bonds[i] = new_i
        tensor_bonds[j] = []

        fq_i, rows_i = info[i]
        fq_j, rows_j = info[j]
        fq = sorted(fq_i + fq_j)
        chunked = False
        if len(fq) == 0:
            batch_seq = [[torch.tensor([0])], [torch.tensor([0])]]
            rows = np.array([-1])
        elif len(fq_i) > 0 and len(fq_j) == 0:
            batch_seq = [[torch.tensor(list(range(len(rows_i))))], [torch.tensor([0])]]
            rows = rows_i
        elif len(fq_j) > 0 and len(fq_i) == 0:
            batch_seq = [[torch.tensor([0])], [torch.tensor(list(range(len(rows_j))))]]
            rows = rows_j
        else:
            loc_i = [fq.index(q) for q in fq_i]
            loc_j = [fq.index(q) for q in fq_j]
            bigger_left = int(len(rows_i) > len(rows_j))
            if table is None:
                table = _BitTable(bitstrings)
            fast = table.ok and len(fq) <= 62
            if fast:   # the same quantities as integers, whole batch at a time (sorted integers = sorted equal-length strings)
                wanted_v = np.unique(table.packed(fq))
                n_wanted = len(wanted_v)
            else:
                wanted = np.unique(_select_bits(bitstrings, fq))
                n_wanted = len(wanted)
            if n_wanted == 2 ** len(fq) or len(fq) + len(new_i) <= sc_target:
                # outer product of the two row sets, optionally followed by a row select
                if fast:
                    w = len(fq)
                    ri, rj = np.asarray(rows_i, dtype=np.int64), np.asarray(rows_j, dtype=np.int64)
                    rows = np.zeros((len(ri), len(rj)), dtype=np.int64)
                    for n_, k in enumerate(loc_i):   # digit n_ of a left row lands at position k of the merged string
                        rows |= (((ri >> (len(fq_i) - 1 - n_)) & 1) << (w - 1 - k))[:, None]
                    for n_, k in enumerate(loc_j):
                        rows |= (((rj >> (len(fq_j) - 1 - n_)) & 1) << (w - 1 - k))[None, :]
                    rows = rows.reshape(-1)
                else:
                    rows = np.array([
                        int(_merge_bits(np.binary_repr(x, len(fq_i)), np.binary_repr(y, len(fq_j)), loc_i, loc_j), 2)
                        for x in rows_i for y in rows_j])
                if n_wanted != len(rows):
                    if fast:
                        sel = np.sort(_first_index(rows, wanted_v))
                        rows = rows[sel]
                    else:
                        sel = np.sort(np.array([np.argwhere(rows == int(s, 2))[0][0] for s in wanted]))
                        rows = np.array([rows[k] for k in sel])
                    batch_seq = [[torch.tensor(sel)], []]
                else:
                    batch_seq = [[], []]
            else:
                # too big for an outer product: gather matching row pairs, in chunks
                if fast:
                    part = np.stack([_sub_bits(wanted_v, len(fq), loc_i), _sub_bits(wanted_v, len(fq), loc_j)])
                    rows = wanted_v.copy()
                    pairs = np.stack([_first_index(np.asarray(rows_i, dtype=np.int64), part[0]),
                                      _first_index(np.asarray(rows_j, dtype=np.int64), part[1])], axis=1)
                else:
                    part = np.stack([
                        np.array([int(s, 2) for s in _select_bits(wanted, loc_i)]),
                        np.array([int(s, 2) for s in _select_bits(wanted, loc_j)])])
                    rows = np.array([int(s, 2) if len(s) > 0 else -1 for s in wanted])
                    pairs = np.array([[np.argwhere(rows_i == bi)[0][0], np.argwhere(rows_j == bj)[0][0]]
                                      for bi, bj in zip(part[0], part[1])])
                perm = np.argsort(pairs[:, 1 - bigger_left])
                pairs = pairs[perm].T.reshape(2, -1)
                batch_seq = [[torch.from_numpy(pairs[0])], [torch.from_numpy(pairs[1])]]
                assert torch.max(batch_seq[0][0]) < len(rows_i)
                assert torch.max(batch_seq[1][0]) < len(rows_j)
                n_chunks = 2 ** ceil(max(0, np.log2(len(rows)) + max(len(bond_i), len(bond_j)) - (sc_target - 2)))
                if n_chunks > 1:
                    length = int(len(rows) / n_chunks)
                    if len(rows) % n_chunks > 0:
                        n_chunks += 1
                    batch_seq = [
                        [batch_seq[0][0][c * length:(c + 1) * length] for c in range(n_chunks)],
                        [batch_seq[1][0][c * length:(c + 1) * length] for c in range(n_chunks)]]
                chunked = True
                rows = rows[perm]
            assert len(rows) == n_wanted

        iy = []
        if len(fq_j):
            has_j = 1
            if chunked:
                ix_right = [-3] + bond_j
                iy.insert(0, -3)
            else:
                ix_right = [-2] + bond_j
                iy.insert(0, -2)
        else:
            has_j = 0
            ix_right = bond_j
        if len(fq_i):
            has_i = 1
            if chunked:
                ix_left = [-3] + bond_i
            else:
                ix_left = [-1] + bond_i
                iy.insert(0, -1)
        else:
            has_i = 0
            ix_left = bond_i
        iy = iy + tensor_bonds[i]
        eq = _equation((ix_left, ix_right), iy, labels)
        if has_i and has_j:
            next_shape = (len(rows),) + (2,) * len(tensor_bonds[i])
            if chunked:
                scheme.append((edge, eq, batch_seq, None, next_shape))
            else:
                scheme.append((edge, eq, batch_seq, (-1,) + (2,) * len(tensor_bonds[i]), next_shape))
        else:
            scheme.append((edge, eq, batch_seq))
        info[i] = (fq, rows)
        if edge == order[-1]:
            out_bits = [np.binary_repr(n, len(final_qubits)) for n in info[i][1]]
    return scheme, tensor_bonds[i], out_bits
