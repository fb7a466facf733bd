"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/artn.h declares, refuses to compute without a GPU (no CPU fallback), and its
host-only planner answers queries."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import artensor_amd as A
from artensor_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "artn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"#ifdef ARTN_DEV_\w+.*?#endif", "", text, flags=re.S)   # (development-only entry points: not the product ABI)
    return sorted(set(re.findall(r"\b(artn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = N.lib()
    names = declared_symbols()
    assert len(names) >= 10
    assert names == N.exported_symbols()
    for name in names:
        assert hasattr(lib, name), name
    assert lib.artn_abi_version() == N.ABI_VERSION
    assert lib.artn_device_count() >= 0


def test_product_library_carries_no_development_entry_points_or_switches():
    """ABI 7 (round 5): the experiments that lost their A/B measurements -- three-step fusion, the split-bf16 arithmetic,
    some thirty planner switches -- are compiled into development builds only (make dev); the product library exports the
    header's entry points and reads a handful of switches from the environment."""
    lib = N.lib()
    dev = os.path.basename(N.LIB_PATH) != "libartn_hip.so" or os.environ.get("ARTN_LIB")
    if not dev and not hasattr(lib, "artn_contract3"):
        assert not N.has("artn_contract3_query")
        assert A.contraction.triple_info("ABCDEFGHIJKLMNOPQRSTUVWX,ABCab->DEFGHIJKLMNOPQRSTUVWXab", (2,) * 24, (2,) * 5,
                             "DEFGHIJKLMNOPQRSTUVWXab,DEFcd->GHIJKLMNOPQRSTUVWXabcd", (2,) * 5,
                             "GHIJKLMNOPQRSTUVWXabcd,GHIef->JKLMNOPQRSTUVWXabcdef", (2,) * 5) is None
    blob = open(N.LIB_PATH, "rb").read()
    switches = set(re.findall(rb"ARTN_[A-Z][A-Z0-9_]{2,}", blob))
    env_like = {s.decode() for s in switches}
    # names that are not environment switches (error text, macros that ended up in strings) are few; the bound is on all of them
    if not hasattr(lib, "artn_contract3"):
        assert len(env_like) < 15, sorted(env_like)


def test_struct_layouts_match_header():
    # sizes the C side was compiled with (ArtnStepDesc: 2 int32 + 4 arrays of 96 int64)
    assert ctypes.sizeof(N.ArtnStepDesc) == 8 + 4 * 96 * 8
    assert ctypes.sizeof(N.ArtnStepInfo) == 10 * 4 + 2 * 8 + 2 * 8 + 4 * 4 + 8 + 8 + 2 * 4   # (+ k3_bits, stage1_reruns: ABI versions 5, 7)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the CPU-only refusal")
def test_no_cpu_fallback_anywhere():
    a = torch.zeros(2, 2, dtype=torch.complex64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        A.contract("ab,bc->ac", a, a)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        A.tensor_contraction({0: a, 1: a}, [((0, 1), "ab,bc->ac")])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        A.tensor_contraction_sparse({0: a, 1: a}, [((0, 1), "ab,bc->ac", [[torch.tensor([0])], [torch.tensor([0])]])])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        A.accumulate(a, a)
    # the raw ABI refuses too: no device -> ARTN_E_NODEVICE, never a host computation
    d, _ = A.contraction._descriptor(("a", "b"), ("b", "c"), ("a", "c"), (2, 2), (2, 1), (2, 2), (2, 1), torch.complex64)
    buf = np.zeros(8, dtype=np.complex64)
    p = buf.ctypes.data_as(ctypes.c_void_p)
    assert N.lib().artn_contract(ctypes.byref(d), p, p, p, None) == -4
    assert b"gfx950" in N.lib().artn_last_error()
    assert N.lib().artn_axpy_c64(p, p, 4, None) == -4
    assert N.lib().artn_sum_axis_c64(p, p, 1, 2, 2, None) == -4
    assert N.lib().artn_contract_gather(ctypes.byref(d), p, p, p, 0, None, 0, None, 0, None, None) == -4
    assert N.lib().artn_contract2(ctypes.byref(d), ctypes.byref(d), p, p, p, p, None) == -4
    assert N.lib().artn_contract2_acc(ctypes.byref(d), ctypes.byref(d), p, p, p, p, None) == -4
    assert N.lib().artn_contract_acc(ctypes.byref(d), p, p, p, None) == -4
    if N.has("artn_contract3"):   # (development builds)
        assert N.lib().artn_contract3(ctypes.byref(d), ctypes.byref(d), ctypes.byref(d), p, p, p, p, p, None) == -4
    assert N.lib().artn_absmax_normalize_c128(p, 4, p, None) == -4
    assert N.lib().artn_gather_rows(p, p, p, 1, 8, 1, None, None) == -4
    assert N.lib().artn_absmax_normalize_c64(p, 4, p, None) == -4


def test_planner_query_and_errors():
    info = A.step_info("ab,bc->ac", (2, 2), (2, 2))
    assert info["kernel"] == N.KERNEL_GENERIC and info["flops"] == 64.0 and info["out_shape"] == (2, 2)
    la = "ABCDEFGHIJKLMNOPQRSTUV"
    info = A.step_info(la + ",DKRUvwxyX->" + "".join(c for c in la if c not in "DKRU") + "vwxyX", (2,) * 22, (2,) * 9)
    assert info["kernel"] == N.KERNEL_BITS_MFMA and info["k_bits"] == 4
    assert info["lds_bytes"] <= 160 * 1024 and info["run_in_bits"] >= 1
    with pytest.raises(RuntimeError):
        A.step_info("ab,bc", (2, 2), (2, 2))
    with pytest.raises(RuntimeError):
        A.step_info("aab,bc->ac", (2, 2, 2), (2, 2))
    with pytest.raises(RuntimeError):
        A.step_info("ab,bc->ad", (2, 2), (2, 2))
    with pytest.raises(RuntimeError):
        A.step_info("ab,bc->ac", (2, 3), (2, 2))


def test_label_tuples_lift_the_alphabet_limit():
    # 60 distinct labels: more than the reference's 50-letter einsum alphabet (contraction.py:9-10)
    la = tuple(range(40))
    lb = tuple(range(36, 60))
    lo = tuple(x for x in la if x < 36) + tuple(x for x in lb if x >= 40)
    info = A.step_info((la, lb, lo), (2,) * 40, (2,) * 24)
    assert info["flops"] == 8.0 * 2.0 ** 60
    with pytest.raises(RuntimeError, match="alphabet"):
        A.einsum_eq_convert(([list(la)], [list(lb)])[0] + [list(lb)], list(lo))


def test_row_indices_are_validated_on_the_host_with_the_reference_semantics():
    """tensors[i][idx] in the reference (contraction.py:149-150, :177-178, :187) raises IndexError for
    an index outside [-rows, rows) and wraps negative ones; _device_index does both once per index
    tensor, before anything is launched."""
    from artensor_amd import contraction as C
    ok = torch.tensor([0, 3, -1, -4, 2])
    dev = C._device_index(ok, "cpu", 4)
    assert dev.tolist() == [0, 3, 3, 0, 2]
    assert C._device_index(ok, "cpu", 4) is dev                 # cached per (index tensor, rows)
    with pytest.raises(RuntimeError, match="row index out of range"):
        C._device_index(torch.tensor([0, 4]), "cpu", 4)
    with pytest.raises(RuntimeError, match="row index out of range"):
        C._device_index(torch.tensor([-5]), "cpu", 4)
    assert C._device_index(torch.tensor([7, 9]), "cpu", None).tolist() == [7, 9]   # unvalidated (kernel flag tests)


def test_precision_is_thread_local_and_caches_are_bounded():
    import threading
    from artensor_amd import contraction as C
    seen = {}

    def other():
        seen["before"] = C.precision.current()
        with C.precision("bf16"):
            seen["inside"] = C.precision.current()
        seen["after"] = C.precision.current()

    with C.precision("bf16"):
        th = threading.Thread(target=other)
        th.start()
        th.join()
        assert C.precision.current() == "bf16"
    assert C.precision.current() is None
    assert seen == {"before": None, "inside": "bf16", "after": None}
    b = C._Bounded(3)
    for k in range(10):
        b[k] = k
        assert len(b) <= 3
    assert b[9] == 9
    assert isinstance(C._desc_cache, C._Bounded) and isinstance(C._plan_cache, C._Bounded)


def test_single_row_index_lists_follow_the_reference_indexing():
    """`t[idx]` with ONE index (reference contraction.py:177-179): the sparse executor takes the row as a view -- negative indices
    count from the end, out of range raises (the reference dies with IndexError, contraction.py:192-195), several rows or rows of
    8 bytes (not 16-byte aligned views) keep the gather path."""
    from artensor_amd import contraction as C
    t = torch.zeros(3, 4, dtype=torch.complex64)
    assert C._single_row(torch.tensor([2]), t) == 2
    assert C._single_row(torch.tensor([-1]), t) == 2 and C._single_row(torch.tensor([-3]), t) == 0
    assert C._single_row(torch.tensor([0, 1]), t) is None
    assert C._single_row(torch.tensor([0]), torch.zeros(3, 1, dtype=torch.complex64)) is None
    assert C._single_row(torch.tensor([0]), torch.zeros((), dtype=torch.complex64)) is None
    with pytest.raises(RuntimeError, match="row index out of range"):
        C._single_row(torch.tensor([3]), t)
    with pytest.raises(RuntimeError, match="row index out of range"):
        C._single_row(torch.tensor([-4]), t)
