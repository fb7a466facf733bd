"""ctypes binding of libartn_hip.so (C ABI: include/artn.h).

PyTorch is plumbing here: it owns device memory (`Tensor.data_ptr()`), the current HIP
stream and, elsewhere, torch.distributed.  All arithmetic of the hot path happens in the
HIP kernels behind the C ABI.  There is NO CPU fallback: if the library is missing or a
tensor is not on a GPU the call raises.
"""
import ctypes
import os
import threading

import torch

ABI_VERSION = 7
ARTN_MAX_LABELS = 96
ARTN_PROGRAM_MAX_EXT = 256
ARTN_C64, ARTN_C128, ARTN_C64_BF16 = 0, 1, 2
KERNEL_GENERIC, KERNEL_BITS_MFMA, KERNEL_GEMM_MFMA, KERNEL_PGEMM, KERNEL_XGEMM = 0, 1, 2, 4, 5

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ARTN_LIB") or os.path.join(_HERE, "libartn_hip.so")  # ARTN_LIB: diagnostic builds


class ArtnStepDesc(ctypes.Structure):
    _fields_ = [
        ("dtype", ctypes.c_int32),
        ("n_labels", ctypes.c_int32),
        ("extent", ctypes.c_int64 * ARTN_MAX_LABELS),
        ("stride_a", ctypes.c_int64 * ARTN_MAX_LABELS),
        ("stride_b", ctypes.c_int64 * ARTN_MAX_LABELS),
        ("stride_c", ctypes.c_int64 * ARTN_MAX_LABELS),
    ]


class ArtnStepInfo(ctypes.Structure):
    _fields_ = [
        ("kernel", ctypes.c_int32),
        ("k_bits", ctypes.c_int32),
        ("m_tile_bits", ctypes.c_int32),
        ("n_tile_bits", ctypes.c_int32),
        ("tile_in_bits", ctypes.c_int32),
        ("tile_out_bits", ctypes.c_int32),
        ("run_in_bits", ctypes.c_int32),
        ("run_out_bits", ctypes.c_int32),
        ("lds_bytes", ctypes.c_int32),
        ("grid", ctypes.c_int32),
        ("n_tiles", ctypes.c_int64),
        ("a_rereads", ctypes.c_int64),
        ("flops", ctypes.c_double),
        ("bytes", ctypes.c_double),
        ("k2_bits", ctypes.c_int32),
        ("n2_tile_bits", ctypes.c_int32),
        ("tile_mid_bits", ctypes.c_int32),
        ("arith", ctypes.c_int32),
        ("mfma_flops", ctypes.c_double),
        ("workspace_bytes", ctypes.c_int64),
        ("k3_bits", ctypes.c_int32),
        ("stage1_reruns", ctypes.c_int32),
    ]


_lib = None
_lock = threading.Lock()

_EXPORTS = {
    "artn_abi_version": (ctypes.c_int, []),
    "artn_last_error": (ctypes.c_char_p, []),
    "artn_device_count": (ctypes.c_int, []),
    "artn_contract_query": (ctypes.c_int, [ctypes.POINTER(ArtnStepDesc), ctypes.POINTER(ArtnStepInfo)]),
    "artn_last_plan_note": (ctypes.c_char_p, []),
    "artn_contract": (ctypes.c_int, [ctypes.POINTER(ArtnStepDesc), ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_void_p]),
    "artn_contract_ws": (ctypes.c_int, [ctypes.POINTER(ArtnStepDesc), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "artn_contract_gather": (ctypes.c_int, [ctypes.POINTER(ArtnStepDesc), ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
                                            ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "artn_contract2_query": (ctypes.c_int, [ctypes.POINTER(ArtnStepDesc), ctypes.POINTER(ArtnStepDesc),
                                            ctypes.POINTER(ArtnStepInfo)]),
    "artn_contract2": (ctypes.c_int, [ctypes.POINTER(ArtnStepDesc), ctypes.POINTER(ArtnStepDesc), ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "artn_contract_acc": (ctypes.c_int, [ctypes.POINTER(ArtnStepDesc), ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p]),
    "artn_contract2_acc": (ctypes.c_int, [ctypes.POINTER(ArtnStepDesc), ctypes.POINTER(ArtnStepDesc), ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "artn_program_record_bytes": (ctypes.c_int64, []),
    "artn_program_image_bytes": (ctypes.c_int64, [ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32]),
    "artn_program_build": (ctypes.c_int, [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_int64]),
    "artn_program_run": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                        ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]),
    "artn_gather_rows": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                        ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "artn_axpy_c64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "artn_sum_axis_c64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                         ctypes.c_int64, ctypes.c_void_p]),
    "artn_axpy_c128": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "artn_sum_axis_c128": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                          ctypes.c_int64, ctypes.c_void_p]),
    "artn_probe_mfma_rate": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]),
    "artn_absmax_normalize_c64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                                 ctypes.c_void_p]),
    "artn_absmax_normalize_c128": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                                  ctypes.c_void_p]),
}


# entry points of DEVELOPMENT builds only (make dev): bound when the loaded library has them
_DEV_EXPORTS = {
    "artn_contract3_query": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "artn_contract3": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
}


def has(name):
    """True when the loaded library exports `name` (the development-only entry points)."""
    try:
        getattr(lib(), name)
        return True
    except AttributeError:
        return False


def exported_symbols():
    """Names include/artn.h declares (used by the CPU test that checks the library exports them)."""
    return sorted(_EXPORTS)


def lib():
    """Load libartn_hip.so (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make` (or __graft_entry__.build()); "
                "artensor_amd has no CPU fallback")
        # torch must be imported first so that its bundled libamdhip64.so (same SONAME)
        # is the HIP runtime this library binds to -- one runtime per process.
        handle = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in _EXPORTS.items():
            if os.environ.get("ARTN_LIB") and not hasattr(handle, name):
                continue  # diagnostic builds of older revisions (tools/libartn_prev.so) may lack newer entry points
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        for name, (res, args) in _DEV_EXPORTS.items():
            if hasattr(handle, name):
                fn = getattr(handle, name)
                fn.restype = res
                fn.argtypes = args
        if handle.artn_abi_version() != ABI_VERSION:
            raise RuntimeError("libartn_hip.so ABI version mismatch")
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().artn_last_error()
        raise RuntimeError(f"artn error {rc}: {msg.decode() if msg else '?'}")


def current_stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu(t, what):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(
            f"{what}: artensor_amd executes on MI355X only (got "
            f"{'a ' + str(t.device) + ' tensor' if isinstance(t, torch.Tensor) else type(t).__name__}); "
            "there is no CPU fallback -- move the tensors to a 'cuda' device")
