"""ctypes binding of libmi_knn.so (C ABI: include/mi_knn.h).

There is no CPU fallback: every compute entry point of this package goes through the HIP
library, and loading fails loudly when the library is missing."""
from __future__ import annotations

import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI_KNN_LIB") or os.path.join(HERE, "libmi_knn.so")   # override: experiments only

MI_F32, MI_BF16 = 0, 1
MI_MAX_K, MI_MAX_CLASSES = 1024, 65536
ABI_VERSION = 8

EXPORTS = [
    "mi_abi_version", "mi_last_error", "mi_profile_enable", "mi_profile_collect", "mi_profile_clock", "mi_debug_force_fallback", "mi_knn_ld", "mi_pack_rows", "mi_knn_topk_workspace_bytes", "mi_knn_topk_workspace_bytes_for_budget",
    "mi_knn_topk", "mi_knn_topk_static", "mi_knn_merge", "mi_knn_vote", "mi_queue_push", "mi_bank_scatter", "mi_moco_workspace_bytes",
    "mi_moco_fwd", "mi_moco_bwd", "mi_queue_push_stash", "mi_bank_append_normalized", "mi_queue_push_wrap", "mi_soft_knn_workspace_bytes", "mi_soft_knn", "mi_kth_largest",
]


class MiKnnError(RuntimeError):
    pass


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MiKnnError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  This package has no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    i64, i32, p, f32, sz = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_float, ctypes.c_size_t
    L.mi_abi_version.restype = i32
    L.mi_last_error.restype = ctypes.c_char_p
    L.mi_knn_ld.restype, L.mi_knn_ld.argtypes = i64, [i64, i32]
    L.mi_profile_enable.restype, L.mi_profile_enable.argtypes = i32, [i32]
    L.mi_profile_collect.restype = i32
    L.mi_profile_collect.argtypes = [i32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(i64), ctypes.POINTER(ctypes.c_double)]
    L.mi_profile_clock.restype, L.mi_profile_clock.argtypes = i32, [p, i64]
    L.mi_debug_force_fallback.restype, L.mi_debug_force_fallback.argtypes = i32, [i32]
    L.mi_pack_rows.restype = i32
    L.mi_pack_rows.argtypes = [p, i64, i32, i64, i64, i32, p, i64, p]
    L.mi_bank_append_normalized.restype = i32
    L.mi_bank_append_normalized.argtypes = [p, i64, i64, i64, i32, p, i64, i64, i64, f32, p]
    L.mi_knn_topk_workspace_bytes.restype = sz
    L.mi_knn_topk_workspace_bytes.argtypes = [i64, i64, i32]
    L.mi_knn_topk_workspace_bytes_for_budget.restype = sz
    L.mi_knn_topk_workspace_bytes_for_budget.argtypes = [i64, i64, i32, sz]
    L.mi_knn_topk.restype = i32
    L.mi_knn_topk.argtypes = [p, i64, p, i64, i32, i64, i64, i64, i32, i64, p, p, p, sz, p]
    L.mi_knn_topk_static.restype = i32
    L.mi_knn_topk_static.argtypes = [p, i64, p, i64, i32, i64, i64, i64, i32, i64, p, p, p]
    L.mi_knn_merge.restype = i32
    L.mi_knn_merge.argtypes = [p, p, i32, i64, i32, i32, p, p, p]
    L.mi_knn_vote.restype = i32
    L.mi_knn_vote.argtypes = [p, p, p, i64, i64, i32, i32, f32, p, p, p]
    L.mi_queue_push.restype = i32
    L.mi_queue_push.argtypes = [p, i64, i64, i64, p, i64, ctypes.POINTER(i64), p]
    L.mi_queue_push_wrap.restype = i32
    L.mi_queue_push_wrap.argtypes = [p, i64, i64, i64, p, i64, ctypes.POINTER(i64), p]
    L.mi_soft_knn_workspace_bytes.restype = sz
    L.mi_soft_knn_workspace_bytes.argtypes = [i64, i64, i64, i64]
    L.mi_soft_knn.restype = i32
    L.mi_soft_knn.argtypes = [p, p, p, p, i64, i64, i64, i64, f32, f32, p, p, sz, p]
    L.mi_kth_largest.restype = i32
    L.mi_kth_largest.argtypes = [p, i64, i64, p, p]
    L.mi_bank_scatter.restype = i32
    L.mi_bank_scatter.argtypes = [p, i64, i64, p, p, i64, p]
    L.mi_moco_workspace_bytes.restype = sz
    L.mi_moco_workspace_bytes.argtypes = [i64, i64, i64]
    L.mi_moco_fwd.restype = i32
    L.mi_moco_fwd.argtypes = [p, p, p, i64, i64, i64, f32, p, p, p, p, sz, p]
    L.mi_moco_bwd.restype = i32
    L.mi_moco_bwd.argtypes = [p, p, p, i64, i64, i64, f32, p, p, p, p, p, p, i64, i64, p, sz, p]
    L.mi_queue_push_stash.restype = i32
    L.mi_queue_push_stash.argtypes = [p, i64, i64, i64, p, i64, p, ctypes.POINTER(i64), p]
    if L.mi_abi_version() != ABI_VERSION:
        raise MiKnnError(f"libmi_knn.so ABI {L.mi_abi_version()} != expected {ABI_VERSION}: rebuild")
    _lib = L
    return L


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().mi_last_error().decode(errors="replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise MiKnnError(f"{what} failed (code {rc}): {msg}")


def stream_ptr(device) -> int:
    import torch
    return torch.cuda.current_stream(device).cuda_stream


def profile_enable(on: bool) -> None:
    check(lib().mi_profile_enable(1 if on else 0), "mi_profile_enable")


def profile_collect(kind: int):
    """(total_ms, launches, total_work) of the recorded launches of `kind` (0 = similarity, 1 = select)."""
    ms, n, work = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_double(0)
    check(lib().mi_profile_collect(kind, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(work)), "mi_profile_collect")
    return ms.value, n.value, work.value
