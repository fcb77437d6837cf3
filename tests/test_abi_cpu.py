"""CPU tests: the C-ABI library loads, exports every symbol include/mi_knn.h declares, and its
argument validation works without a GPU (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mi_knn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    import image_classification_amd as ica
    L = ica._lib.lib()
    declared = _declared_symbols()
    assert declared and set(declared) == set(ica._lib.EXPORTS)
    for sym in declared:
        assert hasattr(L, sym), f"{sym} declared in mi_knn.h but not exported"
    assert L.mi_abi_version() == ica._lib.ABI_VERSION


def test_argument_validation_without_gpu():
    import image_classification_amd as ica
    L = ica._lib.lib()
    assert L.mi_knn_ld(128, ica._lib.MI_F32) == 128 and L.mi_knn_ld(100, ica._lib.MI_F32) == 128
    assert L.mi_knn_ld(100, ica._lib.MI_BF16) == 128 and L.mi_knn_ld(2048, ica._lib.MI_BF16) == 2048
    assert L.mi_knn_topk_workspace_bytes(10000, 50000, 200) > 0
    assert L.mi_knn_topk_workspace_bytes(10, 50, 4096) == 0             # k > MI_MAX_K
    assert L.mi_knn_topk(None, 0, None, 0, 0, 1, 1, 1, 1, 0, None, None, None, 0, None) == -1
    assert b"null pointer" in L.mi_last_error()
    assert L.mi_queue_push(None, 1, 1, 0, None, 1, None, None) == -1
    # the allocation-free variant rejects shapes beyond its fixed scratch before it touches the device
    assert L.mi_knn_topk_static(None, 0, None, 0, 0, 3585, 100000, 128, 200, 0, None, None, None) == -3
    assert b"static scratch" in L.mi_last_error()
    assert L.mi_knn_topk_static(None, 0, None, 0, 0, 100000, 100000, 128, 1, 0, None, None, None) == -3
    assert L.mi_knn_topk_static(None, 0, None, 0, 0, 0, 10, 128, 1, 0, None, None, None) == -1
    assert L.mi_knn_vote(None, None, None, 1, 1, 1, 1, 0.1, None, None, None) == -1


def test_workspace_covers_worst_case_stage():
    """cap >= columns of every stage: candidate lists cannot overflow whatever the data order."""
    import image_classification_amd as ica
    L = ica._lib.lib()
    for B, N, k in [(1, 5000, 7), (300, 60000, 200), (10000, 50000, 200), (50000, 1280000, 200)]:
        ws = L.mi_knn_topk_workspace_bytes(B, N, k)
        b_pad = (B + 255) // 256 * 256
        assert ws >= b_pad * 2048 * 8


def test_workspace_budget_is_monotone_and_capped():
    """More candidate budget never shrinks the workspace, and it stops growing once one stage covers the bank."""
    import image_classification_amd as ica
    L = ica._lib.lib()
    B, N, k = 16384, 1_280_000, 200
    sizes = [L.mi_knn_topk_workspace_bytes_for_budget(B, N, k, g << 30) for g in (1, 8, 16, 64, 1024, 1 << 20)]
    assert all(a <= b for a, b in zip(sizes, sizes[1:])) and sizes[-1] == sizes[-2]
    assert sizes[1] == L.mi_knn_topk_workspace_bytes(B, N, k)
    assert sizes[-1] >= 16384 * 1_280_000 * 8                      # the whole bank in one stage: worst case of every similarity passing
    assert L.mi_knn_topk_workspace_bytes_for_budget(B, N, 2000, 8 << 30) == 0


def test_cpu_tensors_are_rejected_not_silently_computed():
    import torch
    import image_classification_amd as ica
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ica.knn_predict(torch.zeros(2, 32), torch.zeros(32, 8), torch.zeros(8, dtype=torch.long), 3, 2, 0.1)
    mb = ica.MemoryBank()
    mb.register("negative", dim=4, k=8)
    assert mb.offsets["negative"] == 0 and mb.sizes["negative"] == 8 and "negative" in mb.state_dict()
    assert float(mb["negative"].min()) >= 0.0 and float(mb["negative"].max()) < 1.0
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mb.push("negative", torch.zeros(2, 4))
    sb = ica.StorageBank()
    sb.register("test_ys", dim=-1, k=4, dtype=torch.long)
    assert sb["test_ys"].tolist() == [torch.iinfo(torch.int64).min] * 4
