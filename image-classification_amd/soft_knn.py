"""CoMatch soft-kNN pseudo-label smoothing on MI355X -- the arithmetic of ``track_semi/comatch.py:185-189``:

    A = torch.exp(torch.mm(un_w_feature, mem['mb_feature'].t()) / temperature)
    A = A / A.sum(1, keepdim=True)
    probs = alpha * un_w_probs + (1 - alpha) * torch.mm(A, mem['mb_prob'])

i.e. the dense (k = K) form of the exp-weighted vote with soft labels.  One fused kernel (``mi_soft_knn``)
recomputes the similarities tile by tile with fp32 MFMA and never writes the [b, K] matrix ``A``; the two
``mem.push`` calls that follow in the reference (``comatch.py:191-192``) stay ``MemoryBank.push``.
No CPU path: tensors must be on a HIP device."""
from __future__ import annotations

import torch

from . import _lib

__all__ = ["smooth_prob"]


def smooth_prob(un_w_feature: torch.Tensor, mb_feature: torch.Tensor, mb_prob: torch.Tensor,
                un_w_probs: torch.Tensor, temperature: float, alpha: float) -> torch.Tensor:
    """``probs`` of ``comatch.py:187-189``: [b, n_classes] float32."""
    for name, t in (("un_w_feature", un_w_feature), ("mb_feature", mb_feature), ("mb_prob", mb_prob), ("un_w_probs", un_w_probs)):
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise RuntimeError(f"{name} must be a tensor on an MI355X (HIP) device: image-classification_amd has no CPU fallback")
    f = un_w_feature.detach().float().contiguous()
    bf = mb_feature.detach().float().contiguous()
    bp = mb_prob.detach().float().contiguous()
    p = un_w_probs.detach().float().contiguous()
    if f.dim() != 2 or bf.dim() != 2 or bp.dim() != 2 or p.dim() != 2:
        raise ValueError("expected 2-D tensors")
    b, D = f.shape
    K, C = bp.shape
    if bf.shape != (K, D) or p.shape != (b, C):
        raise ValueError(f"shape mismatch: feature {tuple(f.shape)}, mb_feature {tuple(bf.shape)}, mb_prob {tuple(bp.shape)}, probs {tuple(p.shape)}")
    L = _lib.lib()
    ws_bytes = int(L.mi_soft_knn_workspace_bytes(b, K, D, C))
    if ws_bytes == 0:
        raise ValueError(f"unsupported soft-kNN shape b={b} K={K} D={D} C={C} (1 <= C <= 255)")
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=f.device)
    out = torch.empty_like(p)
    with torch.cuda.device(f.device):
        _lib.check(L.mi_soft_knn(f.data_ptr(), bf.data_ptr(), bp.data_ptr(), p.data_ptr(), b, K, D, C,
                                 float(temperature), float(alpha), out.data_ptr(), ws.data_ptr(), ws_bytes,
                                 _lib.stream_ptr(f.device)), "mi_soft_knn")
    return out
