"""Two small-k / order-statistic bank queries next to the kNN monitor.

* :func:`neighbor_label_agreement` -- ``track_nl/betacs_v2.py:159-163``: cosine similarity of a batch against the
  feature bank, ``topk(10)``, the neighbours' predicted labels (``mb['pred'][indices].argmax(-1)``) and how many of
  them agree with each sample's noisy label.  The similarity + top-k is the fused MFMA kernel at k = 10
  (``mi_knn_topk``); the gather / argmax / compare on [b, 10] stay torch ops.  The cosine comes from the third-party
  ``lumo.contrib.nn.functional.batch_cosine_similarity2`` (absent: PARITY UNPINNED); here it is
  ``a.b / (max(|a|, eps) * max(|b|, eps))`` -- identical to any common definition for non-degenerate rows.
* :func:`kth_largest` -- ``track_semi/chmatch.py:252-253,294-295``: ``bank.topk(int(k))[0][-1]`` for a 1-D score
  bank, by device radix select (``mi_kth_largest``), no sort.

No CPU path: tensors must be on a HIP device."""
from __future__ import annotations

import torch

from . import _lib
from .knn import BankBuilder, knn_topk

__all__ = ["neighbor_label_agreement", "kth_largest"]


def _need(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a tensor on an MI355X (HIP) device: image-classification_amd has no CPU fallback")


@torch.no_grad()
def neighbor_label_agreement(feature: torch.Tensor, mb_feature: torch.Tensor, mb_pred: torch.Tensor,
                             noisy_labels: torch.Tensor, k: int = 10, eps: float = 1e-8, compute_dtype="fp32"):
    """Returns ``(eq_cnt [b] int64, values [b, k], indices [b, k])`` of ``betacs_v2.py:159-162``;
    the caller's ``neighbor_mask`` is ``eq_cnt > 5``."""
    for n, t in (("feature", feature), ("mb_feature", mb_feature), ("mb_pred", mb_pred), ("noisy_labels", noisy_labels)):
        _need(t, n)
    K, D = mb_feature.shape
    bank = BankBuilder(K, D, mb_feature.device, compute_dtype=compute_dtype, eps=eps)
    bank.append(mb_feature)
    qn = BankBuilder(feature.shape[0], D, feature.device, compute_dtype="fp32", eps=eps)
    qn.append(feature)
    values, indices = knn_topk(qn.data[:, :D], bank.finish(), int(k))
    mb_p_label = mb_pred[indices].argmax(dim=-1)                               # betacs_v2.py:161
    eq_cnt = torch.eq(noisy_labels.unsqueeze(1), mb_p_label).sum(dim=-1)       # betacs_v2.py:162
    return eq_cnt, values, indices


@torch.no_grad()
def kth_largest(bank: torch.Tensor, k: int) -> torch.Tensor:
    """``bank.topk(k)[0][-1]`` (0-dim float32 tensor) of a 1-D bank; raises like ``topk`` when k is out of range."""
    _need(bank, "bank")
    if bank.dim() != 1:
        raise ValueError(f"expected a 1-D bank, got shape {tuple(bank.shape)}")
    k = int(k)
    if k < 1 or k > bank.numel():
        raise RuntimeError(f"selected index k out of range (k={k}, n={bank.numel()})")
    x = bank.detach().float().contiguous()
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().mi_kth_largest(x.data_ptr(), x.numel(), k, out.data_ptr(), _lib.stream_ptr(x.device)),
                   "mi_kth_largest")
    return out.reshape(())
