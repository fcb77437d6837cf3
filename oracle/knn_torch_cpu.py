"""Torch-CPU port of the reference kNN monitor, used ONLY as the timed ``cpu_baseline`` in bench.py
and as a gloo-test backend.  TEST INFRASTRUCTURE ONLY (see knn_oracle.py header).

Same torch ops in the same order as ``models/knn.py:6-20`` (mm, topk, gather, div+exp, one-hot
scatter, weighted sum, argsort) so the timing is what the reference would cost on these host
cores; the only structural change is an outer loop over query chunks, because at the bench sizes
(50 000 x 1 280 000) the reference's un-chunked [B, N] similarity matrix (256 GB) cannot be
allocated.  Pinned against tests/golden/*.npz in tests/test_oracle_golden.py.
"""
from __future__ import annotations

import torch


@torch.no_grad()
def knn_predict_chunked(feature, feature_bank, feature_labels, classes, knn_k, knn_t, chunk=None):
    """feature [B, D], feature_bank [D, N] (reference layout), labels [N] -> LongTensor [B, classes]."""
    B = feature.shape[0]
    chunk = B if chunk is None else int(chunk)
    out = []
    for s in range(0, B, chunk):
        f = feature[s:s + chunk]
        n = f.shape[0]
        sim = torch.mm(f, feature_bank)                                            # knn.py:6
        w, idx = sim.topk(k=knn_k, dim=-1)                                         # knn.py:8
        lbl = torch.gather(feature_labels.expand(n, -1), dim=-1, index=idx)        # knn.py:10
        w = (w / knn_t).exp()                                                      # knn.py:11
        hot = torch.zeros(n * knn_k, classes, device=lbl.device)                   # knn.py:14
        hot = hot.scatter(dim=-1, index=lbl.view(-1, 1), value=1.0)                # knn.py:16
        scores = torch.sum(hot.view(n, -1, classes) * w.unsqueeze(dim=-1), dim=1)  # knn.py:18
        out.append(scores.argsort(dim=-1, descending=True))                        # knn.py:20
    return torch.cat(out, dim=0)


@torch.no_grad()
def topk_rows(feature, bank_rows, k):
    """CPU stand-in for the fused top-k with the build's tie rule (value desc, row asc);
    used as the local-compute backend of the world_size-2 gloo tests."""
    sim = torch.mm(feature.float(), bank_rows.float().t())
    order = torch.argsort(-sim, dim=1, stable=True)[:, :k]
    return torch.gather(sim, 1, order), order
