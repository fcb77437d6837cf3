"""Weighted-kNN monitor on MI355X -- host-side mirror of the reference's ``models/knn.py``.

``knn_predict(feature, feature_bank, feature_labels, classes, knn_k, knn_t)`` keeps the
reference's positional signature and return contract (``models/knn.py:4-21``; called from
``track_ssl/ssltrainer.py:184-185``): a ``LongTensor[B, classes]`` holding, per query, the class
ids ranked by exp(sim/T)-weighted vote of the k most similar bank rows.  Only column 0 is consumed
by the caller (``ssltrainer.py:187``).

What differs from the reference is *how*: the [B, N] similarity matrix, the [B*k, C] one-hot
matrix and their temporaries are never materialised.  The arithmetic runs in hand-written HIP
kernels behind ``libmi_knn.so`` (``include/mi_knn.h``): a fused MFMA similarity GEMM + streaming
top-k (``mi_knn_topk``) followed by an LDS vote/ranking kernel (``mi_knn_vote``).

There is no CPU path here: tensors must live on a HIP device and the library must be built.
"""
from __future__ import annotations

import os
import weakref
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import _lib

__all__ = ["knn_predict", "knn_predict_full", "knn_topk", "knn_topk_static", "knn_vote", "knn_merge", "check_labels", "BankIndex", "BankBuilder",
           "pack_rows", "DEFAULT_COMPUTE_DTYPE"]

# "fp32": exact float32 products, one k-ascending fma chain per similarity (the parity mode);
# "bf16": inputs rounded to bfloat16, bf16 MFMA with float32 accumulation (the throughput mode).
DEFAULT_COMPUTE_DTYPE = os.environ.get("MI_KNN_DTYPE", "fp32")

# Candidate workspace of one fused top-k call (GiB).  The filter stages are sized by it (a stage's worst case -- every
# similarity passing -- must fit), so more memory = fewer, larger stages = fewer threshold refreshes; 16 of the 288 GB.
WORKSPACE_GIB = int(os.environ.get("MI_KNN_WORKSPACE_GIB", "16"))

_DT = {"fp32": _lib.MI_F32, "f32": _lib.MI_F32, "float32": _lib.MI_F32,
       "bf16": _lib.MI_BF16, "bfloat16": _lib.MI_BF16}
_TORCH_DT = {_lib.MI_F32: torch.float32, _lib.MI_BF16: torch.bfloat16}


def _dtype_code(compute_dtype) -> int:
    key = DEFAULT_COMPUTE_DTYPE if compute_dtype is None else compute_dtype
    if isinstance(key, torch.dtype):
        key = {torch.float32: "fp32", torch.bfloat16: "bf16"}.get(key, str(key))
    try:
        return _DT[str(key).lower()]
    except KeyError:
        raise ValueError(f"compute_dtype must be 'fp32' or 'bf16', got {key!r}") from None


def _require_device(t: torch.Tensor, name: str) -> None:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} is on {t.device}: image-classification_amd computes on an MI355X (HIP) device "
            "only and has no CPU fallback; move the tensor to the GPU")


@dataclass
class PackedRows:
    """K-contiguous rows in the compute layout: ``data`` is [rows, ld] of the compute dtype."""
    data: torch.Tensor
    rows: int
    dim: int
    dtype_code: int

    @property
    def ld(self) -> int:
        return self.data.shape[1]


def pack_rows(x: torch.Tensor, transposed: bool, compute_dtype=None) -> PackedRows:
    """Bring a float matrix into the kernels' layout (``mi_pack_rows``).

    ``transposed=False``: ``x`` is [rows, D].  ``transposed=True``: ``x`` is [D, rows] (the layout
    ``ssltrainer.py:179`` hands to ``knn_predict``).  Zero-copy when ``x`` (or ``x.t()``) already is
    a 16-byte aligned, contiguous [rows, D] matrix of the compute dtype with D a multiple of the
    K-step (32 fp32 / 64 bf16 elements)."""
    _require_device(x, "matrix")
    if x.dim() != 2:
        raise ValueError(f"expected a 2-D matrix, got shape {tuple(x.shape)}")
    code = _dtype_code(compute_dtype)
    L = _lib.lib()
    rows, D = (x.shape[1], x.shape[0]) if transposed else (x.shape[0], x.shape[1])
    if rows == 0 or D == 0:
        raise ValueError(f"empty matrix of shape {tuple(x.shape)}")
    ld = int(L.mi_knn_ld(D, code))
    view = x.t() if transposed else x
    if view.dtype == _TORCH_DT[code] and view.is_contiguous() and ld == D and view.data_ptr() % 16 == 0:
        return PackedRows(view, rows, D, code)
    src = x if x.dtype == torch.float32 else x.float()
    if src.stride(1) != 1 and src.stride(0) == 1:
        # a transposed view of a contiguous matrix: pack from the underlying layout instead
        src, transposed = src.t(), not transposed
    if src.stride(1) != 1:
        src = src.contiguous()
    out = torch.empty((rows, ld), dtype=_TORCH_DT[code], device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(L.mi_pack_rows(src.data_ptr(), src.stride(0), 1 if transposed else 0, rows, D, code,
                                  out.data_ptr(), ld, _lib.stream_ptr(x.device)), "mi_pack_rows")
    return PackedRows(out, rows, D, code)


class BankIndex:
    """A feature bank (or one row shard of it) resident in HBM in the kernels' layout.

    Built once per kNN pass from [N, D] rows (natural output of the memo loop,
    ``ssltrainer.py:172-177``) or from the reference's transposed [D, N] matrix; queried many
    times.  ``idx_base`` is the global row of local row 0 (row-sharded banks)."""

    def __init__(self, packed: PackedRows, idx_base: int = 0):
        self.packed = packed
        self.idx_base = int(idx_base)

    @classmethod
    def from_rows(cls, bank_rows: torch.Tensor, compute_dtype=None, idx_base: int = 0) -> "BankIndex":
        return cls(pack_rows(bank_rows, transposed=False, compute_dtype=compute_dtype), idx_base)

    @classmethod
    def from_reference_layout(cls, feature_bank: torch.Tensor, compute_dtype=None, idx_base: int = 0) -> "BankIndex":
        return cls(pack_rows(feature_bank, transposed=True, compute_dtype=compute_dtype), idx_base)

    @property
    def n(self) -> int:
        return self.packed.rows

    @property
    def dim(self) -> int:
        return self.packed.dim

    @property
    def device(self):
        return self.packed.data.device


class BankBuilder:
    """Builds a :class:`BankIndex` batch by batch, the way ``knn_test`` walks the memo loader
    (``ssltrainer.py:172-179``) -- without the reference's ``F.normalize`` pass, ``torch.cat`` and
    ``.t().contiguous()`` transpose:

        builder = BankBuilder(n_rows, feature_dim, device, compute_dtype="bf16")
        for batch in memo_loader:
            builder.append(to_feature(batch["xs"]))        # L2-normalised and packed in one kernel
        pred = knn_predict(test_feature, builder.finish(), feature_labels, classes, 200, 0.1)

    ``idx_base`` makes the result one row shard of a larger bank."""

    def __init__(self, n_rows: int, dim: int, device, compute_dtype=None, idx_base: int = 0, eps: float = 1e-12):
        self.code = _dtype_code(compute_dtype)
        self.n_rows, self.dim, self.idx_base, self.eps = int(n_rows), int(dim), int(idx_base), float(eps)
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError(f"BankBuilder needs an MI355X (HIP) device, got {device}: there is no CPU fallback")
        self.ld = int(_lib.lib().mi_knn_ld(self.dim, self.code))
        self.data = torch.empty((self.n_rows, self.ld), dtype=_TORCH_DT[self.code], device=device)
        self.filled = 0

    def append(self, feature: torch.Tensor) -> None:
        _require_device(feature, "feature")
        if feature.dim() != 2 or feature.shape[1] != self.dim:
            raise ValueError(f"feature must be [b, {self.dim}], got {tuple(feature.shape)}")
        src = feature.detach()
        if src.dtype != torch.float32:
            src = src.float()
        if src.stride(1) != 1:
            src = src.contiguous()
        with torch.cuda.device(self.data.device):
            _lib.check(_lib.lib().mi_bank_append_normalized(src.data_ptr(), src.stride(0), src.shape[0], self.dim, self.code,
                                                            self.data.data_ptr(), self.ld, self.n_rows, self.filled, self.eps,
                                                            _lib.stream_ptr(self.data.device)), "mi_bank_append_normalized")
        self.filled += src.shape[0]

    def finish(self) -> "BankIndex":
        if self.filled != self.n_rows:
            raise RuntimeError(f"bank holds {self.filled} of {self.n_rows} rows")
        return BankIndex(PackedRows(self.data, self.n_rows, self.dim, self.code), self.idx_base)


def _topk_workspace(L, B: int, N: int, k: int, dev) -> Tuple[torch.Tensor, int]:
    """Candidate workspace of one fused top-k call.  The result never depends on its size (only the number of filter
    stages does), so when the preferred budget (``MI_KNN_WORKSPACE_GIB``, 16 GiB of the 288 GB) does not fit beside
    what a training job already holds, smaller plans are tried -- the library's default 8 GiB plan, then 1 GiB, then
    the minimum (one 2048-column stage) -- before the allocation failure is allowed to propagate."""
    b_pad = -(-B // 256) * 256
    minimum = 2048 * (b_pad * 8 + b_pad // 128 + 1) + 1024 + 4 * b_pad
    budgets, seen = [], set()
    for budget in (WORKSPACE_GIB << 30, 8 << 30, 1 << 30, minimum):
        nbytes = int(L.mi_knn_topk_workspace_bytes_for_budget(B, N, k, max(budget, minimum)))
        if nbytes == 0:
            raise ValueError(f"unsupported kNN shape B={B} N={N} k={k} (1 <= k <= {_lib.MI_MAX_K})")
        if nbytes not in seen:
            seen.add(nbytes)
            budgets.append(nbytes)
    for i, nbytes in enumerate(budgets):
        try:
            return torch.empty(nbytes, dtype=torch.uint8, device=dev), nbytes
        except torch.OutOfMemoryError:
            if i + 1 == len(budgets):
                raise
    raise AssertionError("unreachable")


def knn_topk(feature: torch.Tensor, bank: BankIndex, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Fused similarity + top-k (``models/knn.py:6,8``) -> (values [B, k] f32 desc, indices [B, k] i64).

    Ties go to the lower bank row.  Raises like ``torch.topk`` when ``k > N``."""
    _require_device(feature, "feature")
    if feature.dim() != 2 or feature.shape[1] != bank.dim:
        raise ValueError(f"feature must be [B, {bank.dim}], got {tuple(feature.shape)}")
    if feature.device != bank.device:
        raise RuntimeError("feature and bank are on different devices")
    k = int(k)
    if k > bank.n:
        raise RuntimeError(f"selected index k out of range (k={k} > N={bank.n})")
    q = pack_rows(feature, transposed=False, compute_dtype={0: "fp32", 1: "bf16"}[bank.packed.dtype_code])
    L = _lib.lib()
    B = q.rows
    dev = feature.device
    ws, ws_bytes = _topk_workspace(L, B, bank.n, k, dev)
    val = torch.empty((B, k), dtype=torch.float32, device=dev)
    idx = torch.empty((B, k), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.mi_knn_topk(q.data.data_ptr(), q.ld, bank.packed.data.data_ptr(), bank.packed.ld,
                                 q.dtype_code, B, bank.n, bank.dim, k, bank.idx_base,
                                 val.data_ptr(), idx.data_ptr(), ws.data_ptr(), ws_bytes,
                                 _lib.stream_ptr(dev)), "mi_knn_topk")
    return val, idx


def knn_topk_static(q: PackedRows, bank: BankIndex, k: int, out_val: torch.Tensor, out_idx: torch.Tensor) -> None:
    """``knn_topk`` for callers that cannot allocate (HIP graph capture, serving loops): ``mi_knn_topk_static``.

    Nothing is allocated here or in the library: ``q`` are queries already packed by :func:`pack_rows` in the bank's
    compute dtype, ``out_val`` [B, k] float32 and ``out_idx`` [B, k] int64 are the caller's buffers, and the
    workspace is the library's fixed 64 MiB static scratch (about 18 KiB per query row at k = 200: B <= 3584).
    Calls that overlap on one GPU share that scratch and race -- keep them on one stream."""
    if not isinstance(q, PackedRows) or q.dtype_code != bank.packed.dtype_code or q.dim != bank.dim:
        raise ValueError("q must be PackedRows of the bank's compute dtype and dimension (pack_rows)")
    B, k = q.rows, int(k)
    dev = q.data.device
    for t, dt, name in ((out_val, torch.float32, "out_val"), (out_idx, torch.int64, "out_idx")):
        if t.dtype != dt or tuple(t.shape) != (B, k) or not t.is_contiguous() or t.device != dev:
            raise ValueError(f"{name} must be a contiguous [{B}, {k}] {dt} tensor on {dev}")
    if k > bank.n:
        raise RuntimeError(f"selected index k out of range (k={k} > N={bank.n})")
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().mi_knn_topk_static(q.data.data_ptr(), q.ld, bank.packed.data.data_ptr(), bank.packed.ld,
                                                 q.dtype_code, B, bank.n, bank.dim, k, bank.idx_base,
                                                 out_val.data_ptr(), out_idx.data_ptr(), _lib.stream_ptr(dev)),
                   "mi_knn_topk_static")


def knn_merge(cand_val: torch.Tensor, cand_idx: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Deterministic merge of per-shard candidates [G, B, kc] -> [B, k] (value desc, index asc)."""
    _require_device(cand_val, "cand_val")
    if cand_val.dim() != 3 or cand_val.shape != cand_idx.shape:
        raise ValueError("cand_val / cand_idx must both be [G, B, kc]")
    cand_val = cand_val.contiguous().float()
    cand_idx = cand_idx.contiguous().to(torch.int64)
    G, B, kc = cand_val.shape
    dev = cand_val.device
    val = torch.empty((B, k), dtype=torch.float32, device=dev)
    idx = torch.empty((B, k), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().mi_knn_merge(cand_val.data_ptr(), cand_idx.data_ptr(), G, B, kc, int(k),
                                           val.data_ptr(), idx.data_ptr(), _lib.stream_ptr(dev)), "mi_knn_merge")
    return val, idx


# label tensors already validated: id(tensor) -> (weak reference, tensor version, min label, max label)
_LABELS_OK: Dict[int, tuple] = {}


def check_labels(feature_labels: torch.Tensor, classes: int) -> None:
    """The reference raises (index error in ``scatter``, ``models/knn.py:16``) when a bank label is outside
    ``[0, classes)``; the vote kernel would silently give such a neighbour no vote.  One fused min/max over the labels
    and one host read -- ONCE per label tensor: the (min, max) of a tensor object is remembered until the tensor is
    modified in place (its version counter) or dies, so a caller that votes chunk by chunk, or every monitor pass over
    the same ``feature_labels``, pays the pass and the host synchronisation once.  Skipped while the stream is being
    captured into a HIP graph (no host read is possible there)."""
    if feature_labels.numel() == 0:
        return
    ent = _LABELS_OK.get(id(feature_labels))
    if ent is not None and ent[0]() is feature_labels and ent[1] == feature_labels._version:
        lo, hi = ent[2], ent[3]
    else:
        if feature_labels.is_cuda and torch.cuda.is_current_stream_capturing():
            return
        lo, hi = (int(v) for v in torch.aminmax(feature_labels))
        if len(_LABELS_OK) >= 64:
            _LABELS_OK.clear()
        _LABELS_OK[id(feature_labels)] = (weakref.ref(feature_labels), feature_labels._version, lo, hi)
    if lo < 0 or hi >= int(classes):
        raise IndexError(f"feature_labels span [{lo}, {hi}] but classes={int(classes)}: every label must lie in "
                         f"[0, classes) (models/knn.py:16 raises here too)")


def knn_vote(sim_weight: torch.Tensor, sim_indices: torch.Tensor, feature_labels: torch.Tensor,
             classes: int, knn_t: float, return_scores: bool = False, check: bool = True):
    """Label gather, exp(s/T) weights, class vote, ranking (``models/knn.py:10-20``).

    ``sim_indices`` address ``feature_labels`` (entries < 0 are padding and get no vote); ``check`` validates the
    label range first (:func:`check_labels`), as the reference's ``scatter`` would."""
    _require_device(sim_weight, "sim_weight")
    dev = sim_weight.device
    B, k = sim_weight.shape
    if feature_labels.dim() != 1:
        raise ValueError("feature_labels must be 1-D")
    if check:
        check_labels(feature_labels, classes)           # the caller's tensor object: validated once (see check_labels)
    labels = feature_labels.to(device=dev, dtype=torch.int64).contiguous()
    sim_weight = sim_weight.contiguous()
    sim_indices = sim_indices.contiguous()
    rank = torch.empty((B, int(classes)), dtype=torch.int64, device=dev)
    scores = torch.empty((B, int(classes)), dtype=torch.float32, device=dev) if return_scores else None
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().mi_knn_vote(sim_weight.data_ptr(), sim_indices.data_ptr(), labels.data_ptr(),
                                          labels.numel(), B, k, int(classes), float(knn_t),
                                          scores.data_ptr() if return_scores else None, rank.data_ptr(),
                                          _lib.stream_ptr(dev)), "mi_knn_vote")
    return (rank, scores) if return_scores else rank


def _as_bank(feature_bank, compute_dtype) -> BankIndex:
    if isinstance(feature_bank, BankIndex):
        return feature_bank
    _require_device(feature_bank, "feature_bank")
    if feature_bank.dim() != 2:
        raise ValueError(f"feature_bank must be [D, N], got {tuple(feature_bank.shape)}")
    return BankIndex.from_reference_layout(feature_bank, compute_dtype)


def knn_predict_full(feature, feature_bank, feature_labels, classes, knn_k, knn_t, *, compute_dtype=None):
    """``knn_predict`` plus the intermediates the reference discards (top-k and vote scores)."""
    _require_device(feature, "feature")
    bank = _as_bank(feature_bank, compute_dtype)
    if feature_labels.numel() != bank.n:
        raise ValueError(f"feature_labels has {feature_labels.numel()} entries for a bank of {bank.n} rows")
    val, idx = knn_topk(feature, bank, knn_k)
    rank, scores = knn_vote(val, idx - bank.idx_base if bank.idx_base else idx, feature_labels,
                            classes, knn_t, return_scores=True)
    return {"sim_weight": val, "sim_indices": idx, "pred_scores": scores, "pred_labels": rank}


def knn_predict(feature, feature_bank, feature_labels, classes, knn_k, knn_t, *, compute_dtype=None):
    """Drop-in for ``models.knn.knn_predict`` (same positional arguments, same result contract).

    feature        [B, D] float, L2-normalised by the caller (``ssltrainer.py:142``)
    feature_bank   [D, N] float, the physically transposed bank of ``ssltrainer.py:179``
                   (a ``BankIndex`` is also accepted to skip re-packing between calls)
    feature_labels [N] integer class ids
    returns        LongTensor[B, classes]: class ids by vote score, descending
    """
    _require_device(feature, "feature")
    bank = _as_bank(feature_bank, compute_dtype)
    if feature_labels.numel() != bank.n:
        raise ValueError(f"feature_labels has {feature_labels.numel()} entries for a bank of {bank.n} rows")
    val, idx = knn_topk(feature, bank, knn_k)
    if bank.idx_base:
        idx = idx - bank.idx_base
    return knn_vote(val, idx, feature_labels, classes, knn_t)
