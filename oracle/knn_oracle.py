"""CPU oracle for the feature-bank similarity hot path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the reference algorithm.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the
product package (``image-classification_amd/``) never does and fails loudly when its
HIP library is missing.

Parity pin: every function here is checked against golden vectors generated in the build
container by importing the reference itself (``tests/golden/make_golden.py`` ->
``tests/golden/*.npz``; test: ``tests/test_oracle_golden.py``).  The MoCo loss
(``moco_loss``) has no reference source in the repository (it lives in the third-party
``lumo`` package, absent here) and is therefore **parity unpinned**: it restates the
published MoCo InfoNCE form that the call site ``track_ssl/moco.py:152-156`` implies.

Reference lines followed (all under /root/reference):
  * ``models/knn.py:6``      sim = feature @ feature_bank           -> :func:`similarity`
  * ``models/knn.py:8``      topk(k) values desc + indices          -> :func:`topk_desc`
  * ``models/knn.py:10``     label gather                            -> :func:`knn_predict_full`
  * ``models/knn.py:11``     w = exp(sim / T)  (division, then exp)  -> :func:`vote_weights`
  * ``models/knn.py:14-18``  one-hot weighted class vote             -> :func:`vote_scores`
  * ``models/knn.py:20``     argsort(descending)                     -> :func:`rank_classes`
  * ``models/memory_bank.py:14-32``  StorageBank                     -> :class:`StorageBankOracle`
  * ``models/memory_bank.py:43-72``  MemoryBank (truncating ring)    -> :class:`MemoryBankOracle`
  * ``models/module_utils.py:48-148`` wrapping MemoryBank / LongTensorMemoryBank -> :class:`WrappingQueueOracle`
  * ``track_semi/comatch.py:187-189`` soft-kNN pseudo-label smoothing        -> :func:`comatch_smooth_prob`
  * ``track_ssl/ssltrainer.py:166-190`` harness (normalise, transpose, column-0 accuracy)
                                                                     -> :func:`knn_monitor_accuracy`
  * ``track_ssl/moco.py:103-106,152-163`` queue init / logits / push-before-backward
                                                                     -> :func:`moco_queue_init`, :func:`moco_loss`

Defined refinements (the reference leaves these unspecified):
  * ties in top-k are broken towards the LOWER bank index; ties in the class ranking towards
    the LOWER class index.  torch's CPU ``topk``/``argsort`` do not promise any order there,
    so tests compare tie rows by value, not by index.
  * the vote for one class is summed over neighbours in ascending neighbour rank j.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------- helpers
def bf16_round(x: np.ndarray) -> np.ndarray:
    """Round float32 to the nearest-even bfloat16 and return it upcast to float32.

    Oracle for bf16 mode = the reference fed bf16-rounded-then-upcast inputs."""
    x = np.ascontiguousarray(x, dtype=F32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u >> 16) & 1) + 0x7FFF
    out = ((u + r) & 0xFFFF0000).astype(np.uint32)
    return out.view(F32).reshape(x.shape)


def to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """uint16 bfloat16 bit patterns (round-to-nearest-even) of a float32 array."""
    return (bf16_round(x).view(np.uint32) >> 16).astype(np.uint16)


def from_bf16_bits(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(F32)


def l2_normalize(x: np.ndarray, eps: float = 1e-12) -> np.ndarray:
    """F.normalize(x, dim=1) as used at ssltrainer.py:142,176."""
    x = np.asarray(x, dtype=F32)
    n = np.sqrt((x.astype(np.float64) ** 2).sum(axis=1, keepdims=True)).astype(F32)
    return (x / np.maximum(n, F32(eps))).astype(F32)


# --------------------------------------------------------------------------- kNN (models/knn.py)
def similarity(feature: np.ndarray, feature_bank: np.ndarray) -> np.ndarray:
    """knn.py:6 -- ``[B, D] @ [D, N] -> [B, N]`` in float32 (BLAS accumulation order)."""
    return np.matmul(np.asarray(feature, F32), np.asarray(feature_bank, F32)).astype(F32)


def topk_desc(sim: np.ndarray, k: int):
    """knn.py:8 -- per-row k largest, sorted descending; ties -> lower index first."""
    sim = np.asarray(sim, F32)
    if k > sim.shape[1]:
        raise ValueError("selected index k out of range")  # torch raises RuntimeError here
    order = np.argsort(-sim, axis=1, kind="stable")[:, :k]
    vals = np.take_along_axis(sim, order, axis=1)
    return vals, order.astype(np.int64)


def vote_weights(sim_topk: np.ndarray, knn_t: float) -> np.ndarray:
    """knn.py:11 -- true division by T in float32, then float32 exp."""
    return np.exp((np.asarray(sim_topk, F32) / F32(knn_t)).astype(F32)).astype(F32)


def vote_scores(weights: np.ndarray, labels_topk: np.ndarray, classes: int) -> np.ndarray:
    """knn.py:14-18 -- score[b, c] = sum_j w[b, j] * [label[b, j] == c], float32, j ascending."""
    B, k = weights.shape
    scores = np.zeros((B, classes), dtype=F32)
    rows = np.arange(B)
    for j in range(k):
        scores[rows, labels_topk[:, j]] += weights[:, j]
    return scores


def rank_classes(scores: np.ndarray) -> np.ndarray:
    """knn.py:20 -- full descending class ranking; ties -> lower class index first."""
    return np.argsort(-np.asarray(scores, F32), axis=1, kind="stable").astype(np.int64)


def knn_from_similarity(sim, feature_labels, classes, knn_k, knn_t):
    vals, idx = topk_desc(sim, knn_k)
    lbl = np.asarray(feature_labels, np.int64)[idx]                 # knn.py:10
    w = vote_weights(vals, knn_t)
    scores = vote_scores(w, lbl, classes)
    return {
        "sim_weight": vals, "sim_indices": idx, "sim_labels": lbl,
        "pred_scores": scores, "pred_labels": rank_classes(scores),
    }


def knn_predict_full(feature, feature_bank, feature_labels, classes, knn_k, knn_t):
    """All intermediates of knn.py:4-21.  ``feature_bank`` is [D, N] like the reference."""
    return knn_from_similarity(similarity(feature, feature_bank), feature_labels,
                               classes, knn_k, knn_t)


def knn_predict(feature, feature_bank, feature_labels, classes, knn_k, knn_t):
    """Same signature and return contract as ``models.knn.knn_predict`` (LongTensor[B, classes])."""
    return knn_predict_full(feature, feature_bank, feature_labels, classes, knn_k, knn_t)["pred_labels"]


def merge_topk(cand_val: np.ndarray, cand_idx: np.ndarray, k: int):
    """Deterministic merge of per-shard candidates: [G, B, k'] -> [B, k].

    Order is total: (score descending, global index ascending); entries with idx < 0 are
    padding.  SURVEY 8(e)."""
    G, B, kk = cand_val.shape
    v = np.transpose(cand_val, (1, 0, 2)).reshape(B, G * kk).astype(F32)
    i = np.transpose(cand_idx, (1, 0, 2)).reshape(B, G * kk).astype(np.int64)
    v = np.where(i < 0, -np.inf, v).astype(F32)
    big = np.iinfo(np.int64).max
    isort = np.where(i < 0, big, i)
    out_v = np.empty((B, k), F32)
    out_i = np.empty((B, k), np.int64)
    for b in range(B):
        o = np.lexsort((isort[b], -v[b]))[:k]
        out_v[b], out_i[b] = v[b, o], i[b, o]
    return out_v, out_i


def knn_monitor_accuracy(test_feature, test_ys, bank_rows, bank_labels, classes, knn_k, knn_t):
    """ssltrainer.py:166-190 -- bank rows are L2-normalised, transposed to [D, N], predicted,
    and column 0 of the ranking is compared with the test labels (accuracy in percent)."""
    bank = np.ascontiguousarray(l2_normalize(bank_rows).T)
    pred = knn_predict(test_feature, bank, bank_labels, classes, knn_k, knn_t)
    return float((pred[:, 0] == np.asarray(test_ys)).astype(F32).sum() / pred.shape[0] * 100)


# --------------------------------------------------------------------------- banks (models/memory_bank.py)
class StorageBankOracle:
    """memory_bank.py:8-32.  ``register`` fills with ``ones + inf`` cast to dtype (float -> +inf,
    int64 -> INT64_MIN as observed from the reference's cast); ``scatter`` is ``buf[index] = value``."""

    def __init__(self):
        self.buffers, self.sizes = {}, {}

    def register(self, name, dim, k, dtype=None):
        dtype = np.dtype(dtype or F32)
        shape = (k,) if dim <= 0 else (k, dim)
        if dtype.kind == "f":
            buf = np.full(shape, np.inf, dtype=dtype)
        else:
            buf = np.full(shape, np.iinfo(dtype).min, dtype=dtype)
        self.buffers[name] = buf
        self.sizes[name] = k

    def __getitem__(self, name):
        return self.buffers[name]

    def scatter(self, name, value, index):
        self.buffers[name][np.asarray(index)] = np.asarray(value)


class MemoryBankOracle:
    """memory_bank.py:35-72 -- named ring buffers.  ``push`` TRUNCATES at the end of the buffer
    (overflow rows are dropped) and the pointer becomes ``(ptr + written) % K``."""

    def __init__(self):
        self.buffers, self.offsets, self.sizes = {}, {}, {}

    def register(self, name, dim, k, dtype=None, rng=None):
        rng = rng or np.random.RandomState(0)
        shape = (k,) if dim <= 0 else (k, dim)
        self.buffers[name] = rng.rand(*shape).astype(dtype or F32)   # torch.rand: U[0, 1)
        self.offsets[name] = 0
        self.sizes[name] = k

    def __getitem__(self, name):
        return self.buffers[name]

    def __setitem__(self, name, value):
        self.buffers[name] = np.asarray(value)

    def push(self, name, value):
        assert name in self.offsets
        buf = self.buffers[name]
        value = np.asarray(value)
        assert buf.ndim == value.ndim
        ptr, K = self.offsets[name], self.sizes[name]
        b = value.shape[0]
        if ptr + b > K:
            b = K - ptr
            value = value[:b]
        buf[ptr:ptr + b] = value
        self.offsets[name] = (ptr + b) % K


class WrappingQueueOracle:
    """models/module_utils.py:48-98 (MemoryBank) / :100-148 (LongTensorMemoryBank): the WRAPPING ring.
    ``push`` writes ``item[:right]`` at the cursor and ``item[right:]`` at the front (``:78-85``); a batch
    longer than the queue raises, as the reference's second slice assignment does."""

    def __init__(self, queue_size=65535, feature_dim=128, dtype=F32, rng=None):
        rng = rng or np.random.RandomState(0)
        shape = (queue_size,) if feature_dim is None else (queue_size, feature_dim)
        self.data = rng.rand(*shape).astype(dtype) if np.dtype(dtype).kind == "f" else np.zeros(shape, dtype)
        self.queue_size, self.cursor = queue_size, 0

    def push(self, item):
        item = np.asarray(item)
        isize = len(item)
        if self.cursor + isize > self.queue_size:
            right = self.queue_size - self.cursor
            left = isize - right
            if left > self.queue_size:
                raise RuntimeError("shape mismatch: batch longer than the queue")
            self.data[self.cursor:] = item[:right]
            self.data[:left] = item[right:]
        else:
            self.data[self.cursor:self.cursor + isize] = item
        self.cursor = (self.cursor + isize) % self.queue_size


def ring_push_plan(ptr: int, K: int, b: int):
    """(rows actually written, new pointer) of one truncating push -- memory_bank.py:65-72."""
    w = min(b, K - ptr)
    return w, (ptr + w) % K


# --------------------------------------------------------------------------- CoMatch smoothing (track_semi/comatch.py)
def comatch_smooth_prob(un_w_feature, mb_feature, mb_prob, un_w_probs, temperature, alpha, dtype=F32):
    """comatch.py:187-189 -- A = exp(f @ bank^T / T) (no max subtraction), row-normalised;
    probs = alpha * p + (1 - alpha) * A @ bank_prob.  ``dtype=np.float64`` gives the tolerance reference."""
    f, bf = np.asarray(un_w_feature, dtype), np.asarray(mb_feature, dtype)
    A = np.exp((f @ bf.T) / dtype(temperature)).astype(dtype)
    A = A / A.sum(axis=1, keepdims=True)
    return (dtype(alpha) * np.asarray(un_w_probs, dtype) + dtype(1 - alpha) * (A @ np.asarray(mb_prob, dtype))).astype(dtype)


# --------------------------------------------------------------------------- order statistic / small-k sites
def kth_largest(bank, k: int):
    """chmatch.py:252-253,294-295 -- ``bank.topk(int(k))[0][-1]``: the k-th largest entry of a 1-D bank."""
    x = np.asarray(bank, F32)
    if k < 1 or k > x.size:
        raise ValueError("selected index k out of range")
    return np.sort(x)[::-1][k - 1]


def neighbor_label_agreement(feature, mb_feature, mb_pred, noisy_labels, k=10, eps=1e-8):
    """betacs_v2.py:159-162 -- cosine similarity (PARITY UNPINNED: lumo's batch_cosine_similarity2 is absent;
    a.b / (max(|a|, eps) max(|b|, eps))), top-k, neighbours' argmax label, agreement count with the noisy label."""
    a, b = np.asarray(feature, np.float64), np.asarray(mb_feature, np.float64)
    an = a / np.maximum(np.linalg.norm(a, axis=1, keepdims=True), eps)
    bn = b / np.maximum(np.linalg.norm(b, axis=1, keepdims=True), eps)
    vals, idx = topk_desc((an @ bn.T).astype(F32), k)
    lbl = np.asarray(mb_pred)[idx].argmax(axis=-1)
    return (np.asarray(noisy_labels)[:, None] == lbl).sum(axis=-1), vals, idx


# --------------------------------------------------------------------------- MoCo queue (track_ssl/moco.py)
def moco_queue_init(K: int, D: int, rng=None) -> np.ndarray:
    """moco.py:103-106 -- rows of U[0,1) noise, L2-normalised."""
    rng = rng or np.random.RandomState(0)
    return l2_normalize(rng.rand(K, D).astype(F32))


def moco_loss(query, key, memory, temperature, dtype=np.float64, with_dkey=False):
    """InfoNCE of the MoCo paper as the call site moco.py:152-156 specifies it
    (query_neg=False, key_neg=False, norm=False): PARITY UNPINNED (lumo absent).

      l_pos[i] = q_i . k_i ; l_neg = q @ memory^T ; logits = [l_pos, l_neg] / T
      loss = mean_i ( logsumexp(logits_i) - logits_i[0] )

    Returns (loss, dloss/dquery, lse[b], l_pos[b]) -- plus dloss/dkey with ``with_dkey`` (the symmetric
    branch moco.py:144-148 passes ``key=feat_query``; only the positive logit depends on the key).  The
    gradient w.r.t. memory is not part of the path (the queue holds detached features)."""
    q = np.asarray(query, dtype)
    kk = np.asarray(key, dtype)
    m = np.asarray(memory, dtype)
    b = q.shape[0]
    l_pos = (q * kk).sum(axis=1)
    l_neg = q @ m.T
    logits = np.concatenate([l_pos[:, None], l_neg], axis=1) / dtype(temperature)
    mx = logits.max(axis=1, keepdims=True)
    lse = (mx + np.log(np.exp(logits - mx).sum(axis=1, keepdims=True)))[:, 0]
    loss = (lse - logits[:, 0]).mean()
    p = np.exp(logits - lse[:, None])
    p[:, 0] -= 1.0
    p /= (b * temperature)
    dq = p[:, :1] * kk + p[:, 1:] @ m
    if with_dkey:
        return loss, dq, lse, l_pos, p[:, :1] * q
    return loss, dq, lse, l_pos
