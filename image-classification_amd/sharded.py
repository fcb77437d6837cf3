"""Row-sharded kNN across the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference's ``knn_predict`` is single-device and replicated on every rank;
this module is the scale-out the north star asks for and has no reference counterpart:

  * bank rows are split into G contiguous shards, rank g holds rows ``[lo_g, hi_g)``;
  * queries are replicated (as the reference's ``StorageBank`` already leaves them on every rank);
  * every rank runs the fused similarity + top-k over ITS shard, producing ``(score, global row)``
    candidates ``[B, k]``;
  * ONE exchange step: an all-gather of the candidates (``ncclAllGather`` through
    ``torch.distributed``; 8 bytes x k per query and rank -- 10^2..10^3 x smaller than the bank
    traffic), issued per query chunk so that chunk i's exchange overlaps chunk i+1's GEMM;
  * a deterministic merge by (score desc, global row asc) and the vote, replicated on every rank,
    so each rank ends with the same ``LongTensor[B, classes]`` the reference would return.

Because each dot product is computed identically wherever its bank row lives and the merge order
is total, the result is bit-identical for every G (tests: GPU ``test_index_base_and_merge_*`` and
the world_size-2 gloo test).

The compute primitives are injectable (``ops=``) so the orchestration can be exercised on CPU with
gloo; the default ops are the HIP kernels and fail loudly without the library or a GPU.
"""
from __future__ import annotations

import contextlib
from typing import Optional, Tuple

import torch
import torch.distributed as dist

__all__ = ["shard_bounds", "ShardedKnn", "HipOps"]


def shard_bounds(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row range of ``rank``; the first ``n_total % world`` shards get one extra row."""
    base, extra = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class HipOps:
    """Default compute primitives: the HIP kernels behind libmi_knn.so."""

    def __init__(self, compute_dtype=None):
        from . import knn
        self._knn = knn
        self.compute_dtype = compute_dtype

    def build_bank(self, bank_rows, idx_base):
        return self._knn.BankIndex.from_rows(bank_rows, self.compute_dtype, idx_base=idx_base)

    def bank_rows(self, bank):
        return bank.n

    def topk(self, feature, bank, k):
        return self._knn.knn_topk(feature, bank, k)

    def merge(self, cand_val, cand_idx, k):
        return self._knn.knn_merge(cand_val, cand_idx, k)

    def vote(self, val, idx, labels, classes, knn_t):
        return self._knn.knn_vote(val, idx, labels, classes, knn_t)


class ShardedKnn:
    """kNN monitor over a row-sharded bank.

    bank_rows_local  [n_local, D] rows of this rank's shard (natural [N, D] layout)
    idx_base         global row of local row 0 (``shard_bounds(n_total, world, rank)[0]``)
    """

    def __init__(self, bank_rows_local: torch.Tensor, idx_base: int, group=None, ops=None,
                 compute_dtype=None, query_chunk: int = 8192, n_streams: int = 1):
        self.group = group
        self.world = dist.get_world_size(group) if _dist_on() else 1
        self.rank = dist.get_rank(group) if _dist_on() else 0
        self.ops = ops if ops is not None else HipOps(compute_dtype)
        self.idx_base = int(idx_base)
        self.bank = self.ops.build_bank(bank_rows_local, self.idx_base)
        self.n_local = int(self.ops.bank_rows(self.bank))
        self.query_chunk = int(query_chunk)
        self.n_streams = max(1, int(n_streams))
        sizes = [self.n_local]
        if self.world > 1:
            t = torch.tensor([self.n_local], dtype=torch.int64, device=bank_rows_local.device)
            out = [torch.empty_like(t) for _ in range(self.world)]
            dist.all_gather(out, t, group=group)
            sizes = [int(o.item()) for o in out]
        self.shard_sizes = sizes
        self.n_total = sum(sizes)

    # -- one exchange step ------------------------------------------------------------------
    def _exchange(self, val: torch.Tensor, idx: torch.Tensor):
        """all-gather [B, k] candidates -> [G, B, k]; returns (gathered_val, gathered_idx, works)."""
        if self.world == 1:
            return val[None], idx[None], ()
        G = self.world
        gv = torch.empty((G,) + tuple(val.shape), dtype=val.dtype, device=val.device)
        gi = torch.empty((G,) + tuple(idx.shape), dtype=idx.dtype, device=idx.device)
        if dist.get_backend(self.group) == "nccl":
            w1 = dist.all_gather_into_tensor(gv, val.contiguous(), group=self.group, async_op=True)
            w2 = dist.all_gather_into_tensor(gi, idx.contiguous(), group=self.group, async_op=True)
            return gv, gi, (w1, w2)
        # any other backend (gloo in the tests): synchronous, staged through host memory
        hv, hi = val.cpu().contiguous(), idx.cpu().contiguous()
        lv = [torch.empty_like(hv) for _ in range(G)]
        li = [torch.empty_like(hi) for _ in range(G)]
        dist.all_gather(lv, hv, group=self.group)
        dist.all_gather(li, hi, group=self.group)
        gv.copy_(torch.stack(lv))
        gi.copy_(torch.stack(li))
        return gv, gi, ()

    def _local_candidates(self, f: torch.Tensor, k: int):
        """Local top-k padded to k columns with (-inf, -1) when this shard holds fewer than k rows."""
        kl = min(k, self.n_local)
        val, idx = self.ops.topk(f, self.bank, kl)
        if kl < k:
            pv = torch.full((f.shape[0], k - kl), float("-inf"), dtype=val.dtype, device=val.device)
            pi = torch.full((f.shape[0], k - kl), -1, dtype=idx.dtype, device=idx.device)
            val, idx = torch.cat([val, pv], 1), torch.cat([idx, pi], 1)
        return val, idx

    @torch.no_grad()
    def topk(self, feature: torch.Tensor, k: int):
        """Global top-k (values, global rows) for every query, identical on every rank.

        Query chunks alternate between ``n_streams`` HIP streams: the threshold-refresh kernels,
        launch tails, exchange and merge of one chunk run beside the next chunk's GEMM."""
        if k > self.n_total:
            raise RuntimeError(f"selected index k out of range (k={k} > N={self.n_total})")
        B = feature.shape[0]
        starts = list(range(0, B, self.query_chunk))
        side = _side_streams(feature.device, self.n_streams) if (feature.is_cuda and len(starts) > 1) else None
        main = torch.cuda.current_stream(feature.device) if side else None
        outs = [None] * len(starts)
        pending = None                                   # (chunk number, exchange in flight)

        def on_stream(i):
            return torch.cuda.stream(side[i % len(side)]) if side else contextlib.nullcontext()

        for i, s in enumerate(starts):
            if side and i < len(side):
                side[i].wait_stream(main)                # inputs were produced on the caller's stream
            with on_stream(i):
                val, idx = self._local_candidates(feature[s:s + self.query_chunk], k)
                nxt = self._exchange(val, idx)           # chunk i's all-gather runs beside the next chunk's GEMM
            if pending is not None:
                with on_stream(pending[0]):
                    outs[pending[0]] = self._finish(pending[1], k)
            pending = (i, nxt)
        with on_stream(pending[0]):
            outs[pending[0]] = self._finish(pending[1], k)
        if side:
            for st in side[:len(starts)]:
                main.wait_stream(st)
            for v, ix in outs:
                v.record_stream(main)
                ix.record_stream(main)
        return torch.cat([o[0] for o in outs], 0), torch.cat([o[1] for o in outs], 0)

    def _finish(self, pending, k):
        gv, gi, works = pending
        for w in works:
            w.wait()
        if self.world == 1:
            return gv[0], gi[0]
        return self.ops.merge(gv, gi, k)

    @torch.no_grad()
    def predict(self, feature, feature_labels, classes, knn_k, knn_t):
        """Same result contract as ``knn_predict``: LongTensor[B, classes] on every rank.
        ``feature_labels`` is the FULL [N] label vector (replicated, as in the reference)."""
        if feature_labels.numel() != self.n_total:
            raise ValueError(f"feature_labels has {feature_labels.numel()} entries for a bank of {self.n_total} rows")
        val, idx = self.topk(feature, int(knn_k))
        return self.ops.vote(val, idx, feature_labels, int(classes), float(knn_t))


_SIDE = {}


def _side_streams(device, n):
    """Side streams are created once per device and reused (creation synchronises the device)."""
    key = (torch.device(device).index, n)
    if key not in _SIDE:
        _SIDE[key] = [torch.cuda.Stream(device) for _ in range(n)]
    return _SIDE[key]


def _dist_on() -> bool:
    return dist.is_available() and dist.is_initialized()
