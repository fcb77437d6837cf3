"""Row-sharded kNN across the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference's ``knn_predict`` is single-device and replicated on every rank;
this module is the scale-out the north star asks for and has no reference counterpart:

  * bank rows are split into G contiguous shards, rank g holds rows ``[lo_g, hi_g)``;
  * queries are replicated (as the reference's ``StorageBank`` already leaves them on every rank);
  * every rank runs the fused similarity + top-k over ITS shard, producing ``(score, global row)``
    candidates ``[B, k]``;
  * ONE exchange step per query chunk, issued asynchronously so that chunk i's exchange overlaps
    chunk i+1's GEMM (payload: 8 bytes x k per query and rank -- 10^2..10^3 x smaller than the
    bank traffic).  ``predict`` uses the all-to-all shape (``ncclSend/Recv`` groups through
    ``torch.distributed.all_to_all_single``): rank j receives the candidates of ITS slice of the
    chunk's queries from every rank, merges and votes them once; ``topk`` uses the all-gather
    shape (``ncclAllGather``) and leaves the full top-k on every rank;
  * a deterministic merge by (score desc, global row asc), then the vote; the rankings (or only
    their top-1 column, which is all the caller consumes) are all-gathered at the end.

Because each dot product is computed identically wherever its bank row lives and the merge order
is total, the result is bit-identical for every G (tests: GPU ``test_index_base_and_merge_*`` and
the world_size-2 gloo test).

The compute primitives are injectable (``ops=``) so the orchestration can be exercised on CPU with
gloo; the default ops are the HIP kernels and fail loudly without the library or a GPU.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist

__all__ = ["shard_bounds", "chunk_bounds", "auto_query_chunk", "ShardedKnn", "QuerySlicedKnn", "HipOps"]


def shard_bounds(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row range of ``rank``; the first ``n_total % world`` shards get one extra row."""
    base, extra = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def chunk_bounds(n: int, chunk: int):
    """[lo, hi) query ranges of about ``chunk`` rows.  A short tail (< chunk / 4) is folded into the last
    chunk: a GEMM over a few hundred queries runs far below the rate of a full one and still pays every
    stage's refresh (50 000 = 16384 + 16384 + 17232, not ... + 848)."""
    n, chunk = int(n), int(chunk)
    lows = list(range(0, n, chunk))
    if len(lows) > 1 and n - lows[-1] < chunk // 4:
        lows.pop()
    return [(lo, lows[i + 1] if i + 1 < len(lows) else n) for i, lo in enumerate(lows)]


def auto_query_chunk(n_local: int) -> int:
    """Query rows per fused top-k call when the caller does not choose.  Every chunk pays one threshold bootstrap and
    ramp (small, candidate-dense launches), while a larger chunk leaves fewer bank columns per stage in a given
    candidate workspace.  Large shards need many stages and want them large (16 384-row chunks); a small shard is
    done in a handful of stages whatever their size, so fewer, larger chunks win (measured on 1/2, 1/4 and 1/8 shards
    of the 1.28 M-row bank: 16 384-row chunks are best at 640 k rows, one 50 000-row chunk at 320 k and 160 k)."""
    return 16384 if n_local >= 480_000 else 65536


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
        return self._knn.knn_vote(val, idx, labels, classes, knn_t, check=False)      # validated once per predict()

    def check_labels(self, labels, classes):
        self._knn.check_labels(labels, classes)


# -- collectives: ONE call site each, whatever the backend ------------------------------------
def _on_device(group) -> bool:
    """RCCL (backend "nccl") moves device tensors itself; any other backend (gloo in the CPU tests and in the
    two-ranks-on-one-GPU test) gets the SAME collective call with the SAME split sizes on host copies."""
    return dist.get_backend(group) == "nccl"


def _all_to_all(group, recv: torch.Tensor, send: torch.Tensor, out_splits, in_splits):
    """recv <- all_to_all_single(send); returns the pending work (None once complete)."""
    if _on_device(group):
        return dist.all_to_all_single(recv, send, output_split_sizes=out_splits, input_split_sizes=in_splits,
                                      group=group, async_op=True)
    hrecv = torch.empty(recv.shape, dtype=recv.dtype)
    dist.all_to_all_single(hrecv, send.cpu().contiguous(), output_split_sizes=out_splits,
                           input_split_sizes=in_splits, group=group)
    recv.copy_(hrecv)
    return None


def _all_gather(group, out: torch.Tensor, mine: torch.Tensor, async_op: bool = False):
    """out [G * n, ...] <- all_gather_into_tensor(mine [n, ...]); returns the pending work (None once complete)."""
    if _on_device(group):
        w = dist.all_gather_into_tensor(out, mine.contiguous(), group=group, async_op=async_op)
        return w if async_op else None
    hout = torch.empty(out.shape, dtype=out.dtype)
    dist.all_gather_into_tensor(hout, mine.cpu().contiguous(), group=group)
    out.copy_(hout)
    return None


class ShardedKnn:
    """kNN monitor over a row-sharded bank.

    bank_rows_local  [n_local, D] rows of this rank's shard (natural [N, D] layout)
    idx_base         global row of local row 0 (``shard_bounds(n_total, world, rank)[0]``)
    """

    def __init__(self, bank_rows_local: torch.Tensor, idx_base: int, group=None, ops=None,
                 compute_dtype=None, query_chunk: Optional[int] = 8192, always_exchange: bool = False,
                 shard_sizes=None):
        self.group = group
        self.world = dist.get_world_size(group) if _dist_on() else 1
        self.rank = dist.get_rank(group) if _dist_on() else 0
        self.ops = ops if ops is not None else HipOps(compute_dtype)
        self.idx_base = int(idx_base)
        self.bank = self.ops.build_bank(bank_rows_local, self.idx_base)
        self.n_local = int(self.ops.bank_rows(self.bank))
        self.query_chunk = int(query_chunk) if query_chunk else auto_query_chunk(self.n_local)      # None / 0: by shard size
        # always_exchange: a one-rank group still goes through the collectives and the merge (lets a one-GPU box
        # run the RCCL branches); otherwise a single rank skips them
        self.solo = self.world == 1 and not (always_exchange and _dist_on())
        # shard_sizes: rows of every rank's shard, when the caller knows the partition (shard_bounds): no collective
        # and no host read in the constructor.  Otherwise the sizes are all-gathered once, here.
        if shard_sizes is not None:
            sizes = [int(v) for v in shard_sizes]
            if len(sizes) != (1 if self.solo else self.world) or sizes[self.rank] != self.n_local:
                raise ValueError(f"shard_sizes {sizes} does not describe {self.world} rank(s) with {self.n_local} rows on rank {self.rank}")
        elif self.solo:
            sizes = [self.n_local]
        else:
            t = torch.tensor([self.n_local], dtype=torch.int64, device=bank_rows_local.device)
            out = torch.empty((self.world,), dtype=torch.int64, device=bank_rows_local.device)
            _all_gather(group, out, t)
            sizes = [int(v) for v in out.tolist()]                 # one host read
        self.shard_sizes = sizes
        self.n_total = sum(sizes)

    # -- one exchange step ------------------------------------------------------------------
    def _exchange(self, val: torch.Tensor, idx: torch.Tensor):
        """all-gather [B, k] candidates -> [G, B, k]; returns (gathered_val, gathered_idx, works)."""
        if self.solo:
            return val[None], idx[None], ()
        G = self.world
        gv = torch.empty((G,) + tuple(val.shape), dtype=val.dtype, device=val.device)
        gi = torch.empty((G,) + tuple(idx.shape), dtype=idx.dtype, device=idx.device)
        w1 = _all_gather(self.group, gv.view((G * val.shape[0],) + tuple(val.shape[1:])), val, async_op=True)
        w2 = _all_gather(self.group, gi.view((G * idx.shape[0],) + tuple(idx.shape[1:])), idx, async_op=True)
        return gv, gi, tuple(w for w in (w1, w2) if w is not None)

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
        """Global top-k (values, global rows) for every query, identical on every rank.  Chunk i's all-gather is
        issued asynchronously and runs beside chunk i+1's GEMM."""
        if k > self.n_total:
            raise RuntimeError(f"selected index k out of range (k={k} > N={self.n_total})")
        if self.n_total >= 2 ** 32:
            raise ValueError("mi_knn_merge keys global rows as 32-bit values: N must be < 2^32")
        starts = chunk_bounds(feature.shape[0], self.query_chunk)
        outs = []
        pending = None                                   # exchange in flight
        for s, e in starts:
            val, idx = self._local_candidates(feature[s:e], k)
            nxt = self._exchange(val, idx)
            if pending is not None:
                outs.append(self._finish(pending, k))
            pending = nxt
        outs.append(self._finish(pending, k))
        return torch.cat([o[0] for o in outs], 0), torch.cat([o[1] for o in outs], 0)

    def _finish(self, pending, k):
        gv, gi, works = pending
        for w in works:
            w.wait()
        if self.solo:
            return gv[0], gi[0]
        return self.ops.merge(gv, gi, k)

    # -- query-sliced exchange (predict) ------------------------------------------------------
    def _to_owners(self, val: torch.Tensor, idx: torch.Tensor):
        """ONE exchange step, all-to-all shaped: rank j receives, from every rank, the candidates of
        ITS slice of the chunk's queries (``shard_bounds(c, G, j)``), so merge and vote are done once
        per query instead of once per query and rank, and each rank receives c/G x G x k candidates
        instead of c x G x k.  Payload: (float32 score bits, int32 global row) in one int32 tensor.
        Returns (buffer [G, c_j, 2, k] int32, pending work or None)."""
        G, c, k = self.world, val.shape[0], val.shape[1]
        if self.solo:
            return (val, idx), None                      # one rank, no exchange: the candidates are the result
        lo, hi = shard_bounds(c, G, self.rank)
        send = torch.stack((val.contiguous().view(torch.int32), idx.to(torch.int32)), dim=1)        # [c, 2, k]
        # rows of `send` that go to rank j = rank j's slice of the chunk; every rank sends me MY slice's rows
        in_splits = [shard_bounds(c, G, j)[1] - shard_bounds(c, G, j)[0] for j in range(G)]
        out_splits = [hi - lo] * G
        recv = torch.empty((G * (hi - lo), 2, k), dtype=torch.int32, device=val.device)
        work = _all_to_all(self.group, recv, send, out_splits, in_splits)
        return recv.view(G, hi - lo, 2, k), work

    def _own_slice(self, pending, k, feature_labels, classes, knn_t):
        buf, work = pending
        if work is not None:
            work.wait()
        if self.solo:
            val, idx = buf
            return self.ops.vote(val, idx, feature_labels, classes, knn_t)
        if buf.shape[1] == 0:
            return torch.empty((0, classes), dtype=torch.int64, device=buf.device)
        val = buf[:, :, 0, :].contiguous().view(torch.float32)
        idx = buf[:, :, 1, :].to(torch.int64)
        val, idx = self.ops.merge(val, idx, k)
        return self.ops.vote(val, idx, feature_labels, classes, knn_t)

    @torch.no_grad()
    def predict(self, feature, feature_labels, classes, knn_k, knn_t, gather: str = "full"):
        """The kNN monitor over the sharded bank.  ``feature_labels`` is the FULL [N] label vector
        (replicated, as in the reference).

        gather="full": the reference's result contract on every rank -- LongTensor[B, classes]
                       (the ranking slices are all-gathered: B x classes x 8 bytes per rank).
        gather="top1": LongTensor[B] of the top-1 class per query on every rank -- the only column the
                       caller consumes (``ssltrainer.py:187``); the full rankings stay with the rank
                       that owns the query slice (``self.local_rows``, ``self.local_ranking``)."""
        if gather not in ("full", "top1"):
            raise ValueError("gather must be 'full' or 'top1'")
        if feature_labels.numel() != self.n_total:
            raise ValueError(f"feature_labels has {feature_labels.numel()} entries for a bank of {self.n_total} rows")
        k, classes, knn_t = int(knn_k), int(classes), float(knn_t)
        if k > self.n_total:
            raise RuntimeError(f"selected index k out of range (k={k} > N={self.n_total})")
        if self.n_total >= 2 ** 31:
            raise ValueError("the candidate exchange carries int32 global rows: N must be < 2^31")
        if hasattr(self.ops, "check_labels"):
            self.ops.check_labels(feature_labels, classes)
        B, G = feature.shape[0], self.world
        starts = chunk_bounds(B, self.query_chunk)
        slices, rows = [], []
        pending = None                                   # exchange in flight
        for s, e in starts:
            f = feature[s:e]
            val, idx = self._local_candidates(f, k)
            nxt = self._to_owners(val, idx)              # chunk i's exchange runs beside the next chunk's GEMM
            if pending is not None:
                slices.append(self._own_slice(pending, k, feature_labels, classes, knn_t))
            pending = nxt
            lo, hi = shard_bounds(f.shape[0], G, self.rank)
            rows.append(torch.arange(s + lo, s + hi, device=feature.device))
        slices.append(self._own_slice(pending, k, feature_labels, classes, knn_t))
        self.local_rows = torch.cat(rows)
        self.local_ranking = torch.cat(slices, 0)
        if self.solo:
            return self.local_ranking if gather == "full" else self.local_ranking[:, 0].contiguous()
        # final gather: pad every rank's slice to the same length, then undo the (chunk, owner) interleave
        mine = self.local_ranking if gather == "full" else self.local_ranking[:, :1]
        width = mine.shape[1]
        per_chunk = [[shard_bounds(e - s, G, j) for j in range(G)] for s, e in starts]
        longest = max(sum(hi - lo for lo, hi in (pc[j] for pc in per_chunk)) for j in range(G))
        padded = torch.zeros((longest, width), dtype=torch.int64, device=mine.device)
        padded[:mine.shape[0]] = mine
        allr = torch.empty((G * longest, width), dtype=torch.int64, device=mine.device)
        _all_gather(self.group, allr, padded)
        allr = allr.view(G, longest, width)
        out = torch.empty((B, width), dtype=torch.int64, device=mine.device)
        offs = [0] * G
        for (s, _), pc in zip(starts, per_chunk):
            for j, (lo, hi) in enumerate(pc):
                out[s + lo:s + hi] = allr[j, offs[j]:offs[j] + (hi - lo)]
                offs[j] += hi - lo
        return out if gather == "full" else out[:, 0].contiguous()


class QuerySlicedKnn:
    """The comparison row of SURVEY.md 8(e): every rank holds the WHOLE bank and serves a contiguous slice of the
    queries -- no candidate exchange at all, only the final all-gather of the results.  Not the default: the north
    star prescribes row sharding (and a bank that outgrows one GPU needs it); this form exists so that
    ``bench.py --shard queries`` can put a number beside it.  Results are identical to the single-device path."""

    def __init__(self, bank_rows: torch.Tensor, group=None, ops=None, compute_dtype=None, query_chunk: int = 8192):
        self.group = group
        self.world = dist.get_world_size(group) if _dist_on() else 1
        self.rank = dist.get_rank(group) if _dist_on() else 0
        self.ops = ops if ops is not None else HipOps(compute_dtype)
        self.bank = self.ops.build_bank(bank_rows, 0)
        self.n_total = int(self.ops.bank_rows(self.bank))
        self.query_chunk = int(query_chunk)

    @torch.no_grad()
    def predict(self, feature, feature_labels, classes, knn_k, knn_t, gather: str = "full"):
        if gather not in ("full", "top1"):
            raise ValueError("gather must be 'full' or 'top1'")
        k, classes, knn_t = int(knn_k), int(classes), float(knn_t)
        if k > self.n_total:
            raise RuntimeError(f"selected index k out of range (k={k} > N={self.n_total})")
        if hasattr(self.ops, "check_labels"):
            self.ops.check_labels(feature_labels, classes)
        B, G = feature.shape[0], self.world
        lo, hi = shard_bounds(B, G, self.rank)
        parts = []
        for s, e in chunk_bounds(hi - lo, self.query_chunk):
            val, idx = self.ops.topk(feature[lo + s:lo + e], self.bank, k)
            parts.append(self.ops.vote(val, idx, feature_labels, classes, knn_t))
        mine = torch.cat(parts, 0) if parts else torch.empty((0, classes), dtype=torch.int64, device=feature.device)
        self.local_rows = torch.arange(lo, hi, device=feature.device)
        self.local_ranking = mine
        if gather == "top1":
            mine = mine[:, :1]
        if G == 1:
            return mine if gather == "full" else mine[:, 0].contiguous()
        width = mine.shape[1]
        longest = shard_bounds(B, G, 0)[1]                          # rank 0 holds the longest slice
        padded = torch.zeros((longest, width), dtype=torch.int64, device=mine.device)
        padded[:mine.shape[0]] = mine
        allr = torch.empty((G * longest, width), dtype=torch.int64, device=mine.device)
        _all_gather(self.group, allr, padded)
        allr = allr.view(G, longest, width)
        out = torch.cat([allr[j, :shard_bounds(B, G, j)[1] - shard_bounds(B, G, j)[0]] for j in range(G)], 0)
        return out if gather == "full" else out[:, 0].contiguous()


def _dist_on() -> bool:
    return dist.is_available() and dist.is_initialized()
