"""MoCo queue logits + InfoNCE on MI355X -- the arithmetic behind ``track_ssl/moco.py:152-156``.

The reference calls the third-party ``lumo.contrib.nn.loss.contrastive_loss2`` (source absent
from the repository: parity unpinned).  The call site fixes the contract:

    Lcs = contrastive_loss2(query=feat_query, key=feat_key.detach(),
                            memory=mem['negative'].clone().detach(),
                            query_neg=False, key_neg=False, temperature=0.1, norm=False)

``query`` [b, D] requires grad, ``key`` [b, D] and ``memory`` [K, D] are detached; only the queue
supplies negatives.  The symmetric branch (``moco.py:138-149``) also calls it with ``query=feat_key`` (no grad) and
``key=feat_query`` (requires grad): the gradient w.r.t. ``key`` is returned as well.  MoCo semantics: ``l_pos = q.k``, ``l_neg = q @ memory^T``,
``loss = mean CE([l_pos, l_neg] / T, target 0)``.

:func:`contrastive_loss2` keeps those keyword names for that subset.  The [b, K+1] logits are never
written: the forward is a fused fp32-MFMA GEMM + online log-sum-exp (``mi_moco_fwd``), the backward
recomputes the probabilities tile by tile (``mi_moco_bwd``).

Push-before-backward (``moco.py:159-172``): the reference clones the whole queue every step because
``mem.push`` overwrites it before ``backward``.  Here ``memory`` may be the LIVE buffer
(``mem['negative'].detach()``, no clone): :class:`MemoryBank.push <..memory_bank.MemoryBank>` asks
:func:`protect_rows` first, which stashes only the rows about to be overwritten for every loss whose
backward is still pending; the backward kernel reads those rows from the stash.
"""
from __future__ import annotations

import weakref
from typing import Dict, List, Optional

import torch

from . import _lib

__all__ = ["contrastive_loss2", "moco_queue_loss", "protect_rows"]


class _Pending:
    """Forward done, backward not yet: what backward needs to see the pre-push queue."""

    __slots__ = ("memory", "patch_lo", "patch_rows", "snapshot", "__weakref__")

    def __init__(self, memory: torch.Tensor):
        self.memory = memory
        self.patch_lo = 0
        self.patch_rows: Optional[torch.Tensor] = None     # [patch_n, D] pre-push rows
        self.snapshot: Optional[torch.Tensor] = None       # full pre-push queue (second push before backward)


# queue storage address -> losses whose backward has not run yet
_PENDING: Dict[int, List["weakref.ReferenceType[_Pending]"]] = {}


def _pending_for(ptr: int) -> List[_Pending]:
    refs = _PENDING.get(ptr)
    if not refs:
        return []
    live = [r() for r in refs]
    live = [p for p in live if p is not None]
    if live:
        _PENDING[ptr] = [weakref.ref(p) for p in live]
    else:
        _PENDING.pop(ptr, None)
    return live


def protect_rows(bank: torch.Tensor, lo: int, n: int) -> Optional[torch.Tensor]:
    """Called by ``MemoryBank.push`` before it overwrites ``bank[lo:lo+n]``.

    Returns a stash buffer [n, ...] that the push must fill with the old rows
    (``mi_queue_push_stash``), or None when no loss over this queue awaits its backward."""
    pend = _pending_for(bank.data_ptr())
    if not pend or n <= 0:
        return None
    fresh = [p for p in pend if p.patch_rows is None and p.snapshot is None]
    for p in pend:
        if p in fresh:
            continue
        if p.snapshot is None:
            # a second push before backward: freeze the whole pre-push queue once
            snap = bank.detach().clone()
            snap[p.patch_lo:p.patch_lo + p.patch_rows.shape[0]] = p.patch_rows
            p.snapshot, p.patch_rows = snap, None
    if not fresh:
        return None
    stash = torch.empty((n,) + tuple(bank.shape[1:]), dtype=bank.dtype, device=bank.device)
    for p in fresh:
        p.patch_lo, p.patch_rows = int(lo), stash
    return stash


# Scratch of mi_moco_fwd / mi_moco_bwd (tile partials), cached per (device, stream): both calls use it as pure scratch
# (nothing a forward writes there is read by a backward), every use is ordered by the stream, so ONE buffer serves every
# loss on that stream instead of a fresh torch.empty per call (the forward is ~45 us of kernels: allocations show).
_SCRATCH: Dict[tuple, torch.Tensor] = {}


def _scratch(dev: torch.device, stream_ptr: int, nbytes: int) -> torch.Tensor:
    key = (dev.index, stream_ptr)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _SCRATCH[key] = buf
    return buf


class _MocoQueueLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, query, key, memory, temperature):
        L = _lib.lib()
        q = query.detach().contiguous().float()
        k = key.detach().contiguous().float()
        if not memory.is_contiguous() or memory.dtype != torch.float32:
            raise ValueError("memory must be a contiguous float32 [K, D] tensor")
        b, D = q.shape
        K = memory.shape[0]
        dev = q.device
        need_dq, need_dkey = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if need_dq and (D % 64 or D > 256):
            # fail HERE, not from inside autograd in the middle of a train step: the fused backward w.r.t. the query
            # (mi_moco_bwd) is built for D % 64 == 0, D <= 256 (the reference's feature_dim is 128)
            raise NotImplementedError(f"the backward w.r.t. query needs D % 64 == 0 and D <= 256, got D={D}")
        ws_bytes = int(L.mi_moco_workspace_bytes(b, K, D))
        stream = _lib.stream_ptr(dev)
        ws = _scratch(dev, stream, ws_bytes)
        out = torch.empty(2 * b + 1, dtype=torch.float32, device=dev)      # lse | lpos | loss: one allocation
        lse, lpos, loss = out[:b], out[b:2 * b], out[2 * b:]
        with torch.cuda.device(dev):
            _lib.check(L.mi_moco_fwd(q.data_ptr(), k.data_ptr(), memory.data_ptr(), b, K, D, float(temperature),
                                     lse.data_ptr(), lpos.data_ptr(), loss.data_ptr(), ws.data_ptr(), ws.numel(),
                                     stream), "mi_moco_fwd")
        ctx.save_for_backward(q, k, lse, lpos)
        ctx.temperature = float(temperature)
        ctx.ws_bytes = ws_bytes if need_dq else 0
        ctx.pending = None
        if need_dq:
            # only the query gradient reads the queue again: pushes must protect its rows until backward has run
            pend = _Pending(memory)
            ctx.pending = pend
            _PENDING.setdefault(memory.data_ptr(), []).append(weakref.ref(pend))
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_loss):
        q, k, lse, lpos = ctx.saved_tensors
        need_dq, need_dkey = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        pend: Optional[_Pending] = ctx.pending
        L = _lib.lib()
        b, D = q.shape
        dev = q.device
        g = grad_loss.detach().reshape(1).float().contiguous()
        dq = torch.empty_like(q) if need_dq else None
        dkey = torch.empty_like(k) if need_dkey else None
        memory = patch = None
        K = 1
        if need_dq:
            memory = pend.snapshot if pend.snapshot is not None else pend.memory
            K = memory.shape[0]
            patch = pend.patch_rows
        stream = _lib.stream_ptr(dev)
        ws = _scratch(dev, stream, ctx.ws_bytes) if need_dq else None      # the stream backward runs on (autograd replays the forward's)
        with torch.cuda.device(dev):
            _lib.check(L.mi_moco_bwd(q.data_ptr(), k.data_ptr(), memory.data_ptr() if need_dq else None, b, K, D,
                                     ctx.temperature, lse.data_ptr(), lpos.data_ptr(), g.data_ptr(),
                                     dq.data_ptr() if need_dq else None, dkey.data_ptr() if need_dkey else None,
                                     patch.data_ptr() if patch is not None else None,
                                     pend.patch_lo if patch is not None else 0,
                                     patch.shape[0] if patch is not None else 0,
                                     ws.data_ptr() if need_dq else None, ws.numel() if need_dq else 0,
                                     stream), "mi_moco_bwd")
        if pend is not None:
            # backward has run: pushes no longer need to protect rows for this loss
            refs = _PENDING.get(pend.memory.data_ptr())
            if refs:
                refs[:] = [r for r in refs if r() is not None and r() is not pend]
                if not refs:
                    _PENDING.pop(pend.memory.data_ptr(), None)
            ctx.pending = None
        return dq, dkey, None, None


def moco_queue_loss(query: torch.Tensor, key: torch.Tensor, memory: torch.Tensor, temperature: float = 0.1) -> torch.Tensor:
    """InfoNCE of ``query`` against its positive ``key`` and the negative queue ``memory`` (scalar).

    Differentiable w.r.t. ``query`` (the default call site, ``moco.py:152-156``) and w.r.t. ``key`` (the symmetric
    branch passes ``key=feat_query``, ``moco.py:144-148``).  The queue holds detached features in every call site
    of the reference; a ``memory`` that requires grad raises instead of silently dropping that gradient."""
    for name, t in (("query", query), ("key", key), ("memory", memory)):
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise RuntimeError(f"{name} must be a tensor on an MI355X (HIP) device: image-classification_amd has no CPU fallback")
    if query.dim() != 2 or key.shape != query.shape or memory.dim() != 2 or memory.shape[1] != query.shape[1]:
        raise ValueError(f"expected query/key [b, D] and memory [K, D], got {tuple(query.shape)}, {tuple(key.shape)}, {tuple(memory.shape)}")
    if memory.requires_grad and torch.is_grad_enabled():
        raise NotImplementedError("no gradient w.r.t. the negative queue is built: pass memory.detach() "
                                  "(every call site of the reference holds detached features in the queue)")
    return _MocoQueueLoss.apply(query, key, memory.detach(), float(temperature))


def contrastive_loss2(query, key, memory=None, norm=False, temperature=0.1, query_neg=False, key_neg=False,
                      qk_graph=None, qm_graph=None, eye_one_in_qk=False):
    """The MoCo subset of ``lumo.contrib.nn.loss.contrastive_loss2`` (keyword names as used at
    ``track_ssl/moco.py:139-156``).  Only the queue supplies negatives: ``memory`` is required,
    ``query_neg`` / ``key_neg`` must be False and no graph masks are supported -- the other call
    sites of the lumo function (SimCLR etc.) are outside this path and raise."""
    if memory is None or query_neg or key_neg or qk_graph is not None or qm_graph is not None or eye_one_in_qk:
        raise NotImplementedError("only the MoCo form is built: memory=<queue>, query_neg=False, key_neg=False, no graphs")
    if norm:
        query = torch.nn.functional.normalize(query, dim=-1)
        key = torch.nn.functional.normalize(key, dim=-1)
    return moco_queue_loss(query, key, memory, temperature)
