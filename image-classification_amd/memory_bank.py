"""Feature banks on MI355X -- host-side mirror of the reference's ``models/memory_bank.py``.

``MemoryBank`` (named truncating ring-buffer queues, ``memory_bank.py:35-72``) and ``StorageBank``
(index-addressed stores, ``memory_bank.py:8-32``) keep the reference's method names, argument
meaning and storage model: both are ``nn.Module``s whose storage is *registered buffers*, so
``.to(device)`` and ``state_dict`` behave as in the reference (the ring pointer ``offsets`` is a
plain dict there and is likewise not checkpointed here).

The data movement (``push`` / ``scatter``) runs in HIP kernels (``mi_queue_push``,
``mi_bank_scatter``); buffers must be on a HIP device when those are called -- there is no CPU path.
Under ``torch.distributed`` the pushed rows of all ranks are all-gathered first (RCCL), exactly as
the reference's ``accelerate.utils.gather`` does (``memory_bank.py:29,64``): every rank keeps an
identical replicated queue.
"""
from __future__ import annotations

import ctypes

import torch
import torch.distributed as dist
from torch import nn

from . import _lib

__all__ = ["MemoryBank", "StorageBank", "gather"]


def gather(value: torch.Tensor) -> torch.Tensor:
    """All-gather + concat on dim 0 across ranks; identity in a single process
    (what ``accelerate.utils.gather`` does for the reference)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    value = value.contiguous()
    if value.dim() == 0:
        value = value[None]
    world = dist.get_world_size()
    if dist.get_backend() == "nccl":            # RCCL on ROCm
        out = torch.empty((world * value.shape[0],) + tuple(value.shape[1:]), dtype=value.dtype, device=value.device)
        dist.all_gather_into_tensor(out, value)
        return out
    host = value.cpu()
    parts = [torch.empty_like(host) for _ in range(world)]
    dist.all_gather(parts, host)
    return torch.cat(parts, dim=0).to(value.device)


def _need_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{what}: buffer is on {t.device}; image-classification_amd moves bank rows "
                           "with HIP kernels only (no CPU fallback) -- call .to(device) first")


class StorageBank(nn.Module):
    """``memory_bank.py:8-32``: ``register`` creates a buffer filled with ``ones + inf`` cast to
    ``dtype`` (float -> +inf, int64 -> INT64_MIN); ``scatter`` writes rows at dataset indices."""

    def __init__(self):
        super().__init__()
        self.offset = 0
        self.sizes = {}

    def register(self, name, dim, k, dtype=None):
        if dim <= 0:
            bank = (torch.ones(k) + float("inf")).to(dtype=dtype)
        else:
            bank = (torch.ones(k, dim) + float("inf")).to(dtype=dtype)
        self.register_buffer(name, bank)
        self.sizes[name] = bank.shape[0]

    def __getitem__(self, item):
        return self.__getattr__(item)

    @torch.no_grad()
    def scatter(self, name, value, index):
        bank = self[name]
        _need_gpu(bank, "StorageBank.scatter")
        value = gather(value.detach())
        if isinstance(index, torch.Tensor):
            index = gather(index)
        else:
            index = torch.as_tensor(index)
        index = index.to(device=bank.device, dtype=torch.int64).reshape(-1).contiguous()
        value = value.to(device=bank.device, dtype=bank.dtype).contiguous()
        if value.dim() == bank.dim() - 1:                      # a single row / scalar for every index
            value = value.expand((index.numel(),) + tuple(bank.shape[1:])).contiguous()
        if tuple(value.shape[1:]) != tuple(bank.shape[1:]) or value.shape[0] != index.numel():
            raise RuntimeError(f"shape mismatch: value {tuple(value.shape)} cannot be assigned to "
                               f"{index.numel()} rows of bank {tuple(bank.shape)}")
        row_bytes = bank.element_size() * (bank[0].numel() if bank.dim() > 1 else 1)
        with torch.cuda.device(bank.device):
            _lib.check(_lib.lib().mi_bank_scatter(bank.data_ptr(), bank.shape[0], row_bytes, index.data_ptr(),
                                                  value.data_ptr(), index.numel(), _lib.stream_ptr(bank.device)),
                       "mi_bank_scatter")


class MemoryBank(nn.Module):
    """``memory_bank.py:35-72``: named ring buffers; ``push`` TRUNCATES at the end of the buffer
    (overflow rows are dropped, pointer -> ``(ptr + written) % k``)."""

    def __init__(self, k=None):
        super().__init__()
        self.offset = 0
        self.offsets = {}
        self.sizes = {}

    def register(self, name, dim, k, dtype=None):
        if dim <= 0:
            bank = torch.rand(k, dtype=dtype)
        else:
            bank = torch.rand(k, dim, dtype=dtype)
        self.register_buffer(name, bank)
        self.offsets[name] = 0
        self.sizes[name] = k

    def __setitem__(self, key, value):
        self.__setattr__(key, value)

    def __getitem__(self, item):
        return self.__getattr__(item)

    @torch.no_grad()
    def push(self, name, value):
        assert name in self.offsets
        bank = self[name]
        assert bank.ndim == value.ndim
        _need_gpu(bank, "MemoryBank.push")
        value = gather(value.detach()).to(device=bank.device, dtype=bank.dtype).contiguous()
        if tuple(value.shape[1:]) != tuple(bank.shape[1:]):
            raise RuntimeError(f"shape mismatch: cannot push rows {tuple(value.shape[1:])} into bank rows {tuple(bank.shape[1:])}")
        if not bank.is_contiguous():
            raise RuntimeError("MemoryBank buffers must be contiguous")
        ptr, k = self.offsets[name], self.sizes[name]
        row_bytes = bank.element_size() * (bank[0].numel() if bank.dim() > 1 else 1)
        new_ptr = ctypes.c_int64(0)
        # a MoCo loss over this very buffer may still await its backward (moco.py:159-172 pushes
        # first): save the rows about to be overwritten for it instead of cloning the queue
        from .moco import protect_rows
        stash = protect_rows(bank, ptr, min(value.shape[0], k - ptr))
        with torch.cuda.device(bank.device):
            if stash is not None:
                _lib.check(_lib.lib().mi_queue_push_stash(bank.data_ptr(), k, row_bytes, ptr, value.data_ptr(),
                                                          value.shape[0], stash.data_ptr(), ctypes.byref(new_ptr),
                                                          _lib.stream_ptr(bank.device)), "mi_queue_push_stash")
            else:
                _lib.check(_lib.lib().mi_queue_push(bank.data_ptr(), k, row_bytes, ptr, value.data_ptr(),
                                                    value.shape[0], ctypes.byref(new_ptr), _lib.stream_ptr(bank.device)),
                           "mi_queue_push")
        self.offsets[name] = int(new_ptr.value)
