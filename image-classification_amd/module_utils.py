"""Wrapping feature queues on MI355X -- host-side mirror of ``models/module_utils.py:48-148``.

``MemoryBank`` ([queue_size, feature_dim] float) and ``LongTensorMemoryBank`` ([queue_size] int64) are
``torch.Tensor`` subclasses exactly as in the reference (used as ``self.mb = MemoryBank(...)`` tensors by
supcontrast / ft / infomin / cmc / swav / byol), with the same attributes (``queue_size``, ``cursor``), the same
survival of those attributes through ``.to() / .clone() / deepcopy / .half()``, and the same WRAPPING ``push``:
the part of a batch that does not fit behind the cursor continues at row 0 (unlike
``models/memory_bank.py``'s truncating queue).  ``push`` runs in one HIP kernel (``mi_queue_push_wrap``);
the queue must be on a HIP device -- there is no CPU path."""
from __future__ import annotations

import ctypes

import torch

from . import _lib

__all__ = ["MemoryBank", "LongTensorMemoryBank"]


class _WrappingQueue(torch.Tensor):
    queue_size: int
    cursor: int

    @classmethod
    def _wrap(cls, data: torch.Tensor, queue_size: int, cursor: int):
        self = torch.Tensor._make_subclass(cls, data, False)
        self.queue_size = queue_size
        self.cursor = cursor
        return self

    def to(self, *args, **kwargs):
        ncls = super().to(*args, **kwargs)
        ncls.queue_size = self.queue_size
        ncls.cursor = self.cursor
        return ncls

    def clone(self, *args, **kwargs):
        ncls = super().clone(*args, **kwargs)
        ncls.queue_size = self.queue_size
        ncls.cursor = self.cursor
        return ncls

    def __deepcopy__(self, memo):
        if id(self) in memo:
            return memo[id(self)]
        result = type(self)._wrap(self.tensor().clone(memory_format=torch.preserve_format), self.queue_size, self.cursor)
        memo[id(self)] = result
        return result

    def tensor(self):
        return torch.Tensor._make_subclass(torch.Tensor, self.data, False)

    def half(self, memory_format=None):
        return self.to(torch.float16)

    def _push_rows(self, item: torch.Tensor):
        with torch.no_grad():
            data = self.data
            if not data.is_cuda:
                raise RuntimeError(f"queue is on {data.device}: image-classification_amd moves queue rows with HIP "
                                   "kernels only (no CPU fallback) -- call .to(device) first")
            if not data.is_contiguous():
                raise RuntimeError("queue storage must be contiguous")
            item = item.detach().to(device=data.device, dtype=data.dtype).contiguous()
            if len(item) - (self.queue_size - self.cursor) > self.queue_size:
                # the reference's `self.data[:left] = item[right:]` raises here (module_utils.py:82) -- after its
                # first slice assignment has already been written; this push raises before writing anything
                raise RuntimeError(f"shape mismatch: cannot push {len(item)} rows at cursor {self.cursor} into a queue of {self.queue_size}")
            row_bytes = data.element_size() * (data[0].numel() if data.dim() > 1 else 1)
            new_cursor = ctypes.c_int64(0)
            with torch.cuda.device(data.device):
                _lib.check(_lib.lib().mi_queue_push_wrap(data.data_ptr(), self.queue_size, row_bytes, self.cursor,
                                                         item.data_ptr(), len(item), ctypes.byref(new_cursor),
                                                         _lib.stream_ptr(data.device)), "mi_queue_push_wrap")
            self.cursor = int(new_cursor.value)


class MemoryBank(_WrappingQueue):
    """``module_utils.py:48-98``: ``torch.rand(queue_size, feature_dim)`` queue, wrapping ``push``."""

    def __new__(cls, queue_size=65535, feature_dim=128):
        self = cls._wrap(torch.rand(queue_size, feature_dim), queue_size, 0)
        self.detach_()
        return self

    def push(self, item: torch.Tensor):
        assert item.ndim == 2 and item.shape[1:] == self.shape[1:], f'ndim: {item.ndim} | shape: {item.shape}'
        self._push_rows(item)


class LongTensorMemoryBank(_WrappingQueue):
    """``module_utils.py:100-148``: int64 label queue, same wrapping ``push``."""

    def __new__(cls, queue_size=65535):
        self = cls._wrap(torch.zeros(queue_size, dtype=torch.long), queue_size, 0)
        self.detach_()
        return self

    def push(self, item: torch.Tensor):
        self._push_rows(item)
