#!/usr/bin/env python3
"""BENCH-ONLY: does the clock the chip grants depend on WHEN in a burst a launch runs?  One stage-sized problem
(50 000 queries x 32 768 bank rows x 2048, bf16) is run 12 times back to back after an idle pause, each call with its own
stamp buffer (mi_profile_clock), then again with a 1 ms memory-bound copy between the calls."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import image_classification_amd as ica  # noqa: E402
from image_classification_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
B, N, D, k = 50000, 32768, 2048, 200
x = torch.nn.functional.normalize(torch.randn((N, D), generator=g, device=dev), dim=1)
f = torch.nn.functional.normalize(torch.randn((B, D), generator=g, device=dev), dim=1)
bank = ica.BankIndex.from_rows(x, "bf16")
big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
big2 = torch.empty_like(big)
for _ in range(3):
    ica.knn_topk(f, bank, k)
torch.cuda.synchronize()


def burst(n, between):
    stamps = [torch.zeros((16384, 2), dtype=torch.int64, device=dev) for _ in range(n)]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * n)]
    time.sleep(0.2)
    for i in range(n):
        _lib.check(_lib.lib().mi_profile_clock(stamps[i].data_ptr(), 16384), "clock")
        ev[2 * i].record()
        ica.knn_topk(f, bank, k)
        ev[2 * i + 1].record()
        if between:
            big2.copy_(big)
    torch.cuda.synchronize()
    _lib.check(_lib.lib().mi_profile_clock(None, 0), "clock")
    out = []
    for i in range(n):
        st = stamps[i].cpu().double()
        st = st[st[:, 1] > 0]
        out.append((ev[2 * i].elapsed_time(ev[2 * i + 1]), float((st[:, 0] / st[:, 1]).median()) * 0.1))
    return out


for name, between in (("back to back", False), ("1 GiB copy between calls", True), ("back to back", False)):
    r = burst(12, between)
    print(name)
    print("  ms   : " + " ".join(f"{ms:6.2f}" for ms, _ in r))
    print("  GHz  : " + " ".join(f"{c:6.2f}" for _, c in r), flush=True)
