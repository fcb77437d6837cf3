#!/usr/bin/env python3
"""BENCH-ONLY: run one C4-sized query chunk through libmi_knn_phase.so (tools/ablation/make_phase_variant.py) and print
how a wave's K-steps split into issue / DMA wait / barrier wait.  usage: MI_KNN_LIB=<phase lib> python phase_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import image_classification_amd as ica  # noqa: E402
from image_classification_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
B, N, D, k = 16384, int(os.environ.get("PROBE_N", "327680")), 2048, 200
x = torch.nn.functional.normalize(torch.randn((N, D), generator=g, device=dev), dim=1)
f = torch.nn.functional.normalize(torch.randn((B, D), generator=g, device=dev), dim=1)
DT = os.environ.get("PROBE_DTYPE", "bf16")
bank = ica.BankIndex.from_rows(x, DT)
for _ in range(3):
    ica.knn_topk(f, bank, k)
torch.cuda.synchronize()
cap = 8192 * 32
st = torch.zeros((cap, 2), dtype=torch.int64, device=dev)
_lib.check(_lib.lib().mi_profile_clock(st.data_ptr(), cap), "clock")
for _ in range(3):
    ica.knn_topk(f, bank, k)
torch.cuda.synchronize()
_lib.check(_lib.lib().mi_profile_clock(None, 0), "clock")
r = st.cpu().double().view(-1, 8, 8)              # [workgroup, wave, (issue, dma, barrier, lifetime, prologue, epilogue, slow, hits)]
ok = r[:, :, 3].min(dim=1).values > 0
r = r[ok]
ksteps = D * (2 if DT == "bf16" else 4) // 128
print(f"{r.shape[0]} workgroups, {ksteps} K-steps per tile; median cycles per K-step and wave")
for w in range(8):
    i, d, b, life, pre, post = (float(r[:, w, j].median()) for j in range(6))
    print(f"wave {w}: issue {i / ksteps:7.0f}  dma-wait {d / ksteps:6.0f}  barrier {b / ksteps:6.0f}  sum {(i + d + b) / ksteps:7.0f}   "
          f"tile: lifetime {life:7.0f} prologue {pre:6.0f} epilogue {post:6.0f} (mean {float(r[:, w, 5].mean()):6.0f}) slow entries/tile {float(r[:, w, 6].mean()):5.1f} hit lanes {float(r[:, w, 7].mean()):5.1f}")
tot = r[:, :, :3].sum(dim=2).median()
print(f"shares (all waves): issue {float(r[:, :, 0].sum() / r[:, :, :3].sum()):.3f} dma {float(r[:, :, 1].sum() / r[:, :, :3].sum()):.3f} barrier {float(r[:, :, 2].sum() / r[:, :, :3].sum()):.3f}")
