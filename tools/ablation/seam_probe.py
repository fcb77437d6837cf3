#!/usr/bin/env python3
"""BENCH-ONLY: one C4-sized filter launch through libmi_knn_seam.so (make_seam_variant.py): per CU, the time between the last
instruction of one workgroup and the first of the next, and how long a workgroup needs from its first instruction to the start
of its K loop.  usage: MI_KNN_LIB=<seam lib> python seam_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import image_classification_amd as ica  # noqa: E402
from image_classification_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
B, N, D, k = 16384, 327680, int(os.environ.get("PROBE_D", "2048")), 200
x = torch.nn.functional.normalize(torch.randn((N, D), generator=g, device=dev), dim=1)
f = torch.nn.functional.normalize(torch.randn((B, D), generator=g, device=dev), dim=1)
bank = ica.BankIndex.from_rows(x, "bf16")
for _ in range(3):
    ica.knn_topk(f, bank, k)
torch.cuda.synchronize()
cap = 8192 * 2
st = torch.zeros((cap, 2), dtype=torch.int64, device=dev)
_lib.check(_lib.lib().mi_profile_clock(st.data_ptr(), cap), "clock")
ica.knn_topk(f, bank, k)                           # the last full-size launch of the call leaves its stamps
torch.cuda.synchronize()
_lib.check(_lib.lib().mi_profile_clock(None, 0), "clock")
r = st.cpu().numpy().view(np.uint64).reshape(-1, 4)
r = r[r[:, 1] > 0]
start, end, hw, k0 = r[:, 0].astype(np.int64), r[:, 1].astype(np.int64), r[:, 2], r[:, 3].astype(np.int64)
cu = ((hw >> np.uint64(32)) & np.uint64(0xf)) * np.uint64(1 << 16) + (hw & np.uint64(0xffff) & ~np.uint64(0xff))   # xcc, se/sh/cu bits
gaps, life, pro = [], [], []
for c in np.unique(cu):
    m = cu == c
    o = np.argsort(start[m])
    s_, e_, k_ = start[m][o], end[m][o], k0[m][o]
    keep = (s_[1:] - e_[:-1] > -1000) & (s_[1:] - e_[:-1] < 100000)   # same launch, back to back
    gaps.append((s_[1:] - e_[:-1])[keep])
    life.append(e_ - s_)
    pro.append(k_ - s_)
gaps, life, pro = np.concatenate(gaps), np.concatenate(life), np.concatenate(pro)
print(f"{len(r)} workgroups on {len(np.unique(cu))} CUs (by XCC_ID / HW_ID)")
print(f"workgroup lifetime  median {np.median(life):8.0f} cycles")
print(f"first instruction -> K loop   median {np.median(pro):8.0f} cycles")
print(f"end of one workgroup -> first instruction of the next on the same CU: median {np.median(gaps):8.0f}  p10 {np.percentile(gaps, 10):8.0f}  p90 {np.percentile(gaps, 90):8.0f} cycles ({len(gaps)} seams)")
print(f"seam share of a CU's time: {np.median(gaps) / (np.median(gaps) + np.median(life)):.3f} (dispatch gap) + {np.median(pro) / (np.median(gaps) + np.median(life)):.3f} (prologue)")
