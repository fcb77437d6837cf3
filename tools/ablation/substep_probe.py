#!/usr/bin/env python3
"""BENCH-ONLY: f32 filter launch through libmi_knn_substep.so: cycles per sub-step (32 MFMAs) and wave, per K-step."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import image_classification_amd as ica  # noqa: E402
from image_classification_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
B, N, D, k = 16384, 65536, 2048, 200
x = torch.nn.functional.normalize(torch.randn((N, D), generator=g, device=dev), dim=1)
f = torch.nn.functional.normalize(torch.randn((B, D), generator=g, device=dev), dim=1)
bank = ica.BankIndex.from_rows(x, "fp32")
for _ in range(2):
    ica.knn_topk(f, bank, k)
torch.cuda.synchronize()
cap = 4096 * 64
st = torch.zeros((cap, 2), dtype=torch.int64, device=dev)
_lib.check(_lib.lib().mi_profile_clock(st.data_ptr(), cap), "clock")
ica.knn_topk(f, bank, k)
torch.cuda.synchronize()
_lib.check(_lib.lib().mi_profile_clock(None, 0), "clock")
r = st.cpu().double().view(-1, 8, 16)[:, :, :9]
r = r[r[:, 0, 0] > 0]
ksteps = D * 4 // 128
print(f"{r.shape[0]} workgroups; median cycles per K-step: sub-steps 0..7 (32 MFMAs each: 1024 at the instruction's rate), then wait + barrier")
for w in range(8):
    print(f"wave {w}: " + " ".join(f"{float(r[:, w, i].median()) / ksteps:6.0f}" for i in range(9)))
