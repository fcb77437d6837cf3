#!/usr/bin/env python3
"""BENCH-ONLY: time one C4-sized query chunk through libmi_knn_tauinf.so (make_tauinf_variant.py: nothing passes the
filter, same instruction stream for any data) on operands that toggle the datapath differently, with the in-kernel
clock (mi_profile_clock).  usage: MI_KNN_LIB=<tauinf lib> python toggle_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import image_classification_amd as ica  # noqa: E402
from image_classification_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
B, N, k = 16384, 327680, 200
D = int(os.environ.get("PROBE_D", "2048"))
xr = torch.nn.functional.normalize(torch.randn((N, D), generator=g, device=dev), dim=1)
fr = torch.nn.functional.normalize(torch.randn((B, D), generator=g, device=dev), dim=1)
cases = {f"random x random ({i})": (fr, xr) for i in range(4)} if os.environ.get("PROBE_RANDOM_ONLY") else {
    "random x random": (fr, xr),
    "random queries x zero bank": (fr, torch.zeros_like(xr)),
    "zero x zero": (torch.zeros_like(fr), torch.zeros_like(xr)),
    "one repeated row x one repeated row": (fr[:1].expand(B, D).contiguous(), xr[:1].expand(N, D).contiguous()),
    "random x random (again)": (fr, xr),
}
cap = 8192
for name, (f, x) in cases.items():
    bank = ica.BankIndex.from_rows(x, "bf16")
    for _ in range(2):
        ica.knn_topk(f, bank, k)
    st = torch.zeros((cap, 2), dtype=torch.int64, device=dev)
    _lib.check(_lib.lib().mi_profile_clock(st.data_ptr(), cap), "clock")
    ica.knn_topk(f, bank, k)
    torch.cuda.synchronize()
    _lib.check(_lib.lib().mi_profile_clock(None, 0), "clock")
    r = st.cpu().double()
    r = r[r[:, 1] > 0]
    ghz = float((r[:, 0] / r[:, 1]).median()) * 0.1
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ica.knn_topk(f, bank, k)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    tf = 2.0 * B * N * D / ms / 1e9
    print(f"{name:38s} {ms:7.2f} ms  {tf:6.0f} TFLOP/s (whole call)  clock {ghz:.2f} GHz  -> {tf / (ghz / 2.4 * 2500):.3f} of the MFMA peak at that clock", flush=True)
    del bank
