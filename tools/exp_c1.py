#!/usr/bin/env python3
"""C1 (10k x 50k, D=128) timing loop for profiling."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
import image_classification_amd as ica
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B, N, D, C = 10000, 50000, 128, 100
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
f = F.normalize(torch.randn(B, D, generator=g, device=dev), dim=1)
x = F.normalize(torch.randn(N, D, generator=g, device=dev), dim=1)
lab = torch.randint(0, C, (N,), generator=g, device=dev)
bank_t = x.t().contiguous()
for _ in range(3):
    ica.knn_predict(f, bank_t, lab, C, 200, 0.1, compute_dtype=dt)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ica.knn_predict(f, bank_t, lab, C, 200, 0.1, compute_dtype=dt)
e1.record(); torch.cuda.synchronize()
print(dt, e0.elapsed_time(e1) / 10, "ms per knn_predict")
from image_classification_amd import _lib
_lib.profile_enable(True)
for _ in range(10):
    ica.knn_predict(f, bank_t, lab, C, 200, 0.1, compute_dtype=dt)
torch.cuda.synchronize()
sim_ms, n, fl = _lib.profile_collect(0)
sel_ms, n2, _ = _lib.profile_collect(1)
_lib.profile_enable(False)
print(f"{dt} {os.environ.get('MI_KNN_LIB', 'libmi_knn.so').split('/')[-1]}: similarity kernels {sim_ms / 10 * 1e3:.1f} us per call ({n // 10} launches, {fl / sim_ms / 1e9:.0f} TFLOP/s), "
      f"select kernels {sel_ms / 10 * 1e3:.1f} us per call")
