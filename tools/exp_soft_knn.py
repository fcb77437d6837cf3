#!/usr/bin/env python3
"""Wall time per smooth_prob call (CoMatch soft-kNN, comatch.py:187-189) at the reference's shape and a larger one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
import image_classification_amd as ica

dev = torch.device("cuda:0")
for b, K, D, C in ((448, 14336, 64, 10), (448, 14336, 64, 100), (1024, 65536, 128, 100)):
    f = F.normalize(torch.randn(b, D, device=dev), dim=1)
    bf = F.normalize(torch.randn(K, D, device=dev), dim=1)
    bp = torch.softmax(torch.randn(K, C, device=dev), 1)
    p = torch.softmax(torch.randn(b, C, device=dev), 1)
    for _ in range(20):
        ica.smooth_prob(f, bf, bp, p, 0.2, 0.9)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 500
    e0.record()
    for _ in range(n):
        ica.smooth_prob(f, bf, bp, p, 0.2, 0.9)
    e1.record()
    torch.cuda.synchronize()
    print(f"b={b} K={K} D={D} C={C}: {e0.elapsed_time(e1) / n * 1e3:.1f} us per call", flush=True)
