#!/usr/bin/env python3
"""Latency of small query batches against a large bank (serving-sized calls of knn_predict)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.nn.functional as F
import image_classification_amd as ica
dev = torch.device("cuda:0")
def timed(fn, iters=20, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
g = torch.Generator(device=dev).manual_seed(0)
for N, D in [(1280000, 128), (1280000, 2048)]:
    x = F.normalize(torch.randn((N, D), generator=g, device=dev), dim=1)
    lab = torch.randint(0, 1000, (N,), generator=g, device=dev)
    bank = ica.BankIndex.from_rows(x, "bf16")
    del x
    for B in (1, 64, 256, 512, 1024, 2048, 4096):
        f = F.normalize(torch.randn((B, D), generator=g, device=dev), dim=1)
        ms = timed(lambda: ica.knn_predict(f, bank, lab, 1000, 200, 0.1))
        print(json.dumps({"N": N, "D": D, "B": B, "ms": round(ms, 4), "bank_GBps": round(N * D * 2 / ms / 1e6, 1)}), flush=True)
    del bank
