#!/usr/bin/env python3
"""Eager launches vs HIP-graph replay of the whole kNN monitor pass (launch-bound cases: C1; MoCo-sized batches)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import image_classification_amd as ica  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for tag, B, N, D, C, dt in [("C1", 10000, 50000, 128, 100, "bf16"), ("C1", 10000, 50000, 128, 100, "fp32"),
                            ("small batch", 256, 1280000, 128, 1000, "bf16"), ("C3", 5000, 126000, 2048, 100, "bf16")]:
    g = torch.Generator(device=dev).manual_seed(0)
    f = F.normalize(torch.randn((B, D), generator=g, device=dev), dim=1)
    x = F.normalize(torch.randn((N, D), generator=g, device=dev), dim=1)
    lab = torch.randint(0, C, (N,), generator=g, device=dev)
    bank = ica.BankIndex.from_rows(x, dt)
    run = lambda: ica.knn_predict(f, bank, lab, C, 200, 0.1)
    eager = timed(run)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = run()
    rep = timed(graph.replay)
    print(json.dumps({"case": tag, "B": B, "N": N, "D": D, "dtype": dt, "eager_ms": round(eager, 4), "graph_replay_ms": round(rep, 4)}), flush=True)
