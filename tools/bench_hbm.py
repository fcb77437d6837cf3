#!/usr/bin/env python3
"""Achieved HBM bandwidth of the streaming (non-GEMM) kernels of the path, for DESIGN.md: algorithmic bytes of the
operation (the figures of SURVEY.md 8(d)) / time per call, HIP events around a loop of calls, one MI355X.
Each line is JSON."""
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
    return e0.elapsed_time(e1) / iters * 1e-3          # s


def line(name, nbytes, sec, what):
    print(json.dumps({"kernel": name, "us": round(sec * 1e6, 1), "algorithmic_MB": round(nbytes / 1e6, 2),
                      "GBps": round(nbytes / sec / 1e9, 1), "what": what}), flush=True)


def main():
    g = torch.Generator(device=dev).manual_seed(0)
    # bank pack: [N, D] f32 -> bf16 packed rows (C4 bank: read 4 B, write 2 B per element)
    N, D = 1_280_000, 2048
    x = torch.randn((N, D), generator=g, device=dev)
    t = timed(lambda: ica.BankIndex.from_rows(x, "bf16"), 5, 1)
    line("pack_rows_vec_kernel (bank -> bf16)", N * D * 6, t, f"{N}x{D} f32 -> bf16 packed")
    # normalise + append one memo batch into a packed bank (ssltrainer.py:176-179)
    b = 65536
    builder = ica.BankBuilder(b, D, dev, compute_dtype="bf16")
    xb = x[:b]

    def app():
        builder.filled = 0
        builder.append(xb)
    t = timed(app, 10, 2)
    line("append_normalized_kernel", b * D * 6, t, f"{b}x{D} f32 rows L2-normalised -> bf16 packed")
    del x, builder
    # vote + ranking: B x k (value, index) + labels -> [B, C] ranking
    B, k, C, Nl = 16384, 200, 1000, 1_280_000
    val = torch.rand((B, k), generator=g, device=dev).sort(dim=1, descending=True).values
    idx = torch.randint(0, Nl, (B, k), generator=g, device=dev)
    lab = torch.randint(0, C, (Nl,), generator=g, device=dev)
    t = timed(lambda: ica.knn_vote(val, idx, lab, C, 0.1, check=False))      # the kernel, without the host-side label check
    line("vote_kernel", B * k * 12 + B * k * 8 + B * C * 8, t, f"B={B} k={k} C={C}: values+indices in, label gather, int64 ranking out")
    # cross-shard merge: [G, B, k] candidates -> [B, k]
    G, Bm = 8, 2048
    cv = torch.rand((G, Bm, k), generator=g, device=dev).sort(dim=2, descending=True).values
    ci = torch.randint(0, Nl, (G, Bm, k), generator=g, device=dev)
    t = timed(lambda: ica.knn_merge(cv, ci, k))
    line("merge_wave_kernel (shard merge)", G * Bm * k * 12 + Bm * k * 12, t, f"G={G} B={Bm} k={k}")
    # ... and at the shape ONE rank of an 8-rank C4 step merges: its 1/8 slice of a 50 000-query chunk (sharded.py)
    Bm = 6250
    cv = torch.rand((G, Bm, k), generator=g, device=dev).sort(dim=2, descending=True).values
    ci = torch.randint(0, Nl, (G, Bm, k), generator=g, device=dev)
    t = timed(lambda: ica.knn_merge(cv, ci, k))
    line("merge_wave_kernel (one rank's slice at G=8, C4)", G * Bm * k * 12 + Bm * k * 12, t, f"G={G} B={Bm} k={k}")
    # MoCo queue: truncating ring push (memory_bank.py:65-72) and StorageBank scatter (:32)
    mb = ica.MemoryBank()
    mb.register("negative", dim=128, k=65536)
    mb = mb.to(dev)
    keys = F.normalize(torch.randn((256, 128), generator=g, device=dev), dim=1)
    t = timed(lambda: mb.push("negative", keys), 200, 10)
    line("push_kernel (MoCo enqueue)", 2 * 256 * 128 * 4, t, "256 x 128 f32 keys into a 65536-row ring (launch-latency scale)")
    big = torch.randn((65536, 2048), generator=g, device=dev)
    mb2 = ica.MemoryBank()
    mb2.register("f", dim=2048, k=1 << 20)
    mb2 = mb2.to(dev)
    t = timed(lambda: mb2.push("f", big), 10, 2)
    line("push_kernel (large rows)", 2 * big.numel() * 4, t, "65536 x 2048 f32 rows into a 1M-row ring")
    sb = ica.StorageBank()
    sb.register("test_feature", 2048, 50000)
    sb = sb.to(dev)
    idxs = torch.randperm(50000, generator=g, device=dev)[:4096]
    rows = big[:4096]
    t = timed(lambda: sb.scatter("test_feature", rows, idxs), 50, 5)
    line("scatter_rows_kernel (StorageBank.scatter)", 2 * rows.numel() * 4 + 4096 * 8, t, "4096 x 2048 f32 rows to permuted slots")
    # k-th largest of a 50 000-score bank (chmatch.py:252)
    sc = torch.rand(50000, generator=g, device=dev)
    t = timed(lambda: ica.kth_largest(sc, 40000), 100, 5)
    line("kth_largest_reg_kernel", 50000 * 4, t, "n=50000 k=40000: one workgroup, the bank in registers, bit-sliced radix select (latency bound; the call's host side included)")


if __name__ == "__main__":
    main()
