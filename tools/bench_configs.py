#!/usr/bin/env python3
"""Secondary measurements for DESIGN.md: BASELINE.json configs[0..2] on one MI355X.

  C1  kNN 10k queries x 50k bank, D=128, k=200, C=100          (fp32 parity mode and bf16)
  C2  MoCo queue loss b=256 vs 65536x128 queue: forward, forward+backward, enqueue; HBM GB/s
  C3  kNN 5k queries x 126k bank, D=2048, k=200, C=100         (fp32 and bf16)

Each line is JSON.  `cpu` legs run the torch-CPU port of the reference ops on the host cores
(oracle/knn_torch_cpu.py is test infrastructure; here it is only the thing compared against)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import image_classification_amd as ica  # noqa: E402
from image_classification_amd import _lib  # noqa: E402


def unit_rows(n, d, seed, device):
    g = torch.Generator(device=device).manual_seed(seed)
    return F.normalize(torch.randn((n, d), generator=g, device=device), dim=1)


def timed(fn, iters, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters       # ms


def knn_case(tag, B, N, D, C, k, T, cpu=True):
    dev = torch.device("cuda:0")
    f, x = unit_rows(B, D, 1, dev), unit_rows(N, D, 0, dev)
    lab = torch.randint(0, C, (N,), generator=torch.Generator(device=dev).manual_seed(2), device=dev)
    bank_t = x.t().contiguous()                                       # reference layout [D, N]
    for dt, peak in (("fp32", 157.3), ("bf16", 2500.0)):
        ms = timed(lambda: ica.knn_predict(f, bank_t, lab, C, k, T, compute_dtype=dt), 10, 3)      # wall time: no per-launch events
        _lib.profile_enable(True)                                                                   # kernel rate: a pass of its own
        timed(lambda: ica.knn_predict(f, bank_t, lab, C, k, T, compute_dtype=dt), 5, 0)
        sim_ms, n, flops = _lib.profile_collect(0)
        _lib.profile_enable(False)
        print(json.dumps({"config": tag, "dtype": dt, "queries_per_s": B / ms * 1e3, "ms": ms,
                          "sim_kernel_tflops": flops / sim_ms / 1e9 if sim_ms else None, "peak_tflops": peak,
                          "note": "knn_predict incl. transposed-bank pack, top-k, vote, ranking"}), flush=True)
    # torch on the same GPU (rocBLAS mm + topk + one-hot vote): the reference code path as-is on ROCm
    if B * N * 4 < 8e9:
        from oracle import knn_torch_cpu as kt
        ms = timed(lambda: kt.knn_predict_chunked(f, bank_t, lab, C, k, T), 3, 1)
        print(json.dumps({"config": tag, "dtype": "fp32", "impl": "reference ops on torch-ROCm (same GPU)",
                          "queries_per_s": B / ms * 1e3, "ms": ms}), flush=True)
        if cpu:
            fc, bc, lc = f.cpu(), bank_t.cpu(), lab.cpu()
            t0 = time.perf_counter()
            kt.knn_predict_chunked(fc, bc, lc, C, k, T)
            dt_s = time.perf_counter() - t0
            print(json.dumps({"config": tag, "dtype": "fp32", "impl": "torch CPU port", "cores": torch.get_num_threads(),
                              "queries_per_s": B / dt_s, "ms": dt_s * 1e3}), flush=True)


def moco_case(b=256, K=65536, D=128, T=0.1):
    dev = torch.device("cuda:0")
    q = unit_rows(b, D, 1, dev).requires_grad_(True)
    key = unit_rows(b, D, 2, dev)
    mb = ica.MemoryBank()
    mb.register("negative", dim=D, k=K)
    mb["negative"] = F.normalize(mb["negative"], dim=-1)
    mb = mb.to(dev)
    queue_bytes = K * D * 4

    def fwd():
        return ica.contrastive_loss2(query=q, key=key, memory=mb["negative"].detach(), temperature=T)

    def step():                                   # moco.py:152-172 order: loss, push, backward -- no clone
        q.grad = None
        loss = fwd()
        mb.push("negative", key)
        loss.backward()

    def ref_step():                               # the reference formulation on the same GPU, clone included
        q.grad = None
        mem = mb["negative"].clone().detach()
        logits = torch.cat([(q * key).sum(1, keepdim=True), q @ mem.t()], 1) / T
        loss = F.cross_entropy(logits, torch.zeros(b, dtype=torch.long, device=dev))
        ptr = mb.offsets["negative"]
        mb["negative"][ptr:ptr + b] = key
        mb.offsets["negative"] = (ptr + b) % K
        loss.backward()

    with torch.no_grad():
        ms_f = timed(fwd, 200, 10)
    ms_s = timed(step, 200, 10)                   # wall time: no per-launch events
    _lib.profile_enable(True)                     # kernel times: a pass of its own
    timed(step, 100, 0)
    f_ms, f_n, f_fl = _lib.profile_collect(2)
    b_ms, b_n, b_fl = _lib.profile_collect(3)
    _lib.profile_enable(False)
    ms_p = timed(lambda: mb.push("negative", key), 200, 10)
    ms_r = timed(ref_step, 100, 5)
    out = {"config": "C2 moco b=256 K=65536 D=128", "fwd_ms": ms_f, "step_ms(loss+push+bwd)": ms_s, "push_ms": ms_p,
           "fwd_kernels_ms": f_ms / max(f_n, 1), "bwd_kernels_ms": b_ms / max(b_n, 1),
           "fwd_GBps": queue_bytes / (f_ms / max(f_n, 1)) / 1e6, "fwd_TFLOPs": f_fl / max(f_ms, 1e-9) / 1e9,
           "bwd_GBps": queue_bytes / (b_ms / max(b_n, 1)) / 1e6, "bwd_TFLOPs": b_fl / max(b_ms, 1e-9) / 1e9,
           "torch_rocm_reference_step_ms": ms_r}
    # CPU port: torch.mm + cross_entropy + slice assign
    qc, kc, mc = q.detach().cpu().requires_grad_(True), key.cpu(), mb["negative"].cpu()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        qc.grad = None
        mem = mc.clone()
        logits = torch.cat([(qc * kc).sum(1, keepdim=True), qc @ mem.t()], 1) / T
        loss = F.cross_entropy(logits, torch.zeros(b, dtype=torch.long))
        mc[:b] = kc
        loss.backward()
    out["torch_cpu_step_ms"] = (time.perf_counter() - t0) / n * 1e3
    out["cpu_cores"] = torch.get_num_threads()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["c1", "c2", "c3"]
    _lib.lib()
    if "c2" in which:
        moco_case()
    if "c1" in which:
        knn_case("C1 knn 10000x50000 D=128 C=100", 10000, 50000, 128, 100, 200, 0.1)
    if "c3" in which:
        knn_case("C3 knn 5000x126000 D=2048 C=100", 5000, 126000, 2048, 100, 200, 0.1)
