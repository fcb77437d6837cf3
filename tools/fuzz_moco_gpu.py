#!/usr/bin/env python3
"""Randomised shape fuzz of the MoCo loss (forward, dq, dkey) and of the CoMatch soft-kNN on one GPU, against the float64
restatements in oracle/knn_oracle.py (loss 2e-5, gradients 1e-4 of their maximum, soft-kNN the tolerance of tests/test_soft_knn.py).
usage: tools/fuzz_moco_gpu.py [cases] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import image_classification_amd as ica  # noqa: E402
from oracle import knn_oracle as ko  # noqa: E402


def run(cases=40, seed=0, verbose=True):
    rs = np.random.RandomState(seed)
    t0 = time.time()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    for c in range(cases):
        b = int(rs.choice([1, 2, 63, 64, 65, 130, 256, 300, 517]))
        K = int(rs.choice([1, 63, 64, 65, 257, 1000, 4096, 5000, 16385, 65536]))
        D = int(rs.choice([64, 128, 192, 256]))
        T = float(rs.choice([0.07, 0.1, 0.2, 0.5]))
        q = ko.l2_normalize(rs.randn(b, D))
        k = ko.l2_normalize(q + 0.3 * rs.randn(b, D))
        mem = ko.moco_queue_init(K, D, rs)
        if rs.rand() < 0.5:
            mem[: min(b, K)] = ko.l2_normalize(q[: min(b, K)] + 0.05 * rs.randn(min(b, K), D))
        loss_ref, dq_ref, _, _, dk_ref = ko.moco_loss(q, k, mem, T, with_dkey=True)
        tq, tk = d(q).requires_grad_(True), d(k).requires_grad_(True)
        loss = ica.contrastive_loss2(query=tq, key=tk, memory=d(mem), query_neg=False, key_neg=False, temperature=T, norm=False)
        loss.backward()
        ok = abs(loss.item() - loss_ref) <= 2e-5 * abs(loss_ref)
        ok = ok and np.max(np.abs(tq.grad.cpu().numpy() - dq_ref)) <= 1e-4 * np.max(np.abs(dq_ref))
        ok = ok and np.max(np.abs(tk.grad.cpu().numpy() - dk_ref)) <= 1e-4 * np.max(np.abs(dk_ref)) + 1e-12
        # soft-kNN on a bank of the same shape
        C = int(rs.choice([1, 2, 10, 63, 64, 100, 191, 255]))
        bp = rs.dirichlet(np.ones(C), size=K).astype(np.float32)
        p = rs.dirichlet(np.ones(C), size=b).astype(np.float32)
        out = ica.smooth_prob(d(q), d(mem), d(bp), d(p), 0.2, 0.9).cpu().numpy()
        ref = ko.comatch_smooth_prob(q.astype(np.float32), mem.astype(np.float32), bp, p, 0.2, 0.9, dtype=np.float64)
        ok2 = bool(np.allclose(out, ref, rtol=1e-5, atol=1e-7))
        if verbose:
            print(f"case {c:3d} b={b:4d} K={K:6d} D={D:4d} T={T:.2f} C={C:4d} moco {'ok' if ok else 'MISMATCH'} soft {'ok' if ok2 else 'MISMATCH'}", flush=True)
        assert ok and ok2, f"fuzz case {c} (seed {seed}): b={b} K={K} D={D} T={T} C={C}"
    if verbose:
        print(f"{cases} cases within tolerance in {time.time() - t0:.1f} s")
    return cases


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
