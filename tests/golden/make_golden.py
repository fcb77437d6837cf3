#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Run in the build container only (it needs /root/reference, which does not travel):

    python tests/golden/make_golden.py

* ``models/knn.py`` is imported unmodified (it needs only torch) and ``knn_predict`` is called
  on each case; the returned ranking is stored as ``pred_labels``.  The intermediates the
  function does not return (similarities, top-k, vote scores) are recomputed here with the same
  torch ops and stored too, after asserting that they reproduce the function's own output.
* ``models/memory_bank.py`` imports ``lumo.proc.dist.is_dist`` at module level; ``lumo`` is a
  third-party package that is not installed here.  The symbol is only used by the off-path
  ``batch_shuffle_ddp`` helpers, so the import is satisfied with an inert placeholder module
  (``is_dist -> False``); ``MemoryBank`` / ``StorageBank`` then run as written
  (``accelerate.utils.gather`` is installed and is the identity in a single process).

Only data (inputs + expected outputs) is written; no reference source is copied.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, "..", ".."))

from models.knn import knn_predict  # noqa: E402  (the reference)

from oracle import knn_oracle as ko  # noqa: E402  (only for bf16 rounding / normalisation helpers)


def _save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def _run_knn(name, feature, bank_rows, labels, classes, k, T, store_inputs="f32", extra=None):
    """feature [B, D] f32, bank_rows [N, D] f32 (reference receives its transpose [D, N])."""
    f = torch.from_numpy(np.ascontiguousarray(feature, np.float32))
    bank = torch.from_numpy(np.ascontiguousarray(bank_rows, np.float32)).t().contiguous()
    lab = torch.from_numpy(np.asarray(labels, np.int64))
    torch.set_num_threads(8)
    pred = knn_predict(f, bank, lab, classes, k, T)

    # same ops again, keeping the intermediates
    sim = torch.mm(f, bank)
    w_raw, idx = sim.topk(k=k, dim=-1)
    sim_labels = torch.gather(lab.expand(f.size(0), -1), dim=-1, index=idx)
    w = (w_raw / T).exp()
    one_hot = torch.zeros(f.size(0) * k, classes).scatter(dim=-1, index=sim_labels.view(-1, 1), value=1.0)
    scores = torch.sum(one_hot.view(f.size(0), -1, classes) * w.unsqueeze(dim=-1), dim=1)
    assert torch.equal(scores.argsort(dim=-1, descending=True), pred), name

    # gap between the k-th and (k+1)-th similarity: marks rounding-sensitive rows
    srt = sim.sort(dim=-1, descending=True).values
    gap = (srt[:, k - 1] - srt[:, k]) if k < sim.shape[1] else torch.full((f.size(0),), float("inf"))

    arrays = dict(
        classes=np.int64(classes), knn_k=np.int64(k), knn_t=np.float64(T),
        labels=np.asarray(labels, np.int32),
        pred_labels=pred.numpy().astype(np.int32),
        sim_weight=w_raw.numpy(), sim_indices=idx.numpy().astype(np.int32),
        pred_scores=scores.numpy(), kth_gap=gap.numpy(),
    )
    if store_inputs == "f32":
        arrays.update(feature=f.numpy(), bank_rows=np.ascontiguousarray(bank_rows, np.float32))
    elif store_inputs == "bf16":
        arrays.update(feature_bf16=ko.to_bf16_bits(feature), bank_rows_bf16=ko.to_bf16_bits(bank_rows))
    elif store_inputs == "dyadic":
        arrays.update(extra)
    if store_inputs != "f32" or sim.numel() <= 1 << 16:
        pass
    _save(name, **arrays)


def knn_cases():
    # (i) random N(0,1), L2-normalised
    for name, (B, N, D, C, k, T), seed in [
        ("knn_rand_tiny", (8, 50, 16, 5, 10, 0.1), 0),
        ("knn_rand_cifar100", (64, 1000, 128, 100, 200, 0.1), 1),
        ("knn_rand_c10_4096", (128, 4096, 128, 10, 200, 0.1), 2),
        ("knn_rand_k_eq_n", (4, 12, 8, 3, 12, 0.5), 3),
        ("knn_rand_one_class", (4, 30, 8, 1, 5, 0.1), 4),
        ("knn_rand_ragged", (37, 777, 72, 7, 33, 0.07), 5),
    ]:
        rs = np.random.RandomState(seed)
        f = ko.l2_normalize(rs.randn(B, D))
        x = ko.l2_normalize(rs.randn(N, D))
        lab = rs.randint(0, C, size=N)
        _run_knn(name, f, x, lab, C, k, T)

    # (iii) bf16-rounded-then-upcast inputs: oracle for the bf16 mode
    rs = np.random.RandomState(10)
    B, N, D, C, k, T = 64, 1000, 128, 100, 200, 0.1
    f = ko.bf16_round(ko.l2_normalize(rs.randn(B, D)))
    x = ko.bf16_round(ko.l2_normalize(rs.randn(N, D)))
    lab = rs.randint(0, C, size=N)
    _run_knn("knn_bf16_inputs", f, x, lab, C, k, T, store_inputs="bf16")

    # (iv) clustered features with label structure (realistic vote)
    rs = np.random.RandomState(11)
    B, N, D, C, k, T = 64, 2000, 64, 10, 200, 0.1
    cent = rs.randn(C, D)
    lab = rs.randint(0, C, size=N)
    x = ko.l2_normalize(cent[lab] + 0.5 * rs.randn(N, D))
    qlab = rs.randint(0, C, size=B)
    f = ko.l2_normalize(cent[qlab] + 0.5 * rs.randn(B, D))
    _run_knn("knn_clustered", f, x, lab, C, k, T, extra=None)

    # (ii) exact-arithmetic dyadic grid (every float32 summation order gives the same sums),
    #      wide D and duplicated bank rows (deliberate ties)
    rs = np.random.RandomState(12)
    B, N, D, C, k, T = 16, 768, 2048, 100, 200, 0.1
    fi = rs.randint(-4, 5, size=(B, D)).astype(np.int8)
    xi = rs.randint(-4, 5, size=(N, D)).astype(np.int8)
    dup = rs.randint(0, N // 2, size=64)
    xi[N // 2 + np.arange(64)] = xi[dup]                 # 64 duplicated rows -> exact ties
    lab = rs.randint(0, C, size=N)
    _run_knn("knn_dyadic_wide", fi.astype(np.float32) / 64, xi.astype(np.float32) / 64, lab, C, k, T,
             store_inputs="dyadic", extra=dict(feature_i8=fi, bank_rows_i8=xi, scale=np.float32(1 / 64)))

    rs = np.random.RandomState(13)
    B, N, D, C, k, T = 32, 512, 128, 10, 50, 0.1
    fi = rs.randint(-8, 9, size=(B, D)).astype(np.int8)
    xi = rs.randint(-8, 9, size=(N, D)).astype(np.int8)
    xi[256:384] = xi[0:128]                              # heavy ties
    lab = rs.randint(0, C, size=N)
    _run_knn("knn_dyadic_ties", fi.astype(np.float32) / 64, xi.astype(np.float32) / 64, lab, C, k, T,
             store_inputs="dyadic", extra=dict(feature_i8=fi, bank_rows_i8=xi, scale=np.float32(1 / 64)))


def bank_cases():
    # placeholder for the absent third-party package; see module docstring
    for modname in ("lumo", "lumo.proc", "lumo.proc.dist"):
        sys.modules.setdefault(modname, types.ModuleType(modname))
    sys.modules["lumo.proc.dist"].is_dist = lambda: False
    from models.memory_bank import MemoryBank, StorageBank  # the reference

    # --- MemoryBank.push traces.  Values encode (push id, row) so one snapshot proves provenance.
    out = {}
    for tag, (K, b, dim, n_push) in {
        "k10_b4": (10, 4, 3, 7),
        "k4096_b512": (4096, 512, 2, 11),
        "k65535_b512": (65535, 512, 1, 130),      # wrap with truncation (65535 % 512 != 0)
        "k65536_b256": (65536, 256, 1, 258),
        "k7_b9": (7, 9, 2, 3),                    # batch larger than the queue
        "k12_1d": (12, 5, -1, 6),                 # 1-D bank (dim <= 0)
    }.items():
        torch.manual_seed(0)
        mb = MemoryBank()
        mb.register("q", dim=dim, k=K)
        init = mb["q"].clone().numpy()
        offs = [mb.offsets["q"]]
        for p in range(n_push):
            shape = (b,) if dim <= 0 else (b, dim)
            val = (torch.arange(b, dtype=torch.float32) + 1000.0 * (p + 1)).reshape(-1, *([1] * (len(shape) - 1))).expand(shape).contiguous()
            mb.push("q", val)
            offs.append(mb.offsets["q"])
        final = mb["q"].numpy()
        written = final != init
        out[f"{tag}_params"] = np.array([K, b, dim, n_push], np.int64)
        out[f"{tag}_offsets"] = np.array(offs, np.int64)
        # provenance per row: value of column 0 where overwritten, -1 where the rand init survived
        col0 = final if final.ndim == 1 else final[:, 0]
        w0 = written if written.ndim == 1 else written.any(axis=1)
        out[f"{tag}_final"] = np.where(w0, col0, -1.0).astype(np.float32)
    # dtype long queue
    mbl = MemoryBank()
    mbl.register("ids", dim=-1, k=9, dtype=torch.float64)
    out["dtype_f64_init_ok"] = np.array([mbl["ids"].dtype == torch.float64], np.bool_)
    # register() distribution: U[0,1)
    torch.manual_seed(1)
    mbr = MemoryBank()
    mbr.register("neg", dim=8, k=256)
    r = mbr["neg"].numpy()
    out["rand_min_max_mean"] = np.array([r.min(), r.max(), r.mean()], np.float64)
    _save("memory_bank_push", **out)

    # --- StorageBank register / scatter
    sb = StorageBank()
    sb.register("test_feature", dim=4, k=10)
    sb.register("test_ys", dim=-1, k=10, dtype=torch.long)
    init_f = sb["test_feature"].clone().numpy()
    init_y = sb["test_ys"].clone().numpy()
    rs = np.random.RandomState(3)
    perm = rs.permutation(10)
    vals = rs.randn(10, 4).astype(np.float32)
    ys = rs.randint(0, 5, size=10).astype(np.int64)
    sb.scatter("test_feature", torch.from_numpy(vals[:6]), torch.from_numpy(perm[:6]))
    sb.scatter("test_ys", torch.from_numpy(ys[:6]), torch.from_numpy(perm[:6]))
    mid_f, mid_y = sb["test_feature"].clone().numpy(), sb["test_ys"].clone().numpy()
    sb.scatter("test_feature", torch.from_numpy(vals[6:]), torch.from_numpy(perm[6:]))
    sb.scatter("test_ys", torch.from_numpy(ys[6:]), torch.from_numpy(perm[6:]))
    _save("storage_bank", init_feature=init_f, init_ys=init_y, perm=perm.astype(np.int64), vals=vals, ys=ys,
          mid_feature=mid_f, mid_ys=mid_y, final_feature=sb["test_feature"].numpy(),
          final_ys=sb["test_ys"].numpy(), sizes=np.array([sb.sizes["test_feature"], sb.sizes["test_ys"]]))


def wrapping_queue_cases():
    """models/module_utils.py MemoryBank / LongTensorMemoryBank (the WRAPPING ring).
    The module imports ``lumo.core.BaseParams`` for an unrelated class (ModelParams); the absent
    third-party package is again satisfied with an inert placeholder, the two queue classes run as written."""
    for modname in ("lumo", "lumo.core"):
        sys.modules.setdefault(modname, types.ModuleType(modname))
    sys.modules["lumo.core"].BaseParams = type("BaseParams", (), {})
    from models.module_utils import LongTensorMemoryBank, MemoryBank  # the reference

    out = {}
    for tag, (K, b, dim, n_push) in {
        "k10_b4": (10, 4, 3, 7),                 # wraps with a two-segment write
        "k4096_b512": (4096, 512, 2, 11),
        "k65535_b512": (65535, 512, 1, 130),
        "k7_b7": (7, 7, 2, 3),                   # batch == queue
        "k9_b5": (9, 5, 1, 9),
    }.items():
        torch.manual_seed(0)
        mb = MemoryBank(queue_size=K, feature_dim=dim)
        cursors = [mb.cursor]
        for p in range(n_push):
            val = (torch.arange(b, dtype=torch.float32) + 1000.0 * (p + 1))[:, None].expand(b, dim).contiguous()
            mb.push(val)
            cursors.append(mb.cursor)
        out[f"{tag}_params"] = np.array([K, b, dim, n_push], np.int64)
        out[f"{tag}_cursors"] = np.array(cursors, np.int64)
        final = mb.tensor().numpy()
        out[f"{tag}_final"] = np.where(final[:, 0] >= 1000.0, final[:, 0], -1.0).astype(np.float32)   # rand init < 1
    lb = LongTensorMemoryBank(queue_size=11)
    cur = [lb.cursor]
    for p in range(6):
        lb.push(torch.arange(4, dtype=torch.long) + 100 * (p + 1))
        cur.append(lb.cursor)
    out["long_k11_b4_cursors"] = np.array(cur, np.int64)
    out["long_k11_b4_final"] = lb.tensor().numpy()
    # batch larger than the queue: the reference's second slice assignment cannot fit -> RuntimeError
    try:
        MemoryBank(queue_size=5, feature_dim=2).push(torch.zeros(12, 2))
        too_big = "ok"
    except RuntimeError:
        too_big = "RuntimeError"
    out["oversized_push"] = np.array([too_big])
    # batch longer than the queue but within 2K - cursor: both slice assignments fit, the rows written twice keep
    # the later item rows (K = 10, cursor 5, 12 rows: data[5:] = item[:5], data[:7] = item[5:], cursor 7)
    mb = MemoryBank(queue_size=10, feature_dim=2)
    mb.push(torch.full((5, 2), 500.0))
    mb.push((torch.arange(12, dtype=torch.float32) + 1000.0)[:, None].expand(12, 2).contiguous())
    out["overlong_k10_c5_b12_final"] = mb.tensor().numpy().copy()
    out["overlong_k10_c5_b12_cursor"] = np.array([mb.cursor], np.int64)
    c = MemoryBank(queue_size=6, feature_dim=2)
    c.push(torch.ones(4, 2))
    out["attrs_survive"] = np.array([c.clone().cursor, c.to(torch.float64).cursor, c.clone().queue_size], np.int64)
    _save("wrapping_queue", **out)


def comatch_smooth_cases():
    """track_semi/comatch.py:185-189 is a closure inside CoMatchTrainer.train_step (needs lumo's Trainer: not
    importable).  Its three torch expressions are evaluated here verbatim on torch CPU; only data is stored."""
    out = {}
    for tag, (b, K, D, C, T, alpha, seed) in {
        "cifar10": (448, 1792, 64, 10, 0.2, 0.9, 0),          # batch 64 x mu 7, feature_dim 64
        "cifar100": (192, 1500, 64, 100, 0.2, 0.9, 1),
        "ragged": (37, 333, 128, 7, 0.1, 0.5, 2),
    }.items():
        g = torch.Generator().manual_seed(seed)
        un_w_featrure = torch.nn.functional.normalize(torch.randn(b, D, generator=g), dim=-1)
        mb_feature = torch.nn.functional.normalize(torch.randn(K, D, generator=g), dim=-1)
        mb_feature[: b // 2] = torch.nn.functional.normalize(un_w_featrure[: b // 2] + 0.1 * torch.randn(b // 2, D, generator=g), dim=-1)
        mb_prob = torch.softmax(3 * torch.randn(K, C, generator=g), dim=1)
        un_w_probs = torch.softmax(2 * torch.randn(b, C, generator=g), dim=1)
        A = torch.exp(torch.mm(un_w_featrure, mb_feature.t()) / T)                       # comatch.py:187
        A = A / A.sum(1, keepdim=True)                                                   # comatch.py:188
        probs = alpha * un_w_probs + (1 - alpha) * torch.mm(A, mb_prob)                  # comatch.py:189
        out.update({f"{tag}_feature": un_w_featrure.numpy(), f"{tag}_mb_feature": mb_feature.numpy(),
                    f"{tag}_mb_prob": mb_prob.numpy(), f"{tag}_probs_in": un_w_probs.numpy(),
                    f"{tag}_probs_out": probs.numpy(), f"{tag}_T_alpha": np.array([T, alpha], np.float64)})
    _save("comatch_smooth", **out)


def kth_threshold_cases():
    """track_semi/chmatch.py:252-253 (`thresh, _ = self.mem['fine'].topk(int(k)); thresh[-1]`) evaluated verbatim by
    torch CPU on score banks of the reference's size (mlen 50 000, chmatch.py:37) -- a closure in a lumo Trainer,
    so the expression is executed here, not imported."""
    out = {}
    g = torch.Generator().manual_seed(0)
    banks = {
        "rand": torch.rand(50000, generator=g),                                       # MemoryBank.register init
        "scores": torch.softmax(4 * torch.randn(50000, 10, generator=g), dim=1).max(dim=1).values,
        "ties": (torch.randint(0, 50, (50000,), generator=g).float() / 50.0),
        "small": torch.randn(7, generator=g),
    }
    for name, fine in banks.items():
        n = fine.numel()
        ks = sorted({1, 2, n // 10 + 1, n // 2, int(0.8 * n), n})
        res = []
        for k in ks:
            thresh, _ = fine.topk(int(k))
            res.append(float(thresh[-1]))
        out[f"{name}_bank"] = fine.numpy()
        out[f"{name}_ks"] = np.array(ks, np.int64)
        out[f"{name}_thresh"] = np.array(res, np.float32)
    _save("kth_threshold", **out)


if __name__ == "__main__":
    kth_threshold_cases()
    comatch_smooth_cases()
    knn_cases()
    bank_cases()
    wrapping_queue_cases()
