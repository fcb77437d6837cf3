"""Parity at BASELINE.json's full sizes (GPU only).  The CPU oracle cannot finish these in seconds,
so the checks are size-independent properties plus a sampled recomputation on the device:

  * every row: values sorted descending, indices unique and in range, ties in lower-index order;
  * a sample of query rows is recomputed with torch (float32 matmul over the SAME rounded inputs, top-k)
    and compared: values within 5e-6, index sets equal on rows whose k-th gap exceeds 2e-5 (tolerances and
    why they are tighter than tests/parity.py's worst-case bound: see VAL_ATOL below);
  * row-sharding invariance: G shards with global index bases + deterministic merge == the unsharded result
    bit for bit;
  * the vote's ranking is a permutation of the classes, descending in score.
"""
import numpy as np
import pytest
import torch

from oracle import chain
from oracle import knn_oracle as ko
from tests import parity as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ica():
    import image_classification_amd as pkg
    pkg._lib.lib()
    return pkg


def unit_rows(n, d, seed, chunk=65536):
    g = torch.Generator(device="cuda").manual_seed(seed)
    out = torch.empty((n, d), dtype=torch.float32, device="cuda")
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        out[s:e] = torch.nn.functional.normalize(torch.randn((e - s, d), generator=g, device="cuda"), dim=1)
    return out


def check_rows_wellformed(val, idx, n):
    assert bool((val[:, 1:] <= val[:, :-1]).all()), "values not sorted descending"
    assert int(idx.min()) >= 0 and int(idx.max()) < n
    srt = torch.sort(idx, dim=1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all()), "duplicate neighbour indices in a row"
    same = val[:, 1:] == val[:, :-1]
    assert bool((idx[:, 1:][same] > idx[:, :-1][same]).all()), "tie order is not lower-index-first"


def sampled_reference(feature, bank_rows, rows, k, dtype, chunk=131072):
    """torch float32 top-k of the sampled query rows over the whole bank (inputs rounded like the kernel's)."""
    rnd = (lambda t: t.to(torch.bfloat16).float()) if dtype == "bf16" else (lambda t: t)
    q = rnd(feature[rows])
    best_v = torch.full((len(rows), 0), 0.0, device="cuda")
    best_i = torch.zeros((len(rows), 0), dtype=torch.int64, device="cuda")
    for s in range(0, bank_rows.shape[0], chunk):
        sim = q @ rnd(bank_rows[s:s + chunk]).t()
        v, i = sim.topk(min(k + 1, sim.shape[1]), dim=1)
        best_v, best_i = torch.cat([best_v, v], 1), torch.cat([best_i, i + s], 1)
        v, o = best_v.topk(min(k + 1, best_v.shape[1]), dim=1)
        best_v, best_i = v, torch.gather(best_i, 1, o)
    return best_v, best_i                     # k + 1 columns: the last one gives the k-th gap


# Float32-accumulated dot products of unit vectors computed in two different orders differ by a random-walk
# rounding error of ~sqrt(D) * eps * |terms| (1e-7 at D = 2048), far below the worst-case 4*D*eps of
# tests/parity.py.  At these bank sizes neighbouring order statistics are ~3e-5 apart, so the worst-case bound
# would leave no row "decided"; the full-size contract is therefore: values within VAL_ATOL, index sets equal on
# rows whose k-th gap exceeds GAP_TOL = 4 * VAL_ATOL, and at most one boundary swap on the others.
VAL_ATOL = 5e-6
GAP_TOL = 4 * VAL_ATOL


def compare_sample(val, idx, feature, bank_rows, k, dtype, n_sample=96, seed=0):
    rows = torch.from_numpy(np.random.RandomState(seed).choice(feature.shape[0], n_sample, replace=False)).cuda()
    rv, ri = sampled_reference(feature, bank_rows, rows, k, dtype)
    gap = (rv[:, k - 1] - rv[:, k]).cpu().numpy()
    v, i = val[rows].cpu().numpy(), idx[rows].cpu().numpy()
    rvn, rin = rv[:, :k].cpu().numpy(), ri[:, :k].cpu().numpy()
    assert np.max(np.abs(v - rvn)) <= VAL_ATOL, f"{dtype}: top-k values differ by {np.max(np.abs(v - rvn))}"
    decided = 0
    for b in range(n_sample):
        common = len(set(i[b].tolist()) & set(rin[b].tolist()))
        if gap[b] > GAP_TOL:
            assert common == k, f"{dtype}: sampled row {b}: index set differs ({common} of {k} common, gap {gap[b]})"
            decided += 1
        else:
            assert common >= k - 1, f"{dtype}: sampled row {b}: {k - common} indices differ on an undecided boundary"
    assert decided >= n_sample // 3, f"only {decided} of {n_sample} sampled rows had a decided k-th boundary"


def test_c4_imagenet1k_bf16_full_size(ica):
    """BASELINE configs[3]: 50 000 queries x 1 280 000 x 2048 bank, k=200, bf16, C=1000 (one GPU holds it all)."""
    B, N, D, C, k, T = 50_000, 1_280_000, 2048, 1000, 200, 0.1
    bank_rows = unit_rows(N, D, 0)
    feature = unit_rows(B, D, 1)
    labels = torch.randint(0, C, (N,), generator=torch.Generator(device="cuda").manual_seed(2), device="cuda")
    bank = ica.BankIndex.from_rows(bank_rows, "bf16")
    vals, idxs = [], []
    for s in range(0, B, 16384):
        v, i = ica.knn_topk(feature[s:s + 16384], bank, k)
        vals.append(v)
        idxs.append(i)
    val, idx = torch.cat(vals), torch.cat(idxs)
    check_rows_wellformed(val, idx, N)
    compare_sample(val, idx, feature, bank_rows, k, "bf16")
    # sharding invariance on a slice of the queries: 4 row shards + merge == unsharded, bit for bit
    sub = feature[:2048]
    cv, ci = [], []
    for g in range(4):
        lo, hi = g * N // 4, (g + 1) * N // 4
        v, i = ica.knn_topk(sub, ica.BankIndex.from_rows(bank_rows[lo:hi], "bf16", idx_base=lo), k)
        cv.append(v)
        ci.append(i)
    mv, mi = ica.knn_merge(torch.stack(cv), torch.stack(ci), k)
    assert torch.equal(mv, val[:2048]) and torch.equal(mi, idx[:2048])
    # vote: permutation of the classes, descending in score, top-1 = argmax of the scores
    rank, scores = ica.knn_vote(val[:4096], idx[:4096], labels, C, T, return_scores=True)
    P.check_ranking_prefix(rank.cpu().numpy(), scores.cpu().numpy(), "C4 vote")
    w = torch.exp(val[:4096] / T)
    ref = torch.zeros((4096, C), device="cuda").scatter_add_(1, labels[idx[:4096]], w)
    assert torch.allclose(scores, ref, rtol=1e-4, atol=0)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_c3_imagenet100_full_size(ica, dtype):
    """BASELINE configs[2]: 5 000 queries x 126 000 x 2048 bank, k=200, through the reference's [D, N] layout."""
    B, N, D, C, k, T = 5_000, 126_000, 2048, 100, 200, 0.1
    bank_rows = unit_rows(N, D, 3)
    feature = unit_rows(B, D, 4)
    labels = torch.randint(0, C, (N,), generator=torch.Generator(device="cuda").manual_seed(5), device="cuda")
    out = ica.knn_predict_full(feature, bank_rows.t().contiguous(), labels, C, k, T, compute_dtype=dtype)
    check_rows_wellformed(out["sim_weight"], out["sim_indices"], N)
    compare_sample(out["sim_weight"], out["sim_indices"], feature, bank_rows, k, dtype)
    P.check_ranking_prefix(out["pred_labels"].cpu().numpy(), out["pred_scores"].cpu().numpy(), "C3 vote")
    assert tuple(out["pred_labels"].shape) == (B, C)


def test_c5_oversized_bank_one_shard_fp32(ica):
    """BASELINE configs[4]: 10M x 256 fp32 over 8 GPUs -> ONE rank's shard (1.25M x 256) with its global index base."""
    n_local, D, k, B = 1_250_000, 256, 200, 4096
    base = 5 * n_local                                   # rank 5 of 8
    bank_rows = unit_rows(n_local, D, 6)
    feature = unit_rows(B, D, 7)
    val, idx = ica.knn_topk(feature, ica.BankIndex.from_rows(bank_rows, "fp32", idx_base=base), k)
    check_rows_wellformed(val, idx - base, n_local)
    compare_sample(val, idx - base, feature, bank_rows, k, "fp32", n_sample=64)


def test_c5_oversized_bank_full_size_eight_shards_fp32(ica):
    """BASELINE configs[4] at its real size: the whole 10M x 256 fp32 bank (10.2 GB, it fits one MI355X) cut into the
    eight row shards the 8-GPU job would hold, each searched with its global index base, then ``mi_knn_merge`` --
    bit-equal to ONE unsharded ``mi_knn_topk`` over all 10M rows; plus the sampled torch recomputation."""
    N, D, k, B, G = 10_000_000, 256, 200, 4096, 8
    bank_rows = unit_rows(N, D, 6)
    feature = unit_rows(B, D, 7)
    v_ref, i_ref = ica.knn_topk(feature, ica.BankIndex.from_rows(bank_rows, "fp32"), k)
    check_rows_wellformed(v_ref, i_ref, N)
    sharded = ica.sharded
    cv, ci = [], []
    for g in range(G):
        lo, hi = sharded.shard_bounds(N, G, g)
        v, i = ica.knn_topk(feature, ica.BankIndex.from_rows(bank_rows[lo:hi], "fp32", idx_base=lo), k)
        assert int(i.min()) >= lo and int(i.max()) < hi
        cv.append(v)
        ci.append(i)
    mv, mi = ica.knn_merge(torch.stack(cv), torch.stack(ci), k)
    assert torch.equal(mv, v_ref) and torch.equal(mi, i_ref), "8 shards + merge differ from the unsharded top-k"
    compare_sample(v_ref, i_ref, feature, bank_rows, k, "fp32", n_sample=64)


def test_c1_cifar100_full_size_fp32_bit_exact_on_sampled_rows(ica):
    """BASELINE configs[0] at full size (10 000 queries x 50 000 x 128 bank, C = 100) through the reference's [D, N]
    layout, fp32: 256 sampled query rows are recomputed by the CHAIN ORACLE (oracle/knn_chain.c) against the whole
    bank -- values and indices bit-equal, vote scores within 1e-5 relative, top-1 equal where the scores decide it."""
    B, N, D, C, k, T = 10_000, 50_000, 128, 100, 200, 0.1
    bank_rows = unit_rows(N, D, 8)
    feature = unit_rows(B, D, 9)
    labels = torch.randint(0, C, (N,), generator=torch.Generator(device="cuda").manual_seed(10), device="cuda")
    out = ica.knn_predict_full(feature, bank_rows.t().contiguous(), labels, C, k, T, compute_dtype="fp32")
    check_rows_wellformed(out["sim_weight"], out["sim_indices"], N)
    assert tuple(out["pred_labels"].shape) == (B, C)
    rows = np.sort(np.random.RandomState(1).choice(B, 256, replace=False))
    ref = ko.knn_from_similarity(chain.sim_chain_f32(feature[rows].cpu().numpy(), bank_rows.cpu().numpy()),
                                 labels.cpu().numpy(), C, k, T)
    got = {n: v[torch.from_numpy(rows).cuda()].cpu().numpy() for n, v in out.items()}
    assert np.array_equal(got["sim_weight"], ref["sim_weight"]), "C1: similarities differ from the chain oracle"
    assert np.array_equal(got["sim_indices"], ref["sim_indices"]), "C1: top-k indices differ from the chain oracle"
    P.check_scores(got["pred_scores"], ref["pred_scores"], "C1")
    P.check_ranking_prefix(got["pred_labels"], got["pred_scores"], "C1")
    assert P.check_top1(got["pred_labels"], ref["pred_labels"], ref["pred_scores"], "C1") > 200


def test_c1_cifar100_full_size_bf16_vs_bf16_oracle(ica):
    """The same configuration in the throughput mode against the bf16 oracle (exact products of the bf16-rounded
    inputs, float64 accumulation) on 128 sampled rows: values within 4*D*eps, index sets equal where the k-th gap
    exceeds twice that (tests/parity.py contract)."""
    B, N, D, C, k, T = 10_000, 50_000, 128, 100, 200, 0.1
    bank_rows = unit_rows(N, D, 8)
    feature = unit_rows(B, D, 9)
    labels = torch.randint(0, C, (N,), generator=torch.Generator(device="cuda").manual_seed(10), device="cuda")
    out = ica.knn_predict_full(feature, bank_rows.t().contiguous(), labels, C, k, T, compute_dtype="bf16")
    check_rows_wellformed(out["sim_weight"], out["sim_indices"], N)
    rows = np.sort(np.random.RandomState(2).choice(B, 128, replace=False))
    sim = chain.sim_bf16_f64(ko.to_bf16_bits(feature[rows].cpu().numpy()), ko.to_bf16_bits(bank_rows.cpu().numpy()))
    ref = ko.knn_from_similarity(sim, labels.cpu().numpy(), C, k, T)
    srt = -np.sort(-sim, axis=1)
    gap = srt[:, k - 1] - srt[:, k]
    got = {n: v[torch.from_numpy(rows).cuda()].cpu().numpy() for n, v in out.items()}
    n = P.check_topk(got["sim_weight"], got["sim_indices"], ref["sim_weight"], ref["sim_indices"], gap, D, False, "C1 bf16")
    assert n >= 16, f"only {n} sampled rows had a decided k-th boundary"
    decided = P.decided_rows(gap, D)
    rtol = 1e-5 + P.val_atol(D) / T
    P.check_scores(got["pred_scores"], ref["pred_scores"], "C1 bf16", rtol=rtol, rows=decided)
    P.check_top1(got["pred_labels"], ref["pred_labels"], ref["pred_scores"], "C1 bf16", rtol=rtol, rows=decided)
