"""Parity at BASELINE.json's full sizes (GPU only).  The CPU oracle cannot recompute every query at these sizes
in seconds, so the checks are size-independent properties plus an ORACLE recomputation of a sample of query rows
against the WHOLE bank:

  * every row: values sorted descending, indices unique and in range, ties in lower-index order;
  * a sample of query rows is recomputed by the CPU oracle (oracle/knn_chain.c) over the whole bank, which is fed to
    it in slices of 65 536 rows to bound host memory:
      fp32  -- ``chain.sim_chain_f32`` (the k-ascending fmaf chain): top-k values AND indices bit-equal;
      bf16  -- ``chain.sim_bf16_f64`` (exact products of the bf16-rounded inputs, float64 accumulation): values within
               VAL_ATOL, index sets equal on rows whose k-th gap exceeds GAP_TOL (see below);
    the vote of those rows is checked against ``ko.knn_from_similarity`` (scores, ranking, top-1);
  * row-sharding invariance: G shards with global index bases + deterministic merge == the unsharded result
    bit for bit;
  * the vote's ranking is a permutation of the classes, descending in score.

No value in this file is checked against a torch-ROCm computation: torch only generates the inputs, rounds them to
bf16 for the oracle (round-to-nearest-even, cross-checked against the oracle's own rounding) and moves bytes.
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


SLICE = 65536          # bank rows handed to the oracle at a time (64 k x 2048 float32 = 512 MB of host memory)


def bf16_bits(t):
    """uint16 bfloat16 bit patterns of a float32 device tensor (torch's round-to-nearest-even cast; byte movement)."""
    return t.to(torch.bfloat16).view(torch.int16).cpu().numpy().view(np.uint16)


def oracle_similarity(feature, bank_rows, rows, dtype):
    """[len(rows), N] float32 similarities of the sampled query rows against the WHOLE bank, by the CPU oracle."""
    rows_t = torch.from_numpy(np.asarray(rows)).cuda()
    N = bank_rows.shape[0]
    sim = np.empty((len(rows), N), np.float32)
    if dtype == "bf16":
        q = bf16_bits(feature[rows_t])
        # the device cast and the oracle's own rounding are the same function (checked on the sampled rows)
        assert np.array_equal(q, ko.to_bf16_bits(feature[rows_t].cpu().numpy()))
    else:
        q = feature[rows_t].cpu().numpy()
    for s0 in range(0, N, SLICE):
        e0 = min(N, s0 + SLICE)
        if dtype == "bf16":
            sim[:, s0:e0] = chain.sim_bf16_f64(q, bf16_bits(bank_rows[s0:e0]))
        else:
            sim[:, s0:e0] = chain.sim_chain_f32(q, bank_rows[s0:e0].cpu().numpy())
    return sim


# bf16 mode.  The oracle value is the exactly-rounded sum of the exact products; the MFMA accumulates the same
# products in float32 in a hardware order: a random-walk rounding error of ~sqrt(D) * eps * |terms| (1e-7 at
# D = 2048), far below the worst-case 4*D*eps of tests/parity.py.  At these bank sizes neighbouring order
# statistics are ~3e-5 apart, so the worst-case bound would leave no row "decided"; the full-size bf16 contract is
# therefore: values within VAL_ATOL, index sets equal on rows whose k-th gap exceeds GAP_TOL = 4 * VAL_ATOL, and at
# most one boundary swap on the others.  (fp32 mode needs no tolerance: bit-equal.)
VAL_ATOL = 5e-6
GAP_TOL = 4 * VAL_ATOL


def compare_with_oracle(val, idx, feature, bank_rows, k, dtype, n_sample, seed=0, labels=None, classes=0, T=0.1,
                        scores=None, rank=None, what=""):
    """Sampled query rows: the HIP result against the oracle over the whole bank.  Returns the sampled rows."""
    rows = np.sort(np.random.RandomState(seed).choice(feature.shape[0], n_sample, replace=False))
    rows_t = torch.from_numpy(rows).cuda()
    sim = oracle_similarity(feature, bank_rows, rows, dtype)
    lab = labels.cpu().numpy() if labels is not None else np.zeros(bank_rows.shape[0], np.int64)
    ref = ko.knn_from_similarity(sim, lab, max(classes, 1), k, T)
    v, i = val[rows_t].cpu().numpy(), idx[rows_t].cpu().numpy()
    if dtype == "fp32":
        assert np.array_equal(v, ref["sim_weight"]), f"{what}: similarities differ from the chain oracle"
        assert np.array_equal(i, ref["sim_indices"]), f"{what}: top-k indices differ from the chain oracle"
        decided = np.ones(n_sample, bool)
        rtol = P.SCORE_RTOL
    else:
        kth = -np.partition(-sim, k, axis=1)[:, :k + 1]
        kth.sort(axis=1)
        gap = kth[:, 1] - kth[:, 0]                       # k-th minus (k+1)-th largest
        assert np.max(np.abs(v - ref["sim_weight"])) <= VAL_ATOL, \
            f"{what}: top-k values differ from the bf16 oracle by {np.max(np.abs(v - ref['sim_weight']))}"
        decided = gap > GAP_TOL
        for b in range(n_sample):
            common = len(set(i[b].tolist()) & set(ref["sim_indices"][b].tolist()))
            if decided[b]:
                assert common == k, f"{what}: sampled row {b}: index set differs ({common} of {k} common, gap {gap[b]})"
            else:
                assert common >= k - 1, f"{what}: sampled row {b}: {k - common} indices differ on an undecided boundary"
        assert decided.sum() >= n_sample // 3, f"only {decided.sum()} of {n_sample} sampled rows had a decided k-th boundary"
        rtol = P.SCORE_RTOL + VAL_ATOL / T                # exp(s / T): a value error of VAL_ATOL scales the weight by VAL_ATOL / T
    if scores is not None:
        P.check_scores(scores[rows_t].cpu().numpy(), ref["pred_scores"], what, rtol=rtol, rows=decided)
        got_rank = rank[rows_t].cpu().numpy()
        P.check_ranking_prefix(got_rank, scores[rows_t].cpu().numpy(), what)
        n = P.check_top1(got_rank, ref["pred_labels"], ref["pred_scores"], what, rtol=rtol, rows=decided)
        assert n >= decided.sum() // 2, f"{what}: only {n} sampled rows had a decided top-1"
    return rows


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
    # the packed bank the kernel reads is the round-to-nearest-even bf16 image of the float32 rows
    assert np.array_equal(bank.packed.data[:4096].view(torch.int16).cpu().numpy().view(np.uint16),
                          ko.to_bf16_bits(bank_rows[:4096].cpu().numpy()))
    # 16 sampled queries x the whole 1.28 M-row bank through the bf16 oracle, vote at C = 1000 included
    rank_s, scores_s = ica.knn_vote(val, idx, labels, C, T, return_scores=True)
    compare_with_oracle(val, idx, feature, bank_rows, k, "bf16", n_sample=16, labels=labels, classes=C, T=T,
                        scores=scores_s, rank=rank_s, what="C4 bf16")
    del rank_s, scores_s
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
    # 4096 rows of vote scores against the oracle's vote of the SAME neighbours (ko.vote_scores: models/knn.py:14-18)
    w = ko.vote_weights(val[:4096].cpu().numpy(), T)
    ref = ko.vote_scores(w, labels.cpu().numpy()[idx[:4096].cpu().numpy()], C)
    P.check_scores(scores.cpu().numpy(), ref, "C4 vote")


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_c3_imagenet100_full_size(ica, dtype):
    """BASELINE configs[2]: 5 000 queries x 126 000 x 2048 bank, k=200, through the reference's [D, N] layout."""
    B, N, D, C, k, T = 5_000, 126_000, 2048, 100, 200, 0.1
    bank_rows = unit_rows(N, D, 3)
    feature = unit_rows(B, D, 4)
    labels = torch.randint(0, C, (N,), generator=torch.Generator(device="cuda").manual_seed(5), device="cuda")
    out = ica.knn_predict_full(feature, bank_rows.t().contiguous(), labels, C, k, T, compute_dtype=dtype)
    check_rows_wellformed(out["sim_weight"], out["sim_indices"], N)
    # fp32: 64 sampled queries x the whole bank through the chain oracle, bit-equal; bf16: 32 through the bf16 oracle
    compare_with_oracle(out["sim_weight"], out["sim_indices"], feature, bank_rows, k, dtype,
                        n_sample=64 if dtype == "fp32" else 32, labels=labels, classes=C, T=T,
                        scores=out["pred_scores"], rank=out["pred_labels"], what=f"C3 {dtype}")
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
    compare_with_oracle(val, idx - base, feature, bank_rows, k, "fp32", n_sample=32, what="C5 shard")   # bit-equal


def test_c5_oversized_bank_full_size_eight_shards_fp32(ica):
    """BASELINE configs[4] at its real size: the whole 10M x 256 fp32 bank (10.2 GB, it fits one MI355X) cut into the
    eight row shards the 8-GPU job would hold, each searched with its global index base, then ``mi_knn_merge`` --
    bit-equal to ONE unsharded ``mi_knn_topk`` over all 10M rows; plus 8 sampled queries recomputed by the chain oracle
    over all 10 M rows, bit-equal."""
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
    compare_with_oracle(v_ref, i_ref, feature, bank_rows, k, "fp32", n_sample=8, what="C5 10M")         # bit-equal, all 10 M rows


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
