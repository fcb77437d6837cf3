"""GPU parity tests (run on the MI355X box: ``pytest -m gpu``).  Everything goes through the C ABI
(``libmi_knn.so``) via the package's ctypes binding; the oracle is only the checker."""
import sys

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
    pkg._lib.lib()                       # fail loudly if the HIP library is missing
    return pkg


def dev(x, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def run_full(ica, feature, bank_rows, labels, C, k, T, compute_dtype, layout="reference"):
    f = dev(feature)
    if layout == "reference":
        bank = dev(np.ascontiguousarray(bank_rows.T))                      # [D, N] like ssltrainer.py:179
    elif layout == "view":
        bank = dev(bank_rows).t()                                          # transposed view of [N, D]
    else:
        bank = ica.BankIndex.from_rows(dev(bank_rows), compute_dtype)
    out = ica.knn_predict_full(f, bank, dev(labels), C, k, T, compute_dtype=compute_dtype)
    torch.cuda.synchronize()
    return {n: v.cpu().numpy() for n, v in out.items()}


# ----------------------------------------------------------------------------- fp32: bit-exact vs the chain oracle
@pytest.mark.parametrize("name", P.KNN_CASES)
@pytest.mark.parametrize("layout", ["reference", "rows"])
def test_fp32_bit_exact_vs_chain_oracle(ica, name, layout):
    z = P.load_case(name)
    sim = chain.sim_chain_f32(z["feature"], z["bank_rows"])
    ref = ko.knn_from_similarity(sim, z["labels"], z["classes"], z["knn_k"], z["knn_t"])
    out = run_full(ica, z["feature"], z["bank_rows"], z["labels"], z["classes"], z["knn_k"], z["knn_t"], "fp32", layout)
    # integer / index work and the fp32 similarities: bit-exact
    assert np.array_equal(out["sim_weight"], ref["sim_weight"]), \
        f"{name}: similarities differ, max |d| = {np.max(np.abs(out['sim_weight'] - ref['sim_weight']))}"
    assert np.array_equal(out["sim_indices"], ref["sim_indices"]), f"{name}: top-k indices differ"
    # floating point vote: 1e-5 relative (expf implementations differ in the last ulp)
    P.check_scores(out["pred_scores"], ref["pred_scores"], name)
    P.check_ranking_prefix(out["pred_labels"], out["pred_scores"], name)
    P.check_top1(out["pred_labels"], ref["pred_labels"], ref["pred_scores"], name)
    assert out["pred_labels"].dtype == np.int64 and out["pred_labels"].shape == (z["feature"].shape[0], z["classes"])


@pytest.mark.parametrize("name", P.KNN_CASES)
def test_fp32_vs_reference_golden(ica, name):
    """Against the REFERENCE's own outputs (torch.mm / topk / vote), SURVEY 8(c) contract."""
    z = P.load_case(name)
    D = z["feature"].shape[1]
    exact = name in P.EXACT_CASES
    out = run_full(ica, z["feature"], z["bank_rows"], z["labels"], z["classes"], z["knn_k"], z["knn_t"], "fp32")
    P.check_topk(out["sim_weight"], out["sim_indices"], z["sim_weight"], z["sim_indices"], z["kth_gap"], D, exact, name)
    P.check_lower_index_ties(out["sim_weight"], out["sim_indices"], name)
    rows = P.decided_rows(z["kth_gap"], D, exact)
    P.check_scores(out["pred_scores"], z["pred_scores"], name, rows=rows)
    P.check_top1(out["pred_labels"], z["pred_labels"], z["pred_scores"], name, rows=rows)


def test_knn_predict_signature_and_view_layout(ica):
    z = P.load_case("knn_rand_cifar100")
    f, lab = dev(z["feature"]), dev(z["labels"])
    a = ica.knn_predict(f, dev(np.ascontiguousarray(z["bank_rows"].T)), lab, z["classes"], z["knn_k"], z["knn_t"])
    b = ica.knn_predict(f, dev(z["bank_rows"]).t(), lab, z["classes"], z["knn_k"], z["knn_t"])   # zero-copy view
    assert a.dtype == torch.int64 and tuple(a.shape) == (f.shape[0], z["classes"]) and a.is_cuda
    assert torch.equal(a, b)


# ----------------------------------------------------------------------------- bf16 mode
@pytest.mark.parametrize("name", ["knn_bf16_inputs", "knn_rand_cifar100", "knn_clustered", "knn_rand_ragged"])
def test_bf16_vs_bf16_oracle(ica, name):
    """Oracle for bf16 mode = reference arithmetic on bf16-rounded inputs.  The MFMA's
    internal add order is hardware defined, so values get a tolerance and index sets are compared
    where the k-th gap exceeds it."""
    z = P.load_case(name)
    D = z["feature"].shape[1]
    qb, xb = ko.to_bf16_bits(z["feature"]), ko.to_bf16_bits(z["bank_rows"])
    sim = chain.sim_bf16_f64(qb, xb)
    k = z["knn_k"]
    ref = ko.knn_from_similarity(sim, z["labels"], z["classes"], k, z["knn_t"])
    srt = -np.sort(-sim, axis=1)
    gap = srt[:, k - 1] - srt[:, k] if k < sim.shape[1] else np.full(sim.shape[0], np.inf, np.float32)
    out = run_full(ica, z["feature"], z["bank_rows"], z["labels"], z["classes"], k, z["knn_t"], "bf16")
    n = P.check_topk(out["sim_weight"], out["sim_indices"], ref["sim_weight"], ref["sim_indices"], gap, D, False, name)
    assert n > 0
    rows = P.decided_rows(gap, D)
    # bf16 similarities differ from the float64-accumulated oracle by float32 rounding only;
    # exp(s/T) amplifies that by 1/T: tolerance 1e-5 + D*eps/T relative
    rtol = 1e-5 + P.val_atol(D) / z["knn_t"]
    P.check_scores(out["pred_scores"], ref["pred_scores"], name, rtol=rtol, rows=rows)
    P.check_top1(out["pred_labels"], ref["pred_labels"], ref["pred_scores"], name, rtol=rtol, rows=rows)


@pytest.mark.parametrize("name", P.EXACT_CASES)
def test_bf16_exact_arithmetic_cases_bit_exact(ica, name):
    """Dyadic-grid inputs are bf16-representable and every add order is exact: bf16 == fp32 == oracle."""
    z = P.load_case(name)
    sim = chain.sim_chain_f32(z["feature"], z["bank_rows"])
    ref = ko.knn_from_similarity(sim, z["labels"], z["classes"], z["knn_k"], z["knn_t"])
    out = run_full(ica, z["feature"], z["bank_rows"], z["labels"], z["classes"], z["knn_k"], z["knn_t"], "bf16", "rows")
    assert np.array_equal(out["sim_weight"], ref["sim_weight"])
    assert np.array_equal(out["sim_indices"], ref["sim_indices"])


# ----------------------------------------------------------------------------- multi-stage / adversarial / edge cases
def _rand_case(B, N, D, C, seed):
    rs = np.random.RandomState(seed)
    return ko.l2_normalize(rs.randn(B, D)), ko.l2_normalize(rs.randn(N, D)), rs.randint(0, C, size=N)


@pytest.mark.parametrize("B,N,D,k", [(300, 60000, 64, 200), (1, 5000, 32, 7), (513, 2049, 96, 200), (40, 30000, 160, 1024)])
def test_fp32_multi_stage_bit_exact(ica, B, N, D, k):
    f, x, lab = _rand_case(B, N, D, 10, B + N)
    sim = chain.sim_chain_f32(f, x)
    v_ref, i_ref = chain.topk_rows(sim, k)
    v, i = ica.knn_topk(dev(f), ica.BankIndex.from_rows(dev(x), "fp32"), k)
    assert np.array_equal(v.cpu().numpy(), v_ref) and np.array_equal(i.cpu().numpy(), i_ref)


def test_adversarial_ascending_similarity(ica):
    """Every later bank row beats every earlier one: all elements pass the stale threshold, the
    candidate lists fill to their worst case and must not overflow."""
    rs = np.random.RandomState(7)
    D, N, B, k = 32, 30000, 9, 50
    u = ko.l2_normalize(rs.randn(1, D))[0]
    x = (np.linspace(0.01, 1.0, N, dtype=np.float32)[:, None] * u[None, :]).astype(np.float32)
    f = np.tile(u, (B, 1)).astype(np.float32) * np.linspace(0.5, 1.0, B, dtype=np.float32)[:, None]
    sim = chain.sim_chain_f32(f, x)
    v_ref, i_ref = chain.topk_rows(sim, k)
    v, i = ica.knn_topk(dev(f), ica.BankIndex.from_rows(dev(x), "fp32"), k)
    assert np.array_equal(v.cpu().numpy(), v_ref) and np.array_equal(i.cpu().numpy(), i_ref)


# ----------------------------------------------------------------------------- speculative first threshold + exact fallback
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("B,N,D,k,mod", [(700, 60000, 64, 200, 3), (300, 9000, 128, 50, 1), (1000, 200000, 32, 200, 7)])
def test_forced_fallback_redoes_rows_exactly(ica, dtype, B, N, D, k, mod):
    """mi_knn_topk filters from a speculative first threshold and redoes the rows it failed for.  With
    mi_debug_force_fallback every mod-th row is REPORTED failed, so the fallback path (one whole-bank stage from the safe
    threshold, final refresh into the original rows) runs for real: the result must equal the undisturbed run bit for
    bit, and (fp32) the chain oracle."""
    f, x, lab = _rand_case(B, N, D, 10, B + N + k)
    bank = ica.BankIndex.from_rows(dev(x), dtype)
    v0, i0 = ica.knn_topk(dev(f), bank, k)
    L = ica._lib.lib()
    ica._lib.check(L.mi_debug_force_fallback(mod), "mi_debug_force_fallback")
    try:
        v1, i1 = ica.knn_topk(dev(f), bank, k)
        torch.cuda.synchronize()
    finally:
        ica._lib.check(L.mi_debug_force_fallback(0), "mi_debug_force_fallback")
    assert torch.equal(v0, v1) and torch.equal(i0, i1)
    if dtype == "fp32":
        v_ref, i_ref = chain.topk_rows(chain.sim_chain_f32(f, x), k)
        assert np.array_equal(v1.cpu().numpy(), v_ref) and np.array_equal(i1.cpu().numpy(), i_ref)


@pytest.mark.parametrize("n_bad", [5, 300])
def test_speculation_that_fails_falls_back_exactly(ica, n_bad):
    """A bank the strided sample misjudges: exactly the sampled rows (every floor(N / 2048)-th) resemble the first n_bad
    queries, no other row does.  The sample then promises ~N / 2048 times more near neighbours than the bank holds, the
    speculative threshold lets fewer than k rows through for those queries, and they must come out exact all the same
    (fp32: bit-equal to the chain oracle); the other queries see an ordinary random bank."""
    rs = np.random.RandomState(11)
    B, N, D, k = 400, 50000, 64, 200
    stride = N // 2048
    u = ko.l2_normalize(rs.randn(1, D))[0]
    x = ko.l2_normalize(rs.randn(N, D))
    x[np.arange(2048) * stride] = ko.l2_normalize(u[None, :] + 0.05 * rs.randn(2048, D))
    f = ko.l2_normalize(rs.randn(B, D))
    f[:n_bad] = ko.l2_normalize(u[None, :] + 0.05 * rs.randn(n_bad, D))
    x, f = x.astype(np.float32), f.astype(np.float32)
    v_ref, i_ref = chain.topk_rows(chain.sim_chain_f32(f, x), k)
    assert np.all(np.isin(i_ref[:n_bad], np.arange(2048) * stride))          # their neighbours ARE the sampled rows
    v, i = ica.knn_topk(dev(f), ica.BankIndex.from_rows(dev(x), "fp32"), k)
    assert np.array_equal(v.cpu().numpy(), v_ref) and np.array_equal(i.cpu().numpy(), i_ref)


def test_all_ties_pick_lowest_indices(ica):
    D, N, B, k = 32, 7000, 5, 200
    x = np.tile(np.full((1, D), 0.125, np.float32), (N, 1))
    f = np.full((B, D), 0.25, np.float32)
    for dt in ("fp32", "bf16"):
        v, i = ica.knn_topk(dev(f), ica.BankIndex.from_rows(dev(x), dt), k)
        assert np.array_equal(i.cpu().numpy(), np.tile(np.arange(k), (B, 1)))
        assert np.all(v.cpu().numpy() == np.float32(D * 0.125 * 0.25))


@pytest.mark.parametrize("D,N,B", [(64, 60000, 2100), (32, 40000, 20000)])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_every_element_survives_long_runs_of_short_rows(ica, dtype, D, N, B):
    """Short rows (D = 64: the walk kernel) hand over a row's survivors of a whole RUN of bank tiles as one list.  Here
    every element of the bank passes every threshold (all similarities tie at 2.0, exact in both dtypes), so each list is
    as long as it can get -- run x 256 entries, far beyond one tile's 256 and one refresh batch -- and the k real
    neighbours (twice that similarity) sit at the very end of the bank, behind all of them.  The second shape has 79 query
    tiles, so a workgroup's run is as long as the bank slice of a launch (one run per row)."""
    k, hi = 200, 300
    x = np.full((N, D), 0.125, np.float32)
    x[N - hi:] = 0.25
    f = np.full((B, D), 0.25, np.float32)
    lo_sim, hi_sim = np.float32(D * 0.125 * 0.25), np.float32(D * 0.25 * 0.25)
    v, i = ica.knn_topk(dev(f), ica.BankIndex.from_rows(dev(x), dtype), k)
    assert torch.equal(i, torch.arange(N - hi, N - hi + k, device=i.device).expand(B, k))
    assert bool((v == float(hi_sim)).all())
    # ... and with nothing above the ties: the lowest rows win
    v, i = ica.knn_topk(dev(f), ica.BankIndex.from_rows(dev(x[:N - hi]), dtype), k)
    assert torch.equal(i, torch.arange(k, device=i.device).expand(B, k))
    assert bool((v == float(lo_sim)).all())


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_unwritten_query_rows_of_the_storage_bank(ica, dtype):
    """``StorageBank.register`` fills ``test_feature`` with +inf (memory_bank.py:16-18) and ``knn_test`` hands the
    whole buffer to ``knn_predict`` (ssltrainer.py:184): a row no ``test_step`` has written is all +inf.  Its
    similarities are inf * x summed over mixed signs = NaN.  The reference then returns an ARBITRARY ranking for it
    (torch.topk orders NaN first, exp(NaN) = NaN, and 0 * NaN = NaN poisons every class of the one-hot product,
    knn.py:18; argsort of an all-NaN row is unspecified).  The rule here (mi_knn.h, mi_knn_topk): NaN similarities are
    never selected, the row is padded with (-inf, -1), receives no vote and ranks the classes in ascending order --
    and the written rows that share its 256-row tile are not touched by it."""
    f, x, lab = _rand_case(300, 6000, 64, 7, 31)
    bad = [0, 5, 255, 256, 299]
    f_inf = f.copy()
    f_inf[bad] = np.inf
    C, k, T = 7, 200, 0.1
    good = np.setdiff1d(np.arange(300), bad)
    ref = run_full(ica, f, x, lab, C, k, T, dtype)
    out = run_full(ica, f_inf, x, lab, C, k, T, dtype)
    for n in ref:
        assert np.array_equal(out[n][good], ref[n][good]), f"{n}: a written row changed because a neighbour row is +inf"
    assert np.all(np.isneginf(out["sim_weight"][bad])) and np.all(out["sim_indices"][bad] == -1)
    assert np.all(out["pred_scores"][bad] == 0) and np.array_equal(out["pred_labels"][bad], np.tile(np.arange(C), (len(bad), 1)))
    # what torch does with the same row, for the record in DESIGN.md: all-NaN similarities and all-NaN scores
    sim = torch.from_numpy(f_inf[:1]) @ torch.from_numpy(x.T.copy())
    assert bool(torch.isnan(sim).all())
    # a +inf row against a same-signed bank is +inf, not NaN: selected like any other value (lower rows first)
    xp = np.abs(x)
    v, i = ica.knn_topk(dev(f_inf[:1]), ica.BankIndex.from_rows(dev(xp), dtype), 5)
    assert np.all(np.isposinf(v.cpu().numpy())) and i.cpu().numpy().tolist() == [[0, 1, 2, 3, 4]]


def test_result_does_not_depend_on_the_workspace_size(ica):
    """mi_knn_topk sizes its filter stages by the workspace it is given (mi_knn.h): from the minimum (2048-column
    stages) over the 8 GiB default to a budget that fits the whole bank in one stage, the stage schedule changes
    completely and the result must not -- bit for bit, through the C ABI."""
    f, x, _ = _rand_case(300, 40000, 64, 5, 123)
    k = 200
    bank = ica.BankIndex.from_rows(dev(x), "fp32")
    q = ica.pack_rows(dev(f), False, "fp32")
    L = ica._lib.lib()
    b_pad = 512
    minimum = int(L.mi_knn_topk_workspace_bytes_for_budget(300, 40000, k, 1))           # below the minimum -> the default plan
    sizes = {
        "tiny": b_pad * (8 + k * 8) + 64 * 1024 + b_pad * 2048 * 8 + b_pad * 40 + 8192,  # just above the documented minimum
        "mid": int(L.mi_knn_topk_workspace_bytes_for_budget(300, 40000, k, 64 << 20)),
        "default": int(L.mi_knn_topk_workspace_bytes(300, 40000, k)),
        "one-stage": int(L.mi_knn_topk_workspace_bytes_for_budget(300, 40000, k, 1 << 40)),
    }
    assert minimum == sizes["default"] and sizes["tiny"] < sizes["mid"] <= sizes["default"] == sizes["one-stage"]
    results = {}
    for name, nbytes in sizes.items():
        ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        val = torch.empty((300, k), dtype=torch.float32, device="cuda")
        idx = torch.empty((300, k), dtype=torch.int64, device="cuda")
        ica._lib.check(L.mi_knn_topk(q.data.data_ptr(), q.ld, bank.packed.data.data_ptr(), bank.packed.ld, 0, 300, 40000, 64, k, 0,
                                     val.data_ptr(), idx.data_ptr(), ws.data_ptr(), nbytes, ica._lib.stream_ptr(val.device)), name)
        torch.cuda.synchronize()
        results[name] = (val, idx)
    v_ref, i_ref = chain.topk_rows(chain.sim_chain_f32(f, x), k)
    for name, (val, idx) in results.items():
        assert np.array_equal(val.cpu().numpy(), v_ref) and np.array_equal(idx.cpu().numpy(), i_ref), name
    ws = torch.empty(4096, dtype=torch.uint8, device="cuda")
    rc = L.mi_knn_topk(q.data.data_ptr(), q.ld, bank.packed.data.data_ptr(), bank.packed.ld, 0, 300, 40000, 64, k, 0,
                       val.data_ptr(), idx.data_ptr(), ws.data_ptr(), 4096, ica._lib.stream_ptr(val.device))
    assert rc == -3 and b"below the minimum" in L.mi_last_error()


def test_k_larger_than_n_raises_like_torch(ica):
    f, x, _ = _rand_case(4, 10, 32, 3, 0)
    with pytest.raises(RuntimeError, match="out of range"):
        ica.knn_topk(dev(f), ica.BankIndex.from_rows(dev(x), "fp32"), 11)


def test_bad_arguments(ica):
    f, x, lab = _rand_case(4, 10, 32, 3, 0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ica.knn_predict(torch.from_numpy(f), torch.from_numpy(x.T.copy()), torch.from_numpy(lab), 3, 2, 0.1)
    with pytest.raises(ValueError):
        ica.knn_predict(dev(f), dev(x.T.copy()), dev(lab[:5]), 3, 2, 0.1)
    with pytest.raises(ValueError):
        ica.knn_topk(dev(f[:, :16]), ica.BankIndex.from_rows(dev(x), "fp32"), 2)
    with pytest.raises((ValueError, RuntimeError)):
        ica.knn_topk(dev(f), ica.BankIndex.from_rows(dev(np.tile(x, (300, 1))), "fp32"), 2000)    # k > MI_MAX_K


def test_index_base_and_merge_equals_unsharded(ica):
    """Row-sharded bank: local top-k with global indices, deterministic merge == single-shard result."""
    f, x, lab = _rand_case(70, 9000, 64, 10, 3)
    x[4000:4100] = x[100:200]                               # cross-shard exact ties
    k = 200
    v_ref, i_ref = ica.knn_topk(dev(f), ica.BankIndex.from_rows(dev(x), "fp32"), k)
    for G in (2, 4, 8):
        n = x.shape[0] // G
        vs, is_ = [], []
        for g in range(G):
            lo, hi = g * n, (x.shape[0] if g == G - 1 else (g + 1) * n)
            v, i = ica.knn_topk(dev(f), ica.BankIndex.from_rows(dev(x[lo:hi]), "fp32", idx_base=lo), k)
            vs.append(v)
            is_.append(i)
        v, i = ica.knn_merge(torch.stack(vs), torch.stack(is_), k)
        assert torch.equal(v, v_ref) and torch.equal(i, i_ref), f"G={G}"
        vo, io = ko.merge_topk(torch.stack(vs).cpu().numpy(), torch.stack(is_).cpu().numpy(), k)
        assert np.array_equal(v.cpu().numpy(), vo) and np.array_equal(i.cpu().numpy(), io)


@pytest.mark.parametrize("G,kc,k", [(8, 200, 200), (8, 256, 256), (7, 256, 200), (3, 5, 9), (2, 64, 1), (1, 200, 200), (4, 300, 300)])
def test_merge_kernel_vs_oracle(ica, G, kc, k):
    """``mi_knn_merge`` on synthetic candidate blocks against ``oracle.merge_topk``: both wave-per-row instantiations
    (<= 1664 and <= 2048 candidates per row), the workgroup kernel beyond them (k = 300), values that tie across shards
    (the lower global row wins), the k-th value itself shared by several rows, padding entries (index -1)."""
    rs = np.random.RandomState(G * 1000 + kc)
    B, n_local = 133, 5000
    vals = np.round(rs.rand(G, B, kc).astype(np.float32) * 64) / 64          # 65 distinct values: masses of ties
    vals[:, 5] = 0.5                                                          # one row where EVERY candidate ties
    vals = -np.sort(-vals, axis=2)
    idx = np.stack([g * n_local + np.stack([rs.permutation(n_local)[:kc] for _ in range(B)]) for g in range(G)]).astype(np.int64)
    # within equal values a shard lists the lower row first (what mi_knn_topk emits)
    order = np.lexsort((idx, -vals), axis=2)
    vals, idx = np.take_along_axis(vals, order, 2), np.take_along_axis(idx, order, 2)
    if kc > 4 and G * (kc - 2) >= k:
        vals[0, 7, kc - 2:], idx[0, 7, kc - 2:] = -np.inf, -1                # a shard that held fewer than kc rows
    v, i = ica.knn_merge(dev(vals), dev(idx), k)
    vo, io = ko.merge_topk(vals, idx, k)
    assert np.array_equal(v.cpu().numpy(), vo) and np.array_equal(i.cpu().numpy(), io)


@pytest.mark.parametrize("chunk", [256, 300, 4096])
def test_chunked_driver_equals_single_call(ica, chunk):
    """ShardedKnn (world 1): the query-chunked driver gives the single-call result bit for bit."""
    f, x, lab = _rand_case(1100, 12000, 64, 10, 21)
    k = 200
    v_ref, i_ref = ica.knn_topk(dev(f), ica.BankIndex.from_rows(dev(x), "bf16"), k)
    knn = ica.sharded.ShardedKnn(dev(x), 0, compute_dtype="bf16", query_chunk=chunk, shard_sizes=[12000])
    v, i = knn.topk(dev(f), k)
    assert torch.equal(v, v_ref) and torch.equal(i, i_ref)
    pred = knn.predict(dev(f), dev(lab), 10, k, 0.1)
    ref = ica.knn_predict(dev(f), ica.BankIndex.from_rows(dev(x), "bf16"), dev(lab), 10, k, 0.1)
    assert torch.equal(pred, ref)


@pytest.mark.parametrize("C,k", [(1, 5), (10, 200), (1000, 200), (5000, 64), (65536, 33)])
def test_vote_kernel_vs_oracle(ica, C, k):
    rs = np.random.RandomState(C + k)
    B, N = 37, 5000
    val = -np.sort(-rs.uniform(-1, 1, size=(B, k)).astype(np.float32), axis=1)
    idx = np.stack([rs.permutation(N)[:k] for _ in range(B)]).astype(np.int64)
    lab = rs.randint(0, C, size=N)
    T = 0.1
    rank, scores = ica.knn_vote(dev(val), dev(idx), dev(lab), C, T, return_scores=True)
    w = ko.vote_weights(val, T)
    s_ref = ko.vote_scores(w, lab[idx], C)
    P.check_scores(scores.cpu().numpy(), s_ref, f"C={C}")
    P.check_ranking_prefix(rank.cpu().numpy(), scores.cpu().numpy(), f"C={C}")
    P.check_top1(rank.cpu().numpy(), ko.rank_classes(s_ref), s_ref, f"C={C}")


def test_vote_skips_padding_entries(ica):
    val = dev(np.array([[0.9, 0.5, -np.inf]], np.float32))
    idx = dev(np.array([[2, 0, -1]], np.int64))
    lab = dev(np.array([0, 2, 1], np.int64))
    rank, scores = ica.knn_vote(val, idx, lab, 3, 0.5, return_scores=True)
    s = scores.cpu().numpy()[0]
    assert s[2] == 0 and np.allclose(s[:2], [np.exp(1.0), np.exp(1.8)], rtol=1e-6)
    assert rank.cpu().numpy()[0].tolist() == [1, 0, 2]


def test_labels_outside_the_class_range_raise_like_the_reference(ica):
    """models/knn.py:16 (`scatter` of the one-hot) raises for a label >= classes; the HIP vote would silently drop the
    neighbour, so the host checks the range first -- in knn_predict, knn_vote and the sharded predict."""
    f, x, lab = _rand_case(9, 500, 32, 5, 4)
    bad = lab.copy()
    bad[7] = 5
    with pytest.raises(IndexError, match="classes=5"):
        ica.knn_predict(dev(f), dev(x.T.copy()), dev(bad), 5, 20, 0.1)
    bad[7] = -1
    with pytest.raises(IndexError):
        ica.knn_predict(dev(f), dev(x.T.copy()), dev(bad), 5, 20, 0.1)
    with pytest.raises(IndexError):
        ica.sharded.ShardedKnn(dev(x), 0, compute_dtype="fp32").predict(dev(f), dev(bad), 5, 20, 0.1)
    assert tuple(ica.knn_predict(dev(f), dev(x.T.copy()), dev(lab), 5, 20, 0.1).shape) == (9, 5)


def test_label_range_is_validated_once_per_tensor_and_again_after_an_in_place_change(ica, monkeypatch):
    """check_labels remembers a validated label tensor (no host read per vote) until it is modified in place."""
    f, x, lab = _rand_case(9, 500, 32, 5, 4)
    labels, bank = dev(lab), ica.BankIndex.from_rows(dev(x), "fp32")
    calls = []
    real = torch.aminmax
    monkeypatch.setattr(torch, "aminmax", lambda t: (calls.append(1), real(t))[1])
    for _ in range(3):
        ica.knn_predict(dev(f), bank, labels, 5, 20, 0.1)
    assert len(calls) == 1                                   # one pass + host read for three monitor passes
    with pytest.raises(IndexError, match="classes=4"):       # the remembered range still answers a smaller class count
        ica.knn_predict(dev(f), bank, labels, 4, 20, 0.1)
    assert len(calls) == 1
    labels[3] = 7                                            # in-place change: the version counter moves
    with pytest.raises(IndexError, match="classes=5"):
        ica.knn_predict(dev(f), bank, labels, 5, 20, 0.1)
    assert len(calls) == 2


def test_workspace_allocation_falls_back_to_smaller_plans(ica, monkeypatch):
    """A failed allocation of the preferred candidate workspace retries with smaller plans; the result is the same."""
    f, x, lab = _rand_case(300, 40000, 64, 10, 5)
    bank = ica.BankIndex.from_rows(dev(x), "fp32")
    v_ref, i_ref = ica.knn_topk(dev(f), bank, 50)
    real, sizes = torch.empty, []

    def failing_empty(*a, **kw):
        if kw.get("dtype") is torch.uint8 and isinstance(a[0], int) and a[0] > (64 << 20):
            sizes.append(a[0])
            raise torch.OutOfMemoryError("simulated: HBM held by the training job")
        return real(*a, **kw)

    knn_mod = sys.modules[ica.knn_topk.__module__]
    monkeypatch.setattr(knn_mod.torch, "empty", failing_empty)
    v, i = ica.knn_topk(dev(f), bank, 50)
    monkeypatch.undo()
    assert sizes and sizes == sorted(sizes, reverse=True)    # larger plans were tried first and refused
    assert torch.equal(v, v_ref) and torch.equal(i, i_ref)


def test_deterministic_across_runs(ica):
    f, x, lab = _rand_case(257, 20000, 128, 100, 11)
    a = ica.knn_predict_full(dev(f), dev(x.T.copy()), dev(lab), 100, 200, 0.1, compute_dtype="bf16")
    b = ica.knn_predict_full(dev(f), dev(x.T.copy()), dev(lab), 100, 200, 0.1, compute_dtype="bf16")
    for n in a:
        assert torch.equal(a[n], b[n]), n


def test_pack_rows(ica):
    rs = np.random.RandomState(0)
    x = rs.randn(70, 50).astype(np.float32)
    for dt, tdt, ld in (("fp32", torch.float32, 64), ("bf16", torch.bfloat16, 64)):
        p = ica.pack_rows(dev(x), False, dt)
        pt = ica.pack_rows(dev(np.ascontiguousarray(x.T)), True, dt)
        assert p.data.shape == (70, ld) and p.data.dtype == tdt
        ref = torch.zeros(70, ld)
        ref[:, :50] = torch.from_numpy(x)
        ref = ref.to(tdt)
        assert torch.equal(p.data.cpu(), ref) and torch.equal(pt.data.cpu(), ref)
    y = dev(rs.randn(8, 64).astype(np.float32))
    assert ica.pack_rows(y, False, "fp32").data.data_ptr() == y.data_ptr()           # zero copy


def test_rccl_branches_on_one_rank_group(ica):
    """The RCCL (backend "nccl") branches of ShardedKnn -- all_to_all_single of the packed candidates,
    all_gather_into_tensor of the rankings / top-k -- run on a one-rank group on this one-GPU box and must
    reproduce the unsharded result bit for bit (more ranks: the gloo tests of the same orchestration)."""
    import os
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29617")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        f, x, lab = _rand_case(700, 9000, 64, 10, 5)
        k = 200
        bank = ica.BankIndex.from_rows(dev(x), "bf16")
        ref = ica.knn_predict(dev(f), bank, dev(lab), 10, k, 0.1)
        v_ref, i_ref = ica.knn_topk(dev(f), bank, k)
        knn = ica.sharded.ShardedKnn(dev(x), 0, compute_dtype="bf16", query_chunk=256, always_exchange=True)
        assert not knn.solo and knn.shard_sizes == [9000]
        assert torch.equal(knn.predict(dev(f), dev(lab), 10, k, 0.1), ref)
        assert torch.equal(knn.predict(dev(f), dev(lab), 10, k, 0.1, gather="top1"), ref[:, 0])
        assert torch.equal(knn.local_ranking, ref) and torch.equal(knn.local_rows.cpu(), torch.arange(700))
        v, i = knn.topk(dev(f), k)
        assert torch.equal(v, v_ref) and torch.equal(i, i_ref)
        # the message sizes of a real 8-rank job (bench.py --gpus 8: one 50 000-row chunk against a 160 000 x 2048 shard,
        # 80 MB of packed candidates through all_to_all_single, 50 000 top-1 labels through all_gather_into_tensor)
        g = torch.Generator(device="cuda").manual_seed(5)
        xb = torch.nn.functional.normalize(torch.randn((160_000, 2048), generator=g, device="cuda"), dim=1)
        fb = torch.nn.functional.normalize(torch.randn((50_000, 2048), generator=g, device="cuda"), dim=1)
        labb = torch.randint(0, 1000, (160_000,), generator=g, device="cuda")
        big = ica.sharded.ShardedKnn(xb, 0, compute_dtype="bf16", query_chunk=None, always_exchange=True)
        top1 = big.predict(fb, labb, 1000, k, 0.1, gather="top1")
        ref_big = ica.knn_predict(fb, big.bank, labb, 1000, k, 0.1)
        assert torch.equal(top1, ref_big[:, 0]) and torch.equal(big.local_ranking, ref_big)
    finally:
        dist.destroy_process_group()


_SWEEP = [(B, N, D, k) for B, N, D, k in [
    (5, 2048, 32, 1), (70, 2049, 32, 63), (33, 2300, 64, 64), (257, 4097, 32, 65), (64, 9001, 96, 255), (129, 8193, 32, 256),
    (31, 12345, 64, 257), (700, 3000, 32, 200), (16, 70000, 32, 200), (300, 2047, 36, 199), (2, 300, 20, 300), (90, 26000, 256, 100)]]


@pytest.mark.parametrize("B,N,D,k", _SWEEP)
def test_fp32_shape_sweep_bit_exact(ica, B, N, D, k):
    """Odd shapes around every boundary of the refresh kernels (k at 1 / 63-65 / 255-257: wave-per-row vs
    workgroup select; N at 2047-2049: bootstrap only vs one filter stage; ragged tiles; D with padding), with a
    clustered bank so that many rows carry far more than 64 candidates per stage: values and indices bit-equal to
    the chain oracle."""
    rs = np.random.RandomState(B * 7 + N + k)
    centers = rs.randn(8, D)
    f = ko.l2_normalize(centers[rs.randint(0, 8, size=B)] + 0.3 * rs.randn(B, D))
    x = ko.l2_normalize(centers[rs.randint(0, 8, size=N)] + 0.3 * rs.randn(N, D))
    x[N // 2:N // 2 + 5] = x[:5]                                  # exact ties, lower row first
    sim = chain.sim_chain_f32(f, x)
    v_ref, i_ref = chain.topk_rows(sim, k)
    v, i = ica.knn_topk(dev(f), ica.BankIndex.from_rows(dev(x), "fp32"), k)
    assert np.array_equal(v.cpu().numpy(), v_ref) and np.array_equal(i.cpu().numpy(), i_ref)


def test_knn_predict_replays_from_a_hip_graph(ica):
    """The whole monitor pass (query pack, bootstrap, every filter stage and refresh, vote, ranking) is stream
    ordered -- no host synchronisation, no allocation inside the library -- so it captures into a HIP graph
    (torch.cuda.CUDAGraph) and replays on new query values with the eager result, bit for bit."""
    f, x, lab = _rand_case(600, 20000, 64, 10, 77)
    f2 = _rand_case(600, 20000, 64, 10, 78)[0]
    bank = ica.BankIndex.from_rows(dev(x), "bf16")
    labels = dev(lab)
    fin = dev(f)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                                  # warm-up outside the capture (library load, allocator)
        ica.knn_predict(fin, bank, labels, 10, 200, 0.1)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = ica.knn_predict(fin, bank, labels, 10, 200, 0.1)
    for feats in (f, f2):
        fin.copy_(dev(feats))
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ica.knn_predict(dev(feats), bank, labels, 10, 200, 0.1))
