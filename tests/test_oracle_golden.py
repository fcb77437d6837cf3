"""CPU tests: the oracle (numpy restatement + exact-chain C) against the reference's golden vectors."""
import numpy as np
import pytest

from oracle import chain
from oracle import knn_oracle as ko
from tests import parity as P


@pytest.mark.parametrize("name", P.KNN_CASES)
def test_numpy_oracle_matches_reference(name):
    z = P.load_case(name)
    D = z["feature"].shape[1]
    exact = name in P.EXACT_CASES
    out = ko.knn_predict_full(z["feature"], np.ascontiguousarray(z["bank_rows"].T), z["labels"],
                              z["classes"], z["knn_k"], z["knn_t"])
    P.check_topk(out["sim_weight"], out["sim_indices"], z["sim_weight"], z["sim_indices"],
                 z["kth_gap"], D, exact=exact, what=name)
    P.check_lower_index_ties(out["sim_weight"], out["sim_indices"], name)
    rows = P.decided_rows(z["kth_gap"], D, exact)
    P.check_scores(out["pred_scores"], z["pred_scores"], name, rows=rows)
    n = P.check_top1(out["pred_labels"], z["pred_labels"], z["pred_scores"], name, rows=rows)
    assert n > 0 or name in P.EXACT_CASES
    P.check_ranking_prefix(out["pred_labels"], out["pred_scores"], name)
    assert out["pred_labels"].shape == z["pred_labels"].shape and out["pred_labels"].dtype == np.int64


@pytest.mark.parametrize("name", P.KNN_CASES)
def test_chain_oracle_matches_reference(name):
    """The k-ascending fmaf chain (the fp32 kernel's defined order) against torch.mm's values."""
    z = P.load_case(name)
    D = z["feature"].shape[1]
    exact = name in P.EXACT_CASES
    sim = chain.sim_chain_f32(z["feature"], z["bank_rows"])
    out = ko.knn_from_similarity(sim, z["labels"], z["classes"], z["knn_k"], z["knn_t"])
    P.check_topk(out["sim_weight"], out["sim_indices"], z["sim_weight"], z["sim_indices"],
                 z["kth_gap"], D, exact=exact, what=name)
    rows = P.decided_rows(z["kth_gap"], D, exact)
    P.check_scores(out["pred_scores"], z["pred_scores"], name, rows=rows)
    P.check_top1(out["pred_labels"], z["pred_labels"], z["pred_scores"], name, rows=rows)
    v2, i2 = chain.topk_rows(sim, z["knn_k"])
    assert np.array_equal(v2, out["sim_weight"]) and np.array_equal(i2, out["sim_indices"])


@pytest.mark.parametrize("name", ["knn_rand_tiny", "knn_rand_cifar100", "knn_clustered", "knn_rand_ragged"])
def test_torch_cpu_port_matches_reference(name):
    """The timed cpu_baseline port (oracle/knn_torch_cpu.py), un-chunked and chunked."""
    import torch
    from oracle import knn_torch_cpu as kt
    z = P.load_case(name)
    f = torch.from_numpy(z["feature"])
    bank = torch.from_numpy(np.ascontiguousarray(z["bank_rows"].T))
    lab = torch.from_numpy(z["labels"])
    full = kt.knn_predict_chunked(f, bank, lab, z["classes"], z["knn_k"], z["knn_t"]).numpy()
    assert np.array_equal(full, z["pred_labels"])
    part = kt.knn_predict_chunked(f, bank, lab, z["classes"], z["knn_k"], z["knn_t"], chunk=5).numpy()
    P.check_top1(part, z["pred_labels"], z["pred_scores"], name)


def test_exact_cases_are_order_independent():
    for name in P.EXACT_CASES:
        z = P.load_case(name)
        a = chain.sim_chain_f32(z["feature"], z["bank_rows"])
        b = ko.similarity(z["feature"], np.ascontiguousarray(z["bank_rows"].T))
        c = (z["feature"].astype(np.float64) @ z["bank_rows"].astype(np.float64).T).astype(np.float32)
        assert np.array_equal(a, b) and np.array_equal(a, c)
        # ties exist and the reference resolved them differently or not -- values must still agree
        assert (np.diff(np.sort(a, axis=1), axis=1) == 0).any()


def test_bf16_oracle_pair():
    z = P.load_case("knn_bf16_inputs")
    qb, xb = ko.to_bf16_bits(z["feature"]), ko.to_bf16_bits(z["bank_rows"])
    assert np.array_equal(ko.from_bf16_bits(qb), z["feature"])          # inputs are bf16-representable
    s64 = chain.sim_bf16_f64(qb, xb)
    s32 = chain.sim_chain_f32(z["feature"], z["bank_rows"])
    assert np.max(np.abs(s64 - s32)) <= P.val_atol(z["feature"].shape[1])
    x = np.array([1.0, 1.00390625, 1.005859375, -3.1415927, 0.0], np.float32)
    assert np.array_equal(ko.bf16_round(x), np.array([1.0, 1.0, 1.0078125, -3.140625, 0.0], np.float32))


def test_merge_topk_is_sharding_invariant():
    rs = np.random.RandomState(0)
    sim = rs.randn(5, 400).astype(np.float32)
    sim[:, 100:140] = sim[:, 300:340]          # cross-shard ties
    k = 32
    v_ref, i_ref = ko.topk_desc(sim, k)
    for G in (1, 2, 4, 8):
        n = 400 // G
        cv = np.stack([ko.topk_desc(sim[:, g * n:(g + 1) * n], k)[0] for g in range(G)])
        ci = np.stack([ko.topk_desc(sim[:, g * n:(g + 1) * n], k)[1] + g * n for g in range(G)])
        v, i = ko.merge_topk(cv, ci, k)
        assert np.array_equal(v, v_ref) and np.array_equal(i, i_ref)


def test_topk_k_larger_than_n_raises():
    with pytest.raises(ValueError):
        ko.topk_desc(np.zeros((2, 3), np.float32), 4)


# ----------------------------------------------------------------------------- banks
def _replay(tag, z):
    K, b, dim, n_push = [int(v) for v in z[f"{tag}_params"]]
    mb = ko.MemoryBankOracle()
    mb.register("q", dim=dim, k=K)
    mb["q"] = np.full_like(mb["q"], -1.0)
    offs = [mb.offsets["q"]]
    for p in range(n_push):
        shape = (b,) if dim <= 0 else (b, dim)
        val = np.broadcast_to((np.arange(b, dtype=np.float32) + 1000.0 * (p + 1)).reshape(-1, *([1] * (len(shape) - 1))), shape)
        mb.push("q", val)
        offs.append(mb.offsets["q"])
    final = mb["q"] if mb["q"].ndim == 1 else mb["q"][:, 0]
    return np.array(offs), final


@pytest.mark.parametrize("tag", ["k10_b4", "k4096_b512", "k65535_b512", "k65536_b256", "k7_b9", "k12_1d"])
def test_memory_bank_push_trace(tag):
    z = np.load(P.GOLDEN + "/memory_bank_push.npz")
    offs, final = _replay(tag, z)
    assert np.array_equal(offs, z[f"{tag}_offsets"])
    assert np.array_equal(final, z[f"{tag}_final"])
    K, b = int(z[f"{tag}_params"][0]), int(z[f"{tag}_params"][1])
    ptr = 0
    for o in z[f"{tag}_offsets"][1:]:
        w, ptr = ko.ring_push_plan(ptr, K, b)
        assert ptr == o and 0 < w <= b


def test_memory_bank_known_truncation():
    """K=10, b=4 -> pointers 0,4,8,0,4 and the third push writes only 2 rows."""
    z = np.load(P.GOLDEN + "/memory_bank_push.npz")
    assert z["k10_b4_offsets"][:5].tolist() == [0, 4, 8, 0, 4]
    mn, mx, mean = z["rand_min_max_mean"]
    assert 0.0 <= mn and mx < 1.0 and abs(mean - 0.5) < 0.05       # register() = U[0, 1)


def test_storage_bank_matches_reference():
    z = np.load(P.GOLDEN + "/storage_bank.npz")
    sb = ko.StorageBankOracle()
    sb.register("test_feature", dim=4, k=10)
    sb.register("test_ys", dim=-1, k=10, dtype=np.int64)
    assert np.array_equal(sb["test_feature"], z["init_feature"]) and np.all(np.isinf(sb["test_feature"]))
    assert np.array_equal(sb["test_ys"], z["init_ys"])
    sb.scatter("test_feature", z["vals"][:6], z["perm"][:6])
    sb.scatter("test_ys", z["ys"][:6], z["perm"][:6])
    assert np.array_equal(sb["test_feature"], z["mid_feature"]) and np.array_equal(sb["test_ys"], z["mid_ys"])
    sb.scatter("test_feature", z["vals"][6:], z["perm"][6:])
    sb.scatter("test_ys", z["ys"][6:], z["perm"][6:])
    assert np.array_equal(sb["test_feature"], z["final_feature"]) and np.array_equal(sb["test_ys"], z["final_ys"])
    assert sb.sizes["test_feature"] == int(z["sizes"][0])


def test_moco_loss_restatement_against_torch():
    """Parity unpinned (lumo absent): check the restatement against torch's own CE + autograd."""
    import torch
    import torch.nn.functional as F
    rs = np.random.RandomState(0)
    q = ko.l2_normalize(rs.randn(16, 32))
    k = ko.l2_normalize(rs.randn(16, 32))
    mem = ko.moco_queue_init(100, 32, rs)
    loss, dq, lse, lpos = ko.moco_loss(q, k, mem, 0.1)
    tq = torch.tensor(q, dtype=torch.float64, requires_grad=True)
    logits = torch.cat([(tq * torch.tensor(k, dtype=torch.float64)).sum(1, keepdim=True),
                        tq @ torch.tensor(mem, dtype=torch.float64).t()], dim=1) / 0.1
    tl = F.cross_entropy(logits, torch.zeros(16, dtype=torch.long))
    tl.backward()
    assert abs(loss - tl.item()) < 1e-12
    assert np.max(np.abs(dq - tq.grad.numpy())) < 1e-12
