"""GPU tests of the MoCo queue logits / InfoNCE (track_ssl/moco.py:152-163) through the C ABI.

Parity status: the reference's arithmetic lives in the third-party ``lumo`` package, which is
absent -> PARITY UNPINNED against the reference itself.  Checked here against the
float64 restatement of the MoCo-paper form in ``oracle/knn_oracle.py::moco_loss`` and against
``torch.mm`` + ``F.cross_entropy`` autograd on the same device.  Floating point: loss / lse within
2e-5 relative, dq within 1e-4 of max|dq| (fp32 MFMA, exp in fp32)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import knn_oracle as ko

pytestmark = pytest.mark.gpu

LOSS_RTOL = 2e-5
DQ_RTOL = 1e-4


@pytest.fixture(scope="module")
def ica():
    import image_classification_amd as pkg
    pkg._lib.lib()
    return pkg


def make(b, K, D, seed, sharp=False):
    rs = np.random.RandomState(seed)
    q = ko.l2_normalize(rs.randn(b, D))
    k = ko.l2_normalize(q + 0.3 * rs.randn(b, D))
    mem = ko.moco_queue_init(K, D, rs)
    if sharp:                                   # a few queue rows nearly equal to queries: peaked softmax
        mem[: min(b, K)] = ko.l2_normalize(q[: min(b, K)] + 0.05 * rs.randn(min(b, K), D))
    return q, k, mem


def run(ica, q, k, mem, T):
    tq = torch.from_numpy(q).cuda().requires_grad_(True)
    loss = ica.contrastive_loss2(query=tq, key=torch.from_numpy(k).cuda(), memory=torch.from_numpy(mem).cuda(),
                                 query_neg=False, key_neg=False, temperature=T, norm=False)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.item()), tq.grad.cpu().numpy()


@pytest.mark.parametrize("b,K,D,T,sharp", [
    (256, 4096, 128, 0.1, False),        # reference defaults (moco.py:33-39)
    (256, 65536, 128, 0.1, False),       # BASELINE configs[1]
    (37, 1000, 64, 0.2, False),          # ragged everything
    (300, 5000, 256, 0.07, True),
    (1, 64, 64, 0.1, True),
    (130, 257, 192, 0.5, False),
])
def test_loss_and_grad_vs_float64_oracle(ica, b, K, D, T, sharp):
    q, k, mem = make(b, K, D, b + K + D, sharp)
    loss_ref, dq_ref, _, _ = ko.moco_loss(q, k, mem, T)
    loss, dq = run(ica, q, k, mem, T)
    assert abs(loss - loss_ref) <= LOSS_RTOL * abs(loss_ref), (loss, loss_ref)
    scale = np.max(np.abs(dq_ref))
    assert np.max(np.abs(dq - dq_ref)) <= DQ_RTOL * scale, np.max(np.abs(dq - dq_ref)) / scale


def test_matches_torch_autograd_on_device(ica):
    b, K, D, T = 256, 8192, 128, 0.1
    q, k, mem = make(b, K, D, 3)
    tq = torch.from_numpy(q).cuda().requires_grad_(True)
    tk, tm = torch.from_numpy(k).cuda(), torch.from_numpy(mem).cuda()
    logits = torch.cat([(tq * tk).sum(1, keepdim=True), tq @ tm.t()], 1) / T
    ref = F.cross_entropy(logits, torch.zeros(b, dtype=torch.long, device="cuda"))
    ref.backward()
    loss, dq = run(ica, q, k, mem, T)
    assert abs(loss - ref.item()) <= LOSS_RTOL * abs(ref.item())
    g = tq.grad.cpu().numpy()
    assert np.max(np.abs(dq - g)) <= DQ_RTOL * np.max(np.abs(g))


def test_upstream_gradient_scales(ica):
    q, k, mem = make(64, 2048, 128, 9)
    _, dq1 = run(ica, q, k, mem, 0.1)
    tq = torch.from_numpy(q).cuda().requires_grad_(True)
    loss = ica.moco_queue_loss(tq, torch.from_numpy(k).cuda(), torch.from_numpy(mem).cuda(), 0.1)
    (loss * 3.0).backward()
    assert np.allclose(tq.grad.cpu().numpy(), 3.0 * dq1, rtol=1e-6, atol=0)


@pytest.mark.parametrize("ptr0,pushes", [(0, 1), (3968, 1), (4000, 1), (0, 2)])
def test_push_before_backward_uses_pre_push_queue(ica, ptr0, pushes):
    """moco.py:152-172: loss, then mem.push, then backward.  With the LIVE queue (no clone) the
    gradient must equal the one computed against a clone taken before the push -- including a push
    that truncates at the end of the ring (ptr0 = 4000, b = 128 > 96 free rows) and a second push
    before backward (symmetric variants push twice)."""
    b, K, D, T = 128, 4096, 128, 0.1
    q, k, mem = make(b, K, D, 17 + ptr0)
    bank = ica.MemoryBank()
    bank.register("negative", dim=D, k=K)
    bank["negative"] = torch.from_numpy(mem)
    bank = bank.cuda()
    bank.offsets["negative"] = ptr0
    tk = torch.from_numpy(k).cuda()

    tq_ref = torch.from_numpy(q).cuda().requires_grad_(True)
    ica.contrastive_loss2(query=tq_ref, key=tk, memory=bank["negative"].clone().detach(), temperature=T).backward()

    tq = torch.from_numpy(q).cuda().requires_grad_(True)
    loss = ica.contrastive_loss2(query=tq, key=tk, memory=bank["negative"].detach(), temperature=T)
    orc = ko.MemoryBankOracle()
    orc.register("negative", dim=D, k=K)
    orc["negative"] = mem.copy()
    orc.offsets["negative"] = ptr0
    for i in range(pushes):
        rows = tk if i == 0 else torch.from_numpy(q).cuda()
        bank.push("negative", rows)
        orc.push("negative", rows.cpu().numpy())
    assert bank.offsets["negative"] == orc.offsets["negative"]
    assert np.array_equal(bank["negative"].cpu().numpy(), orc["negative"])          # the push itself is unchanged
    loss.backward()
    assert torch.equal(tq.grad, tq_ref.grad)                                        # same kernel, same inputs: bitwise


def test_symmetric_branch_carries_the_key_gradient(ica):
    """moco.py:138-149 (params.symmetric): Lcsa = loss(query=feat_query, key=feat_key) and
    Lcsb = loss(query=feat_key, key=feat_query) -- feat_key comes out of a no_grad block, so Lcsb's whole
    gradient flows through ``key``.  d(Lcsa + Lcsb)/d feat_query against the float64 oracle and torch autograd."""
    b, K, D, T = 96, 3000, 128, 0.1
    q, k, mem = make(b, K, D, 41)
    _, dq_a, _, _ = ko.moco_loss(q, k, mem, T)
    loss_b, _, _, _, dkey_b = ko.moco_loss(k, q, mem, T, with_dkey=True)
    tq = torch.from_numpy(q).cuda().requires_grad_(True)
    tk, tm = torch.from_numpy(k).cuda(), torch.from_numpy(mem).cuda()
    la = ica.contrastive_loss2(query=tq, key=tk, memory=tm, query_neg=False, key_neg=False, temperature=T, norm=False)
    lb = ica.contrastive_loss2(query=tk, key=tq, memory=tm, query_neg=False, key_neg=False, temperature=T, norm=False)
    assert abs(lb.item() - loss_b) <= LOSS_RTOL * abs(loss_b)
    (la + lb).backward()
    ref = dq_a + dkey_b
    assert np.max(np.abs(tq.grad.cpu().numpy() - ref)) <= DQ_RTOL * np.max(np.abs(ref))
    # the same sum from torch autograd on the device
    tq2 = torch.from_numpy(q).cuda().requires_grad_(True)

    def torch_loss(qq, kk):
        logits = torch.cat([(qq * kk).sum(1, keepdim=True), qq @ tm.t()], 1) / T
        return F.cross_entropy(logits, torch.zeros(b, dtype=torch.long, device="cuda"))
    (torch_loss(tq2, tk) + torch_loss(tk, tq2)).backward()
    g = tq2.grad.cpu().numpy()
    assert np.max(np.abs(tq.grad.cpu().numpy() - g)) <= DQ_RTOL * np.max(np.abs(g))
    # key-only gradient needs no pass over the queue and therefore not the query backward's D limits
    q3, k3, mem3 = make(20, 500, 96, 43)
    t3 = torch.from_numpy(q3).cuda().requires_grad_(True)
    ica.contrastive_loss2(query=torch.from_numpy(k3).cuda(), key=t3, memory=torch.from_numpy(mem3).cuda(), temperature=T).backward()
    dkey3 = ko.moco_loss(k3, q3, mem3, T, with_dkey=True)[4]
    assert np.max(np.abs(t3.grad.cpu().numpy() - dkey3)) <= DQ_RTOL * np.max(np.abs(dkey3))


def test_gradients_that_are_not_built_fail_at_the_forward(ica):
    """A queue that requires grad, and a query gradient at a feature_dim the fused backward is not built for, raise
    when the loss is CALLED -- not from inside autograd in the middle of a train step, and never silently."""
    q, k, mem = make(8, 64, 96, 2)
    tq, tk, tm = torch.from_numpy(q).cuda(), torch.from_numpy(k).cuda(), torch.from_numpy(mem).cuda()
    with pytest.raises(NotImplementedError, match="negative queue"):
        ica.contrastive_loss2(query=tq, key=tk, memory=tm.clone().requires_grad_(True))
    with pytest.raises(NotImplementedError, match="D % 64"):
        ica.contrastive_loss2(query=tq.clone().requires_grad_(True), key=tk, memory=tm)
    loss = ica.contrastive_loss2(query=tq, key=tk, memory=tm)                    # no gradient asked: D = 96 is fine
    assert abs(loss.item() - ko.moco_loss(q, k, mem, 0.1)[0]) <= LOSS_RTOL * abs(loss.item())


def test_deterministic(ica):
    q, k, mem = make(200, 30000, 128, 5)
    a = run(ica, q, k, mem, 0.1)
    b_ = run(ica, q, k, mem, 0.1)
    assert a[0] == b_[0] and np.array_equal(a[1], b_[1])


def test_unsupported_forms_and_cpu_tensors_fail_loudly(ica):
    q, k, mem = make(8, 64, 64, 1)
    tq, tk, tm = torch.from_numpy(q), torch.from_numpy(k), torch.from_numpy(mem)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ica.contrastive_loss2(query=tq, key=tk, memory=tm)
    with pytest.raises(NotImplementedError):
        ica.contrastive_loss2(query=tq.cuda(), key=tk.cuda())                     # SimCLR form: outside the path
    with pytest.raises(NotImplementedError):
        ica.contrastive_loss2(query=tq.cuda(), key=tk.cuda(), memory=tm.cuda(), query_neg=True)
    with pytest.raises(ValueError):
        ica.contrastive_loss2(query=tq.cuda(), key=tk.cuda(), memory=tm[:, :32].contiguous().cuda())
