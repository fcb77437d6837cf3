"""GPU tests of the MemoryBank / StorageBank mirrors (models/memory_bank.py) through the C ABI."""
import numpy as np
import pytest
import torch

from oracle import knn_oracle as ko
from tests import parity as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ica():
    import image_classification_amd as pkg
    pkg._lib.lib()
    return pkg


@pytest.mark.parametrize("tag", ["k10_b4", "k4096_b512", "k65535_b512", "k65536_b256", "k7_b9", "k12_1d"])
def test_push_trace_matches_reference(ica, tag):
    """Replay the reference's recorded push traces (tests/golden/memory_bank_push.npz)."""
    z = np.load(P.GOLDEN + "/memory_bank_push.npz")
    K, b, dim, n_push = [int(v) for v in z[f"{tag}_params"]]
    mb = ica.MemoryBank()
    mb.register("q", dim=dim, k=K)
    mb["q"] = torch.full_like(mb["q"], -1.0)
    mb = mb.cuda()
    offs = [mb.offsets["q"]]
    for p in range(n_push):
        shape = (b,) if dim <= 0 else (b, dim)
        val = (torch.arange(b, dtype=torch.float32) + 1000.0 * (p + 1)).reshape(-1, *([1] * (len(shape) - 1))).expand(shape)
        mb.push("q", val.cuda())
        offs.append(mb.offsets["q"])
    final = mb["q"].cpu().numpy()
    final = final if final.ndim == 1 else final[:, 0]
    assert offs == z[f"{tag}_offsets"].tolist()
    assert np.array_equal(final, z[f"{tag}_final"])


def test_push_rows_are_bitwise_copies(ica):
    rs = np.random.RandomState(0)
    mb = ica.MemoryBank()
    mb.register("negative", dim=128, k=1000)
    mb = mb.cuda()
    orc = ko.MemoryBankOracle()
    orc.register("negative", dim=128, k=1000)
    orc["negative"] = mb["negative"].cpu().numpy().copy()
    for _ in range(9):
        v = rs.randn(256, 128).astype(np.float32)
        mb.push("negative", torch.from_numpy(v).cuda())
        orc.push("negative", v)
        assert mb.offsets["negative"] == orc.offsets["negative"]
    assert np.array_equal(mb["negative"].cpu().numpy(), orc["negative"])
    assert "negative" in mb.state_dict()                       # registered buffer, rides in state_dict
    with pytest.raises(AssertionError):
        mb.push("missing", torch.zeros(1, 128).cuda())
    with pytest.raises(AssertionError):
        mb.push("negative", torch.zeros(128).cuda())


def test_push_other_dtypes_and_unaligned(ica):
    mb = ica.MemoryBank()
    mb.register("ids", dim=-1, k=11, dtype=torch.float64)
    mb.register("odd", dim=3, k=7)
    mb = mb.cuda()
    a = mb["ids"].cpu().numpy().copy()
    mb.push("ids", torch.arange(5, dtype=torch.float64).cuda())
    mb.push("ids", torch.arange(5, 10, dtype=torch.float64).cuda())
    mb.push("ids", torch.arange(10, 15, dtype=torch.float64).cuda())       # only 1 row fits
    a[:10] = np.arange(10)
    a[10] = 10
    assert np.array_equal(mb["ids"].cpu().numpy(), a) and mb.offsets["ids"] == 0
    v = torch.arange(15, dtype=torch.float32).reshape(5, 3).cuda()
    mb.push("odd", v)
    assert torch.equal(mb["odd"][:5], v) and mb.offsets["odd"] == 5


def test_storage_bank_matches_reference(ica):
    z = np.load(P.GOLDEN + "/storage_bank.npz")
    sb = ica.StorageBank()
    sb.register("test_feature", dim=4, k=10)
    sb.register("test_ys", dim=-1, k=10, dtype=torch.long)
    assert np.array_equal(sb["test_feature"].numpy(), z["init_feature"])
    assert np.array_equal(sb["test_ys"].numpy(), z["init_ys"])
    sb = sb.cuda()
    c = lambda a: torch.from_numpy(a).cuda()
    sb.scatter("test_feature", c(z["vals"][:6]), c(z["perm"][:6]))
    sb.scatter("test_ys", c(z["ys"][:6]), c(z["perm"][:6]))
    assert np.array_equal(sb["test_feature"].cpu().numpy(), z["mid_feature"])
    assert np.array_equal(sb["test_ys"].cpu().numpy(), z["mid_ys"])
    sb.scatter("test_feature", c(z["vals"][6:]), c(z["perm"][6:]))
    sb.scatter("test_ys", c(z["ys"][6:]), c(z["perm"][6:]))
    assert np.array_equal(sb["test_feature"].cpu().numpy(), z["final_feature"])
    assert np.array_equal(sb["test_ys"].cpu().numpy(), z["final_ys"])
    assert sb.sizes["test_feature"] == int(z["sizes"][0])


def test_monitor_flow_like_ssltrainer(ica):
    """ssltrainer.py:133-190 end to end: scatter test features by id, build the bank, predict, top-1 accuracy."""
    rs = np.random.RandomState(5)
    C, D, N, B = 10, 64, 3000, 500
    cent = rs.randn(C, D)
    lab = rs.randint(0, C, size=N)
    x = cent[lab] + 0.7 * rs.randn(N, D)
    ys = rs.randint(0, C, size=B)
    f = ko.l2_normalize(cent[ys] + 0.7 * rs.randn(B, D))
    sb = ica.StorageBank()
    sb.register("test_feature", dim=D, k=B)
    sb.register("test_ys", dim=-1, k=B, dtype=torch.long)
    sb = sb.cuda()
    perm = rs.permutation(B)
    for s in range(0, B, 128):
        ids = perm[s:s + 128]
        sb.scatter("test_feature", torch.from_numpy(f[ids]).cuda(), torch.from_numpy(ids).cuda())
        sb.scatter("test_ys", torch.from_numpy(ys[ids]).cuda(), torch.from_numpy(ids).cuda())
    bank = torch.nn.functional.normalize(torch.from_numpy(x.astype(np.float32)).cuda(), dim=1).t().contiguous()
    pred = ica.knn_predict(sb["test_feature"], bank, torch.from_numpy(lab).cuda(), C, 200, 0.1)
    acc = torch.eq(pred[:, 0], sb["test_ys"]).float().sum().item() / B * 100
    acc_ref = ko.knn_monitor_accuracy(f, ys, x.astype(np.float32), lab, C, 200, 0.1)
    assert abs(acc - acc_ref) <= 100.0 / B + 1e-9, (acc, acc_ref)


@pytest.mark.parametrize("dtype,D", [("fp32", 128), ("bf16", 128), ("fp32", 50), ("bf16", 2048)])
def test_bank_builder_normalises_and_packs(ica, dtype, D):
    """Next row 8(f)1: ssltrainer.py:172-179 (F.normalize per memo batch, cat, t().contiguous()) in one kernel per
    batch.  Floating point: each stored element within 2 float32 ulp of the float64 oracle normalisation (the
    norm's summation order differs from torch's); bf16 storage = that value rounded to nearest-even, so at most
    one bf16 ulp apart and equal almost everywhere."""
    rs = np.random.RandomState(D)
    N = 1000
    x = (rs.randn(N, D) * rs.uniform(0.1, 30.0, size=(N, 1))).astype(np.float32)
    x[17] = 0.0                                             # zero row: 0 / max(0, eps) = 0, as F.normalize
    b = ica.BankBuilder(N, D, "cuda", compute_dtype=dtype)
    for s in range(0, N, 192):                              # ragged last batch
        b.append(torch.from_numpy(x[s:s + 192]).cuda())
    bank = b.finish()
    got = bank.packed.data.float().cpu().numpy()
    ref = ko.l2_normalize(x)
    assert got.shape[1] == int(ica._lib.lib().mi_knn_ld(D, 0 if dtype == "fp32" else 1))
    assert np.all(got[:, D:] == 0) and np.all(got[17] == 0)
    if dtype == "fp32":
        assert np.max(np.abs(got[:, :D] - ref) / np.maximum(np.abs(ref), 1e-30)) <= 3e-7
    else:
        refb = ko.bf16_round(ref)
        assert np.max(np.abs(got[:, :D] - refb)) <= np.max(np.abs(refb)) * 2.0 ** -7
        assert np.mean(got[:, :D] == refb) > 0.995
    with pytest.raises(ValueError):
        b.append(torch.zeros(1, D).cuda())                  # bank is full
    with pytest.raises(ValueError):
        ica.BankBuilder(10, D, "cuda", compute_dtype=dtype).append(torch.zeros(3, D + 1).cuda())


def test_bank_builder_flow_equals_reference_flow(ica):
    """knn_test with the builder == knn_test with F.normalize + cat + t().contiguous() + knn_predict (top-1 labels)."""
    rs = np.random.RandomState(9)
    C, D, N, B = 10, 128, 4000, 600
    cent = rs.randn(C, D)
    lab = rs.randint(0, C, size=N)
    x = (cent[lab] + 0.7 * rs.randn(N, D)).astype(np.float32)
    f = torch.from_numpy(ko.l2_normalize(cent[rs.randint(0, C, size=B)] + 0.7 * rs.randn(B, D))).cuda()
    tl = torch.from_numpy(lab).cuda()
    xb = torch.from_numpy(x).cuda()
    ref_bank = torch.cat([torch.nn.functional.normalize(xb[s:s + 512], dim=1) for s in range(0, N, 512)], 0).t().contiguous()
    ref = ica.knn_predict(f, ref_bank, tl, C, 200, 0.1)
    b = ica.BankBuilder(N, D, "cuda", compute_dtype="fp32")
    for s in range(0, N, 512):
        b.append(xb[s:s + 512])
    got = ica.knn_predict(f, b.finish(), tl, C, 200, 0.1)
    assert (got[:, 0] == ref[:, 0]).float().mean().item() >= 0.995       # 1-ulp bank differences can flip a near-tie
