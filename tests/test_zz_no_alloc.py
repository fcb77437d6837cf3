"""The allocation-free entry point ``mi_knn_topk_static`` (include/mi_knn.h) against ``mi_knn_topk``.

It has no counterpart in the reference; the bar is the one of the entry point it mirrors: bit-equal values and
indices (same kernels, only the workspace differs: the library's static per-GPU scratch instead of a caller
buffer).  Two calls in a row share that scratch, the second with a smaller input: whatever the first call left
behind must not leak into the second."""
import numpy as np
import pytest
import torch

from oracle import chain
from oracle import knn_oracle as ko

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ica():
    import image_classification_amd as pkg
    pkg._lib.lib()
    return pkg


def _case(B, N, D, seed):
    rs = np.random.RandomState(seed)
    centers = rs.randn(6, D)
    f = ko.l2_normalize(centers[rs.randint(0, 6, size=B)] + 0.3 * rs.randn(B, D))
    x = ko.l2_normalize(centers[rs.randint(0, 6, size=N)] + 0.3 * rs.randn(N, D))
    return f, x


def _static(ica, f, bank, k):
    q = ica.pack_rows(f, False, {0: "fp32", 1: "bf16"}[bank.packed.dtype_code])
    val = torch.full((f.shape[0], k), float("nan"), dtype=torch.float32, device="cuda")
    idx = torch.full((f.shape[0], k), -7, dtype=torch.int64, device="cuda")
    before = torch.cuda.memory_allocated()
    ica.knn_topk_static(q, bank, k, val, idx)
    assert torch.cuda.memory_allocated() == before, "the allocation-free entry point went through torch's allocator"
    torch.cuda.synchronize()
    return val, idx


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_static_scratch_topk_equals_workspace_topk_two_calls_in_a_row(ica, dtype):
    k = 200
    f1, x1 = _case(1500, 40000, 96, 1)              # several filter stages, > 64 candidates per row in the early ones
    f2, x2 = _case(70, 3000, 64, 2)                 # second call: smaller in every dimension, one filter stage
    for f, x in ((f1, x1), (f2, x2)):
        ft, bank = torch.from_numpy(f).cuda(), ica.BankIndex.from_rows(torch.from_numpy(x).cuda(), dtype, idx_base=11)
        v_ref, i_ref = ica.knn_topk(ft, bank, k)
        v, i = _static(ica, ft, bank, k)
        assert torch.equal(v, v_ref) and torch.equal(i, i_ref), f"{dtype}: static-scratch result differs ({f.shape[0]} queries)"
        if dtype == "fp32":                          # and the mirrored entry point's own bar: bit-exact vs the chain oracle
            vo, io = chain.topk_rows(chain.sim_chain_f32(f, x), k)
            assert np.array_equal(v.cpu().numpy(), vo) and np.array_equal(i.cpu().numpy() - 11, io)


def test_static_scratch_rejects_shapes_beyond_its_limit(ica):
    f, x = _case(8, 3000, 64, 3)
    bank = ica.BankIndex.from_rows(torch.from_numpy(x).cuda(), "fp32")
    big = torch.zeros((3585, 64), dtype=torch.float32, device="cuda")
    q = ica.pack_rows(big, False, "fp32")
    val = torch.empty((3585, 200), dtype=torch.float32, device="cuda")
    idx = torch.empty((3585, 200), dtype=torch.int64, device="cuda")
    with pytest.raises(ica._lib.MiKnnError, match="static scratch"):
        ica.knn_topk_static(q, bank, 200, val, idx)
