"""models/module_utils.py wrapping queues: oracle vs the reference's recorded traces (CPU),
HIP path vs both (GPU)."""
import copy

import numpy as np
import pytest
import torch

from oracle import knn_oracle as ko
from tests import parity as P

TAGS = ["k10_b4", "k4096_b512", "k65535_b512", "k7_b7", "k9_b5"]


def _golden():
    return np.load(P.GOLDEN + "/wrapping_queue.npz")


def _vals(b, dim, p):
    return np.repeat((np.arange(b, dtype=np.float32) + 1000.0 * (p + 1))[:, None], dim, axis=1)


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_matches_reference_trace(tag):
    z = _golden()
    K, b, dim, n_push = [int(v) for v in z[f"{tag}_params"]]
    q = ko.WrappingQueueOracle(K, dim)
    cursors = [q.cursor]
    for p in range(n_push):
        q.push(_vals(b, dim, p))
        cursors.append(q.cursor)
    assert cursors == z[f"{tag}_cursors"].tolist()
    assert np.array_equal(np.where(q.data[:, 0] >= 1000.0, q.data[:, 0], -1.0).astype(np.float32), z[f"{tag}_final"])


def test_oracle_long_queue_and_oversized_push():
    z = _golden()
    q = ko.WrappingQueueOracle(11, None, dtype=np.int64)
    cur = [q.cursor]
    for p in range(6):
        q.push(np.arange(4, dtype=np.int64) + 100 * (p + 1))
        cur.append(q.cursor)
    assert cur == z["long_k11_b4_cursors"].tolist() and np.array_equal(q.data, z["long_k11_b4_final"])
    assert z["oversized_push"][0] == "RuntimeError"
    with pytest.raises(RuntimeError):
        ko.WrappingQueueOracle(5, 2).push(np.zeros((12, 2), np.float32))
    # longer than the queue but within 2K - cursor: the reference's two slice assignments both fit
    q = ko.WrappingQueueOracle(10, 2)
    q.push(np.full((5, 2), 500.0, np.float32))
    q.push(np.repeat((np.arange(12, dtype=np.float32) + 1000.0)[:, None], 2, axis=1))
    assert np.array_equal(q.data, z["overlong_k10_c5_b12_final"]) and q.cursor == int(z["overlong_k10_c5_b12_cursor"][0])


def test_attributes_survive_like_the_reference():
    import image_classification_amd as ica
    z = _golden()
    c = ica.module_utils.MemoryBank(queue_size=6, feature_dim=2)
    c.cursor = 4                                   # the reference's trace pushed 4 rows first
    assert [c.clone().cursor, c.to(torch.float64).cursor, c.clone().queue_size] == z["attrs_survive"].tolist()
    d = copy.deepcopy(c)
    assert type(d) is type(c) and d.cursor == 4 and d.queue_size == 6 and torch.equal(d.tensor(), c.tensor())
    assert type(c.tensor()) is torch.Tensor and c.half().dtype == torch.float16
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        c.push(torch.ones(2, 2))                   # CPU queue: the product path fails loudly


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_hip_push_matches_reference_trace(tag):
    import image_classification_amd as ica
    z = _golden()
    K, b, dim, n_push = [int(v) for v in z[f"{tag}_params"]]
    q = ica.module_utils.MemoryBank(K, dim).to("cuda")
    assert type(q) is ica.module_utils.MemoryBank and q.queue_size == K
    cursors = [q.cursor]
    for p in range(n_push):
        q.push(torch.from_numpy(_vals(b, dim, p)).cuda())
        cursors.append(q.cursor)
    assert cursors == z[f"{tag}_cursors"].tolist()
    col0 = q.tensor()[:, 0].cpu().numpy()
    assert np.array_equal(np.where(col0 >= 1000.0, col0, -1.0).astype(np.float32), z[f"{tag}_final"])


@pytest.mark.gpu
def test_hip_long_queue_random_rows_and_errors():
    import image_classification_amd as ica
    z = _golden()
    lb = ica.module_utils.LongTensorMemoryBank(11).to("cuda")
    cur = [lb.cursor]
    for p in range(6):
        lb.push(torch.arange(4, dtype=torch.long) + 100 * (p + 1))      # CPU item: moved to the queue's device
        cur.append(lb.cursor)
    assert cur == z["long_k11_b4_cursors"].tolist()
    assert np.array_equal(lb.tensor().cpu().numpy(), z["long_k11_b4_final"])
    rs = np.random.RandomState(0)
    q = ica.module_utils.MemoryBank(1000, 128).to("cuda")
    orc = ko.WrappingQueueOracle(1000, 128)
    orc.data = q.tensor().cpu().numpy().copy()
    for n in (256, 256, 256, 256, 999, 1000, 1):
        v = rs.randn(n, 128).astype(np.float32)
        q.push(torch.from_numpy(v).cuda())
        orc.push(v)
        assert q.cursor == orc.cursor
    assert np.array_equal(q.tensor().cpu().numpy(), orc.data)           # bitwise row copies
    assert q.cursor == 24
    with pytest.raises(RuntimeError, match="shape mismatch"):
        q.push(torch.zeros(1977, 128).cuda())                           # left = 1977 - 976 = 1001 > K, as the reference
    # longer than the queue, within 2K - cursor: the reference's recorded result (rows written twice keep the later row)
    q = ica.module_utils.MemoryBank(10, 2).to("cuda")
    q.push(torch.full((5, 2), 500.0).cuda())
    q.push((torch.arange(12, dtype=torch.float32) + 1000.0)[:, None].expand(12, 2).contiguous().cuda())
    assert np.array_equal(q.tensor().cpu().numpy(), z["overlong_k10_c5_b12_final"]) and q.cursor == int(z["overlong_k10_c5_b12_cursor"][0])
    v = rs.randn(1900, 128).astype(np.float32)                          # and at size: 1900 rows into 1000 from cursor 24
    q = ica.module_utils.MemoryBank(1000, 128).to("cuda")
    orc = ko.WrappingQueueOracle(1000, 128)
    orc.data = q.tensor().cpu().numpy().copy()
    for item in (v[:24], v):
        q.push(torch.from_numpy(item).cuda())
        orc.push(item)
    assert q.cursor == orc.cursor == (24 + 1900) % 1000 and np.array_equal(q.tensor().cpu().numpy(), orc.data)
    with pytest.raises(AssertionError):
        q.push(torch.zeros(4, 64).cuda())
