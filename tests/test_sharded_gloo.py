"""world_size-2 (and 3) tests of the sharded kNN orchestration.

CPU leg: gloo + oracle-backed compute ops (the HIP kernels cannot run here) -- checks the shard
bounds, global indices, chunked exchange, merge and vote wiring against the single-process oracle.
The exchange goes through the SAME ``all_to_all_single`` / ``all_gather_into_tensor`` calls, with the same
split sizes, as the RCCL path (``ShardedKnn._all_to_all`` / ``_all_gather``); only the staging differs.
GPU leg (-m gpu): the same flow with the real HIP ops, two ranks sharing the box's one GPU."""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from oracle import knn_oracle as ko
from tests import sharded_worker as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, mode):
    port = _free_port()
    with tempfile.TemporaryDirectory() as d:
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "sharded_worker.py"),
                                   str(r), str(world), str(port), mode, d], cwd=ROOT) for r in range(world)]
        try:
            rcs = [p.wait(timeout=300) for p in procs]
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        assert rcs == [0] * world, rcs
        outs = [dict(np.load(os.path.join(d, f"rank{r}.npz"))) for r in range(world)]
    f, x, lab, C = W.make_case()
    k, T = 40, 0.1
    ref = ko.knn_predict_full(f, np.ascontiguousarray(x.T), lab, C, k, T)
    return outs, ref


def _check(outs, ref, exact_values):
    for o in outs:
        assert np.array_equal(o["idx"], ref["sim_indices"])
        if exact_values:
            assert np.array_equal(o["val"], ref["sim_weight"])
        else:
            assert np.allclose(o["val"], ref["sim_weight"], atol=1e-5)
        assert np.array_equal(o["pred"][:, 0], ref["pred_labels"][:, 0])
        assert np.array_equal(o["pred"], outs[0]["pred"])           # identical on every rank
        assert np.array_equal(o["top1"], o["pred"][:, 0])
    owned = np.sort(np.concatenate([o["local_rows"] for o in outs]))
    assert np.array_equal(owned, np.arange(outs[0]["pred"].shape[0]))   # every query has exactly one owner
    # uneven chunkings (7-row chunks with a folded tail; 1-row chunks = zero-row slices on all ranks but one):
    # the same rankings as the 16-row chunking above, on every rank
    for qc, nq in ((7, 50), (1, 5)):
        for o in outs:
            assert np.array_equal(o[f"pred_qc{qc}"], outs[0]["pred"][:nq]), f"query_chunk={qc}"
            assert np.array_equal(o[f"top1_qc{qc}"], outs[0]["pred"][:nq, 0])
        owned = np.sort(np.concatenate([o[f"rows_qc{qc}"] for o in outs]))
        assert np.array_equal(owned, np.arange(nq))
    assert any(o["rows_qc1"].size == 0 or o["rows_qc1"].size < 5 for o in outs)


def test_shard_bounds_cover_everything():
    import importlib
    sh = importlib.import_module("image-classification_amd.sharded")
    for n, g in [(10, 3), (1281167, 8), (7, 8), (1280000, 4)]:
        b = [sh.shard_bounds(n, g, r) for r in range(g)]
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(g - 1))
        assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_knn_gloo_cpu(world):
    outs, ref = _run(world, "cpu")
    _check(outs, ref, exact_values=False)


@pytest.mark.gpu
def test_sharded_knn_two_ranks_one_gpu():
    outs, ref = _run(2, "gpu")
    from oracle import chain
    f, x, lab, C = W.make_case()
    sim = chain.sim_chain_f32(f, x)
    ref = ko.knn_from_similarity(sim, lab, C, 40, 0.1)
    _check(outs, ref, exact_values=True)
