"""Worker for the multi-process sharded-kNN tests (spawned by test_sharded_*.py)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import knn_oracle as ko          # noqa: E402
from oracle import knn_torch_cpu as kt       # noqa: E402


class OracleOps:
    """CPU stand-ins for the HIP primitives (tests only): lets gloo exercise the orchestration."""

    def build_bank(self, rows, idx_base):
        return rows.float(), int(idx_base)

    def bank_rows(self, bank):
        return bank[0].shape[0]

    def topk(self, feature, bank, k):
        v, i = kt.topk_rows(feature, bank[0], k)
        return v, i + bank[1]

    def merge(self, gv, gi, k):
        v, i = ko.merge_topk(gv.numpy(), gi.numpy(), k)
        return torch.from_numpy(v), torch.from_numpy(i)

    def vote(self, val, idx, labels, classes, knn_t):
        w = ko.vote_weights(val.numpy(), knn_t)
        s = ko.vote_scores(w, labels.numpy()[idx.numpy()], classes)
        return torch.from_numpy(ko.rank_classes(s))


def make_case(seed=0, B=50, N=1203, D=32, C=7):
    rs = np.random.RandomState(seed)
    f = ko.l2_normalize(rs.randn(B, D))
    x = ko.l2_normalize(rs.randn(N, D))
    x[700:760] = x[10:70]                      # exact ties across the shard boundary
    lab = rs.randint(0, C, size=N)
    return f, x, lab, C


def main(rank, world, port, use_gpu, out_dir):
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=True)      # a hang names its line instead of timing out silently
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import image_classification_amd as ica
        from image_classification_amd import sharded  # noqa: F401  (alias import path)
        sh = sys.modules["image-classification_amd.sharded"]
        f, x, lab, C = make_case()
        k, T = 40, 0.1
        lo, hi = sh.shard_bounds(x.shape[0], world, rank)
        dev = torch.device("cuda:0") if use_gpu else torch.device("cpu")
        ops = None if use_gpu else OracleOps()
        knn = sh.ShardedKnn(torch.from_numpy(x[lo:hi]).to(dev), lo, ops=ops, compute_dtype="fp32", query_chunk=16)
        assert knn.n_total == x.shape[0] and knn.shard_sizes[rank] == hi - lo
        # a caller that knows the partition passes the shard sizes: the constructor then runs no collective
        sizes = [b - a for a, b in (sh.shard_bounds(x.shape[0], world, r) for r in range(world))]
        knn2 = sh.ShardedKnn(torch.from_numpy(x[lo:hi]).to(dev), lo, ops=ops, compute_dtype="fp32", query_chunk=16, shard_sizes=sizes)
        assert knn2.shard_sizes == knn.shard_sizes and knn2.n_total == knn.n_total
        try:
            sh.ShardedKnn(torch.from_numpy(x[lo:hi]).to(dev), lo, ops=ops, compute_dtype="fp32", shard_sizes=[1] * world)
            raise AssertionError("wrong shard_sizes accepted")
        except ValueError:
            pass
        val, idx = knn.topk(torch.from_numpy(f).to(dev), k)
        pred = knn.predict(torch.from_numpy(f).to(dev), torch.from_numpy(lab).to(dev), C, k, T)
        top1 = knn.predict(torch.from_numpy(f).to(dev), torch.from_numpy(lab).to(dev), C, k, T, gather="top1")
        assert torch.equal(knn.local_ranking, pred[knn.local_rows])     # the owner's slices are the full result's rows
        # comparison row of SURVEY 8(e): whole bank on every rank, queries sliced, no candidate exchange
        qs = sh.QuerySlicedKnn(torch.from_numpy(x).to(dev), ops=ops, compute_dtype="fp32", query_chunk=16)
        qpred = qs.predict(torch.from_numpy(f).to(dev), torch.from_numpy(lab).to(dev), C, k, T)
        qtop1 = qs.predict(torch.from_numpy(f).to(dev), torch.from_numpy(lab).to(dev), C, k, T, gather="top1")
        assert torch.equal(qpred, pred) and torch.equal(qtop1, top1)
        # uneven chunks: 50 queries in chunks of 7 leave a last chunk of 8 (the 1-row tail is folded in), and with
        # query_chunk=1 every chunk is ONE query, so all ranks but one own a zero-row slice of it -- the split sizes
        # of the all_to_all_single and the interleave undo of the final gather see 0-row pieces
        small = {"local_rows": knn.local_rows.cpu().numpy()}
        for qc, nq in ((7, 50), (1, 5)):
            knn.query_chunk = qc
            fq = torch.from_numpy(f[:nq]).to(dev)
            small[f"pred_qc{qc}"] = knn.predict(fq, torch.from_numpy(lab).to(dev), C, k, T).cpu().numpy()
            small[f"top1_qc{qc}"] = knn.predict(fq, torch.from_numpy(lab).to(dev), C, k, T, gather="top1").cpu().numpy()
            small[f"rows_qc{qc}"] = knn.local_rows.cpu().numpy()
            v2, i2 = knn.topk(fq, k)
            # (CPU stand-in ops: MKL's summation order moves with the chunk shape, so values get a tolerance there)
            assert torch.equal(i2, idx[:nq]) and (torch.equal(v2, val[:nq]) if use_gpu else torch.allclose(v2, val[:nq], atol=1e-6))
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), val=val.cpu().numpy(), idx=idx.cpu().numpy(),
                 pred=pred.cpu().numpy(), top1=top1.cpu().numpy(), **small)
        # MemoryBank.push all-gathers the keys of every rank first (memory_bank.py:64)
        if use_gpu:
            mb = ica.MemoryBank()
            mb.register("negative", dim=4, k=10)
            mb = mb.to(dev)
            mb.push("negative", torch.full((3, 4), float(rank + 1), device=dev))
            got = mb["negative"][:3 * world, 0].cpu().numpy()
            assert got.tolist() == [float(r + 1) for r in range(world) for _ in range(3)], got
            assert mb.offsets["negative"] == 3 * world
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == "gpu", sys.argv[5])
