#!/usr/bin/env python3
"""bench.py -- kNN-monitor throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full kNN monitor pass over one batch of synthetic input, i.e. everything the
reference does in ``knn_predict`` (models/knn.py:4-21) for ALL queries: pack the (this rank's shard
of the) float32 feature bank and the float32 queries into the compute layout, fused similarity +
top-k, (N > 1: one RCCL exchange of the (score, index) candidates per query chunk -- rank j receives
its slice of the chunk's queries from every rank -- + deterministic merge), exp(s/T)-weighted class
vote and full class ranking; the top-1 labels of all queries end on every rank (the only column the
reference's caller consumes, ssltrainer.py:187), the full [B, classes] ranking stays distributed over
the ranks by query slice.  Inputs are resident in HBM as float32 when
the timed region starts.  Workload = BASELINE.json configs[3] (the configuration the metric is
quoted on): 50 000 queries x 1 280 000 x 2048 bank, k=200, T=0.1, 1000 classes, bf16 MFMA with
float32 accumulation.  N GPUs row-shard the SAME bank (strong scaling); value = queries/s of the
whole job.

Prints ONE JSON line on rank 0 (see the contract in the task statement) with two extra objects:
``roofline`` (dominant kernel = the fused similarity/filter kernel, timed per launch with HIP
events on its own stream inside the timed region by the library's mi_profile_* hooks) and
``cpu_baseline`` (torch-CPU port of the reference path, bounded sample, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}       # dense MFMA peaks, MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)      # the last warm-up step has no allocation stalls: the timed region starts on a warm clock
    ap.add_argument("--queries", type=int, default=50_000)
    ap.add_argument("--bank", type=int, default=1_280_000)
    ap.add_argument("--dim", type=int, default=2048)
    ap.add_argument("--classes", type=int, default=1000)
    ap.add_argument("--k", type=int, default=200)
    ap.add_argument("--temperature", type=float, default=0.1)
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--query-chunk", type=int, default=0, help="query rows per fused top-k call; 0 = by shard size (sharded.auto_query_chunk)")
    ap.add_argument("--emulate-rank-of", type=int, default=0, metavar="G",
                    help="one GPU only: do what ONE rank of a G-rank row-sharded job does -- all queries against a bank of "
                         "--bank / G rows, then merge (G candidate blocks) + vote for 1 / G of the queries; no exchange "
                         "(tools/shard_cost.sh: upper bound of the scaling efficiency)")
    ap.add_argument("--no-clock-probe", action="store_true", help="skip the in-kernel clock measurement after the timed region")
    ap.add_argument("--shard", choices=["rows", "queries"], default="rows",
                    help="rows: the bank is row-sharded (north star, default).  queries: comparison row of SURVEY 8(e) -- "
                         "every rank holds the whole bank and serves a slice of the queries, no candidate exchange")
    ap.add_argument("--data", choices=["random", "clustered", "clustered-sorted"], default="random",
                    help="random: N(0,1) rows, L2-normalised (the metric's workload).  clustered: class centroid + 0.5 N(0,1) "
                         "(SURVEY 8d realism variant); clustered-sorted: the same with the bank in class order (an ImageFolder memo loader)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal on a one-GPU box: run the multi-rank code path (RCCL process group, candidate all-to-all, "
                         "merge, label all-gather, barriers, max-over-ranks) with a process group of ONE rank")
    ap.add_argument("--rehearse-shared-gpu", action="store_true",
                    help="REHEARSAL ONLY (not a measurement): with --gpus N > 1 every rank uses cuda:0 and the process group is gloo "
                         "(RCCL refuses two ranks on one device) -- runs the N-rank launcher, step loop, exchange and timing code on a one-GPU box")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    return ap.parse_args()


def unit_rows(n, d, seed, device, chunk=65536):
    """[n, d] float32 rows, N(0,1) then L2-normalised (ssltrainer.py:142,176), generated on device."""
    g = torch.Generator(device=device).manual_seed(seed)
    out = torch.empty((n, d), dtype=torch.float32, device=device)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        x = torch.randn((e - s, d), generator=g, device=device, dtype=torch.float32)
        out[s:e] = torch.nn.functional.normalize(x, dim=1)
    return out


def clustered_rows(labels, d, centroids, seed, device, chunk=65536):
    """[n, d] float32 rows: normalise(centroid[label] + 0.5 N(0,1)), generated on device."""
    g = torch.Generator(device=device).manual_seed(seed)
    n = labels.numel()
    out = torch.empty((n, d), dtype=torch.float32, device=device)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        x = centroids[labels[s:e]] + 0.5 * torch.randn((e - s, d), generator=g, device=device, dtype=torch.float32)
        out[s:e] = torch.nn.functional.normalize(x, dim=1)
    return out


def cpu_baseline(feature, bank_rows, labels, args, seconds):
    """Torch-CPU port of the reference path on a bounded sample of the SAME workload."""
    from oracle import knn_torch_cpu as kt
    threads = torch.get_num_threads()
    chunk = 256
    f_cpu = feature[: chunk * 16].cpu()
    bank_t = bank_rows.t().contiguous().cpu()          # the reference's [D, N] layout (ssltrainer.py:179)
    lab_cpu = labels.cpu()
    kt.knn_predict_chunked(f_cpu[:8], bank_t, lab_cpu, args.classes, args.k, args.temperature)   # warm-up
    done, t0 = 0, time.perf_counter()
    while done < f_cpu.shape[0]:
        kt.knn_predict_chunked(f_cpu[done:done + chunk], bank_t, lab_cpu, args.classes, args.k, args.temperature)
        done += chunk
        if time.perf_counter() - t0 >= seconds:
            break
    dt = time.perf_counter() - t0
    return {
        "value": done / dt, "unit": "queries/s", "cores": threads, "kind": "port",
        "sample": f"{done} of {args.queries} queries ({done // chunk} chunks of {chunk}) x full "
                  f"{args.bank}x{args.dim} fp32 bank, {dt:.1f} s; cost is linear in the query count",
    }


def launcher_command(gpus, argv, port=None, python=None):
    """(command, environment) that starts ``gpus`` ranks of this script, one per GPU, on this node.

    ``python bench.py --gpus N`` outside a torch.distributed launcher (no WORLD_SIZE) is a request for an N-rank
    job: the parent starts the ranks as FRESH child processes through ``torch.distributed.run`` BEFORE it makes any
    GPU call of its own (a process that has initialised the GPU must never be replaced or forked into ranks), relays
    their output and exits with their status.  The reference has no counterpart (models/knn.py:4-21 is single-device)."""
    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    cmd = [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(gpus)}",
           "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "16")
    for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)                      # the launcher sets them for every rank
    return cmd, env


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no GPU call has been made in this process (importing torch makes none)
        cmd, env = launcher_command(args.gpus, sys.argv[1:])
        raise SystemExit(subprocess.call(cmd, env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; "
                         f"run `python bench.py --gpus N` (it starts its own ranks) or give torch.distributed.run "
                         f"--nproc-per-node equal to --gpus")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False")
    dev_index = 0 if args.rehearse_shared_gpu else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist_on = world > 1 or args.force_dist
    if args.force_dist and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    if world > 1:
        if args.rehearse_shared_gpu:
            dist.init_process_group("gloo")
        else:
            if torch.cuda.device_count() < world:
                raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible (one rank per GPU)")
            dist.init_process_group("nccl", device_id=device)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: process group of {dist.get_world_size()} ranks for --gpus {args.gpus}")

    import image_classification_amd as ica
    from image_classification_amd import _lib
    import importlib
    sharded = importlib.import_module("image-classification_amd.sharded")
    _lib.lib()

    B, N, D, C, k, T = args.queries, args.bank, args.dim, args.classes, args.k, args.temperature
    by_rows = args.shard == "rows"
    emu = args.emulate_rank_of
    if emu and (world > 1 or not by_rows or N % emu):
        raise SystemExit("--emulate-rank-of is a one-GPU, row-sharded emulation and needs bank rows divisible by G")
    lo, hi = sharded.shard_bounds(N, world, rank) if by_rows else (0, N)
    # every rank's shard size is known from the partition: no size exchange (and no host read) inside a step
    shard_sizes = [b - a for a, b in (sharded.shard_bounds(N, world, r) for r in range(world))]
    if emu:
        lo, hi = sharded.shard_bounds(N, emu, 0)                            # rank 0's shard of the G-rank job
    labels = torch.randint(0, C, (N,), generator=torch.Generator(device=device).manual_seed(2), device=device)
    if args.data == "random":
        bank_rows = unit_rows(hi - lo, D, 1000 + (rank if by_rows else 0), device)   # this rank's shard (or the whole bank), [n_local, D] f32
        feature = unit_rows(B, D, 1, device)                               # replicated queries
    else:
        if args.data == "clustered-sorted":
            labels = labels.sort().values
        centroids = torch.randn((C, D), generator=torch.Generator(device=device).manual_seed(3), device=device)
        bank_rows = clustered_rows(labels[lo:hi], D, centroids, 1000 + (rank if by_rows else 0), device)
        qlab = torch.randint(0, C, (B,), generator=torch.Generator(device=device).manual_seed(4), device=device)
        feature = clustered_rows(qlab, D, centroids, 1, device)
    torch.cuda.synchronize()

    def emulated_rank_step():
        """Rank 0 of an `emu`-rank job without the exchange: local candidates for ALL queries, then merge + vote for
        its 1 / emu slice of every chunk (the other ranks' candidate blocks are stood in for by its own, re-based so
        that the keys stay distinct)."""
        knn = sharded.ShardedKnn(bank_rows, lo, compute_dtype=args.dtype, query_chunk=args.query_chunk or None)
        outs = []
        for s0, e0 in sharded.chunk_bounds(B, knn.query_chunk):
            val, idx = knn._local_candidates(feature[s0:e0], k)
            a, b2 = sharded.shard_bounds(e0 - s0, emu, 0)
            cv = val[a:b2].unsqueeze(0).expand(emu, -1, -1).contiguous()
            ci = torch.stack([idx[a:b2] + g * (hi - lo) for g in range(emu)])
            mv, mi = ica.knn_merge(cv, ci, k)
            outs.append(ica.knn_vote(mv, mi, labels, C, T, check=False))
        ranking = torch.cat(outs, 0)
        return ranking[:, 0].contiguous(), ranking

    def step():
        if emu:
            return emulated_rank_step()
        if by_rows:
            knn = sharded.ShardedKnn(bank_rows, lo, compute_dtype=args.dtype, query_chunk=args.query_chunk or None,
                                     shard_sizes=shard_sizes, always_exchange=args.force_dist)
        else:
            knn = sharded.QuerySlicedKnn(bank_rows, compute_dtype=args.dtype, query_chunk=args.query_chunk or 16384)
        top1 = knn.predict(feature, labels, C, k, T, gather="top1")
        return top1, knn.local_ranking

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pred, ranking = step()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    sim_ms, sim_launches, sim_flops = _lib.profile_collect(0)
    sel_ms, sel_launches, _ = _lib.profile_collect(1)
    _lib.profile_enable(False)
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert ranking.shape[1] == C and (emu or tuple(pred.shape) == (B,))   # top-1 everywhere, full ranking with the slice owner

    # In-kernel clock of the dominant kernel (MI355X lowers its clock under MFMA load: MI355X_MICROARCH.md, DVFS
    # give-back item 6): two more untimed steps through the STAMPED instantiation of the same kernel, right after
    # the timed region; clock = d(s_memtime) / d(s_memrealtime) x 100 MHz, median over the workgroups.
    clock_ghz = None
    if not args.no_clock_probe:                      # every rank: the probe steps contain the job's collectives
        stamps = torch.zeros((16384, 2), dtype=torch.int64, device=device)
        _lib.check(_lib.lib().mi_profile_clock(stamps.data_ptr(), stamps.shape[0]), "mi_profile_clock")
        try:
            for _ in range(2):
                step()
            torch.cuda.synchronize()
        finally:
            _lib.check(_lib.lib().mi_profile_clock(None, 0), "mi_profile_clock")
        st = stamps.cpu().double()
        ok = st[:, 1] > 0
        if bool(ok.any()):
            clock_ghz = float((st[ok, 0] / st[ok, 1]).median()) * 0.1
    if dist_on:
        dist.barrier()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        achieved = sim_flops / (sim_ms * 1e-3) / 1e12 if sim_ms > 0 else 0.0
        peak = PEAK_TFLOPS[args.dtype]
        # traffic is NOT measured in this run: it is the recorded result of the rocprofv3 --pmc passes named in
        # traffic_source (PMC collection needs its own profiled runs); null when no record matches this workload
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and not emu:
            try:
                rec = json.load(open(tpath)).get(f"{args.dtype}:{B}x{N}x{D}:g{world}", {})
                traffic = rec.get("hbm_bytes_per_launch")
                traffic_source = rec.get("source") if traffic is not None else None
            except Exception:
                traffic = None
        out = {
            "metric": "kNN queries/s @1.28Mx2048 bank,k=200",
            "value": B * args.steps / elapsed,
            "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "bf16" if args.dtype == "bf16" else "f32",
            "data": "synthetic" if args.data == "random" else f"synthetic ({args.data})",
            "config": {
                "workload": f"BASELINE configs[3]: kNN monitor, {B} queries x {N}x{D} bank, k={k}, T={T}, "
                            f"{C} classes, {args.dtype} MFMA with fp32 accumulate; step = pack + fused sim/top-k"
                            f"{' + RCCL candidate all-to-all + merge' if dist_on and by_rows else ''} + vote + ranking",
                "queries": B, "bank_rows": N, "dim": D, "k": k, "classes": C,
                "ranks": dist.get_world_size() if dist_on else 1,        # size of the process group that did the work
                **({"rehearsal": "--force-dist: multi-rank code path on a one-rank RCCL group"} if args.force_dist else {}),
                **({"rehearsal": f"--rehearse-shared-gpu: {world} ranks on ONE GPU over gloo; not a measurement"}
                   if args.rehearse_shared_gpu and world > 1 else {}),
                **({"emulated_rank_of": emu, "rank_bank_rows": hi - lo} if emu else {}),
                "parallelism": (f"bank rows sharded over {world} GPU(s); merge/vote sharded by query slice; top-1 labels all-gathered"
                                if by_rows else
                                f"comparison row: whole bank on each of {world} GPU(s), queries sliced, no candidate exchange; top-1 labels all-gathered"),
            },
            "roofline": {
                "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_source,
                "sustained_clock_ghz": clock_ghz, "spec_clock_ghz": 2.4,
                "peak_at_sustained_clock": peak * clock_ghz / 2.4 if clock_ghz else None,
                "frac_of_sustained": achieved / (peak * clock_ghz / 2.4) if clock_ghz else None,
                "clock_source": ("in-kernel s_memtime / s_memrealtime stamps of a stamped instantiation of the same kernel, "
                                 "2 untimed steps right after the timed region, median over workgroups (mi_profile_clock)"
                                 if clock_ghz else None),
                "kernel": "mi::sim_kernel (fused similarity GEMM + top-k filter)",
                "launches": sim_launches, "avg_launch_ms": sim_ms / max(sim_launches, 1),
                "algorithmic_flops_per_launch": sim_flops / max(sim_launches, 1),
                "share_of_step_time": sim_ms / (elapsed * 1e3),
                "select_kernel_ms_per_step": sel_ms / args.steps,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(feature, bank_rows, labels, args, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
