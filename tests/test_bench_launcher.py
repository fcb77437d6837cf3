"""bench.py --gpus N must run N ranks: the launcher's command / environment, the world-size check, and one real
start of two ranks (which stop at once here: no GPU is visible to them).  CPU only."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("n", [2, 8])
def test_launcher_command_and_env(n, monkeypatch):
    monkeypatch.setenv("RANK", "3")                      # stale launcher variables must not leak into the ranks
    monkeypatch.setenv("WORLD_SIZE", "5")
    b = _bench()
    argv = ["--gpus", str(n), "--steps", "20", "--warmup", "5"]
    cmd, env = b.launcher_command(n, argv, port=29731, python="python3")
    assert cmd[:3] == ["python3", "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and f"--nproc-per-node={n}" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29731"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv                           # the ranks get the caller's flags unchanged
    assert "RANK" not in env and "WORLD_SIZE" not in env and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    cmd2, _ = b.launcher_command(n, argv)                # a free port is picked when none is given
    assert 1024 < int(cmd2[cmd2.index("--master-port") + 1]) < 65536


def test_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 4" in r.stderr and "WORLD_SIZE=2" in r.stderr


def test_gpus_2_starts_two_ranks_and_relays_their_status():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself.  With no GPU visible each rank
    stops with bench.py's own message, and the parent exits non-zero: both facts prove the ranks were started."""
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs an MI355X") >= 2, r.stderr[-2000:]


@pytest.mark.gpu
def test_bench_multi_rank_code_path_on_a_one_rank_rccl_group():
    """`bench.py --force-dist`: the code a rank of an N-GPU job runs -- RCCL process group, candidate all-to-all, merge,
    top-1 all-gather, barriers, max-over-ranks -- on a process group of ONE rank (two RCCL ranks cannot share the box's
    one GPU), on a small workload; the JSON line must come out complete."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--queries", "3000", "--bank", "60000",
                        "--dim", "256", "--classes", "100", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-clock-probe"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["config"]["ranks"] == 1 and "rehearsal" in line["config"] and line["n_gpus"] == 1
    assert line["value"] > 0 and line["roofline"]["launches"] > 0 and "all-to-all" in line["config"]["workload"]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 3])
def test_bench_n_ranks_sharing_the_one_gpu(n):
    """`python bench.py --gpus N --rehearse-shared-gpu`: the PLAIN command starts its N ranks itself; every rank runs the
    multi-rank step loop (row shards, candidate exchange, merge + vote of its query slice, top-1 all-gather, barriers,
    max-over-ranks, the clock-probe steps with their collectives) on cuda:0 over gloo, since RCCL refuses two ranks on one
    device.  A rehearsal of the code path the 8-GPU run takes, not a measurement: rank 0's line must come out complete
    and say so, and every rank must exit cleanly (no rank-0-only collective, no deadlock: the timeout is the check)."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--rehearse-shared-gpu", "--queries", "3000",
                        "--bank", "60001", "--dim", "256", "--classes", "100", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == n and line["config"]["ranks"] == n and "not a measurement" in line["config"]["rehearsal"]
    assert line["value"] > 0 and line["roofline"]["launches"] > 0 and line["scaling"] == "strong"
