"""bench.py launched the way the multi-GPU runs launch it: `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`, one rank per process, rank 0
printing the one JSON line.  Two ranks on a small grid; on a one-GPU box they share device 0 over gloo
(`--backend gloo`, messages staged through the host), where two or more devices are visible they run on RCCL.
Marked gpu."""
import json
import os
import subprocess
import sys

import pytest

from tests.mp_util import free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_runs_under_torch_distributed_run():
    import torch
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--exp", "10", "--backend", backend, "--no-cpu-baseline", "--roofline-iters", "2", "--overlap-min-rows", "512"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["config"]["grid"] == [2 * 2048 + 1, 2049] and d["config"]["backend"] == backend
    assert d["config"]["exchange_overlap"]["rounds_per_cycle"] >= 2
    assert d["value"] > 0 and 0 < d["roofline"]["frac"] < 1 and 0 < d["cycle"]["frac_of_peak"] < 1
    assert d["metric"] == "unknowns*V-cycles/s" and d["unit"] == "unknowns*V-cycles/s"
    # what a first multi-GPU run is read and re-tuned from: the exchange round and the all-gather timed at start-up (the
    # distribution depth was chosen from them), and every rank's own record
    cal = d["config"]["comm"]["calibration"]
    assert cal["t_round_us"] > 0 and cal["gather_gbps"] > 0 and cal["round_gpu_us"] > 0 and cal["allgather_gpu_us"] > 0
    assert d["config"]["comm"]["mode"] == ("native" if backend == "nccl" else "staged") == cal["comm_mode"]
    assert d["config"]["exchange_overlap"]["min_rows_per_rank"] == 512 and d["config"]["exchange_overlap"]["passes_cut"] > 0
    assert [r["rank"] for r in d["per_rank"]] == [0, 1]
    for r in d["per_rank"]:
        assert r["host_issue_ms_per_cycle"] > 0 and r["finest_pass_ms"] > 0 and r["ms_per_step_local"] > 0
        assert r["rounds_per_cycle"] == d["config"]["exchange_overlap"]["rounds_per_cycle"] and r["passes_cut"] > 0
    assert d["host_issue_ms_per_cycle"] > 0
    assert d["pcie"]["bytes"] == 2048 * 2049 * 8 and d["pcie"]["h2d_gbps"] > 0 and d["pcie"]["d2h_gbps"] > 0


def test_bench_takes_the_distribution_depth_from_the_command_line():
    import torch
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--exp", "9", "--backend", backend, "--no-cpu-baseline", "--roofline-iters", "2", "--dist-levels", "2", "--no-pcie"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["config"]["distributed_levels"] == 2 and d["config"]["comm"]["calibration"] is None and d["pcie"] is None


def test_bench_refuses_several_gpus_without_a_launcher():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True,
                         text=True, timeout=120, cwd=ROOT, env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK")})
    assert out.returncode != 0 and "torch.distributed.run" in (out.stderr + out.stdout)
