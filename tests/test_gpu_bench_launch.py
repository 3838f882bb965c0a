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
           "--exp", "10", "--backend", backend, "--no-cpu-baseline", "--roofline-iters", "2"]
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


def test_bench_refuses_several_gpus_without_a_launcher():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True,
                         text=True, timeout=120, cwd=ROOT, env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK")})
    assert out.returncode != 0 and "torch.distributed.run" in (out.stderr + out.stdout)
