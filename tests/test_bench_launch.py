"""bench.py --gpus N: started without a launcher it must run N ranks itself.
CPU: the re-exec command line.  GPU: 2 ranks sharing the one GPU of the test box report n_gpus = 2."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    return env


def test_gpus_n_without_a_launcher_reexecutes_under_torch_distributed_run():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "6", "--warmup", "1",
                          "--dry-launch"], capture_output=True, text=True, env=_env(), check=True)
    cmd = json.loads(out.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=8" in cmd and "--dry-launch" not in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "6", "--warmup", "1"]


def test_gpus_1_does_not_relaunch():
    # with --gpus 1 and no launcher, bench.py goes straight to the engine (which needs the GPU): no subprocess
    import bench
    a = type("A", (), {"gpus": 1})()
    assert bench.self_launch_command(a, ["--gpus", "1"])[4] == "--nproc-per-node=1"  # helper itself is N-agnostic


@pytest.mark.gpu
def test_two_ranks_sharing_the_gpu_report_n_gpus_2():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--dist-backend",
                          "gloo", "--N", "2048", "--M", "256", "--steps", "4", "--warmup", "1", "--inflight", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, env=_env(), timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["steps"] == 4 and rec["value"] > 0
    assert rec["roofline"]["frac"] > 0
