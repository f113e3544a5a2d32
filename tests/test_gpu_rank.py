"""GPU: the sharded predictive sweep with one process per GPU and no PyTorch (include/gpx.h gpx_rank_*,
gpax_amd/launch.py; gpax/models/gp.py:392-395 is the axis being sharded).  On the 1-GPU test box: RCCL with one rank
(unique id, ncclCommInitRank, broadcast, all-reduce) in-process, and the multi-process control flow — rendezvous,
broadcast, ragged / empty blocks, gather, NaN rows — with 2 and 3 ranks sharing GPU 0 over the file transport; with more
GPUs visible, one rank per GPU over RCCL."""
import os
import subprocess
import sys

import numpy as np
import pytest

from bench_inputs import synthetic_problem, synthetic_theta_samples

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    return {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GPX_RDZV_DIR")}


def test_one_rank_over_rccl_equals_the_single_gpu_sweep(engine):
    from gpax_amd import _lib
    uid = _lib.rccl_unique_id()
    assert len(uid) == 128
    rk = _lib.Rank(0, 0, 1, unique_id=uid, inflight=2)
    info = rk.info()
    assert info == {"rank": 0, "nranks": 1, "inflight": 2, "transport": "rccl", "rccl_version": info["rccl_version"]}
    assert info["rccl_version"] > 0
    rk.barrier()
    assert np.array_equal(rk.allreduce_max([3.0, -1.0]), [3.0, -1.0])
    X, y, Xn, _ = synthetic_problem(700, 2, 130, seed=0)
    th = synthetic_theta_samples(9, 2, seed=1)
    eps = np.random.default_rng(2).standard_normal((9, 2, 130))
    got = rk.predict_sweep(1, 700, 2, 9, 130, 2, False, 1e-6, X=X, ells=th["k_length"], scales=th["k_scale"],
                           noises=th["noise"], yres=y, Xnew=Xn, eps=eps, want_var=True)
    engine.set_train(X)
    want = engine.predict_sweep(1, th["k_length"], th["k_scale"], th["noise"], y, Xn, False, 1e-6, eps, want_var=True)
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g, w)
    rk.close()


def test_one_rank_rehearses_every_rccl_call_of_the_path(engine, monkeypatch):
    """ncclCommInitRank with > 1 rank cannot run on this box, but every RCCL CALL of the collective can —
    GPX_RANK_FORCE_COLLECTIVES=1 makes a one-rank communicator issue the flag all-reduce, the payload broadcast and a
    send / receive group instead of skipping them: datatypes, counts, streams and group nesting are RCCL-checked, and the
    results are still the single-GPU sweep's, bit for bit."""
    from gpax_amd import _lib
    monkeypatch.setenv("GPX_RANK_FORCE_COLLECTIVES", "1")
    rk = _lib.Rank(0, 0, 1, unique_id=_lib.rccl_unique_id(), inflight=2)
    assert rk.collective_calls() == 0
    rk.barrier()                                   # all-reduce
    assert np.array_equal(rk.allreduce_max([3.0, -1.0]), [3.0, -1.0])
    assert np.array_equal(rk.bcast(np.arange(7.0)), np.arange(7.0))
    n0 = rk.collective_calls()
    assert n0 == 3
    X, y, Xn, _ = synthetic_problem(700, 2, 130, seed=0)
    th = synthetic_theta_samples(9, 2, seed=1)
    eps = np.random.default_rng(2).standard_normal((9, 2, 130))
    got = rk.predict_sweep(1, 700, 2, 9, 130, 2, False, 1e-6, X=X, ells=th["k_length"], scales=th["k_scale"],
                           noises=th["noise"], yres=y, Xnew=Xn, eps=eps, want_var=True)
    # per sweep: argument-agreement all-reduce, broadcast, failure-flag all-reduce, send + receive
    assert rk.collective_calls() - n0 == 5
    engine.set_train(X)
    want = engine.predict_sweep(1, th["k_length"], th["k_scale"], th["noise"], y, Xn, False, 1e-6, eps, want_var=True)
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g, w)
    rk.close()


@pytest.mark.parametrize("ranks", [2, 3])
def test_ranks_sharing_the_gpu_over_the_file_transport(ranks):
    from gpax_amd import launch
    rc = subprocess.run([sys.executable, "-c",
                         "import sys; sys.path.insert(0, %r); from gpax_amd import launch; "
                         "sys.exit(launch.spawn_ranks(%r, ['--share-gpu'], %d, timeout=900))"
                         % (ROOT, os.path.join(ROOT, "tools", "rank_check.py"), ranks)],
                        capture_output=True, text=True, env=_env(), timeout=1000)
    assert rc.returncode == 0, rc.stdout[-1500:] + rc.stderr[-3000:]
    assert "rank_check ok: %d ranks, transport file" % ranks in rc.stdout


def test_a_rank_that_dies_before_the_sweep_fails_the_others_instead_of_hanging_them():
    """rank 1 of 2 (file transport, shared GPU) leaves without a word before the collective; rank 0 must
    come back from gpx_rank_predict_sweep with an error within the transport's time-out (GPX_RANK_FILE_TIMEOUT)."""
    import time
    env = _env()
    env["GPX_RANK_FILE_TIMEOUT"] = "4"
    t0 = time.monotonic()
    rc = subprocess.run([sys.executable, "-c",
                         "import sys; sys.path.insert(0, %r); from gpax_amd import launch; "
                         "sys.exit(launch.spawn_ranks(%r, ['--share-gpu', '--die-rank', '1'], 2, timeout=300))"
                         % (ROOT, os.path.join(ROOT, "tools", "rank_check.py"))],
                        capture_output=True, text=True, env=env, timeout=400)
    assert rc.returncode == 0, rc.stdout[-1500:] + rc.stderr[-3000:]
    assert "rank 0: peer lost detected" in rc.stdout and "timed out waiting for" in rc.stdout
    assert time.monotonic() - t0 < 120


def test_one_rank_per_visible_gpu_over_rccl():
    from gpax_amd import _lib
    n_dev = _lib.visible_device_count()
    if n_dev < 2:
        pytest.skip("one GPU visible: RCCL across ranks needs >= 2")
    rc = subprocess.run([sys.executable, "-c",
                         "import sys; sys.path.insert(0, %r); from gpax_amd import launch; "
                         "sys.exit(launch.spawn_ranks(%r, [], %d, timeout=900))"
                         % (ROOT, os.path.join(ROOT, "tools", "rank_check.py"), n_dev)],
                        capture_output=True, text=True, env=_env(), timeout=1000)
    assert rc.returncode == 0, rc.stdout[-1500:] + rc.stderr[-3000:]
    assert "transport rccl" in rc.stdout
