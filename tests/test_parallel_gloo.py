"""CPU, world_size = 2 and 3 (gloo): the sample-sharded sweep over a caller-supplied communicator
(gpax_amd.parallel.predict_sharded; the torch.distributed implementation of the protocol lives OUTSIDE the product,
tools/torch_comm.py) gathers exactly the single-process result.  The engine here is the checker-backed stand-in (no GPU
in this container).  The product's own multi-process path — _lib.Rank, RCCL inside the library, no torch — is covered by
tests/test_launch.py (CPU: rendezvous / agreement / launcher) and tests/test_gpu_rank.py (GPU)."""
import os
import socket

import numpy as np
import pytest

from gpax_amd.parallel import shard_range


def test_shard_range_partitions_exactly():
    for S in [0, 1, 7, 8, 1000]:
        for world in [1, 2, 3, 8]:
            blocks = [shard_range(S, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == S
            assert all(b[1] == c[0] for b, c in zip(blocks[:-1], blocks[1:]))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_weighted_blocks_follow_the_library_rule():
    """parallel.shard_ranges_weighted restates gpx_shard_ranges_weighted (the blocks the ranks of a calibrated library sweep
    work off, csrc/rccl_bind.h): the same partition for the same weights, equal weights = shard_range."""
    from gpax_amd import _lib
    from gpax_amd.parallel import shard_ranges_weighted
    rng = np.random.default_rng(1)
    for S in (0, 1, 5, 40, 999, 1000):
        for world in (1, 2, 3, 8):
            assert shard_ranges_weighted(S, np.ones(world)) == [shard_range(S, r, world) for r in range(world)]
            for _ in range(5):
                w = rng.uniform(0.85, 1.15, world)
                assert shard_ranges_weighted(S, w) == _lib.shard_ranges_weighted(S, w)
            assert shard_ranges_weighted(S, [1.0] * (world - 1) + [0.0]) == [shard_range(S, r, world) for r in range(world)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _problem():
    from oracle import cpu_ref as ref
    import bench_inputs

    N, d, M, S, n = 60, 2, 17, 5, 2
    X, y, Xn, _ = bench_inputs.synthetic_problem(N, d, M, seed=3)
    samples = bench_inputs.synthetic_theta_samples(S, d, seed=1)
    eps = np.random.default_rng(2).standard_normal((S, n, M))
    return X, y, Xn, samples, eps


def _worker(rank, world, port, outdir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gpax_amd.parallel import predict_sharded
    from tools.torch_comm import Communicator
    from tests.oracle_engine import OracleEngine

    comm = Communicator()
    args = _problem() if rank == 0 else (None, None, None, None, None)
    res = predict_sharded(OracleEngine(), 1, *args, False, 1e-6, comm)
    # the same sweep with blocks sized by (made-up) rank speeds: 5 samples over weights (1, 0.5[, 1.5]) = blocks of 3 + 2 /
    # 2 + 1 + 2 — whoever computes a sample, the gathered result is the same
    wres = predict_sharded(OracleEngine(), 1, *args, False, 1e-6, comm, weights=[1.0, 0.5, 1.5][:world] if rank == 0 else None)
    if rank == 0:
        means, draws, infos = res
        np.savez(os.path.join(outdir, "out.npz"), means=means, draws=draws, infos=infos, wmeans=wres[0], wdraws=wres[1],
                 winfos=wres[2])
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_sweep_equals_single_process(tmp_path, world):
    import torch.multiprocessing as mp

    from oracle import cpu_ref as ref

    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "out.npz")
    X, y, Xn, samples, eps = _problem()
    _, yy, means = ref.predict(X, y, Xn, samples, eps, False, kernel="Matern", route="chol")
    np.testing.assert_allclose(got["means"], means, rtol=1e-10)
    np.testing.assert_allclose(got["draws"], yy, rtol=1e-9, atol=1e-12)
    assert got["infos"].shape == (5,) and np.all(got["infos"] == 0)
    np.testing.assert_array_equal(got["wmeans"], got["means"])
    np.testing.assert_array_equal(got["wdraws"], got["draws"])
    np.testing.assert_array_equal(got["winfos"], got["infos"])


def _model_worker(rank, world, port, outdir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gpax_amd import ExactGP, _lib
    from gpax_amd.utils import get_keys
    from tests.oracle_engine import OracleEngine
    from tools.torch_comm import Communicator

    _lib.set_engine(OracleEngine())
    m = ExactGP(2, "Matern", mean_fn=lambda x: 0.3 * x[:, 0])
    if rank == 0:  # only rank 0 holds data and samples
        X, y, Xn, samples, _ = _problem()
        m.X_train, m.y_train = m._set_data(X, y)
        res = m.predict_distributed(get_keys()[1], Xn, samples, n=2, comm=Communicator())
        single = m.predict(get_keys()[1], Xn, samples, n=2)
        np.savez(os.path.join(outdir, "model.npz"), ym=res[0], ys=res[1], ym1=single[0], ys1=single[1])
    else:
        assert m.predict_distributed(None, None, comm=Communicator()) is None
    dist.barrier()
    dist.destroy_process_group()


def test_model_predict_distributed_equals_predict(tmp_path):
    import torch.multiprocessing as mp

    mp.spawn(_model_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "model.npz")
    np.testing.assert_allclose(got["ym"], got["ym1"], rtol=1e-12)
    np.testing.assert_allclose(got["ys"], got["ys1"], rtol=1e-12)
    assert got["ys"].shape == (5, 2, 17)
