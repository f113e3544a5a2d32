"""One process per GPU, torch-free: run under a launcher that sets RANK / LOCAL_RANK / WORLD_SIZE
(`python -m torch.distributed.run --nproc-per-node N tools/rank_check.py`, or gpax_amd.launch.spawn_ranks as
tests/test_gpu_rank.py does).  Every rank joins the library's communicator (gpx_rank_*), rank 0 runs the collective
predictive sweep against the plain single-GPU sweep on its own device and requires bit-identical results.
  --share-gpu   every rank on GPU 0 over the file transport (a 1-GPU box; RCCL refuses duplicate devices)
Cases: blocks of unequal size, an empty block (S < ranks), per-sample residuals, predict_in_batches blocks, a non-PD
theta, variances only (n = 0)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from bench_inputs import synthetic_problem, synthetic_theta_samples  # noqa: E402
from gpax_amd import _lib, launch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--share-gpu", action="store_true")
    ap.add_argument("--inflight", type=int, default=2)
    ap.add_argument("--die-rank", type=int, default=-1, help="this rank leaves (exit code 0, no goodbye) before the first "
                    "sweep: the others must come back from the collective with an error, not hang")
    a = ap.parse_args()
    env = launch.rank_env()
    assert env is not None, "start me under a launcher (RANK / WORLD_SIZE)"
    rk = launch.init_rank(env, device=0 if a.share_gpu else env.local_rank, inflight=a.inflight,
                          transport="file" if a.share_gpu else None, timeout=120.0)
    root = env.rank == 0
    info = rk.info()
    assert info["nranks"] == env.world and info["rank"] == env.rank
    assert np.array_equal(rk.allreduce_max([float(env.rank), -float(env.rank)]), [env.world - 1.0, 0.0])
    b = rk.bcast(np.arange(5.0) if root else np.zeros(5))
    assert np.array_equal(b, np.arange(5.0))
    eng = _lib.Engine(0 if a.share_gpu else env.local_rank) if root else None
    # blocks sized by measured GPU speed (gpx_rank_calibrate; launch.init_rank has calibrated once already): rank 1 reports
    # a tenth of its rate (the ranks may share one GPU here and probe it at the same time: anything milder drowns in that),
    # so after the clamp to [0.85, 1.15] of the mean it gets the smaller block — and every case below still has to come out
    # bit-identical, which it only does when all ranks cut the same blocks
    if env.world > 1:
        assert getattr(rk, "speeds", None) is not None and len(rk.speeds) == env.world
        if env.rank == 1:
            os.environ["GPX_RANK_SPEED_SCALE"] = "0.1"
        sp = rk.calibrate()
        os.environ.pop("GPX_RANK_SPEED_SCALE", None)
        assert abs(sp.mean() - 1.0) < 0.16 and sp.min() >= 0.85 and sp.max() <= 1.15 and sp[1] == sp.min() and sp[1] < sp[0]
        blocks = _lib.shard_ranges_weighted(40 * env.world, sp)
        sizes = [hi - lo for lo, hi in blocks]
        assert sum(sizes) == 40 * env.world and sizes[1] == min(sizes) and sizes[1] < 40 and blocks[0][0] == 0
        assert all(blocks[r][1] == blocks[r + 1][0] for r in range(env.world - 1))
    if a.die_rank >= 0:
        import time
        if env.rank == a.die_rank:
            os._exit(0)
        t0 = time.monotonic()
        kw = {}
        if root:
            X, y, Xn, _ = synthetic_problem(300, 2, 40, seed=1)
            th = synthetic_theta_samples(2 * env.world, 2, seed=2)
            kw = dict(X=X, ells=th["k_length"], scales=th["k_scale"], noises=th["noise"], yres=y, Xnew=Xn)
        try:
            rk.predict_sweep(1, 300, 2, 2 * env.world, 40, 0, False, 1e-6, want_var=True, **kw)
            print(f"rank {env.rank}: the sweep RETURNED although rank {a.die_rank} is gone", flush=True)
            sys.exit(5)
        except RuntimeError as ex:
            print(f"rank {env.rank}: peer lost detected after {time.monotonic() - t0:.1f} s: {ex}", flush=True)
        os._exit(0)  # no finalize: the communicator is broken
    cases = [  # N, d, M, S, n, kind, per-sample yres, m_slice, want_var, bad theta
        (300, 3, 70, 4 * env.world + 1, 1, 0, True, 32, False, True),
        (3100, 2, 130, 2 * env.world + 1, 2, 1, False, 0, True, False),
        (300, 2, 40, max(1, env.world - 1), 1, 1, False, 0, False, False),  # an empty block on the last rank
        (500, 2, 64, 3 * env.world, 0, 1, False, 0, True, False),           # variances only
        (260, 2, 48, 2 * env.world + 1, 1, 2, False, 0, False, False),      # periodic kernel: d + 1 values per theta
    ]
    for ci, (N, d, M, S, n, kind, per_y, m_slice, want_var, bad) in enumerate(cases):
        kw = {}
        if root:
            X, y, Xn, _ = synthetic_problem(N, d, M, seed=5 + ci)
            th = synthetic_theta_samples(S, d, seed=6 + ci)
            if kind == 2:  # packed (lengthscales.., period)
                th["k_length"] = np.concatenate([th["k_length"], np.full((S, 1), 2.5) + 0.1 * np.arange(S)[:, None]], axis=1)
            if bad:
                th["k_scale"][S // 2] = -1.0  # not positive definite: NaN rows + info, never an abort
            eps = np.random.default_rng(7 + ci).standard_normal((S, n, M)) if n else None
            yres = np.stack([y + 0.01 * k for k in range(S)]) if per_y else y
            kw = dict(X=X, ells=th["k_length"], scales=th["k_scale"], noises=th["noise"], yres=yres, Xnew=Xn, eps=eps)
        got = rk.predict_sweep(kind, N, d, S, M, n, False, 1e-6, yres_rows=S if per_y else 1, want_var=want_var,
                               m_slice=m_slice, **kw)
        if root:
            eng.set_train(X)
            want = eng.predict_sweep(kind, th["k_length"], th["k_scale"], th["noise"], yres, Xn, False, 1e-6, eps,
                                     want_var=want_var, m_slice=m_slice)
            for g, w in zip(got, want):
                np.testing.assert_array_equal(g, w)
            if bad:
                assert want[2][S // 2] != 0 and np.isnan(got[0][S // 2]).all()
        else:
            assert got is None
    # a bad array on the ROOT (the only rank that holds arrays): every rank leaves the collective with an error — the
    # others must not wait for a broadcast that never comes
    try:
        rk.predict_sweep(1, 300, 2, 3, 40, 1, False, 1e-6, X=np.zeros((7, 2)) if root else None)
        raise SystemExit(f"rank {env.rank}: a mis-shaped X on rank 0 went through")
    except (ValueError, RuntimeError, TypeError):
        pass
    rk.barrier()  # ... and the communicator is still usable
    # the model API: ExactGP.predict_distributed over this communicator against ExactGP.predict on rank 0 alone
    from gpax_amd import ExactGP
    from gpax_amd.utils import get_keys
    _lib.set_engine(None)
    m = ExactGP(2, "Matern", mean_fn=lambda x: 0.3 * x[:, 0])
    m._device = 0 if a.share_gpu else env.local_rank
    if root:
        X, y, Xn, _ = synthetic_problem(400, 2, 50, seed=9)
        th = synthetic_theta_samples(2 * env.world + 1, 2, seed=10)
        m.X_train, m.y_train = m._set_data(X, y)
        got = m.predict_distributed(get_keys()[1], Xn, dict(th), n=2, comm=rk)
        want = m.predict(get_keys()[1], Xn, dict(th), n=2)
        np.testing.assert_array_equal(got[0], want[0])
        np.testing.assert_array_equal(got[1], want[1])
    else:
        assert m.predict_distributed(None, None, comm=rk) is None
    rk.barrier()
    if root:
        print(f"rank_check ok: {env.world} ranks, transport {info['transport']}, rccl {info['rccl_version']}, "
              f"{len(cases)} cases bit-identical to the single-GPU sweep", flush=True)
    launch.finalize(env, rk)


if __name__ == "__main__":
    main()
