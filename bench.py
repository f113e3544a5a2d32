#!/usr/bin/env python
"""
bench.py — headline benchmark of the MI355X exact-GP hot path (driver contract).

A "step" = one pass of the per-sample predictive pipeline (ExactGP.predict's vmap body,
gpax/models/gp.py:279-293,351-399) for one theta on synthetic data resident in HBM:
    Gram (N x N) -> blocked fp64-MFMA Cholesky (+ lml) -> k_pX -> TRSM -> mean / cov (SYRK)
    -> chol(cov) -> 1 MVN draw.
Workload = BASELINE.json configs[2] ("C3"): Matern-5/2, N=16384, d=2, M=1024.
value = posteriors per second over all ranks (weak scaling: every rank runs K steps on its own
theta samples; the sweep has no data-path collective, only the final gather of results).

    python bench.py --gpus N --steps K --warmup W
N > 1: one rank per GPU under torch.distributed.run (RCCL).  The driver launches the ranks itself; when bench.py
is started plainly with --gpus N > 1 (no WORLD_SIZE in the environment) it re-executes itself under
`python -m torch.distributed.run --nproc-per-node N` and relays the ranks' output.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 2.4 GHz x 32 flop/clk/SIMD


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=36)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--N", type=int, default=16384)
    ap.add_argument("--d", type=int, default=2)
    ap.add_argument("--M", type=int, default=1024)
    ap.add_argument("--kernel", default="Matern")
    ap.add_argument("--inflight", type=int, default=3, help="theta samples in flight per GPU (libgpx contexts)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to "
                    "exercise the multi-rank control flow on a box with fewer GPUs than ranks)")
    ap.add_argument("--share-gpu", action="store_true", help="testing: every rank uses GPU 0")
    ap.add_argument("--cpu-baseline-N", type=int, default=0, help="override the CPU sample size")
    ap.add_argument("--cpu-budget-s", type=float, default=150.0,
                    help="budget of the CPU leg; the same-workload sample (one posterior at the bench N) runs when its "
                         "estimate fits, else N is halved")
    ap.add_argument("--dry-launch", action="store_true",
                    help="print the torch.distributed.run command --gpus N would re-execute under, and exit")
    return ap.parse_args()


def self_launch_command(a, argv):
    """The command line `bench.py --gpus N` (N > 1, started without a launcher) re-executes itself under: one rank
    per GPU, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def cpu_baseline(N, d, M, kernel, budget_s=150.0):
    """The oracle (CPU restatement of the reference algorithm) timed on the host cores of this box on a bounded
    sample of the SAME workload: ONE posterior + draw at the bench N (Cholesky route — the algorithm the GPU runs),
    halving N only if the estimate exceeds the budget.  Beside it: the reference-faithful explicit-inverse route
    (gpax/models/gp.py:271-273) at the largest N whose estimate fits ~30 s, and the Cholesky route on ONE core."""
    from bench_inputs import synthetic_problem
    from oracle import cpu_ref as ref
    try:
        from threadpoolctl import threadpool_info
        pools = threadpool_info()
        threads = max([p.get("num_threads", 1) for p in pools] or [1])
        blas = ";".join(sorted({f"{p.get('internal_api')} {p.get('version')} ({p.get('threading_layer', 'n/a')}, "
                                f"{p.get('num_threads')} threads)" for p in pools}))
    except Exception:
        threads, blas = os.cpu_count() or 1, "unknown"
    # calibrate on a small case (N^3 scaling), then pick the sample size
    Xc, yc, Xn, p = synthetic_problem(2048, d, M, seed=0)
    eps = np.zeros((1, M))
    t0 = time.perf_counter()
    ref.predict_one(Xc, yc, Xn, p, eps, False, kernel=kernel, jitter=1e-6, route="chol")
    t_small = time.perf_counter() - t0
    Ns = N
    while Ns > 2048 and t_small * (Ns / 2048.0) ** 3 * 0.5 > budget_s:
        Ns //= 2
    X, y, Xnew, p = synthetic_problem(Ns, d, M, seed=0)
    t0 = time.perf_counter()
    ref.predict_one(X, y, Xnew, p, eps, False, kernel=kernel, jitter=1e-6, route="chol")
    dt = time.perf_counter() - t0
    Ni = Ns
    while Ni > 1024 and 6.0 * dt * (Ni / float(Ns)) ** 3 > 30.0:
        Ni //= 2
    Xi, yi, Xni, pi_ = synthetic_problem(Ni, d, M, seed=0)
    t0 = time.perf_counter()
    ref.predict_one(Xi, yi, Xni, pi_, eps, False, kernel=kernel, jitter=1e-6, route="inv")
    dti = time.perf_counter() - t0
    inv = {"value": 1.0 / dti, "unit": f"posteriors/s at N={Ni}", "seconds": dti, "N": Ni,
           "route": "explicit inverse, as gpax/models/gp.py:271-273"}
    one = None
    try:
        from threadpoolctl import threadpool_limits
        N1 = 2048
        X1, y1, Xn1, p1 = synthetic_problem(N1, d, M, seed=0)
        with threadpool_limits(limits=1):
            t0 = time.perf_counter()
            ref.predict_one(X1, y1, Xn1, p1, eps, False, kernel=kernel, jitter=1e-6, route="chol")
            dt1 = time.perf_counter() - t0
        one = {"value": 1.0 / dt1, "unit": f"posteriors/s at N={N1}", "seconds": dt1, "N": N1, "cores": 1,
               "all_cores_seconds_at_this_N": t_small}
    except Exception:
        pass
    return {
        "value": 1.0 / dt,
        "unit": "posteriors/s" if Ns == N else f"posteriors/s at N={Ns}",
        "cores": int(threads),
        "kind": "port",
        "sample": (f"1 posterior+draw (oracle/cpu_ref.py predict_one, Cholesky route) at N={Ns}, d={d}, M={M}"
                   f"{' = the bench workload' if Ns == N else ' (bench N halved to fit the CPU budget)'}: {dt:.2f} s; "
                   f"BLAS: {blas}; os.cpu_count()={os.cpu_count()}"),
        "seconds": dt,
        "N": Ns,
        "same_workload_as_value": Ns == N,
        "blas": blas,
        "inv_route": inv,
        "one_core": one,
    }


def gram_bytes_written(N, Np):
    """Bytes one lower-tile Gram build actually stores (32 x 512 tiles with j0 <= i0 + 31, gram.hip:49), next to the
    8 N^2 the algorithmic figure credits (the symmetric build may claim the full matrix)."""
    total = 0
    for i0 in range(0, N, 32):
        rows = min(32, N - i0)
        ntiles = min((i0 + 31) // 512 + 1, (Np + 511) // 512)
        total += rows * min(ntiles * 512, Np) * 8
    return total


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly: launch the N ranks ourselves (one process per GPU over RCCL) and relay their output
        import subprocess
        argv = [x for x in sys.argv[1:] if x != "--dry-launch"]
        cmd = self_launch_command(a, argv)
        if a.dry_launch:
            print(json.dumps({"launch": cmd}))
            return
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.run(cmd, env=env).returncode)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    if "WORLD_SIZE" in os.environ and "RANK" in os.environ:  # launched by torch.distributed.run (any N, also 1)
        # torch first: its bundled HIP runtime must be the one libgpx binds to (same SONAME)
        import torch  # noqa: F811
        import torch.distributed as dist  # noqa: F811
        if a.share_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if a.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=a.dist_backend)

    from bench_inputs import synthetic_problem, synthetic_theta_samples  # BASELINE.md §3 workloads
    from gpax_amd import _lib

    if world != a.gpus and rank == 0:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}", file=sys.stderr)
    eng = _lib.Engine(local_rank)
    kind = _lib.kernel_kind(a.kernel)
    N, d, M = a.N, a.d, a.M
    X, y, Xnew, p = synthetic_problem(N, d, M, seed=0)
    K, W = a.steps, a.warmup
    # every rank sweeps its own block of theta samples (contiguous shard of the global table)
    thetas = synthetic_theta_samples(world * (K + W), d, seed=1)
    lo = rank * (K + W)
    sl_w = slice(lo, lo + W)
    sl_k = slice(lo + W, lo + W + K)

    # Several theta samples in flight per GPU: independent libgpx contexts on the same device fill the
    # latency-bound tail of one sample's pipeline with the GEMM-heavy head of another.
    import threading
    n_fl = max(1, min(a.inflight, K // 2))  # at least two steps per context
    engines = [eng] + [_lib.Engine(local_rank) for _ in range(n_fl - 1)]
    # resident state: X, yres, Xnew, eps on the device before the timed region
    for e in engines:
        e.set_train(X)
        lml, info = e.factor(kind, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
        e.posterior(Xnew, p["noise"], 1e-6, want_cov=True)
        e.mvn_draw(np.random.default_rng(2).standard_normal((1, M)))

    def barrier():
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
        for e in engines:
            e.synchronize()

    def sweep(sl):
        """The steps of slice `sl`, split in contiguous blocks over the in-flight contexts."""
        idx = np.arange(sl.start, sl.stop)
        parts = [idx[(len(idx) * i) // n_fl:(len(idx) * (i + 1)) // n_fl] for i in range(n_fl)]
        ev = [0.0] * n_fl

        def work(i):
            if len(parts[i]):
                ev[i] = engines[i].sweep_resident(kind, thetas["k_length"][parts[i]], thetas["k_scale"][parts[i]],
                                                  thetas["noise"][parts[i]], False, 1e-6, 1)

        ts = [threading.Thread(target=work, args=(i,)) for i in range(n_fl)]
        for t_ in ts:
            t_.start()
        for t_ in ts:
            t_.join()
        return max(ev)

    if W > 0:
        sweep(sl_w)
        if W < n_fl:  # make sure every context has run the pipeline once before the timed region
            for e in engines:
                e.sweep_resident(kind, thetas["k_length"][sl_w][:1], thetas["k_scale"][sl_w][:1],
                                 thetas["noise"][sl_w][:1], False, 1e-6, 1)
    barrier()
    t0 = time.perf_counter()
    ev_ms = sweep(sl_k)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if a.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel (MFMA GEMM, trailing SYRK of the Cholesky) --------
        eng.profile_enable(True)
        eng.profile_reset()
        eng.time_stage(_lib.STAGE_PREDICT, 1)
        n_l, ms, flops = eng.profile_read(_lib.PROF_GEMM_TRAILING)
        alg_bytes = eng.profile_read_bytes(_lib.PROF_GEMM_TRAILING)
        n_o, ms_o, flops_o = eng.profile_read(_lib.PROF_GEMM_OTHER)
        n_p, ms_p, _ = eng.profile_read(_lib.PROF_POTF2)
        n_g, ms_g, bytes_g = eng.profile_read(_lib.PROF_GRAM)
        eng.profile_enable(False)
        achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        # stage timings (device events), fit step = lml + analytic gradient (one leapfrog of NUTS)
        stages = {}
        for name, st in [("gram", _lib.STAGE_GRAM), ("potrf", _lib.STAGE_POTRF), ("fit_step", _lib.STAGE_FITSTEP),
                         ("posterior", _lib.STAGE_POSTERIOR), ("predict", _lib.STAGE_PREDICT)]:
            eng.time_stage(st, 1)  # warm-up
            stages[name + "_ms"] = float(np.median([eng.time_stage(st, 1) for _ in range(5)]))
        # HBM traffic of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
        # (profiles/r01/traffic.json; FETCH_SIZE doubled per the gfx950 correction, MI355X_MICROARCH.md §HBM)
        traffic, traffic_note = None, None
        for rnd in ("r02", "r01"):
            tj = os.path.join(ROOT, "profiles", rnd, "traffic.json")
            if os.path.exists(tj) and (N, d, M) == (16384, 2, 1024):
                t = json.load(open(tj))
                if "FETCH_SIZE" in t and "WRITE_SIZE" in t:
                    traffic = (2.0 * t["FETCH_SIZE"]["avg_per_launch"] + t["WRITE_SIZE"]["avg_per_launch"]) * 1024.0
                    traffic_note = (f"bytes per launch, profiles/{rnd}/{{fetch,write}}.md: (2*FETCH_SIZE + WRITE_SIZE) KB"
                                    + ("" if rnd == "r02" else " (previous round's kernel)"))
                    break
        post_flops = N ** 3 / 3 + N * N * M + N * M * M + 2 * N * N + 2 * N * M + M ** 3 / 3 + M * M
        out = {
            "metric": f"exactgp_posteriors_per_sec_N{N}_d{d}",
            "value": world * K / dt,
            "unit": "posteriors/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": dt / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"C3: ExactGP {a.kernel} N={N} d={d} M={M}, 1 MVN draw per theta sample "
                                   "(BASELINE.json configs[2]); per-rank theta shard, inputs resident in HBM",
                       "parallelism": f"sample-sharded x{world}, {n_fl} samples in flight per GPU"},
            "roofline": {
                "bound": "mfma",
                "kernel": "gpx::gemm_nt128_kernel<1,1> (Cholesky trailing SYRK, lower tiles, LDS-direct staging; K = 1024 while the "
                          "factorisation is GEMM-bound, 512 in the chain-bound tail)",
                "achieved": achieved,
                "peak": FP64_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                "traffic": traffic,
                "traffic_note": traffic_note,
                "alg_bytes_per_launch_avg": (alg_bytes / n_l) if n_l else None,
                "launches": n_l,
                "avg_launch_ms": ms / n_l if n_l else None,
                "alg_flops_per_launch_avg": flops / n_l if n_l else None,
            },
            "event_ms_longest_context": ev_ms,
            "inflight_per_gpu": n_fl,
            "pipeline_tflops": post_flops / (dt / K) / 1e12,
            "pipeline_frac_of_fp64_peak": post_flops / (dt / K) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
            "stages": stages,
            "fit_step_tflops": N ** 3 / (stages["fit_step_ms"] * 1e-3) / 1e12,
            "potrf_tflops": (N ** 3 / 3) / (stages["potrf_ms"] * 1e-3) / 1e12,
            "kernel_classes_ms_per_predict": {"gemm_trailing": ms, "gemm_other": ms_o, "potf2": ms_p, "gram": ms_g},
            "gram_alg_GBps": bytes_g / (ms_g * 1e-3) / 1e9 if ms_g > 0 else None,
            "gram_written_GBps": (gram_bytes_written(N, (N + 1 + 127) // 128 * 128) / (stages["gram_ms"] * 1e-3) / 1e9),
            "gram_note": "alg = 8 N^2 credited to the symmetric build (SURVEY 8d); written = bytes the lower 32x512 "
                         "tiles actually store, over the stand-alone Gram stage",
            "mfma_f64_microbench_tflops": eng.mfma_f64_peak(),
            "lml_check": lml,
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_baseline_N or N, d, M, a.kernel, a.cpu_budget_s)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
