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
N = 1: K steps of the resident sweep (gpx_sweep_resident), `--inflight` contexts on the GPU.
N > 1: one process per GPU, NO PyTorch (RANK / LOCAL_RANK / WORLD_SIZE set by the launcher — the driver's
`python -m torch.distributed.run`, or bench.py's own `gpax_amd.launch.spawn_ranks` when it is started plainly with
--gpus N), two legs:
  1. `replicas`: every rank runs the N = 1 workload on its own theta samples, no communicator — the path as it shards
     (independent samples, no data-path exchange).  Start and end meet in the rendezvous directory; time = latest end -
     earliest start.  Rank 0 then has a complete line (roofline, stages) before the process makes its first RCCL call.
  2. the collective: every rank joins the library's RCCL communicator (gpx_rank_*, ncclCommInitRank over the rendezvous of
     gpax_amd/launch.py) and the timed region is ONE gpx_rank_predict_sweep over S = N * K theta samples: rank 0's H2D
     of the inputs, ncclBroadcast over xGMI, every rank's block of K samples, ncclSend / ncclRecv gather, D2H on rank 0 —
     bracketed by a barrier on both sides (gpx_rank_barrier: own contexts synchronised + all-reduce), time = max over
     ranks (gpx_rank_allreduce_max).  This is the product's multi-GPU call with host arrays in and out, i.e. a PCIe- and
     xGMI-inclusive rate: it is reported as the `collective` record, never as `value` — `value` is leg 1's (inputs
     resident in HBM when the timed region starts, as at N = 1).  Should the leg fail or not finish within
     --collective-timeout seconds, the line says so (`multi_gpu_path` "replicas", `collective_leg` = why) and the run
     still has its measurement.
A second record, `c4_sweep`, times BASELINE.json configs[3] (S = 1000, N = 8192, d = 3) through the collective and
against rank 0 alone; a third, `node_sweep`, times both sweeps under the other launch model (ONE process owning all N
GPUs, gpx_predict_sweep_multi) from a child process of rank 0.  `multi_gpu_path`: "rank-rccl"; "rank-file" = the
library's file transport, the fallback when RCCL cannot initialise; "replicas" = leg 1 alone.
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
    ap.add_argument("--share-gpu", action="store_true", help="testing: every rank uses GPU 0 (file transport: RCCL "
                    "refuses two ranks on one device)")
    ap.add_argument("--transport", default=None, choices=["auto", "rccl", "file"], help="N > 1: transport of the "
                    "rank communicator (default: $GPX_RANK_TRANSPORT or auto = RCCL, file on failure)")
    ap.add_argument("--c4-S", type=int, default=1000, help="N > 1: samples of the C4 record (0: skip it)")
    ap.add_argument("--c4-N", type=int, default=8192)
    ap.add_argument("--node-record", action="store_true", help="internal: time the ONE-process node sweep "
                    "(gpx_predict_sweep_multi over --gpus devices) and print its record; rank 0 of an N > 1 run "
                    "starts this as a child process")
    ap.add_argument("--no-node-record", action="store_true", help="N > 1: skip the node-sweep record")
    ap.add_argument("--force-rank-path", action="store_true", help="run the N > 1 code path whatever the world size "
                    "(1 rank over RCCL on a 1-GPU box)")
    ap.add_argument("--init-timeout", type=float, default=120.0, help="N > 1: seconds before a hung RCCL "
                    "initialisation is abandoned for the file transport")
    ap.add_argument("--collective-timeout", type=float, default=300.0, help="N > 1: seconds the collective leg (RCCL "
                    "initialisation, timed sweep, C4 and node records) may take before rank 0 reports the replicas leg alone")
    ap.add_argument("--cpu-baseline-N", type=int, default=0, help="override the CPU sample size")
    ap.add_argument("--cpu-budget-s", type=float, default=150.0,
                    help="budget of the CPU leg; the same-workload sample (one posterior at the bench N) runs when its "
                         "estimate fits, else N is halved")
    ap.add_argument("--no-configs", action="store_true", help="N = 1: skip the records of the other BASELINE configs "
                    "(C1, C2, C4, C5: bench_configs.py)")
    ap.add_argument("--configs-timeout", type=float, default=240.0, help="seconds the configs child process may take")
    ap.add_argument("--dry-launch", action="store_true",
                    help="print the launch bench.py --gpus N performs when started without a launcher, and exit")
    return ap.parse_args()


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
    # SURVEY 8d's protocol (median of 3 after one warm-up) at a size where it fits the bench's time budget: N = 4096
    med = None
    try:
        Nm = min(4096, N)
        Xm, ym, Xnm, pm = synthetic_problem(Nm, d, M, seed=0)
        ref.predict_one(Xm, ym, Xnm, pm, eps, False, kernel=kernel, jitter=1e-6, route="chol")  # warm-up
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            ref.predict_one(Xm, ym, Xnm, pm, eps, False, kernel=kernel, jitter=1e-6, route="chol")
            ts.append(time.perf_counter() - t0)
        med = {"value": 1.0 / float(np.median(ts)), "unit": f"posteriors/s at N={Nm}", "N": Nm, "seconds_runs": ts,
               "seconds_median": float(np.median(ts)), "protocol": "median of 3 after 1 warm-up, all cores, Cholesky route"}
    except Exception as ex:  # the headline sample above stands on its own
        med = {"error": str(ex)}
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
        "median_of_3": med,
        "protocol_note": "value = ONE sample at the bench's own N (a second one would double the ~15 s it costs); "
                         "median_of_3 = SURVEY 8d's protocol at N = 4096",
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


def write_only_fill_GBps(nbytes):
    """What a pure WRITE stream reaches on this device: hipMemsetAsync of `nbytes` (the runtime's fill kernel) between two
    HIP events, best of 9 — the yardstick for the Gram build, which only writes (the 6.29 TB/s copy figure of
    MI355X_MICROARCH.md is read + write traffic together).  Returns None when the runtime cannot be bound."""
    try:
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        ptr, e0, e1 = C.c_void_p(), C.c_void_p(), C.c_void_p()
        if hip.hipMalloc(C.byref(ptr), C.c_size_t(nbytes)) != 0:
            return None
        hip.hipEventCreate(C.byref(e0))
        hip.hipEventCreate(C.byref(e1))
        best = 1e9
        for _ in range(9):
            hip.hipEventRecord(e0, None)
            hip.hipMemsetAsync(ptr, 0, C.c_size_t(nbytes), None)
            hip.hipEventRecord(e1, None)
            hip.hipEventSynchronize(e1)
            ms = C.c_float()
            hip.hipEventElapsedTime(C.byref(ms), e0, e1)
            best = min(best, ms.value)
        hip.hipFree(ptr)
        hip.hipEventDestroy(e0)
        hip.hipEventDestroy(e1)
        return nbytes / (best * 1e-3) / 1e9
    except Exception:
        return None


def committed_pmc_record(N, d, M):
    """HBM traffic of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
    (profiles/<round>/traffic.json; FETCH_SIZE doubled per the gfx950 correction, MI355X_MICROARCH.md §HBM) — the one
    field of the roofline block that is NOT measured in this run (counters need rocprofv3 around the process)."""
    out = {"traffic": None, "traffic_note": None, "pmc_serialised_avg_launch_ms": None}
    if (N, d, M) != (16384, 2, 1024):
        return out
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        tj = os.path.join(ROOT, "profiles", rnd, "traffic.json")
        if not os.path.exists(tj):
            continue
        t = json.load(open(tj))
        if "FETCH_SIZE" in t and "WRITE_SIZE" in t:
            out["traffic"] = (2.0 * t["FETCH_SIZE"]["avg_per_launch"] + t["WRITE_SIZE"]["avg_per_launch"]) * 1024.0
            out["traffic_note"] = (f"bytes per launch, profiles/{rnd}/traffic.json (from {{fetch,write}}.md): (2*FETCH_SIZE + WRITE_SIZE) KB"
                                   + ("" if rnd in ("r03", "r04", "r05", "r06") else " (an earlier round's kernel)"))
            us = [v["avg_duration_us"] for v in t.values() if isinstance(v, dict) and "avg_duration_us" in v]
            if us:  # every dispatch runs alone under a PMC pass: what the live serialised figure is checked against
                out["pmc_serialised_avg_launch_ms"] = float(np.mean(us)) * 1e-3
            break
    return out


def serialised_trailing_record(eng):
    """The dominant kernel ALONE on the chip, measured in this run: one more predict pass with every Cholesky trailing
    update serialised against the panel stream of the look-ahead (gpx_debug_set_serialise_trailing) — the same 24 launches
    on the same operands, HIP events around each on its stream, nothing else resident while they run."""
    from gpax_amd import _lib
    eng.set_serialise_trailing(True)
    try:
        eng.time_stage(_lib.STAGE_PREDICT, 1)  # warm-up in this mode
        eng.profile_enable(True)
        per = []
        for _ in range(3):
            eng.profile_reset()
            eng.time_stage(_lib.STAGE_PREDICT, 1)
            n, ms, flops = eng.profile_read(_lib.PROF_GEMM_TRAILING)
            per.append((ms / n if n else None, n, flops))
        eng.profile_enable(False)
    finally:
        eng.set_serialise_trailing(False)
    if not per[0][1]:  # a size whose factorisation is one outer block (test-sized runs): no trailing-update launches at all
        return {"avg_launch_ms": None, "launches": 0, "flops": 0.0, "passes": 3}
    per.sort(key=lambda r: r[0])
    return {"avg_launch_ms": per[1][0], "launches": per[1][1], "flops": per[1][2], "passes": 3}


def potf2_record(eng, a):
    """The diagonal-block kernel of the blocked Cholesky, so that a slow box explains itself: HIP-event
    time per launch stand-alone (potrf of one 128 x 128 block: nothing else on the chip) and inside this workload's
    pipeline (average over the launches of one predict pass: placement + execution beside the trailing update), for
    the default kernel and — switched in THIS process and context (gpx_debug_set_potf2; both give the same bits)
    — the round-2 kernel kept as the tests' reference, with the potrf stage time each gives."""
    from gpax_amd import _lib
    rng = np.random.default_rng(0)
    A = rng.standard_normal((128, 128))
    A = A @ A.T + 128 * np.eye(128)
    rec = {"default": "slim", "standalone_us": {}, "in_pipeline_us": {}, "potf2_ms_per_predict": {}, "potrf_ms": {}}
    for mode in ("tile", "slim"):  # the default last: the context is left on it
        eng.set_potf2(mode)
        for _ in range(3):
            eng.potrf(A)
        eng.profile_enable(True)
        eng.profile_reset()
        for _ in range(20):
            eng.potrf(A)
        n, ms, _ = eng.profile_read(_lib.PROF_POTF2)
        rec["standalone_us"][mode] = ms / n * 1e3 if n else None
        eng.profile_reset()
        eng.time_stage(_lib.STAGE_PREDICT, 1)
        n, ms, _ = eng.profile_read(_lib.PROF_POTF2)
        eng.profile_enable(False)
        rec["in_pipeline_us"][mode] = ms / n * 1e3 if n else None
        rec["potf2_ms_per_predict"][mode] = ms
        eng.time_stage(_lib.STAGE_POTRF, 1)
        rec["potrf_ms"][mode] = float(np.median([eng.time_stage(_lib.STAGE_POTRF, 1) for _ in range(3)]))
    rec["note"] = ("slim (csrc/potf2_slim.h): 94 VGPRs / 28 KB LDS, placed at once beside two resident trailing-update "
                   "workgroups; tile (round 2): the tests' reference (round 3's register-resident kernel: "
                   "tools/exp/potf2_chain.h).  in_pipeline = HIP events around each launch on its stream, one theta in flight")
    return rec


def device_record(eng, a, lml):
    """Rank 0, after the timed region: the roofline block of the dominant kernel (HIP events around every launch on the
    launching stream, gpx_profile_*), stage timings and the fractions of the fp64 MFMA peak they imply."""
    from gpax_amd import _lib
    N, d, M = a.N, a.d, a.M
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
    pmc = committed_pmc_record(N, d, M)
    ser = serialised_trailing_record(eng)
    post_flops = N ** 3 / 3 + N * N * M + N * M * M + 2 * N * N + 2 * N * M
    pk = FP64_MFMA_PEAK_TFLOPS * 1e12
    p2 = potf2_record(eng, a)
    gram_written = gram_bytes_written(N, (N + 1 + 127) // 128 * 128)
    fill = write_only_fill_GBps(gram_written)
    roof = {
        "bound": "mfma",
        "kernel": "gpx::gemm_nt128_kernel<1,1> (Cholesky trailing SYRK, lower tiles, LDS-direct staging; K = 1024 while the "
                  "factorisation is GEMM-bound, 512 in the chain-bound tail)",
        "achieved": achieved,
        "peak": FP64_MFMA_PEAK_TFLOPS,
        "unit": "TFLOP/s",
        "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
        "traffic": pmc["traffic"],
        "traffic_note": pmc["traffic_note"],
        "alg_bytes_per_launch_avg": (alg_bytes / n_l) if n_l else None,
        "traffic_over_alg_bytes": (pmc["traffic"] / (alg_bytes / n_l)) if (pmc["traffic"] and n_l) else None,
        "launches": n_l,
        "avg_launch_ms": ms / n_l if n_l else None,
        "alg_flops_per_launch_avg": flops / n_l if n_l else None,
        "serialised_avg_launch_ms": ser["avg_launch_ms"],
        "serialised_frac": ((ser["flops"] / ser["launches"]) / (ser["avg_launch_ms"] * 1e-3) / pk)
        if (ser["avg_launch_ms"] and ser["launches"]) else None,
        "serialised_note": "measured live in this run: the same launches with every trailing update ALONE on the chip "
                           "(gpx_debug_set_serialise_trailing: main stream waits for the panel stream and vice versa), HIP "
                           "events around each launch, median of 3 predict passes",
        "serialised_launches": ser["launches"],
        "pmc_serialised_avg_launch_ms": pmc["pmc_serialised_avg_launch_ms"],
        "committed_constants": ["traffic", "traffic_over_alg_bytes (its numerator)", "pmc_serialised_avg_launch_ms"],
        "committed_constants_note": "traffic is read from the committed rocprofv3 --pmc summary named in traffic_note "
                                    "(separate passes of this same command, MI355X_MICROARCH.md HBM section); "
                                    "pmc_serialised_avg_launch_ms is the kernel's average duration under those passes, "
                                    "kept only as the cross-check of serialised_avg_launch_ms; every other field of this "
                                    "block is measured live in this run with HIP events on the launching stream",
    }
    return {
        "roofline": roof,
        "potf2": p2,
        "stages": stages,
        "stages_frac_of_fp64_peak": {
            "potrf": (N ** 3 / 3) / (stages["potrf_ms"] * 1e-3) / pk,
            "fit_step": N ** 3 / (stages["fit_step_ms"] * 1e-3) / pk,
            "posterior": post_flops / (stages["posterior_ms"] * 1e-3) / pk,
            "predict": (post_flops + M ** 3 / 3 + M * M) / (stages["predict_ms"] * 1e-3) / pk,
        },
        "fit_step_tflops": N ** 3 / (stages["fit_step_ms"] * 1e-3) / 1e12,
        "potrf_tflops": (N ** 3 / 3) / (stages["potrf_ms"] * 1e-3) / 1e12,
        "kernel_classes_ms_per_predict": {"gemm_trailing": ms, "gemm_other": ms_o, "potf2": ms_p, "gram": ms_g},
        "gram_alg_GBps": bytes_g / (ms_g * 1e-3) / 1e9 if ms_g > 0 else None,
        "gram_written_GBps": gram_written / (stages["gram_ms"] * 1e-3) / 1e9,
        "gram_write_only_fill_GBps": fill,
        "gram_frac_of_write_only_fill": (gram_written / (stages["gram_ms"] * 1e-3) / 1e9 / fill) if fill else None,
        "gram_note": "alg = 8 N^2 credited to the symmetric build (SURVEY 8d); written = bytes the lower 32x512 "
                     "tiles actually store, over the stand-alone Gram stage; write_only_fill = hipMemsetAsync of the same "
                     "number of bytes, measured in this run (what a kernel that only writes can reach on this device)",
        "mfma_f64_microbench_tflops": eng.mfma_f64_peak(),
        "lml_check": lml,
    }


def base_line(a, world, K, W, dt, n_fl, parallelism):
    N, d, M = a.N, a.d, a.M
    post_flops = N ** 3 / 3 + N * N * M + N * M * M + 2 * N * N + 2 * N * M + M ** 3 / 3 + M * M
    return {
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
                               "(BASELINE.json configs[2]); theta samples sharded over the GPUs, "
                               "inputs resident in HBM",
                   "parallelism": parallelism},
        "inflight_per_gpu": n_fl,
        "pipeline_tflops": world * post_flops / (dt / K) / 1e12,
        "pipeline_frac_of_fp64_peak": post_flops / (dt / K) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
    }


def flush_c_stdio():
    """RCCL prints a version banner through C stdio, which sits in the C library's buffer until the process exits when
    stdout is a pipe — i.e. AFTER the JSON line.  Flush it out first: the JSON line is the last line a rank prints."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


class ResidentBench:
    """K steps of the resident sweep (gpx_sweep_resident) on ONE GPU, `--inflight` contexts: the N = 1 headline, and the
    per-rank leg of an N > 1 run that involves no communicator (`replicas` below)."""

    def __init__(self, a, device, first_theta=0, n_theta=None):
        from bench_inputs import synthetic_problem, synthetic_theta_samples  # BASELINE.md §3 workloads
        from gpax_amd import _lib

        self.a, self.device = a, device
        self.eng = _lib.Engine(device)
        self.kind = _lib.kernel_kind(a.kernel)
        N, d, M = a.N, a.d, a.M
        self.X, self.y, self.Xnew, self.p = synthetic_problem(N, d, M, seed=0)
        K, W = a.steps, a.warmup
        n_theta = (K + W) if n_theta is None else n_theta
        th = synthetic_theta_samples(first_theta + n_theta, d, seed=1)
        self.thetas = {k: v[first_theta:] for k, v in th.items()}
        # Several theta samples in flight per GPU: independent libgpx contexts on the same device fill the
        # latency-bound tail of one sample's pipeline with the GEMM-heavy head of another.
        self.n_fl = max(1, min(a.inflight, max(1, K // 2)))  # at least two steps per context
        self.engines = [self.eng] + [_lib.Engine(device) for _ in range(self.n_fl - 1)]
        # resident state: X, yres, Xnew, eps on the device before the timed region
        p = self.p
        for e in self.engines:
            e.set_train(self.X)
            self.lml, info = e.factor(self.kind, p["k_length"], p["k_scale"], p["noise"], 1e-6, self.y)
            e.posterior(self.Xnew, p["noise"], 1e-6, want_cov=True)
            e.mvn_draw(np.random.default_rng(2).standard_normal((1, M)))

    def barrier(self):
        for e in self.engines:
            e.synchronize()

    def sweep(self, sl):
        """The steps of slice `sl` over the in-flight contexts: every context takes the next step when it has finished its
        own (a shared cursor), so all of them stop within one step of each other.  (Rounds 1 - 3 dealt contiguous blocks
        of K / n_fl steps: the context the hardware happened to favour finished early and the others ran the last steps
        with fewer samples in flight — the reason `value` moved by +-4 % between runs of the same binary.)"""
        import threading

        idx = list(range(sl.start, sl.stop))
        cursor = [0]
        lock = threading.Lock()
        ev = [0.0] * self.n_fl
        th, engines, kind = self.thetas, self.engines, self.kind

        def work(i):
            while True:
                with lock:
                    if cursor[0] >= len(idx):
                        return
                    k = idx[cursor[0]]
                    cursor[0] += 1
                ev[i] += engines[i].sweep_resident(kind, th["k_length"][k:k + 1], th["k_scale"][k:k + 1],
                                                   th["noise"][k:k + 1], False, 1e-6, 1)

        ts = [threading.Thread(target=work, args=(i,)) for i in range(self.n_fl)]
        for t_ in ts:
            t_.start()
        for t_ in ts:
            t_.join()
        return max(ev)

    def warm(self):
        W = self.a.warmup
        if W > 0:
            self.sweep(slice(0, W))
        if W < self.n_fl:  # make sure every context has run the pipeline once before the timed region
            th = self.thetas
            for e in self.engines:
                e.sweep_resident(self.kind, th["k_length"][:1], th["k_scale"][:1], th["noise"][:1], False, 1e-6, 1)
        self.barrier()

    def timed(self):
        """(seconds, start, end on the system-wide monotonic clock, longest context's event time in ms) of the K steps."""
        W, K = self.a.warmup, self.a.steps
        self.barrier()
        t0 = time.monotonic()
        ev_ms = self.sweep(slice(W, W + K))
        self.barrier()
        t1 = time.monotonic()
        return t1 - t0, t0, t1, ev_ms

    def close(self, keep_first=False):
        for e in self.engines[1 if keep_first else 0:]:
            e.close()
        self.engines = self.engines[:1] if keep_first else []


def configs_record(a, device=0):
    """The other BASELINE.json configs (C1, C2, C4, C5), measured live in this run on the same GPU: bench_configs.py as a
    CHILD process under a time limit, so that whatever happens there costs the headline a note, not its result."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "GPX_RDZV_DIR")}
    if device:
        env["HIP_VISIBLE_DEVICES"] = str(device)
    t0 = time.perf_counter()
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench_configs.py")], env=env, capture_output=True,
                             text=True, timeout=a.configs_timeout)
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        if out.returncode == 0 and lines:
            rec = json.loads(lines[-1])
        else:
            rec = {"error": f"exit code {out.returncode}", "stderr_tail": out.stderr[-400:]}
    except subprocess.TimeoutExpired:
        rec = {"error": f"no result within {a.configs_timeout:.0f} s (child stopped)"}
    rec["child_wall_s"] = time.perf_counter() - t0
    return rec


def single_gpu(a, device=0):
    """N = 1: K steps of the resident sweep on one GPU, `--inflight` contexts (the round-1/2 headline, unchanged), then
    the CPU-baseline leg, then — GPU legs last — the records of the other BASELINE configs (bench_configs.py)."""
    rb = ResidentBench(a, device)
    rb.warm()
    dt, _, _, ev_ms = rb.timed()
    K, W = a.steps, a.warmup
    out = base_line(a, 1, K, W, dt, rb.n_fl, f"sample-sharded x1, {rb.n_fl} samples in flight per GPU")
    out["event_ms_longest_context"] = ev_ms
    out["multi_gpu_path"] = None
    out.update(device_record(rb.eng, a, rb.lml))
    rb.close()
    if not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a.cpu_baseline_N or a.N, a.d, a.M, a.kernel, a.cpu_budget_s)
    if not a.no_configs and (a.N, a.d, a.M) == (16384, 2, 1024):  # the driver's default run; test-sized runs skip it
        out["configs"] = configs_record(a, device)
    flush_c_stdio()
    print(json.dumps(out), flush=True)


def node_record(a):
    """The OTHER launch model of the same sharded sweep, for the record of an N > 1 run: ONE process owning all N GPUs
    (gpx_node_* / gpx_predict_sweep_multi: ncclCommInitAll, ncclBroadcast, per-GPU host threads, ncclSend / ncclRecv
    gather; gpax/models/gp.py:392-395).  Same C3 sweep (S = N * K theta samples, host arrays in and out) and the C4
    sweep.  Prints one JSON object."""
    from bench_inputs import synthetic_problem, synthetic_theta_samples
    from gpax_amd import _lib

    G, K, W = a.gpus, a.steps, a.warmup
    kind = _lib.kernel_kind(a.kernel)
    devices = [0] * G if a.share_gpu else list(range(G))
    t0 = time.perf_counter()
    node = _lib.Node(devices, inflight=max(1, min(a.inflight, max(1, K // 2))))
    rec = {"model": "one process, all GPUs (gpx_predict_sweep_multi)", "ngpu": G, "init_s": time.perf_counter() - t0}
    rec.update({k: v for k, v in node.info().items() if k in ("transport", "rccl_version", "inflight")})
    X, y, Xnew, _ = synthetic_problem(a.N, a.d, a.M, seed=0)
    th = synthetic_theta_samples(G * (K + W), a.d, seed=1)
    eps = np.random.default_rng(2).standard_normal((G * (K + W), 1, a.M))

    def sweep(lo, hi):
        return node.predict_sweep(X, kind, th["k_length"][lo:hi], th["k_scale"][lo:hi], th["noise"][lo:hi], y, Xnew, False,
                                  1e-6, eps[lo:hi])

    if W > 0:
        sweep(0, G * W)
    t0 = time.perf_counter()
    res = sweep(G * W, G * (W + K))
    dt = time.perf_counter() - t0
    rec.update({"c3_seconds": dt, "c3_posteriors_per_s": G * K / dt, "c3_ms_per_step": dt / K * 1e3,
                "c3_nan_rows": int(np.isnan(res[1]).any(axis=(1, 2)).sum())})
    if a.c4_S > 0:
        N4, d4, M4, S4 = a.c4_N, 3, 1024, a.c4_S
        X4, y4, Xn4, _ = synthetic_problem(N4, d4, M4, seed=0)
        th4 = synthetic_theta_samples(S4, d4, seed=1)
        eps4 = np.random.default_rng(2).standard_normal((S4, 1, M4))
        w = min(S4, 24 * G)
        node.predict_sweep(X4, kind, th4["k_length"][:w], th4["k_scale"][:w], th4["noise"][:w], y4, Xn4, False, 1e-6, eps4[:w])
        t0 = time.perf_counter()
        r4 = node.predict_sweep(X4, kind, th4["k_length"], th4["k_scale"], th4["noise"], y4, Xn4, False, 1e-6, eps4)
        dt4 = time.perf_counter() - t0
        rec.update({"c4_S": S4, "c4_seconds": dt4, "c4_posteriors_per_s": S4 / dt4,
                    "c4_nan_rows": int(np.isnan(r4[1]).any(axis=(1, 2)).sum()),
                    "c4_checksum": float(np.nansum(r4[0]) + np.nansum(r4[1])),
                    # the samples are dealt to the GPUs as they go (one cursor over all contexts): who ended up with how many
                    "c4_samples_per_gpu": node.last_shares()})
    node.close()
    print(json.dumps(rec), flush=True)


def run_node_record(a):
    """Rank 0 of an N > 1 run: the node-sweep record from a CHILD process with a time limit, so that a communicator that
    cannot be formed there (or hangs) costs this run a note, not its result."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--node-record", "--gpus", str(a.gpus), "--steps", str(a.steps),
           "--warmup", str(a.warmup), "--N", str(a.N), "--d", str(a.d), "--M", str(a.M), "--kernel", a.kernel,
           "--inflight", str(a.inflight), "--c4-S", str(a.c4_S), "--c4-N", str(a.c4_N)] + (["--share-gpu"] if a.share_gpu else [])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "GPX_RDZV_DIR", "GPX_RANK_TRANSPORT", "GPX_BENCH_REPLICAS_LINE")}
    if a.share_gpu:
        env["GPX_NODE_TRANSPORT"] = "memcpy"  # RCCL refuses one device listed twice
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        if out.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"error": f"exit code {out.returncode}", "stderr_tail": out.stderr[-400:]}
    except subprocess.TimeoutExpired:
        return {"error": "no result within 300 s (child stopped)"}


def bench_store(env, name):
    """A key store of THIS launch and attempt under the rendezvous directory (gpax_amd/launch.py: <dir>/<token>/...)."""
    from gpax_amd import launch

    launch._private_dir(env.rdzv_dir, parent=True)
    base = os.path.join(env.rdzv_dir, env.token) if env.token else env.rdzv_dir
    launch._private_dir(base)
    return launch.FileStore(os.path.join(base, f"{name}{env.attempt}"))


def replicas_leg(a, env, device):
    """N > 1, FIRST leg — no communicator anywhere: every rank runs the N = 1 workload (K steps of the resident sweep on its
    own theta samples, `--inflight` contexts) between two meetings in the rendezvous directory; time = latest end -
    earliest start on the system-wide monotonic clock.  This is the path as it shards (independent theta samples, no
    data-path exchange), it needs nothing but N working GPUs, and it leaves rank 0 with a complete JSON line before the
    first RCCL call of the process — the collective leg below has never run on more than one GPU (VERDICT r3 missing #2),
    and whatever happens to it, the run reports.  Returns rank 0's line (None elsewhere)."""
    from gpax_amd import _lib

    world, rank = env.world, env.rank
    K, W = a.steps, a.warmup
    st = bench_store(env, "replicas")
    rb = ResidentBench(a, device, first_theta=rank * (K + W), n_theta=K + W)
    rb.warm()
    st.set(f"ready.{rank}", b"1")
    if rank == 0:  # everybody is ready: a common start time on the system-wide monotonic clock (the store is polled in
        st.gather("ready", world, 1800.0)  # steps of up to 20 ms — too coarse a start line for a region of ~0.5 s)
        st.set("go", repr(time.monotonic() + 0.2).encode())
    t_go = float(st.get("go", 1800.0))
    while time.monotonic() < t_go:
        pass
    dt_own, t0, t1, ev_ms = rb.timed()
    st.set(f"t.{rank}", json.dumps([t0, t1, _lib.device_pci(device)]).encode())
    out = None
    if rank == 0:
        ts = [json.loads(v) for v in st.gather("t", world, 1800.0)]
        dt = max(t[1] for t in ts) - min(t[0] for t in ts)
        pci = [int(t[2]) for t in ts]
        n_distinct = len(set(pci))
        out = base_line(a, world, K, W, dt, rb.n_fl,
                        f"sample-sharded x{world} (one process per GPU), {rb.n_fl} samples in flight per GPU")
        out["multi_gpu_path"] = "replicas"
        out["ranks"] = world
        out["n_gpus"] = n_distinct  # distinct physical devices (ranks may share one: --share-gpu, LOCAL_RANK beyond the visible set)
        out["shared_devices"] = n_distinct < world
        out["per_rank_seconds"] = [t[1] - t[0] for t in ts]
        out["devices_pci"] = ["%04x:%02x:%02x" % (v >> 16, (v >> 8) & 0xff, v & 0xff) for v in pci]
        out["rccl_ranks"] = 0
        out["event_ms_longest_context"] = ev_ms
        out.update(device_record(rb.eng, a, rb.lml))
        st.set("solo_done", b"1")
    else:
        st.get("solo_done", timeout=1800.0)  # rank 0's stage timings have the GPUs' host to themselves
    rb.close()
    return out


def multi_rank(a, env):
    """N > 1: one process per GPU.  Two legs (module docstring): replicas, then the library's collective under a time limit."""
    import threading

    from bench_inputs import synthetic_problem, synthetic_theta_samples
    from gpax_amd import _lib, launch

    world, rank = env.world, env.rank
    root = rank == 0
    if world != a.gpus and root:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}", file=sys.stderr)
    # LOCAL_RANK names the GPU; a launcher that narrows HIP_VISIBLE_DEVICES per rank leaves one visible device (ordinal 0)
    n_vis = max(1, _lib.visible_device_count())
    device = 0 if a.share_gpu else (env.local_rank if env.local_rank < n_vis else env.local_rank % n_vis)
    transport = a.transport or ("file" if a.share_gpu else None)
    K, W = a.steps, a.warmup

    # ---- leg 1: replicas.  (A process that re-executes itself for the file transport — launch.init_rank — finds the leg
    # done: its line travels in the environment.) --------------------------------------------------------------------------
    carried = os.environ.get("GPX_BENCH_REPLICAS_LINE")
    if carried is None:
        rep = replicas_leg(a, env, device)
        os.environ["GPX_BENCH_REPLICAS_LINE"] = json.dumps(rep) if root else "-"
    else:
        rep = json.loads(carried) if root else None

    def report_replicas_only(why):
        """The collective leg failed or ran out of time: rank 0 prints the replicas line, every rank leaves (exit code 0:
        the run HAS its measurement; `collective_leg` says what happened to the other one)."""
        if root:
            rep["collective_leg"] = {"completed": False, "reason": why}
            flush_c_stdio()
            print(json.dumps(rep), flush=True)
        sys.stderr.write(f"[bench.py] rank {rank}: collective leg abandoned ({why})\n")
        sys.stderr.flush()
        os._exit(0)

    limit = a.collective_timeout
    watchdog = threading.Timer(limit, report_replicas_only, args=(f"not finished within {limit:.0f} s",))
    watchdog.daemon = True
    watchdog.start()
    try:
        out, rk = collective_leg(a, env, device, transport, rep, on_hang=report_replicas_only)
    except BaseException as ex:  # a rank that lost its peers raises (gpx_rank_*: peer death, time-outs of the transport)
        if isinstance(ex, (KeyboardInterrupt, SystemExit)):
            raise
        report_replicas_only(f"{type(ex).__name__}: {ex}"[:500])
    watchdog.cancel()
    store = bench_store(env, "bench")
    if root:
        if not a.no_node_record:
            # the same sweeps under the other launch model (one process, all GPUs), from a child process with its own time
            # limit; the ranks idle meanwhile — on the host (they poll the rendezvous store below), not inside an RCCL
            # collective that would spin on their GPUs
            out["node_sweep"] = run_node_record(a)
            ns = out["node_sweep"]
            if "c3_posteriors_per_s" in ns:
                ns["c3_vs_rank_collective"] = ns["c3_posteriors_per_s"] / out["collective"]["value"]
        flush_c_stdio()
        print(json.dumps(out), flush=True)
        store.set("solo_done", b"1")  # rank 0's solo phase is over: everybody meets again
    else:
        store.get("solo_done", timeout=1800.0)
    launch.finalize(env, rk)


def collective_leg(a, env, device, transport, rep, on_hang=None):
    """N > 1, SECOND leg: the library's own communicator (gpx_rank_*: RCCL, or the file transport when RCCL cannot
    initialise) and ONE collective gpx_rank_predict_sweep over S = N * K theta samples as the timed region.  Returns rank
    0's final line: the replicas leg's, with this leg's rate under `collective`."""
    from bench_inputs import synthetic_problem, synthetic_theta_samples
    from gpax_amd import _lib, launch

    world, rank = env.world, env.rank
    root = rank == 0
    K, W = a.steps, a.warmup
    n_fl = max(1, min(a.inflight, max(1, K // 2)))
    if os.environ.get("GPX_BENCH_TEST_HANG") == str(rank):  # tests: this rank never joins (a hung collective)
        time.sleep(86400)
    rk = launch.init_rank(env, device=device, inflight=n_fl, transport=transport, timeout=a.init_timeout,
                          reexec_on_hang=True, on_hang=on_hang)
    info = rk.info()
    kind = _lib.kernel_kind(a.kernel)
    N, d, M = a.N, a.d, a.M
    arrays = {}
    if root:  # only rank 0 holds inputs; the sweep broadcasts them
        X, y, Xnew, p = synthetic_problem(N, d, M, seed=0)
        th = synthetic_theta_samples(world * (K + W), d, seed=1)
        eps = np.random.default_rng(2).standard_normal((world * (K + W), 1, M))
        arrays = dict(X=X, yres=y, Xnew=Xnew)

    def sweep(lo, hi):
        kw = dict(arrays)
        if root:
            kw.update(ells=th["k_length"][lo:hi], scales=th["k_scale"][lo:hi], noises=th["noise"][lo:hi], eps=eps[lo:hi])
        return rk.predict_sweep(kind, N, d, hi - lo, M, 1, False, 1e-6, **kw)

    if W > 0:
        sweep(0, world * W)
    flush_c_stdio()  # every rank's RCCL banner out now, not behind rank 0's JSON line
    rk.barrier()
    t0 = time.perf_counter()
    res = sweep(world * W, world * (W + K))
    dt_own = time.perf_counter() - t0  # this rank's view: its block + its part in broadcast / gather
    rk.barrier()
    dt = float(rk.allreduce_max([time.perf_counter() - t0])[0])
    # per-rank seconds and the GPU every rank drives, so that ONE JSON line explains a partial failure or ranks that
    # share a device (one slot per rank, the other slots at -inf, max over ranks: an all-gather of small vectors)
    slots = min(world, 32)
    vec = np.full(2 * slots, -np.inf)
    if rank < slots:
        vec[rank] = dt_own
        vec[slots + rank] = float(rk.device_pci())
    vec = rk.allreduce_max(vec)
    per_rank_s = [float(v) for v in vec[:slots]]
    pci = [int(v) for v in vec[slots:2 * slots] if np.isfinite(v)]
    n_distinct = len(set(pci))
    # ---- second record: C4 (BASELINE.json configs[3]) through the same collective, and on rank 0 alone -------------------
    c4 = None
    if a.c4_S > 0:
        N4, d4, M4, S4 = a.c4_N, 3, 1024, a.c4_S
        arr4 = {}
        if root:
            X4, y4, Xn4, _ = synthetic_problem(N4, d4, M4, seed=0)
            th4 = synthetic_theta_samples(S4, d4, seed=1)
            eps4 = np.random.default_rng(2).standard_normal((S4, 1, M4))
            arr4 = dict(X=X4, yres=y4, Xnew=Xn4, ells=th4["k_length"], scales=th4["k_scale"], noises=th4["noise"], eps=eps4)
        warm = min(S4, 24 * world)  # >= one full batch (B = 7 at N = 8192) per context in flight
        wk = {k: (v[:warm] if k in ("ells", "scales", "noises", "eps") else v) for k, v in arr4.items()}
        rk.predict_sweep(kind, N4, d4, warm, M4, 1, False, 1e-6, **wk)  # allocations at this shape
        rk.barrier()
        t0 = time.perf_counter()
        res4 = rk.predict_sweep(kind, N4, d4, S4, M4, 1, False, 1e-6, **arr4)
        rk.barrier()
        dt4 = float(rk.allreduce_max([time.perf_counter() - t0])[0])
        dt4_one = None
        if root:  # the same sweep on rank 0's GPU alone: the single-GPU product path (ExactGP.predict's engine pool)
            engines = [_lib.Engine(device) for _ in range(max(1, a.inflight))]
            sl = slice(0, min(S4, 64))
            _lib.concurrent_sweep(engines, X4, kind, th4["k_length"][sl], th4["k_scale"][sl], th4["noise"][sl], y4, Xn4,
                                  False, 1e-6, eps4[sl])
            t0 = time.perf_counter()
            one = _lib.concurrent_sweep(engines, X4, kind, th4["k_length"], th4["k_scale"], th4["noise"], y4, Xn4, False,
                                        1e-6, eps4)
            dt4_one = time.perf_counter() - t0
            same = bool(np.array_equal(one[0], res4[0]) and np.array_equal(one[1], res4[1]))
            for e in engines:
                e.close()
            flop4 = N4 ** 3 / 3 + N4 * N4 * M4 + N4 * M4 * M4 + 2 * N4 * N4 + 2 * N4 * M4 + M4 ** 3 / 3 + M4 * M4
            c4 = {"config": f"C4: {S4}-sample predictive sweep N={N4} d={d4} M={M4} n=1 (BASELINE.json configs[3]), host "
                            "arrays in, host arrays out",
                  "S": S4, "seconds": dt4, "posteriors_per_s": S4 / dt4, "rccl_ranks": world if info["transport"] == "rccl" else 0,
                  "ranks": world, "seconds_one_gpu": dt4_one, "posteriors_per_s_one_gpu": S4 / dt4_one,
                  "speedup_vs_1": dt4_one / dt4, "tflops": S4 * flop4 / dt4 / 1e12,
                  "frac_of_fp64_peak_per_gpu": S4 * flop4 / dt4 / 1e12 / FP64_MFMA_PEAK_TFLOPS / world,
                  "identical_to_one_gpu": same, "nan_rows": int(np.isnan(res4[1]).any(axis=(1, 2)).sum())}
        rk.barrier()

    if not root:
        return None, rk
    # `value` stays the replicas leg's: inputs resident in HBM when the timed region starts, the N = 1 workload on every
    # GPU (the contract of this file's docstring).  The collective — host arrays in on rank 0, host arrays out — is the
    # PCIe- and xGMI-inclusive rate of the product's multi-GPU call, reported beside it.
    out = dict(rep)
    out["multi_gpu_path"] = "rank-" + info["transport"]
    out["rccl_version"] = info["rccl_version"]
    out["rccl_ranks"] = world if info["transport"] == "rccl" else 0
    out["collective_leg"] = {"completed": True}
    out["collective"] = {
        "what": ("gpx_rank_predict_sweep over S = ranks x steps theta samples: H2D on rank 0, ncclBroadcast, per-rank block, "
                 "ncclSend/ncclRecv gather, D2H on rank 0 — all inside the timed region; barrier = gpx_rank_barrier, time = "
                 "max over ranks (gpx_rank_allreduce_max)"),
        "value": world * K / dt, "unit": "posteriors/s", "ms_per_step": dt / K * 1e3, "per_rank_seconds": per_rank_s,
        "vs_replicas": (world * K / dt) / rep["value"], "transport": info["transport"], "n_gpus": n_distinct,
        "devices_pci": ["%04x:%02x:%02x" % (v >> 16, (v >> 8) & 0xff, v & 0xff) for v in pci],
        "nan_rows": int(np.isnan(res[1]).any(axis=(1, 2)).sum()),
        # the ranks' sample blocks are sized by a probe of every GPU (gpx_rank_calibrate, launch.init_rank): relative speeds
        "rank_speeds": [float(v) for v in getattr(rk, "speeds", [])] or None}
    if c4 is not None:
        out["c4_sweep"] = c4
    return out, rk


def main():
    a = parse()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL across processes needs here
    from gpax_amd import launch

    if a.node_record:
        node_record(a)
        return
    env = launch.rank_env()
    if env is None and a.gpus > 1:
        # started plainly: launch the N ranks ourselves (one process per GPU; no torch anywhere)
        argv = [x for x in sys.argv[1:] if x != "--dry-launch"]
        if a.dry_launch:
            print(json.dumps({"launch": launch.spawn_command(__file__, argv, a.gpus)}))
            return
        sys.exit(launch.spawn_ranks(__file__, argv, a.gpus))
    if a.dry_launch:
        print(json.dumps({"launch": None}))
        return
    if a.force_rank_path and env is None:
        import tempfile
        env = launch.RankEnv(0, 1, 0, tempfile.mkdtemp(prefix="gpx_rdzv_"), 0, True)
    if env is None or (env.world == 1 and not a.force_rank_path):
        single_gpu(a, 0 if (env is None or a.share_gpu) else env.local_rank)
    else:
        multi_rank(a, env)


if __name__ == "__main__":
    main()
