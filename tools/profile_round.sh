#!/bin/bash
# Regenerate the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r06
#   kernel-trace + stats of the default bench command, separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_*), the C4
#   sweep (S = 1000) with its own kernel trace, and the round's bench line; summarised on the box (the rocpd sqlite
#   databases stay in /tmp; only .md / .json summaries come back under gpurun_out/prof — copy them to profiles/<round>).
# NOTE (round 1): a single pass with five TCC_* derived counters on `bench.py --steps 1` did not finish within 10
# minutes on this pool — keep L2 counters out of this script.  --pmc passes carry --kernel-trace only (gpurun refuses
# --pmc together with the hip / hsa / memory trace domains).
RND=${1:-r06}
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/prof
rm -rf $O && mkdir -p $O
run() {
  name=$1; cmd=$2; shift 2
  rm -rf /tmp/prof_$name
  timeout 600 rocprofv3 "$@" -d /tmp/prof_$name -- $cmd > $O/under_$name.json 2> $O/$name.err
  db=$(find /tmp/prof_$name -name '*.db' | head -1)
  echo "## rocprofv3 $name pass: rocprofv3 $* -- $cmd" > $O/$name.md
  python tools/rocpd_summary.py "$db" $O/$name.md $O/traffic.json > /dev/null 2>> $O/$name.err
  tail -c 300 $O/under_$name.json | head -c 300; echo
}
B="python bench.py --no-cpu-baseline --no-configs"  # (the configs block — C1, C2, C4, C5 — is part of the final bench line below)
run trace "$B" --kernel-trace --stats
# one theta in flight: the dominant kernel's average duration here is the one bench.py's roofline block measures
# (its profile pass runs on a single context); with 3 contexts in flight (pass above) concurrent kernels stretch
run trace1 "$B --inflight 1" --kernel-trace --stats
run fetch "$B --steps 1 --warmup 0" --kernel-trace --pmc FETCH_SIZE
run write "$B --steps 1 --warmup 0" --kernel-trace --pmc WRITE_SIZE
run sq "$B --steps 1 --warmup 0" --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
if [ -n "$PROFILE_CORE_ONLY" ]; then
  # the passes bench.py's roofline block reads (trace, fetch, write, sq) + the mid-N timelines + the bench line: what has to be
  # re-measured when only the launch geometry of the GEMM kernels changed
  for N in 2048 4096; do
    rm -rf /tmp/prof_sn$N
    timeout 200 rocprofv3 --kernel-trace -d /tmp/prof_sn$N -- python tools/smalln_timeline.py run $N > $O/smalln_$N.txt 2>&1
    python tools/smalln_timeline.py show "$(find /tmp/prof_sn$N -name '*.db' | head -1)" $O/smalln_timeline_$N.md > /dev/null 2>> $O/smalln_$N.txt
  done
  bash tools/exp/sgp_trace2.sh $O/sgptrace > /dev/null 2>&1 && cp $O/sgptrace/c5_step_timeline.md $O/c5_step_timeline.md
  python bench.py > $O/bench_$RND.json 2> $O/bench_$RND.err
  cut -c1-400 $O/bench_$RND.json
  ls $O
  exit 0
fi
# C4: the 1000-sample sweep at N = 8192, d = 3 through ExactGP.predict, plain and under the kernel trace
python tools/c4_sweep.py > $O/c4_sweep.json 2> $O/c4_sweep.err
run c4trace "python tools/c4_sweep.py" --kernel-trace --stats
# fit + predict wall-clock through the model API (BASELINE metric, first half), the roctx marker trace, the ASan smoke
rm -f $O/fit_predict_wallclock.json
python tools/fit_predict_wallclock.py C2 C3 > $O/fit_predict_wallclock.log 2>&1
python tools/fit_predict_wallclock.py C1 >> $O/fit_predict_wallclock.log 2>&1
rm -rf /tmp/prof_roctx
GPX_ROCTX=1 timeout 300 rocprofv3 --marker-trace --kernel-trace --stats -d /tmp/prof_roctx -- python bench.py --no-cpu-baseline --steps 2 --warmup 1 --inflight 1 > $O/under_roctx.json 2> $O/roctx.err
db=$(find /tmp/prof_roctx -name '*.db' | head -1)
python - "$db" > $O/roctx.md 2>> $O/roctx.err <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
print("## roctx ranges (GPX_ROCTX=1) under rocprofv3 --marker-trace --kernel-trace: bench.py --steps 2 --warmup 1 --inflight 1\n")
names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
if "regions" not in names:
    print("(no `regions` view in this rocpd schema)")
else:
    # ROCm 7.2 rocpd: roctx ranges are rows of `regions` named roctxThreadRangeA, the message sits in extdata (JSON)
    rows = cur.execute("select json_extract(extdata, '$.message') as msg, count(*), sum(duration), avg(duration) from regions "
                       "where category like 'MARKER%' group by msg order by sum(duration) desc").fetchall()
    print("| range | count | total ms (host side: the time the stage's launches take to enqueue) | avg us |\n|---|---|---|---|")
    for n, c, s, a in rows:
        print(f"| `{n}` | {c} | {s / 1e6:.3f} | {a / 1e3:.1f} |")
PY
# C5: viSparseGP / viGP on the 512 x 512 image (flop model in the tool's docstring), and the launch timeline of one sparse step
python tools/c5_bench.py > $O/c5_sparse.json 2> $O/c5_sparse.err
bash tools/exp/sgp_trace2.sh $O/sgptrace > /dev/null 2>&1 && cp $O/sgptrace/c5_step_timeline.md $O/c5_step_timeline.md
# the N > 1 code path of bench.py on this 1-GPU box: one rank over RCCL (the collective sweep incl. the C4 record), and two
# ranks sharing the GPU over the file transport (control flow only: the two ranks halve the GPU between them)
python bench.py --force-rank-path --steps 12 --warmup 3 > $O/bench_rank1_rccl.json 2> $O/bench_rank1_rccl.err
python bench.py --gpus 2 --share-gpu --steps 8 --warmup 2 --c4-S 200 > $O/bench_2ranks_shared_gpu.json 2> $O/bench_2ranks_shared_gpu.err
# round 4: the diagonal-block kernel inside the pipeline — per dispatch wait / execution from the kernel trace (default
# kernel and the round-3 one), the in-kernel phase trace of the default kernel (trace build of the library), the Gram
# build alone with its HBM counters, the fit step at the sizes gpax is mostly used at, and the RCCL calls one rank issues
for m in slim; do
  rm -rf /tmp/prof_pw_$m
  GPX_POTF2=$m timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_pw_$m -- $B --steps 6 --warmup 1 --inflight 1 > /dev/null 2> $O/pw_$m.err
  echo "## GPX_POTF2=$m: rocprofv3 --kernel-trace -- $B --steps 6 --warmup 1 --inflight 1" >> $O/potf2_wait.md
  python tools/potf2_wait.py "$(find /tmp/prof_pw_$m -name '*.db' | head -1)" $O/potf2_wait.md > /dev/null 2>> $O/pw_$m.err
done
if [ -f gpax_amd/lib/libgpx_trace.so ]; then
  GPX_LIB=gpax_amd/lib/libgpx_trace.so timeout 300 python tools/potf2_trace.py > $O/potf2_phase_trace.json 2> $O/potf2_phase_trace.err
fi
python tools/gram_bench.py > $O/gram.json 2> $O/gram.err
run gram_write "python tools/gram_bench.py" --kernel-trace --pmc WRITE_SIZE
run gram_fetch "python tools/gram_bench.py" --kernel-trace --pmc FETCH_SIZE
python tools/fit_small_bench.py > $O/fit_small.json 2> $O/fit_small.log
# round 6: the fused chain step (potf2 + panel TRSM in one launch) against the three-launch step, its per-phase stamps (trace
# build), and NUTS through the model API with the transition loop in Python / in the library
python tools/potf2_trsm_ab.py > $O/potf2_trsm_ab.txt 2>&1
if [ -f gpax_amd/lib/libgpx_trace.so ]; then
  GPX_LIB=gpax_amd/lib/libgpx_trace.so timeout 300 python tools/potf2_trsm_trace.py 2048 4096 5120 > $O/potf2_trsm_trace.txt 2>&1
fi
python tools/nuts_native_time.py > $O/nuts_native_time.txt 2>&1
python tools/lat_gemm_bench.py > $O/lat_gemm.json 2> $O/lat_gemm.txt
for N in 25 128 512 2048 4096; do
  rm -rf /tmp/prof_sn$N
  timeout 200 rocprofv3 --kernel-trace -d /tmp/prof_sn$N -- python tools/smalln_timeline.py run $N > $O/smalln_$N.txt 2>&1
  python tools/smalln_timeline.py show "$(find /tmp/prof_sn$N -name '*.db' | head -1)" $O/smalln_timeline_$N.md > /dev/null 2>> $O/smalln_$N.txt
done
GPX_RANK_FORCE_COLLECTIVES=1 NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,COLL,P2P NCCL_DEBUG_FILE=$O/rank1_rccl_nccl_debug.log python bench.py --force-rank-path --steps 4 --warmup 1 --c4-S 24 --no-node-record > $O/bench_rank1_rccl_forced_collectives.json 2> $O/rank1_rccl_forced.err
python bench.py > $O/bench_$RND.json 2> $O/bench_$RND.err
cut -c1-400 $O/bench_$RND.json
ls -la $O
