#!/bin/bash
# Regenerate the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   kernel-trace + stats of the default bench command, and separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_*),
#   summarised on the box (the rocpd sqlite databases stay in /tmp; only .md / .json summaries come back).
# NOTE (round 1): a single pass with five TCC_* derived counters (TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
# TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum) on `bench.py --steps 1` did not finish within 10 minutes on this pool —
# keep L2 counters out of this script, or collect one per pass on a much shorter workload with its own timeout.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/prof
rm -rf $O && mkdir -p $O
run() {
  name=$1; args=$2; shift 2
  rm -rf /tmp/prof_$name
  rocprofv3 "$@" -d /tmp/prof_$name -- python bench.py --no-cpu-baseline $args > $O/bench_under_$name.json 2> $O/$name.err
  db=$(find /tmp/prof_$name -name '*.db' | head -1)
  echo "## rocprofv3 $name pass: rocprofv3 $* -- python bench.py --no-cpu-baseline $args" > $O/$name.md
  python tools/rocpd_summary.py "$db" $O/$name.md $O/traffic.json > /dev/null 2>> $O/$name.err
  tail -c 300 $O/bench_under_$name.json | head -c 300; echo
}
run trace "" --kernel-trace --stats
# one theta in flight: the dominant kernel's average duration here is the one bench.py's roofline block measures
# (its profile pass runs on a single context); with 3 contexts in flight (pass above) concurrent kernels stretch
run trace1 "--inflight 1" --kernel-trace --stats
run fetch "--steps 1 --warmup 0" --pmc FETCH_SIZE
run write "--steps 1 --warmup 0" --pmc WRITE_SIZE
run sq "--steps 1 --warmup 0" --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
python bench.py > $O/bench_r01.json 2> $O/bench_r01.err
cut -c1-400 $O/bench_r01.json
ls -la $O
