# XCD-aware tile order of the trailing update (GPX_TILE_ORDER=R[,b]) against the plain grid: bench timing, then FETCH_SIZE / WRITE_SIZE
# and SQ counters of the trailing kernel under separate --pmc passes (one theta, every dispatch alone)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05/tile_order; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-configs"
for ord in plain 2,2 4,2 2,4 1,2 8,2 2,1; do
  if [ $ord = plain ]; then unset GPX_TILE_ORDER; else export GPX_TILE_ORDER=$ord; fi
  timeout 300 $B --steps 18 > $O/bench_$ord.json 2>/dev/null
  python - $O/bench_$ord.json $ord <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("order %-6s value %.2f  potrf %.2f predict %.2f  trailing avg launch %.1f us frac %.3f  classes %s" % (sys.argv[2], r["value"], r["stages"]["potrf_ms"], r["stages"]["predict_ms"], r["roofline"]["avg_launch_ms"] * 1e3, r["roofline"]["frac"], {k: round(v, 2) for k, v in r["kernel_classes_ms_per_predict"].items()}))
PY
  for pmc in FETCH_SIZE WRITE_SIZE "SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $pmc | cut -d' ' -f1)
    rm -rf /tmp/to_p
    timeout 300 rocprofv3 --kernel-trace --pmc $pmc -d /tmp/to_p -- $B --steps 1 --warmup 0 > /dev/null 2> $O/pmc_${ord}_$tag.err
    db=$(find /tmp/to_p -name '*.db' | head -1)
    python - $db $ord "$pmc" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%gemm_nt128_kernel<1,%' or kernel_name like '%gemm_nt128_xcd_kernel%' group by counter_name").fetchall()
for cn, n, avg, dur in rows:
    print("   order %-6s %-26s dispatches %d avg per launch %.6g  avg duration alone %.1f us" % (sys.argv[2], cn, n, avg, dur / 1e3))
PY
  done
done
