#!/bin/bash
# A/B of two builds of libgpx on ONE box: alternating bench runs (no CPU leg, no configs block) and one kernel-trace pass
# per library with a single theta in flight (the in-pipeline average duration of every kernel).
#   bash tools/ab_lib.sh <out-dir> <other-lib.so> [rounds]
# The other library is loaded through GPX_LIB (gpax_amd/_lib.py); same ABI required.
O=${1:-gpurun_out/ab}; OTHER=${2:-gpax_amd/lib/libgpx_r05.so}; R=${3:-3}
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p $O
B="python bench.py --no-cpu-baseline --no-configs"
for i in $(seq 1 $R); do
  $B > $O/new_$i.json 2> $O/new_$i.err
  GPX_LIB=$OTHER $B > $O/old_$i.json 2> $O/old_$i.err
done
for which in new old; do
  rm -rf /tmp/prof_$which
  if [ $which = old ]; then export GPX_LIB=$OTHER; else unset GPX_LIB; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$which -- $B --inflight 1 > $O/under_trace1_$which.json 2> $O/trace1_$which.err
  db=$(find /tmp/prof_$which -name '*.db' | head -1)
  echo "## rocprofv3 --kernel-trace --stats -- $B --inflight 1   ($which library)" > $O/trace1_$which.md
  python tools/rocpd_summary.py "$db" $O/trace1_$which.md > /dev/null 2>> $O/trace1_$which.err
done
unset GPX_LIB
python - $O $R <<'PY'
import json, sys
o, r = sys.argv[1], int(sys.argv[2])
for w in ("new", "old"):
    for i in range(1, r + 1):
        try:
            p = json.loads(open(f"{o}/{w}_{i}.json").read().strip().splitlines()[-1])
            kc = p.get("kernel_classes_ms_per_predict", {})
            print(w, i, "value %.2f ms/step %.2f trailing_frac %.3f potrf_tflops %s gemm_other %s" % (
                p["value"], p["ms_per_step"], p["roofline"]["frac"], p.get("potrf_tflops"), kc.get("gemm_other")))
        except Exception as e:
            print(w, i, "failed", e)
PY
