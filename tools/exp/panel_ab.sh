#!/bin/bash
# A/B of the cooperative panel-chain kernel (GPX_PANEL_KERNEL = 0 launches / 1 tail / 2 everywhere): stage timings of
# bench.py at C3, one process per variant, ABAB.  Usage (GPU box): bash tools/exp/panel_ab.sh [outdir]
O=${1:-gpurun_out/panel_ab}
mkdir -p $O
for rep in 1 2; do
  for v in 0 1 2; do
    GPX_PANEL_KERNEL=$v timeout 300 python bench.py --no-cpu-baseline --steps 9 --warmup 3 > $O/bench_pk${v}_$rep.json 2> $O/bench_pk${v}_$rep.err
    python - $O/bench_pk${v}_$rep.json $v <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
s = d["stages"]
print(f"panel={sys.argv[2]} potrf {s['potrf_ms']:.2f} predict {s['predict_ms']:.2f} fit {s['fit_step_ms']:.2f} post/s {d['value']:.2f} "
      f"frac {d['roofline']['frac']:.3f} classes {d['kernel_classes_ms_per_predict']}")
PY
  done
done
