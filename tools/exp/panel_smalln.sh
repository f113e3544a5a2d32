#!/bin/bash
# cooperative panel kernel at small N (few rows below every block): stage timings, variant per process
O=${1:-gpurun_out/panel_smalln}
mkdir -p $O
for N in 1024 2048 3072 4096 6144; do
  for v in "0 16" "2 16" "2 32" "2 8"; do
    set -- $v
    GPX_PANEL_KERNEL=$1 GPX_PANEL_MAX_FAR=$2 timeout 300 python bench.py --N $N --M 256 --no-cpu-baseline --steps 6 --warmup 2 --inflight 1 > $O/b_${N}_$1_$2.json 2> $O/b_${N}_$1_$2.err
    python - $O/b_${N}_$1_$2.json $N "$v" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
s = d["stages"]
print(f"N={sys.argv[2]} panel,maxfar={sys.argv[3]}: potrf {s['potrf_ms']:.3f} predict {s['predict_ms']:.3f} fit {s['fit_step_ms']:.3f} ms")
PY
  done
done
for v in "0 16" "1 16" "1 8" "1 24"; do
  set -- $v
  GPX_PANEL_KERNEL=$1 GPX_PANEL_MAX_FAR=$2 timeout 300 python bench.py --no-cpu-baseline --steps 9 --warmup 3 > $O/c3_$1_$2.json 2> $O/c3_$1_$2.err
  python - $O/c3_$1_$2.json "$v" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
s = d["stages"]
print(f"C3 panel,maxfar={sys.argv[2]}: potrf {s['potrf_ms']:.2f} predict {s['predict_ms']:.2f} fit {s['fit_step_ms']:.2f} post/s {d['value']:.2f}")
PY
done
for v in 0 2; do
  GPX_PANEL_KERNEL=$v GPX_PANEL_MAX_FAR=16 timeout 300 python tools/c5_bench.py 2> $O/c5_$v.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C5 sparse panel=$v: bound', round(d['bound']['ms'],2), 'bound+grad', round(d['bound_and_gradient']['ms'],2), 'ms')"
done
