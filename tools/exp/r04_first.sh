#!/bin/bash
# round 4, first GPU call: the slim potf2 kernel — bit-identity, stand-alone time, in-pipeline A/B (one process per variant)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04a
O=gpurun_out/r04a
timeout 600 python -m pytest tests/test_gpu_edges.py -x -q -k "three_diagonal or lazy_far" > $O/edges.log 2>&1; echo "edges rc=$?" | tee -a $O/summary.txt
for m in slim chain tile; do GPX_POTF2=$m timeout 120 python tools/potf2_time.py >> $O/potf2_time.txt 2>&1; done
for rep in 1 2; do for m in slim chain; do
  GPX_POTF2=$m timeout 300 python bench.py --no-cpu-baseline --steps 24 > $O/bench_${m}_$rep.json 2> $O/bench_${m}_$rep.err
done; done
tail -3 $O/edges.log; cat $O/potf2_time.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04a/bench_*.json')):
    try:
        j=json.loads(open(f).read().strip().split('\n')[-1])
        st=j.get('stages',{})
        print(f, j['value'], {k:st.get(k) for k in ('potrf_ms','predict_ms','fit_step_ms')}, j.get('kernel_classes_ms_per_predict'))
    except Exception as e: print(f, 'ERR', e)
PY
