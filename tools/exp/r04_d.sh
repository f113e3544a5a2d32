#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04d; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_parity_fullsize.py tests/test_gpu_sparse.py tests/test_gpu_rank.py tests/test_bench_launch.py tests/test_gpu_edges.py -x -q --durations=12 -k "c3_gradient or vector_valued or rank or bench or three_diag or c4_shape or lazy_far" > $O/tests.log 2>&1; echo "tests rc=$?"
tail -22 $O/tests.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r04d/bench.json').read().strip().split('\n')[-1])
print('value', j['value'], 'stages', j['stages'])
print('potf2', json.dumps(j['potf2'], indent=0))
print('roofline', {k:j['roofline'][k] for k in ('achieved','frac','avg_launch_ms','serialised_frac')})
print('classes', j['kernel_classes_ms_per_predict'])
cb=j['cpu_baseline']; print('cpu', cb['value'], cb['seconds'], cb['median_of_3'])
PY
timeout 600 python tools/c5_bench.py > $O/c5_sparse.json 2> $O/c5.err; echo "c5 rc=$?"
python - <<'PY'
import json
j=json.load(open('gpurun_out/r04d/c5_sparse.json'))
print({k:(j[k]['ms'], j[k].get('frac_of_fp64_peak')) for k in ('bound','bound_and_gradient','posterior_all_pixels')}, j['viSparseGP_api']['ms_per_svi_step'], j['viGP_exact_api']['ms_per_svi_step'])
PY
