#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04i; mkdir -p $O; rm -f $O/*
timeout 600 python -m pytest tests/test_gpu_edges.py tests/test_gpu_golden.py tests/test_gpu_kernels.py tests/test_gpu_exactgp.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for m in slim chain tile; do GPX_POTF2=$m timeout 120 python tools/potf2_time.py 2>&1 | head -1; done
GPX_LIB=gpax_amd/lib/libgpx_trace.so timeout 300 python tools/potf2_trace.py > $O/slim_trace.json 2> $O/slim_trace.err
python - <<'PY'
import json
for g in json.load(open('gpurun_out/r04i/slim_trace.json')):
    print(g['group'][:60], 'in_kernel', round(g['in_kernel_us'],1), 'chain sums', g['chain_sum_us'], 'upd', g['worker_update_window_us_per_panel'][:3])
PY
for rep in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline > $O/bench_$rep.json 2> $O/bench_$rep.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04i/bench_*.json')):
    j=json.loads(open(f).read().strip().split('\n')[-1])
    print(f.split('/')[-1], round(j['value'],2), {k:round(v,2) for k,v in j['stages'].items()}, {k:round(v,1) for k,v in j['kernel_classes_ms_per_predict'].items()}, j['potf2']['in_pipeline_us'], j['potf2']['standalone_us'], j['potf2']['potrf_ms'])
PY
for N in 128 512 2048; do python tools/smalln_timeline.py run $N | tail -1; done
