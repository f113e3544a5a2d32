cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py tests/test_gpu_fit_small.py tests/test_gpu_periodic.py tests/test_gpu_exactgp.py tests/test_gpu_sparse.py tests/test_gpu_edges.py -x -q -m gpu > $O/tests_w.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|Error" $O/tests_w.log | tail -3
for i in 1 2; do timeout 300 python tools/gram_bench.py 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('gram median %.4f best %.4f ms written %.0f GB/s best %.0f  fill %.0f  frac %.3f' % (d['median_ms'], d['best_ms'], d['written_GBps'], d['best_written_GBps'], d['write_only_fill_GBps'], d['frac_of_write_only_fill']))"; done
timeout 300 python bench.py --no-cpu-baseline --no-configs --steps 18 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['stages'], r['gram_written_GBps'], r['gram_write_only_fill_GBps'], r['lml_check'])"
