cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
bash tools/exp/r05_f.sh 2>&1 | grep -v "^   grad\|^   scale" 
timeout 600 python tools/fit_small_bench.py > $O/fit_small.json 2> $O/fit_small.log; cat $O/fit_small.log
timeout 900 python -m pytest tests/test_gpu_exactgp.py tests/test_gpu_models.py tests/test_gpu_vgp.py tests/test_gpu_periodic.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py tests/test_gpu_edges.py tests/test_gpu_reference_notebook.py -x -q -m gpu > $O/tests_h.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|Error" $O/tests_h.log | tail -3
timeout 300 python bench_configs.py C1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['C1'])"
