cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fit_small.py tests/test_gpu_vgp.py tests/test_gpu_exactgp.py -x -q -m gpu > $O/tests_l.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|Error|assert" $O/tests_l.log | tail -5
timeout 600 python tools/fit_small_bench.py > $O/fit_small.json 2> $O/fit_small.log; cat $O/fit_small.log
timeout 300 python bench_configs.py C1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['C1']['host_api_fit_step'], d['C1']['fit_s'])"
