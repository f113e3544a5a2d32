cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py tests/test_gpu_edges.py -x -q -m gpu > $O/tests_m.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|Error" $O/tests_m.log | tail -3
timeout 300 python tools/gram_bench.py > $O/gram.json 2> $O/gram.err; cat $O/gram.json | cut -c1-900
