cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_edges.py tests/test_gpu_exactgp.py tests/test_gpu_vgp.py tests/test_gpu_models.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py tests/test_gpu_mngp.py tests/test_gpu_hskgp.py -x -q -m gpu > $O/tests_q.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|Error" $O/tests_q.log | tail -3
S=8192 CTX=1 timeout 600 python tools/small_n_sweep.py 2>&1 | tail -5
S=4096 CTX=1 SIZES=128,1,100 timeout 600 python tools/small_n_sweep.py 2>&1 | tail -1
