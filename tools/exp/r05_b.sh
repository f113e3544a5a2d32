cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
bash tools/exp/sgp_trace2.sh $O/sgptrace_ride
timeout 600 python tools/lat_gemm_bench.py > $O/lat_gemm.json 2> $O/lat_gemm.log; tail -14 $O/lat_gemm.log
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_edges.py tests/test_gpu_exactgp.py -x -q -m gpu > $O/tests_b.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/tests_b.log | tail -2
