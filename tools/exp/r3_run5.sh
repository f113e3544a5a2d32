cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e
mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -12 $O/gpu_tests.log
