# the round-end checks of the driver, run by hand: the whole GPU suite, the smoke entry, the default bench line
cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/full}
mkdir -p $O
timeout 2700 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; cut -c1-300 $O/bench_n1.json
