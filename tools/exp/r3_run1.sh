cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
O=gpurun_out/r03a
timeout 1500 python -m pytest tests/test_gpu_rank.py tests/test_gpu_node.py tests/test_bench_launch.py -x -q -m gpu > $O/tests_rank.log 2>&1; echo "tests rc=$?" >> $O/tests_rank.log
tail -5 $O/tests_rank.log
timeout 600 python bench.py --force-rank-path --steps 12 --warmup 3 > $O/bench_rank1_rccl.json 2> $O/bench_rank1_rccl.err; echo "rank1 rc=$?"
tail -c 1500 $O/bench_rank1_rccl.json; tail -5 $O/bench_rank1_rccl.err
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
cut -c1-600 $O/bench_default.json
