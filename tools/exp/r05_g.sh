cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 600 python tools/fit_small_bench.py > $O/fit_small.json 2> $O/fit_small.log; cat $O/fit_small.log
