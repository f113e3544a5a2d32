cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
bash tools/exp/r05_f.sh 2>&1 | grep "bitwise\|host fit\|lml fused" 
timeout 600 python tools/fit_small_bench.py > $O/fit_small.json 2> $O/fit_small.log; cat $O/fit_small.log
timeout 600 python tools/fit_predict_wallclock.py C1 2>&1 | tail -3
