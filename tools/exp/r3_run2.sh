cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_edges.py -x -q -m gpu -k "panel or lazy" > $O/tests_panel.log 2>&1; echo "panel tests rc=$?"; tail -15 $O/tests_panel.log
bash tools/exp/panel_ab.sh $O 2>&1 | tee $O/panel_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity_fullsize.py -x -q -m gpu -k "c5_visparse" -s > $O/tests_c5.log 2>&1; echo "c5 tests rc=$?"; grep -E "C5 sparse|passed|failed|Error" $O/tests_c5.log | tail -12
timeout 600 python tools/c5_bench.py > $O/c5_sparse.json 2> $O/c5_sparse.err; echo "c5 bench rc=$?"; cut -c1-1500 $O/c5_sparse.json
