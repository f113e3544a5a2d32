cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c
mkdir -p $O
GPX_DEBUG=1 timeout 600 python -m pytest tests/test_gpu_edges.py -x -q -m gpu -k "panel" -s > $O/tests_panel.log 2>&1; echo "panel tests rc=$?"; grep -E "gpx\]|passed|failed|Error|assert" $O/tests_panel.log | tail -15
if grep -q "1 passed" $O/tests_panel.log; then
  timeout 600 python -m pytest tests/test_gpu_edges.py -x -q -m gpu -k "lazy" > $O/tests_lazy.log 2>&1; echo "lazy rc=$?"; tail -3 $O/tests_lazy.log
  bash tools/exp/panel_ab.sh $O 2>&1 | tee $O/panel_ab.txt
fi
