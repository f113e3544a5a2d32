# the round-end checks of the driver, run by hand: the whole GPU suite, the smoke entry
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 2700 python -m pytest tests -x -q -m gpu --durations=25 > $O/gputest_round5.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gputest_round5.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
