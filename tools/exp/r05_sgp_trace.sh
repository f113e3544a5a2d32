cd $GRAFT_REPO_ROOT
bash tools/exp/sgp_trace.sh gpurun_out/r05/sgptrace_ride
