cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
O=gpurun_out/kinv; mkdir -p $O
for N in 1024 3072 4096 6144; do
  rm -rf /tmp/kp$N
  timeout 200 rocprofv3 --kernel-trace -d /tmp/kp$N -- python tools/smalln_timeline.py run $N > $O/run_$N.txt 2>&1
  python tools/smalln_timeline.py show "$(find /tmp/kp$N -name '*.db' | head -1)" $O/tl_$N.md > /dev/null 2>> $O/run_$N.txt
  head -1 $O/tl_$N.md; tail -8 $O/tl_$N.md | cut -c1-110
done
