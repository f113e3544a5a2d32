cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/exp/fit_host.py
rm -rf /tmp/fh; NS=2048 R=100 timeout 300 rocprofv3 --hip-trace --stats -d /tmp/fh -- python tools/exp/fit_host.py > /tmp/fh.log 2>&1; tail -2 /tmp/fh.log
f=$(find /tmp/fh -name '*hip_api_stats.csv' | head -1); echo $f; head -25 "$f" | cut -c1-160
ls /tmp/fh/* | head
