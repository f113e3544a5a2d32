#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04b; mkdir -p $O; rm -f $O/*
for m in slim chain; do for inf in 1 3; do
  rm -rf /tmp/pw_$m$inf
  GPX_POTF2=$m timeout 300 rocprofv3 --kernel-trace -d /tmp/pw_$m$inf -- python bench.py --no-cpu-baseline --steps 6 --warmup 1 --inflight $inf > $O/under_$m$inf.json 2> $O/err_$m$inf.txt
  db=$(find /tmp/pw_$m$inf -name '*.db' | head -1)
  echo "## GPX_POTF2=$m, --inflight $inf" >> $O/potf2_wait.md
  python tools/potf2_wait.py "$db" $O/potf2_wait.md > /dev/null 2>> $O/err_$m$inf.txt
  if [ $inf = 1 ]; then python tools/timeline_dump.py "$db" $O/timeline_$m.csv >> $O/err_$m$inf.txt 2>&1; python tools/timeline_analyze.py $O/timeline_$m.csv > $O/timeline_$m.md 2>> $O/err_$m$inf.txt; fi
done; done
cat $O/potf2_wait.md; head -12 $O/timeline_slim.md; head -12 $O/timeline_chain.md
