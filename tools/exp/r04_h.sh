#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04h; mkdir -p $O; rm -f $O/*
timeout 600 python -m pytest tests/test_gpu_rank.py tests/test_bench_launch.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
for rep in 1 2 3 4 5 6; do timeout 300 python bench.py --no-cpu-baseline > $O/bench_$rep.json 2> $O/bench_$rep.err; done
python - <<'PY'
import json,glob
v=[]
for f in sorted(glob.glob('gpurun_out/r04h/bench_*.json')):
    j=json.loads(open(f).read().strip().split('\n')[-1]); v.append(j['value'])
    print(f.split('/')[-1], round(j['value'],2), round(j['stages']['potrf_ms'],2), round(j['stages']['predict_ms'],2), round(j['event_ms_longest_context'],1))
import statistics as st
print('mean', st.mean(v), 'min', min(v), 'max', max(v), 'spread +-%', 100*(max(v)-min(v))/2/st.mean(v))
PY
GPX_RANK_FORCE_COLLECTIVES=1 NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,COLL,P2P NCCL_DEBUG_FILE=$O/rank1_rccl_nccl_debug.log python bench.py --force-rank-path --steps 4 --warmup 1 --c4-S 24 --no-node-record > $O/rank1_rccl_forced.json 2> $O/rank1_rccl_forced.err
grep -c "Broadcast\|AllReduce\|Send\|Recv" $O/rank1_rccl_nccl_debug.log; grep "Broadcast\|AllReduce\|Send:\|Recv:" $O/rank1_rccl_nccl_debug.log | cut -c1-220 | head -12
tail -c 400 $O/rank1_rccl_forced.json
