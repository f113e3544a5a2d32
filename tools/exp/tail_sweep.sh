cd $GRAFT_REPO_ROOT
O=gpurun_out/tail; mkdir -p $O
for cfg in "72 2" "48 2" "96 2" "72 3" "56 2" "72 2"; do
  set -- $cfg
  GPX_TAIL_TILES=$1 GPX_LAZY_GROUP=$2 timeout 200 python bench.py --no-cpu-baseline --steps 24 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('tail=$1 lazy=$2 value %.2f potrf %.3f predict %.3f fit %.2f' % (d['value'], d['stages']['potrf_ms'], d['stages']['predict_ms'], d['stages']['fit_step_ms']))"
done 2>&1 | tee $O/tail.log
