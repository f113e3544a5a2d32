cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/linvt}
mkdir -p $O
bash tools/exp/fit_trace.sh $O 2>&1 | grep -E "big0_0p|span"
for rep in 1 2; do for v in tree sweep; do
  GPX_LINVT=$v timeout 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=d['stages']; print('C3 $v potrf %.2f predict %.2f fit %.2f' % (s['potrf_ms'], s['predict_ms'], s['fit_step_ms']))"
done; done
