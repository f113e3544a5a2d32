cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in 400 200 100; do
  GPX_SMALL_TILES_MAX=$v timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 6 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=d['stages']; print('stm=$v value %.2f potrf %.2f predict %.2f classes %s' % (d['value'], s['potrf_ms'], s['predict_ms'], d['kernel_classes_ms_per_predict']))"
done; done
