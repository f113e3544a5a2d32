cd $GRAFT_REPO_ROOT
for N in 512 1024 2048 4096 6144; do for v in 16 0; do
  GPX_SMALL_BK=$v timeout 300 python bench.py --N $N --M 256 --no-cpu-baseline --steps 6 --warmup 2 --inflight 1 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=d['stages']; print('N=$N bk=$v potrf %.3f predict %.3f fit %.3f ms' % (s['potrf_ms'], s['predict_ms'], s['fit_step_ms']))"
done; done
for v in 16 0; do echo "== batched sweeps, GPX_SMALL_BK=$v"; GPX_SMALL_BK=$v timeout 300 python tools/small_n_sweep.py 2>&1 | grep -E " 1 ctx| 3 ctx" ; done
for v in 16 0; do GPX_SMALL_BK=$v timeout 300 python bench.py --no-cpu-baseline --steps 9 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=d['stages']; print('C3 bk=$v potrf %.2f predict %.2f fit %.2f post/s %.2f' % (s['potrf_ms'], s['predict_ms'], s['fit_step_ms'], d['value']))"; done
