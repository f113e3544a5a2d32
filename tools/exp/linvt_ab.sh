cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/linvt}
mkdir -p $O
# gradient parity first: the oracle comparisons of the fit step (single + batched, ragged N)
timeout 900 python -m pytest tests -x -q -m gpu -k "grad or fit or lml or svi or vi or sparse" > $O/t.log 2>&1; echo "grad tests rc=$?"; tail -3 $O/t.log
for rep in 1 2; do for v in tree sweep; do
  GPX_LINVT=$v timeout 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=d['stages']; print('C3 $v potrf %.2f predict %.2f fit %.2f' % (s['potrf_ms'], s['predict_ms'], s['fit_step_ms']))"
done; done
for N in 500 1024 2048 4096 8192; do for v in tree sweep; do
  GPX_LINVT=$v timeout 300 python bench.py --N $N --M 256 --no-cpu-baseline --steps 4 --warmup 2 --inflight 1 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=d['stages']; print('N=$N $v potrf %.3f fit %.3f ms' % (s['potrf_ms'], s['fit_step_ms']))"
done; done
