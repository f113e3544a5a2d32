cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/smallbk}
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_edges.py -x -q -m gpu -k "lazy" > $O/t.log 2>&1; echo "lazy test rc=$?"; tail -2 $O/t.log
for rep in 1 2; do for v in 16 32; do
  GPX_SMALL_BK=$v timeout 300 python bench.py --no-cpu-baseline --steps 9 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=d['stages']; print('C3 bk=$v potrf %.2f predict %.2f fit %.2f post/s %.2f classes %s' % (s['potrf_ms'], s['predict_ms'], s['fit_step_ms'], d['value'], d['kernel_classes_ms_per_predict']))"
done; done
for N in 512 1024 2048 4096; do for v in 16 32; do
  GPX_SMALL_BK=$v timeout 300 python bench.py --N $N --M 256 --no-cpu-baseline --steps 6 --warmup 2 --inflight 1 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=d['stages']; print('N=$N bk=$v potrf %.3f predict %.3f fit %.3f ms' % (s['potrf_ms'], s['predict_ms'], s['fit_step_ms']))"
done; done
