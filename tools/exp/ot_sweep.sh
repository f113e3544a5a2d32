cd $GRAFT_REPO_ROOT
O=gpurun_out/ot; mkdir -p $O
for n in 5120 6144 8192; do
for ot in auto 64 auto 64; do
  if [ $ot = auto ]; then unset GPX_OUTER_TILES; else export GPX_OUTER_TILES=$ot; fi
  timeout 200 python bench.py --N $n --M 256 --no-cpu-baseline --steps 6 --warmup 2 --inflight 1 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('N=$n OT=$ot potrf %.3f fit_step %.3f predict %.3f' % (d['stages']['potrf_ms'], d['stages']['fit_step_ms'], d['stages']['predict_ms']))"
done; done 2>&1 | tee $O/ot4.log
