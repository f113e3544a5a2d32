cd $GRAFT_REPO_ROOT
O=gpurun_out/ot; mkdir -p $O
for n in 8192 16384; do
for ot in auto 4 auto 4; do
  if [ $ot = auto ]; then unset GPX_OUTER_TILES; else export GPX_OUTER_TILES=$ot; fi
  timeout 200 python bench.py --N $n --M 2048 --no-cpu-baseline --steps 12 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('N=$n OT=$ot potrf %.3f fit_step %.3f predict %.3f value %.2f' % (d['stages']['potrf_ms'], d['stages']['fit_step_ms'], d['stages']['predict_ms'], d['value']))"
done; done 2>&1 | tee $O/ot3.log
