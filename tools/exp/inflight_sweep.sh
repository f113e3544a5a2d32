cd $GRAFT_REPO_ROOT
for rep in 1 2; do for f in 3 4 5 6; do
  timeout 300 python bench.py --no-cpu-baseline --steps 36 --warmup 6 --inflight $f 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('inflight=$f post/s %.2f ms/step %.2f pipeline frac %.3f' % (d['value'], d['ms_per_step'], d['pipeline_frac_of_fp64_peak']))"
done; done
