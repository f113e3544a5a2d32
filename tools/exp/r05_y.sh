cd $GRAFT_REPO_ROOT
for b in 7 8 7 8 16; do
GPX_SWEEP_BATCH=$b timeout 600 python bench_configs.py C4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B=$b C4', round(d['C4']['posteriors_per_s'],2), d['C4']['batch'], d['C4']['checksum'])"
done
