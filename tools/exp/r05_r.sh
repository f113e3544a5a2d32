cd $GRAFT_REPO_ROOT
timeout 600 python bench_configs.py C4 C2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C4', d['C4']['posteriors_per_s'], d['C4']['batch'], 'C2 sweep', d['C2']['sweep_posteriors_per_s'], d['C2']['stages'])"
timeout 600 python bench_configs.py C4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C4', d['C4']['posteriors_per_s'], d['C4']['batch'])"
