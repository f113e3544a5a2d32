# round 5: the ride-along sparse forward pass (GPX_SGP_SOLVE=ride, default) against round 4's (inverse): tests, C5 record, soak
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "sparse or sgp or Sparse or forward_pass" > $O/sgp_tests.log 2>&1; echo "sparse tests rc=$?"; tail -3 $O/sgp_tests.log
for v in ride inverse; do
GPX_SGP_SOLVE=$v timeout 300 python bench_configs.py C5 > $O/c5_$v.json 2> $O/c5_$v.err
python - $O/c5_$v.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["C5"]
if "error" in d: print(sys.argv[2], d); raise SystemExit
print('%s bound %.2f ms (%.3f)  bound+grad %.2f ms (%.3f)  posterior %.1f ms  api step %.2f ms  checksum %.12f' % (sys.argv[2], d['sparse_bound']['ms'], d['sparse_bound']['frac_of_fp64_peak'], d['sparse_bound_and_gradient']['ms'], d['sparse_bound_and_gradient']['frac_of_fp64_peak'], d['sparse_posterior_all_pixels']['ms'], d['viSparseGP_api']['ms_per_svi_step'], d['checksum']))
PY
done
timeout 600 python tools/exp/sgp_soak.py 2>&1 | tail -4
