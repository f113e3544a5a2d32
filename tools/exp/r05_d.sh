# sparse ride-along with CU reservation for the chain (GPX_SGP_RESERVE)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
for r in 0 32 16 64; do
GPX_SGP_RESERVE=$r timeout 300 python bench_configs.py C5 > $O/c5_res$r.json 2> $O/c5_res$r.err
python - $O/c5_res$r.json "reserve $r" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["C5"]
if "error" in d: print(sys.argv[2], d); raise SystemExit
print('%s bound %.2f ms (%.3f)  bound+grad %.2f ms (%.3f)  posterior %.1f ms  api step %.2f ms  checksum %.12f' % (sys.argv[2], d['sparse_bound']['ms'], d['sparse_bound']['frac_of_fp64_peak'], d['sparse_bound_and_gradient']['ms'], d['sparse_bound_and_gradient']['frac_of_fp64_peak'], d['sparse_posterior_all_pixels']['ms'], d['viSparseGP_api']['ms_per_svi_step'], d['checksum']))
PY
done
GPX_SGP_RESERVE=32 bash tools/exp/sgp_trace2.sh $O/sgptrace_res32
