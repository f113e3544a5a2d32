# A/B: sparse forward pass {ride, inverse} x latency GEMM {r5, r1} on the C5 record; the C3 headline under r5 / r1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
for lg in r5 r1; do for v in ride inverse; do
GPX_LAT_GEMM=$lg GPX_SGP_SOLVE=$v timeout 300 python bench_configs.py C5 > $O/c5_${v}_$lg.json 2> $O/c5_${v}_$lg.err
python - $O/c5_${v}_$lg.json "$v $lg" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["C5"]
if "error" in d: print(sys.argv[2], d); raise SystemExit
print('%s bound %.2f ms (%.3f)  bound+grad %.2f ms (%.3f)  posterior %.1f ms  api step %.2f ms  exact step %.1f ms checksum %.12f' % (sys.argv[2], d['sparse_bound']['ms'], d['sparse_bound']['frac_of_fp64_peak'], d['sparse_bound_and_gradient']['ms'], d['sparse_bound_and_gradient']['frac_of_fp64_peak'], d['sparse_posterior_all_pixels']['ms'], d['viSparseGP_api']['ms_per_svi_step'], d['viGP_exact_api']['ms_per_svi_step'], d['checksum']))
PY
done; done
for lg in r5 r1 r5 r1; do
GPX_LAT_GEMM=$lg timeout 600 python bench.py --no-cpu-baseline --no-configs > $O/bench_$lg.json 2>/dev/null
python - $O/bench_$lg.json $lg <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "value %.2f  potrf %.2f fit %.2f predict %.2f  roofline %.3f  classes %s" % (r["value"], r["stages"]["potrf_ms"], r["stages"]["fit_step_ms"], r["stages"]["predict_ms"], r["roofline"]["frac"], {k: round(v, 2) for k, v in r["kernel_classes_ms_per_predict"].items()}))
PY
done
for lg in r5 r1; do GPX_LAT_GEMM=$lg timeout 300 python bench_configs.py C2 C1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lg', d['C2']['stages'], d['C1']['fit_s'], d['C1']['host_api_fit_step'])"; done
