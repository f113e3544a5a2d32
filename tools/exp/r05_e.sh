cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_edges.py tests/test_gpu_kernels.py tests/test_gpu_models.py -x -q -m gpu > $O/tests_e.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|Error" $O/tests_e.log | tail -3
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-configs > $O/bench_auto$i.json 2>/dev/null
python - $O/bench_auto$i.json auto <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "value %.2f  potrf %.2f fit %.2f predict %.2f  roofline %.3f  classes %s" % (r["value"], r["stages"]["potrf_ms"], r["stages"]["fit_step_ms"], r["stages"]["predict_ms"], r["roofline"]["frac"], {k: round(v, 2) for k, v in r["kernel_classes_ms_per_predict"].items()}))
PY
done
timeout 300 python bench_configs.py C2 C5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['C2']['stages'], d['C2']['sweep_posteriors_per_s']); c=d['C5']; print(c['sparse_bound']['ms'], c['sparse_bound_and_gradient']['ms'], c['viSparseGP_api']['ms_per_svi_step'])"
