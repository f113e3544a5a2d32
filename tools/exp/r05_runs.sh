# four consecutive default runs of bench.py (configs block included) on one box: the spread of the headline and of the configs
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
: > $O/bench_four_runs_one_box.jsonl
for i in 1 2 3 4; do timeout 600 python bench.py 2>/dev/null | tail -1 >> $O/bench_four_runs_one_box.jsonl; done
python - <<'PY'
import json
for ln in open("gpurun_out/r05/bench_four_runs_one_box.jsonl"):
    r = json.loads(ln); c = r["configs"]
    print("value %.2f potrf %.2f frac %.3f | C1 nuts512 %.2fs nutsN25 %.3fs step25 %.1fus | C2 potrf %.3f fit %.3f sweep %.0f | C4 %.1f | C5 bound %.2f step %.2f svi %.2f exact %.1f" % (
        r["value"], r["stages"]["potrf_ms"], r["roofline"]["frac"], c["C1"]["fit_s"], c["C1"]["host_api_fit_step"]["nuts_200_200_N25_s"], c["C1"]["host_api_fit_step"]["fit_step_ms_N25"] * 1e3,
        c["C2"]["stages"]["potrf_ms"], c["C2"]["stages"]["fit_step_ms"], c["C2"]["sweep_posteriors_per_s"], c["C4"]["posteriors_per_s"],
        c["C5"]["sparse_bound"]["ms"], c["C5"]["sparse_bound_and_gradient"]["ms"], c["C5"]["viSparseGP_api"]["ms_per_svi_step"], c["C5"]["viGP_exact_api"]["ms_per_svi_step"]))
PY
