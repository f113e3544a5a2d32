# round 5, first lease: the default bench line with the configs block, the edge / abi / sparse GPU tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
timeout 900 python bench.py > $O/bench_a.json 2> $O/bench_a.err; echo "bench rc=$?"; cut -c1-400 $O/bench_a.json
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r05/bench_a.json").read().strip().splitlines()[-1])
print(json.dumps(r.get("configs"), indent=1)[:6000])
PY
timeout 1200 python -m pytest tests/test_abi.py tests/test_gpu_edges.py tests/test_gpu_sparse.py tests/test_gpu_node.py -x -q -m gpu > $O/tests_a.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests_a.log
