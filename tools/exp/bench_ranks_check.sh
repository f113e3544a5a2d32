cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/ranks}
mkdir -p $O
timeout 900 python -m pytest tests/test_bench_launch.py -x -q -m gpu > $O/t.log 2>&1; echo "bench launch tests rc=$?"; tail -3 $O/t.log
timeout 900 python bench.py --gpus 2 --share-gpu --steps 6 --warmup 2 --c4-S 100 > $O/b2.json 2> $O/b2.err; echo "rc=$?"
python - $O/b2.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "n_gpus", "multi_gpu_path")}, d["node_sweep"])
PY
