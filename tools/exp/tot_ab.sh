# narrower outer blocks in the chain-bound tail (GPX_TAIL_OUTER_TILES), variant per process, ABAB
mkdir -p gpurun_out/r2
timeout 300 python -m pytest tests/test_gpu_edges.py -m gpu -x -q 2>&1 | tail -3
for r in 1 2; do for v in "0 72" "2 72" "1 72" "2 96" "2 48"; do
set -- $v
GPX_TAIL_OUTER_TILES=$1 GPX_TAIL_TILES=$2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2/tot$1_$2_$r.json 2>gpurun_out/r2/tot$1_$2_$r.err
python - <<PY
import json
d=json.load(open("gpurun_out/r2/tot$1_$2_$r.json"))
st=d.get("stages_ms") or d.get("stages") or {}
print("tot=$1 tail=$2", "value %.2f"%d["value"], "frac %.3f"%d["roofline"]["frac"], "launches", d["roofline"]["launches"], {k:round(v,2) for k,v in d.items() if isinstance(v,float) and ("ms" in k)}, {k:(round(v,2) if isinstance(v,float) else v) for k,v in st.items()} if isinstance(st,dict) else "")
PY
done; done
