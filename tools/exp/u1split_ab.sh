# U1 split (GPX_U1_SPLIT: 1 tail, 2 everywhere), variant per process, ABAB
mkdir -p gpurun_out/r2
timeout 300 python -m pytest tests/test_gpu_edges.py tests/test_gpu_exactgp.py -m gpu -x -q 2>&1 | tail -3
for r in 1 2; do for v in 0 1 2; do
GPX_U1_SPLIT=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2/u1s${v}_$r.json 2>gpurun_out/r2/u1s${v}_$r.err
python - <<PY
import json
d=json.load(open("gpurun_out/r2/u1s${v}_$r.json"))
st=d.get("stages") or {}
print("u1split=$v", "value %.2f"%d["value"], "frac %.3f"%d["roofline"]["frac"], "launches", d["roofline"]["launches"], {k:(round(v,2) if isinstance(v,float) else v) for k,v in st.items()})
PY
done; done
