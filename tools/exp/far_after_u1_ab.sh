# far update of a tail block started only after U1 of the same block (GPX_FAR_AFTER_U1 = tile-row threshold), ABAB
mkdir -p gpurun_out/r2
timeout 300 python -m pytest tests/test_gpu_edges.py -m gpu -x -q 2>&1 | tail -2
for r in 1 2; do for v in 0 40 56 72; do
GPX_FAR_AFTER_U1=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2/fau${v}_$r.json 2>gpurun_out/r2/fau${v}_$r.err
python - <<PY
import json
d=json.load(open("gpurun_out/r2/fau${v}_$r.json"))
print("far_after_u1=$v", "value %.2f"%d["value"], "frac %.3f"%d["roofline"]["frac"], {k:round(v,2) for k,v in d["stages"].items()})
PY
done; done
