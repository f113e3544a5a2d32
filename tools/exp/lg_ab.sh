# adaptive schedule: lazy group size in the GEMM-bound head (tail = 72 tile rows), variant per process, ABAB
mkdir -p gpurun_out/r2
for r in 1 2; do for v in 2 3 4; do
GPX_LAZY_GROUP=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2/alg${v}_$r.json 2>gpurun_out/r2/alg${v}_$r.err
python - <<PY
import json
d=json.load(open("gpurun_out/r2/alg${v}_$r.json"))
print("G=$v", "value %.2f"%d["value"], "frac %.3f"%d["roofline"]["frac"], "launches", d["roofline"]["launches"], "avg_launch %.4f"%d["roofline"]["avg_launch_ms"], {k:round(v,2) for k,v in d.items() if isinstance(v,float) and ("ms" in k)})
PY
done; done
