# GPX_FAR_AFTER_U1 in the TRSM sweeps too: fit step / posterior stage timings, ABAB
mkdir -p gpurun_out/r2
timeout 300 python -m pytest tests/test_gpu_edges.py tests/test_gpu_exactgp.py -m gpu -x -q 2>&1 | tail -2
for N in 16384 8192 4096; do for r in 1 2; do for v in 0 40; do
GPX_FAR_AFTER_U1=$v timeout 300 python bench.py --N $N --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r2/fauF${N}_${v}_$r.json 2>gpurun_out/r2/fauF${N}_${v}_$r.err
python - <<PY
import json
d=json.load(open("gpurun_out/r2/fauF${N}_${v}_$r.json"))
print("N=$N far_after_u1=$v", "value %.2f"%d["value"], {k:round(v,3) for k,v in d["stages"].items()})
PY
done; done; done
