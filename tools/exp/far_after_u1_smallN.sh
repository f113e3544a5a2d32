# GPX_FAR_AFTER_U1 at smaller N (single-sample stage timings of bench.py)
mkdir -p gpurun_out/r2
for N in 4096 8192 12288; do for r in 1 2; do for v in 0 40; do
GPX_FAR_AFTER_U1=$v timeout 300 python bench.py --N $N --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r2/fauN${N}_${v}_$r.json 2>gpurun_out/r2/fauN${N}_${v}_$r.err
python - <<PY
import json
d=json.load(open("gpurun_out/r2/fauN${N}_${v}_$r.json"))
print("N=$N far_after_u1=$v", "value %.2f"%d["value"], {k:round(v,3) for k,v in d["stages"].items()})
PY
done; done; done
