# theta samples in flight per GPU (bench.py --inflight), variant per process
mkdir -p gpurun_out/r2
for r in 1 2; do for v in 2 3 4 6; do
timeout 300 python bench.py --steps 48 --warmup 6 --inflight $v --no-cpu-baseline > gpurun_out/r2/infl${v}_$r.json 2>gpurun_out/r2/infl${v}_$r.err
python - <<PY
import json
d=json.load(open("gpurun_out/r2/infl${v}_$r.json"))
print("inflight=$v", "value %.2f"%d["value"], "ms/step %.2f"%d["ms_per_step"], "pipeline frac %.3f"%d["pipeline_frac_of_fp64_peak"])
PY
done; done
