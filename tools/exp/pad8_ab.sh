# grid.x of the big-tile GEMM padded to a multiple of 8 (GPX_GRID_PAD8): same row-major order, but XCD x then holds tile
# columns x, x + 8, ... in every tile row.  ABAB bench + a FETCH_SIZE pass of each.
mkdir -p gpurun_out/r2
for r in 1 2; do for v in 0 1; do
GPX_GRID_PAD8=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2/pad${v}_$r.json 2>gpurun_out/r2/pad${v}_$r.err
python - <<PY
import json
d=json.load(open("gpurun_out/r2/pad${v}_$r.json"))
print("pad8=$v", "value %.2f"%d["value"], "frac %.3f"%d["roofline"]["frac"], "avg_launch %.4f"%d["roofline"]["avg_launch_ms"], {k:round(v,2) for k,v in d["stages"].items()})
PY
done; done
export TMPDIR=/tmp
for v in 0 1; do
rm -rf /tmp/pf$v
GPX_GRID_PAD8=$v timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf$v -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>/tmp/pf$v.err
db=$(find /tmp/pf$v -name '*.db' | head -1)
echo "pad8=$v"; python tools/rocpd_summary.py $db | grep DOMINANT | cut -c1-300
done
