# anti-locality tile order (GPX_TILE_SWIZZLE=3: the 64 workgroups an XCD holds share no A and no B panel) against grid order:
# ABAB bench + a FETCH_SIZE pass.
mkdir -p gpurun_out/r2
timeout 300 python -m pytest tests/test_gpu_edges.py -m gpu -x -q 2>&1 | tail -2
for r in 1 2; do for v in 0 3; do
GPX_TILE_SWIZZLE=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2/swz3${v}_$r.json 2>gpurun_out/r2/swz3${v}_$r.err
python - <<PY
import json
d=json.load(open("gpurun_out/r2/swz3${v}_$r.json"))
print("swz=$v", "value %.2f"%d["value"], "frac %.3f"%d["roofline"]["frac"], "avg_launch %.4f"%d["roofline"]["avg_launch_ms"], {k:round(v,2) for k,v in d["stages"].items()})
PY
done; done
export TMPDIR=/tmp
for v in 3; do
rm -rf /tmp/pf$v
GPX_TILE_SWIZZLE=$v timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf$v -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>/tmp/pf$v.err
db=$(find /tmp/pf$v -name '*.db' | head -1)
echo "swz=$v"; python tools/rocpd_summary.py $db | grep DOMINANT | cut -c1-300
done
