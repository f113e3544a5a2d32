cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${1:-gpurun_out/fittrace}; mkdir -p $O
rm -rf /tmp/ft; timeout 300 rocprofv3 --kernel-trace -d /tmp/ft -- python tools/exp/fit_trace.py > $O/log 2>&1
db=$(find /tmp/ft -name '*.db' | head -1)
python tools/timeline_dump.py $db $O/fit_tree.csv
python - $O/fit_tree.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
# after the last potf2: the gradient part
last = max(i for i, r in enumerate(rows) if r['name'] == 'potf2')
for r in rows[last + 1:]:
    if float(r['dur_us']) > 30: print(r['name'], r['start_us'], r['dur_us'], r['grid_x'], r['grid_y'])
print("grad part span ms", (float(rows[-1]['end_us']) - float(rows[last]['end_us'])) / 1e3, "total", float(rows[-1]['end_us']) / 1e3)
PY
