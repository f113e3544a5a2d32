# kernel timeline of one C3 factorisation with the round-3 defaults (wave-specialised potf2)
O=${1:-gpurun_out/r03t}
mkdir -p $O
export TMPDIR=/tmp
rm -rf /tmp/tl3
timeout 200 rocprofv3 --kernel-trace -d /tmp/tl3 -- python tools/timeline.py 16384 > /dev/null 2>/tmp/tl3.err
db=$(find /tmp/tl3 -name '*.db' | head -1)
python tools/timeline_dump.py $db $O/timeline_c3.csv
python tools/timeline_analyze.py $O/timeline_c3.csv > $O/timeline_c3.txt
head -14 $O/timeline_c3.txt
# the chain of one tail block, kernel by kernel (around t = 26 ms)
python - $O/timeline_c3.csv <<'PY' | tee $O/tail_block.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
sel = [r for r in rows if 26000 <= float(r["start_us"]) <= 26700]
for r in sel:
    print(f"{r['name']:10s} q{r['queue']} {float(r['start_us']):9.1f} {float(r['dur_us']):7.1f}  grid {r['grid_x']}x{r['grid_y']}")
PY
