cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${1:-gpurun_out/sgptrace}; mkdir -p $O
rm -rf /tmp/st; timeout 300 rocprofv3 --kernel-trace -d /tmp/st -- python tools/exp/sgp_trace.py > $O/log 2>&1
db=$(find /tmp/st -name '*.db' | head -1)
python - $db $O/sgp.csv <<'PY'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select name, stream_id, start, end, grid_x, grid_y, grid_z from kernels order by start").fetchall()
gi = [i for i, r in enumerate(rows) if 'gram' in r[0]]
# last step: starts at the third-from-last... find the first gram kernel of the last step = after the last grad kernel of step 2
# simpler: split by big time gaps (> 200 us of host work between steps)
starts = [0]
for i in range(1, len(rows)):
    if rows[i][2] - rows[i - 1][3] > 150e3: starts.append(i)
rows = rows[starts[-1]:]
t0 = rows[0][2]
with open(sys.argv[2], 'w') as f:
    for n, s, a, b, gx, gy, gz in rows:
        short = re.sub(r'\(.*', '', n).replace('gpx::', '')[:60]
        line = f"{short},{s},{(a - t0) / 1e3:.1f},{(b - a) / 1e3:.1f},{gx},{gy},{gz}"
        f.write(line + "\n")
        if (b - a) > 40e3: print(line)
print(len(rows), "kernels, span ms", (rows[-1][3] - t0) / 1e6, "kernel sum ms", sum(r[3] - r[2] for r in rows) / 1e6)
PY
