# kernel timeline of ONE viSparseGP step (bound + gradient) at C5 size: rocprofv3 --kernel-trace, the last step's launches
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
O=${1:-gpurun_out/sgptrace}; mkdir -p $O
rm -rf /tmp/st; timeout 300 rocprofv3 --kernel-trace -d /tmp/st -- python tools/exp/sgp_trace.py > $O/log 2>&1
db=$(find /tmp/st -name '*.db' | head -1)
python - $db $O/c5_step_timeline.md <<'PY'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select name, start, end, grid_x, grid_y, grid_z from kernels order by start").fetchall()
# a step starts at the Gram build of Kuu: the gram kernel that follows a host gap (steps end with a stream synchronisation)
starts = [i for i in range(1, len(rows)) if rows[i][1] - rows[i - 1][2] > 100e3]
rows = rows[starts[-1]:] if starts else rows
t0 = rows[0][1]
lines = [f"one viSparseGP step (bound + gradient), C5 size: {len(rows)} launches, span {(rows[-1][2] - t0) / 1e6:.3f} ms, "
         f"kernel time {sum(r[2] - r[1] for r in rows) / 1e6:.3f} ms", "",
         "runs of the same kernel and grid are folded into one row", "",
         "| start us | span us | launches | kernel time us | kernel (grid) |", "|---|---|---|---|---|"]
run = None
def flush(run):
    lines.append(f"| {run[1]:.1f} | {run[4] - run[1]:.1f} | {run[2]} | {run[3]:.1f} | `{run[0]}` |")
for n, a, b, gx, gy, gz in rows:
    short = re.sub(r'\(.*', '', n).replace('void ', '').replace('gpx::', '')[:70]
    key = f"{short} ({gx} x {gy} x {gz})"
    a_us, d = (a - t0) / 1e3, (b - a) / 1e3
    if run and run[0] == key:
        run[2] += 1; run[3] += d; run[4] = a_us + d
    else:
        if run: flush(run)
        run = [key, a_us, 1, d, a_us + d]
flush(run)
open(sys.argv[2], "w").write("\n".join(lines) + "\n")
print(lines[0])
PY
