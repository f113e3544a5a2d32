# kernel timeline of ONE viSparseGP step (bound + gradient) at C5 size with the STREAM of every launch (three streams since
# round 5): rocprofv3 --kernel-trace, the last step = from the last Kuu Gram build (the launch before pad_identity_kernel)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
O=${1:-gpurun_out/sgptrace}; mkdir -p $O
rm -rf /tmp/st; timeout 300 rocprofv3 --kernel-trace -d /tmp/st -- python tools/exp/sgp_trace.py > $O/log 2>&1
db=$(find /tmp/st -name '*.db' | head -1)
python - $db $O/c5_step_timeline.md <<'PY'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select name, start, end, grid_x, grid_y, grid_z, stream_id from kernels order by start").fetchall()
pads = [i for i, r in enumerate(rows) if "pad_identity" in r[0]]
rows = rows[pads[-1] - 1:]
t0 = rows[0][1]
streams = {}
for r in rows:
    streams.setdefault(r[6], len(streams))
lines = [f"one viSparseGP step (bound + gradient), C5 size: {len(rows)} launches on {len(streams)} streams, span {(max(r[2] for r in rows) - t0) / 1e6:.3f} ms, "
         f"kernel time {sum(r[2] - r[1] for r in rows) / 1e6:.3f} ms", "",
         "stream = order of first use in the step; runs of the same kernel and grid on one stream are folded into one row", "",
         "| start us | end us | stream | launches | kernel time us | kernel (grid) |", "|---|---|---|---|---|---|"]
run = None
def flush(run):
    lines.append(f"| {run[1]:.1f} | {run[4]:.1f} | {run[5]} | {run[2]} | {run[3]:.1f} | `{run[0]}` |")
for n, a, b, gx, gy, gz, st in rows:
    short = re.sub(r'\(.*', '', n).replace('void ', '').replace('gpx::', '')[:70]
    key = f"{short} ({gx // 256} x {gy} x {gz})"
    a_us, d = (a - t0) / 1e3, (b - a) / 1e3
    if run and run[0] == key and run[5] == streams[st]:
        run[2] += 1; run[3] += d; run[4] = a_us + d
    else:
        if run: flush(run)
        run = [key, a_us, 1, d, a_us + d, streams[st]]
flush(run)
open(sys.argv[2], "w").write("\n".join(lines) + "\n")
print(lines[0])
PY
