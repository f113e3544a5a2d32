"""Per-launch efficiency of the Cholesky trailing update at C3 from a rocprofv3 kernel trace (rocpd sqlite):
usage: trailing_per_launch.py <results.db>  — prints, for the last posterior in the trace, tiles / flops / duration /
TFLOP/s of each of the 31 gemm_nt_kernel<1> launches."""
import sqlite3, sys
N, M, TILE, OUT = 16384, 1024, 128, 4
Np = ((N + 1 + TILE - 1) // TILE) * TILE
nblk, extra = Np // TILE, M // TILE
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select start, end - start from kernels where name like '%gemm_nt_kernel<1,%' order by start").fetchall()
nouter = (nblk + OUT - 1) // OUT
launches = []
for k in range(nouter):
    ob, oe = k * OUT, min(k * OUT + OUT, nblk)
    oe2 = min(oe + OUT, nblk)
    if oe >= nblk or oe2 >= nblk:
        continue
    r0, c0, c1 = oe2, oe2, nblk
    entries = 0
    for t in range(c1 - c0):
        first = max(c0 + t, r0)
        entries += (nblk + extra - first - 1) * TILE * TILE + 0.5 * TILE * (TILE + 1)
    tiles = sum(nblk + extra - max(c0 + t, r0) for t in range(c1 - c0))
    launches.append((k, tiles, 2.0 * (oe - ob) * TILE * entries))
last = rows[-len(launches):]
print(f"{len(launches)} trailing launches per factorisation; trace has {len(rows)} dispatches")
print("  k   tiles  waves(512 slots)   GFLOP      us    TFLOP/s")
tot_f = tot_t = 0.0
for (k, tiles, fl), (_, dur) in zip(launches, last):
    tot_f += fl; tot_t += dur
    print(f"{k:3d} {tiles:7d} {tiles / 512:10.2f} {fl / 1e9:12.1f} {dur / 1e3:8.1f} {fl / dur / 1e3:9.1f}")
print(f"flop-weighted: {tot_f / tot_t / 1e3:.1f} TFLOP/s")
