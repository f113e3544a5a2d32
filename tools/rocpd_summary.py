"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) run: per-kernel stats (--kernel-trace --stats) and,
for a --pmc pass, per-kernel counter sums/averages.  Usage: rocpd_summary.py <results.db> [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    for name, n, s, a, mn, mx in rows:
        short = name if len(name) < 90 else name[:87] + "..."
        lines.append(f"| `{short}` | {n} | {s / 1e6:.3f} | {a / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * s / tot:.1f} |")
    try:
        pmc = cur.execute("select name, counter_name, count(*), sum(counter_value), avg(counter_value) "
                          "from pmc_events group by name, counter_name order by sum(counter_value) desc").fetchall()
    except Exception as e:  # schema differences
        pmc = []
        lines.append(f"\n(pmc query failed: {e})")
    if pmc:
        lines += ["", "| kernel | counter | dispatches | sum | avg per dispatch |", "|---|---|---|---|---|"]
        for name, cn, n, s, a in pmc:
            short = name if len(name) < 70 else name[:67] + "..."
            lines.append(f"| `{short}` | {cn} | {n} | {s:.6g} | {a:.6g} |")
    # the Cholesky trailing SYRK launches of gemm_nt_kernel: square grids of (Np/128 - 4k) tiles
    try:
        rows = cur.execute("select grid_size_x/256, grid_size_y, grid_size_z, counter_name, value, duration "
                           "from counters_collection where kernel_name like 'gpx::gemm_nt_kernel%'").fetchall()
    except Exception:
        rows = []
    if rows:
        gmax = max(r[0] for r in rows if r[0] == r[1] and r[2] == 1)
        trail = [r for r in rows if r[0] == r[1] and r[2] == 1 and (gmax - r[0]) % 4 == 0]
        if trail:
            cname = trail[0][3]
            n = len(trail)
            tot = sum(r[4] for r in trail)
            lines += ["", f"Cholesky trailing-update launches (square grids {gmax}, {gmax - 4}, ... tiles): "
                          f"{n} dispatches, {cname} sum = {tot:.6g} KB, avg per launch = {tot / n:.6g} KB, "
                          f"avg duration under PMC = {sum(r[5] for r in trail) / n / 1e3:.1f} us"]
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "a").write(text)
    print(text)


if __name__ == "__main__":
    main()
