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
    # the dominant kernel has its own symbol: gpx::gemm_nt128_kernel<1, EPI> (Cholesky trailing SYRK; round 1:
    # gemm_nt_kernel<1, ...>)
    try:
        rows = cur.execute("select counter_name, count(*), sum(value), avg(value), avg(duration) from counters_collection "
                           "where kernel_name like '%gemm_nt128_kernel<1,%' or kernel_name like '%gemm_nt128_swz_kernel<1,%' "
                           "or kernel_name like '%gemm_nt_kernel<1,%' "
                           "group by counter_name").fetchall()
    except Exception:
        rows = []
    for cname, n, tot, avg, dur in rows:
        lines += ["", f"DOMINANT trailing-update kernel: {n} dispatches, {cname} sum = {tot:.6g}, avg per launch = {avg:.6g}, "
                      f"avg duration under PMC = {dur / 1e3:.1f} us"]
        if len(sys.argv) > 3:
            import json
            import os
            d = json.load(open(sys.argv[3])) if os.path.exists(sys.argv[3]) else {}
            d[cname] = {"dispatches": n, "avg_per_launch": avg, "sum": tot, "avg_duration_us": dur / 1e3}
            json.dump(d, open(sys.argv[3], "w"), indent=1)
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "a").write(text)
    print(text)


if __name__ == "__main__":
    main()
