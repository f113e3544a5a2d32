"""Per dispatch of the diagonal-block kernel (potf2) in a rocprofv3 --kernel-trace database (rocpd sqlite): how long it
EXECUTED (end - start on the device) and how long it WAITED to start after the kernel before it on its own stream had
ended (dispatch latency + waiting for a place on a CU), next to whether a trailing update was running meanwhile.
VERDICT r3 item 1a: `potf2_standalone_us` vs `potf2_in_pipeline_us`, split into the two.

    rocprofv3 --kernel-trace -d /tmp/p -- python bench.py --no-cpu-baseline --steps 4 --warmup 1 --inflight 1
    python tools/potf2_wait.py $(find /tmp/p -name '*.db') [out.md]
"""
import sqlite3
import statistics as st
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, stream_id, start, end from kernels order by start").fetchall()
    by_stream = {}
    for n, s, a, b in rows:
        by_stream.setdefault(s, []).append((a, b, n))
    trail = sorted((a, b) for n, s, a, b in rows if "gemm_nt128_kernel<1" in n)
    out = []
    recs = []
    for s, ks in by_stream.items():
        for (pa, pb, pn), (a, b, n) in zip(ks, ks[1:]):
            if "potf2" not in n:
                continue
            wait = (a - pb) / 1e3
            ex = (b - a) / 1e3
            # a trailing update overlapping the moment this kernel became ready?
            busy = any(ta <= pb < tb for ta, tb in trail)
            recs.append((wait, ex, busy, pn))
    if not recs:
        print("no potf2 dispatches found")
        return
    kname = next(n for n, *_ in rows if "potf2" in n).split("(")[0]

    def line(tag, sel):
        if not sel:
            return f"| {tag} | 0 | | | | |"
        w = [r[0] for r in sel]
        e = [r[1] for r in sel]
        return (f"| {tag} | {len(sel)} | {st.median(w):.1f} | {st.mean(w):.1f} | {st.median(e):.1f} | {st.mean(e):.1f} | "
                f"{sum(w) / 1e3:.2f} + {sum(e) / 1e3:.2f} |")

    out.append(f"kernel `{kname}`: {len(recs)} dispatches that follow another kernel on their stream")
    out.append("")
    out.append("| dispatches | n | wait median us | wait mean us | exec median us | exec mean us | total wait + exec ms |")
    out.append("|---|---|---|---|---|---|---|")
    out.append(line("all", recs))
    out.append(line("a trailing update running when it became ready", [r for r in recs if r[2]]))
    out.append(line("no trailing update running", [r for r in recs if not r[2]]))
    text = "\n".join(out)
    print(text)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "a") as f:
            f.write(text + "\n\n")


if __name__ == "__main__":
    main()
