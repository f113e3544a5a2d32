"""Overlap analysis of one factorisation's kernel timeline (tools/timeline.py + timeline_dump.py CSV): how long the
trailing-update stream is busy / idle, what the panel chain is doing meanwhile, per outer block."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    for k in ("start_us", "end_us", "dur_us"):
        r[k] = float(r[k])
trail = [r for r in rows if r["name"].startswith("big1_")]
chain = [r for r in rows if not r["name"].startswith("big1_") and r["name"] not in ("gram_kernel", "augment_kernel", "lml_terms_kernel")]
span = max(r["end_us"] for r in rows) - min(r["start_us"] for r in rows)
busy = sum(r["dur_us"] for r in trail)
gaps = [b["start_us"] - a["end_us"] for a, b in zip(trail, trail[1:])]
print(f"span {span / 1e3:.2f} ms; trailing launches {len(trail)}: busy {busy / 1e3:.2f} ms, gaps between consecutive launches "
      f"{sum(g for g in gaps if g > 0) / 1e3:.2f} ms (max {max(gaps):.0f} us); first trailing launch starts at "
      f"{trail[0]['start_us'] / 1e3:.2f} ms, last ends at {trail[-1]['end_us'] / 1e3:.2f} ms")
by = {}
for r in chain:
    by.setdefault(r["name"], [0, 0.0])
    by[r["name"]][0] += 1
    by[r["name"]][1] += r["dur_us"]
for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"  chain {k:12s} {n:5d} launches {t / 1e3:7.2f} ms  avg {t / n:6.1f} us")
# chain idle time: union of chain kernel intervals
iv = sorted((r["start_us"], r["end_us"]) for r in chain)
cov, cur_s, cur_e = 0.0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        cov += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
cov += cur_e - cur_s
print(f"chain streams busy (union) {cov / 1e3:.2f} ms of {span / 1e3:.2f}")
print("per trailing launch: start, dur, gap before (us)")
prev = None
for i, r in enumerate(trail):
    g = r["start_us"] - prev if prev is not None else 0.0
    print(f"  {i:2d} {r['start_us']:9.1f} {r['dur_us']:8.1f} {g:8.1f}")
    prev = r["end_us"]
