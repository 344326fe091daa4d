"""tuning aid: where the device idles -- union of kernel intervals and the largest gaps of a rocprofv3 kernel trace (rocpd .db):
    python scripts/prof_gaps.py <results.db> [last_ms]"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = sorted(cur.execute("select start, end, name from kernels"))
last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
t_end = max(r[1] for r in rows)
rows = [r for r in rows if r[0] >= t_end - last_ms * 1e6]
span = (rows[-1][1] - rows[0][0]) / 1e3
busy, cur_s, cur_e, gaps = 0.0, rows[0][0], rows[0][1], []
prev_name = rows[0][2]
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(((s - cur_e) / 1e3, prev_name.split("(")[0][:40], n.split("(")[0][:40]))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e: prev_name = n
busy += cur_e - cur_s
print("last %.1f ms of the trace: %d kernels, span %.1f us, device busy (union) %.1f us, sum of kernel times %.1f us" % (last_ms, len(rows), span, busy / 1e3, sum(e - s for s, e, _ in rows) / 1e3))
gaps.sort(reverse=True)
print("idle in gaps: %.1f us over %d gaps; median gap %.2f us" % (sum(g[0] for g in gaps), len(gaps), sorted(g[0] for g in gaps)[len(gaps) // 2] if gaps else 0))
for g in gaps[:12]:
    print("  %8.1f us between %-40s and %s" % g)
