#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace result database (rocpd sqlite) into profiles/<name>_kernel_stats.csv/.md
usage: python scripts/prof_summary.py gpurun_out/prof/r5_results.db profiles/r01_decode"""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
with open(out + "_kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], "%.1f" % r[3], r[4], r[5], "%.2f" % (100.0 * r[2] / tot)])
shapes = list(cur.execute("select name, grid_x, grid_y, workgroup_x, lds_size, vgpr_count, count(*), avg(end-start) from kernels group by name, grid_x, grid_y order by 1, 2"))
with open(out + "_kernel_stats.md", "w") as f:
    f.write("| kernel | calls | avg us | min us | max us | % of GPU time |\n|---|---|---|---|---|---|\n")
    for r in rows:
        f.write("| `%s` | %d | %.2f | %.2f | %.2f | %.1f |\n" % (r[0][:90], r[1], r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot))
    f.write("\nper launch shape:\n\n| kernel | grid (threads) | wg | LDS B | VGPR | calls | avg us |\n|---|---|---|---|---|---|---|\n")
    for r in shapes:
        f.write("| `%s` | %d x %d | %d | %d | %d | %d | %.2f |\n" % (r[0][:60], r[1], r[2], r[3], r[4], r[5], r[6], r[7] / 1e3))
print(open(out + "_kernel_stats.md").read())
