#!/usr/bin/env python3
"""Per-kernel totals of a `rocprofv3 --kernel-trace --stats` run (reads the .db it leaves): usage kernel_stats.py <dir>"""
import glob, os, sqlite3, sys
db = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True))[0]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("SELECT name, COUNT(*), SUM(duration), AVG(duration), MIN(duration), MAX(duration) FROM kernels GROUP BY name ORDER BY SUM(duration) DESC").fetchall()
tot = sum(r[2] for r in rows)
print("# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline   (first pass + warm-up + timed steps)")
print("# total kernel time %.3f ms in %d launches" % (tot / 1e6, sum(r[1] for r in rows)))
print("# %-88s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
for r in rows:
    print("%-90s %7d %12.1f %10.2f %10.2f %10.2f %6.2f" % (r[0][:90], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot))
