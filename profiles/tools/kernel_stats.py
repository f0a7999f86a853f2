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

# K1 (k_sweep<3, *>: stage A's fused sweep) on the shard itself: the run starts with a tiny warm-up data set (kernel loading), whose launch is left out
try:
    k1 = [r[0] for r in cur.execute("SELECT duration FROM kernels WHERE name LIKE '%k_sweep<3,%' OR name LIKE '%k_sweep_lean<3>%'").fetchall()]
    if k1:
        full = [d for d in k1 if d > 0.6 * max(k1)]
        print("# K1 k_sweep<3, *> / k_sweep_lean<3>: %d launches on the shard (of %d), mean %.2f us" % (len(full), len(k1), sum(full) / len(full) / 1e3))
except Exception as ex:
    print("# (no K1 line: %s)" % ex)

# GPU timeline: how much of the span between the first and the last kernel of the timed steps is spent inside kernels, and how the
# idle time is distributed (launch gaps vs host waits); the last `--steps` passes are delimited by the k_prepare launches
try:
    ev = cur.execute("SELECT name, start, end FROM kernels ORDER BY start").fetchall()
    # first big kernel of pga_begin: k_genome_sort* (stage A's orders may take two launches: k_genome_sort2 + k_genome_sort2d) or, for
    # genomes beyond the per-genome sorts (the 110 k-hit assemblies of configs[4]), k_prepare in front of the multi-workgroup radix sort
    # (round 5: k_prepare runs once per upload; the multi-workgroup path of pga_begin now opens with k_score_key -- 64-bit score keys -- or k_xkey)
    def first_of_begin(i):
        n = ev[i][0]
        if n.startswith("k_genome_sort"): return not (i and ev[i - 1][0].startswith("k_genome_sort"))
        if n.startswith("k_score_key"): return True
        if n.startswith("k_xkey"): return not any(ev[j][0].startswith("k_score_key") for j in range(max(0, i - 80), i))
        return False
    starts = [i for i in range(len(ev)) if first_of_begin(i)]
    if len(starts) >= 3:
        a, b = starts[-2], starts[-1]  # one whole pass: from one k_prepare to the next
        seg = ev[a:b]
        span = seg[-1][2] - seg[0][1]
        busy = sum(e[2] - e[1] for e in seg)
        gaps = [seg[i + 1][1] - seg[i][2] for i in range(len(seg) - 1)]
        big = [g for g in gaps if g > 15000]
        print("# one pass (from the first kernel of pga_begin to the next one): %d launches, span %.3f ms, inside kernels %.3f ms (%.0f %%), idle %.3f ms: %d gaps > 15 us (host waits) = %.3f ms, the other %d gaps = %.3f ms (mean %.2f us)"
              % (len(seg), span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, len(big), sum(big) / 1e6, len(gaps) - len(big), (sum(gaps) - sum(big)) / 1e6,
                 (sum(gaps) - sum(big)) / 1e3 / max(1, len(gaps) - len(big))))
        # where the host waits are: (kernel before the gap -> kernel after it), count, mean gap
        kinds = {}
        for i in range(len(seg) - 1):
            g = seg[i + 1][1] - seg[i][2]
            if g > 15000:
                k = (seg[i][0].split("(")[0][:40], seg[i + 1][0].split("(")[0][:40])
                kinds.setdefault(k, []).append(g)
        for k, v in sorted(kinds.items(), key=lambda kv: -sum(kv[1])):
            print("#   wait  %-40s -> %-40s  x%-3d mean %.1f us" % (k[0], k[1], len(v), sum(v) / len(v) / 1e3))
except Exception as ex:  # older databases
    print("# (no timeline: %s)" % ex)

# stage A of the last whole pass, launch by launch (start relative to the pass's first kernel, duration)
try:
    if len(starts) >= 3:
        seg = ev[starts[-2]:starts[-1]]
        t0 = seg[0][1]
        last = max(i for i, e in enumerate(seg) if e[0].startswith("k_subopt2") or e[0].startswith("k_chain") or e[0].startswith("k_genome_filters"))
        print("# stage A of that pass (pga_begin + pga_ingest), launch by launch: start us, duration us, kernel")
        for e in seg[:last + 1]:
            print("#   %9.1f %9.1f  %s" % ((e[1] - t0) / 1e3, (e[2] - e[1]) / 1e3, e[0][:70]))
except Exception as ex:
    print("# (no stage-A timeline: %s)" % ex)
