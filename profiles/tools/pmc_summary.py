#!/usr/bin/env python3
"""Summarise rocprofv3 output databases: per kernel the dispatch count, average duration and, if the run collected
counters (--pmc), the per-dispatch average of every counter.  usage: pmc_summary.py <dir-with-.db> [kernel-substring ...]"""
import glob, os, sqlite3, sys
from collections import defaultdict


def cols(cur, view):
    cur.execute("SELECT * FROM %s LIMIT 1" % view)
    return [d[0] for d in cur.description]


def main():
    root, pats = sys.argv[1], sys.argv[2:]
    for db in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
        con = sqlite3.connect(db)
        cur = con.cursor()
        views = [r[0] for r in cur.execute("SELECT name FROM sqlite_master WHERE type IN ('view','table')")]
        print("#", os.path.relpath(db, root))
        if "kernels" in views:
            c = cols(cur, "kernels")
            kn = "name" if "name" in c else [x for x in c if "name" in x][0]
            dur = "duration" if "duration" in c else None
            q = "SELECT %s, COUNT(*), AVG(%s) FROM kernels GROUP BY %s ORDER BY SUM(%s) DESC" % (kn, dur or "end-start", kn, dur or "end-start")
            for name, n, avg in cur.execute(q):
                if pats and not any(p in name for p in pats): continue
                print("dur\t%s\t%d\t%.1f ns" % (name[:70], n, avg))
        if "pmc_events" in views:
            c = cols(cur, "pmc_events")
            kn = [x for x in c if x in ("name", "kernel_name")] or [x for x in c if "name" in x and "counter" not in x]
            cn = [x for x in c if x in ("counter_name", "pmc_name")] or [x for x in c if "counter" in x and "name" in x]
            vn = [x for x in c if x in ("counter_value", "value")] or [x for x in c if "value" in x]
            dn = [x for x in c if x in ("dispatch_id", "event_id")] or [x for x in c if "dispatch" in x]
            if not (kn and cn and vn):
                print("# pmc_events columns:", c)
                continue
            acc, nd = defaultdict(float), defaultdict(set)
            for name, ctr, val, did in cur.execute("SELECT %s, %s, %s, %s FROM pmc_events" % (kn[0], cn[0], vn[0], dn[0] if dn else "0")):
                if pats and not any(p in name for p in pats): continue
                acc[(name, ctr)] += val or 0
                nd[(name, ctr)].add(did)
            for (name, ctr) in sorted(acc):
                n = max(1, len(nd[(name, ctr)]))
                print("pmc\t%s\t%s\t%d\t%.1f" % (name[:70], ctr, n, acc[(name, ctr)] / n))
        con.close()


if __name__ == "__main__":
    main()
