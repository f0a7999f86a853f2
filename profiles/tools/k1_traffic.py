#!/usr/bin/env python3
"""HBM bytes per launch of K1 (k_sweep<1, true>) from the two PMC passes.  rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB
(1 unit = 1024 B); on gfx950 FETCH_SIZE tallies 64 B per request where a wide coalesced read moves 128 B, so it is doubled
(MI355X_MICROARCH.md, HBM section).  usage: k1_traffic.py <pmc_summary.txt> <bench.json>"""
import json, sys
fetch = write = None
for line in open(sys.argv[1]):
    f = line.rstrip("\n").split("\t")
    if len(f) >= 5 and f[0] == "pmc" and "k_sweep<1" in f[1]:
        if f[2] == "FETCH_SIZE": fetch = float(f[4])
        if f[2] == "WRITE_SIZE": write = float(f[4])
b = json.load(open(sys.argv[2]))
hits = b["roofline"]["hits_per_launch"]
tot = int((2 * fetch + write) * 1024)
print(json.dumps({"kernel": "k_sweep<1, *>", "hits_per_launch": hits, "fetch_kb_raw": fetch, "write_kb": write, "bytes_per_launch": tot,
                  "bytes_per_hit": round(tot / hits, 1), "note": "FETCH_SIZE x2 (gfx950 correction), separate --pmc passes of `python bench.py --no-cpu-baseline`"}))
