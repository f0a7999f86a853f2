#!/usr/bin/env python3
"""HBM-side bytes per launch of every kernel from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same command.
rocprofv3 reports both in KB (1 unit = 1024 B); on gfx950 FETCH_SIZE tallies 64 B per request and a request is a 128-byte L2 line,
so it is doubled.  Calibrated this round with kernels of known byte counts (profiles/r04_counter_calibration.txt): x 2.000 for coalesced
reads of 4 and of 16 bytes per lane, x 1.000 for WRITE_SIZE of coalesced writes; a gather through a permutation makes one request
per L2 miss whatever the item's width, so x 2 gives the FABRIC-side bytes of a gather kernel too -- which the 256 MiB Infinity Cache
may serve: for kernels that gather (marked "gather" below) the figure is an upper bound of the HBM traffic.  A command launches each kernel on several data sizes (warm-up set, workload): only the dispatches within a factor 2 of
the kernel's largest counter value are averaged (= the launches on the large shard).
usage: pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <hits per launch> [min KB]"""
import glob, os, sqlite3, sys


def per_dispatch(root, counter):
    db = sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True))[0]
    cur = sqlite3.connect(db).cursor()
    cur.execute("SELECT * FROM pmc_events LIMIT 1")
    c = [d[0] for d in cur.description]
    kn = ([x for x in c if x in ("name", "kernel_name")] or [x for x in c if "name" in x and "counter" not in x])[0]
    cn = ([x for x in c if x in ("counter_name", "pmc_name")] or [x for x in c if "counter" in x and "name" in x])[0]
    vn = ([x for x in c if x in ("counter_value", "value")] or [x for x in c if "value" in x])[0]
    dn = ([x for x in c if x in ("dispatch_id", "event_id")] or [x for x in c if "dispatch" in x])[0]
    acc = {}
    for name, ctr, val, did in cur.execute("SELECT %s, %s, %s, %s FROM pmc_events" % (kn, cn, vn, dn)):
        if ctr == counter:
            acc[(name, did)] = acc.get((name, did), 0.0) + (val or 0.0)
    out = {}
    for (name, did), v in acc.items():
        out.setdefault(name, []).append(v)
    return out


def top(vals):
    m = max(vals)
    sel = [v for v in vals if v >= 0.5 * m]
    return sum(sel) / len(sel), len(sel)


f, w = per_dispatch(sys.argv[1], "FETCH_SIZE"), per_dispatch(sys.argv[2], "WRITE_SIZE")
hits = int(sys.argv[3])
min_kb = float(sys.argv[4]) if len(sys.argv) > 4 else 4096.0
rows = []
for name in sorted(set(f) | set(w)):
    fk, nf = top(f.get(name, [0.0]))
    wk, nw = top(w.get(name, [0.0]))
    if 2 * fk + wk < min_kb:
        continue
    rows.append((2 * fk + wk, name, fk, wk, nf, nw))
GATHER = ("k_zrec", "k_rep_fill", "k_n_local", "k_mark_hits_z", "k_gene_arcs", "OutHalfArcs", "k_to_file", "k_zpos_y", "k_pack_yrec", "rs_scatter", "k_vtx", "k_post_part", "k_genome_sort", "k_unblock")
print("# L2 <-> fabric traffic per launch on the large shard (%d hits): read = 2 x FETCH_SIZE KB, write = WRITE_SIZE KB (factors calibrated: profiles/r04_counter_calibration.txt)" % hits)
print("# 'gather': the kernel gathers / scatters through a permutation -- one 128-byte line per L2 miss, possibly served by the Infinity Cache: an UPPER bound of its HBM bytes")
print("# %-80s %10s %10s %8s %8s %6s" % ("kernel", "read MB", "write MB", "rd B/hit", "wr B/hit", "n"))
for tot, name, fk, wk, nf, nw in sorted(rows, reverse=True):
    print("%-82s %10.1f %10.1f %8.1f %8.1f %3d/%-3d %s" % (name[:82], 2 * fk / 1024, wk / 1024, 2 * fk * 1024 / hits, wk * 1024 / hits, nf, nw, "gather" if any(g in name for g in GATHER) else ""))
