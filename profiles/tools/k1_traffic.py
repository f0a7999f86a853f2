#!/usr/bin/env python3
"""HBM bytes per launch of K1 (k_sweep<3, *>) from the two PMC passes of `python bench.py --no-cpu-baseline --no-extra-legs`, per
kernel flavour and shard size.  rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB (1 unit = 1024 B); on gfx950 FETCH_SIZE tallies
64 B per request where a wide coalesced read moves 128 B, so it is doubled (MI355X_MICROARCH.md, HBM section).  The bench launches
k_sweep<3, false> on three shard sizes (the tiny warm-up set, the bench workload, the past-L3 shard of the roofline leg) and
k_sweep<3, true> or k_sweep_lean<3> (whichever the upload's density picked) on the human-shaped shard: the dispatches are told apart by their counter values (each size is > 5x the previous
one).  Every entry carries the sha256 of k_sweep.hpp it was measured with: bench.py only reports it as `traffic` while the source
is the same.
usage: k1_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <bench.json>"""
import glob, hashlib, json, os, sqlite3, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from k1_traffic_lib import per_dispatch  # noqa: E402


def groups(vals):  # consecutive values within a factor 3 form one shard size
    out = []
    for v in vals:
        if out and v <= 3 * out[-1][0]:
            out[-1].append(v)
        else:
            out.append([v])
    return [sum(g) / len(g) for g in out], [len(g) for g in out]


b = json.load(open(sys.argv[3]))
sha = hashlib.sha256(open(os.path.join(ROOT, "pangene_amd", "csrc", "hip", "k_sweep.hpp"), "rb").read()).hexdigest()[:16]
KNAME = {"false": "k_sweep<3, false>", "true": "k_sweep<3, true>", "lean": "k_sweep_lean<3>"}
flav = lambda r: "lean" if "k_sweep_lean" in r["kernel"] else "true" if "<3, true>" in r["kernel"] else "false"
want = {"false": [b["roofline"]["hits_per_launch"]], "true": [], "lean": []}
if "also_at_bench_size" in b["roofline"]:
    want["false"].insert(0, b["roofline"]["also_at_bench_size"]["hits_per_launch"])
if b.get("human_shard") and b["human_shard"].get("roofline"):
    want[flav(b["human_shard"]["roofline"])].append(b["human_shard"]["roofline"]["hits_per_launch"])
if b.get("full_size") and b["full_size"].get("roofline"):  # bench.py --workload config3 | config4: the full-size leg behind the default workload
    fr = b["full_size"]["roofline"]
    want[flav(fr)].append(fr["hits_per_launch"])
out = []
for flavour, sizes in want.items():
    if not sizes:
        continue
    f, nf = groups(per_dispatch(sys.argv[1], "FETCH_SIZE", KNAME[flavour]))
    w, nw = groups(per_dispatch(sys.argv[2], "WRITE_SIZE", KNAME[flavour]))
    ent = []
    for k, hits in enumerate(reversed(sizes)):  # largest group = largest shard
        if k >= len(f) or k >= len(w):
            break
        fk, wk = f[-1 - k], w[-1 - k]
        tot = int((2 * fk + wk) * 1024)
        ent.append({"kernel": KNAME[flavour], "flavour": flavour, "k_sweep_sha16": sha, "hits_per_launch": hits, "fetch_kb_raw": round(fk, 1), "write_kb": round(wk, 1),
                    "bytes_per_launch": tot, "bytes_per_hit": round(tot / hits, 1), "dispatches": [nf[-1 - k], nw[-1 - k]],
                    "note": "FETCH_SIZE x2 (gfx950 correction), separate --pmc passes of `python bench.py --no-cpu-baseline --no-extra-legs --steps 2 --warmup 0`"})
    out += ent[::-1]
print(json.dumps(out, indent=1))
