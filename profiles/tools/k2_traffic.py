#!/usr/bin/env python3
"""HBM bytes per launch of the arc round's kernels (k_walk, k_gene_arcs_big, k_gene_arcs_wave, k_sweep<0, and k_rep_fill of the branch step) on the
12 M-hit shard, from the two PMC passes of `python bench.py ... --genomes-per-gpu 1250 --steps 1 --warmup 0` (FETCH_SIZE doubled: gfx950, see
k1_traffic.py).  The launches of the shard are told from those of the small warm-up sets by their counter values (within a factor 8 of the
largest); the mean over them is what bench.py's K2 figures are averages over as well (live lists shrink the later rounds' launches).
Every entry carries the sha256 of k_genes.hpp: bench.py reports `frac_by_counters` only while the source is the same.
usage: k2_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <hits of the shard>"""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from k1_traffic_lib import per_dispatch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hits = int(sys.argv[3])
sha = hashlib.sha256(open(os.path.join(ROOT, "pangene_amd", "csrc", "hip", "k_genes.hpp"), "rb").read()).hexdigest()[:16]
out = []
for key, pat in (("walk_scan", "k_walk<"), ("gene_arcs_big", "k_gene_arcs_big"), ("gene_arcs_wave", "k_gene_arcs_wave"), ("sweep0", "k_sweep<0"), ("rep_fill", "k_loop_front2")):  # (rep_fill: the branch step's records -- k_rep_fill inside the queued rounds' second front launch, with k_br_wave<1>)
    f, w = per_dispatch(sys.argv[1], "FETCH_SIZE", pat), per_dispatch(sys.argv[2], "WRITE_SIZE", pat)
    if not f or not w:
        continue
    tot = sorted(2 * a + b for a, b in zip(f, w))  # (per_dispatch returns the values sorted: the k-th smallest fetch belongs with the k-th smallest write only roughly -- means are what is kept)
    fbig = [x for x in f if x * 8 >= f[-1]]
    wbig = [x for x in w if x * 8 >= w[-1]]
    fk, wk = sum(fbig) / len(fbig), sum(wbig) / len(wbig)
    b = int((2 * fk + wk) * 1024)
    out.append({"kernel": key, "pattern": pat, "k_genes_sha16": sha, "hits_of_the_shard": hits, "launches": [len(fbig), len(wbig)], "fetch_kb_raw": round(fk, 1), "write_kb": round(wk, 1),
                "bytes_per_launch": b, "bytes_per_hit_of_the_shard": round(b / hits, 1)})
print(json.dumps(out, indent=1))
