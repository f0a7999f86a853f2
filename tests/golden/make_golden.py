#!/usr/bin/env python3
"""Generates the committed parity fixtures (run in the build container, where /root/reference exists).

Inputs : tests/golden/<set>/*.paf.gz  -- C4 is the reference's own test/C4 data; the others come from the
         seeded generators in pangene_amd/synth.py.
Outputs: tests/golden/expected.json   -- md5 of what the UNTOUCHED reference (oracle/_ref/pangene_ref, built
         by oracle/Makefile from /root/reference) prints for every (set, option variant); for --bed outputs
         both the md5 of the bytes and of the sorted lines (line order exposes the unstable sort);
         tests/golden/<set>.gfa.gz  -- the default-option GFA itself, for debugging.
"""
import gzip, hashlib, json, os, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from pangene_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "pangene_ref")
VARIANTS = [[], ["-p0", "-a1"], ["-a2"], ["-E"], ["-J"], ["-S"], ["-F"], ["--ori-sc"], ["-w"], ["-a2", "-E"], ["-D", "300", "-C", "2"], ["-G"],
            ["-c", "3", "-g", "6"], ["-T", "3"], ["-f", "0.2"], ["-e", "0.9", "-l", "0.8"], ["-m", "0.5"],
            ["-b", "0.2", "-B", "0.1", "-y", "0.3"], ["-b", "0.01", "-r", "1"],
            # pg_n_local's local_dist / local_count boundary inside the fixture shapes (cs ties among walkable hits)
            ["-D", "1000", "-C", "1", "-p0", "-a1"], ["-D", "600", "-C", "3", "-F"], ["-S", "-D", "600", "-C", "3"],
            ["--bed=raw"], ["--bed=flag"], ["--bed=walk"]]
SETS = {
    "bact20": lambda: synth.bact(20, 500, seed=1),
    "human8": lambda: synth.human(8, 300, iso=3.0, seed=1, n_chr=6),
    "human8f": lambda: synth.human(8, 300, iso=3.0, seed=2, n_chr=4, frag=True),
    "dense": lambda: synth.dense(3),          # giant spanning hit + pile-ups: the sweep's slow-list / overflow paths
    "manydoms": lambda: synth.many_doms(2),   # > 8 dominators per gene: the spill path of the vertex fold
}
for s in range(6):
    SETS["fuzz%d" % s] = (lambda s=s: synth.fuzz(s, harsh=(s % 2 == 0)))
# seeds on which the default tie-order mode differed from the reference before H2b became a trigger
SETS["fuzz7115"] = lambda: synth.fuzz(7115, harsh=False)
SETS["fuzz7115h"] = lambda: synth.fuzz(7115, harsh=True)
SETS["fuzz7126"] = lambda: synth.fuzz(7126, harsh=False)
# what the generators never emit on their own: non-positive scores, duplicated alignments, strand flips, fs / st
# tags, dropped lines, lines not grouped by protein
SETS["mut0"] = lambda: synth.mutate(synth.fuzz(100, harsh=False), 1)
SETS["mut1"] = lambda: synth.mutate(synth.bact(12, 200, seed=5), 2)
SETS["mut2"] = lambda: synth.mutate(synth.human(6, 150, iso=2.0, seed=3, n_chr=3, frag=True), 3)
# contig coordinates beyond 32 bits (pangene.h:71: int64_t cs, cm, ce): the device layout keeps 32-bit planes and cuts such contigs into
# virtual contigs (include/pangene_hip.h, pga_genome_block_t)
SETS["wide0"] = lambda: synth.widen(synth.human(8, 300, iso=3.0, seed=4, n_chr=5), 1)
SETS["wide1"] = lambda: synth.widen(synth.bact(12, 300, seed=7), 2, p_gap=0.02)
SETS["wide2"] = lambda: synth.widen(synth.fuzz(3, harsh=True), 3, p_gap=0.3)
SETS["wide3"] = lambda: synth.widen(synth.mutate(synth.human(6, 200, iso=2.0, seed=9, n_chr=3, frag=True), 4), 4, p_gap=0.3)


def files_of(name):
    d = os.path.join(HERE, name)
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".paf.gz") or f.endswith(".paf"))


def main():
    for name, gen in SETS.items():
        d = os.path.join(HERE, name)
        if not os.path.isdir(d):
            synth.write_files(gen(), d, gz=True)
    # option lists used by the -X/-I/-P variants
    extra = {"human8": [["-X", "G00003,G00007:T1"], ["-I", "G00010", "-P", "G00020,G00021"]],
             "C4": [["-P", "C4B"], ["-X", "CYP21A2"], ["-d", "."]]}
    exp = {}
    for name in ["C4"] + list(SETS):
        fs = files_of(name)
        exp[name] = {}
        for v in VARIANTS + extra.get(name, []):
            r = subprocess.run([REF] + v + fs, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True)
            key = " ".join(v)
            ent = {"md5": hashlib.md5(r.stdout).hexdigest(), "bytes": len(r.stdout)}
            if any(x.startswith("--bed") for x in v):
                ent["md5_sorted"] = hashlib.md5(b"\n".join(sorted(r.stdout.split(b"\n")))).hexdigest()
            exp[name][key] = ent
            if not v:
                with open(os.path.join(HERE, name + ".gfa.gz"), "wb") as f:
                    f.write(gzip.compress(r.stdout, mtime=0))  # mtime 0: regenerating does not change the bytes
    with open(os.path.join(HERE, "expected.json"), "w") as f:
        json.dump(exp, f, indent=1, sort_keys=True)
    print("wrote expected.json for", len(exp), "sets")


if __name__ == "__main__":
    main()
