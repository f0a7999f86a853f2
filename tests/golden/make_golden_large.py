#!/usr/bin/env python3
"""Reference md5s of the FULL-SIZE workloads (run in the build container, where /root/reference exists; ~10 CPU-minutes):
the inputs are too large to commit, so the tests regenerate them from the seeded generators (pangene_amd/synth.py) on the GPU
box and compare md5s.  Writes tests/golden/expected_large.json:
  human47x20k   BASELINE configs[2] stand-in  synth.human(47, 20000, iso=1.0, seed=1, frag=True)   (multi-exon K1 flavour)
  bact1250x5k   configs[3] per-GPU shard      synth.bact(1250, 5000, seed=1)                        (~12 M hits)
  human25x20k_iso5.5  configs[4] per-GPU shard  synth.human(25, 20000, iso=5.5, seed=1, frag=True), -p0 -a1 (~2.8 M hits, ~110 k proteins)
"""
import hashlib, json, os, subprocess, sys, tempfile, time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from pangene_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "pangene_ref")
SETS = {
    "human47x20k": (lambda: synth.human(47, 20000, iso=1.0, seed=1, frag=True), [[], ["-p0", "-a1"]]),
    "bact1250x5k": (lambda: synth.bact(1250, 5000, seed=1), [[]]),
    # the per-GPU shard of BASELINE configs[4] (200 assemblies x ~110 k all-isoform proteins over 8 GPUs, -p0 -a1): 25 x 20 k genes x 5.5 isoforms
    "human25x20k_iso5.5": (lambda: synth.human(25, 20000, iso=5.5, seed=1, frag=True), [["-p0", "-a1"]]),
    # BASELINE configs[3] and configs[4] at their FULL stated size (one MI355X holds either): hours of reference CPU time, the files
    # are written by parallel generator processes (a dict instead of a generator = arguments of synth.write_files_parallel)
    "bact10000x5k": (dict(kind="bact", G=10000, P=5000, seed=1), [[]]),
    "human200x20k_iso5.5": (dict(kind="human", G=200, Q=20000, iso=5.5, seed=1, frag=True), [["-p0", "-a1"]]),
}
DEFAULT = ["human47x20k", "bact1250x5k", "human25x20k_iso5.5"]


def sl(b):
    return hashlib.md5(b"\n".join(l for l in b.split(b"\n") if l[:1] in (b"S", b"L"))).hexdigest()


def main():
    out = {}
    only = sys.argv[1:] or DEFAULT
    p = os.path.join(HERE, "expected_large.json")
    if os.path.exists(p):
        out = json.load(open(p))
    for name in only:
        gen, variants = SETS[name]
        with tempfile.TemporaryDirectory(prefix="pg_large_") as td:
            fs = synth.write_files_parallel(out_dir=td, **gen) if isinstance(gen, dict) else synth.write_files(gen(), td)
            out[name] = {}
            for v in variants:
                t0 = time.time()
                r = subprocess.run([REF] + v + fs, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True)
                out[name][" ".join(v)] = {"md5": hashlib.md5(r.stdout).hexdigest(), "sl_md5": sl(r.stdout), "bytes": len(r.stdout),
                                           "n_S": sum(1 for l in r.stdout.split(b"\n") if l[:1] == b"S"), "reference_wall_s": round(time.time() - t0, 1)}
                print(name, v, out[name][" ".join(v)], flush=True)
        cur = json.load(open(p)) if os.path.exists(p) else {}          # two of these may run side by side
        cur[name] = out[name]
        with open(p, "w") as f:
            json.dump(cur, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
