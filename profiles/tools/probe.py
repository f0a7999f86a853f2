#!/usr/bin/env python3
"""One data set through the path, for tuning runs on a GPU box:  probe.py <shape> [--opt "-p0 -a1"] [--runs N] [--lib path] [--md5 name]
  shape   bact:G:P  |  human:G:Q:iso   (seeded synthetic sets of pangene_amd/synth.py, cached under $TMPDIR between calls on one box)
  --lib   another build of libpangene_amd.so (e.g. the -DPGA_SW_PROFILE one: pangene_amd/lib/prof/libpangene_amd.so)
  --md5   key of tests/golden/expected_large.json to compare the GFA with
Prints the wall time of stages A+B+C of every run (pg_last_path_seconds) and the md5 of the GFA."""
import argparse, ctypes as C, hashlib, json, os, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pangene_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("shape"); ap.add_argument("--opt", default=""); ap.add_argument("--runs", type=int, default=2); ap.add_argument("--lib", default=""); ap.add_argument("--md5", default="")
ap.add_argument("--exact", type=int, default=1)
a = ap.parse_args()
f = a.shape.split(":")
d = os.path.join(tempfile.gettempdir(), "pangene_probe_" + a.shape.replace(":", "_"))
if not os.path.exists(d + ".done"):
    if f[0] == "bact": synth.write_files_parallel("bact", d, G=int(f[1]), P=int(f[2]), seed=1)
    else: synth.write_files_parallel("human", d, G=int(f[1]), Q=int(f[2]), iso=float(f[3]), seed=1, frag=True)
    open(d + ".done", "w").close()
files = sorted(os.path.join(d, x) for x in os.listdir(d))
if a.lib: capi.LIB_HIP = os.path.abspath(a.lib)
lib = capi.load()
C.c_int.in_dll(lib, "pg_verbose").value = 0
lib.pg_set_exact_mode(a.exact)
for r in range(a.runs):
    t0 = time.time()
    out = capi.run(lib, files, a.opt.split())
    print("run %d: path %.2f ms (%d hits, %d attempt(s)), whole call %.2f s, md5 %s" % (r, lib.pg_last_path_seconds() * 1e3, lib.pg_last_path_hits(), lib.pg_last_attempts(), time.time() - t0, hashlib.md5(out).hexdigest()), flush=True)
if a.md5:
    e = json.load(open(os.path.join(ROOT, "tests", "golden", "expected_large.json"))).get(a.md5, {}).get(a.opt)
    print("expected (reference): %s -> %s" % (e and e["md5"], "IDENTICAL" if e and e["md5"] == hashlib.md5(out).hexdigest() else "DIFFERENT"))
