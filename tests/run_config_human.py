#!/usr/bin/env python3
"""One-off parity + timing run for the human-shaped configs (BASELINE configs[2] stand-in: 47 haplotypes x ~20k genes,
multi-exon, isoforms; real HPRC PAFs are not available offline).  Needs a GPU and oracle/_ref/pangene_ref.
    python tests/run_config_human.py [G] [Q] [iso] [extra pangene options...]
Prints one JSON line: hits, md5 of the GFA from the HIP path and from the untouched reference, timings."""
import ctypes as C, hashlib, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pangene_amd import capi, synth

G = int(sys.argv[1]) if len(sys.argv) > 1 else 47
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
iso = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
opts = sys.argv[4:]
d = os.path.join(tempfile.gettempdir(), "pangene_human_%d_%d_%g" % (G, Q, iso))
t0 = time.time()
if not os.path.isdir(d):
    synth.write_files(synth.human(G, Q, iso=iso, seed=5, frag=True), d)
files = sorted(os.path.join(d, f) for f in os.listdir(d))
t_gen = time.time() - t0
lib = capi.load()
C.c_int.in_dll(lib, "pg_verbose").value = 1
t0 = time.time(); out = capi.run(lib, files, opts); t_hip = time.time() - t0
path_s, hits = lib.pg_last_path_seconds(), lib.pg_last_path_hits()
t0 = time.time(); out2 = capi.run(lib, files, opts); t_hip2 = time.time() - t0
ph = (C.c_double * 16)(); nph = lib.pg_phase_times(ph, 16)
phases = {lib.pg_phase_name(i).decode(): round(ph[i] * 1e3, 2) for i in range(nph)}
res = {"G": G, "Q": Q, "iso": iso, "opts": opts, "hits": hits, "gen_s": round(t_gen, 1), "hip_total_s": round(t_hip, 2), "hip_path_ms": round(path_s * 1e3, 1),
       "hip_path_ms_2nd": round(lib.pg_last_path_seconds() * 1e3, 1), "hip_md5": hashlib.md5(out).hexdigest(), "rerun_identical": out == out2, "host_phases_ms_2nd": phases}
ref = os.path.join(ROOT, "oracle", "_ref", "pangene_ref")
if os.path.exists(ref):
    t0 = time.time(); r = subprocess.run([ref] + opts + files, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL); t_ref = time.time() - t0
    res.update(ref_s=round(t_ref, 2), ref_md5=hashlib.md5(r.stdout).hexdigest(), identical=(r.stdout == out))
    if r.stdout != out:
        open(os.path.join(ROOT, "gpurun_out", "human_hip.gfa"), "wb").write(out); open(os.path.join(ROOT, "gpurun_out", "human_ref.gfa"), "wb").write(r.stdout)
print(json.dumps(res))
