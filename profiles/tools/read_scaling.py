#!/usr/bin/env python3
"""Batch PAF reader on the GPU box's host: wall time and rate by thread count, with and without the huge-page arena.
    python profiles/tools/read_scaling.py [genomes] [proteins]        (one process per setting: the switches are read once)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, os, time, ctypes as C
sys.path.insert(0, %r)
from pangene_amd import capi
d = sys.argv[1]
fs = sorted(os.path.join(d, f) for f in os.listdir(d))
lib = capi.load()
C.c_int.in_dll(lib, "pg_verbose").value = 0
lib.pg_device_warm()
best = None
for rep in range(3):
    opt = capi.parse_args(lib, [])
    dd = lib.pg_data_init()
    t = time.time(); capi.read_files(lib, opt, dd, fs); dt = time.time() - t
    best = dt if best is None or dt < best else best
    lib.pg_data_destroy(dd)
print("RESULT %%s threads=%%s arena=%%s best_of_3 %%.3f s" %% (os.environ.get("TAG", ""), os.environ.get("PANGENE_READ_THREADS", "default"), "no" if os.environ.get("PANGENE_NO_ARENA") else "yes", best))
''' % ROOT
if __name__ == "__main__":
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    sys.path.insert(0, ROOT)
    from pangene_amd import synth
    d = "/tmp/pg_read_scaling_%dx%d" % (G, P)
    if not os.path.isdir(d):
        synth.write_files_parallel("bact", d, G=G, P=P, seed=11)
    nb = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d))
    print("# %d files, %.2f GB of PAF text" % (len(os.listdir(d)), nb * 1e-9), flush=True)
    for arena in (True, False):
        for th in ("8", "16", "32", "64") if arena else ("16",):
            env = dict(os.environ, PANGENE_READ_THREADS=th, PANGENE_TIMING="1")
            if not arena:
                env["PANGENE_NO_ARENA"] = "1"
            r = subprocess.run([sys.executable, "-c", CHILD, d], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            err = [l for l in r.stderr.decode().split("\n") if l.startswith("[pg_read_paf_batch]")]
            print(r.stdout.decode().strip().split("\n")[-1] if r.returncode == 0 else "FAILED: " + r.stderr.decode()[-500:], flush=True)
            if err:
                print("   " + err[-1], flush=True)
