#!/usr/bin/env python3
"""One-off wide parity sweep on a GPU box: HIP path vs the oracle backend (same host driver) on many fresh fuzz / bacterial /
human-shaped seeds and option variants, both tie-order modes.  Prints one line per mismatch and a summary; exit code 1 on any.
    python tests/fuzz_hip_vs_oracle.py [first_seed] [n_seeds]"""
import ctypes as C, hashlib, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pangene_amd import capi, synth

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
hip, ora = capi.load(), capi.load(oracle_host=True)
for lib in (hip, ora):
    C.c_int.in_dll(lib, "pg_verbose").value = 0
VARIANTS = [[], ["-p0", "-a1"], ["-S"], ["-F"], ["-E"], ["-b", "0.2", "-B", "0.1", "-y", "0.3"], ["--bed=flag"], ["-f", "0.2"],
            ["-D", "300", "-C", "2"], ["-D", "1000", "-C", "1", "-p0", "-a1"], ["-D", "600", "-C", "3", "-F"], ["-S", "-D", "600", "-C", "3"]]
bad = tot = 0
base = tempfile.mkdtemp(prefix="pg_fuzz_")
for s in range(first, first + n):
    sets = {"fuzz": synth.fuzz(s, harsh=bool(s & 1)), "bact": synth.bact(6 + s % 5, 150 + 37 * (s % 7), seed=s)}
    if s % 4 == 0:
        sets["human"] = synth.human(4 + s % 3, 120, iso=2.0 + (s % 3), seed=s, n_chr=3, frag=bool(s & 8))
    for name, gen in sets.items():
        fs = synth.write_files(gen, os.path.join(base, "%s%d" % (name, s)))
        for v in VARIANTS:
            for mode in (1, 2):
                hip.pg_set_exact_mode(mode); ora.pg_set_exact_mode(mode)
                a, b = capi.run(hip, fs, v), capi.run(ora, fs, v)
                if "--bed=flag" in v and mode == 1:  # line order of --bed is only pinned in mode all
                    a, b = b"\n".join(sorted(a.split(b"\n"))), b"\n".join(sorted(b.split(b"\n")))
                tot += 1
                if a != b:
                    bad += 1
                    a2, b2 = capi.run(hip, fs, v), capi.run(ora, fs, v)  # which side moved?
                    if "--bed=flag" in v and mode == 1:
                        a2, b2 = b"\n".join(sorted(a2.split(b"\n"))), b"\n".join(sorted(b2.split(b"\n")))
                    dump = os.environ.get("PG_FUZZ_DUMP")
                    if dump:
                        os.makedirs(dump, exist_ok=True)
                        tag = "%s%d_v%d_m%d" % (name, s, VARIANTS.index(v), mode)
                        open(os.path.join(dump, tag + ".hip"), "wb").write(a); open(os.path.join(dump, tag + ".ora"), "wb").write(b)
                        open(os.path.join(dump, tag + ".hip2"), "wb").write(a2)
                    print("  again: hip %s its first answer, hip %s the oracle; oracle %s its first answer" % ("==" if a2 == a else "!=", "==" if a2 == b else "!=", "==" if b2 == b else "!="), flush=True)
                    print("MISMATCH seed %d set %s variant %r mode %d (%d vs %d bytes, md5 %s)" % (s, name, v, mode, len(a), len(b), hashlib.md5(a).hexdigest()[:8]), flush=True)
print("fuzz sweep: %d comparisons, %d mismatches" % (tot, bad))
sys.exit(1 if bad else 0)
