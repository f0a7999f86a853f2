#!/usr/bin/env python3
"""Wide parity sweep on a GPU box: HIP path vs the oracle backend (same host driver) on fresh fuzz / bacterial / human-shaped /
mutated seeds and option variants, both tie-order modes.  As a script it prints one line per mismatch and a summary (exit code 1
on any); tests/test_hip_parity.py runs a bounded slice of it inside `pytest -m gpu` (test_fresh_seed_fuzz_hip_vs_oracle).
    python tests/fuzz_hip_vs_oracle.py [first_seed] [n_seeds]"""
import ctypes as C, hashlib, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from pangene_amd import capi, synth
import oracle_host

VARIANTS = [[], ["-p0", "-a1"], ["-S"], ["-F"], ["-E"], ["-b", "0.2", "-B", "0.1", "-y", "0.3"], ["--bed=flag"], ["-f", "0.2"],
            ["-D", "300", "-C", "2"], ["-D", "1000", "-C", "1", "-p0", "-a1"], ["-D", "600", "-C", "3", "-F"], ["-S", "-D", "600", "-C", "3"]]


def sets_of(s, human=True, mutated=True):
    sets = {"fuzz": synth.fuzz(s, harsh=bool(s & 1)), "bact": synth.bact(6 + s % 5, 150 + 37 * (s % 7), seed=s)}
    if human and s % 4 == 0:
        sets["human"] = synth.human(4 + s % 3, 120, iso=2.0 + (s % 3), seed=s, n_chr=3, frag=bool(s & 8))
    if mutated:  # non-positive scores, duplicates, strand flips, fs / st tags, dropped and ungrouped lines (synth.mutate)
        sets["mutfuzz"] = synth.mutate(synth.fuzz(s + 500000, harsh=bool(s & 2)), s)
        if s % 3 == 0:
            sets["mutbact"] = synth.mutate(synth.bact(5 + s % 4, 120 + 29 * (s % 5), seed=s + 1), s + 1)
    # contig coordinates beyond 32 bits (synth.widen): virtual contigs in the packer, the wide record forms on the device
    sets["widefuzz"] = synth.widen(synth.fuzz(s + 700000, harsh=bool(s & 1)), s, p_gap=0.3)
    if human and s % 4 == 2:
        sets["widehuman"] = synth.widen(synth.human(4 + s % 3, 100, iso=2.0, seed=s + 7, n_chr=3, frag=bool(s & 8)), s + 1)
    return sets


def sweep(hip, ora, first, n, base, variants=VARIANTS, modes=(1, 2), human=True, mutated=True, log=print):
    """returns (comparisons, list of mismatch descriptions)"""
    bad, tot = [], 0
    for s in range(first, first + n):
        for name, gen in sets_of(s, human, mutated).items():
            fs = synth.write_files(gen, os.path.join(base, "%s%d" % (name, s)))
            for v in variants:
                for mode in modes:
                    hip.pg_set_exact_mode(mode); ora.pg_set_exact_mode(mode)
                    a, b = capi.run(hip, fs, v), capi.run(ora, fs, v)
                    if "--bed=flag" in v and mode == 1:  # line order of --bed is only pinned in mode all
                        a, b = b"\n".join(sorted(a.split(b"\n"))), b"\n".join(sorted(b.split(b"\n")))
                    tot += 1
                    if a != b:
                        a2, b2 = capi.run(hip, fs, v), capi.run(ora, fs, v)  # which side moved?
                        if "--bed=flag" in v and mode == 1:
                            a2, b2 = b"\n".join(sorted(a2.split(b"\n"))), b"\n".join(sorted(b2.split(b"\n")))
                        dump = os.environ.get("PG_FUZZ_DUMP")
                        if dump:
                            os.makedirs(dump, exist_ok=True)
                            tag = "%s%d_v%d_m%d" % (name, s, variants.index(v), mode)
                            open(os.path.join(dump, tag + ".hip"), "wb").write(a); open(os.path.join(dump, tag + ".ora"), "wb").write(b)
                            open(os.path.join(dump, tag + ".hip2"), "wb").write(a2)
                        log("  again: hip %s its first answer, hip %s the oracle; oracle %s its first answer" % ("==" if a2 == a else "!=", "==" if a2 == b else "!=", "==" if b2 == b else "!="))
                        bad.append("seed %d set %s variant %r mode %d (%d vs %d bytes, md5 %s)" % (s, name, v, mode, len(a), len(b), hashlib.md5(a).hexdigest()[:8]))
                        log("MISMATCH " + bad[-1])
    hip.pg_set_exact_mode(1); ora.pg_set_exact_mode(1)
    return tot, bad


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    hip, ora = capi.load(), oracle_host.load()
    for lib in (hip, ora):
        C.c_int.in_dll(lib, "pg_verbose").value = 0
    tot, bad = sweep(hip, ora, first, n, tempfile.mkdtemp(prefix="pg_fuzz_"), log=lambda m: print(m, flush=True))
    print("fuzz sweep: %d comparisons, %d mismatches" % (tot, len(bad)))
    sys.exit(1 if bad else 0)
