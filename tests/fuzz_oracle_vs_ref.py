#!/usr/bin/env python3
"""CPU-side parity sweep: host driver + plain-C oracle backend vs the untouched reference binary (oracle/_ref/pangene_ref)
on fresh fuzz seeds x option variants x tie-order modes.  The variants include the -D/-C settings that put pg_n_local's
local_dist / local_count boundary (branch.c:31-46) inside the fuzz shapes, i.e. the H2b channel of SURVEY.md 9.1.
Prints one line per mismatch and a summary; exit code 1 on any mismatch.

    python tests/fuzz_oracle_vs_ref.py [first_seed] [n_seeds] [modes, e.g. 1,2]

Needs /root/reference to have been compiled into oracle/_ref (this container; `make -C oracle ref`)."""
import ctypes as C, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from pangene_amd import capi, synth
import oracle_host

VARIANTS = [[], ["-p0", "-a1"], ["-S"], ["-D", "300", "-C", "2"], ["-D", "1000", "-C", "1", "-p0", "-a1"], ["-D", "600", "-C", "3", "-F"],
            ["-S", "-D", "600", "-C", "3"]]
REF = os.path.join(ROOT, "oracle", "_ref", "pangene_ref")


def sweep(first, n, modes=(1, 2), variants=VARIANTS, shapes=(False, True), verbose=True):
    ora = oracle_host.load()
    C.c_int.in_dll(ora, "pg_verbose").value = 0
    bad, tot = [], 0
    with tempfile.TemporaryDirectory(prefix="pg_fuzz_ref_") as base:
        for s in range(first, first + n):
            for harsh in shapes:
                fs = synth.write_files(synth.fuzz(s, harsh=harsh), os.path.join(base, "f%d_%d" % (s, harsh)))
                for v in variants:
                    want = subprocess.run([REF] + v + fs, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
                    for mode in modes:
                        ora.pg_set_exact_mode(mode)
                        tot += 1
                        if capi.run(ora, fs, v) != want:
                            bad.append((s, harsh, tuple(v), mode))
                            if verbose:
                                print("MISMATCH seed %d harsh %d variant %r mode %d" % (s, harsh, v, mode), flush=True)
    return tot, bad


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 7000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    modes = tuple(int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 else (1, 2)
    tot, bad = sweep(first, n, modes)
    by_var = {}
    for b in bad:
        by_var[(b[2], b[3])] = by_var.get((b[2], b[3]), 0) + 1
    for k in sorted(by_var):
        print("  %-40s mode %d: %d" % (" ".join(k[0]), k[1], by_var[k]))
    print("oracle-host vs reference: %d comparisons, %d mismatches (seeds %d..%d)" % (tot, len(bad), first, first + n - 1))
    sys.exit(1 if bad else 0)
