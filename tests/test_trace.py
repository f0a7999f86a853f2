"""Step-by-step parity: with PANGENE_TRACE=<file> the host driver writes, after every step of the path (ingest, post_process,
vertices, each pg_gen_arc round, each branch-marking step, each pg_flt_high_occ), one line of hashes over the per-hit state arrays
the backend holds (file order).  The HIP backend and the oracle backend, driven over the same input, must write the same lines;
when an end-to-end md5 test goes red, the first differing line names the step -- the group of kernels -- that broke."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT, golden_files

_RUN = r'''
import sys, ctypes as C
sys.path.insert(0, %r); sys.path.insert(0, %r)
from pangene_amd import capi
import oracle_host
lib = (oracle_host.load() if sys.argv[1] == "oracle" else capi.load()); C.c_int.in_dll(lib, "pg_verbose").value = 0
lib.pg_set_exact_mode(int(sys.argv[2]))
capi.run(lib, sys.argv[4:], sys.argv[3].split())
''' % (ROOT, os.path.join(ROOT, "tests"))

FIELDS = ["flags", "rank", "score_dom", "pid_dom", "pid_dom0", "pos_x", "pos_y"]


def _trace(tmp_path, backend, mode, variant, files):
    path = str(tmp_path / ("trace_%s.txt" % backend))
    r = subprocess.run([sys.executable, "-c", _RUN, backend, str(mode), variant] + files, env=dict(os.environ, PANGENE_TRACE=path),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    rows = []
    for line in open(path):
        t = line.rstrip("\n").split("\t")
        rows.append((t[0], int(t[1]), dict(x.split("=") for x in t[2:])))
    return rows


def test_trace_of_the_oracle_backend_has_every_step(built, tmp_path):
    rows = _trace(tmp_path, "oracle", 1, "", golden_files("C4"))
    steps = [(s, r) for s, r, _ in rows]
    assert steps[:5] == [("ingest", 0), ("post_process", 0), ("gen_vtx+flag_vtx", 0), ("gen_arc", 1), ("flt_high_occ", 1)]
    assert steps.count(("gen_arc", 17)) == 1 and ("mark_branch", 17) in steps and ("flt_high_occ", 17) in steps  # 15 branch rounds: 3..17
    assert all(set(h) == set(FIELDS) for _, _, h in rows)
    assert rows == _trace(tmp_path, "oracle", 1, "", golden_files("C4"))  # and it is a function of the input


@pytest.mark.parametrize("name,variant", [("C4", ""), ("fuzz7126", "-D 300 -C 2"), ("human8f", "-p0 -a1")])
def test_round_filter_on_the_host_and_on_the_backend_agree(built, tmp_path, name, variant):
    """A branch round either leaves pg_flt_high_occ's tests to the backend (branch_decide_filter: one byte per segment comes back) or
    makes them on the host from the round's counters, degrees and n_dist_loci (PANGENE_ROUND_FILTER_HOST=1, also the route of sharded
    runs and of runs with log lines): same states after every step, with the oracle backend here and with the HIP backend in
    test_arc_round_paths_agree."""
    files = golden_files(name)
    a = _trace(tmp_path, "oracle", 1, variant, files)
    os.environ["PANGENE_ROUND_FILTER_HOST"] = "1"
    try:
        b = _trace(tmp_path, "oracle", 1, variant, files)
    finally:
        del os.environ["PANGENE_ROUND_FILTER_HOST"]
    assert a == b and len(a) > 40


@pytest.mark.gpu
@pytest.mark.parametrize("name,variant,mode", [("C4", "", 1), ("bact20", "", 1), ("human8f", "-p0 -a1", 1), ("human8", "-S", 2), ("fuzz7126", "-D 300 -C 2", 1),
                                               ("fuzz3", "-F", 0), ("dense", "", 1), ("manydoms", "", 1)])
def test_hip_trace_equals_oracle_trace(built, tmp_path, name, variant, mode):
    files = golden_files(name)
    a, b = _trace(tmp_path, "hip", mode, variant, files), _trace(tmp_path, "oracle", mode, variant, files)
    assert [(s, r) for s, r, _ in a] == [(s, r) for s, r, _ in b]
    for (step, rnd, ha), (_, _, hb) in zip(a, b):
        diff = [f for f in FIELDS if ha[f] != hb[f]]
        assert not diff, "first difference after step %s (round %d): %s" % (step, rnd, ", ".join(diff))
