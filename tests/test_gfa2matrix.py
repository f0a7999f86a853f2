"""pangene.js gfa2matrix (reference pangene.js:1168-1247).  pg_write_matrix (the per-hit reduction runs on the
backend: the plain-C oracle here, the HIP kernels in the -m gpu twin) and pg_gfa2matrix_file (text route) against the
plain-Python restatement in oracle/gfa2matrix_ref.py and against entries of the reference's test/C4 data counted by hand."""
import ctypes as C
import gzip
import os
import sys

import pytest

from conftest import ROOT, GOLD, golden_files
from pangene_amd import capi

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gfa2matrix_ref as ref  # noqa: E402


def _matrix_checks(lib, tmp_path, name, variant):
    fs = golden_files(name)
    gfa = capi.run(lib, fs, variant.split())
    gfa_lines = gfa.decode().split("\n")
    for cn in (False, True):
        want = ref.gfa2matrix(gfa_lines, copy_number=cn)
        got = capi.run(lib, fs, variant.split() + (["--matrix=count"] if cn else ["--matrix"])).decode()
        assert got == want and got.startswith("Gene\t")
        p = tmp_path / ("g%d.gfa" % cn)
        p.write_bytes(gfa)
        out = tmp_path / ("m%d.txt" % cn)
        lib.pg_set_output(str(out).encode())
        assert lib.pg_gfa2matrix_file(str(p).encode(), 1 if cn else 0, None, 0) == 0
        lib.pg_set_output(None)
        assert out.read_text() == want


@pytest.mark.parametrize("name,variant", [("C4", ""), ("bact20", ""), ("human8f", "-p0 -a1"), ("fuzz3", "-S"), ("manydoms", "")])
def test_matrix_from_memory_and_from_file_equal_the_restatement(built, tmp_path, name, variant):
    lib = capi.load(oracle_host=True)
    C.c_int.in_dll(lib, "pg_verbose").value = 0
    _matrix_checks(lib, tmp_path, name, variant)


@pytest.mark.gpu
@pytest.mark.parametrize("name,variant", [("C4", ""), ("bact20", ""), ("human8f", "-p0 -a1"), ("human8", "-S")])
def test_matrix_on_the_gpu(built, tmp_path, name, variant):
    lib = capi.load()
    C.c_int.in_dll(lib, "pg_verbose").value = 0
    _matrix_checks(lib, tmp_path, name, variant)


def test_c4_entries_counted_by_hand():
    """the reference's test/C4 graph (tests/golden/C4.gfa.gz = its output on its own data): 7 genes x 33 assemblies.  Read off the
    W-lines: bonobo's and gorilla's walks pass CYP21A2 twice, every other gene once; GRCh38 has each gene once."""
    with gzip.open(os.path.join(GOLD, "C4.gfa.gz"), "rt") as f:
        lines = f.read().split("\n")
    m = [l.split("\t") for l in ref.gfa2matrix(lines, copy_number=True).split("\n") if l]
    assert m[0][0] == "Gene" and len(m[0]) == 1 + 33 and [r[0] for r in m[1:]] == ["DXO", "STK19", "C4A", "C4B", "CYP21A2", "TNXB", "ATF6B"]
    col = {a: i for i, a in enumerate(m[0])}
    row = {r[0]: r for r in m[1:]}
    assert row["CYP21A2"][col["bonobo#0"]] == "2" and row["CYP21A2"][col["gorilla#0"]] == "2" and row["CYP21A2"][col["GRCh38#0"]] == "1"
    assert all(row[g][col["GRCh38#0"]] == "1" for g in row) and row["C4B"][col["bonobo#0"]] == "1"
    p = [l.split("\t") for l in ref.gfa2matrix(lines, copy_number=False).split("\n") if l]
    assert {x for r in p[1:] for x in r[1:]} <= {"0", "1"}


def test_cluster_file_merges_paralogs(built, tmp_path):
    """-d: the members of a CD-HIT cluster are added to its representative and not printed (pangene.js:1198-1234)"""
    lib = capi.load(oracle_host=True)
    gfa = os.path.join(GOLD, "C4.gfa.gz")
    cl = tmp_path / "x.clstr"
    cl.write_text(">Cluster 0\n0\t1744aa, >C4A:ENSP1... *\n1\t1744aa, >C4B:ENSP2... at 99.43%\n>Cluster 1\n0\t500aa, >DXO:P... *\n")
    with gzip.open(gfa, "rt") as f:
        lines = f.read().split("\n")
    for cn in (0, 1):
        want = ref.gfa2matrix(lines, copy_number=bool(cn), clstr_lines=cl.read_text().split("\n"))
        out = tmp_path / ("o%d" % cn)
        lib.pg_set_output(str(out).encode())
        assert lib.pg_gfa2matrix_file(gfa.encode(), cn, str(cl).encode(), 0) == 0
        lib.pg_set_output(None)
        assert out.read_text() == want and "\nC4B\t" not in want and "\nC4A\t" in want
    rows = {l.split("\t")[0]: l.split("\t")[1:] for l in ref.gfa2matrix(lines, True, cl.read_text().split("\n")).split("\n") if l}
    assert rows["C4A"][0] == "2"  # GRCh38: C4A + C4B
