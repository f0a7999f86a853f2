"""pangene.js gfa2matrix (reference pangene.js:1168-1247).  pg_write_matrix (the per-hit reduction runs on the
backend: the plain-C oracle here, the HIP kernels in the -m gpu twin) and pg_gfa2matrix_file (text route) against the
plain-Python restatement in oracle/gfa2matrix_ref.py and against entries of the reference's test/C4 data counted by hand."""
import ctypes as C
import gzip
import os
import sys

import pytest

from conftest import ROOT, GOLD, golden_files
import oracle_host  # tests/oracle_host.py: the checker build of the host driver
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


@pytest.mark.parametrize("name,variant", [("C4", ""), ("bact20", ""), ("human8f", "-p0 -a1"), ("fuzz3", "-S"), ("manydoms", ""), ("wide0", ""), ("wide3", "-S")])
def test_matrix_from_memory_and_from_file_equal_the_restatement(built, tmp_path, name, variant):
    lib = oracle_host.load()
    C.c_int.in_dll(lib, "pg_verbose").value = 0
    _matrix_checks(lib, tmp_path, name, variant)


@pytest.mark.gpu
@pytest.mark.parametrize("name,variant", [("C4", ""), ("bact20", ""), ("human8f", "-p0 -a1"), ("human8", "-S"), ("wide0", ""), ("wide3", "-S")])
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
    lib = oracle_host.load()
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


# ---- anchors on data the reference itself holds (test/bubble/*.gfa, copied to tests/golden/bubble/): expected matrices derived by
# hand from pangene.js:1168-1247.  k8 is not in this image, so the script itself never ran here: the restatement stays "unpinned
# against the script", but it no longer rests on one data set.
BUBBLE = os.path.join(GOLD, "bubble")


def _file_matrix(lib, tmp_path, gfa, cn, clstr=None, print_cd=False):
    out = tmp_path / ("m_%s_%d_%d_%d.txt" % (gfa, cn, clstr is not None, print_cd))
    lib.pg_set_output(str(out).encode())
    rc = lib.pg_gfa2matrix_file(os.path.join(BUBBLE, gfa).encode(), 1 if cn else 0, os.path.join(BUBBLE, clstr).encode() if clstr else None, 1 if print_cd else 0)
    lib.pg_set_output(None)
    assert rc == 0
    got = out.read_text()
    with open(os.path.join(BUBBLE, gfa)) as f:
        lines = f.read().split("\n")
    cl = open(os.path.join(BUBBLE, clstr)).read().split("\n") if clstr else None
    assert got == ref.gfa2matrix(lines, copy_number=cn, clstr_lines=cl, print_cd=print_cd)  # the Python restatement agrees
    return got


@pytest.fixture(scope="module")
def ora_lib(built):
    lib = oracle_host.load()
    C.c_int.in_dll(lib, "pg_verbose").value = 0
    return lib


def test_bubble_graphs_without_walks(ora_lib, tmp_path):
    """test/bubble/t1-1.gfa and t2-1.gfa have S- and L-lines only: no assembly, so `print('Gene', asm_a.join("\\t"))` prints the
    header with an empty second field and every row is the name and an empty field (segments in S-line order)"""
    assert _file_matrix(ora_lib, tmp_path, "t1-1.gfa", False) == "Gene\t\nCAPNS2\t\nCES1\t\nCES5A\t\nGNAO1\t\nSLC6A2\t\n"
    assert _file_matrix(ora_lib, tmp_path, "t2-1.gfa", True) == "Gene\t\n" + "".join("s%d\t\n" % i for i in range(1, 8))


def test_bubble_graph_with_hand_written_walks(ora_lib, tmp_path):
    """test/bubble/t1-8c.gfa (10 segments) + five W-lines written for this test (tests/golden/bubble/t1-8c.walks.gfa): assemblies in
    first-seen order A#1, A#2, B#1; counts read off the walks -- ETDB is stepped on twice by A#1 (once per contig), CT45A1 three times
    by B#1 (a self-loop walked three times)"""
    rows = ["RTL8C", "CT55", "ETDB", "ETDC", "INTS6L", "RTL8A", "SMIM10L2B", "ZNF449", "ZNF75D", "CT45A1"]
    cnt = {"RTL8C": (1, 1, 0), "CT55": (1, 0, 1), "ETDB": (2, 0, 1), "ETDC": (1, 0, 0), "INTS6L": (0, 0, 1), "RTL8A": (1, 1, 0), "SMIM10L2B": (1, 0, 1),
           "ZNF449": (1, 0, 0), "ZNF75D": (1, 0, 0), "CT45A1": (0, 0, 3)}

    def table(vals, names):
        return "Gene\tA#1\tA#2\tB#1\n" + "".join("%s\t%d\t%d\t%d\n" % ((n,) + tuple(vals[n])) for n in names)
    assert _file_matrix(ora_lib, tmp_path, "t1-8c.walks.gfa", True) == table(cnt, rows)
    assert _file_matrix(ora_lib, tmp_path, "t1-8c.walks.gfa", False) == table({k: tuple(min(1, x) for x in v) for k, v in cnt.items()}, rows)
    # -d: RTL8A is merged into RTL8C and ETDC into ETDB (their rows disappear, their counts are added BEFORE the clamp); NOPE is not in the graph
    merged = dict(cnt, RTL8C=(2, 2, 0), ETDB=(3, 0, 1))
    left = [n for n in rows if n not in ("RTL8A", "ETDC")]
    assert _file_matrix(ora_lib, tmp_path, "t1-8c.walks.gfa", True, "t1-8c.clstr") == table(merged, left)
    assert _file_matrix(ora_lib, tmp_path, "t1-8c.walks.gfa", False, "t1-8c.clstr") == table({k: tuple(min(1, x) for x in v) for k, v in merged.items()}, left)
    assert _file_matrix(ora_lib, tmp_path, "t1-8c.walks.gfa", False, "t1-8c.clstr", print_cd=True) == "RTL8A\tRTL8C\nETDC\tETDB\nNOPE\tETDB\n"


def test_cluster_file_corner_cases(ora_lib, tmp_path):
    """(a) `for (const g in paralog)` visits the integer-like key "10" before "A" although "A" was inserted first: B gets the ORIGINAL
    row of 10 (1, 1), then 10 gets A's (2, 0) -- in insertion order B would have received (3, 1).  (b) the greedy `(\\S+)\\.\\.\\.` takes
    the LAST `...` of the token: the member is named "G1...x:q", whose gene "G1...x" is not a segment, so Z is merged into nothing and
    G1 keeps its row.  (c) a line that does not match the pattern is skipped."""
    assert _file_matrix(ora_lib, tmp_path, "intkeys.gfa", True, "intkeys.clstr") == "Gene\ts#1\tt#1\nB\t2\t1\nG1\t0\t1\n"
    assert _file_matrix(ora_lib, tmp_path, "intkeys.gfa", False, "intkeys.clstr", print_cd=True) == "A\t10\n10\tB\nZ\tG1...x\n"
