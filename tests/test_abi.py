"""C-ABI checks that need no GPU: struct layouts equal the reference's pangene.h, and the product
library loads and exports every symbol that include/*.h declares."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import pytest

from conftest import ROOT, golden_files
import oracle_host  # tests/oracle_host.py: the checker build of the host driver

SIZES = {"pg_opt_t": 128, "pg_hit_t": 88, "pg_exon_t": 8, "pg_prot_t": 32, "pg_gene_t": 16, "pg_ctg_t": 16, "pg_genome_t": 56,
         "pg_data_t": 72, "pg_seg_t": 32, "pg_arc_t": 32, "pg_graph_t": 56, "pg128_t": 16, "pga_arc_part_t": 40}


def test_struct_sizes_and_offsets():
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "pangene_amd.h"\n#include "pangene_hip.h"\nint main(void){\n'
    for k in SIZES:
        src += 'printf("%s %%zu\\n", sizeof(%s));\n' % (k, k)
    src += 'printf("cs %zu cm %zu ce %zu\\n", offsetof(pg_hit_t, cs), offsetof(pg_hit_t, cm), offsetof(pg_hit_t, ce));\n'
    src += 'pg_hit_t h; unsigned *w = (unsigned*)((char*)&h + 60); *w = 0; h.weak_br = 3; h.rev = 1; h.shadow = 1; printf("bits %u\\n", *w);\nreturn 0;}\n'
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "a.c")
        open(c, "w").write(src)
        subprocess.run(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), c, "-o", os.path.join(td, "a")], check=True)
        out = subprocess.run([os.path.join(td, "a")], stdout=subprocess.PIPE, check=True, text=True).stdout.split("\n")
    got = dict(l.split()[:2] for l in out if l and l.split()[0] in SIZES)
    assert {k: int(v) for k, v in got.items()} == SIZES
    assert "cs 64 cm 72 ce 80" in out
    assert "bits %d" % (0x1 | 0x80 | 0x600) in out  # bit positions of pangene.h:70 == PGA_F_*


def _declared(header, prefix_re):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(prefix_re, txt)))


def test_product_library_exports_every_declared_symbol(built):
    lib = C.CDLL(os.path.join(ROOT, "pangene_amd", "lib", "libpangene_amd.so"))
    names = _declared("pangene_amd.h", r"\b(pg_[a-z_0-9]+)\s*\(")
    names = [n for n in names if n not in ("pg_exchange_t",)]
    pfx = _declared("pangene_hip.h", r"pfx##_([a-z_]+)\(")
    names += ["pga_" + n for n in pfx] + ["pga_backend", "pga_set_stream", "pga_timing_reset", "pga_timing_get"]
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # pg_verbose is exported as data with the reference's default (sys.c: 3) -- asked in a fresh process: the variable is the process's,
    # and tests that ran before this one in the same interpreter may have set it
    code = "import ctypes as C; print(C.c_int.in_dll(C.CDLL(%r), 'pg_verbose').value)" % os.path.join(ROOT, "pangene_amd", "lib", "libpangene_amd.so")
    assert subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, check=True).stdout.split()[-1] == b"3"


def test_oracle_exports_the_same_abi(built):
    lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    pfx = _declared("pangene_hip.h", r"pfx##_([a-z_]+)\(")
    missing = [n for n in pfx if not hasattr(lib, "pgo_" + n)]
    assert not missing, missing


def test_product_fails_loudly_without_gpu(built):
    """No CPU fallback: on a machine without a GPU the path reports an error instead of computing."""
    import torch
    if torch.cuda.is_available():
        return
    from pangene_amd import capi
    from conftest import golden_files
    lib = capi.load()
    C.c_int.in_dll(lib, "pg_verbose").value = 0
    try:
        capi.run(lib, golden_files("C4"), [])
    except RuntimeError as e:
        assert "backend status" in str(e)
    else:
        raise AssertionError("the product path produced output without a GPU")


def test_reference_main_c_builds_and_runs_on_this_library(built, expected, tmp_path):
    """Drop-in at source level: the reference's own, unmodified main.c (compiled where it lies) links against this library's
    pangene.h-compatible surface and, on top of the host driver + oracle backend, prints the reference's GFA for test/C4."""
    import hashlib, subprocess
    ref_dir = "/root/reference"
    if not os.path.exists(os.path.join(ref_dir, "main.c")):
        pytest.skip("reference sources not present (GPU box)")
    shim = tmp_path / "shim"
    shim.mkdir()
    (shim / "pgpriv.h").write_text('#include "pangene_amd.h"\n')
    exe = str(tmp_path / "pangene_main_c")
    lib_dir = os.path.join(ROOT, "tests", "_build")
    r = subprocess.run(["gcc", "-std=c99", "-O2", "-I", str(shim), "-I", os.path.join(ROOT, "include"), "-I", ref_dir, os.path.join(ref_dir, "main.c"),
                        "-o", exe, "-L", lib_dir, "-lpangene_oraclehost", "-Wl,-rpath," + lib_dir, "-lm"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    env = dict(os.environ, PANGENE_EXACT="all")
    out = subprocess.run([exe] + golden_files("C4"), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, check=True).stdout
    assert hashlib.md5(out).hexdigest() == expected["C4"][""]["md5"]
    out = subprocess.run([exe, "-p0", "-a1"] + golden_files("C4"), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, check=True).stdout
    assert hashlib.md5(out).hexdigest() == expected["C4"]["-p0 -a1"]["md5"]


def test_device_layout_limits_are_reported_not_asserted(tmp_path):
    """include/pangene_hip.h documents the device layout limits.  Contig coordinates beyond 2^31 (pangene.h:71 has int64) are NOT one
    of them any more: the packer cuts such contigs into virtual contigs and the GFA is the reference's (the wide* fixtures; here a
    3 Gb contig).  What remains out of reach is ONE cluster of overlapping hits that spans more than 2^31 bp: that must come back as
    PGA_ERR_RANGE from the packing step -- on any backend, before a device is touched -- and as pg_last_error() != 0 with an empty
    graph, never as an abort or a silent truncation."""
    import ctypes as C
    from pangene_amd import capi
    lib = oracle_host.load()
    C.c_int.in_dll(lib, "pg_verbose").value = 0
    big = 3_000_000_000
    p = tmp_path / "big.paf"
    p.write_text("g1\t100\t0\t100\t+\tc1\t%d\t%d\t%d\t300\t300\t0\tms:i:400\tcg:Z:100M\n" % (big + 1000, big, big + 300)
                 + "g2\t100\t0\t100\t+\tc1\t%d\t10\t310\t300\t300\t0\tms:i:410\tcg:Z:100M\n" % (big + 1000))
    out = capi.run(lib, [str(p)], ["-p0", "-a1"])
    assert lib.pg_last_error() == 0
    # what oracle/_ref/pangene_ref prints for this file (the distance 2 999 999 990 goes through graph.c:73's int32_t and the
    # double -> int32_t conversion of graph.c:141)
    assert out == (b"S\tg1\t*\tLN:i:100\tng:i:1\tnc:i:1\tc1:i:1\tc2:i:0\tpp:Z:g1\nS\tg2\t*\tLN:i:100\tng:i:1\tnc:i:1\tc1:i:1\tc2:i:0\tpp:Z:g2\n"
                   b"L\tg1\t-\tg2\t-\t0M\tng:i:1\tnc:i:1\tad:i:-2147483647\ts1:i:400\ts2:i:410\nL\tg2\t+\tg1\t+\t0M\tng:i:1\tnc:i:1\tad:i:-2147483647\ts1:i:410\ts2:i:400\n"
                   b"W\tbig\t0\tc1\t*\t*\t>g2>g1\tlf:B:i,0,0\n")
    # three hits of 1e9 bp each (one huge intron), each overlapping the next: a cluster of 2.4e9 bp that no cut can divide
    span = 1_000_000_000
    lines = ""
    for k in range(3):
        st = 10 + k * 700_000_000
        lines += "h%d\t100\t0\t100\t+\tc1\t%d\t%d\t%d\t300\t300\t0\tms:i:400\tcg:Z:50M%dN50M\n" % (k, 4 * span, st, st + span, span - 300)
    p.write_text(lines)
    with pytest.raises(RuntimeError, match="status -2"):
        capi.run(lib, [str(p)], [])
    assert lib.pg_last_error() == -2
    q = tmp_path / "ok.paf"  # and the library is usable afterwards
    q.write_text("g1\t100\t0\t100\t+\tc1\t1000\t10\t310\t300\t300\t0\tms:i:400\tcg:Z:100M\n")
    assert capi.run(lib, [str(q)], []).startswith(b"S\tg1")


def _in_place_edit_runs(lib, tmp_path, tag=""):
    """(base, after bump_score, after move_exon): --bed=raw of human8f, with one field of one record edited between pg_read_paf and pg_post_process"""
    import ctypes as C
    from pangene_amd import capi

    class Hit(C.Structure):
        _fields_ = [("pid", C.c_int32), ("qs", C.c_int32), ("qe", C.c_int32), ("cid", C.c_int32), ("mlen", C.c_int32), ("blen", C.c_int32), ("lof", C.c_int32),
                    ("rank", C.c_int32), ("score_ori", C.c_int32), ("score_adj", C.c_int32), ("score_dom", C.c_int32), ("n_exon", C.c_int32), ("off_exon", C.c_int32),
                    ("pid_dom", C.c_int32), ("pid_dom0", C.c_int32), ("bits", C.c_uint32), ("cs", C.c_int64), ("cm", C.c_int64), ("ce", C.c_int64)]

    class Genome(C.Structure):
        _fields_ = [("n_ctg", C.c_int32), ("m_ctg", C.c_int32), ("ctg", C.c_void_p), ("n_hit", C.c_int32), ("m_hit", C.c_int32), ("hit", C.POINTER(Hit)),
                    ("n_exon", C.c_int32), ("m_exon", C.c_int32), ("exon", C.POINTER(C.c_int32)), ("label", C.c_void_p)]

    class Data(C.Structure):
        _fields_ = [("d_ctg", C.c_void_p), ("d_gene", C.c_void_p), ("d_prot", C.c_void_p), ("n_genome", C.c_int32), ("m_genome", C.c_int32), ("genome", C.POINTER(Genome)),
                    ("n_gene", C.c_int32), ("m_gene", C.c_int32), ("gene", C.c_void_p), ("n_prot", C.c_int32), ("m_prot", C.c_int32), ("prot", C.c_void_p)]

    assert C.sizeof(Hit) == 88 and C.sizeof(Genome) == 56 and C.sizeof(Data) == 72
    C.c_int.in_dll(lib, "pg_verbose").value = 0
    files = golden_files("human8f")
    n_run = [0]

    def run(edit):
        opt = capi.parse_args(lib, ["--bed=raw"])
        n_run[0] += 1
        out = str(tmp_path / ("o%s%d.bed" % (tag, n_run[0])))
        lib.pg_set_output(out.encode())
        d = lib.pg_data_init()
        try:
            capi.read_files(lib, opt, d, files)
            if edit:
                edit(C.cast(d, C.POINTER(Data)).contents)
            lib.pg_post_process(C.byref(opt), d)
            assert lib.pg_last_error() == 0
            lib.pg_write_bed(d, 0)
        finally:
            lib.pg_data_destroy(d)
            lib.pg_set_output(None)
        return open(out, "rb").read()

    def bump_score(dd):  # a hit of a multi-isoform gene loses most of its score: it is no longer the isoform that survives
        g = dd.genome[1]
        assert g.n_hit > 300
        g.hit[3].score_adj = 1

    def move_exon(dd):  # the first exon of hit 5 of genome 2 shrinks to one base: CDS lengths and overlaps change
        g = dd.genome[2]
        h = g.hit[5]
        g.exon[2 * h.off_exon + 1] = g.exon[2 * h.off_exon] + 1

    return run(None), run(bump_score), run(move_exon)


@pytest.mark.gpu
def test_in_place_edit_is_seen_by_the_device_path(built, tmp_path):
    """The same on the HIP backend (round 6: the signature check runs WHILE the blocks travel to the device, and a stale pack costs a
    second upload): byte for byte what the checker build prints for the same edits."""
    from pangene_amd import capi
    want = _in_place_edit_runs(oracle_host.load(), tmp_path, "c")
    got = _in_place_edit_runs(capi.load(), tmp_path, "g")
    assert want[0] != want[1] and want[0] != want[2]
    assert got == want


@pytest.mark.gpu
def test_forty_thousand_rank0_hits_of_one_protein_are_counted_exactly(built, tmp_path):
    """Stage B's per-protein sums through LDS (k_post_part_lds) keep a workgroup's two counts in the halves of one 32-bit word and empty a half into
    the global table when it reaches 2^15: 40 000 rank-0 hits of ONE protein in one workgroup's stretch (ranks edited in place, as the public
    pg_data_t allows) must give what the checker build prints.  tests/support/rank0_flood.py, in processes of their own (PANGENE_POST is read once)."""
    md5 = {}
    for which in ("oracle", "hip"):
        env = dict(os.environ, PANGENE_POST="lds")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "support", "rank0_flood.py"), which, str(tmp_path)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        md5[which] = [ln for ln in r.stdout.splitlines() if ln.startswith("md5 ")][-1]
    assert md5["hip"] == md5["oracle"]


def test_in_place_edit_between_read_and_post_process_is_seen(built, tmp_path):
    """pg_read_paf packs every genome for the device while the next file is parsed; the reference reads g->hit at pg_post_process time
    (graph.c:7-32), so a caller may edit the public pg_data_t in between.  The pack carries a signature over EVERY record it was made
    from (round 4 sampled every 257th): an edit of one field of one hit -- here hit 3's score_adj, then an exon boundary -- must give
    the output of a run that read the edited values, not the stale pack's."""
    base, a, b = _in_place_edit_runs(oracle_host.load(), tmp_path)
    assert a != base and b != base and a != b
    assert len(a.split(b"\n")) == len(base.split(b"\n")) == len(b.split(b"\n"))  # (the same hits, other flags and scores)
