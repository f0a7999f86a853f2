"""The batch PAF reader (reference read.c:107-236 called once per file, main.c:121-122) against per-file reads.

pg_read_paf_batch parses files on threads, resolves names against frozen snapshots of the dictionaries, leaves to the sequential
commit only what a file is the FIRST to bring (new names; attribute values that differ from the snapshot; its own entries for ids
somebody changed since the snapshot -- the change log), and puts the hit / exon arrays into one huge-page arena.  Whatever the
thread count, the state it leaves -- ids, names, every attribute of every gene and protein, every field of every hit and exon --
must be exactly what n sequential pg_read_paf calls leave."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from pangene_amd import capi, synth
import oracle_host  # tests/oracle_host.py: the checker build of the host driver


class Prot(C.Structure):
    _fields_ = [("name", C.c_char_p), ("len", C.c_int32), ("gid", C.c_int32), ("rep", C.c_int32), ("n", C.c_int32), ("avg", C.c_int32), ("mx", C.c_int32)]


class Gene(C.Structure):
    _fields_ = [("name", C.c_char_p), ("bits", C.c_uint32), ("rep_pid", C.c_int32)]


class Ctg(C.Structure):
    _fields_ = [("name", C.c_char_p), ("len", C.c_int64)]


class Genome(C.Structure):
    _fields_ = [("n_ctg", C.c_int32), ("m_ctg", C.c_int32), ("ctg", C.POINTER(Ctg)), ("n_hit", C.c_int32), ("m_hit", C.c_int32), ("hit", C.c_void_p),
                ("n_exon", C.c_int32), ("m_exon", C.c_int32), ("exon", C.c_void_p), ("label", C.c_char_p)]


class Data(C.Structure):
    _fields_ = [("d_ctg", C.c_void_p), ("d_gene", C.c_void_p), ("d_prot", C.c_void_p), ("n_genome", C.c_int32), ("m_genome", C.c_int32), ("genome", C.POINTER(Genome)),
                ("n_gene", C.c_int32), ("m_gene", C.c_int32), ("gene", C.POINTER(Gene)), ("n_prot", C.c_int32), ("m_prot", C.c_int32), ("prot", C.POINTER(Prot))]


def state_of(lib, files, argv, **kw):
    """everything a read leaves behind, as one comparable object"""
    opt = capi.parse_args(lib, argv)
    d = lib.pg_data_init()
    try:
        assert capi.read_files(lib, opt, d, files, **kw) == 0
        D = C.cast(d, C.POINTER(Data)).contents
        genes = [(D.gene[i].name, D.gene[i].bits) for i in range(D.n_gene)]
        prots = [(D.prot[i].name, D.prot[i].len, D.prot[i].gid) for i in range(D.n_prot)]
        genomes = []
        for j in range(D.n_genome):
            g = D.genome[j]
            hits = hashlib.md5(C.string_at(g.hit, 88 * g.n_hit)).hexdigest() if g.n_hit else ""
            exons = hashlib.md5(C.string_at(g.exon, 8 * g.n_exon)).hexdigest() if g.n_exon else ""
            genomes.append((g.label, g.n_hit, g.n_exon, hits, exons, [(g.ctg[c].name, g.ctg[c].len) for c in range(g.n_ctg)]))
        return genes, prots, genomes
    finally:
        lib.pg_data_destroy(d)


def moving_set(tmp_path, n_files=36, n_prot=2500):
    """a bacterial-shaped set (big enough for the arena: > 8 MB of text) in which later files change what earlier ones said:
    protein lengths (prot.len is the LAST file's, gene.len the maximum: read.c:168-177), names that only late files hold,
    a file without any line, lines of one protein apart from each other"""
    rng = np.random.default_rng(5)
    out = []
    for j, (name, text) in enumerate(synth.bact(n_files, n_prot, seed=21)):
        lines = text.splitlines()
        if j >= 3:
            for k in rng.choice(len(lines), size=25, replace=False):  # another length for the protein, and with it another coverage test
                f = lines[k].split("\t")
                f[1] = str(int(f[1]) + int(rng.integers(1, 40)) * (1 if j % 2 else -1) * (1 if int(f[1]) > 60 else 0) + (j % 3))
                lines[k] = "\t".join(f)
        if j >= 5:
            for k in rng.choice(len(lines), size=8, replace=False):  # names nobody has seen: a new protein of a known gene, a new gene
                f = lines[k].split("\t")
                f[0] = (f[0] + ":iso%d" % j) if k % 2 else ("NEW%d_%d" % (j, k))
                lines[k] = "\t".join(f)
        if j == 7:
            lines = []
        if j % 4 == 1:
            lines = [lines[i] for i in rng.permutation(len(lines))]
        p = tmp_path / ("m%03d.paf" % j)
        p.write_text("".join(l + "\n" for l in lines))
        out.append(str(p))
    return out


@pytest.mark.parametrize("argv", [[], ["-P", "P000003,P000007"], ["-d", ":"]])
def test_batch_state_equals_sequential_state(built, tmp_path, argv):
    lib = oracle_host.load()
    C.c_int.in_dll(lib, "pg_verbose").value = 0
    files = moving_set(tmp_path)
    assert sum(os.path.getsize(f) for f in files) > (8 << 20)  # the arena is in play
    want = state_of(lib, files, argv, batch=False)
    assert len(want[1]) > 2500 and any(b"NEW" in n for n, _ in want[0])
    for nt in (1, 3, 16):
        got = state_of(lib, files, argv, batch=True, n_threads=nt)
        assert got[0] == want[0], "genes differ with %d threads" % nt
        assert got[1] == want[1], "proteins differ with %d threads" % nt
        assert got[2] == want[2], "genomes differ with %d threads" % nt


def test_score_adj_fast_path_is_the_long_double_route(built, tmp_path):
    """read.c:216 computes score_adj in long double through expl(); the reader takes exp() in double and falls back to expl() only when
    the value lies within a few ulp of a whole number.  100 000 random (score, identity, coverage) triples against Python's own
    extended-precision-free check: the integer part of score * e^x + .499 computed with 50 digits."""
    from decimal import Decimal, getcontext
    getcontext().prec = 60
    rng = np.random.default_rng(9)
    lib = oracle_host.load()
    C.c_int.in_dll(lib, "pg_verbose").value = 0
    n = 20000
    plen = rng.integers(50, 3000, n)
    qs = rng.integers(0, 20, n)
    qe = plen - rng.integers(0, 20, n)
    blen = rng.integers(100, 9000, n)
    mlen = (blen * rng.uniform(0.6, 1.0, n)).astype(np.int64)
    sc = rng.integers(1, 20000, n)
    lines = []
    for i in range(n):
        span = 3 * int(plen[i])
        lines.append("G%d:p\t%d\t%d\t%d\t+\tc1\t100000000\t%d\t%d\t%d\t%d\t0\tms:i:%d\tcg:Z:%dM\n" % (i, plen[i], qs[i], qe[i], 1000 * i, 1000 * i + span, mlen[i], blen[i], sc[i], plen[i]))
    p = tmp_path / "s.paf"
    p.write_text("".join(lines))
    opt = capi.parse_args(lib, ["-e", "0", "-l", "0"])
    d = lib.pg_data_init()
    try:
        assert capi.read_files(lib, opt, d, [str(p)]) == 0
        D = C.cast(d, C.POINTER(Data)).contents
        g = D.genome[0]
        assert g.n_hit == n
        raw = np.frombuffer(C.string_at(g.hit, 88 * n), dtype=np.int32).reshape(n, 22)
        got = raw[:, 9]  # score_adj
        coef = Decimal(opt.score_adj_coef)
        for i in range(n):
            div = 1.0 - float(mlen[i]) / float(blen[i])
            unc = 1.0 - float(qe[i] - qs[i]) / float(plen[i])
            x = Decimal(-opt.score_adj_coef * (div + unc))  # the argument is a double in the reference too
            want = int(Decimal(int(sc[i])) * x.exp() + Decimal(0.499))
            assert got[i] == want, (i, got[i], want)
    finally:
        lib.pg_data_destroy(d)


def _adversarial_lines(rng, n_ctg=2):
    """PAF lines the format allows and no generator of synth.py emits: numbers the way strtol takes them (blanks, signs, trailing
    junk, 19+ digits), tags in any order / twice / cut short, every CIGAR operation of miniprot (M I D N U V F G X =), CIGARs that
    with zero-length operations, '*' strands, blank and short lines, CR LF, names that hold the intron letters N U V"""
    lines = []
    for i in range(260):
        g = "GENE%d" % (i % 37)
        name = "%s:UV%dN" % (g, i % 3)                       # gene<delim>protein; intron letters in names
        plen = 40 + ((i % 37) * 7 + (i % 3) * 11) % 300      # one length per protein name (read.c:175 asserts it)
        qs, qe = (i % 5), plen - (i % 3)
        strand = "+-"[i % 2] if i % 53 else "*"
        ctg = "ctgN%d" % (i % n_ctg)
        cs = 1000 + 977 * i
        kind = i % 11
        if kind == 0:   cg, span = "%dM" % plen, 3 * plen
        elif kind == 1: cg, span = "%dM30N%dM" % (plen // 2, plen - plen // 2), 3 * plen + 30
        elif kind == 2: cg, span = "%dM31U%dM" % (plen // 2, plen - plen // 2), 3 * plen + 31
        elif kind == 3: cg, span = "%dM32V%dM" % (plen // 2, plen - plen // 2), 3 * plen + 32
        elif kind == 4: cg, span = "%dM1F%dM" % (plen // 2, plen - plen // 2), 3 * plen + 1
        elif kind == 5: cg, span = "%dM2G%dM" % (plen // 2, plen - plen // 2), 3 * plen + 2
        elif kind == 6: cg, span = "%dM3I2D%dM" % (plen // 2, plen - plen // 2), 3 * plen + 6
        elif kind == 7: cg, span = "%dX%d=" % (plen // 2, plen - plen // 2), 3 * plen
        elif kind == 8: cg, span = "%dM40N%dM50N%dM" % (plen // 3, plen // 3, plen - 2 * (plen // 3)), 3 * plen + 90
        elif kind == 9: cg, span = "0M%dM00N1I" % plen, 3 * plen          # zero-length operations (a CIGAR that does not span the alignment is an assert in the reference, read.c:75: not fed)
        else:           cg, span = "%dM" % plen, 3 * plen
        ce = cs + span
        mlen, blen = span - (i % 7), span
        ms = [str(50 + (i * 13) % 900), "+%d" % (60 + i), " %d" % (70 + i), "-%d" % (5 + i % 9), "0", "%d junk" % (80 + i), "99999999999999999999"][i % 7]
        num = lambda v, k: [str(v), "+%d" % v, " %d" % v, "%dx" % v, "%d " % v, "0%d" % v][k % 6] if (i % 17 == 3) else str(v)
        cols = [name, num(plen, i), num(qs, i + 1), num(qe, i + 2), strand, ctg, ("123456789012345678901" if i % 17 == 5 else "0000000000000000005000000" if i % 17 == 6 else num(5_000_000, i + 3)), num(cs, i + 4), num(ce, i + 5), num(mlen, i + 1), num(blen, i + 2), "0"]
        tags = ["ms:i:" + ms, "cg:Z:" + cg]
        if i % 13 == 1: tags.append("fs:i:%d" % (i % 3))
        if i % 19 == 2: tags.append("st:i:1")
        if i % 5 == 0: tags.reverse()
        if i % 29 == 4: tags.append("ms:i:7")                  # a second score tag: the last one counts (read.c:212)
        if i % 31 == 5: tags = ["ms:i", "cg:"] + tags           # cut short
        if i % 37 == 6: tags = tags[:1]                          # no CIGAR: dropped
        if i % 23 == 7: tags.insert(0, "AS:i:12")
        line = "\t".join(cols + tags)
        if i % 41 == 8: line = "\t".join(cols[:7])              # a short line
        if i % 43 == 9: line = ""
        if i % 47 == 10: line += "\t"
        lines.append(line + ("\r\n" if i % 9 == 0 else "\n"))
    order = rng.permutation(len(lines)) if rng is not None else range(len(lines))
    return "".join(lines[k] for k in order)


@pytest.mark.parametrize("variant", [[], ["-S"], ["-p0", "-a1"], ["--bed=raw"], ["-e", "0.3", "-l", "0.2"]])
def test_adversarial_paf_lines_against_the_reference_binary(built, tmp_path, variant):
    """the reader's short cuts (digits parsed and delimited in one walk, tags told apart by their bytes, CIGAR numbers without strtol,
    exp() with a guard band, names resolved against a snapshot) on lines written to miss them, against what the untouched reference
    prints for the same files -- through the batch reader (plain and gzipped) and the per-file reader"""
    import gzip
    import subprocess
    from conftest import ROOT
    ref = os.path.join(ROOT, "oracle", "_ref", "pangene_ref")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/pangene_ref not built")
    lib = oracle_host.load()
    C.c_int.in_dll(lib, "pg_verbose").value = 0
    files = []
    for j in range(4):
        text = _adversarial_lines(np.random.default_rng(100 + j) if j % 2 else None)
        p = tmp_path / ("a%d.paf" % j)
        p.write_text(text, newline="")
        files.append(str(p))
        if j == 3:
            with gzip.open(str(p) + ".gz", "wt", newline="") as f:
                f.write(text)
            files[-1] = str(p) + ".gz"
    want = subprocess.run([ref] + variant + files, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    assert want.returncode == 0 and len(want.stdout) > 500
    lib.pg_set_exact_mode(2)
    assert capi.run(lib, files, variant, batch=True) == want.stdout
    assert capi.run(lib, files, variant, batch=False) == want.stdout
