"""Seeded synthetic miniprot-PAF generators for the pangene graph-construction path.

The reference ships no benchmark inputs besides test/C4, so the shapes named in
BASELINE.json are produced here:

  bact(G, P)      : bacterial pangenome -- one contig per genome, single-exon hits, paralog families whose
                    members have equal length (=> many hits with identical (cs, ce): tie groups),
                    core/accessory genes, inversions.  bact(100, 5000) ~ 1.0 M hits = BASELINE configs[1].
  human(G, Q, iso): human-shaped -- multi-exon genes with N/U/V introns, isoforms that share exons,
                    tandem CNV, tandem paralog pairs, processed single-exon copies, nested genes,
                    frameshifts, `sample#hap#ctg` contig names (stand-in for HPRC; real HPRC PAFs are
                    not available offline).
  fuzz(seed)      : tiny adversarial sets on a coarse coordinate grid (exercises every tie channel of
                    SURVEY.md section 9.1).

Every generator yields (file_name, text) per genome; RNG stream = f(seed, genome index), so any
sub-range of genomes can be produced independently (needed for sharding across ranks).
Only the PAF columns and tags pangene reads are emitted (read.c:128-236 of the reference lists them:
cols 1-11, ms:i, fs:i, st:i, cg:Z).
"""
from __future__ import annotations

import gzip
import os
from typing import Iterator, List, Tuple

import numpy as np

__all__ = ["bact", "human", "fuzz", "odd_exons", "write_files"]


def _rng(seed: int, *stream: int) -> np.random.Generator:
    return np.random.default_rng([int(seed)] + [int(s) + 1000003 for s in stream])


# ----------------------------------------------------------------------------------------------
# bacterial shape
# ----------------------------------------------------------------------------------------------
class _BactModel:
    """Genome-independent part of bact(G, P): protein lengths, families, base order, gaps."""

    def __init__(self, P: int, seed: int):
        r = _rng(seed, -1)
        self.P = P
        # families: sizes drawn so that mean hits/protein/genome ~ 2.0 (each member of a size-s family
        # hits all s loci when they are present)
        fam = np.empty(P, dtype=np.int64)
        sizes = []
        i = 0
        while i < P:
            u = r.random()
            s = 1 if u < 0.66 else int(r.integers(2, 7))
            s = min(s, P - i)
            fam[i:i + s] = len(sizes)
            sizes.append(s)
            i += s
        self.fam = fam
        self.n_fam = len(sizes)
        fam_len = r.integers(80, 601, size=self.n_fam)
        self.L = fam_len[fam]                                  # aa; equal inside a family
        # pairwise similarity of paralogs inside one family (one scalar per family)
        self.fam_sim = r.uniform(0.972, 0.999, size=self.n_fam)
        self.order = r.permutation(P)                          # shared base gene order
        self.strand = r.integers(0, 2, size=P)                 # per-gene strand in the base order
        core = r.random(P) < 0.8
        self.freq = np.where(core, 1.0, r.uniform(0.05, 0.95, size=P))
        gap = r.integers(20, 301, size=P)
        u = r.random(P)
        gap = np.where(u < 0.10, -r.integers(1, 12, size=P), gap)        # short overlaps, like real operons
        big = u > 0.985                                                   # substantial overlaps (30-70 %)
        gap = np.where(big, -(self.L * 3 * r.uniform(0.3, 0.7, size=P)).astype(np.int64), gap)
        self.gap = gap
        self.names = np.array(["P%06d" % i for i in range(P)])
        # members of each family, for fast lookup
        o = np.argsort(fam, kind="stable")
        self.fam_members_sorted = o
        self.fam_start = np.searchsorted(fam[o], np.arange(self.n_fam + 1))


def _bact_genome(m: _BactModel, seed: int, j: int) -> str:
    r = _rng(seed, j)
    P = m.P
    present = r.random(P) < m.freq
    order = m.order[present[m.order]]
    strand = m.strand[order].copy()
    n = len(order)
    for _ in range(int(r.integers(0, 4))):                     # 0-3 inversions of 2-200 genes
        if n < 4:
            break
        ln = int(min(n - 1, r.integers(2, 201)))
        a = int(r.integers(0, n - ln))
        order[a:a + ln] = order[a:a + ln][::-1].copy()
        strand[a:a + ln] = 1 - strand[a:a + ln][::-1]
    L3 = m.L[order] * 3
    gap = m.gap[order] + r.integers(-2, 3, size=n)
    gap = np.maximum(gap, -(L3 * 7 // 10))
    step = L3 + gap
    step = np.maximum(step, 1)                                 # keep starts strictly increasing
    cs = int(r.integers(1000, 5000)) + np.concatenate(([0], np.cumsum(step[:-1])))
    ce = cs + L3
    ctg_len = int(ce.max() + r.integers(1000, 5000))
    ctg = "g%d#0#chr1" % j
    div = r.uniform(0.0, 0.15)
    iden_locus = np.clip(1.0 - div * r.uniform(0.5, 1.5, size=n), 0.55, 1.0)
    locus_of_gene = np.full(P, -1, dtype=np.int64)
    locus_of_gene[order] = np.arange(n)

    # hits: every protein p (present or not) x every present locus of its family
    fam_loci_gene = order                                       # gene sitting at each locus
    fam_of_locus = m.fam[fam_loci_gene]
    # loci grouped by family
    lo = np.argsort(fam_of_locus, kind="stable")
    lstart = np.searchsorted(fam_of_locus[lo], np.arange(m.n_fam + 1))
    lines: List[str] = []
    noise = r.uniform(0.985, 1.0, size=4 * n + 16)
    ni = 0
    for f in range(m.n_fam):
        loci = lo[lstart[f]:lstart[f + 1]]
        if len(loci) == 0:
            continue
        members = m.fam_members_sorted[m.fam_start[f]:m.fam_start[f + 1]]
        for p in members:
            L = int(m.L[p])
            hits = []
            for q in loci:
                own = fam_loci_gene[q] == p
                idn = iden_locus[q] if own else iden_locus[q] * m.fam_sim[f] * noise[ni % len(noise)]
                ni += 1
                hits.append((idn, int(q)))
            best = max(h[0] for h in hits)
            hits = [h for h in hits if h[0] >= 0.97 * best]
            hits.sort(key=lambda h: (-h[0], h[1]))
            name = m.names[p]
            for idn, q in hits:
                blen = 3 * L
                mlen = int(blen * idn)
                ms = int(1.6 * mlen)
                lines.append("%s\t%d\t0\t%d\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t0\tms:i:%d\tcg:Z:%dM\n" % (
                    name, L, L, "+-"[strand[q]], ctg, ctg_len, cs[q], ce[q], mlen, blen, ms, L))
    return "".join(lines)


def bact(G: int, P: int, seed: int = 1, first: int = 0, last: int | None = None) -> Iterator[Tuple[str, str]]:
    """Yield (file name, PAF text) for genomes first..last-1 of the bact(G, P) set."""
    m = _BactModel(P, seed)
    last = G if last is None else last
    for j in range(first, last):
        yield ("g%05d.paf" % j, _bact_genome(m, seed, j))


# ----------------------------------------------------------------------------------------------
# human shape
# ----------------------------------------------------------------------------------------------
class _HumanModel:
    def __init__(self, Q: int, iso: float, seed: int, n_chr: int = 24):
        r = _rng(seed, -2)
        self.Q, self.n_chr = Q, n_chr
        self.genes = []
        chr_of = r.integers(0, n_chr, size=Q)
        for gid in range(Q):
            ne = int(min(40, 1 + r.exponential(8.0)))
            if r.random() < 0.003:
                ne = int(r.integers(100, 140))
            ex = r.integers(15, 91, size=ne)                                 # aa per exon
            intr = r.integers(90, 4001, size=max(ne - 1, 0))
            ityp = r.integers(0, 3, size=max(ne - 1, 0))                     # 0:N 1:U 2:V
            n_iso = int(min(8, 1 + r.poisson(max(iso - 1.0, 0.0))))
            isos = [list(range(ne))]
            for _ in range(n_iso - 1):
                keep = list(range(ne))
                if ne >= 3:
                    for __ in range(int(r.integers(1, 3))):
                        if len(keep) > 2:
                            k = int(r.integers(1, len(keep) - 1))
                            if r.random() < 0.3:
                                k = 0                                         # alternative first exon
                            keep.pop(k)
                isos.append(keep)
            self.genes.append(dict(
                chr=int(chr_of[gid]), strand=int(r.integers(0, 2)), ex=ex, intr=intr, ityp=ityp, isos=isos,
                spacer=int(r.integers(2000, 50001)),
                nested=bool(r.random() < 0.02) and gid > 0,
                paralog=bool(r.random() < 0.04),                             # tandem near-identical pair with next gene
                processed=bool(r.random() < 0.04 and ne > 1),                # single-exon processed copy elsewhere
                pp_chr=int(r.integers(0, n_chr)), pp_pos=int(r.integers(10_000, 3_000_000)),
                sim=float(r.uniform(0.93, 0.995)),
                name="G%05d" % gid))


def _cigar_and_span(ex, intr, ityp, keep, strand, fs_at=-1):
    """CIGAR in transcript order + genomic (start offset, span) of isoform `keep` of a gene."""
    # genomic layout of the full gene (forward coordinates): exon i starts at gs[i]
    ne = len(ex)
    gs = np.zeros(ne, dtype=np.int64)
    x = 0
    for i in range(ne):
        gs[i] = x
        x += 3 * int(ex[i])
        if i < ne - 1:
            x += int(intr[i])
    ge = gs + 3 * ex.astype(np.int64)
    ops = []
    first, lastx = keep[0], keep[-1]
    for a, i in enumerate(keep):
        ops.append("%dM" % int(ex[i]))
        if a == fs_at:
            ops.append("1F")
        if a + 1 < len(keep):
            nxt = keep[a + 1]
            gapnt = int(gs[nxt] - ge[i])
            t = int(ityp[i]) if i < len(ityp) else 0
            if t == 0 or gapnt < 8:
                ops.append("%dN" % gapnt)
            else:                                               # split codon: borrow 3 nt from the intron
                ops.append("%d%s" % (gapnt, "UV"[t - 1]))
    span = int(ge[lastx] - gs[first]) + (1 if fs_at >= 0 else 0)
    if strand:
        ops = ops[::-1]
    return "".join(ops), int(gs[first]), span


def _human_genome(m: _HumanModel, seed: int, j: int, frag: bool) -> str:
    r = _rng(seed, j)
    hap = j % 2 + 1
    sample = "S%04d" % (j // 2)
    div = r.uniform(0.0, 0.03)
    pos = [int(r.integers(5000, 50000)) for _ in range(m.n_chr)]
    # per-chromosome fragment boundaries
    lines: List[str] = []
    chr_len = [0] * m.n_chr
    recs = []                                                   # (gid, chr, gene start, strand, iden, copy#)
    prev_locus = None
    for gid, g in enumerate(m.genes):
        if r.random() < 0.02:
            continue                                            # gene absent in this genome
        c = g["chr"]
        full_span = int(3 * g["ex"].sum() + g["intr"].sum())
        if g["nested"] and prev_locus is not None and prev_locus[0] == c and prev_locus[2] > full_span + 200:
            start = prev_locus[1] + int(r.integers(50, prev_locus[2] - full_span - 50))   # inside previous gene
        else:
            start = pos[c] + g["spacer"] + int(r.integers(-500, 501))
            pos[c] = start + full_span
        idn = float(np.clip(1.0 - div * r.uniform(0.2, 2.0), 0.6, 1.0))
        recs.append((gid, c, start, g["strand"], idn, 0))
        prev_locus = (c, start, full_span)
        if r.random() < 0.03:                                   # tandem CNV: second copy right after
            s2 = pos[c] + int(r.integers(1000, 5000))
            pos[c] = s2 + full_span
            recs.append((gid, c, s2, g["strand"], idn * float(r.uniform(0.985, 1.0)), 1))
    for c in range(m.n_chr):
        chr_len[c] = pos[c] + 100_000
    nfrag = int(r.integers(8, 20)) if frag else 1

    def ctg_of(c, x, span):
        if nfrag == 1:
            return "%s#%d#chr%d" % (sample, hap, c + 1), x, chr_len[c]
        fl = chr_len[c] // nfrag + 1
        k = x // fl
        if (x + span) // fl != k:                               # hit would straddle a break: keep it in fragment k
            return "%s#%d#chr%d_%d" % (sample, hap, c + 1, k), x - k * fl, fl + span
        return "%s#%d#chr%d_%d" % (sample, hap, c + 1, k), x - k * fl, fl + span

    by_prot = {}                                                # (gid, iso) -> list of hit tuples
    order_keys = []

    def add(gid, k, tup):
        key = (gid, k)
        if key not in by_prot:
            by_prot[key] = []
            order_keys.append(key)
        by_prot[key].append(tup)

    for (gid, c, start, strand, idn, cp) in recs:
        g = m.genes[gid]
        targets = [(gid, 1.0)]
        if g["paralog"] and gid + 1 < m.Q and len(m.genes[gid + 1]["ex"]) == len(g["ex"]):
            targets.append((gid + 1, g["sim"]))                 # proteins of the next gene also hit here
        if gid > 0 and m.genes[gid - 1]["paralog"] and len(m.genes[gid - 1]["ex"]) == len(g["ex"]):
            targets.append((gid - 1, m.genes[gid - 1]["sim"]))
        for (tg, sim) in targets:
            tgene = m.genes[tg]
            for k, keep in enumerate(tgene["isos"]):
                if tg != gid:                                   # paralog protein on this locus: use this locus' exon frame
                    keep = [e for e in keep if e < len(g["ex"])]
                    if not keep:
                        continue
                fs = int(r.integers(0, len(keep))) if r.random() < 0.01 else -1
                cg, off, span = _cigar_and_span(g["ex"], g["intr"], g["ityp"], keep, strand, fs)
                plen = int(sum(int(tgene["ex"][e]) for e in tgene["isos"][k]))
                alen = int(sum(int(g["ex"][e]) for e in keep))
                qs, qe = 0, min(plen, alen)
                if r.random() < 0.03:
                    qe = int(plen * r.uniform(0.4, 0.9))
                    qe = max(1, min(qe, alen))
                ii = idn * sim * float(r.uniform(0.995, 1.0))
                if r.random() < 0.01:
                    ii *= float(r.uniform(0.4, 0.6))            # junk, filtered by -e
                blen = 3 * alen + (1 if fs >= 0 else 0)
                mlen = int(3 * alen * ii)
                ms = int(1.7 * mlen) - 11 * (len(keep) - 1)
                name, x, cl = ctg_of(c, start + off, span)
                add(tg, k, (ms, "%s:T%d\t%d\t%d\t%d\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t0\tms:i:%d\tfs:i:%d\tst:i:0\tcg:Z:%s\n" % (
                    tgene["name"], k, plen, qs, qe, "+-"[strand], name, cl, x, x + span, mlen, blen, max(ms, 1),
                    1 if fs >= 0 else 0, cg)))
    # processed single-exon copies
    for gid, g in enumerate(m.genes):
        if not g["processed"] or r.random() < 0.3:
            continue
        c = g["pp_chr"]
        for k, keep in enumerate(g["isos"]):
            plen = int(sum(int(g["ex"][e]) for e in keep))
            x = g["pp_pos"] + int(r.integers(-3, 4)) * 3
            ii = 0.90 * float(r.uniform(0.97, 1.0))
            mlen = int(3 * plen * ii)
            ms = int(1.7 * mlen)
            name, xx, cl = ctg_of(c, x, 3 * plen)
            add(gid, k, (ms, "%s:T%d\t%d\t0\t%d\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t0\tms:i:%d\tfs:i:0\tst:i:0\tcg:Z:%dM\n" % (
                g["name"], k, plen, plen, "+-"[int(r.integers(0, 2))], name, cl, xx, xx + 3 * plen, mlen, 3 * plen, ms, plen)))
    for key in order_keys:
        hs = by_prot[key]
        hs.sort(key=lambda t: -t[0])
        lines.extend(t[1] for t in hs)
    return "".join(lines)


def human(G: int, Q: int, iso: float = 1.0, seed: int = 1, first: int = 0, last: int | None = None,
          frag: bool = False, n_chr: int = 24) -> Iterator[Tuple[str, str]]:
    m = _HumanModel(Q, iso, seed, n_chr)
    last = G if last is None else last
    for j in range(first, last):
        yield ("h%05d.paf" % j, _human_genome(m, seed, j, frag))


# ----------------------------------------------------------------------------------------------
# adversarial fuzz
# ----------------------------------------------------------------------------------------------
def fuzz(seed: int, harsh: bool = True) -> Iterator[Tuple[str, str]]:
    """3-8 genomes, 12-40 genes x 1-3 isoforms x 0-4 hits on a 300-bp grid (SURVEY.md section 10 iii)."""
    r = _rng(seed, -3)
    G = int(r.integers(3, 9))
    n_gene = int(r.integers(12, 41))
    n_ctg = int(r.integers(1, 3))
    grid = int(r.integers(25, 201))
    lens = [30, 60, 100] if harsh else list(range(30, 200, 7))
    idens = [1.0, 0.9, 0.8] if harsh else [x / 100 for x in range(70, 101)]
    genes = []
    for g in range(n_gene):
        n_iso = int(r.integers(1, 4))
        ne = int(r.integers(1, 5))
        genes.append((n_iso, ne))
    for j in range(G):
        rj = _rng(seed, j)
        lines = []
        for g, (n_iso, ne) in enumerate(genes):
            base = int(rj.integers(0, grid)) * 300 + 1000
            c = int(rj.integers(0, n_ctg))
            strand = int(rj.integers(0, 2))
            for k in range(n_iso):
                nh = int(rj.integers(0, 5))
                hits = []
                for h in range(nh):
                    e = int(rj.integers(1, ne + 1)) if rj.random() < 0.7 else 1
                    ex = [int(rj.choice(lens)) for _ in range(e)]
                    intr = [int(rj.choice([90, 300, 600])) for _ in range(e - 1)]
                    if h == 0:
                        x = base + (0 if rj.random() < 0.6 else int(rj.integers(0, 4)) * 300)
                        cc, ss = c, strand
                    else:
                        x = int(rj.integers(0, grid)) * 300 + 1000
                        cc, ss = int(rj.integers(0, n_ctg)), int(rj.integers(0, 2))
                    ops = []
                    span = 0
                    for a in range(e):
                        ops.append("%dM" % ex[a]); span += 3 * ex[a]
                        if a + 1 < e:
                            t = int(rj.integers(0, 3))
                            ops.append("%d%s" % (intr[a], "NUV"[t])); span += intr[a]
                    if ss:
                        ops = ops[::-1]
                    plen = sum(ex)
                    idn = float(rj.choice(idens))
                    mlen = int(3 * plen * idn)
                    ms = int(float(rj.choice([1.5, 1.6, 1.6])) * mlen)
                    hits.append((ms, "G%03d:T%d\t%d\t0\t%d\t%s\tS%d#1#c%d\t%d\t%d\t%d\t%d\t%d\t0\tms:i:%d\tcg:Z:%s\n" % (
                        g, k, plen, plen, "+-"[ss], j, cc, 400000, x, x + span, mlen, 3 * plen, ms, "".join(ops))))
                hits.sort(key=lambda t: -t[0])
                lines.extend(t[1] for t in hits)
        yield ("f%02d.paf" % j, "".join(lines))


def odd_exons(seed: int) -> Iterator[Tuple[str, str]]:
    """Exon lists that are NOT sorted and disjoint: U / V introns shorter than 3 bp make the next exon start before the last one ends
    (read.c:59-62: st = x + 1, en = x + l - 2), zero-length introns make exons touch, `0M` between two introns makes an exon of no
    length.  miniprot does not write such lines; the reference takes them (its pg_hit_overlap only asserts l_inter <= l_union), and the
    device path must then merge step by step like it (cds_inter_ref) instead of taking its shortcuts.  Piles of overlapping
    multi-exon isoforms of few genes, so that most pairs are evaluated."""
    r = _rng(seed, -7)
    G = int(r.integers(3, 7))
    n_gene = int(r.integers(6, 15))
    genes = [(int(r.integers(2, 5)), int(r.integers(2, 7))) for _ in range(n_gene)]
    for j in range(G):
        rj = _rng(seed, j)
        lines = []
        for g, (n_iso, ne) in enumerate(genes):
            base = 1000 + 900 * int(rj.integers(0, 12))
            strand = int(rj.integers(0, 2))
            for k in range(n_iso):
                for h in range(int(rj.integers(1, 3))):
                    e = int(rj.integers(1, ne + 1))
                    ops, span = [], 0
                    for a in range(e):
                        L = int(rj.choice([0, 10, 20, 40]))
                        ops.append("%dM" % L); span += 3 * L
                        if a + 1 < e:
                            il = int(rj.choice([0, 1, 2, 3, 60, 300]))
                            ops.append("%d%s" % (il, "NUV"[int(rj.integers(0, 3))])); span += il
                    if span == 0:
                        ops, span = ["10M"], 30
                    if strand:
                        ops = ops[::-1]
                    plen = max(1, sum(int(o[:-1]) for o in ops if o[-1] == "M"))
                    x = base + 30 * int(rj.integers(0, 6))
                    mlen = int(3 * plen * float(rj.choice([1.0, 0.9])))
                    ms = int(1.6 * mlen)
                    lines.append((g, -ms, "G%03d:T%d\t%d\t0\t%d\t%s\tS%d#1#c0\t%d\t%d\t%d\t%d\t%d\t0\tms:i:%d\tcg:Z:%s\n" % (
                        g, k, plen, plen, "+-"[strand], j, 400000, x, x + span, mlen, 3 * plen, ms, "".join(ops))))
        lines.sort(key=lambda t: (t[0], t[1]))
        yield ("o%02d.paf" % j, "".join(t[2] for t in lines))


def dense(seed: int = 1, G: int = 4, n_small: int = 700, pile: int = 48) -> Iterator[Tuple[str, str]]:
    """Stress shape for the interval sweep: per genome one contig with (a) a lattice of small single-exon hits that each
    overlap a few neighbours, (b) one giant two-exon hit whose intron spans the whole lattice (every hit then has a
    partner hundreds of array slots away), (c) pile-ups of `pile` different genes on one locus (hundreds of overlapping
    pairs inside 64 consecutive hits) in single- and multi-exon flavour, (d) exact duplicates of one protein."""
    for j in range(G):
        rj = _rng(seed, 7000 + j)
        lines = []

        def add(name, plen, x, ops, span, ms, strand="+", idn=1.0):
            mlen = int(3 * plen * idn)
            lines.append("%s\t%d\t0\t%d\t%s\tD%d#0#c0\t%d\t%d\t%d\t%d\t%d\t0\tms:i:%d\tcg:Z:%s\n" % (
                name, plen, plen, strand, j, 3000000, x, x + span, mlen, 3 * plen, ms, ops))

        for g in range(n_small):  # (a)
            if rj.random() < 0.1:
                continue
            plen = int(rj.choice([60, 100, 140]))
            x = 5000 + g * 150 + int(rj.integers(0, 3)) * 30
            add("s%04d" % g, plen, x, "%dM" % plen, 3 * plen, int(4.8 * plen * float(rj.choice([1.0, 0.95, 0.9]))), "+-"[int(rj.integers(0, 2))])
        add("giant", 200, 4000, "100M%dN100M" % (n_small * 150 + 3000), 600 + n_small * 150 + 3000, 900 + j)  # (b)
        for k in range(pile):  # (c) single exon, then three exons
            plen = 100 + (k % 5)
            add("p%03d" % k, plen, 400000 + (k % 4) * 30, "%dM" % plen, 3 * plen, 400 + int(rj.integers(0, 60)), "+-"[k & 1])
            add("m%03d" % k, 90, 600000 + (k % 3) * 30, "30M200N30M300U30M", 270 + 500, 350 + int(rj.integers(0, 60)), "+-"[k & 1])
        for k in range(3):  # (d) the same protein, same score, same place
            add("dup", 120, 800000, "120M", 360, 500)
        yield ("d%02d.paf" % j, "".join(lines))


def many_doms(seed: int = 1, G: int = 24, n_x: int = 6) -> Iterator[Tuple[str, str]]:
    """Genes x0..x{n_x-1} stand alone in the even genomes and lose, in odd genome j, to a stronger gene y{i}_{(j // 2) % 11}
    on the same locus: each x is dominant often enough to be selected first and then marks cells of 11 distinct dominators
    (more than the eight slots per gene the device fold of pg_gen_vtx keeps); y genes that only ever dominate an x end up
    with all their genomes marked (vertex.c:70: y == x, not selected) unless they also occur on their own (every third).
    A background of ordinary genes provides vertices and arcs."""
    for j in range(G):
        rj = _rng(seed, 9000 + j)
        lines = []

        def add(name, plen, x, ms, strand="+"):
            lines.append("%s\t%d\t0\t%d\t%s\tM%d#0#c0\t%d\t%d\t%d\t%d\t%d\t0\tms:i:%d\tcg:Z:%dM\n" % (
                name, plen, plen, strand, j, 1000000, x, x + 3 * plen, 3 * plen, 3 * plen, ms, plen))

        for b in range(60):
            if rj.random() < 0.9:
                add("bg%02d" % b, 100, 2000 + b * 1000, 480, "+-"[b & 1])
        for i in range(n_x):
            x = 100000 + i * 5000
            add("x%d" % i, 100, x, 300)
            if j & 1:
                add("y%d_%d" % (i, (j // 2) % 11), 110, x - 15, 520)
            for k in range(0, 11, 3):  # these dominators also stand alone in every genome
                if not ((j & 1) and k == (j // 2) % 11):
                    add("y%d_%d" % (i, k), 110, 200000 + (i * 11 + k) * 1000, 520)
        yield ("m%02d.paf" % j, "".join(lines))


def mutate(gen: Iterator[Tuple[str, str]], seed: int, p_score: float = 0.06, p_dup: float = 0.03, p_flip: float = 0.03, p_tag: float = 0.03,
           p_drop: float = 0.02, p_shuffle: float = 0.3) -> Iterator[Tuple[str, str]]:
    """Inputs the generators above never produce but the format allows: non-positive `ms:i:` scores (graph.c:133 running maxima
    start at 0; a negative score_adj takes the 64-bit score-key route), exact duplicates of an alignment, flipped strands, `fs:i:` /
    `st:i:` tags (read.c:217-220), missing lines, and -- in about a third of the files -- lines that are not grouped by protein."""
    for j, (name, text) in enumerate(gen):
        r = _rng(seed, 5000 + j)
        out = []
        for line in text.splitlines():
            if r.random() < p_drop:
                continue
            f = line.split("\t")
            if r.random() < p_score:
                f = [("ms:i:%d" % int(r.choice([0, 1, -5, -100]))) if x.startswith("ms:i:") else x for x in f]
            if r.random() < p_flip:
                f[4] = "-" if f[4] == "+" else "+"
            if r.random() < p_tag:
                f.insert(12, "fs:i:%d" % int(r.integers(1, 3)) if r.random() < 0.5 else "st:i:1")
            out.append("\t".join(f))
            if r.random() < p_dup:
                out.append(out[-1])
        if r.random() < p_shuffle:
            out = [out[i] for i in r.permutation(len(out))]
        yield (name, "".join(x + "\n" for x in out))


def widen(gen: Iterator[Tuple[str, str]], seed: int, p_gap: float = 0.15, p_shift: float = 0.5) -> Iterator[Tuple[str, str]]:
    """Contig coordinates beyond 32 bits (pangene.h:71 keeps cs / cm / ce in int64_t; read.c:202-204 parses them with strtol): every
    contig is stretched at some of its HIT-FREE gaps by 1.3e9, 2.2e9 or 4.7e9 bp -- so that distances between neighbouring hits pass
    2^31 (graph.c:73 narrows them to int32_t) and 2^32 -- and about half of the contigs start beyond 2^31 as a whole.  Hits keep
    their lengths, their order and their overlaps."""
    jumps = [1_300_000_000, 2_200_000_000, 4_700_000_000, (1 << 32)]
    for j, (name, text) in enumerate(gen):
        r = _rng(seed, 7000 + j)
        lines = [l.split("\t") for l in text.splitlines()]
        per_ctg = {}
        for i, f in enumerate(lines):
            if len(f) > 8 and f[7].isdigit() and f[8].isdigit():
                per_ctg.setdefault(f[5], []).append(i)
        for ctg, idx in per_ctg.items():
            idx.sort(key=lambda i: (int(lines[i][7]), int(lines[i][8])))
            off = int(r.choice([2_500_000_000, 5_000_000_000])) if r.random() < p_shift else 0
            reach = -1
            new = {}
            for i in idx:
                cs, ce = int(lines[i][7]), int(lines[i][8])
                if reach >= 0 and cs > reach and r.random() < p_gap:
                    off += int(r.choice(jumps))
                reach = max(reach, ce)
                new[i] = (cs + off, ce + off)
            end = max(int(lines[idx[-1]][6]) + off, max(e for _, e in new.values()) + 1) if lines[idx[-1]][6].isdigit() else None
            for i in idx:
                lines[i][7], lines[i][8] = str(new[i][0]), str(new[i][1])
                if end is not None:
                    lines[i][6] = str(end)
        yield (name, "".join("\t".join(f) + "\n" for f in lines))


def write_files(gen: Iterator[Tuple[str, str]], out_dir: str, gz: bool = False) -> List[str]:
    os.makedirs(out_dir, exist_ok=True)
    paths = []
    for name, text in gen:
        p = os.path.join(out_dir, name + (".gz" if gz else ""))
        if gz:
            with gzip.open(p, "wt", compresslevel=6) as f:
                f.write(text)
        else:
            with open(p, "w") as f:
                f.write(text)
        paths.append(p)
    return paths


def _cpu_budget() -> int:
    """cores this process may really use: the affinity mask, capped by the control group's CPU quota (a box that shows 256 hardware
    threads may grant 16 cores of CPU time: more workers than that only get the whole group throttled)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except Exception:
        pass
    return max(1, n)


def write_files_parallel(kind: str, out_dir: str, n_proc: int = 0, **kw) -> List[str]:
    """write_files for the big sets, by a few fresh interpreters side by side (every genome is seeded on its own, so any split of
    [0, G) gives the same files; fresh processes -- not forks -- because the caller may hold a GPU context).
    kind: "bact" (G, P, seed) or "human" (G, Q, iso, seed, frag)."""
    import subprocess, sys
    G = int(kw["G"])
    n_proc = n_proc or max(1, min(G, min(_cpu_budget(), 64)))
    os.makedirs(out_dir, exist_ok=True)
    procs = []
    for k in range(n_proc):
        a, b = G * k // n_proc, G * (k + 1) // n_proc
        if a == b:
            continue
        arg = dict(kw, first=a, last=b)
        code = "import sys; sys.path.insert(0, %r); from pangene_amd import synth; synth.write_files(synth.%s(**%r), %r)" % (
            os.path.dirname(os.path.dirname(os.path.abspath(__file__))), kind, arg, out_dir)
        procs.append(subprocess.Popen([sys.executable, "-c", code]))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("a generator process failed")
    return sorted(os.path.join(out_dir, f) for f in os.listdir(out_dir))
