// k_sweep.hpp -- the interval-dominance sweep (K1): pg_shadow, pg_flt_ov_isoform, pg_hit_overlap.
// Included by pga_backend.hip (one translation unit); uses the context types, BLOCK / WAVE and dev_prims.hpp from there.
#pragma once

// ------------------------------------------------------------------------------------------------
// the interval-dominance sweep: pg_shadow (overlap.c:101-178) and pg_flt_ov_isoform (58-93)
// ------------------------------------------------------------------------------------------------
// Packed per-hit records for the sweep: a partner costs 16-byte loads instead of a dozen 4-byte ones.
//   A = {cs, seg, ce, pm}   B = {rk, gid, cds, pid}   C = {rank, n_exon, off_exon, score_ori}
// (A.xy read as one 64-bit word is seg << 32 | cs: the sort key of the X order, non-decreasing along the array)
// C is only needed for multi-exon hits, for two hits with the same score key and for score_dom.
__global__ __launch_bounds__(BLOCK) void k_pack_rec(const int32_t *seg, const int32_t *cs, const int32_t *ce, const int32_t *pm, const int32_t *rk,
                                                      const int32_t *gid, const int32_t *cds, const int32_t *rank, const int32_t *nex, const int32_t *offx,
                                                      const int32_t *pid, const int32_t *sori, int n, int4 *A, int4 *B, int4 *C, uint32_t *flags)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	const int sg = seg[h], c0 = cs[h];
	A[h] = make_int4(c0, sg, ce[h], pm[h]);
	// static mark of the cs sort's tie groups (an X neighbour with the same contig and start): only their members can have
	// their place in pg_gen_rep_pos's count depend on the reference's unstable sort (hazard H2b, k_rep_fill)
	const bool tie = (h > 0 && seg[h - 1] == sg && cs[h - 1] == c0) || (h + 1 < n && seg[h + 1] == sg && cs[h + 1] == c0);
	const uint32_t f = flags[h], nf = tie ? f | F_CSTIE : f & ~F_CSTIE;
	if (nf != f) flags[h] = nf;
	B[h] = make_int4(rk[h], gid[h], cds[h], pid[h]);
	C[h] = make_int4(rank[h], nex[h], offx[h], sori[h]);
}

// record C carries a copy of the rank (pg_flag_pseudo / pg_post_process change it)
__global__ __launch_bounds__(BLOCK) void k_pack_rank(const int32_t *rank, int n, int4 *C)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h < n) ((int32_t *)&C[h])[0] = rank[h];
}

// the static tie marks of the cs order from the records alone (after an order override moved hits)
__global__ __launch_bounds__(BLOCK) void k_cstie(const int4 *A, int n, uint32_t *flags)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	const int4 a = A[h];
	bool tie = false;
	if (h > 0) { const int4 p = A[h - 1]; tie = p.y == a.y && p.x == a.x; }
	if (!tie && h + 1 < n) { const int4 q = A[h + 1]; tie = q.y == a.y && q.x == a.x; }
	const uint32_t f = flags[h], nf = tie ? f | F_CSTIE : f & ~F_CSTIE;
	if (nf != f) flags[h] = nf;
}

// pm (record A, word 3) = running maximum of ce inside a contig, as a segmented scan over the records
struct InSegMaxA { const int4 *A; __device__ __forceinline__ SegMax operator()(int64_t i) const { const int4 a = A[i]; return SegMax{a.y, a.z}; } };
struct OutSegMaxA { int4 *A; __device__ __forceinline__ void operator()(int64_t i, SegMax in, SegMax) const { ((int32_t *)&A[i])[3] = in.v; } };

struct SweepView {
	const int4 *A, *B, *C; const int32_t *sori; const int2 *exon;
	uint32_t *flags; int32_t *pdom, *sdom;
	int32_t *pdom0; // MODE 3 only
	int n; double min_ov; int check_strand;
	int literal; // some exon list of the shard is not sorted and disjoint (introns shorter than 3 bp under U / V, read.c:59-62): every merge takes the reference's steps one by one
	int stage_c; // some hit of the shard has several exons: stage the C records with the others
	int64_t *hz;
	int64_t *slow_cnt; int32_t *slow_list; // work list for k_sweep_slow
	long long *prof; // PGA_SW_PROFILE builds only
	int dbg;         // PGA_SW_PROFILE builds only (PGA_SW_DBG): 1 = no exact merge in the epilogue, 2 = no merge in the pair evaluation, 4 = no score_dom arithmetic
	int32_t *hz_list; // hz[10] (= dcnt[14]) counts its entries
	int init_dom; // MODE 1, first sweep of a run: filtered hits get pid_dom = -1, score_dom = 0 (read.c:133-134) here, nobody wrote them before
	Gate gate; // the pg_shadow of an arc round inside pga_branch_loop may have nothing to do (no flag changed since the last one)
	// Round 6, live lists: the sweeps of the rounds (MODE 0) run over COMPACT copies of the records -- the hits without flt when the lists were built,
	// in X order, pm recomputed over them -- and reach the per-hit state (flag word, pid_dom) at the hit's X position through xmap; NULL: A / B / C
	// are the X-order records themselves.  A filtered hit takes no part in any pair (overlap.c:112,127), so the pairs are the same.
	const int32_t *xmap;
};
#define SW_X(v, h) ((v).xmap ? (v).xmap[h] : (h)) /* where the state of the hit at (compact) position h lives */

// pg_hit_overlap (overlap.c:6-42) step by step as the reference takes them: for shards with an exon list that is not sorted and disjoint
// (k_prepare looks), where the shortcuts of cds_inter_t would not add up to the same number
__device__ __forceinline__ int cds_inter_ref(const int2 *__restrict__ ex, int oa, int na, int ca, int ea_end, int ob, int nb, int cb, int eb_end)
{
	if (!(ca < eb_end && ea_end > cb)) return 0;
	int ia = 0, ib = 0, inter = 0;
	int2 xa = ex[oa], xb = ex[ob];
	while (true) {
		int s0 = ca + xa.x, e0 = ca + xa.y, s1 = cb + xb.x, e1 = cb + xb.y;
		bool adv_a;
		if (s0 < s1) {
			if (e0 < e1) { int o = e0 - s1; inter += o > 0 ? o : 0; adv_a = true; }
			else { inter += e1 - s1; adv_a = false; }
		} else {
			if (e1 < e0) { int o = e1 - s0; inter += o > 0 ? o : 0; adv_a = false; }
			else { inter += e0 - s0; adv_a = true; }
		}
		if (adv_a) { if (++ia >= na) break; xa = ex[oa + ia]; }
		else { if (++ib >= nb) break; xb = ex[ob + ib]; }
	}
	return inter;
}

// CDS intersection of hit a (exons ea[na], start ca) and hit b: pg_hit_overlap, overlap.c:6-42, for sorted disjoint exon lists.
// X says where the exon lists live (global memory, or the tile's copy in LDS); `need`: the merge may stop as soon as the
// intersection has reached it -- the sum only grows, and a pair's test is "at least so many bp" (overlap.c:132,134-136: any
// overlap for two hits of one gene, min_ov_ratio of the shorter CDS otherwise) -- so the value returned is the exact length when
// it is below `need` and some length >= need otherwise.  INT32_MAX asks for the exact length.
struct XGlobal { const int2 *__restrict__ p; __device__ __forceinline__ int2 operator()(int i) const { return p[i]; } };
template <class X>
__device__ __forceinline__ int cds_inter_t(const X ex, int oa, int na, int ca, int ea_end, int ob, int nb, int cb, int eb_end, int need, int *steps = nullptr)
{
	if (!(ca < eb_end && ea_end > cb)) return 0;
	if (na == 1 && nb == 1) { // single-exon x single-exon: plain interval intersection
		int s = ca > cb ? ca : cb, e = ea_end < eb_end ? ea_end : eb_end;
		return e > s ? e - s : 0;
	}
	int ia = 0, ib = 0, inter = 0;
	// A long list first skips the exons that end before the other hit begins: each would be a step of the reference's merge that
	// adds nothing (a gene inside an intron of a 140-exon gene is ~70 such steps, and a batch of lanes takes as long as its
	// longest merge).  For sorted disjoint lists the sum is the CDS intersection whatever the order the exon pairs are visited in.
	if (na > 8) {
		int lo = 0, hi = na; // first exon of a that ends behind cb
		while (lo < hi) { const int mid = (lo + hi) >> 1; if (ca + ex(oa + mid).y > cb) hi = mid; else lo = mid + 1; }
		if (lo >= na) return 0;
		ia = lo;
	}
	if (nb > 8) {
		int lo = 0, hi = nb;
		while (lo < hi) { const int mid = (lo + hi) >> 1; if (cb + ex(ob + mid).y > ca) hi = mid; else lo = mid + 1; }
		if (lo >= nb) return 0;
		ib = lo;
	}
	int2 xa = ex(oa + ia), xb = ex(ob + ib);
	// One step of the reference's merge (overlap.c:22-39) adds the overlap of the two current exons and advances the list whose
	// exon ends first (a on equal ends unless it started first).  Written without branches -- the lanes of a wave are in different
	// merges, and a step made of divergent branches cost ~70 instructions (870 cycles with three waves on a SIMD) -- and with three
	// shortcuts that leave the sum what it is for sorted disjoint lists (read.c:59-70 only ever moves forward): an exon present in
	// both lists (isoforms of one gene share most of theirs) advances both at once -- the reference's next step would find the
	// next exon of a clear of b's and advance b, adding nothing --; a list whose current exon starts behind the other hit's end has
	// nothing left to add; and the merge stops once `need` is reached.
	while (true) {
#ifdef PGA_SW_PROFILE
		if (steps) ++*steps;
#endif
		const int s0 = ca + xa.x, e0 = ca + xa.y, s1 = cb + xb.x, e1 = cb + xb.y;
		const int lo = s0 > s1 ? s0 : s1, hi = e0 < e1 ? e0 : e1, o = hi - lo;
		inter += o > 0 ? o : 0;
		const bool adv_a = e0 < e1 || (e0 == e1 && s0 >= s1), adv_b = !adv_a || (e0 == e1 && s0 == s1);
		ia += adv_a ? 1 : 0, ib += adv_b ? 1 : 0;
		const bool done = inter >= need || ia >= na || ib >= nb;
		xa = ex(oa + (ia < na ? ia : na - 1)), xb = ex(ob + (ib < nb ? ib : nb - 1));
		if (done || ca + xa.x >= eb_end || cb + xb.x >= ea_end) break;
	}
	return inter;
}
__device__ __forceinline__ int cds_inter(const int2 *__restrict__ ex, int literal, int oa, int na, int ca, int ea_end, int ob, int nb, int cb, int eb_end)
{
	return literal ? cds_inter_ref(ex, oa, na, ca, ea_end, ob, nb, cb, eb_end) : cds_inter_t(XGlobal{ex}, oa, na, ca, ea_end, ob, nb, cb, eb_end, INT32_MAX);
}

struct SwHit { // the hit a thread works for
	int sg, cs, ce, gid, cds, rank, nex, offx, weak; uint32_t fl; uint32_t sc;
};
struct SwBest { bool lose, iso; uint32_t best; int j, ov, pid, cds, cs; };

// Thread-per-hit form of one pair, used by k_sweep_slow: partner p (records a/b/c, flags fp, array index pi) of hit t;
// EARLIER: p precedes t in the array.  overlap.c:126-154 (pg_shadow) / 76-87 (pg_flt_ov_isoform).
__device__ __forceinline__ int4 sw_scse(int4 r) { return make_int4(r.y, r.x, r.z, r.w); } // record A -> (seg, cs, ce, pm)

template <int MODE, bool EARLIER>
__device__ __forceinline__ void sw_pair(const SweepView &v, const SwHit &t, SwBest &r, const int4 a, const uint32_t fp, const int4 b, const int4 c, int pi, bool ok)
{
	ok = ok && !(fp & PGA_F_FLT);
	if (v.check_strand) ok = ok && !((fp ^ t.fl) & PGA_F_REV);
	const bool same_gene = b.y == t.gid;
	const int x = !ok ? 0 : EARLIER ? cds_inter(v.exon, v.literal, c.z, c.y, a.y, a.z, t.offx, t.nex, t.cs, t.ce)
	                                : cds_inter(v.exon, v.literal, t.offx, t.nex, t.cs, t.ce, c.z, c.y, a.y, a.z);
	ok = ok && x > 0; // overlap.c:132
	const uint32_t sp = (uint32_t)b.x;
	// "i" of the reference is the later hit of the pair: i loses if (si < sj || (si == sj && rank_i > rank_j))
	const uint32_t s_i = EARLIER ? t.sc : sp, s_j = EARLIER ? sp : t.sc;
	const int rk_i = EARLIER ? t.rank : c.x, rk_j = EARLIER ? c.x : t.rank;
	bool i_loses = s_i < s_j || (s_i == s_j && rk_i > rk_j);
	{
		const int m = t.cds < b.z ? t.cds : b.z;
		// cov_short < min_ov_ratio (overlap.c:134-136).  For the default 0.5 the test is exactly 2x < m: x/m is within
		// 2^-32 of 0.5 only when it equals it, far above double rounding; other ratios take the IEEE division.
		bool too_short;
		if (v.min_ov == 0.5) too_short = 2u * (uint32_t)x < (uint32_t)m;
		else too_short = (double)x / (m > 0 ? m : 1) < v.min_ov;
		ok = ok && (same_gene || !too_short);
		const int wk_p = (int)((fp & PGA_F_WEAK_MASK) >> PGA_F_WEAK_SHIFT);
		const int wk_i = EARLIER ? t.weak : wk_p, wk_j = EARLIER ? wk_p : t.weak;
		i_loses = (!same_gene && wk_i != wk_j) ? wk_i > wk_j : i_loses; // overlap.c:139-147
	}
	const bool t_loses = ok && (EARLIER ? i_loses : !i_loses);
	r.lose = r.lose || t_loses;
	if (MODE == 3) r.iso = r.iso || (t_loses && same_gene); // pg_flt_ov_isoform's pairs are pg_shadow's same-gene pairs, with the same loser (overlap.c:76-87 vs 126-147)
	// dominator = best-scoring winner, first in array order on ties (overlap.c:150,153).  Earlier partners are visited in
	// DEscending index order, so an equal score replaces; later partners in ascending order, so it does not.
	const bool upd = t_loses && (EARLIER ? (sp > 0 && sp >= r.best) : (sp > r.best));
	if (t_loses && sp == r.best && sp > 0 && a.y == r.cs) { atomicAdd((unsigned long long *)&v.hz[3], 1ull); hz_note(&v.hz[10], v.hz_list, t.sg); } // hazard H3 (both winners in one (contig, cs) tie group), rare
	r.best = upd ? sp : r.best, r.j = upd ? pi : r.j, r.ov = upd ? x : r.ov, r.pid = upd ? b.w : r.pid, r.cds = upd ? b.z : r.cds, r.cs = upd ? a.y : r.cs;
}

__device__ __forceinline__ void wave_sync() // LDS hand-over between lanes of ONE wave (the LDS queue of a wave is in order)
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// exclusive prefix sum over the wave of a small count (c < 256), one ballot per bit: no LDS traffic, no cross-lane moves
__device__ __forceinline__ int wave_scan_small(int c, int *total)
{
	int off = 0, tot = 0;
#pragma unroll
	for (int b = 0; b < 8; ++b) {
		const unsigned long long mk = __ballot((c >> b) & 1);
		off += (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u)) << b;
		tot += __popcll(mk) << b;
	}
	*total = tot;
	return off;
}

// The interval-dominance sweep as an LDS pair list.
//
// A workgroup stages SW_TILE consecutive hits (cs order) plus SW_HALO neighbours on each side (36 B/hit, 52 when the C
// records are needed; coalesced 16-byte loads) and after ONE barrier its waves work independently: a wave owns 64 hits
// and looks at a window of SW_HALO more slots on each side.  Because hits are cs-sorted inside a contig, the later
// partners of a hit are a contiguous run; the runs are counted and expanded, k-th partners of all slots together, into a
// list of (earlier, later) slot pairs with at least one member among the wave's hits.  The list is evaluated one pair per
// lane (full lanes, every pair once -- a thread-per-hit walk evaluates each pair twice and runs as long as the busiest
// lane) and in chunks of at most SW_WCAP pairs, so a wave never runs out of list (round 5; a pile of isoforms used to send the
// whole wave to k_sweep_slow).  The outcome reaches the loser as ONE 64-bit LDS atomicMax of (winner's score rank, "lost"
// bit, inverted winner slot): the maximum is the best-scoring winner and, among equals, the first in array order
// (overlap.c:150).  Pairs across a wave or tile border are evaluated by both sides, each updating only its own hit: no global
// atomics, no inter-wave synchronisation.  Hits whose partners reach beyond the window go to a work list for k_sweep_slow.
//
// STAGE_C (some hit of the shard has several exons), round 5: **the exon lists of the tile live in LDS.**  pg_hit_overlap
// (overlap.c:6-42) is a two-pointer merge of two exon lists: a chain of dependent 8-byte loads, one pair per lane, every lane
// somewhere else -- out of global memory that was 87 M wave-level gathers and 8.2 ms on the 21.9 M isoform-rich hits of
// BASELINE configs[4] (0.5 TB/s of traffic: the kernel waited, it did not stream).  Now every live hit of the tile that
// overlaps a neighbour at all (the only ones a pair can name) has its list copied once, by eight lanes a list, into SW_XCAP
// entries of LDS behind two more barriers, and the merges run out of LDS; lists longer than SW_XMAX exons and what does not
// fit are still read where they are.  A merge stops as soon as the pair's test is decided (cds_inter_t: any overlap for two
// hits of one gene -- isoforms of a pile mostly share their first exon --, half of the shorter CDS otherwise), and the pairs of
// different genes, whose merges are the long ones, are set aside and evaluated together afterwards: a batch of 64 lanes takes
// as long as its longest merge.
// MODE 0: pg_shadow(cal_dom_sc=0); 1: pg_shadow(cal_dom_sc=1) (pg_post_process); (2 was pg_flt_ov_isoform alone: the two-launch form of stage A, retired in round 5)
// 3: stage A's two sweeps in ONE (read.c:248-254): pg_shadow(cal_dom_sc=1), the reset of read.c:249-253 and pg_flt_ov_isoform.  Both
//    enumerate the same overlapping pairs over the same flt flags (nothing between them sets flt); a same-gene pair has the same
//    loser in both (overlap.c:139 ignores weak_br and min_ov_ratio for one gene, and both compare score then rank), so the isoform
//    outcome is one more bit per hit: "lost a same-gene pair".  Written here: pid_dom0 (= the dominator's protein), pid_dom = -1,
//    score_dom, flt_iso_ov; the shadow flag ends 0 for every hit (read.c:252).  flt itself is NOT set here -- other workgroups
//    still read the flags of their halo hits -- but by the per-genome filter kernel that follows (overlap.c:89-91).
constexpr int SW_HALO = 32, SW_TILE = 256, SW_LDS = SW_TILE + 2 * SW_HALO, SW_WCAP = 512, SW_NW = SW_TILE / 64;
constexpr int SW_XCAP = 3584;            // exon entries of a tile in LDS (28 KB: three workgroups per CU; a tile of 320 hits with 7.7 exons each has 2 460, one in forty of the isoform-rich set more than this)
constexpr int SW_XMAX = 255;             // longest list staged (wave_scan_small counts to 255)
constexpr uint32_t SW_XNONE = 0xffffu;   // "not in LDS"
struct XLds { const int2 *p; __device__ __forceinline__ int2 operator()(int i) const { return p[i]; } };

constexpr int SW_NSTAMP = 12;
#ifdef PGA_SW_PROFILE
#define SW_DBG(bit) (!(v.dbg & (bit)))
#else
#define SW_DBG(bit) true
#endif
#ifdef PGA_SW_PROFILE // tuning build: s_memtime stamps of lane 0 of every wave at the phase boundaries
#define SW_STAMP(k) do { if (v.prof && (threadIdx.x & 63) == 0) v.prof[((long long)blockIdx.x * SW_NW + (threadIdx.x >> 6)) * SW_NSTAMP + (k)] = clock64(); } while (0)
#else
#define SW_STAMP(k) do { } while (0)
#endif

// epilogue of a hit, overlap.c:157-175.  The hit at index 0 of a genome is never reset (loop starts at 1, overlap.c:108).
template <int MODE>
__device__ __forceinline__ void sw_finish(const SweepView &v, int h, uint32_t fl, bool lose, bool has_dom, int pid_w, int ov, int cds_h, int cds_w, int sori_h, int sori_w, bool iso = false)
{
	if (MODE == 3) {
		const uint32_t nf = (fl & ~PGA_F_SHADOW) | (iso ? PGA_F_ISO_OV : 0u); // read.c:252; overlap.c:83,85
		if (nf != fl) v.flags[h] = nf;
		v.pdom0[h] = has_dom ? pid_w : -1, v.pdom[h] = -1; // read.c:251-252
		v.sdom[h] = has_dom ? (int32_t)(sori_h * (1.0 - (double)ov / cds_h) + sori_w * ((double)ov / cds_w) + .499) : -1; // overlap.c:161,170
		return;
	}
	uint32_t nf = (fl & F_HEAD) ? fl : (fl & ~PGA_F_SHADOW);
	if (lose) nf |= PGA_F_SHADOW;
	const int hx = MODE == 0 ? SW_X(v, h) : h; // (only the sweeps of the rounds run on the compact records)
	if (nf != fl) v.flags[hx] = nf;
	v.pdom[hx] = has_dom ? pid_w : -1;
	if (MODE == 1) {
		int sd = -1;
		if (has_dom) sd = (int32_t)(sori_h * (1.0 - (double)ov / cds_h) + sori_w * ((double)ov / cds_w) + .499); // overlap.c:170
		v.sdom[h] = sd;
	}
}

template <int MODE, bool STAGE_C, bool LISTS_IN_LDS>
__device__ __forceinline__ void sweep_tile(const SweepView &v)
{
	if (gate_closed(v.gate)) return; // (k_sweep_slow then finds an empty list and only resets the next sweep's counter)
	static_assert(2 * SW_HALO == 64 && SW_NW == 4 && SW_LDS <= 1024 && 64 + 2 * SW_HALO <= 128 && SW_WCAP >= 64 + 3 * SW_HALO, "the slots past SW_TILE are staged one array per wave; window-relative slot ids are packed in 7 bits, winner slots in 10; a level of the pair list has at most 64 + SW_HALO entries");
	constexpr bool STAGE_ORI = (MODE == 1 || MODE == 3) && !STAGE_C; // score_dom needs score_ori: out of the C records when they are staged, else staged alone
	// The exon lists go to LDS for the sweeps of stage A and pg_post_process (every hit is live, every pile is evaluated, every loser wants its
	// exact overlap).  The sweeps of the arc rounds (MODE 0) see what the filters left -- one isoform a gene -- and merge a few lists per tile:
	// they read them where they are and keep the smaller footprint (six workgroups per CU instead of three: 113 us against 33 on the
	// 2.7 M-hit isoform-rich shard when they staged as well).
	// Staging them costs LDS: three workgroups on a CU instead of six.  Where few hits overlap a neighbour at all (one isoform a gene) the
	// kernel is made of the record loads and little else, and the occupancy is worth more than the lists in LDS: k_sweep_lean (same code,
	// LISTS_IN_LDS = false) runs there -- the host picks by the shard's measured density, see k_list_density.
	constexpr bool STAGE_X = STAGE_C && LISTS_IN_LDS && (MODE == 1 || MODE == 3);
	__shared__ int4 sA[SW_LDS + 4], sB[SW_LDS], sC[STAGE_C ? SW_LDS : 1]; // sA: four sentinel slots close the array
	__shared__ uint32_t sF[SW_LDS];
	__shared__ int32_t sOri[STAGE_ORI ? SW_LDS : 1];
	__shared__ uint16_t sPairAll[SW_NW][SW_WCAP]; // (earlier slot - window start) << 7 | (later slot - first own slot): both < 96
	__shared__ unsigned long long sKeyAll[SW_NW][64];
	__shared__ uint32_t sIsoAll[MODE == 3 ? SW_NW : 1][64]; // MODE 3: the hit lost a same-gene pair (pg_flt_ov_isoform's mark)
	__shared__ int2 sX[STAGE_X ? SW_XCAP : 1];      // the tile's exon lists
	__shared__ uint16_t sXo[STAGE_X ? SW_LDS : 1];  // where a slot's list starts in sX, or SW_XNONE
	__shared__ int32_t sXt[2 * SW_NW], sXe[SW_NW];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	uint16_t *sPair = sPairAll[wave];
	unsigned long long *sKey = sKeyAll[wave];
	uint32_t *sIso = sIsoAll[MODE == 3 ? wave : 0];
	const int tile = blockIdx.x, base = tile * SW_TILE - SW_HALO;
	SW_STAMP(0);
	{
		const int g = base + tid;
		int4 a = make_int4(0, -2, 0, 0), b = make_int4(0, 0, 0, 0), c = b; // slots outside the array: contig -2, filtered
		uint32_t f = PGA_F_FLT;
		int32_t so = 0;
		if (g >= 0 && g < v.n) {
			a = v.A[g], b = v.B[g], f = v.flags[MODE == 0 ? SW_X(v, g) : g];
			if (STAGE_C) c = v.C[g];
			if (STAGE_ORI) so = v.sori[g];
		}
		// the 2 * SW_HALO slots past SW_TILE: one array per wave, so that no wave has more to stage than the others
		const int l2 = SW_TILE + lane, g2 = base + l2;
		const bool in2 = lane < 2 * SW_HALO && g2 >= 0 && g2 < v.n;
		if (wave == 0) sA[l2] = in2 ? v.A[g2] : make_int4(0, -2, 0, 0);
		else if (wave == 1) sB[l2] = in2 ? v.B[g2] : make_int4(0, 0, 0, 0);
		else if (wave == 2) sF[l2] = in2 ? v.flags[MODE == 0 ? SW_X(v, g2) : g2] : PGA_F_FLT;
		else if (STAGE_C) sC[l2] = in2 ? v.C[g2] : make_int4(0, 0, 0, 0);
		else if (STAGE_ORI) sOri[l2] = in2 ? v.sori[g2] : 0;
		sA[tid] = a, sB[tid] = b, sF[tid] = f;
		if (STAGE_C) sC[tid] = c;
		if (STAGE_ORI) sOri[tid] = so;
	}
	if (tid < 4) sA[SW_LDS + tid] = make_int4(0, -2, 0, 0);
	sKey[lane] = 0;
	if (MODE == 3) sIso[lane] = 0;
	SW_STAMP(1);
	__syncthreads();
	SW_STAMP(2);
	if (STAGE_X) {
		// The exon lists into LDS.  A slot is staged when it is live, its list is short enough and it overlaps a neighbour at all:
		// the earlier member l of an overlapping pair (l, m) sees cs[l+1] <= cs[m] < ce[l], the later one pm[m-1] >= ce[l] > cs[m]
		// (pm = running maximum of ce inside the contig), so both members of every pair the waves will name pass the test.
		// Wave w looks after the slots 64 w ... 64 w + 63 and SW_TILE + 16 w ... + 15.
		auto want = [&](int s) -> int {
			if (sF[s] & PGA_F_FLT) return 0;
			const int ne = sC[s].y;
			if (ne > SW_XMAX) return 0;
			const int4 a = sA[s], nx = sA[s + 1];
			bool near = nx.y == a.y && nx.x < a.z;
			if (!near && s > 0) { const int4 pv = sA[s - 1]; near = pv.y == a.y && pv.w > a.x; }
			return near ? ne : 0;
		};
		const int s1 = tid, s2 = SW_TILE + 16 * wave + (lane & 15);
		const int n1 = want(s1), n2 = lane < 16 ? want(s2) : 0;
		int t1, t2;
		int o1 = wave_scan_small(n1, &t1), o2 = wave_scan_small(n2, &t2);
		if (lane == 0) sXt[wave] = t1, sXt[SW_NW + wave] = t2;
		SW_STAMP(3);
		__syncthreads();
		{
			int b1 = 0, b2 = 0;
#pragma unroll
			for (int w = 0; w < SW_NW; ++w) { const int x1 = sXt[w], x2 = sXt[SW_NW + w]; b1 += w < wave ? x1 : 0, b2 += x1 + (w < wave ? x2 : 0); }
			o1 += b1, o2 += b2;
		}
		// every entry of sX first learns where it comes from (its slot's thread writes the source indices: LDS stores, nothing
		// waits for them), then the threads share the entries evenly: SW_XCAP / SW_TILE independent 8-byte loads each, all in flight
		// together (a loop over the slots that loads as it goes waits for memory once per slot: 5.7 ms instead of 8.2, no more)
		const bool in1 = n1 > 0 && o1 + n1 <= SW_XCAP, in2 = n2 > 0 && o2 + n2 <= SW_XCAP;
		sXo[s1] = (uint16_t)(in1 ? (uint32_t)o1 : SW_XNONE);
		if (lane < 16) sXo[s2] = (uint16_t)(in2 ? (uint32_t)o2 : SW_XNONE);
		if (in1) { const int z = sC[s1].z; for (int e = 0; e < n1; ++e) sX[o1 + e].x = z + e; }
		if (in2) { const int z = sC[s2].z; for (int e = 0; e < n2; ++e) sX[o2 + e].x = z + e; }
		{ // the entries in use end where the last list that fitted ends (offsets only grow: behind a list that does not fit none does)
			const int e1 = wave_max(in1 ? o1 + n1 : 0), e2 = wave_max(in2 ? o2 + n2 : 0);
			if (lane == 0) sXe[wave] = e1 > e2 ? e1 : e2;
		}
		SW_STAMP(4);
		__syncthreads();
		{
			int x_tot = 0;
#pragma unroll
			for (int w = 0; w < SW_NW; ++w) { const int x = sXe[w]; x_tot = x > x_tot ? x : x_tot; }
			constexpr int PER = SW_XCAP / SW_TILE;
			static_assert(PER * SW_TILE == SW_XCAP, "entries per thread");
			int src[PER]; int2 val[PER];
#pragma unroll
			for (int k = 0; k < PER; ++k) src[k] = sX[tid + k * SW_TILE].x;
#pragma unroll
			for (int k = 0; k < PER; ++k) if (tid + k * SW_TILE < x_tot) val[k] = v.exon[src[k]];
#pragma unroll
			for (int k = 0; k < PER; ++k) if (tid + k * SW_TILE < x_tot) sX[tid + k * SW_TILE] = val[k];
		}
		SW_STAMP(5);
		__syncthreads();
		SW_STAMP(6);
	}
	// ---- from here on every wave is on its own ----
	const int lo = SW_HALO + wave * 64, wend = lo + 64 + SW_HALO; // own slots [lo, lo+64), window [lo-SW_HALO, wend)
	// Later partners of a slot l: the run (l, e) with e = the first slot whose sort key (contig, cs) is not below
	// (contig_l, ce_l); the keys are non-decreasing, so four candidates are tested per round trip to LDS and the tests are
	// independent.  Lane t looks after its own slot and, the first SW_HALO lanes, after a slot of the left context, whose
	// run matters from the wave's first hit on.
	const int l1 = lo + lane, l0 = lo - SW_HALO + (lane & (SW_HALO - 1));
	int m1 = l1 + 1, m0 = lo, c1, c0;
	{
		const int4 a1 = sA[l1], a0 = sA[l0];
		const unsigned long long t1 = (unsigned long long)(uint32_t)a1.y << 32 | (uint32_t)a1.z, t0 = (unsigned long long)(uint32_t)a0.y << 32 | (uint32_t)a0.z;
		bool go1 = !(sF[l1] & PGA_F_FLT), go0 = lane < SW_HALO && !(sF[l0] & PGA_F_FLT);
		const int f1 = m1;
		while (go0 || go1) {
			unsigned long long q1[4], q0[4];
#pragma unroll
			for (int u = 0; u < 4; ++u) q1[u] = *(const unsigned long long *)&sA[m1 + u], q0[u] = *(const unsigned long long *)&sA[m0 + u];
			int n1 = 0, n0 = 0;
#pragma unroll
			for (int u = 0; u < 4; ++u) n1 += q1[u] < t1 ? 1 : 0, n0 += q0[u] < t0 ? 1 : 0;
			n1 = go1 ? n1 : 0, n0 = go0 ? n0 : 0;
			m1 += n1, m0 += n0;
			go1 = n1 == 4 && m1 < wend, go0 = n0 == 4 && m0 < wend;
		}
		c1 = (m1 < wend ? m1 : wend) - f1, c0 = (m0 < wend ? m0 : wend) - lo;
	}
	SW_STAMP(7);
	const int ovmode = v.min_ov == 0.5 ? 0 : v.min_ov <= 0.0 ? 1 : 2; // how cov_short < min_ov_ratio (overlap.c:134-136) is tested, see below
	// One pair: slot l precedes slot m in the array (l is "j", m is "i" of overlap.c:126-154 / 76-87).  DEFER: a pair of two
	// genes that needs a merge of exon lists is not evaluated but reported back (the caller collects those).
	auto eval = [&](auto DEFER, const uint32_t w) -> bool {
		const int l = lo - SW_HALO + (int)(w >> 7), m = lo + (int)(w & 127u);
		const uint32_t fj = sF[l], fi = sF[m];
		const int csj = sA[l].x, cej = sA[l].z, csi = sA[m].x, cei = sA[m].z;
		const int4 bj = sB[l], bi = sB[m]; // {rk, gid, cds, pid}
		bool ok = !((fj | fi) & PGA_F_FLT);
		if (v.check_strand) ok = ok && !((fj ^ fi) & PGA_F_REV);
		const bool same_gene = bj.y == bi.y;
		int x;
		{
			const int s0 = csj > csi ? csj : csi, e0 = cej < cei ? cej : cei;
			x = e0 > s0 ? e0 - s0 : 0; // single-exon x single-exon: the CDS intersection is the interval intersection; else: the spans must overlap (overlap.c:12)
		}
		ok = ok && x > 0;
		bool i_loses = (uint32_t)bi.x < (uint32_t)bj.x;
		// what the pair's test needs of the intersection: x >= need.  overlap.c:132 (x > 0) and, for two genes, cov_short <
		// min_ov_ratio (134-136): for the default 0.5 that is exactly 2x < min(cds) -- x/m is within 2^-32 of 0.5 only when it
		// equals it, far above double rounding --, for a ratio <= 0 never; other ratios take the IEEE division on the exact length.
		const int mn = bi.z < bj.z ? bi.z : bj.z;
		const bool thr = same_gene || ovmode != 2;
		int need = 1;
		if (!same_gene) {
			const uint32_t half = ((uint32_t)mn >> 1) + ((uint32_t)mn & 1u);
			need = ovmode == 0 ? (half > 0x7fffffffu ? INT32_MAX : (half > 1u ? (int)half : 1)) : ovmode == 1 ? 1 : INT32_MAX;
		}
		if (STAGE_C) {
			// the C records only when a pair of the wave needs them: multi-exon hits, or two hits with the same score key
			// (the same protein with the same score) whose order the rank decides
			const bool multi = ok && ((fj | fi) & F_MULTI), tie = ok && bi.x == bj.x;
			if (decltype(DEFER)::value && multi && need > 1) return true;
			if (__ballot(multi || tie)) {
				if (multi || tie) {
					const int4 cj = sC[l], ci = sC[m]; // {rank, n_exon, off_exon, score_ori}
					if (multi && SW_DBG(2)) {
						const uint32_t xj = STAGE_X ? sXo[l] : SW_XNONE, xi = STAGE_X ? sXo[m] : SW_XNONE;
						if (v.literal) x = cds_inter_ref(v.exon, cj.z, cj.y, csj, cej, ci.z, ci.y, csi, cei);
						else if (STAGE_X && xj != SW_XNONE && xi != SW_XNONE) x = cds_inter_t(XLds{sX}, (int)xj, cj.y, csj, cej, (int)xi, ci.y, csi, cei, need);
						else x = cds_inter_t(XGlobal{v.exon}, cj.z, cj.y, csj, cej, ci.z, ci.y, csi, cei, need);
					}
					if (tie) i_loses = ci.x > cj.x; // rank_i > rank_j
				}
			}
		} else {
			const bool tie = ok && bi.x == bj.x;
			if (__ballot(tie)) {
				if (tie) i_loses = v.C[base + m].x > v.C[base + l].x; // rank_i > rank_j
			}
		}
		if (thr) ok = ok && x >= need;
		else ok = ok && x > 0 && !((double)x / (mn > 0 ? mn : 1) < v.min_ov);
		{
			const uint32_t wk_i = fi & PGA_F_WEAK_MASK, wk_j = fj & PGA_F_WEAK_MASK;
			i_loses = (!same_gene && wk_i != wk_j) ? wk_i > wk_j : i_loses; // overlap.c:139-147
		}
		const int L = i_loses ? m : l, W = i_loses ? l : m, Lt = L - lo;
		if (ok && (unsigned)Lt < 64u) {
			const uint32_t rw = (uint32_t)(i_loses ? bj.x : bi.x);
			const unsigned long long key = (unsigned long long)rw << 32 | 0x80000000u | (uint32_t)(1023 - W);
			const unsigned long long old = atomicMax(&sKey[Lt], key);
			if (MODE == 3 && same_gene) sIso[Lt] = 1u; // (plain stores of one value)
			if (rw != 0 && (uint32_t)(old >> 32) == rw && sA[1023 - (int)((uint32_t)old & 1023u)].x == sA[W].x) { atomicAdd((unsigned long long *)&v.hz[3], 1ull); hz_note(&v.hz[10], v.hz_list, sA[L].y); } // hazard H3: two winners with one key; array order only decides between members of one (contig, cs) tie group
		}
		return false;
	};
	// The pair list, k-th partners of all slots together: their places follow from one ballot, no prefix sum needed.  A level
	// adds at most 64 + SW_HALO pairs; the list is evaluated whenever the next level might not fit.
	for (int k = 0;;) {
		int tot = 0;
		bool more = true;
		while (tot + 64 + SW_HALO <= SW_WCAP) {
			const unsigned long long mk0 = __ballot(c0 > k), mk1 = __ballot(c1 > k);
			if ((mk0 | mk1) == 0) { more = false; break; }
			const int at0 = tot + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk0, 0u));
			if (c0 > k) sPair[at0] = (uint16_t)((lane & (SW_HALO - 1)) << 7 | k);
			tot += __popcll(mk0);
			const int at1 = tot + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk1, 0u));
			if (c1 > k) sPair[at1] = (uint16_t)((SW_HALO + lane) << 7 | (lane + 1 + k));
			tot += __popcll(mk1);
			++k;
		}
		wave_sync();
		if (STAGE_C) {
			// first everything that is decided at once or after a few exons; the long merges (two genes) move to the front of the
			// list -- a write index never passes the read index of the same batch -- and run together afterwards
			int nh = 0;
			for (int p0 = 0; p0 < tot; p0 += 64) {
				const int p = p0 + lane;
				const uint32_t w = p < tot ? sPair[p] : 0u;
				const bool heavy = p < tot && eval(std::true_type{}, w);
				const unsigned long long mh = __ballot(heavy);
				if (heavy) sPair[nh + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mh >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mh, 0u))] = (uint16_t)w;
				nh += __popcll(mh);
			}
			wave_sync();
			for (int p = lane; p < nh; p += 64) (void)eval(std::false_type{}, sPair[p]);
		} else {
			for (int p = lane; p < tot; p += 64) (void)eval(std::false_type{}, sPair[p]);
		}
		wave_sync();
		if (!more) break;
	}
	SW_STAMP(8);
	int n_steps = 0;
	{
		// Epilogue of the wave's 64 hits (overlap.c:157-175).  score_dom needs the EXACT CDS overlap with the dominator
		// (overlap.c:161,170): one merge per lane out of LDS.  (Tried on the GPU and not kept: the exons of the wave's hits as
		// 64 x 7.7 independent tasks, one lane an exon, each bisecting the dominator's list and adding what overlaps into its hit's
		// LDS counter -- balanced where a lane's merge takes as long as the wave's longest gene, 30 steps on average against 7
		// of its own, but 17-20 k cycles per wave against 15.6 k for the merges: with three waves on a SIMD every LDS round trip and
		// every instruction is paid in full, and a bisection makes more of both than the merge steps it saves.)
		const int h = tile * SW_TILE + wave * 64 + lane, lh = lo + lane;
		const uint32_t fl = sF[lh];
		if (h < v.n && !(fl & PGA_F_FLT)) { // filtered hits keep stale shadow/pid_dom (overlap.c:112)
			const int4 a = sA[lh]; // {cs, seg, ce, pm}
			// partners outside the window?  (pm = running max of ce is non-decreasing inside a contig)
			const int4 w0 = sA[lo - SW_HALO], w1 = sA[wend - 1];
			const bool open = (w0.y == a.y && w0.w > a.x) || (w1.y == a.y && w1.x < a.z);
			if (open) {
				const unsigned long long at = atomicAdd((unsigned long long *)v.slow_cnt, 1ull);
				v.slow_list[at] = h;
			} else {
				const unsigned long long key = sKey[lane];
				const bool lose = key != 0, has_dom = (key >> 32) != 0;
				int pid_w = -1, ov = 0, cds_w = 1, so_w = 0, so_h = 0, cds_h = 1;
				if (has_dom) {
					const int W = 1023 - (int)(key & 1023u);
					const int4 bw = sB[W];
					pid_w = bw.w, cds_w = bw.z;
					if (MODE == 1 || MODE == 3) {
						const int4 aw = sA[W], cw = STAGE_C ? sC[W] : make_int4(0, 1, 0, sOri[W]), c2 = STAGE_C ? sC[lh] : make_int4(0, 1, 0, sOri[lh]);
						so_w = cw.w, so_h = c2.w, cds_h = sB[lh].z;
						const int s0 = aw.x > a.x ? aw.x : a.x, e0 = aw.z < a.z ? aw.z : a.z;
						ov = e0 > s0 ? e0 - s0 : 0;
						if (STAGE_C && ((fl | sF[W]) & F_MULTI) && SW_DBG(1)) { // the exact length this time; the earlier hit goes first, as in the pair evaluation
							const bool wf = W < lh;
							const uint32_t xw = STAGE_X ? sXo[W] : SW_XNONE, xh = STAGE_X ? sXo[lh] : SW_XNONE;
							if (STAGE_X && xw != SW_XNONE && xh != SW_XNONE && !v.literal)
								ov = cds_inter_t(XLds{sX}, (int)(wf ? xw : xh), wf ? cw.y : c2.y, wf ? aw.x : a.x, wf ? aw.z : a.z,
								                 (int)(wf ? xh : xw), wf ? c2.y : cw.y, wf ? a.x : aw.x, wf ? a.z : aw.z, INT32_MAX, &n_steps);
							else
								ov = cds_inter(v.exon, v.literal, wf ? cw.z : c2.z, wf ? cw.y : c2.y, wf ? aw.x : a.x, wf ? aw.z : a.z,
								               wf ? c2.z : cw.z, wf ? c2.y : cw.y, wf ? a.x : aw.x, wf ? a.z : aw.z);
						}
					}
				}
				sw_finish<MODE>(v, h, fl, lose, has_dom && SW_DBG(4), pid_w, ov, cds_h, cds_w, so_h, so_w, MODE == 3 && sIso[lane] != 0u);
			}
		} else if (MODE == 1 && v.init_dom && h < v.n) v.pdom[h] = -1, v.sdom[h] = 0;
		else if (MODE == 3 && h < v.n) v.pdom0[h] = -1, v.pdom[h] = -1, v.sdom[h] = 0; // a hit filtered before the sweeps keeps what read.c:133-134 gave it
	}
#ifdef PGA_SW_PROFILE
	{ const int mx = wave_max(n_steps), sm = wave_sum(n_steps); if (v.prof && lane == 0) v.prof[((long long)blockIdx.x * SW_NW + wave) * SW_NSTAMP + 10] = mx, v.prof[((long long)blockIdx.x * SW_NW + wave) * SW_NSTAMP + 11] = sm; }
#endif
	SW_STAMP(9);
}

template <int MODE, bool STAGE_C>
__global__ __launch_bounds__(SW_TILE) void k_sweep(SweepView v) { sweep_tile<MODE, STAGE_C, true>(v); }
// the sweeps of stage A and pg_post_process on a shard of multi-exon hits that rarely overlap: the exon lists stay in global memory
template <int MODE>
__global__ __launch_bounds__(SW_TILE) void k_sweep_lean(SweepView v) { sweep_tile<MODE, true, false>(v); }

// What k_sweep would stage, measured on a sample: the exons of the hits that overlap an X-order neighbour (the `want` test of sweep_tile), summed
// over every `stride`-th tile (one workgroup a CU: one round of cold loads).  Once per upload (the keys never change); the host reads the sum and picks k_sweep or k_sweep_lean.
__global__ __launch_bounds__(SW_TILE) void k_list_density(const int4 *A, const int4 *C, int n, int stride, int64_t *sum)
{
	const int h = blockIdx.x * stride * SW_TILE + threadIdx.x;
	int ne = 0;
	if (h < n) { // (four independent loads: a sampled tile is cold in every cache and in the TLB)
		const int4 a = A[h], nx = A[h + 1 < n ? h + 1 : h], pv = A[h > 0 ? h - 1 : h], c = C[h];
		const bool near = (h + 1 < n && nx.y == a.y && nx.x < a.z) || (h > 0 && pv.y == a.y && pv.w > a.x);
		ne = near ? c.y : 0;
	}
	ne = wave_sum(ne);
	if ((threadIdx.x & 63) == 0 && ne) atomicAdd((unsigned long long *)sum, (unsigned long long)ne);
}

// The rare hits k_sweep could not finish inside its LDS window: one thread per listed hit walks all its partners in
// global memory, in both directions (the plain thread-per-hit formulation of the sweep).
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_sweep_slow(SweepView v, long long *next_cnt, int32_t *clear2 = nullptr /* two more words to clear: the hand-out counters of the gene kernels behind the arc rounds' sweep */)
{
	const long long n_slow = *v.slow_cnt;
	if (blockIdx.x == 0 && threadIdx.x == 0) { *next_cnt = 0; if (clear2) clear2[0] = 0, clear2[1] = 0; } // the counter the NEXT sweep will use (ping-pong; nobody reads it now)
	for (long long q = blockIdx.x * (long long)BLOCK + threadIdx.x; q < n_slow; q += (long long)gridDim.x * BLOCK) {
		const int h = v.slow_list[q];
		const uint32_t fl = v.flags[MODE == 0 ? SW_X(v, h) : h];
		SwHit t;
		const int4 ch = v.C[h];
		{
			const int4 a = sw_scse(v.A[h]), b = v.B[h];
			t.sg = a.x, t.cs = a.y, t.ce = a.z, t.gid = b.y, t.cds = b.z, t.rank = ch.x, t.nex = ch.y, t.offx = ch.z;
			t.weak = (int)((fl & PGA_F_WEAK_MASK) >> PGA_F_WEAK_SHIFT), t.fl = fl;
			t.sc = (uint32_t)b.x;
		}
		SwBest r = { false, false, 0, -1, 0, -1, 0, -1 };
		// partners before h: every j with ce_j > cs_h.  pm (running max of ce) is non-decreasing inside a contig, so the
		// walk stops at the first j whose pm is <= cs_h.
		for (int j = h - 1; j >= 0; --j) {
			const int4 a = sw_scse(v.A[j]);
			if (a.x != t.sg || a.w <= t.cs) break;
			sw_pair<MODE, true>(v, t, r, a, v.flags[MODE == 0 ? SW_X(v, j) : j], v.B[j], v.C[j], j, a.z > t.cs);
		}
		// partners after h: every i with cs_i < ce_h
		for (int i = h + 1; i < v.n; ++i) {
			const int4 a = sw_scse(v.A[i]);
			if (a.x != t.sg || a.y >= t.ce) break;
			sw_pair<MODE, false>(v, t, r, a, v.flags[MODE == 0 ? SW_X(v, i) : i], v.B[i], v.C[i], i, true);
		}
		sw_finish<MODE>(v, h, fl, r.lose, r.best > 0, r.pid, r.ov, t.cds, r.cds, ch.w, (MODE == 1 || MODE == 3) && r.best > 0 ? v.C[r.j].w : 0, r.iso);
	}
}
