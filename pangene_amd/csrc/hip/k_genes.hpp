// k_genes.hpp -- pg_gen_arc (graph.c:87-177) and pg_mark_branch_flt_hit (branch.c:108-145) on a GENE-major index, without a global sort.
// Included by pga_backend.hip (one translation unit); uses the context types, BLOCK / WAVE and dev_prims.hpp from there.
//
// The reference sorts all 2 x (#adjacencies) temp arcs by (v, w) every round (graph.c:127,151).  The source vertex of a temp
// arc is an oriented GENE, and which hits belong to a gene never changes: with the hits indexed gene-major once per run
// (Z order = (gene, genome, X position); zoff[] = CSR offsets), every temp arc leaving the two vertices of gene g comes from
// a hit of g -- the arc v -> w from the hit that plays v (its successor adjacency), the mirrored arc w^1 -> v^1 from the hit
// that plays w (its predecessor adjacency).  So:
//   (A) the scan over the cm order that finds each walkable hit's walkable predecessor also leaves, per hit, two 16-byte
//       half-arc records (successor / predecessor adjacency: target vertex, distance, the two scores);
//   (B) one workgroup per gene reads its hits' half-arcs (sequential in Z order), collapses the (arc, genome) duplicates
//       (graph.c:128-145; genomes are contiguous inside a gene), sums over the genomes in an LDS hash table keyed by
//       (orientation, target) (graph.c:153-169: integer sums, order-free), sorts the few entries and writes the gene's arcs;
//       it also counts the gene's walkable hits and genomes (graph.c:125-126) -- no global atomics except one per gene;
//   (C) the arcs of gene g stay where the workgroup put them -- its own stretch [2 zoff[g], ...) of the table arrays, sorted by
//       target -- and the per-vertex ranges vs[] / ve[] point there: the steps that read the table (branch marking, hit marking,
//       the degree filter) go through those ranges, so the table need not be contiguous.  Only when somebody wants it as ONE
//       array sorted by x = v << 32 | w (the exchange of a sharded run, the host after the last round) do a scan over the segments
//       and a copy compact it: segments are numbered in gene order (vertex.c:85-94), so the stretches are already in x order.
// pg_mark_branch_flt_hit then needs no walk at all: every hit looks its own two half-arcs up in the arc lists of its own
// gene's vertices and raises its own weak_br.
#pragma once

// (the half-arc record format -- HA_NONE, HA_TAG_SHIFT, ha_valid, ha_walk -- lives in k_common.hpp: k_branch.hpp reads the records too)

// ------------------------------------------------------------------------------------------------
// the gene-major index (static per run: rebuilt only when an order override moves hits)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_zkey(const int32_t *gid, int n, uint64_t *key, uint32_t *val)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h < n) key[h] = (uint64_t)(uint32_t)gid[h], val[h] = (uint32_t)h;
}

// LIVE LISTS (pga_ctx::live_on): the index and the walk's list over the hits that are not filtered when they are built.  One scan over the X order
// marks the members (F_MEMBER travels with the hit from then on) and numbers them (lx[x] = members before x, lx[n] = their number), a second one
// over the cm order writes their X positions in that order (ylist); the sort keys of the index are emitted for members only.
struct InLiveX { const uint32_t *flags; __device__ __forceinline__ I32 operator()(int64_t i) const { return I32{(flags[i] & PGA_F_FLT) ? 0 : 1}; } };
struct OutLiveX {
	uint32_t *flags; int32_t *lx; const int32_t *gid; uint64_t *key; uint32_t *val; int64_t n;
	__device__ __forceinline__ void operator()(int64_t i, I32 in, I32 ex) const
	{
		const uint32_t f = flags[i], nf = (f & PGA_F_FLT) ? (f & ~F_MEMBER) : (f | F_MEMBER);
		if (nf != f) flags[i] = nf;
		lx[i] = ex.v;
		if (!(f & PGA_F_FLT)) key[ex.v] = (uint64_t)(uint32_t)gid[i], val[ex.v] = (uint32_t)i;
		if (i == n - 1) lx[n] = in.v;
	}
};
struct InLiveY { const uint32_t *flags; const int32_t *yperm; __device__ __forceinline__ I32 operator()(int64_t y) const { return I32{(flags[yperm[y]] & PGA_F_FLT) ? 0 : 1}; } };
struct OutLiveY {
	const uint32_t *flags; const int32_t *yperm; int32_t *ylist;
	__device__ __forceinline__ void operator()(int64_t y, I32, I32 ex) const { const int x = yperm[y]; if (!(flags[x] & PGA_F_FLT)) ylist[ex.v] = x; }
};
// the sweep's records of the members, compact (SweepView::xmap): record i = the member with lx == i
__global__ __launch_bounds__(BLOCK) void k_live_records(const uint32_t *flags, const int32_t *lx, const int4 *A, const int4 *B, const int4 *C, int n, int4 *cA, int4 *cB, int4 *cC, int32_t *cx)
{
	const int x = blockIdx.x * BLOCK + threadIdx.x;
	if (x >= n || !(flags[x] & F_MEMBER)) return;
	const int i = lx[x];
	cA[i] = A[x], cB[i] = B[x], cC[i] = C[x], cx[i] = x;
}
// ... of the hits an order override has moved (lpos: their places among the members, k_ovl_pos; -1: not a member)
__global__ __launch_bounds__(BLOCK) void k_ovl_records(const int32_t *ov_pos, const int32_t *lpos, int64_t t, const int4 *A, const int4 *B, const int4 *C, int4 *cA, int4 *cB, int4 *cC, int32_t *cx)
{
	const int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (k >= t || lpos[k] < 0) return;
	const int i = lpos[k], x = ov_pos[k];
	cA[i] = A[x], cB[i] = B[x], cC[i] = C[x], cx[i] = x;
}
// pm of the compact records over the listed places (a contig's members are consecutive in the list's order of places; -1 entries are no-ops)
struct InSegMaxListL { const int4 *A; const int32_t *lpos; __device__ __forceinline__ SegMax operator()(int64_t i) const { const int p = lpos[i]; if (p < 0) return SegMax{SEG_EMPTY, 0}; const int4 a = A[p]; return SegMax{a.y, a.z}; } };
struct OutSegMaxListL { int4 *A; const int32_t *lpos; __device__ __forceinline__ void operator()(int64_t i, SegMax in, SegMax) const { const int p = lpos[i]; if (p >= 0) ((int32_t *)&A[p])[3] = in.v; } };

// the members' number for the host (dcnt[10]) with the other counters
__global__ void k_mail_live(const int32_t *lx, int64_t n, int64_t *dcnt, int64_t *host_box)
{
	if (threadIdx.x == 0) dcnt[10] = lx[n];
	__syncthreads();
	if (threadIdx.x < 16) sys_store(&host_box[threadIdx.x], dcnt[threadIdx.x]);
}

// The gene-major index as separate 4-byte planes (each kernel reads only the planes it needs):
//   zx[z] = X position; zy[z] = local genome << 1 | rev, bit 31 = the only hit of its gene in its genome (nearly all are: the
//   group logic of the gene kernels has nothing to do for it); zg[z] = gene; zst[z] = {cm, contig segment} (static);  zpos[x] = z
struct ZIndex { int32_t *zx, *zy, *zg; int2 *zst; int32_t *zpos; };
__global__ __launch_bounds__(BLOCK) void k_zrec(const uint32_t *perm, const uint64_t *ks, const int32_t *ctg_base, int n_genome, const uint32_t *flags, const int32_t *cm, const int32_t *seg, int n, ZIndex o)
{
	// Three gathers through the permutation per hit (flags, cm, seg: a 128-byte line each; round 3 made seven and moved 717 B/hit): the
	// genome follows from the contig segment (a search in the small ctg_base table), and the neighbours' (gene, genome) -- "is this hit
	// alone in its group?" -- come from the neighbouring LANES; only the wave's two border lanes fetch theirs.
	const int z = blockIdx.x * BLOCK + threadIdx.x, lane = threadIdx.x & 63;
	const bool v = z < n;
	const int x = v ? (int)perm[z] : 0;
	const int sg = v ? seg[x] : 0;
	const int gn = v ? genome_of(ctg_base, n_genome, sg) : -1;
	const unsigned long long kz = v ? ks[z] : ~0ull;
	int gn_p = __shfl_up(gn, 1, WAVE), gn_n = __shfl_down(gn, 1, WAVE);
	unsigned long long k_p = (unsigned long long)(unsigned)__shfl_up((int)(unsigned)(kz >> 32), 1, WAVE) << 32 | (unsigned)__shfl_up((int)(unsigned)kz, 1, WAVE);
	unsigned long long k_n = (unsigned long long)(unsigned)__shfl_down((int)(unsigned)(kz >> 32), 1, WAVE) << 32 | (unsigned)__shfl_down((int)(unsigned)kz, 1, WAVE);
	if (lane == 0) { const bool h = v && z > 0; k_p = h ? ks[z - 1] : ~0ull; gn_p = h ? genome_of(ctg_base, n_genome, seg[(int)perm[z - 1]]) : -1; }
	if (lane == 63) { const bool h = v && z + 1 < n; k_n = h ? ks[z + 1] : ~0ull; gn_n = h ? genome_of(ctg_base, n_genome, seg[(int)perm[z + 1]]) : -1; }
	if (!v) return;
	const bool grp_prev = z > 0 && k_p == kz && gn_p == gn, grp_next = z + 1 < n && k_n == kz && gn_n == gn;
	o.zx[z] = x, o.zg[z] = (int)kz;
	o.zy[z] = gn << 1 | ((flags[x] & PGA_F_REV) ? 1 : 0) | ((grp_prev || grp_next) ? 0 : (int)0x80000000);
	o.zst[z] = make_int2(cm[x], sg);
	o.zpos[x] = z;
}

__global__ __launch_bounds__(BLOCK) void k_zoff(const uint64_t *ks, int n, int Q, int32_t *zoff) // zoff[g] = first z with gene >= g
{
	int g = blockIdx.x * BLOCK + threadIdx.x;
	if (g > Q) return;
	int lo = 0, hi = n;
	while (lo < hi) { const int mid = (lo + hi) >> 1; if (ks[mid] < (uint64_t)g) lo = mid + 1; else hi = mid; }
	zoff[g] = lo;
}

// What the walk reads of a hit, in Y (cm) order, ONE 32-byte record: W0 = {contig segment, vertex = gene << 1 | rev, cm, score_ori},
// W1 = {score_dom, gene of pid_dom0's protein (-1: none), gene-major position z, 0}.  Round 4 read two 16-byte records out of two
// arrays and the position out of a third: with one hit in five walkable (the late rounds of a bacterial shard) that is three 128-byte
// lines per walkable hit for 36 useful bytes -- 103 B/hit read where ~45 are needed.  With virtual contigs (vfirst != NULL) "segment" is the
// contig's FIRST piece and "cm" the low word of the true 64-bit cm (k_pack_yrec in k_arcs.hpp says why that is enough).
struct WrecSrc { const int32_t *yperm, *seg, *gid, *cm, *sori, *sdom, *pdom0, *prot_gid; const uint32_t *flags; const int32_t *zpos; const int32_t *vfirst; const int64_t *vbase; };
__device__ __forceinline__ void pack_wrec_one(const WrecSrc &s, int64_t y, int4 *W)
{
	const int a = s.yperm[y], p0 = s.pdom0[a];
	int sg = s.seg[a], cmv = s.cm[a];
	if (s.vfirst) { cmv = (int)(unsigned)((unsigned long long)s.vbase[sg] + (unsigned long long)(long long)cmv); sg = s.vfirst[sg]; }
	W[2 * y] = make_int4(sg, s.gid[a] << 1 | ((s.flags[a] & PGA_F_REV) ? 1 : 0), cmv, s.sori[a]);
	W[2 * y + 1] = make_int4(s.sdom[a], p0 < 0 ? -1 : s.prot_gid[p0], s.zpos[a], 0);
}
__global__ __launch_bounds__(BLOCK) void k_pack_wrec(WrecSrc s, int n, int4 *W)
{
	const int y = blockIdx.x * BLOCK + threadIdx.x;
	if (y < n) pack_wrec_one(s, y, W);
}
// ... of the positions an order override moved hits to or from (whole contigs: a contig has the same index range in both orders)
__global__ __launch_bounds__(BLOCK) void k_pack_wrec_list(WrecSrc s, const int32_t *pos, int64_t T, int4 *W)
{
	const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (t < T && pos[t] >= 0) pack_wrec_one(s, pos[t], W); // (< 0: a listed hit that is not in the live lists)
}

// ------------------------------------------------------------------------------------------------
// (A) the walk (graph.c:103-122), cm order: previous walkable hit -> the two half-arc records of every walkable hit
// ------------------------------------------------------------------------------------------------
// One launch (round 5; rounds 2-4: a three-launch scan -- reduce, tile sums, output step -- that read every flag twice).  The previous
// walkable hit of a position is the running maximum of the walkable marks before it; what a tile needs from the tiles in front of it is
// ONE number, the last walkable position before its first, and with walkable hits every few positions that is a glance backwards
// (wave 0 looks at 64 positions at a time while the other waves load the tile's own flags), not a scan.
// Positions a thread (WK_IPT): 8 / 4 / 2 / 1 -> 277 / 246 / 227 / 214 us at 12.1 M hits, 31 / 27 / 25 / 23 us at 955 k -- more workgroups in flight beat
// fewer carries --, but at 96.6 M hits (configs[3] on one GPU: rounds in which few hits are walkable, so the look backwards for the carry is
// most of a workgroup's work) one position a thread took 891 us a launch where four take 439.  The host picks by the shard's size (WK_FEW_FROM).
constexpr int WK_FEW_FROM = 1 << 25; // hits from which a thread takes four positions
struct Walk {
	const uint32_t *__restrict__ flags; const int32_t *__restrict__ yperm; const int4 *__restrict__ W; const int32_t *__restrict__ g2s;
	uint32_t *__restrict__ hfk, *__restrict__ hbk; int4 *__restrict__ hfp, *__restrict__ hbp; // key word / payload {distance, score of this hit, score of the other, 0} of the two half-arcs
	uint32_t tag; int ori; int n; int64_t *dcnt; int32_t *hz_list; Gate gate;
};
__device__ __forceinline__ int walk_score(const int4 w0, const int4 w1, int ori, const int32_t *g2s)
{ // pg_get_score, graph.c:82-85: score_ori unless the dominator's gene is not a vertex and score_dom is at least as large
	return (ori || w0.w > w1.x || w1.y < 0 || g2s[w1.y] >= 0) ? w0.w : w1.x;
}
template <int WK_IPT>
__global__ __launch_bounds__(BLOCK) void k_walk(Walk a)
{
	constexpr int WK_TILE = BLOCK * WK_IPT;
	__shared__ int32_t s_wave[BLOCK / WAVE], s_carry;
	if (gate_closed(a.gate)) return;
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const int64_t tile0 = (int64_t)blockIdx.x * WK_TILE, base = tile0 + (int64_t)tid * WK_IPT;
	int mark[WK_IPT];
#pragma unroll
	for (int k = 0; k < WK_IPT; ++k) { const int64_t i = base + k; mark[k] = (i < a.n && !(a.flags[a.yperm[i]] & (PGA_F_FLT | PGA_F_SHADOW))) ? (int)i : -1; } // graph.c:108
	if (w == 0) { // the last walkable position in front of the tile
		int carry = -1;
		for (int64_t j0 = tile0 - 1; j0 >= 0 && carry < 0; j0 -= 64) {
			const int64_t j = j0 - lane;
			const bool ok = j >= 0 && !(a.flags[a.yperm[j]] & (PGA_F_FLT | PGA_F_SHADOW));
			const unsigned long long mk = __ballot(ok);
			if (mk) carry = (int)(j0 - (__ffsll((long long)mk) - 1));
		}
		if (lane == 0) s_carry = carry;
	}
	int t_max = mark[0];
#pragma unroll
	for (int k = 1; k < WK_IPT; ++k) t_max = t_max > mark[k] ? t_max : mark[k];
	I32 incl = wave_scan_incl(I32{t_max}, OpMax{}, lane);
	if (lane == 63) s_wave[w] = incl.v;
	__syncthreads();
	int run = s_carry; // exclusive running maximum in front of this thread's items
	for (int k = 0; k < w; ++k) run = run > s_wave[k] ? run : s_wave[k];
	{ const int pv = incl.shfl_up(1).v; if (lane > 0) run = run > pv ? run : pv; }
	// every item's predecessor first (registers only), then the records of all items in flight together, then the gene lookups of the
	// scores, then the stores: four dependent trips to memory instead of two per item
	int prev[WK_IPT];
#pragma unroll
	for (int k = 0; k < WK_IPT; ++k) { prev[k] = run; run = mark[k] >= 0 ? mark[k] : run; }
	int4 A0[WK_IPT], A1[WK_IPT], B0[WK_IPT], B1[WK_IPT];
#pragma unroll
	for (int k = 0; k < WK_IPT; ++k) {
		const int i = mark[k] >= 0 ? mark[k] : 0, pp = prev[k] >= 0 ? prev[k] : i;
		if (mark[k] >= 0) A0[k] = a.W[2 * (int64_t)i], A1[k] = a.W[2 * (int64_t)i + 1], B0[k] = a.W[2 * (int64_t)pp], B1[k] = a.W[2 * (int64_t)pp + 1];
	}
	int SA[WK_IPT], SB[WK_IPT];
#pragma unroll
	for (int k = 0; k < WK_IPT; ++k)
		if (mark[k] >= 0 && prev[k] >= 0 && B0[k].x == A0[k].x) SA[k] = walk_score(A0[k], A1[k], a.ori, a.g2s), SB[k] = walk_score(B0[k], B1[k], a.ori, a.g2s);
#pragma unroll
	for (int k = 0; k < WK_IPT; ++k) {
		if (mark[k] < 0) continue;
		const int p = prev[k];
		const int4 a0 = A0[k], a1 = A1[k], b0 = B0[k], b1 = B1[k];
		const int zi = a1.z, zp = b1.z;
		uint32_t key = a.tag << HA_TAG_SHIFT | HA_NONE;
		if (p >= 0 && b0.x == a0.x) { // same contig: adjacency p -> i (graph.c:113-121)
			const uint32_t wv = (uint32_t)a0.y, v = (uint32_t)b0.y;
			const int sa = SA[k], sb = SB[k], d = (int)((unsigned)a0.z - (unsigned)b0.z); // (low words of 64-bit coordinates when the shard has virtual contigs)
			if (a0.z == b0.z) { atomicAdd((unsigned long long *)&a.dcnt[5], 1ull); hz_note(&a.dcnt[14], a.hz_list, a0.x); } // hazard H2a: equal cm
			a.hfk[zp] = a.tag << HA_TAG_SHIFT | wv, a.hfp[zp] = make_int4(d, sb, sa, 0); // v -> w,     s1 = score(v), s2 = score(w) (graph.c:117)
			key = a.tag << HA_TAG_SHIFT | (v ^ 1u), a.hbp[zi] = make_int4(d, sa, sb, 0); // w^1 -> v^1, s1 = score(w), s2 = score(v) (graph.c:119)
		}
		a.hbk[zi] = key; // written for EVERY walkable hit: "carries the round's tag" = "is walkable in this round"
	}
}

// The walk again for the contigs an order override has just rearranged (pga_override_order: whole contigs, a few thousand hits of
// millions), with the tag of the walk that stands: every listed position writes its OWN two records -- the adjacency with its nearest
// walkable predecessor and with its nearest walkable successor inside the contig, found by looking along the contig -- or, when it is
// not walkable, the tag 0 that no round carries.  Round 4 walked the whole shard again after every override: 109 walks per pass of
// the isoform-rich 200-assembly set, 47 of them in front of an arc round that needs one anyway.
__global__ __launch_bounds__(BLOCK) void k_walk_list(Walk a, const int32_t *pos, int64_t T)
{
	const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (t >= T) return;
	const int i = pos[t];
	if (i < 0) return; // a listed hit that is not in the live lists
	auto walkable = [&](int y) { return !(a.flags[a.yperm[y]] & (PGA_F_FLT | PGA_F_SHADOW)); };
	const int4 a0 = a.W[2 * (int64_t)i], a1 = a.W[2 * (int64_t)i + 1];
	const int zi = a1.z;
	if (!walkable(i)) { a.hbk[zi] = 0, a.hfk[zi] = 0; return; }
	int p = -1, q = -1;
	for (int j = i - 1; j >= 0; --j) { if (a.W[2 * (int64_t)j].x != a0.x) break; if (walkable(j)) { p = j; break; } }
	for (int j = i + 1; j < a.n; ++j) { if (a.W[2 * (int64_t)j].x != a0.x) break; if (walkable(j)) { q = j; break; } }
	const int sa = walk_score(a0, a1, a.ori, a.g2s);
	uint32_t key = a.tag << HA_TAG_SHIFT | HA_NONE;
	if (p >= 0) { // adjacency p -> i as the later hit sees it (graph.c:119)
		const int4 b0 = a.W[2 * (int64_t)p], b1 = a.W[2 * (int64_t)p + 1];
		const int sb = walk_score(b0, b1, a.ori, a.g2s), d = (int)((unsigned)a0.z - (unsigned)b0.z);
		if (a0.z == b0.z) { atomicAdd((unsigned long long *)&a.dcnt[5], 1ull); hz_note(&a.dcnt[14], a.hz_list, a0.x); } // hazard H2a: equal cm
		key = a.tag << HA_TAG_SHIFT | ((uint32_t)b0.y ^ 1u), a.hbp[zi] = make_int4(d, sa, sb, 0);
	}
	a.hbk[zi] = key;
	if (q >= 0) { // adjacency i -> q as the earlier hit sees it (graph.c:117)
		const int4 c0 = a.W[2 * (int64_t)q], c1 = a.W[2 * (int64_t)q + 1];
		const int sc = walk_score(c0, c1, a.ori, a.g2s), d = (int)((unsigned)c0.z - (unsigned)a0.z);
		a.hfk[zi] = a.tag << HA_TAG_SHIFT | (uint32_t)c0.y, a.hfp[zi] = make_int4(d, sa, sc, 0);
	} else a.hfk[zi] = 0;
}

// ------------------------------------------------------------------------------------------------
// (B) one wave per gene (one workgroup for a gene with many hits or many neighbours)
// ------------------------------------------------------------------------------------------------
constexpr int GA_CAP_WAVE = 128, GA_CAP = 512; // distinct (orientation, target) pairs of one gene the LDS tables hold; more than GA_CAP = the round takes the sort path
constexpr int GA_WAVE_HITS = 512;              // genes with more hits go to the second kernel straight away
constexpr int GA_BIG_STAGE = 2048;             // hits the workgroup kernel stages at a time

struct GeneArcs {
	const int32_t *zy; const int32_t *zoff; const uint32_t *hfk, *hbk; const int4 *hfp, *hbp; const int32_t *g2s;
	int Q, S; uint32_t tag; int cap_log2; // table size actually used (<= GA_CAP; tests shrink it to reach the overflow paths)
	int32_t *seg_cnt, *seg_gid;       // [2S] n_genome then tot_cnt (graph.c:125-126); [S] gene of each segment
	pga_arc_part_t *stage; int4 *gmeta; // arcs of a gene at stage[gmeta.x ...): gmeta = {base, #arcs leaving (sid, +), #arcs leaving (sid, -), 0}
	// what the branch steps read, at the same (sparse) positions: x, rounded s1 (graph.c:171), target gene, weak_br = 0; per oriented vertex its range, degree, "has a weak arc" = 0
	uint64_t *ax; int32_t *s1, *agid; uint8_t *aw; int32_t *vs, *ve, *deg; uint8_t *vwk;
	int32_t *h_round;                   // pinned host memory (or NULL): seg_cnt[2S] then deg[2S] for the host, written straight from here
	int32_t *big_list;                  // [Q] the genes left to the workgroup kernel, in the order the first kernel's workgroups gave up on them
	int32_t *big_ctl;                   // [2] their number; how many of them have been taken (both cleared by the k_sweep_slow in front of every arc round)
	int64_t *dcnt;                      // [3] invariant, [9] genes that overflowed GA_CAP
	Gate gate; int32_t *tag_out;        // (pga_branch_loop) the round may have nothing to do; tag_out: where a round that does run leaves its tag
};


template <int CAP, int STAGE> struct GeneTable {
	uint32_t key[CAP]; int32_t ng[CAP], tot[CAP]; unsigned long long sd[CAP], s1[CAP], s2[CAP]; uint16_t dense[CAP];
	int32_t zy[STAGE]; uint32_t fx[STAGE], bx[STAGE]; // staged window of the gene's hits: genome << 1 | rev, word 0 of the two half-arcs
	int n_tot, n_gen, over, m, m0, base;
};

// The group logic (which hits of the gene lie in the same genome, which of their half-arcs share a key) only needs three words
// per hit; they are staged in LDS a window at a time, so that looking at a neighbour costs no trip to memory.  Outside the
// window (groups that straddle a window border: rare) the same words come from global memory.
template <int CAP, int STAGE> struct GeneWin {
	const GeneArcs &a; const GeneTable<CAP, STAGE> &T; int lo, hi;
	__device__ __forceinline__ int zy(int z) const { return ((z >= lo && z < hi) ? T.zy[z - lo] : a.zy[z]) & 0x7fffffff; }
	__device__ __forceinline__ uint32_t fx(int z) const { return (z >= lo && z < hi) ? T.fx[z - lo] : a.hfk[z]; }
	__device__ __forceinline__ uint32_t bx(int z) const { return (z >= lo && z < hi) ? T.bx[z - lo] : a.hbk[z]; }
};

// a thread's sums for one key -> the gene's table (false: no room)
template <int CAP, class Table>
__device__ __forceinline__ bool ga_flush(Table &T, int cap, int cap_log2, uint32_t key, int ng, int tot, unsigned long long sd, unsigned long long s1, unsigned long long s2)
{
	uint32_t slot = (key * 2654435761u) >> (32 - cap_log2);
	int probes = 0;
	for (; probes < cap; ++probes, slot = (slot + 1) & (cap - 1)) {
		const uint32_t old = atomicCAS(&T.key[slot], 0xffffffffu, key);
		if (old == 0xffffffffu || old == key) break;
	}
	if (probes == cap) return false;
	atomicAdd(&T.ng[slot], ng); atomicAdd(&T.tot[slot], tot);
	atomicAdd(&T.sd[slot], sd); atomicAdd(&T.s1[slot], s1); atomicAdd(&T.s2[slot], s2);
	return true;
}

// NT cooperating threads (one wave: 64, one workgroup: 256), tid in [0, NT).  Returns false when the table overflowed.
template <int NT, int CAP, int STAGE>
__device__ __forceinline__ bool gene_arcs_one(const GeneArcs &a, GeneTable<CAP, STAGE> &T, const int g, const int sid, const int tid, const int cap_log2)
{
	constexpr int HALO = STAGE >= 1024 ? 32 : 0; // the first kernel stages a whole gene at once
	const int z0 = a.zoff[g], z1 = a.zoff[g + 1], cap = 1 << cap_log2;
	for (int k = tid; k < cap; k += NT) T.key[k] = 0xffffffffu, T.ng[k] = 0, T.tot[k] = 0, T.sd[k] = 0, T.s1[k] = 0, T.s2[k] = 0;
	if (tid == 0) T.n_tot = 0, T.n_gen = 0, T.over = 0, T.m = 0, T.m0 = 0;
	int n_tot = 0, n_gen = 0; // this thread's walkable hits / genomes it counted (graph.c:125-126)
	uint32_t pk[2] = { 0xffffffffu, 0xffffffffu }; // per direction: the key whose run this thread is summing, and the sums
	int png[2] = { 0, 0 }, ptot[2] = { 0, 0 };
	unsigned long long psd[2] = { 0, 0 }, ps1[2] = { 0, 0 }, ps2[2] = { 0, 0 };
	for (int c0 = z0; c0 < z1; c0 += STAGE - 2 * HALO) {
		const int c1 = c0 + (STAGE - 2 * HALO) < z1 ? c0 + (STAGE - 2 * HALO) : z1;
		GeneWin<CAP, STAGE> W = { a, T, c0 - HALO > z0 ? c0 - HALO : z0, c1 + HALO < z1 ? c1 + HALO : z1 };
		if (NT == 64) wave_sync(); else __syncthreads(); // the previous window is done with (and the table is clear)
		for (int z = W.lo + tid; z < W.hi; z += NT) T.zy[z - W.lo] = a.zy[z], T.fx[z - W.lo] = a.hfk[z], T.bx[z - W.lo] = a.hbk[z]; // 12 bytes a hit (zy bit 31: singleton group)
		if (NT == 64) wave_sync(); else __syncthreads();
		for (int z = c0 + tid; z < c1; z += NT) {
			if (!hx_walk(T.bx[z - W.lo], a.tag)) continue;
			const int zyw = T.zy[z - W.lo], zy = zyw & 0x7fffffff, genome = zy >> 1, rev = zy & 1;
			// the hits of this gene in this genome: [gs, ge) around z -- z alone as a rule (static mark), then nothing to look at
			int gs = z, ge = z + 1;
			bool first = true; // first walkable hit of the group: it counts the genome (graph.c:125)
			if (zyw >= 0) {
				while (gs > z0 && (W.zy(gs - 1) >> 1) == genome) --gs;
				while (ge < z1 && (W.zy(ge) >> 1) == genome) ++ge;
				for (int q = gs; q < z; ++q) first = first && !hx_walk(W.bx(q), a.tag);
			}
			n_tot += 1, n_gen += first ? 1 : 0;
#pragma unroll
			for (int dir = 0; dir < 2; ++dir) {
				const uint32_t hx = dir ? T.bx[z - W.lo] : T.fx[z - W.lo];
				if (!hx_valid(hx, a.tag)) continue;
				const uint32_t key = (uint32_t)(dir ? !rev : rev) << HA_TAG_SHIFT | (hx & HA_NONE);
				// level 1 (graph.c:128-145): the first item of the group with this key collapses the group's items with the same key
				// (the two half-arcs of one hit never share a key: their orientation bits differ)
				bool leader = true;
				int n = 1;
				for (int q = gs; q < ge && leader && ge - gs > 1; ++q) {
					if (q == z) continue;
					const int rq = W.zy(q) & 1;
#pragma unroll
					for (int d2 = 0; d2 < 2; ++d2) {
						const uint32_t ox = d2 ? W.bx(q) : W.fx(q);
						if (!hx_valid(ox, a.tag) || ((uint32_t)(d2 ? !rq : rq) << HA_TAG_SHIFT | (ox & HA_NONE)) != key) continue;
						if (q < z) { leader = false; break; }
						++n;
					}
				}
				if (!leader) continue;
				const int4 h = dir ? a.hbp[z] : a.hfp[z]; // the payload: distance and the two scores (tried: asked for with the staged words, by the thread
				// that will evaluate the hit -- one trip to memory less per workgroup, but every hit's payloads instead of the walkable leaders': 48 -> 52 us)
				int m1 = h.y, m2 = h.z;
				unsigned long long sd = (unsigned long long)(long long)h.x;
				if (n > 1) // rare: the same adjacency twice in one genome
					for (int q = z + 1; q < ge; ++q) {
						const int rq = W.zy(q) & 1;
#pragma unroll
						for (int d2 = 0; d2 < 2; ++d2) {
							const uint32_t ox = d2 ? W.bx(q) : W.fx(q);
							if (!hx_valid(ox, a.tag) || ((uint32_t)(d2 ? !rq : rq) << HA_TAG_SHIFT | (ox & HA_NONE)) != key) continue;
							const int4 o = d2 ? a.hbp[q] : a.hfp[q];
							sd += (unsigned long long)(long long)o.x, m1 = m1 > o.y ? m1 : o.y, m2 = m2 > o.z ? m2 : o.z;
						}
					}
				m1 = m1 > 0 ? m1 : 0, m2 = m2 > 0 ? m2 : 0; // the reference's running maxima start at 0 (graph.c:133)
				const int dg = cvt_i32_x86((double)sd / n + .499); // graph.c:141: the sum is a uint64_t there, and so is its conversion
				// level 2 (graph.c:153-169): sums over the genomes, LDS table keyed by (orientation, target).  A lane's hits lie in
				// different genomes and mostly have the same neighbour: runs of one key are summed in registers and reach the table
				// as ONE set of atomics (a gene with 2 400 hits and ten neighbours would otherwise send 24 000 atomics to ten slots).
				if (key != pk[dir]) {
					if (pk[dir] != 0xffffffffu && !ga_flush<CAP>(T, cap, cap_log2, pk[dir], png[dir], ptot[dir], psd[dir], ps1[dir], ps2[dir])) T.over = 1;
					pk[dir] = key, png[dir] = 0, ptot[dir] = 0, psd[dir] = 0, ps1[dir] = 0, ps2[dir] = 0;
				}
				png[dir] += 1, ptot[dir] += n;
				psd[dir] += (unsigned long long)(long long)dg * (unsigned long long)n;
				ps1[dir] += (unsigned long long)(long long)m1, ps2[dir] += (unsigned long long)(long long)m2;
			}
		}
	}
	// (Tried in round 4: the lanes' last runs combined per wave before they reach the table -- one set of atomics per (wave, key) instead
	// of one per lane.  Slower: k_gene_arcs_big 273 -> 292 us at 12 M hits, configs[1] 5.58 -> 6.11 ms per pass.  The LDS serialises
	// same-address atomics cheaply; five wave reductions per key cost more than they save.)
#pragma unroll
	for (int dir = 0; dir < 2; ++dir)
		if (pk[dir] != 0xffffffffu && !ga_flush<CAP>(T, cap, cap_log2, pk[dir], png[dir], ptot[dir], psd[dir], ps1[dir], ps2[dir])) T.over = 1;
	{ // one LDS atomic per wave, not per hit
		const int wt = wave_sum(n_tot), wg = wave_sum(n_gen);
		if ((tid & 63) == 0) { if (NT == 64) T.n_tot = wt, T.n_gen = wg; else atomicAdd(&T.n_tot, wt), atomicAdd(&T.n_gen, wg); }
	}
	if (NT == 64) wave_sync(); else __syncthreads();
	if (T.over) return false;
	for (int k = tid; k < cap; k += NT)
		if (T.key[k] != 0xffffffffu) { const int at = atomicAdd(&T.m, 1); T.dense[at] = (uint16_t)k; if (!(T.key[k] >> HA_TAG_SHIFT)) atomicAdd(&T.m0, 1); }
	if (NT == 64) wave_sync(); else __syncthreads();
	const int m = T.m;
	if (tid == 0) {
		const int b = 2 * z0, m0 = T.m0; // the gene's own stretch of the table: it has at most two half-arcs per hit (no allocation, no atomic)
		T.base = b;
		a.seg_cnt[sid] = T.n_gen, a.seg_cnt[a.S + sid] = T.n_tot, a.seg_gid[sid] = g;
		a.gmeta[sid] = make_int4(b, m0, m - m0, 0);
		a.vs[2 * sid] = b, a.ve[2 * sid] = b + m0, a.vs[2 * sid + 1] = b + m0, a.ve[2 * sid + 1] = b + m;
		a.deg[2 * sid] = m0, a.deg[2 * sid + 1] = m - m0;
		a.vwk[2 * sid] = 0, a.vwk[2 * sid + 1] = 0;
		if (a.h_round) {
			a.h_round[sid] = T.n_gen, a.h_round[a.S + sid] = T.n_tot; // pinned host memory, plain stores (see sys_store in k_common.hpp)
			a.h_round[2 * a.S + 2 * sid] = m0, a.h_round[2 * a.S + 2 * sid + 1] = m - m0;
		}
	}
	if (NT == 64) wave_sync(); else __syncthreads();
	for (int e = tid; e < m; e += NT) { // rank among the gene's entries = place in its stretch (keys are distinct)
		const int k = T.dense[e];
		const uint32_t key = T.key[k];
		int r = 0;
		for (int j = 0; j < m; ++j) r += T.key[T.dense[j]] < key;
		pga_arc_part_t o;
		const uint32_t t = key & HA_NONE;
		o.x = (uint64_t)((uint32_t)sid << 1 | (key >> HA_TAG_SHIFT)) << 32 | (uint32_t)((uint32_t)a.g2s[t >> 1] << 1 | (t & 1u));
		o.n_genome = T.ng[k], o.tot_cnt = T.tot[k], o.sum_dist = T.sd[k], o.sum_s1 = (int64_t)T.s1[k], o.sum_s2 = (int64_t)T.s2[k];
		const int at = T.base + r;
		a.stage[at] = o;
		a.ax[at] = o.x;
		a.s1[at] = (int32_t)((double)o.sum_s1 / o.n_genome + .499); // graph.c:171
		a.agid[at] = (int32_t)(t >> 1);
		a.aw[at] = 0;
	}
	return true;
}

// one workgroup per gene: at the usual sizes (a few hundred hits of a gene in the shard) its threads see every hit in one or two goes
// (128 threads: a workgroup is a chain of dependent loads whatever its width, and a CU holds 14 of these against 8 of 256 threads -- a shard of
// 20 000 genes x 137 hits 218 -> 170 us, configs[1] 49.6 -> 47.9; one wave a gene, which the CU holds no more of -- the LDS tables -- 62)
constexpr int GA_WAVE_NT = 128;
// (a template over its three sizes so that other shapes can be measured side by side: PANGENE_GA_WAVE picks one, arc_round_genes)
template <int NT, int CAP, int HITS, int CAP_LOG2>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(7))) void k_gene_arcs_wave_t(GeneArcs a)
{
	__shared__ GeneTable<CAP, HITS> T;
	if (gate_closed(a.gate)) return;
	const int g = blockIdx.x, tid = threadIdx.x;
	const int sid = a.g2s[g], z0 = a.zoff[g], z1 = a.zoff[g + 1];
	if (sid < 0) { // not a vertex: none of its hits may be walkable (graph.c:111)
		for (int z = z0 + tid; z < z1; z += NT)
			if (hx_walk(a.hbk[z], a.tag)) atomicAdd((unsigned long long *)&a.dcnt[3], 1ull), a.dcnt[11] = 1; // ([11]: sticky, for rounds nobody looks at one by one: pga_branch_loop)
		return;
	}
	const int cl = a.cap_log2 < CAP_LOG2 ? a.cap_log2 : CAP_LOG2;
	const bool done = z1 - z0 <= HITS && gene_arcs_one<NT, CAP, HITS>(a, T, g, sid, tid, cl);
	if (tid == 0 && !done) a.big_list[atomicAdd(&a.big_ctl[0], 1)] = g; // many hits, or many neighbours: the second kernel takes it
}

// (512 threads a gene: the LDS tables allow three of these workgroups on a CU whatever their width -- 12 waves of 256 threads, 24 of 512; 328 -> 258 us
// at 12.1 M hits, where every gene comes here; 1 024 threads: two workgroups a CU, 305 us; a shard that sends nothing here pays 1-4 us more for the empty pass)
constexpr int GA_BIG_NT = 512;
__global__ __launch_bounds__(GA_BIG_NT) __attribute__((amdgpu_waves_per_eu(6))) void k_gene_arcs_big(GeneArcs a)
{
	__shared__ GeneTable<GA_CAP, GA_BIG_STAGE> T;
	if (gate_closed(a.gate)) return;
	if (a.tag_out && blockIdx.x == 0 && threadIdx.x == 0) *a.tag_out = (int32_t)a.tag;
	// The genes the wave kernel left, handed out one at a time (round 6).  Rounds 3-5 gave workgroup b the genes b, b + grid, ... and skipped the small
	// ones by a flag: on the 12.1 M-hit shard, where 2 327 of 5 000 genes come here at ~70 us each and 768 workgroups fit the chip, the luck of
	// that draw decided the launch's length.
	// (Workgroup b starts with item b; only who finishes one asks the counter for the next -- a returning atomic on ONE word costs ~50 ns each at
	// the memory side: 2 048 workgroups asking at once made an otherwise empty launch take 115 us.)
	__shared__ int s_next;
	const int n_big = a.big_ctl[0];
	for (int at = blockIdx.x; at < n_big; ) {
		const int g = a.big_list[at];
		const int sid = a.g2s[g];
		if (!gene_arcs_one<GA_BIG_NT, GA_CAP, GA_BIG_STAGE>(a, T, g, sid, threadIdx.x, a.cap_log2) && threadIdx.x == 0) // a hub gene: this round is redone on the sort path
			atomicAdd((unsigned long long *)&a.dcnt[9], 1ull), a.dcnt[11] = 1, a.gmeta[sid] = make_int4(0, 0, 0, 0); // (the whole round is repeated: nothing else to leave behind)
		if (threadIdx.x == 0) s_next = (int)gridDim.x + atomicAdd(&a.big_ctl[1], 1);
		__syncthreads();
		at = s_next;
		__syncthreads(); // (s_next is written again one item on)
	}
}

// ------------------------------------------------------------------------------------------------
// (C) place every gene's arcs (segment order = gene order) and derive what the branch steps read
// ------------------------------------------------------------------------------------------------
struct InGmeta { const int4 *gm; __device__ __forceinline__ I32 operator()(int64_t i) const { const int4 m = gm[i]; return I32{m.y + m.z}; } };

// the table as ONE array sorted by x: one wave per segment copies the gene's stretch to its place (off[] = exclusive scan of the
// segments' arc counts); the last segment also leaves the table size in dcnt[10] and mails the counters
__global__ __launch_bounds__(BLOCK) void k_arc_compact(const int4 *gmeta, const int32_t *off, int S, const pga_arc_part_t *stage, pga_arc_part_t *arcs, int64_t *dcnt, int64_t *host_box)
{
	const int lane = threadIdx.x & 63;
	for (int sid = blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6); sid < S; sid += gridDim.x * (BLOCK / WAVE)) {
		const int4 m = gmeta[sid];
		const int o = off[sid], n = m.y + m.z;
		for (int i = lane; i < n; i += WAVE) arcs[o + i] = stage[m.x + i];
		if (lane == 0 && sid == S - 1) {
			dcnt[10] = o + n;
			for (int t = 0; t < 16; ++t) sys_store(&host_box[t], dcnt[t]);
		}
	}
}

// the same into the rank's slot of a sharded round's all-gather (k_arcs.hpp, XS_HDR): the table, the segment counters and the table's size
// (gate: a queued round of a sharded run whose own arc round found nothing to do leaves its slot as the round before left it)
// Header word 3 of the slot: did this rank raise a hit's weak_br in this round (stamp[1] == round)?  The fixed point of the queued rounds (dev_prims.hpp:
// Gate) is a property of ALL ranks' hits: k_xs_sum_rank ORs the ranks' words and stamps the round for everybody.  Written whether the gate is open or not.
__global__ __launch_bounds__(BLOCK) void k_xs_compact(const int4 *gmeta, const int32_t *off, int S, const pga_arc_part_t *stage, const int32_t *seg_cnt, int32_t *slot, int64_t arc_cap, const int64_t *dcnt, Gate gate = Gate{nullptr, 0},
                                                        const int32_t *stamp = nullptr /* Gate::w, or NULL */, int round = 0)
{
	if (blockIdx.x == 0 && threadIdx.x == 0) slot[3] = (stamp && stamp[1] == round) ? 1 : 0;
	if (gate_closed(gate)) return;
	const int lane = threadIdx.x & 63;
	int32_t *segc = slot + XS_HDR;
	pga_arc_part_t *arcs = (pga_arc_part_t *)(slot + XS_HDR + xs_seg_words(S));
	for (int sid = blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6); sid < S; sid += gridDim.x * (BLOCK / WAVE)) {
		const int4 m = gmeta[sid];
		const int o = off[sid], n = m.y + m.z;
		if ((int64_t)o + n <= arc_cap)
			for (int i = lane; i < n; i += WAVE) arcs[o + i] = stage[m.x + i];
		if (lane == 0) {
			segc[sid] = seg_cnt[sid], segc[S + sid] = seg_cnt[S + sid];
			if (sid == S - 1) { slot[0] = o + n, slot[1] = dcnt[9] != 0, slot[2] = dcnt[3] != 0; for (int t = 4; t < XS_HDR; ++t) slot[t] = 0; } // ([3]: this kernel's first thread, above) // [1] a hub gene overflowed its LDS table, [2] an invariant was violated
		}
	}
}

// ------------------------------------------------------------------------------------------------
// pg_mark_branch_flt_hit (branch.c:108-145): a hit is marked by the weak arcs among its own two half-arcs
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_mark_hits_z(const int32_t *zx, const int32_t *zy, const int32_t *zg, const uint32_t *hfk, const uint32_t *hbk, uint32_t tag, int n, const int32_t *g2s,
                                                        const uint64_t *ax, const uint8_t *aw, const int32_t *vs, const int32_t *ve, const uint8_t *vwk,
                                                        uint32_t *flags, int64_t *cnt, int then_filter, Gate gate, int32_t *stamp /* Gate::w of the loop, or NULL */, int round)
{
	int z = blockIdx.x * BLOCK + threadIdx.x;
	bool marked = false;
	if (gate_closed(gate)) return; // (uniform; a round that counts for the log never comes with a gate)
	if (z < n) {
		const uint32_t kb = hbk[z], kf = hfk[z]; // 16 bytes a hit, all independent loads
		const int y = zy[z], g = zg[z];
		if (hx_walk(kb, tag)) {
			const int sid = g2s[g], rev = y & 1;
			int nw = 0;
#pragma unroll
			for (int dir = 0; dir < 2; ++dir) {
				const uint32_t h = dir ? kb : kf;
				if (!hx_valid(h, tag) || sid < 0) continue;
				const uint32_t u = (uint32_t)sid << 1 | (uint32_t)(dir ? !rev : rev), t = h & HA_NONE;
				if (!vwk[u]) continue; // the vertex has no weak out-arc
				const uint32_t w = (uint32_t)g2s[t >> 1] << 1 | (t & 1u);
				const int e = arc_weak_v(ax, aw, vs, ve, u, w); // dir 0: arc v -> w marks the earlier hit (branch.c:128-130); dir 1: w^1 -> v^1 marks the later one (131-133)
				nw = nw > e ? nw : e;
			}
			if (nw) { // rare: only now is the hit's flag word touched
				const int x = zx[z];
				const uint32_t f = flags[x];
				// (then_filter: PG_SET_FILTER(weak_br == 2), graph.c:309 -- only a hit marked here can newly have weak_br == 2)
				if (nw > (int)((f & PGA_F_WEAK_MASK) >> PGA_F_WEAK_SHIFT)) {
					flags[x] = (f & ~PGA_F_WEAK_MASK) | (uint32_t)nw << PGA_F_WEAK_SHIFT | ((then_filter && nw == 2) ? PGA_F_FLT : 0u);
					if (stamp) stamp[1] = round; // a hit's weak_br went up: the state of the loop changed in this round
				}
				marked = true;
			}
		}
		if (cnt && !marked) marked = (flags[zx[z]] & PGA_F_WEAK_MASK) != 0; // log-only counter (branch.c:137-139): every hit with weak_br != 0
	}
	if (cnt) {
		const unsigned long long mk = __ballot(marked);
		if (mk && (threadIdx.x & 63) == (unsigned)__ffsll((long long)mk) - 1) atomicAdd((unsigned long long *)cnt, (unsigned long long)__popcll(mk));
	}
}
