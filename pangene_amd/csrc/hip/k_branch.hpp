// k_branch.hpp -- branch.c on the device.
// Included by pga_backend.hip (one translation unit); uses the context types, BLOCK / WAVE and dev_prims.hpp from there.
#pragma once

// ------------------------------------------------------------------------------------------------
// branch.c on device: pg_gen_rep_pos (6-29), pg_n_local (31-46), pg_mark_branch_flt_hit (108-145)
// ------------------------------------------------------------------------------------------------
// pg_gen_rep_pos (branch.c:6-29) as ONE scan over the X order: the input is the walkable mark of a hit (computed on the fly), the
// exclusive sum is its rank among the walkable hits (rx), and the output step also records the hit as its gene's representative in
// its genome -- the last walkable hit of a gene in array order wins (branch.c:22-23 overwrite), hence the atomicMax of h + 1.
struct InWalkX { const uint32_t *flags; __device__ __forceinline__ I32 operator()(int64_t i) const { return I32{(flags[i] & (PGA_F_FLT | PGA_F_SHADOW)) ? 0 : 1}; } };
struct OutRank {
	int32_t *rx; const uint32_t *flags;
	__device__ __forceinline__ void operator()(int64_t i, I32, I32 ex) const
	{
		rx[i] = ex.v | ((flags[i] & F_CSTIE) ? (int32_t)0x80000000 : 0); // bit 31: member of a cs tie group (k_rep_fill looks closer)
	}
};

// The same ranks per GENOME in one launch: k_rep_fill only ever uses a hit's rank inside its genome (rx[h] - rx[first hit of the genome]), so no
// carry has to cross a genome and the scan needs no second launch -- one workgroup a genome, 1024 hits a step, the wave ranks by ballots.
// (rx[first hit of a genome] = 0 here.)  Used while no genome is long enough for its workgroup to become the launch's tail (pga_rep_pos).
constexpr int RK_T = 1024;
__device__ __forceinline__ void rank_genome_body(const uint32_t *flags, const int32_t *goff, int32_t *rx, const int g, int (*wtot)[RK_T / WAVE])
{
	const int h0 = goff[g], h1 = goff[g + 1], tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
	int carry = 0, par = 0;
	for (int base = h0; base < h1; base += RK_T, par ^= 1) { // (two sets of wave totals used in turn: one barrier a step)
		const int i = base + tid;
		const uint32_t f = i < h1 ? flags[i] : (uint32_t)PGA_F_FLT;
		const unsigned long long m = __ballot(!(f & (PGA_F_FLT | PGA_F_SHADOW)));
		if (lane == 0) wtot[par][w] = __popcll(m);
		__syncthreads();
		int pre = carry, tot = 0;
#pragma unroll
		for (int k = 0; k < RK_T / WAVE; ++k) { const int t = wtot[par][k]; tot += t; if (k < w) pre += t; }
		if (i < h1) rx[i] = (pre + __popcll(m & lt)) | ((f & F_CSTIE) ? (int32_t)0x80000000 : 0);
		carry += tot;
	}
}
__global__ __launch_bounds__(RK_T) void k_rank_genome(const uint32_t *flags, const int32_t *goff, int32_t *rx, Gate gate)
{
	__shared__ int wtot[2][RK_T / WAVE];
	if (gate_closed(gate)) return;
	rank_genome_body(flags, goff, rx, blockIdx.x, wtot);
}

// Position record of (gene, genome): {contig, rank among the walkable hits of the genome, cm} of the gene's LAST walkable hit in
// the genome's array order (branch.c:22-23 overwrite).  COMPACT (every genome has < 4096 contigs and < 2^20 hits, decided once in
// create): 8 bytes {cm, local contig << 20 | rank}, half the L2 traffic of pg_n_local, which reads two records per (pair, genome);
// otherwise 16 bytes {global contig, rank, cm, interval}.  Absent: -1.
// WIDE (a genome of the shard came with virtual contigs, pga_genome_block_t: coordinates of 64 bits): 16 bytes {contig = the segment id of
// the contig's first piece, rank | bit 31 "has an interval", low and high word of the true cm}, the interval in iv[] as in the compact form.
constexpr int RP_FULL = 0, RP_COMPACT = 1, RP_WIDE = 2;
//
// Filled from the gene-major index: the hits of (gene, genome) are adjacent there, in array order, so the last hit of each
// group finds the group's last walkable hit (walkable = its half-arc record carries the round's tag) and also writes the "absent"
// records of the genomes up to the next group -- every record is written exactly once, nothing is cleared, nothing is atomic.
//
// Tie order (hazard H2b).  The reference's unstable sort may permute the hits sharing (contig, cs); the walkable
// ones among them receive consecutive values of the counter r (branch.c:14,22-24) in whatever order they end up, so the r of
// a representative inside such a group is only known up to [r - nb, r + na] (nb / na = walkable members before / after it in
// the canonical order).  Representatives with nb + na > 0 carry bit 31 of their cm word and the interval nb << 16 | na (side
// table iv[] in the compact form); k_n_local raises the hazard where it matters.  Two walkable hits of ONE gene in a group
// (-S: opposite strands) make the choice of the representative itself order-dependent (branch.c:22-23): hazard at once.
// The (contig, cs) tie groups of the cs order are static -- a cs override only permutes hits INSIDE a group --, so where a member's group begins and
// ends is looked up once per pass here instead of in every round by every representative (k_rep_fill walked the 16-byte records to either side: on
// isoform-rich data nearly every live hit shares its start with five filtered isoforms -- ~300 us a round at the 21.9 M hits of configs[4]).
__global__ __launch_bounds__(BLOCK) void k_tie_bounds(const int4 *A, const uint32_t *flags, int n, int2 *tg)
{
	const int x = blockIdx.x * BLOCK + threadIdx.x;
	if (x >= n || !(flags[x] & F_CSTIE)) return;
	const int4 ah = A[x]; // {cs, seg, ce, pm}
	int ta = x, tb = x + 1;
	while (ta > 0) { const int4 ap = A[ta - 1]; if (ap.y != ah.y || ap.x != ah.x) break; --ta; }
	while (tb < n) { const int4 ap = A[tb]; if (ap.y != ah.y || ap.x != ah.x) break; ++tb; }
	tg[x] = make_int2(ta, tb);
}

struct RepFill {
	const int2 *tg; // [N] k_tie_bounds (members of a tie group only)
	int64_t n_ent; int GL, Q, N /* hits of the shard (X positions) */, NZ /* entries of the gene-major index: N, or the members of the live lists */; const int32_t *zx, *zy, *zg; const int2 *zst; const int32_t *zoff; const uint32_t *hbk; uint32_t tag;
	const int4 *A; const int32_t *gid; const uint32_t *flags; const int32_t *rx, *goff, *ctg_base;
	void *rp_out; int32_t *iv; int64_t *dcnt; int32_t *hz_list;
	const int32_t *vfirst; const int64_t *vbase; // [contig segments] of the shard (RP_WIDE only)
	Gate gate;
};

template <int FORM>
__device__ __forceinline__ void rep_absent(const RepFill &a, int64_t e0, int n)
{
	for (int k = 0; k < n; ++k) {
		if (FORM == RP_COMPACT) ((int2 *)a.rp_out)[e0 + k] = make_int2(0, -1); else ((int4 *)a.rp_out)[e0 + k] = make_int4(-1, 0, 0, 0);
	}
}
// With live lists (CLEARED): every record "absent" first, as one coalesced fill (round 6).  Rounds 3-5 had the last hit of every (gene, genome) group write the absent records of
// the genomes up to the next group, one thread in a loop -- nothing was cleared, nothing written twice; but once the index holds the live hits
// only, a gene that lost its vertex has no entry at all and ONE thread wrote the records of all its genomes: 1 250 scattered stores in a row
// for each of 2 673 genes of the 12.1 M-hit shard, the kernel 0.19 -> ~1 ms.  The fill is 50 MB there: ~15 us.
template <int FORM>
__global__ __launch_bounds__(BLOCK) void k_rep_clear(void *rp_out, int64_t n_ent, Gate gate)
{
	if (gate_closed(gate)) return;
	const int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (e >= n_ent) return;
	if (FORM == RP_COMPACT) ((int2 *)rp_out)[e] = make_int2(0, -1); else ((int4 *)rp_out)[e] = make_int4(-1, 0, 0, 0);
}

// CLEARED: k_rep_clear ran first (live lists); otherwise this kernel writes every record itself, the absent ones too (one launch a round less)
template <int FORM, bool CLEARED>
__device__ __forceinline__ void rep_fill_body(const RepFill &a, const int t)
{
	if (!CLEARED && t < a.Q && a.zoff[t] == a.zoff[t + 1]) rep_absent<FORM>(a, (int64_t)t * a.GL, a.GL); // a gene without hits in this shard
	if (t >= a.NZ) return;
	const int z = t;
	// the loads are issued in as few dependent rounds as possible, from 4-byte planes in gene-major order (the kernel is bound by
	// latency and sectors, not by arithmetic): round 1 = genome / gene of z and of its two neighbours, z's walkable mark
	const int g = a.zg[z], y = a.zy[z] & 0x7fffffff, j = y >> 1;
	const bool has_n = z + 1 < a.NZ;
	const int gn = has_n ? a.zg[z + 1] : -1, yn = has_n ? (a.zy[z + 1] & 0x7fffffff) : 0;
	const uint32_t kb = a.hbk[z];
	if (gn == g && (yn >> 1) == j) return; // not the last hit of its (gene, genome) group
	// round 2: everything that hangs on the gene, the genome or the hit itself
	const int z0 = a.zoff[g], gj = a.goff[j], cb = a.ctg_base[j], xz = a.zx[z];
	const int2 st_z = a.zst[z]; // {cm, contig segment}
	const int64_t e = (int64_t)g * a.GL + j;
	// (CLEARED: the records of the genomes without a hit of this gene, and of the groups without a walkable hit, are "absent" already)
	if (!CLEARED) { // the genomes without a hit of this gene: before the first group, and between this group and the next
		int gs = z;
		while (gs > z0 && ((a.zy[gs - 1] & 0x7fffffff) >> 1) == j) --gs;
		if (gs == z0 && j > 0) rep_absent<FORM>(a, (int64_t)g * a.GL, j);
		const int jn = gn == g ? (yn >> 1) : a.GL;
		if (jn > j + 1) rep_absent<FORM>(a, e + 1, jn - j - 1);
	}
	int q = z;
	if (!hx_walk(kb, a.tag)) { // the group's last walkable hit
		q = z - 1;
		while (q >= z0 && ((a.zy[q] & 0x7fffffff) >> 1) == j && !hx_walk(a.hbk[q], a.tag)) --q;
		if (q < z0 || ((a.zy[q] & 0x7fffffff) >> 1) != j) { if (!CLEARED) rep_absent<FORM>(a, e, 1); return; }
	}
	const int h = q == z ? xz : a.zx[q];
	const int2 st = q == z ? st_z : a.zst[q];
	// round 3: the one gather into cs order -- the hit's rank among the walkable hits (and the rank at the genome's start)
	const int rxh = a.rx[h], r = (rxh & 0x7fffffff) - (a.rx[gj] & 0x7fffffff);
	int ivl = 0;
	if (rxh < 0) { // member of a static tie group [ta, tb): its walkable members on either side, from the walkable ranks at the group's ends
		const int hi = a.goff[j + 1];
		const int2 tgb = a.tg[h];
		const int ta = tgb.x, tb = tgb.y; // (a contig never crosses a genome: gj <= ta, tb <= hi)
		const int re = tb < hi ? (a.rx[tb] & 0x7fffffff) : (a.rx[tb - 1] & 0x7fffffff) + ((a.flags[tb - 1] & (PGA_F_FLT | PGA_F_SHADOW)) ? 0 : 1); // (tb < hi: the ranks may be per genome, k_rank_genome)
		const int nb = (rxh & 0x7fffffff) - (a.rx[ta] & 0x7fffffff), na = re - (rxh & 0x7fffffff) - 1;
		if (nb + na > 0) { // rare: walkable hits do share this start
			bool same_gene = false; // two walkable hits of ONE gene in the group: the representative itself depends on the tie order
			for (int p = ta; p < tb; ++p) same_gene = same_gene || (p != h && a.gid[p] == g && !(a.flags[p] & (PGA_F_FLT | PGA_F_SHADOW)));
			if (same_gene || nb > 0xffff || na > 0xffff) {
				atomicAdd((unsigned long long *)&a.dcnt[6], 1ull);
				hz_note(&a.dcnt[14], a.hz_list, st.y);
			} else ivl = nb << 16 | na;
		}
	}
	const int cmw = st.x | (ivl ? (int)0x80000000 : 0);
	if (FORM == RP_COMPACT) {
		((int2 *)a.rp_out)[e] = make_int2(cmw, (st.y - cb) << 20 | r);
		if (ivl) a.iv[e] = ivl;
	} else if (FORM == RP_WIDE) {
		const long long cm64 = (long long)st.x + a.vbase[st.y]; // the piece's base back on (branch.c:23 keeps 64 bits)
		((int4 *)a.rp_out)[e] = make_int4(a.vfirst[st.y], r | (ivl ? (int)0x80000000 : 0), (int)(unsigned)(unsigned long long)cm64, (int)((unsigned long long)cm64 >> 32));
		if (ivl) a.iv[e] = ivl;
	} else ((int4 *)a.rp_out)[e] = make_int4(st.y, r, cmw, ivl);
}
template <int FORM, bool CLEARED>
__global__ __launch_bounds__(BLOCK) void k_rep_fill(RepFill a)
{
	if (gate_closed(a.gate)) return;
	rep_fill_body<FORM, CLEARED>(a, blockIdx.x * BLOCK + threadIdx.x);
}

// hazard H2b inside pg_n_local: the pair's distance test failed, so the count test |r1 - r2| <= local_count decides, and at
// least one r is only known up to an interval: is the answer the same over the whole interval?  (rare path)
struct NLocalHz { const int32_t *iv; const int32_t *ctg_base; int64_t *dcnt; int32_t *list; };

__device__ __noinline__ bool nl_hazard(const NLocalHz &z, int cc, int iv1, int iv2, int seg1, int seg2, int local_count) // true: the same for every tie order
{
	const int lo = cc - (iv1 >> 16) - (iv2 & 0xffff), hi = cc + (iv1 & 0xffff) + (iv2 >> 16);
	const bool all_in = lo >= -local_count && hi <= local_count, all_out = hi < -local_count || lo > local_count;
	if (all_in || all_out) return true;
	atomicAdd((unsigned long long *)&z.dcnt[6], 1ull);
	if (iv1) hz_note(&z.dcnt[14], z.list, seg1);
	if (iv2) hz_note(&z.dcnt[14], z.list, seg2);
	return false;
}

// pg_n_local (branch.c:31-46).  Its callers only ever ask whether the count is zero (branch.c:76 and 86), so the search over
// the genomes stops at the first genome in which the pair is local FOR CERTAIN (a hit whose local_count test depends on the tie
// order -- hazard H2b -- does not stop it); cnt[k] = the hits seen until then: > 0 iff pg_n_local > 0, and sums over shards keep
// that property.  Sixteen lanes work on one pair (sixteen genomes per step, one coalesced 128-byte read per record table), four
// pairs per wave; as a rule the first step settles a pair, so the cost no longer grows with the number of genomes.
// (Lanes a pair, measured in round 6 -- 8 / 16 / 32 / 64: configs[1], 100 genomes, 4.62 / 4.58 / 4.73 / 5.00 ms a pass; the shard of 1 250 genomes
// 24.6 / 23.1 / 22.4 / 22.3: where a gene is absent from most genomes the first sixteen settle little.  The host picks by the number of genomes.
// Tried the other way too, four pairs a group in flight with their loads batched: k_n_local 31 -> 43 us at configs[1]; more waves beat longer ones.)
constexpr int nl_lanes_for(int GL) { return GL <= 256 ? 16 : GL <= 640 ? 32 : 64; }

template <int FORM, int NL_LANES>
__global__ __launch_bounds__(BLOCK) void k_n_local(const int32_t *pairs, int64_t n_cap, const int64_t *np_dev, int GL, const void *rp_in,
                                                     int local_dist, int local_count, int frag_mode, int32_t *cnt, NLocalHz hz, Gate gate)
{
	constexpr int NL_PAIRS = WAVE / NL_LANES;
	constexpr unsigned long long NL_MASK = NL_LANES == 64 ? ~0ull : ((1ull << (NL_LANES & 63)) - 1);
	if (gate_closed(gate)) return;
	const int lane = threadIdx.x & 63, grp = lane / NL_LANES, sub = lane & (NL_LANES - 1);
	int64_t n_pair = np_dev ? *np_dev : n_cap; // the count may still be on its way to the host: it is read here
	if (n_pair > n_cap) return; // more pairs than the list holds: some stretches of it were never written, and the host repeats the step with room
	const int64_t stride = (int64_t)gridDim.x * (BLOCK / WAVE) * NL_PAIRS;
	for (int64_t k0 = ((int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6)) * NL_PAIRS; k0 < n_pair; k0 += stride) {
		const int64_t k = k0 + grp;
		const bool have = k < n_pair;
		const int64_t kk = have ? k : n_pair - 1;
		const int64_t g1 = (int64_t)pairs[2 * kk] * GL, g2 = (int64_t)pairs[2 * kk + 1] * GL;
		int c = 0;
		bool open = have; // still searching (uniform over the sixteen lanes of the pair)
		for (int j0 = 0; j0 < GL; j0 += NL_LANES) {
			if (__ballot(open) == 0) break;
			const int j = j0 + sub;
			const bool in = open && j < GL;
			const int jj = j < GL ? j : 0;
			bool hit = false, sure = false;
			if (FORM == RP_WIDE) {
				const int4 a = ((const int4 *)rp_in)[g1 + jj], b = ((const int4 *)rp_in)[g2 + jj]; // {contig, rank | bit 31, cm low, cm high}
				const long long d = (long long)((unsigned long long)(unsigned)a.w << 32 | (unsigned)a.z) - (long long)((unsigned long long)(unsigned)b.w << 32 | (unsigned)b.z); // branch.c:40
				const int cc = (a.y & 0x7fffffff) - (b.y & 0x7fffffff);
				const bool cmp = in && a.x >= 0 && b.x >= 0 && (frag_mode || a.x == b.x);
				const bool near = d >= -(long long)local_dist && d <= (long long)local_dist;
				hit = cmp && (near || (cc >= -local_count && cc <= local_count));
				sure = hit;
				if (cmp && !near && (a.y | b.y) < 0) sure = nl_hazard(hz, cc, a.y < 0 ? hz.iv[g1 + jj] : 0, b.y < 0 ? hz.iv[g2 + jj] : 0, a.x, b.x, local_count) && hit;
			} else if (FORM == RP_COMPACT) {
				const int2 a = ((const int2 *)rp_in)[g1 + jj], b = ((const int2 *)rp_in)[g2 + jj];
				const int d = (a.x & 0x7fffffff) - (b.x & 0x7fffffff); // cm < 2^31: the difference fits
				const int cc = (a.y & 0xfffff) - (b.y & 0xfffff);
				const bool cmp = in && (a.y | b.y) >= 0 && (frag_mode || ((a.y ^ b.y) >> 20) == 0);
				const bool near = d >= -local_dist && d <= local_dist;
				hit = cmp && (near || (cc >= -local_count && cc <= local_count));
				sure = hit;
				if (cmp && !near && (a.x | b.x) < 0) // rare: an r of the pair depends on the tie order (H2b)
					sure = nl_hazard(hz, cc, a.x < 0 ? hz.iv[g1 + jj] : 0, b.x < 0 ? hz.iv[g2 + jj] : 0, hz.ctg_base[jj] + (a.y >> 20), hz.ctg_base[jj] + (b.y >> 20), local_count) && hit;
			} else {
				const int4 a = ((const int4 *)rp_in)[g1 + jj], b = ((const int4 *)rp_in)[g2 + jj];
				const int d = (a.z & 0x7fffffff) - (b.z & 0x7fffffff);
				const int cc = a.y - b.y;
				const bool cmp = in && a.x >= 0 && b.x >= 0 && (frag_mode || a.x == b.x);
				const bool near = d >= -local_dist && d <= local_dist;
				hit = cmp && (near || (cc >= -local_count && cc <= local_count));
				sure = hit;
				if (cmp && !near && (a.z | b.z) < 0) sure = nl_hazard(hz, cc, a.w, b.w, a.x, b.x, local_count) && hit;
			}
			const unsigned long long mh = __ballot(hit), ms = __ballot(sure);
			c += __popcll((mh >> (grp * NL_LANES)) & NL_MASK);
			if ((ms >> (grp * NL_LANES)) & NL_MASK) open = false;
		}
		if (have && sub == 0) cnt[k] = c;
	}
}

// ------------------------------------------------------------------------------------------------
// pg_mark_branch_flt_arc (branch.c:48-106) on the arc table: one thread per oriented vertex
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_br_prep(const uint64_t *ax, int64_t n_arc, const int32_t *seg_gid, int32_t *agid, int32_t *vs, int32_t *ve)
{
	int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n_arc) return;
	const uint64_t x = ax[i];
	const uint32_t v = (uint32_t)(x >> 32);
	agid[i] = seg_gid[(uint32_t)x >> 1];
	if (i == 0 || (uint32_t)(ax[i - 1] >> 32) != v) vs[v] = (int32_t)i;
	if (i == n_arc - 1 || (uint32_t)(ax[i + 1] >> 32) != v) ve[v] = (int32_t)i + 1;
}

// the round's arc table -> what branch marking reads (see pga_arc_set_current)
__global__ __launch_bounds__(BLOCK) void k_seg_gid(const int32_t *g2s, int Q, int n_seg, int32_t *seg_gid)
{
	int g = blockIdx.x * BLOCK + threadIdx.x;
	if (g < Q) { int s = g2s[g]; if (s >= 0 && s < n_seg) seg_gid[s] = g; }
}

__global__ __launch_bounds__(BLOCK) void k_cur_prep(const pga_arc_part_t *arcs, int64_t n_arc, const int32_t *seg_gid, uint64_t *ax, int32_t *s1, int32_t *agid,
                                                      int32_t *vs, int32_t *ve)
{
	int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n_arc) return;
	const pga_arc_part_t a = arcs[i];
	const uint32_t v = (uint32_t)(a.x >> 32);
	ax[i] = a.x;
	s1[i] = (int32_t)((double)a.sum_s1 / a.n_genome + .499); // graph.c:171
	agid[i] = seg_gid[(uint32_t)a.x >> 1];
	if (i == 0 || (uint32_t)(arcs[i - 1].x >> 32) != v) vs[v] = (int32_t)i;
	if (i == n_arc - 1 || (uint32_t)(arcs[i + 1].x >> 32) != v) ve[v] = (int32_t)i + 1;
}

// pga_arc_set_current in one launch, the table's size in device memory (sharded rounds): per arc x / rounded s1 / target gene / weak_br = 0;
// the first arc of a vertex finds the end of the vertex's run and writes its range, degree and "no weak arc", and the same (empty) for the
// vertices without arcs before it; the vertices behind the last arc are left to the threads beyond the table (the grid has n_vtx of them).
__global__ __launch_bounds__(BLOCK) void k_curx_table(const pga_arc_part_t *arcs, const int64_t *n_dev, int64_t cap, const int32_t *seg_gid, int n_vtx, uint64_t *ax, int32_t *s1, int32_t *agid, uint8_t *aw,
                                                        int32_t *vs, int32_t *ve, int32_t *deg, uint8_t *vwk)
{
	const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x, n_arc = *n_dev;
	if (i >= n_arc) { // a spare thread: one vertex behind the last one that has arcs
		if (i < cap) return;
		const int64_t u = (n_arc ? (int64_t)(uint32_t)(arcs[n_arc - 1].x >> 32) + 1 : 0) + (i - cap);
		if (u < n_vtx) vs[u] = 0, ve[u] = 0, deg[u] = 0, vwk[u] = 0;
		return;
	}
	const pga_arc_part_t a = arcs[i];
	const uint32_t v = (uint32_t)(a.x >> 32);
	ax[i] = a.x;
	s1[i] = (int32_t)((double)a.sum_s1 / a.n_genome + .499); // graph.c:171
	agid[i] = seg_gid[(uint32_t)a.x >> 1];
	aw[i] = 0;
	const int64_t pv = i ? (int64_t)(uint32_t)(arcs[i - 1].x >> 32) : -1;
	if (pv == (int64_t)v) return;
	int64_t e = i + 1;
	while (e < n_arc && (uint32_t)(arcs[e].x >> 32) == v) ++e;
	vs[v] = (int32_t)i, ve[v] = (int32_t)e, deg[v] = (int32_t)(e - i), vwk[v] = 0;
	for (int64_t u = pv + 1; u < (int64_t)v; ++u) vs[u] = 0, ve[u] = 0, deg[u] = 0, vwk[u] = 0;
}

__global__ __launch_bounds__(BLOCK) void k_deg(const int32_t *vs, const int32_t *ve, int n_vtx, int32_t *deg)
{
	int v = blockIdx.x * BLOCK + threadIdx.x;
	if (v < n_vtx) deg[v] = ve[v] - vs[v];
}

// number of pg_n_local calls of vertex v: n_max * n_weak (branch.c:70-75) + n(n-1)/2 (branch.c:83-88)
__device__ __forceinline__ int br_count_one(const int v, const int32_t *vs, const int32_t *ve, const int32_t *s1, const double bd)
{
	const int a0 = vs[v], n = ve[v] - a0;
	if (n < 2) return 0;
	int max_s1 = 0, n_max = 0, n_weak = 0;
	for (int i = 0; i < n; ++i) max_s1 = max_s1 > s1[a0 + i] ? max_s1 : s1[a0 + i];
	for (int i = 0; i < n; ++i) {
		n_max += s1[a0 + i] == max_s1;
		n_weak += (1.0 - (double)s1[a0 + i] / max_s1) > bd; // branch.c:71-72
	}
	return n_max * n_weak + n * (n - 1) / 2;
}
__global__ __launch_bounds__(BLOCK) void k_br_count(int n_vtx, const int32_t *vs, const int32_t *ve, const int32_t *s1, double bd, int32_t *pc, Gate gate)
{
	const int v = blockIdx.x * BLOCK + threadIdx.x;
	if (v >= n_vtx || gate_closed(gate)) return;
	pc[v] = br_count_one(v, vs, ve, s1, bd);
}

// Exclusive prefix sums of the vertices' pair counts in ONE workgroup: a graph has thousands of vertices, not millions, and the
// general scan costs two launches plus one more for the total.  The total goes to dcnt[15] and, with the other counters, to the
// host's mailbox.  (Graphs beyond PO_THREADS * PO_MAX_ITEMS vertices take the general scan.)
constexpr int PO_THREADS = 1024, PO_MAX_ITEMS = 64;
__global__ __launch_bounds__(PO_THREADS) void k_pair_offsets(const int32_t *pc, int n, int32_t *poff, int64_t *dcnt, int64_t *host_box, long long cap /* room in the pair list, or < 0 */, Gate gate)
{
	__shared__ int32_t part[PO_THREADS];
	if (gate_closed(gate)) return; // (uniform: the whole workgroup leaves)
	const int t = threadIdx.x, per = (n + PO_THREADS - 1) / PO_THREADS, i0 = t * per, i1 = i0 + per < n ? i0 + per : n;
	int32_t s = 0;
	for (int i = i0; i < i1; ++i) s += pc[i];
	part[t] = s;
	__syncthreads();
	for (int d = 1; d < PO_THREADS; d <<= 1) { // inclusive scan of the partial sums
		const int32_t v = t >= d ? part[t - d] : 0;
		__syncthreads();
		part[t] += v;
		__syncthreads();
	}
	int32_t run = part[t] - s;
	for (int i = i0; i < i1; ++i) { poff[i] = run; run += pc[i]; }
	if (t == PO_THREADS - 1) { dcnt[15] = part[t]; if (cap >= 0 && part[t] > cap) { dcnt[11] = 1; if (part[t] > dcnt[16]) dcnt[16] = part[t]; } } // [11]: sticky "a queued round could not be completed" (pga_branch_loop); [16]: the longest list that did not fit
	__syncthreads();
	if (t < 16) sys_store(&host_box[t], dcnt[t]);
}

// sequential form (one lane), used for vertices with more than 64 arcs.  MODE 1: write pairs; 2: decide.
template <int MODE>
__device__ void br_vertex_seq(int a0, int n, const int32_t *s1, const int32_t *agid, double bd, int64_t k, int32_t *pairs, const int32_t *cnt,
                              double bdist, double bcut, uint8_t *weak, int32_t *grp, int32_t *ndl_out, int64_t *dcnt)
{
	int max_s1 = 0;
	for (int i = 0; i < n; ++i) max_s1 = max_s1 > s1[a0 + i] ? max_s1 : s1[a0 + i];
	for (int i = 0; i < n; ++i) {
		const double r = 1.0 - (double)s1[a0 + i] / max_s1;
		if (!(r > bd)) continue;
		int n_local = 0;
		for (int j = 0; j < n; ++j) {
			if (s1[a0 + j] != max_s1) continue;
			if (MODE == 1) pairs[2 * k] = agid[a0 + j], pairs[2 * k + 1] = agid[a0 + i];
			if (MODE == 2) n_local += cnt[k];
			++k;
		}
		if (MODE == 2) {
			weak[a0 + i] = ((n_local == 0 && r > bdist) || r > bcut) ? 2 : 1;
			if (dcnt) atomicAdd((unsigned long long *)&dcnt[weak[a0 + i] - 1], 1ull);
		}
	}
	int n_group = 0;
	for (int i = 0; i < n; ++i) {
		if (MODE == 2 && grp[a0 + i] == 0) grp[a0 + i] = ++n_group;
		for (int j = i + 1; j < n; ++j) {
			if (MODE == 1) pairs[2 * k] = agid[a0 + i], pairs[2 * k + 1] = agid[a0 + j];
			if (MODE == 2 && cnt[k] > 0 && grp[a0 + j] == 0) grp[a0 + j] = grp[a0 + i];
			++k;
		}
	}
	if (MODE == 2) *ndl_out = n_group;
}

// A vertex with more than 64 arcs (a hub of a many-genome shard before pg_flt_high_occ has thinned the graph: the 12.1 M-hit shard has vertices with
// hundreds in its first rounds), by the whole wave instead of one lane (br_vertex_seq: 1.2 ms for one such launch): the arcs in chunks of 64, the
// same pair numbering.  Part 2's grouping is sequential in i by definition (the first i that is local with j names j's group); what a lane keeps of it
// is the group mark of its own j's, in the wave's slice of LDS (BR_WIDE_CAP arcs; beyond that the sequential form).
constexpr int BR_WIDE_CAP = 2048;
template <int MODE>
__device__ void br_vertex_wide(const int lane, int a0, int n, const int32_t *s1, const int32_t *agid, double bd, int64_t k0, int32_t *pairs, const int32_t *cnt,
                               double bdist, double bcut, uint8_t *weak, uint16_t *grp /* LDS, [n] */, int32_t *ndl_out, int64_t *dcnt)
{
	const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
	int max_s1 = 0;
	for (int c = 0; c < n; c += WAVE) { const int j = c + lane; const int v = j < n ? s1[a0 + j] : 0; max_s1 = max_s1 > v ? max_s1 : v; }
	max_s1 = wave_max(max_s1);
	int n_max = 0;
	for (int c = 0; c < n; c += WAVE) { const int j = c + lane; n_max += __popcll(__ballot(j < n && s1[a0 + j] == max_s1)); }
	// part 1 (branch.c:70-77): every weak arc i (ascending) against every best-scoring arc j (ascending)
	int64_t k = k0;
	for (int ci = 0; ci < n; ci += WAVE) {
		const int ii = ci + lane;
		const double r_l = ii < n ? 1.0 - (double)s1[a0 + ii] / max_s1 : 0.0;
		unsigned long long m_weak = __ballot(ii < n && r_l > bd);
		for (; m_weak; m_weak &= m_weak - 1, k += n_max) {
			const int bit = __ffsll((long long)m_weak) - 1, i = ci + bit;
			const int gid_i = agid[a0 + i];
			bool any_local = false;
			int base = 0;
			for (int cj = 0; cj < n; cj += WAVE) {
				const int j = cj + lane;
				const bool is_max = j < n && s1[a0 + j] == max_s1;
				const unsigned long long mm = __ballot(is_max);
				const int64_t kk = k + base + __popcll(mm & lt);
				if (MODE == 1) { if (is_max) pairs[2 * kk] = agid[a0 + j], pairs[2 * kk + 1] = gid_i; }
				else any_local = any_local || __ballot(is_max && cnt[kk] != 0) != 0;
				base += __popcll(mm);
			}
			if (MODE == 2 && lane == bit) {
				const int wk = ((!any_local && r_l > bdist) || r_l > bcut) ? 2 : 1;
				weak[a0 + i] = (uint8_t)wk;
				if (dcnt) atomicAdd((unsigned long long *)&dcnt[wk - 1], 1ull); // log only
			}
		}
	}
	// part 2 (branch.c:82-90): all i < j pairs, row i behind i * n - i (i + 1) / 2 earlier pairs
	if (MODE == 2) { for (int j = lane; j < n; j += WAVE) grp[j] = 0; wave_sync(); }
	int n_group = 0;
	for (int i = 0; i < n; ++i) {
		const int64_t row = k + (int64_t)i * n - (int64_t)i * (i + 1) / 2 - i - 1; // + j = the pair (i, j)
		if (MODE == 1) {
			const int gid_i = agid[a0 + i];
			for (int j = i + 1 + lane; j < n; j += WAVE) pairs[2 * (row + j)] = gid_i, pairs[2 * (row + j) + 1] = agid[a0 + j];
		} else {
			int gi = grp[i]; // (wave-uniform)
			if (gi == 0) { gi = ++n_group; if (lane == 0) grp[i] = (uint16_t)gi; }
			for (int j = i + 1 + lane; j < n; j += WAVE) if (grp[j] == 0 && cnt[row + j] > 0) grp[j] = (uint16_t)gi;
			wave_sync(); // (the marks of this row are read by the next one)
		}
	}
	if (MODE == 2 && lane == 0) *ndl_out = n_group;
}

// One WAVE per oriented vertex; lane j holds arc j (score, target gene, group mark) in registers and arcs are
// broadcast with shuffles, so the O(n^2) pair loops of branch.c:70-90 touch memory only for the pair list
// (coalesced stores, MODE 1) or the all-reduced counts (coalesced loads, MODE 2).
template <int MODE>
__device__ __forceinline__ void br_wave_body(const int v, const int lane, uint16_t *s_grp /* MODE 2: the workgroup's */, const int64_t k0 /* poff[v] */, const int32_t *vs, const int32_t *ve, const int32_t *s1g, const int32_t *agidg, double bd,
                                             int32_t *pairs, int64_t pair_cap, const int32_t *pcnt, const int32_t *cnt, double bdist, double bcut,
                                             uint8_t *weak, int32_t *grpg, int32_t *ndl, int64_t *dcnt, uint8_t *vwk)
{
	const int a0 = vs[v], n = ve[v] - a0;
	if (n < 2) { if (MODE == 2 && lane == 0) ndl[v] = 0; return; } // (every n_dist_loci entry is written: nothing to clear beforehand)
	if (MODE == 1 && k0 + pcnt[v] > pair_cap) return; // would run past the list: left out -- the total then exceeds the capacity too, the host sees that and repeats the round with room
	if (n > WAVE && n <= BR_WIDE_CAP) {
		int32_t g = 0;
		br_vertex_wide<MODE>(lane, a0, n, s1g, agidg, bd, k0, pairs, cnt, bdist, bcut, weak, MODE == 2 ? s_grp + (threadIdx.x >> 6) * BR_WIDE_CAP : (uint16_t *)nullptr, &g, dcnt);
		if (MODE == 2 && lane == 0) ndl[v] = g, vwk[v] = 1;
		return;
	}
	if (n > WAVE) {
		if (lane == 0) {
			int32_t g = 0;
			if (MODE == 2) for (int i = 0; i < n; ++i) grpg[a0 + i] = 0;
			br_vertex_seq<MODE>(a0, n, s1g, agidg, bd, k0, pairs, cnt, bdist, bcut, weak, grpg, &g, dcnt);
			if (MODE == 2) ndl[v] = g, vwk[v] = 1;
		}
		return;
	}
	const bool in = lane < n;
	const int my_s1 = in ? s1g[a0 + lane] : 0, my_gid = in ? agidg[a0 + lane] : 0;
	const int max_s1 = wave_max(my_s1);
	const double r = in ? 1.0 - (double)my_s1 / max_s1 : 0.0; // branch.c:71
	const bool is_weak = in && r > bd, is_max = in && my_s1 == max_s1;
	const unsigned long long m_weak = __ballot(is_weak), m_max = __ballot(is_max);
	if (MODE == 2 && lane == 0 && m_weak) vwk[v] = 1;
	const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
	const int n_max = __popcll(m_max), mrank = __popcll(m_max & lt);
	// part 1 (branch.c:70-77): for every weak arc i (ascending), one pair per best-scoring arc j (ascending)
	int wb = 0;
	for (unsigned long long m = m_weak; m; m &= m - 1, ++wb) {
		const int i = __ffsll((long long)m) - 1;
		const int gid_i = __builtin_amdgcn_readlane(my_gid, i); // i is wave-uniform
		const int64_t k = k0 + (int64_t)wb * n_max + mrank;
		if (MODE == 1) { if (is_max) pairs[2 * k] = my_gid, pairs[2 * k + 1] = gid_i; }
		else {
			const bool none_local = __ballot(is_max && cnt[k] != 0) == 0; // the counts are >= 0: their sum is 0 iff all are
			if (lane == i) {
				const int wk = ((none_local && r > bdist) || r > bcut) ? 2 : 1;
				weak[a0 + i] = (uint8_t)wk;
				if (dcnt) atomicAdd((unsigned long long *)&dcnt[wk - 1], 1ull); // log only
			}
		}
	}
	// part 2 (branch.c:82-90): all i<j pairs, row i starts after i*n - i(i+1)/2 earlier pairs
	const int64_t k2 = k0 + (int64_t)n_max * __popcll(m_weak);
	int grp = 0, n_group = 0;
	for (int i = 0; i < n; ++i) {
		const int64_t k = k2 + (int64_t)i * n - (int64_t)i * (i + 1) / 2 + (lane - i - 1);
		if (MODE == 1) {
			const int gid_i = __builtin_amdgcn_readlane(my_gid, i);
			if (lane > i && in) pairs[2 * k] = gid_i, pairs[2 * k + 1] = my_gid;
		} else {
			int gi = __builtin_amdgcn_readlane(grp, i);
			if (gi == 0) { gi = ++n_group; if (lane == i) grp = gi; } // uniform: every lane sees the same gi
			if (lane > i && in && grp == 0 && cnt[k] > 0) grp = gi;
		}
	}
	if (MODE == 2 && lane == 0) ndl[v] = n_group; // (pinned host memory in the unsharded path: plain stores, released when the host asks the runtime about the stream)
}
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_br_wave(int n_vtx, const int32_t *vs, const int32_t *ve, const int32_t *s1g, const int32_t *agidg, double bd,
                                                     const int32_t *poff, int32_t *pairs, int64_t pair_cap /* MODE 1: room in pairs[] */, const int32_t *pcnt /* MODE 1: pairs of each vertex */, const int32_t *cnt, double bdist, double bcut,
                                                     uint8_t *weak, int32_t *grpg, int32_t *ndl, int64_t *dcnt, uint8_t *vwk /* MODE 2: vertex has a weak arc */,
                                                     const int64_t *np_dev = nullptr /* MODE 2: the number of pairs, when the list has a capacity (pair_cap) */, Gate gate = Gate{nullptr, 0})
{
	__shared__ uint16_t s_grp[MODE == 2 ? (BLOCK / WAVE) * BR_WIDE_CAP : 1];
	const int v = blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (v >= n_vtx || gate_closed(gate)) return;
	if (MODE == 2 && np_dev && *np_dev > pair_cap) return; // the list overflowed: there are no counts to read, the host repeats the step with room
	br_wave_body<MODE>(v, lane, s_grp, poff[v], vs, ve, s1g, agidg, bd, pairs, pair_cap, pcnt, cnt, bdist, bcut, weak, grpg, ndl, dcnt, vwk);
}

// ------------------------------------------------------------------------------------------------
// The front of a queued branch round in two launches instead of five (round 6; pga_branch_loop).  pg_gen_rep_pos (ranks -> records) and the pair
// enumeration (counts -> offsets -> list) are two chains that share nothing until pg_n_local reads both: the links of the same depth go
// into one launch, workgroups of either kind side by side.
//   k_loop_front1 (RK_T threads):  [0, nbc)      pair counts of 1 024 vertices + their offsets INSIDE the workgroup + the workgroup's total (btot)
//                                  [nbc, +GL)    k_rank_genome's workgroup of genome b - nbc
//                                  the rest      k_rep_clear (live lists only)
//   k_loop_front2 (BLOCK threads): [0, nb_rep)   k_rep_fill
//                                  the rest      k_br_wave<1>, every wave adding the totals of the workgroups in front of its vertex's (at most 64:
//                                                the loop takes graphs of up to 65 536 vertices) -- there is no launch for the offsets any more
// ------------------------------------------------------------------------------------------------
struct LoopFront {
	int n_vtx, nbc, GL; const int32_t *vs, *ve, *s1; double bd; int32_t *pc, *poff, *btot; // counts
	const uint32_t *flags; const int32_t *goff; int32_t *rx;                                 // ranks
	void *rp_out; int64_t n_ent; int clear_bytes;                                            // clear (0: none, 8 / 16: record size)
	Gate gate;
};
__global__ __launch_bounds__(RK_T) void k_loop_front1(LoopFront a)
{
	__shared__ int wtot[2][RK_T / WAVE];
	if (gate_closed(a.gate)) return;
	const int b = blockIdx.x, tid = threadIdx.x;
	if (b < a.nbc) {
		const int v = b * RK_T + tid, lane = tid & 63, w = tid >> 6;
		const int pc = v < a.n_vtx ? br_count_one(v, a.vs, a.ve, a.s1, a.bd) : 0;
		const I32 incl = wave_scan_incl(I32{pc}, OpSum{}, lane);
		if (lane == 63) wtot[0][w] = incl.v;
		__syncthreads();
		int pre = 0, tot = 0;
#pragma unroll
		for (int k = 0; k < RK_T / WAVE; ++k) { const int t = wtot[0][k]; tot += t; if (k < w) pre += t; }
		if (v < a.n_vtx) a.pc[v] = pc, a.poff[v] = pre + incl.v - pc;
		if (tid == 0) a.btot[b] = tot;
	} else if (b < a.nbc + a.GL) {
		rank_genome_body(a.flags, a.goff, a.rx, b - a.nbc, wtot);
	} else {
		const int64_t e = (int64_t)(b - a.nbc - a.GL) * RK_T + tid;
		if (e < a.n_ent) { if (a.clear_bytes == 8) ((int2 *)a.rp_out)[e] = make_int2(0, -1); else ((int4 *)a.rp_out)[e] = make_int4(-1, 0, 0, 0); }
	}
}
struct LoopFront2 {
	int nb_rep, n_vtx, nbc; const int32_t *vs, *ve, *s1, *agid; double bd; int32_t *poff; const int32_t *btot, *pc; int32_t *pairs; int64_t pair_cap; int64_t *dcnt;
};
template <int FORM, bool CLEARED>
__global__ __launch_bounds__(BLOCK) void k_loop_front2(RepFill rf, LoopFront2 a)
{
	if (gate_closed(rf.gate)) return;
	if ((int)blockIdx.x < a.nb_rep) { rep_fill_body<FORM, CLEARED>(rf, blockIdx.x * BLOCK + threadIdx.x); return; }
	const int v = (blockIdx.x - a.nb_rep) * (BLOCK / WAVE) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (v >= a.n_vtx) return;
	const int part = lane < a.nbc ? a.btot[lane] : 0, bv = v / RK_T;
	const int base = wave_sum(lane < bv ? part : 0);
	const int64_t k0 = (int64_t)base + a.poff[v];
	if (lane == 0) a.poff[v] = (int32_t)k0; // (what k_br_wave<2> reads; only this wave touches the entry)
	if (v == 0) { // k_pair_offsets' other results: the number of pairs, and "a queued round could not be completed" (a list beyond its capacity)
		const int tot = wave_sum(part);
		if (lane == 0) { a.dcnt[15] = tot; if (tot > a.pair_cap) { a.dcnt[11] = 1; if (tot > a.dcnt[16]) a.dcnt[16] = tot; } } // ([16]: the longest list that did not fit -- pga_branch_loop's next attempt makes room)
	}
	br_wave_body<1>(v, lane, nullptr, k0, a.vs, a.ve, a.s1, a.agid, a.bd, a.pairs, a.pair_cap, a.pc, nullptr, 0.0, 0.0, nullptr, nullptr, nullptr, a.dcnt, nullptr);
}

// pg_flt_high_occ's three tests (graph.c:226-258) for every segment of the round, from what the round left on the device:
// seg_cnt[S + s] = tot_cnt (graph.c:126), deg[] = out-degree of each oriented vertex, ndl[] = n_dist_loci of the branch step
__global__ __launch_bounds__(BLOCK) void k_round_filter(int S, const int32_t *seg_cnt, const int32_t *deg, const int32_t *ndl, int max_tot_cnt, int max_degree, int max_dist_loci, uint8_t *del)
{
	const int s = blockIdx.x * BLOCK + threadIdx.x;
	if (s >= S) return;
	const int l0 = ndl[2 * s], l1 = ndl[2 * s + 1];
	del[s] = (seg_cnt[S + s] > max_tot_cnt || deg[2 * s] > max_degree || deg[2 * s + 1] > max_degree || (l0 > l1 ? l0 : l1) > max_dist_loci) ? 1 : 0;
}

// pga_branch_loop: k_round_filter's tests and their consequences in one launch.  A deleted segment keeps its number: its gene loses its
// vertex, its two vertices their arcs and counters (nothing refers to them from then on: hits of the gene are filtered next).
__global__ __launch_bounds__(BLOCK) void k_round_del(int S, const int32_t *ndl, int max_tot_cnt, int max_degree, int max_dist_loci, const int32_t *seg_gid, int32_t *g2s, int32_t *vs, int32_t *ve, int32_t *deg, int32_t *seg_cnt, uint8_t *vwk, uint8_t *alive, int4 *gmeta /* or NULL */,
                 int32_t *stamp = nullptr /* Gate::w, or NULL */, int round = 0)
{
	const int s = blockIdx.x * BLOCK + threadIdx.x;
	if (s >= S || !alive[s]) return;
	const int l0 = ndl[2 * s], l1 = ndl[2 * s + 1];
	if (!(seg_cnt[S + s] > max_tot_cnt || deg[2 * s] > max_degree || deg[2 * s + 1] > max_degree || (l0 > l1 ? l0 : l1) > max_dist_loci)) return; // k_round_filter's tests
	alive[s] = 0;
	if (stamp) stamp[0] = round; // the state of the loop changed in this round (every writer of the round writes the same value)
	g2s[seg_gid[s]] = -1;
	vs[2 * s] = ve[2 * s] = vs[2 * s + 1] = ve[2 * s + 1] = 0;
	deg[2 * s] = deg[2 * s + 1] = 0;
	seg_cnt[s] = seg_cnt[S + s] = 0;
	vwk[2 * s] = vwk[2 * s + 1] = 0;
	if (gmeta) gmeta[s] = make_int4(0, 0, 0, 0); // (sharded rounds compact the genes' stretches: this one has none from now on)
}

__device__ __forceinline__ int arc_weak(const uint64_t *ax, const uint8_t *aw, int64_t n, uint64_t x) // pg_get_arc, pgpriv.h:99-107
{
	int64_t lo = 0, hi = n;
	while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (ax[mid] < x) lo = mid + 1; else hi = mid; }
	return (lo < n && ax[lo] == x) ? aw[lo] : 0;
}

// pg_get_arc as in the reference (pgpriv.h:99-107): scan the few arcs leaving v; vs/ve = arc range of each vertex
__device__ __forceinline__ int arc_weak_v(const uint64_t *ax, const uint8_t *aw, const int32_t *vs, const int32_t *ve, uint32_t v, uint32_t w)
{
	const int e = ve[v];
	for (int i = vs[v]; i < e; i += 4) { // four targets per round trip (the list is a handful of arcs)
		const uint32_t t0 = (uint32_t)ax[i], t1 = i + 1 < e ? (uint32_t)ax[i + 1] : ~0u, t2 = i + 2 < e ? (uint32_t)ax[i + 2] : ~0u, t3 = i + 3 < e ? (uint32_t)ax[i + 3] : ~0u;
		const int k = t0 == w ? 0 : t1 == w ? 1 : t2 == w ? 2 : t3 == w ? 3 : -1;
		if (k >= 0) return aw[i + k];
	}
	return 0;
}

__global__ __launch_bounds__(BLOCK) void k_mark_hits(const int32_t *val, const int32_t *prev, const int4 *YA, const int4 *YB, const int32_t *g2s, int n,
                                                       const uint64_t *ax, const uint8_t *aw, int64_t n_arc, const int32_t *vs, const int32_t *ve, const uint8_t *vwk, int32_t *weak_new)
{
	int y = blockIdx.x * BLOCK + threadIdx.x;
	if (y >= n || val[y] < 0) return;
	int p = prev[y];
	if (p < 0) return;
	const int4 aA = YA[y], bA = YA[p];
	if (aA.x != bA.x) return; // branch.c:124
	const int aw_ = YB[y].w, bw_ = YB[p].w; // X position << 1 | rev
	uint32_t w = (uint32_t)g2s[aA.y] << 1 | (uint32_t)(aw_ & 1);
	uint32_t v = (uint32_t)g2s[bA.y] << 1 | (uint32_t)(bw_ & 1);
	if (vwk && !vwk[v] && !vwk[w ^ 1]) return; // neither vertex has a weak out-arc (the common case): nothing to look up
	int e1 = vs ? arc_weak_v(ax, aw, vs, ve, v, w) : arc_weak(ax, aw, n_arc, (uint64_t)v << 32 | w);                       // branch.c:128-130: marks the earlier hit
	if (e1) atomicMax(&weak_new[bw_ >> 1], e1);
	int e2 = vs ? arc_weak_v(ax, aw, vs, ve, w ^ 1, v ^ 1) : arc_weak(ax, aw, n_arc, (uint64_t)(w ^ 1) << 32 | (v ^ 1)); // branch.c:131-133: marks this hit
	if (e2) atomicMax(&weak_new[aw_ >> 1], e2);
}

__global__ __launch_bounds__(BLOCK) void k_weak_merge(uint32_t *flags, const int32_t *weak_new, int n, int64_t *cnt)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	const bool in = h < n;
	if (!in) h = n - 1;
	uint32_t f = flags[h];
	int cur = in ? (int)((f & PGA_F_WEAK_MASK) >> PGA_F_WEAK_SHIFT) : 0, nw = in ? weak_new[h] : 0;
	if (nw > cur) { cur = nw; flags[h] = (f & ~PGA_F_WEAK_MASK) | (uint32_t)nw << PGA_F_WEAK_SHIFT; }
	if (cnt) { // log-only counter (branch.c:137-139): one atomic per wave, and only when somebody asks
		const unsigned long long m = __ballot(cur != 0);
		if (m && (threadIdx.x & 63) == (unsigned)__ffsll((long long)m) - 1) atomicAdd((unsigned long long *)cnt, (unsigned long long)__popcll(m));
	}
}
