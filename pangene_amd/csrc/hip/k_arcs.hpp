// k_arcs.hpp -- pg_gen_arc: walk marks, adjacency arcs, two-level collapse, cross-shard merge.
// Included by pga_backend.hip (one translation unit); uses the context types, BLOCK / WAVE and dev_prims.hpp from there.
#pragma once

// ------------------------------------------------------------------------------------------------
// pg_gen_arc, per-genome part (graph.c:97-146)
// ------------------------------------------------------------------------------------------------
constexpr int SEGCNT_COPIES = 64;
__global__ __launch_bounds__(BLOCK) void k_segcnt_sum(int32_t *seg_cnt, int n2s)
{
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n2s) return;
	int t = 0;
	for (int k = 0; k < SEGCNT_COPIES; ++k) t += seg_cnt[(int64_t)k * n2s + i];
	seg_cnt[i] = t;
}

// walkable = !flt && !shadow; val[y] = y if the y-th hit in cm order is walkable else -1
// walkable mark of Y position y: y itself if the hit there is neither filtered nor shadowed, else -1 (graph.c:108).  The exclusive
// running maximum of the marks is the previous walkable hit; a mark is recovered from the scan itself (the marks grow with y, so
// the inclusive maximum differs from the exclusive one exactly at walkable positions): no separate marking pass.
struct InWalk {
	const uint32_t *flags; const int32_t *yperm;
	__device__ __forceinline__ I32 operator()(int64_t i) const { return I32{(flags[yperm[i]] & (PGA_F_FLT | PGA_F_SHADOW)) ? -1 : (int32_t)i}; }
};
struct OutPrev {
	int32_t *val, *prev;
	__device__ __forceinline__ void operator()(int64_t i, I32 incl, I32 ex) const { val[i] = incl.v != ex.v ? incl.v : -1, prev[i] = ex.v; }
};
struct InKeyHead { const uint64_t *key; __device__ __forceinline__ I32 operator()(int64_t i) const { return I32{(i == 0 || key[i] != key[i - 1]) ? 1 : 0}; } };

// Static per-hit fields in Y (cm) order, packed once per run: the arc kernels walk the hits in that order and would otherwise
// gather every field through yperm.   YA = {seg, gid, genome, cm}   YB = {score_ori, score_dom, gene of pid_dom0's protein
// (-1: none), X position << 1 | rev}
// With virtual contigs (pga_genome_block_t; vfirst != NULL) "seg" is the segment id of the contig's FIRST piece -- the walk's "same
// contig" test (graph.c:114, branch.c:124) then sees the contig, not the piece -- and "cm" the low 32 bits of the true 64-bit cm: the
// walk only ever takes the difference of two of them into an int32_t (graph.c:73,117: p->dist = a->cm - vpos), and the low word of a
// difference is the difference of the low words.  (Equal low words of different cm -- 2^32 bp apart -- raise a needless H2a hazard.)
__global__ __launch_bounds__(BLOCK) void k_pack_yrec(const int32_t *yperm, const int32_t *seg, const int32_t *gid, const int32_t *gnm, const int32_t *cm,
                                                       const int32_t *sori, const int32_t *sdom, const int32_t *pdom0, const int32_t *prot_gid, const uint32_t *flags,
                                                       int n, int4 *YA, int4 *YB, const int32_t *vfirst, const int64_t *vbase)
{
	int y = blockIdx.x * BLOCK + threadIdx.x;
	if (y >= n) return;
	const int a = yperm[y], p0 = pdom0[a];
	int sg = seg[a], cmv = cm[a];
	if (vfirst) { cmv = (int)(unsigned)((unsigned long long)vbase[sg] + (unsigned long long)(long long)cmv); sg = vfirst[sg]; }
	YA[y] = make_int4(sg, gid[a], gnm[a], cmv);
	YB[y] = make_int4(sori[a], sdom[a], p0 < 0 ? -1 : prot_gid[p0], a << 1 | (flags[a] & PGA_F_REV ? 1 : 0));
}

// has_arc[y] = 1 if walkable y has a walkable predecessor on the same contig; also per-segment counts
// (graph.c:113,125-126) and hazard H2a (equal cm of two consecutive walkable hits)
__global__ __launch_bounds__(BLOCK) void k_arc_flag(const int32_t *val, const int32_t *prev, const int4 *YA, const int32_t *g2s, int n, int S, int32_t *has, int32_t *seg_cnt,
                                                      uint32_t *seen, int64_t words_per_genome, int64_t *dcnt, int32_t *hz_list)
{
	int y = blockIdx.x * BLOCK + threadIdx.x;
	if (y >= n) return;
	int out = 0;
	if (val[y] >= 0) {
		const int4 ra = YA[y];
		const int sid = g2s[ra.y];
		if (sid < 0) atomicAdd((unsigned long long *)&dcnt[3], 1ull); // graph.c:111
		else {
			int32_t *copy = seg_cnt + (int64_t)(blockIdx.x & (SEGCNT_COPIES - 1)) * 2 * S; // 64 copies: 64x less contention per address
			atomicAdd(&copy[S + sid], 1);
			uint32_t old = atomicOr(&seen[(int64_t)ra.z * words_per_genome + (sid >> 5)], 1u << (sid & 31));
			if (!(old >> (sid & 31) & 1u)) atomicAdd(&copy[sid], 1);
		}
		const int p = prev[y];
		if (p >= 0) {
			const int4 rb = YA[p];
			if (rb.x == ra.x) {
				out = 1;
				if (rb.w == ra.w) { atomicAdd((unsigned long long *)&dcnt[5], 1ull); hz_note(&dcnt[14], hz_list, ra.x); }
			}
		}
	}
	has[y] = out;
}

__device__ __forceinline__ int arc_score(const int4 yb, int ori, const int32_t *g2s)
{ // pg_get_score, graph.c:82-85: score_ori unless the dominator's gene is not a vertex and score_dom is at least as large
	return (ori || yb.x > yb.y || yb.z < 0 || g2s[yb.z] >= 0) ? yb.x : yb.y;
}

struct ArcEmit {
	const int32_t *has, *slot, *prev; const int4 *YA, *YB; const int32_t *g2s;
	uint64_t *key; uint32_t *idx; int4 *pay; // payload {dist, s1, s2, genome}
	int n, ori, vbits;
};

__global__ __launch_bounds__(BLOCK) void k_arc_emit(ArcEmit e)
{
	int y = blockIdx.x * BLOCK + threadIdx.x;
	if (y >= e.n || !e.has[y]) return;
	const int p = e.prev[y];
	const int4 aA = e.YA[y], bA = e.YA[p], aB = e.YB[y], bB = e.YB[p];
	uint32_t w = (uint32_t)e.g2s[aA.y] << 1 | (uint32_t)(aB.w & 1);
	uint32_t v = (uint32_t)e.g2s[bA.y] << 1 | (uint32_t)(bB.w & 1);
	int sa = arc_score(aB, e.ori, e.g2s);
	int sb = arc_score(bB, e.ori, e.g2s);
	int d = (int)((unsigned)aA.w - (unsigned)bA.w), g = aA.z; // (unsigned: with virtual contigs the words are the low halves of 64-bit coordinates)
	int64_t o = (int64_t)e.slot[y] * 2;
	e.key[o] = (uint64_t)v << e.vbits | w;           e.idx[o] = (uint32_t)o;         // v -> w      (graph.c:117)
	e.pay[o] = make_int4(d, sb, sa, g);
	e.key[o + 1] = (uint64_t)(w ^ 1) << e.vbits | (v ^ 1); e.idx[o + 1] = (uint32_t)(o + 1); // w^1 -> v^1 (graph.c:119)
	e.pay[o + 1] = make_int4(d, sa, sb, g);
}

__global__ __launch_bounds__(BLOCK) void k_arc_gather(const uint32_t *idx, int64_t m, const int4 *pay, int4 *opay)
{
	int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i < m) opay[i] = pay[idx[i]]; // one random 16-byte read per temp arc
}

__global__ __launch_bounds__(BLOCK) void k_arc_head(const uint64_t *key, int64_t m, int32_t *head)
{
	int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i >= m) return;
	head[i] = (i == 0 || key[i] != key[i - 1]) ? 1 : 0;
}

// Two-level collapse of the sorted temp arcs (graph.c:128-175).  Equal keys are adjacent and, inside one key,
// grouped by genome (stable sort of a genome-major emission).
// level 1: the first element of every (key, genome) run collapses its run -- almost always a single element --
//          into (n, rounded mean dist * n, max s1, max s2) stored at its own position; other positions hold zeros;
// level 2: one wave per distinct key sums those records over the key's run with coalesced strided reads.
__global__ __launch_bounds__(BLOCK) void k_arc_l1(const uint64_t *key, int64_t m, const int4 *pay, const int32_t *slot, int32_t *run_start,
                                                    int32_t *o_n, uint64_t *o_dn, int32_t *o_s1, int32_t *o_s2)
{
	int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i >= m) return;
	const uint64_t k = key[i];
	const int4 p = pay[i];
	const int g = p.w;
	if (i == 0 || key[i - 1] != k) run_start[slot[i]] = (int32_t)i; // head of the key's run
	if (i > 0 && key[i - 1] == k && pay[i - 1].w == g) { o_n[i] = 0, o_dn[i] = 0, o_s1[i] = 0, o_s2[i] = 0; return; }
	int n = 1, m1 = p.y > 0 ? p.y : 0, m2 = p.z > 0 ? p.z : 0; // the reference's running maxima start at 0 (graph.c:133): non-positive scores count as 0
	uint64_t sd = (uint64_t)(int64_t)p.x;
	for (int64_t j = i + 1; j < m && key[j] == k; ++j) { // almost always empty: one adjacency per (arc, genome)
		const int4 q = pay[j];
		if (q.w != g) break;
		sd += (uint64_t)(int64_t)q.x;
		m1 = m1 > q.y ? m1 : q.y;
		m2 = m2 > q.z ? m2 : q.z;
		++n;
	}
	const int dg = cvt_i32_x86((double)sd / n + .499); // graph.c:141 (sd: uint64_t as there)
	o_n[i] = n, o_dn[i] = (uint64_t)(int64_t)dg * (uint64_t)n, o_s1[i] = m1, o_s2[i] = m2;
}

__global__ __launch_bounds__(BLOCK) void k_arc_l2(const uint64_t *key, int64_t m, int64_t n_run, const int32_t *run_start, const int32_t *c_n, const uint64_t *c_dn,
                                                    const int32_t *c_s1, const int32_t *c_s2, int vbits, pga_arc_part_t *out)
{
	const int64_t w = (int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	if (w >= n_run) return;
	const int64_t st = run_start[w], en = w + 1 < n_run ? run_start[w + 1] : m;
	int ng = 0, tot = 0;
	uint64_t sd = 0;
	int64_t a1 = 0, a2 = 0;
	for (int64_t j = st + lane; j < en; j += WAVE) {
		const int n = c_n[j];
		ng += n > 0, tot += n, sd += c_dn[j], a1 += c_s1[j], a2 += c_s2[j];
	}
	ng = wave_sum(ng), tot = wave_sum(tot);
	sd = wave_sum64(sd), a1 = (int64_t)wave_sum64((unsigned long long)a1), a2 = (int64_t)wave_sum64((unsigned long long)a2);
	if (lane == 0) {
		const uint64_t k = key[st];
		pga_arc_part_t r;
		r.x = (k >> vbits) << 32 | (k & ((1ull << vbits) - 1));
		r.n_genome = ng, r.tot_cnt = tot, r.sum_dist = sd, r.sum_s1 = a1, r.sum_s2 = a2;
		out[w] = r;
	}
}


// ------------------------------------------------------------------------------------------------
// cross-shard merge of arc tables (after the all-gather): gather valid entries, sort by x, wave-per-run sums
// ------------------------------------------------------------------------------------------------
// Every rank's table arrives sorted by x with unique keys, so the merged order needs no sort: the place of an entry is
// the number of entries before it in all the tables (binary searches; equal keys keep rank order).
struct MergeLists { int32_t W; int64_t slot_sz; const int64_t *off; }; // off[r] = entries of ranks < r, off[W] = total

__device__ __forceinline__ int64_t mg_bound(const pga_arc_part_t *a, int64_t n, uint64_t x, bool upper)
{
	int64_t lo = 0, hi = n;
	while (lo < hi) {
		const int64_t mid = (lo + hi) >> 1;
		const uint64_t y = a[mid].x;
		if (upper ? y <= x : y < x) lo = mid + 1; else hi = mid;
	}
	return lo;
}

__global__ __launch_bounds__(BLOCK) void k_mg_rank(const pga_arc_part_t *g, MergeLists L, uint64_t *key, uint32_t *val)
{
	const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i >= L.off[L.W]) return;
	int r = 0;
	while (L.off[r + 1] <= i) ++r; // the table entry i belongs to (W is small)
	const int64_t k = i - L.off[r], src = r * L.slot_sz + k;
	const uint64_t x = g[src].x;
	int64_t pos = k;
	for (int q = 0; q < L.W; ++q)
		if (q != r) pos += mg_bound(g + q * L.slot_sz, L.off[q + 1] - L.off[q], x, q < r);
	key[pos] = x, val[pos] = (uint32_t)src;
}

struct InMgHead { const uint64_t *key; __device__ __forceinline__ I32 operator()(int64_t i) const { return I32{(i == 0 || key[i] != key[i - 1]) ? 1 : 0}; } };

__global__ __launch_bounds__(BLOCK) void k_mg_count(const uint64_t *key, const int32_t *slot, int64_t m, int64_t *box) // number of distinct keys
{
	if (blockIdx.x == 0 && threadIdx.x == 0) *box = slot[m - 1] + ((m == 1 || key[m - 1] != key[m - 2]) ? 1 : 0); // slot = exclusive count of run heads
}

__global__ __launch_bounds__(BLOCK) void k_mg_runstart(const uint64_t *key, const int32_t *slot, int64_t m, int32_t *run_start)
{
	int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i < m && (i == 0 || key[i] != key[i - 1])) run_start[slot[i]] = (int32_t)i;
}

__global__ __launch_bounds__(BLOCK) void k_mg_sum(const pga_arc_part_t *g, const uint32_t *val, int64_t m, int64_t n_run, const int32_t *run_start, pga_arc_part_t *out)
{
	const int64_t w = (int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	if (w >= n_run) return;
	const int64_t st = run_start[w], en = w + 1 < n_run ? run_start[w + 1] : m;
	int ng = 0, tot = 0;
	uint64_t sd = 0, x = 0;
	int64_t a1 = 0, a2 = 0;
	for (int64_t j = st + lane; j < en; j += WAVE) {
		const pga_arc_part_t p = g[val[j]];
		x = p.x, ng += p.n_genome, tot += p.tot_cnt, sd += p.sum_dist, a1 += p.sum_s1, a2 += p.sum_s2;
	}
	ng = wave_sum(ng), tot = wave_sum(tot);
	sd = wave_sum64(sd), a1 = (int64_t)wave_sum64((unsigned long long)a1), a2 = (int64_t)wave_sum64((unsigned long long)a2);
	if (lane == 0) { // lane 0 always owns element st
		pga_arc_part_t r;
		r.x = x, r.n_genome = ng, r.tot_cnt = tot, r.sum_dist = sd, r.sum_s1 = a1, r.sum_s2 = a2;
		out[w] = r;
	}
}

// ------------------------------------------------------------------------------------------------
// the same merge inside the sharded form of pga_branch_loop: every count stays in device memory
// ------------------------------------------------------------------------------------------------
// A rank's slot of the round's all-gather, in 32-bit words: [0] arcs in its table (the true number, also when it is beyond
// the slot's capacity), [1] / [2] the rank's round is void (hub gene / invariant), [3..15] 0; [16, 16 + 2S) its segment counters (graph.c:125-126); then, 8-byte aligned, arc_cap table
// entries sorted by x.  Grids are sized by capacities the host knows; what is really there is read from the slots.
constexpr int XS_HDR = 16;
__host__ __device__ inline int64_t xs_seg_words(int S) { return ((int64_t)2 * S + 1) & ~(int64_t)1; }
__host__ __device__ inline int64_t xs_slot_words(int S, int64_t arc_cap) { return XS_HDR + xs_seg_words(S) + arc_cap * (int64_t)(sizeof(pga_arc_part_t) / sizeof(int32_t)); }
struct XSlots {
	const int32_t *all; int64_t slot_words, arc_cap; int W, S;
	__device__ __forceinline__ const pga_arc_part_t *arcs(int r) const { return (const pga_arc_part_t *)(all + r * slot_words + XS_HDR + xs_seg_words(S)); }
};

// One launch after the all-gather: the global segment counters (sum over the slots), the place of every table entry in the merged
// order (every rank's table arrives sorted by x with distinct keys: binary searches in the other tables, equal keys keep rank order)
// and, by thread 0, off[r] = entries of ranks < r for the kernels that follow (a table beyond its slot raises the sticky flag: the
// round, and with it the run, is void), the largest table and the longest pair list seen (the next run's capacities), the ranks' flags.
__global__ __launch_bounds__(BLOCK) void k_xs_sum_rank(XSlots X, int32_t *seg_cnt, int64_t *off, int64_t *dcnt, int64_t *xstat, uint64_t *key, uint32_t *val, int32_t *stamp = nullptr /* Gate::w of the queued rounds, or NULL */, int round = 0)
{
	const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i == 0 && stamp) { // some rank marked a hit in this round (header word 3, written by k_xs_compact): the state of the loop changed for everybody
		bool any = false;
		for (int r = 0; r < X.W; ++r) any = any || X.all[r * X.slot_words + 3] != 0;
		if (any) stamp[1] = round;
	}
	if (i < 2 * X.S) {
		int32_t t = 0;
		for (int r = 0; r < X.W; ++r) t += X.all[r * X.slot_words + XS_HDR + i];
		seg_cnt[i] = t;
	}
	// A table beyond its slot was not copied in full: what its slot holds is no table.  The round is void then (sticky flag) -- and
	// EMPTY, so that whatever still runs on it before the host gets to know runs on nothing instead of on leftovers.
	bool void_round = false;
	for (int q = 0; q < X.W; ++q) void_round = void_round || X.all[q * X.slot_words] > X.arc_cap;
	if (i == 0) {
		int64_t run = 0, mx = 0;
		off[0] = 0;
		for (int r = 0; r < X.W; ++r) {
			const int64_t n = X.all[r * X.slot_words];
			mx = mx > n ? mx : n;
			if (X.all[r * X.slot_words + 1]) xstat[3] = 1;
			if (X.all[r * X.slot_words + 2]) xstat[4] = 1;
			run += void_round ? 0 : n, off[r + 1] = run;
		}
		if (void_round) dcnt[11] = 1, xstat[2] = 1;
		if (mx > xstat[1]) xstat[1] = mx;
		if (dcnt[15] > xstat[0]) xstat[0] = dcnt[15];
	}
	if (void_round) return;
	// (every thread derives the tables' sizes from the W headers itself: off[] is only written here)
	int r = -1;
	int64_t base = 0;
	for (int q = 0; q < X.W; ++q) {
		const int64_t n = X.all[q * X.slot_words];
		if (i < base + n) { r = q; break; }
		base += n;
	}
	if (r < 0) return;
	const int64_t k = i - base;
	const uint64_t x = X.arcs(r)[k].x;
	int64_t pos = k;
	for (int q = 0; q < X.W; ++q)
		if (q != r) pos += mg_bound(X.arcs(q), (int64_t)X.all[q * X.slot_words], x, q < r);
	key[pos] = x, val[pos] = (uint32_t)(r * X.arc_cap + k);
}

struct InMgHeadN { const uint64_t *key; const int64_t *n; __device__ __forceinline__ I32 operator()(int64_t i) const { return I32{(i < *n && (i == 0 || key[i] != key[i - 1])) ? 1 : 0}; } };

// The merged table: the first entry of a run of equal keys sums the run (at most W entries: a rank's keys are distinct) into its place
// slot[i] = number of runs before it; the last entry leaves the number of runs = the table's size.
__global__ __launch_bounds__(BLOCK) void k_mgx_heads_sum(XSlots X, const uint64_t *key, const uint32_t *val, const int32_t *slot, const int64_t *m_dev, pga_arc_part_t *out, int64_t *n_run)
{
	const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x, m = *m_dev;
	if (i == 0 && m == 0) *n_run = 0;
	if (i >= m) return;
	const uint64_t x = key[i];
	const bool head = i == 0 || key[i - 1] != x;
	if (i == m - 1) *n_run = slot[i] + (head ? 1 : 0);
	if (!head) return;
	pga_arc_part_t r;
	{ const uint32_t v = val[i]; r = X.arcs((int)(v / X.arc_cap))[v % X.arc_cap]; }
	for (int64_t j = i + 1; j < m && key[j] == x; ++j) {
		const uint32_t v = val[j];
		const pga_arc_part_t p = X.arcs((int)(v / X.arc_cap))[v % X.arc_cap];
		r.n_genome += p.n_genome, r.tot_cnt += p.tot_cnt, r.sum_dist += p.sum_dist, r.sum_s1 += p.sum_s1, r.sum_s2 += p.sum_s2;
	}
	out[slot[i]] = r;
}

// what the caller is told at the end (summed over the ranks): [0] some queued round is void, [1] an invariant was violated, [2] an
// exchange buffer was too small (learnable: the capacities follow xstat), [3] a hub gene overflowed its LDS table
__global__ void k_xs_flags(int64_t *dcnt, int64_t *xstat, long long pair_cap, int32_t *out4)
{
	if (threadIdx.x == 0 && blockIdx.x == 0) {
		if (dcnt[15] > xstat[0]) xstat[0] = dcnt[15];
		out4[0] = dcnt[11] != 0, out4[1] = dcnt[3] != 0, out4[2] = (xstat[2] != 0 || xstat[0] > pair_cap), out4[3] = dcnt[9] != 0;
	}
}
