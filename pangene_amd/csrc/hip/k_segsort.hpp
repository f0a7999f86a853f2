// k_segsort.hpp -- pg_hit_sort (hit.c:29-64) for both orders, and every per-hit constant of stage A, in ONE launch:
// one workgroup per genome, the genome's sort keys resident in LDS ("contig-segmented radix sort": a genome's hits are one
// contiguous block of the file order, and inside it the order is (contig, cs) resp. (contig, cm)).
// Included by pga_backend.hip (one translation unit); uses the context types, BLOCK / WAVE and dev_prims.hpp from there.
//
// What it replaces (the round-2 form of pga_begin): k_prepare, k_xkey, 5 x (rs_hist, rs_rowscan, rs_scatter), k_gather,
// k_inv_only, the segmented max scan (3 launches), k_pack_rec, k_ykey, 5 x (rs_*) and two copies -- ~40 launches that moved
// every key 13 times through HBM.  Here a genome's hits are read once (coalesced, file order), sorted as 16-bit indices in
// LDS with a stable LSD radix sort (one byte of the key per pass, the byte plane staged in LDS; 16 waves, in-wave ranks by
// ballots), permuted THROUGH LDS (plane by plane: coalesced read, LDS gather, coalesced write) and written once.
//
// LDS budget per workgroup for np items (np = largest genome of the shard, rounded up to 64):
//   [idx0: 2 np] [S: max(3 np + 16 KiB, 4 np)] [head / tie bit arrays: np / 4] [256]
//   S while sorting = {second index array 2 np, byte plane np, per-wave digit counters 16 x 256 x 4};  S while permuting = one
//   staged 32-bit plane.  160 KiB hold np = 25 600; a 10 k-hit bacterial genome needs 69 KiB (two workgroups per CU).
// Genomes beyond that (or shards whose score keys need 64 bits) take the multi-workgroup radix path of round 2 (pga_begin).
#pragma once

constexpr int GS_T = 1024, GS_NW = GS_T / WAVE, GS_HIST = GS_NW * 256;
constexpr int GS_NP_MAX = 25600;

static inline size_t gs_lds_bytes(int np)
{
	const size_t s = std::max<size_t>(3 * (size_t)np + sizeof(uint32_t) * GS_HIST, 4 * (size_t)np);
	return 2 * (size_t)np + s + (size_t)np / 4 + 256;
}

struct GenomeSort {
	const int32_t *up; int64_t N;  // file-order planes: plane f at up + f * N (k_unblock)
	const int32_t *goff, *ctg_base; const int2 *exon; const int32_t *prot_gid; const uint8_t *gene_pref; const int32_t *hrank;
	int rk_shift, cs_bits, cm_bits, ctg_bits, np, n_genome;
	HitArrays o; int32_t *pm, *inv, *yperm, *headpos; int4 *A, *B, *C;
};

struct GsLds { uint16_t *cur, *alt; uint8_t *dig; uint32_t *whist, *stage, *wtot; unsigned long long *head, *tie; int2 *wagg; };

__device__ __forceinline__ uint32_t gs_block_excl(uint32_t v, uint32_t *wtot)
{
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	uint32_t incl = v;
#pragma unroll
	for (int d = 1; d < WAVE; d <<= 1) { const uint32_t u = __shfl_up(incl, d, WAVE); if (lane >= d) incl += u; }
	if (lane == 63) wtot[w] = incl;
	__syncthreads();
	uint32_t carry = 0;
	for (int k = 0; k < w; ++k) carry += wtot[k];
	__syncthreads();
	return carry + incl - v;
}

// Stable LSD radix passes over bits [0, bits) of plane[item] - sub, one byte (or what is left) per pass.  The items are the
// 16-bit indices in L.cur (ping-pong with L.alt); wave w owns a contiguous span of the current order, so "earlier wave, then
// earlier lane" is the input order and equal digits keep it.
__device__ __forceinline__ void gs_sort_bits(GsLds &L, const int n, const int32_t *plane, const int sub, const int bits)
{
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const int span = (((n + GS_NW - 1) / GS_NW) + 63) & ~63;
	const int lo = w * span, hi = lo + span < n ? lo + span : n;
	const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
	for (int shift = 0; shift < bits; shift += 8) {
		const int b = bits - shift < 8 ? bits - shift : 8;
		const uint32_t mask = (1u << b) - 1u;
		for (int i = tid; i < n; i += GS_T) L.dig[i] = (uint8_t)(((uint32_t)(plane[i] - sub) >> shift) & mask);
		for (int k = tid; k < GS_HIST; k += GS_T) L.whist[k] = 0;
		__syncthreads();
		for (int i = lo + lane; i < hi; i += WAVE) atomicAdd(&L.whist[w * 256 + L.dig[L.cur[i]]], 1u);
		__syncthreads();
		{ // exclusive scan of the counters in (digit, wave) order: thread t = digit t / 4, waves 4 (t % 4) ...
			const int d = tid >> 2, w0 = (tid & 3) * 4;
			const uint32_t c0 = L.whist[(w0 + 0) * 256 + d], c1 = L.whist[(w0 + 1) * 256 + d], c2 = L.whist[(w0 + 2) * 256 + d], c3 = L.whist[(w0 + 3) * 256 + d];
			const uint32_t ex = gs_block_excl(c0 + c1 + c2 + c3, L.wtot);
			L.whist[(w0 + 0) * 256 + d] = ex, L.whist[(w0 + 1) * 256 + d] = ex + c0, L.whist[(w0 + 2) * 256 + d] = ex + c0 + c1, L.whist[(w0 + 3) * 256 + d] = ex + c0 + c1 + c2;
		}
		__syncthreads();
		for (int j0 = lo; j0 < hi; j0 += WAVE) {
			const int i = j0 + lane;
			const bool v = i < hi;
			const uint32_t item = v ? L.cur[i] : 0u, d = v ? L.dig[item] : 0u;
			unsigned long long peers = __ballot(v);
			for (int bb = 0; bb < b; ++bb) {
				const bool bit = (d >> bb) & 1u;
				const unsigned long long bal = __ballot(bit);
				peers &= bit ? bal : ~bal;
			}
			const uint32_t base = v ? L.whist[w * 256 + d] : 0u;
			const int r = __popcll(peers & lt);
			wave_sync();
			if (v) {
				L.alt[base + r] = (uint16_t)item;
				if (r == 0) L.whist[w * 256 + d] = base + (uint32_t)__popcll(peers);
			}
			wave_sync();
		}
		__syncthreads();
		uint16_t *t = L.cur; L.cur = L.alt; L.alt = t;
	}
}

__global__ __launch_bounds__(GS_T, 8) void k_genome_sort(GenomeSort a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char gs_mem[];
	const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const int gb = a.goff[g], n = a.goff[g + 1] - gb, np = a.np;
	if (tid == 0) { a.headpos[g] = gb; if (g == a.n_genome - 1) a.headpos[g + 1] = gb + n; }
	if (n == 0) return;
	const size_t s_bytes = 3 * (size_t)np + sizeof(uint32_t) * GS_HIST > 4 * (size_t)np ? 3 * (size_t)np + sizeof(uint32_t) * GS_HIST : 4 * (size_t)np;
	uint16_t *const idx0 = (uint16_t *)gs_mem;
	unsigned char *const S = gs_mem + 2 * (size_t)np;
	GsLds L;
	L.cur = idx0, L.alt = (uint16_t *)S, L.dig = S + 2 * (size_t)np, L.whist = (uint32_t *)(S + 3 * (size_t)np), L.stage = (uint32_t *)S;
	L.head = (unsigned long long *)(S + s_bytes), L.tie = L.head + np / 64;
	L.wtot = (uint32_t *)(L.tie + np / 64), L.wagg = (int2 *)(L.wtot + GS_NW);
	const int64_t N = a.N;
	const int32_t *f_pid = a.up + gb, *f_cid = a.up + N + gb, *f_rank = a.up + 2 * N + gb, *f_sori = a.up + 3 * N + gb, *f_sadj = a.up + 4 * N + gb, *f_nex = a.up + 5 * N + gb,
		*f_offx = a.up + 6 * N + gb, *f_cs = a.up + 7 * N + gb, *f_ce = a.up + 8 * N + gb, *f_cm = a.up + 9 * N + gb;
	const uint8_t *f_rev = (const uint8_t *)(a.up + 14 * N) + gb;
	const int cb = a.ctg_base[g];

	// ---- X order: pg_hit_sort(g, 0) = by (contig, cs), ties in file order (the reference's own tie order is replayed later where it matters) ----
	for (int i = tid; i < n; i += GS_T) L.cur[i] = (uint16_t)i;
	__syncthreads();
	gs_sort_bits(L, n, f_cs, 0, a.cs_bits);
	gs_sort_bits(L, n, f_cid, 0, a.ctg_bits);
	if (L.cur != idx0) { // the permutation phase wants the order in the first array (S becomes the staging area)
		for (int i = tid; i < n; i += GS_T) idx0[i] = L.cur[i];
		__syncthreads();
	}
	const uint16_t *const ix = idx0;
	uint32_t *const st = L.stage;

	// ---- every plane through LDS: coalesced read in file order, gather in LDS, coalesced write in X order ----
#define GS_STAGE(expr) do { for (int i = tid; i < n; i += GS_T) st[i] = (uint32_t)(expr); __syncthreads(); } while (0)
#define GS_EMIT(dst) do { for (int x = tid; x < n; x += GS_T) (dst)[gb + x] = (int32_t)st[ix[x]]; __syncthreads(); } while (0)
	// contig: segment ids, and where a contig starts in X order (bit array)
	GS_STAGE(f_cid[i]);
	for (int x0 = w * WAVE; x0 < n; x0 += GS_T) {
		const int x = x0 + lane;
		const bool v = x < n;
		const uint32_t cx = v ? st[ix[x]] : 0u, cp = (v && x > 0) ? st[ix[x - 1]] : ~0u;
		if (v) a.o.seg[gb + x] = cb + (int32_t)cx;
		const unsigned long long hb = __ballot(v && cx != cp);
		if (lane == 0) L.head[x0 >> 6] = hb;
	}
	__syncthreads();
	// cs, and the static marks of the cs sort's tie groups (hazard H2b, see k_rep_fill)
	GS_STAGE(f_cs[i]);
	for (int x0 = w * WAVE; x0 < n; x0 += GS_T) {
		const int x = x0 + lane;
		const bool v = x < n;
		bool tie = false;
		if (v) {
			const uint32_t c0 = st[ix[x]];
			a.o.cs[gb + x] = (int32_t)c0;
			const bool hd = (L.head[x >> 6] >> (x & 63)) & 1ull, hn = x + 1 < n ? ((L.head[(x + 1) >> 6] >> ((x + 1) & 63)) & 1ull) : true;
			tie = (!hd && st[ix[x - 1]] == c0) || (!hn && st[ix[x + 1]] == c0);
		}
		const unsigned long long tb = __ballot(tie);
		if (lane == 0) L.tie[x0 >> 6] = tb;
	}
	__syncthreads();
	// ce, and pm = its running maximum inside a contig (bounds the sweep's look-back): a segmented inclusive max scan over the
	// X order, wave w over a contiguous span, two sweeps (span aggregates, then the scan proper with the carry of the spans before)
	GS_STAGE(f_ce[i]);
	{
		const int span = (((n + GS_NW - 1) / GS_NW) + 63) & ~63;
		const int lo = w * span, hi = lo + span < n ? lo + span : n;
		for (int sweep = 0; sweep < 2; ++sweep) {
			int cv = INT32_MIN, cf = 0;
			if (sweep) for (int k = 0; k < w; ++k) { const int2 q = L.wagg[k]; cv = q.x ? q.y : (cv > q.y ? cv : q.y); }
			for (int j0 = lo; j0 < hi; j0 += WAVE) {
				const int x = j0 + lane;
				const bool v = x < hi;
				const int e0 = v ? (int)st[ix[x]] : INT32_MIN;
				int e = e0, f = v ? (int)((L.head[x >> 6] >> (x & 63)) & 1ull) : 0;
#pragma unroll
				for (int d = 1; d < WAVE; d <<= 1) {
					const int ue = __shfl_up(e, d, WAVE), uf = __shfl_up(f, d, WAVE);
					if (lane >= d) { if (!f) e = e > ue ? e : ue; f |= uf; }
				}
				if (!f) e = e > cv ? e : cv;
				if (sweep && v) a.o.ce[gb + x] = e0, a.pm[gb + x] = e;
				cv = __shfl(e, 63, WAVE), cf |= __shfl(f, 63, WAVE);
			}
			if (!sweep) {
				if (lane == 0) L.wagg[w] = make_int2(cf, cv);
				__syncthreads();
			}
		}
		__syncthreads();
	}
	GS_STAGE(f_cm[i]); GS_EMIT(a.o.cm);
	GS_STAGE(f_pid[i]); GS_EMIT(a.o.pid);
	GS_STAGE(a.prot_gid[f_pid[i]]); GS_EMIT(a.o.gid);
	// the comparison key of overlap.c:137 in 32 bits (see pga_ctx::rk_shift)
	GS_STAGE((uint32_t)f_sadj[i] << a.rk_shift | (uint32_t)a.gene_pref[a.prot_gid[f_pid[i]]] << (a.rk_shift - 1) | (uint32_t)a.hrank[f_pid[i]]); GS_EMIT(a.o.rk);
	GS_STAGE(f_rank[i]); GS_EMIT(a.o.rank);
	GS_STAGE(f_sori[i]); GS_EMIT(a.o.sori);
	GS_STAGE(f_sadj[i]); GS_EMIT(a.o.sadj);
	GS_STAGE(f_nex[i]); GS_EMIT(a.o.nex);
	GS_STAGE(f_offx[i]); GS_EMIT(a.o.offx);
	for (int i = tid; i < n; i += GS_T) { // pg_cds_len, overlap.c:45-51
		const int ne = f_nex[i], ox = f_offx[i];
		int len = 0;
		for (int e = 0; e < ne; ++e) { const int2 q = a.exon[ox + e]; len += q.y - q.x; }
		st[i] = (uint32_t)len;
	}
	__syncthreads();
	GS_EMIT(a.o.cds);
	GS_STAGE((f_rev[i] ? PGA_F_REV : 0u) | (f_nex[i] != 1 ? F_MULTI : 0u));
	for (int x = tid; x < n; x += GS_T) {
		a.o.flags[gb + x] = st[ix[x]] | (x == 0 ? F_HEAD : 0u) | (((L.tie[x >> 6] >> (x & 63)) & 1ull) ? F_CSTIE : 0u);
		a.o.fidx[gb + x] = ix[x], a.o.gnm[gb + x] = g;
		a.o.sdom[gb + x] = 0, a.o.pdom[gb + x] = -1, a.o.pdom0[gb + x] = 0; // read.c:133-134
	}
	__syncthreads();
	for (int x = tid; x < n; x += GS_T) st[ix[x]] = (uint32_t)x; // file index -> X position
	__syncthreads();
	for (int i = tid; i < n; i += GS_T) a.inv[gb + i] = gb + (int32_t)st[i];
	__syncthreads(); // everything this workgroup wrote is in L2 now (the barrier waits for the stores)
#undef GS_STAGE
#undef GS_EMIT
	// the packed sweep records (k_sweep.hpp): A = {cs, seg, ce, pm}  B = {rk, gid, cds, pid}  C = {rank, n_exon, off_exon, score_ori}
	for (int x = tid; x < n; x += GS_T) {
		const int h = gb + x;
		a.A[h] = make_int4(a.o.cs[h], a.o.seg[h], a.o.ce[h], a.pm[h]);
		a.B[h] = make_int4(a.o.rk[h], a.o.gid[h], a.o.cds[h], a.o.pid[h]);
		a.C[h] = make_int4(a.o.rank[h], a.o.nex[h], a.o.offx[h], a.o.sori[h]);
	}
	// ---- Y order: pg_hit_sort(g, 1) = by (contig, cm), ties in X order; the items are X positions now ----
	L.cur = idx0, L.alt = (uint16_t *)S;
	for (int x = tid; x < n; x += GS_T) L.cur[x] = (uint16_t)x;
	__syncthreads();
	gs_sort_bits(L, n, a.o.cm + gb, 0, a.cm_bits);
	gs_sort_bits(L, n, a.o.seg + gb, cb, a.ctg_bits);
	for (int y = tid; y < n; y += GS_T) a.yperm[gb + y] = gb + (int32_t)L.cur[y];
}
