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
// Latency: a workgroup alternates between global memory and LDS some forty times, so the planes are software-pipelined -- the
// loads of plane p + 1 are issued before plane p is gathered, stores are never waited for (the barriers between the phases
// only order LDS: gs_bar), and a key plane is loaded once for all its radix passes (thread t keeps items t, t + 1024, ... in
// registers: K of them, the kernel's template parameter).
//
// LDS budget per workgroup for np items (np = largest genome of the shard, rounded up to 64):
//   [idx0: 2 np] [S: max(3 np + 16 KiB, 4 np)] [head / tie bit arrays: np / 4] [256]
//   S while sorting = {second index array 2 np, byte plane np, per-wave digit counters 16 x 256 x 4};  S while permuting = one
//   staged 32-bit plane.  160 KiB hold np = 25 600; a 10 k-hit bacterial genome needs 69 KiB (two workgroups per CU).
// Genomes beyond that (or shards whose score keys need 64 bits) take the multi-workgroup radix path of round 2 (pga_begin).
#pragma once

constexpr int GS_T = 1024, GS_NW = GS_T / WAVE, GS_HIST = GS_NW * 256;
constexpr int GS_K_SMALL = 14, GS_K_BIG = 25;        // items per thread of the two instantiations
constexpr int GS_NP_MAX = GS_K_BIG * GS_T;           // 25 600

static inline size_t gs_lds_bytes(int np)
{
	const size_t s = std::max<size_t>(3 * (size_t)np + sizeof(uint32_t) * GS_HIST, 4 * (size_t)np);
	return 2 * (size_t)np + s + (size_t)np / 4 + 256;
}

struct GenomeSort {
	const int32_t *up; int64_t N;  // file-order planes: plane f at up + f * N (k_unblock)
	const int32_t *goff, *ctg_base;
	int cs_bits, cm_bits, ctg_bits, np, n_genome;
	HitArrays o; int32_t *yperm, *headpos; int4 *A, *B, *C;
	long long *prof; // tuning aid (PANGENE_GS_PROF=1): 16 time stamps per workgroup
	const int32_t *glist; // k_segsort2.hpp: the genomes this launch sorts (workgroup b takes glist[b]); NULL = genome b
	// k_segsort2.hpp, round 6 -- CONTIG BINS: a workgroup's unit is a run of consecutive contigs of one genome instead of a whole genome, so that
	// genomes beyond what the LDS holds (a human assembly: 110 000 hits in a few hundred contigs) are still sorted there ("contig-segmented radix
	// sort", hit.c:37-53 buckets by contig first as well).  The planes in `up` are then grouped by contig (pga_create: stable, once per upload --
	// file order inside a contig stands, which is all the tie order needs), a unit is one contiguous range of them and of the X order, and
	// plane 17 holds each hit's file index.  bins[b] = {first position, hits, genome | bit 31 = the genome's first bin, first contig (local id)}.
	const int4 *bins;
	int y_fixup; // k_segsort2.hpp: the cm order by transpositions out of the cs order (gs2_body) -- 0: always by radix passes
};

struct GsLds { uint16_t *cur, *alt; uint8_t *dig; uint32_t *whist, *stage, *wtot; unsigned long long *head, *tie; int2 *wagg; };

#define GS_STAMP(k) do { if (a.prof && threadIdx.x == 0) a.prof[(long long)blockIdx.x * 32 + (k)] = (long long)__builtin_readcyclecounter(); } while (0)

// a workgroup barrier that orders LDS only: global loads and stores stay in flight across it (__syncthreads() drains them)
__device__ __forceinline__ void gs_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ uint32_t gs_block_excl(uint32_t v, uint32_t *wtot)
{
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	uint32_t incl = v;
#pragma unroll
	for (int d = 1; d < WAVE; d <<= 1) { const uint32_t u = __shfl_up(incl, d, WAVE); if (lane >= d) incl += u; }
	if (lane == 63) wtot[w] = incl;
	gs_bar();
	uint32_t carry = 0;
	for (int k = 0; k < w; ++k) carry += wtot[k];
	gs_bar();
	return carry + incl - v;
}

// Stable LSD radix passes over bits [0, bits) of the keys, one byte (or what is left) per pass.  key[u] = key of item
// tid + u * GS_T (registers).  The items are the 16-bit indices in L.cur (ping-pong with L.alt); wave w owns a contiguous span
// of the current order (at most K steps of 64), so "earlier wave, then earlier step, then earlier lane" is the input order and
// equal digits keep it.
// One pass = four phases with a barrier each, and NO dependent LDS round trip per step: (1) the byte plane is staged; (2) every
// step of a wave finds, with one ballot per bit, the lanes that share its digit; the first of them adds their number to the
// (wave, digit) counter and gets back the count of the wave's earlier steps (LDS atomics of one wave execute in program order;
// the returned value is not waited for); (3) the counters are scanned in (digit, wave) order; (4) position = scanned counter +
// earlier steps + lanes before me: one store per item.
template <int K>
__device__ __forceinline__ void gs_sort_bits(GsLds &L, const int n, const uint32_t (&key)[K], const int bits, long long *prof = nullptr)
{
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const int span = (((n + GS_NW - 1) / GS_NW) + 63) & ~63;
	const int lo = w * span, hi = lo + span < n ? lo + span : n;
	const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
	for (int shift = 0; shift < bits; shift += 8) {
		const int b = bits - shift < 8 ? bits - shift : 8;
		const uint32_t mask = (1u << b) - 1u;
#pragma unroll
		for (int u = 0; u < K; ++u) { const int i = tid + u * GS_T; if (i < n) L.dig[i] = (uint8_t)((key[u] >> shift) & mask); }
		for (int k = tid; k < GS_HIST; k += GS_T) L.whist[k] = 0;
		gs_bar();
		if (prof && shift == 0 && tid == 0) prof[16] = (long long)__builtin_readcyclecounter();
		uint32_t pk[K], ret[K]; // per step: item | lanes before me with my digit << 16 | first lane with my digit << 22; what that lane's atomic returned
		{
			uint32_t dg[K];
#pragma unroll
			for (int s = 0; s < K; ++s) { const int i = lo + s * WAVE + lane; pk[s] = i < hi ? L.cur[i] : 0u; }
#pragma unroll
			for (int s = 0; s < K; ++s) dg[s] = L.dig[pk[s]];
#pragma unroll
			for (int s = 0; s < K; ++s) {
				ret[s] = 0;
				if (lo + s * WAVE >= hi) continue; // wave-uniform
				const bool v = lo + s * WAVE + lane < hi;
				const uint32_t d = dg[s];
				unsigned long long peers = __ballot(v);
				for (int bb = 0; bb < b; ++bb) {
					const bool bit = (d >> bb) & 1u;
					const unsigned long long bal = __ballot(bit);
					peers &= bit ? bal : ~bal;
				}
				const uint32_t r = (uint32_t)__popcll(peers & lt), ldr = v ? (uint32_t)__ffsll((long long)peers) - 1u : (uint32_t)lane;
				if (v && r == 0) ret[s] = atomicAdd(&L.whist[w * 256 + d], (uint32_t)__popcll(peers));
				pk[s] |= r << 16 | ldr << 22;
			}
		}
		if (prof && shift == 0 && tid == 0) prof[17] = (long long)__builtin_readcyclecounter();
		gs_bar();
		if (prof && shift == 0 && tid == 0) prof[18] = (long long)__builtin_readcyclecounter();
		{ // exclusive scan of the counters in (digit, wave) order: thread t = digit t / 4, waves 4 (t % 4) ...
			const int d = tid >> 2, w0 = (tid & 3) * 4;
			const uint32_t c0 = L.whist[(w0 + 0) * 256 + d], c1 = L.whist[(w0 + 1) * 256 + d], c2 = L.whist[(w0 + 2) * 256 + d], c3 = L.whist[(w0 + 3) * 256 + d];
			const uint32_t ex = gs_block_excl(c0 + c1 + c2 + c3, L.wtot);
			L.whist[(w0 + 0) * 256 + d] = ex, L.whist[(w0 + 1) * 256 + d] = ex + c0, L.whist[(w0 + 2) * 256 + d] = ex + c0 + c1, L.whist[(w0 + 3) * 256 + d] = ex + c0 + c1 + c2;
		}
		gs_bar();
		if (prof && shift == 0 && tid == 0) prof[19] = (long long)__builtin_readcyclecounter();
		{
			uint32_t pos[K];
#pragma unroll
			for (int s = 0; s < K; ++s) { // (every lane takes part in the shuffles)
				const uint32_t within = (uint32_t)__shfl((int)ret[s], (int)((pk[s] >> 22) & 63u), WAVE);
				pos[s] = within + ((pk[s] >> 16) & 63u) + L.whist[w * 256 + L.dig[pk[s] & 0xffffu]];
			}
#pragma unroll
			for (int s = 0; s < K; ++s) if (lo + s * WAVE + lane < hi) L.alt[pos[s]] = (uint16_t)(pk[s] & 0xffffu);
		}
		if (prof && shift == 0 && tid == 0) prof[20] = (long long)__builtin_readcyclecounter();
		gs_bar();
		if (prof && shift == 0 && tid == 0) prof[21] = (long long)__builtin_readcyclecounter();
		uint16_t *t = L.cur; L.cur = L.alt; L.alt = t;
	}
}

template <int K, int D>
__device__ __forceinline__ void gs_body(const GenomeSort &a, unsigned char *gs_mem)
{
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
	// the file-order planes in the order they are permuted (plane f of the shard at up + f * N: 0 pid, 1 contig, 2 rank, 3 score_ori,
	// 4 score_adj, 5 n_exon, 6 off_exon, 7 cs, 8 ce, 9 cm; static per-hit constants derived once in pga_create: 12 gene, 13 CDS
	// length, 15 score key, 16 rev / multi-exon flag bits)
	constexpr int NPL = 15;
	const int32_t *const pl[NPL] = { a.up + N + gb, a.up + 7 * N + gb, a.up + 8 * N + gb, a.up + gb, a.up + 12 * N + gb, a.up + 15 * N + gb, a.up + 13 * N + gb, a.up + 2 * N + gb,
	                                 a.up + 3 * N + gb, a.up + 4 * N + gb, a.up + 5 * N + gb, a.up + 6 * N + gb, a.up + 16 * N + gb, a.up + 9 * N + gb, a.up + N + gb };
	const int cb = a.ctg_base[g];
	uint32_t R[D][K]; // D planes in registers (loaded D planes ahead of their use): element u belongs to item tid + u * GS_T
	GS_STAMP(0);

#define GS_LOADP(q) do { if ((q) < NPL) { _Pragma("unroll") for (int u = 0; u < K; ++u) { const int i = tid + u * GS_T; R[(q) % D][u] = i < n ? (uint32_t)pl[(q) < NPL ? (q) : 0][i] : 0u; } } } while (0)
	// ---- X order: pg_hit_sort(g, 0) = by (contig, cs), ties in file order (the reference's own tie order is replayed later where it matters) ----
	if (D >= 2) { GS_LOADP(0); GS_LOADP(1); } else GS_LOADP(1); // (one register set: cs first, the contig plane after its passes)
	if (D > 2) GS_LOADP(2);
#pragma unroll
	for (int u = 0; u < K; ++u) { const int i = tid + u * GS_T; if (i < n) L.cur[i] = (uint16_t)i; }
	gs_bar();
	gs_sort_bits<K>(L, n, R[1 % D], a.cs_bits, a.prof ? a.prof + (long long)blockIdx.x * 32 : nullptr);
	GS_STAMP(1);
	if (D < 2) GS_LOADP(0);
	gs_sort_bits<K>(L, n, R[0], a.ctg_bits);
	GS_STAMP(2);
	if (L.cur != idx0) { // the permutation phase wants the order in the first array (S becomes the staging area)
		uint32_t t[K];
#pragma unroll
		for (int u = 0; u < K; ++u) { const int i = tid + u * GS_T; t[u] = i < n ? L.cur[i] : 0u; }
#pragma unroll
		for (int u = 0; u < K; ++u) { const int i = tid + u * GS_T; if (i < n) idx0[i] = (uint16_t)t[u]; }
		gs_bar();
	}
	uint32_t *const st = L.stage;
	uint32_t V[K]; // the plane being written, in X order

	// ---- every plane through LDS: coalesced read in file order (issued D planes ahead), gather in LDS, coalesced write in X order ----
	// GS_BEGIN(p): plane p goes from its registers into the staging area, the registers are refilled with plane p + D
#define GS_BEGIN(p) do { _Pragma("unroll") for (int u = 0; u < K; ++u) { const int i = tid + u * GS_T; if (i < n) st[i] = R[(p) % D][u]; } gs_bar(); GS_LOADP((p) + D); } while (0)
#define GS_GET() do { uint32_t j_[K]; _Pragma("unroll") for (int u = 0; u < K; ++u) { const int x = tid + u * GS_T; j_[u] = x < n ? idx0[x] : 0u; } \
		_Pragma("unroll") for (int u = 0; u < K; ++u) V[u] = st[j_[u]]; } while (0)
#define GS_OUT(dst) do { _Pragma("unroll") for (int u = 0; u < K; ++u) { const int x = tid + u * GS_T; if (x < n) (dst)[gb + x] = (int32_t)V[u]; } } while (0)
#define GS_PLANE(p, dst) do { GS_BEGIN(p); GS_GET(); GS_OUT(dst); gs_bar(); } while (0)
	// plane 0, contig: segment ids, and where a contig starts in X order (bit array)
	GS_BEGIN(0);
	GS_GET();
#pragma unroll
	for (int u = 0; u < K; ++u) {
		const int x = tid + u * GS_T;
		if (x - lane >= n) break; // wave-uniform
		const bool v = x < n;
		const uint32_t cp = (v && x > 0) ? st[idx0[x - 1]] : ~0u;
		if (v) a.o.seg[gb + x] = cb + (int32_t)V[u];
		const unsigned long long hb = __ballot(v && V[u] != cp);
		if (lane == 0) L.head[x >> 6] = hb;
	}
	gs_bar();
	GS_STAMP(3);
	// plane 1, cs: the static marks of the cs sort's tie groups (hazard H2b, see k_rep_fill); the value itself only lives in record A
	GS_BEGIN(1);
	GS_GET();
	uint32_t CSX[K];
#pragma unroll
	for (int u = 0; u < K; ++u) {
		const int x = tid + u * GS_T;
		CSX[u] = V[u];
		if (x - lane >= n) continue;
		const bool v = x < n;
		bool tie = false;
		if (v) {
			const bool hd = (L.head[x >> 6] >> (x & 63)) & 1ull, hn = x + 1 < n ? (bool)((L.head[(x + 1) >> 6] >> ((x + 1) & 63)) & 1ull) : true;
			tie = (!hd && st[idx0[x - 1]] == V[u]) || (!hn && st[idx0[x + 1]] == V[u]);
		}
		const unsigned long long tb = __ballot(tie);
		if (lane == 0) L.tie[x >> 6] = tb;
	}
	gs_bar();
	GS_STAMP(4);
	// plane 2, ce, and pm = its running maximum inside a contig (bounds the sweep's look-back): a segmented inclusive max scan over
	// the X order, wave w over a contiguous span, two sweeps (span aggregates, then the scan proper with the carry of the spans
	// before); the result goes back into the staging area, from where every thread takes its own positions for record A
	GS_BEGIN(2);
	GS_GET(); // V = ce in X order
	gs_bar();
	{
		const int span = (((n + GS_NW - 1) / GS_NW) + 63) & ~63;
		const int lo = w * span, hi = lo + span < n ? lo + span : n;
		for (int sweep = 0; sweep < 2; ++sweep) {
			int cv = INT32_MIN, cf = 0;
			if (sweep) for (int k = 0; k < w; ++k) { const int2 q = L.wagg[k]; cv = q.x ? q.y : (cv > q.y ? cv : q.y); }
			for (int j0 = lo; j0 < hi; j0 += WAVE) {
				const int x = j0 + lane;
				const bool v = x < hi;
				const uint32_t jx = v ? idx0[x] : 0u;
				int e = v ? (int)st[jx] : INT32_MIN, f = v ? (int)((L.head[x >> 6] >> (x & 63)) & 1ull) : 0;
#pragma unroll
				for (int d = 1; d < WAVE; d <<= 1) {
					const int ue = __shfl_up(e, d, WAVE), uf = __shfl_up(f, d, WAVE);
					if (lane >= d) { if (!f) e = e > ue ? e : ue; f |= uf; }
				}
				if (!f) e = e > cv ? e : cv;
				cv = __shfl(e, 63, WAVE), cf |= __shfl(f, 63, WAVE);
				if (sweep) { wave_sync(); if (v) st[jx] = (uint32_t)e; } // (each staging slot is read and rewritten by the same lane)
			}
			if (!sweep) {
				if (lane == 0) L.wagg[w] = make_int2(cf, cv);
				gs_bar();
			}
		}
		gs_bar();
	}
	{ // record A = {cs, seg, ce, pm} (k_sweep.hpp)
		uint32_t PM[K];
		{ uint32_t j_[K];
#pragma unroll
		  for (int u = 0; u < K; ++u) { const int x = tid + u * GS_T; j_[u] = x < n ? idx0[x] : 0u; }
#pragma unroll
		  for (int u = 0; u < K; ++u) PM[u] = st[j_[u]]; }
#pragma unroll
		for (int u = 0; u < K; ++u) {
			const int x = tid + u * GS_T;
			if (x < n) a.A[gb + x] = make_int4((int)CSX[u], a.o.seg[gb + x], (int)V[u], (int)PM[u]);
		}
	}
	gs_bar();
	GS_STAMP(5);
	GS_PLANE(3, a.o.pid);
	GS_PLANE(4, a.o.gid);
	GS_BEGIN(5); GS_GET(); gs_bar(); // score key: only in record B
#pragma unroll
	for (int u = 0; u < K; ++u) CSX[u] = V[u];
	GS_BEGIN(6); GS_GET(); gs_bar(); // CDS length: only in record B
#pragma unroll
	for (int u = 0; u < K; ++u) { // record B = {rk, gid, cds, pid}: gid and pid come back out of their planes at the end
		const int x = tid + u * GS_T;
		if (x < n) a.B[gb + x] = make_int4((int)CSX[u], 0, (int)V[u], 0);
	}
	GS_PLANE(7, a.o.rank);
	GS_PLANE(8, a.o.sori); // (also a plane: the single-exon flavour of the sweep stages it alone)
	GS_PLANE(9, a.o.sadj);
	GS_PLANE(10, a.o.nex);
	GS_BEGIN(11); GS_GET(); gs_bar(); // off_exon: only in record C = {rank, n_exon, off_exon, score_ori}; the other three come back out of their planes
#pragma unroll
	for (int u = 0; u < K; ++u) {
		const int x = tid + u * GS_T;
		if (x < n) ((int32_t *)&a.C[gb + x])[2] = (int32_t)V[u];
	}
	GS_STAMP(6);
	GS_BEGIN(12);
	GS_GET();
#pragma unroll
	for (int u = 0; u < K; ++u) {
		const int x = tid + u * GS_T;
		if (x >= n) break;
		a.o.flags[gb + x] = V[u] | (x == 0 ? F_HEAD : 0u) | (((L.tie[x >> 6] >> (x & 63)) & 1ull) ? F_CSTIE : 0u);
		a.o.fidx[gb + x] = (int32_t)idx0[x], a.o.gnm[gb + x] = g;
	}
	gs_bar();
	GS_STAMP(7);
	// ---- Y order: pg_hit_sort(g, 1) = by (contig, cm), ties in X order; the items are X positions now, the keys cm and contig in X order ----
	GS_PLANE(13, a.o.cm);
#pragma unroll
	for (int u = 0; u < K; ++u) CSX[u] = V[u];
	GS_BEGIN(14); GS_GET(); gs_bar();
#undef GS_LOADP
#undef GS_BEGIN
#undef GS_GET
#undef GS_OUT
#undef GS_PLANE
	GS_STAMP(8);
	L.cur = idx0, L.alt = (uint16_t *)S;
#pragma unroll
	for (int u = 0; u < K; ++u) { const int x = tid + u * GS_T; if (x < n) L.cur[x] = (uint16_t)x; }
	gs_bar();
	gs_sort_bits<K>(L, n, CSX, a.cm_bits);
	GS_STAMP(9);
	gs_sort_bits<K>(L, n, V, a.ctg_bits);
	GS_STAMP(10);
#pragma unroll
	for (int u = 0; u < K; ++u) { const int y = tid + u * GS_T; if (y < n) a.yperm[gb + y] = gb + (int32_t)L.cur[y]; }
	__syncthreads(); // everything this workgroup wrote is in L2 now (this barrier waits for the stores)
	GS_STAMP(11);
	// the words of records B and C that also exist as planes (gid, pid; rank, n_exon, score_ori) are filled in from there
	for (int x0 = tid; x0 < n; x0 += 4 * GS_T) {
		int32_t q[4][5];
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const int h = gb + (x0 + u * GS_T < n ? x0 + u * GS_T : x0);
			q[u][0] = a.o.gid[h], q[u][1] = a.o.pid[h], q[u][2] = a.o.rank[h], q[u][3] = a.o.nex[h], q[u][4] = a.o.sori[h];
		}
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			if (x0 + u * GS_T >= n) break;
			const int h = gb + x0 + u * GS_T;
			((int32_t *)&a.B[h])[1] = q[u][0], ((int32_t *)&a.B[h])[3] = q[u][1];
			((int32_t *)&a.C[h])[0] = q[u][2], ((int32_t *)&a.C[h])[1] = q[u][3], ((int32_t *)&a.C[h])[3] = q[u][4];
		}
	}
	GS_STAMP(12);
}

// two instantiations: up to 14 items per thread with the loads two planes ahead, up to 25 with the loads one plane ahead (registers)
__global__ __launch_bounds__(GS_T, 4) void k_genome_sort(GenomeSort a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char gs_mem[];
	gs_body<GS_K_SMALL, 2>(a, gs_mem);
}
__global__ __launch_bounds__(GS_T, 4) void k_genome_sort_big(GenomeSort a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char gs_mem_big[];
	gs_body<GS_K_BIG, 1>(a, gs_mem_big);
}
