// k_segsort2.hpp -- the round-4 form of k_genome_sort (k_segsort.hpp holds the round-3 form, still used for genomes of more than
// GS2_NP_MAX hits): pg_hit_sort (hit.c:29-64) for both orders and every per-hit constant of stage A, one workgroup per genome,
// the genome's sort keys resident in LDS.
//
// What changed, and why.  The round-3 kernel needs 126 VGPRs: with 1024 threads that is ONE workgroup per CU (4 waves per SIMD is
// all 512 registers allow), so every one of its ~60 barrier phases idles the whole CU -- it ran at ~2.4 TB/s of traffic, latency
// bound.  This form is built for 64 VGPRs -- two workgroups (32 waves) per CU, each covering the other's barriers and loads:
//   * ten items per thread (np <= 10 240: the bacterial shape), and never more than four ten-element arrays live;
//   * (contig, cs) and (contig, cm) sorted as ONE composite key when contig and coordinate bits fit 32 (the contig pass of the
//     round-3 form was a full pass for one bit): 3 + 3 radix passes at 5-6 Mb genomes instead of 4 + 4;
//   * the 16-byte records A / B / C are assembled in registers over four consecutive plane phases and written whole -- the round-3
//     form wrote B and C partly and patched the other words in with a last pass of 4-byte stores into 16-byte records.
// Layout of the LDS (np items): [idx0: 2 np] [S: max(3 np + 16 KiB, 4 np)] [head / tie bit arrays: np / 4] [256].
#pragma once

constexpr int GS2_T = 1024, GS2_K = 10, GS2_NP_MAX = GS2_K * GS2_T; // 10 240
constexpr int GS2_K_BIG = 14, GS2_NP_BIG = GS2_K_BIG * GS2_T;        // 14 336
constexpr int GS2_FIX_IT = 6;                                        // double passes of the cm order's transposition fix-up before the radix passes take over

template <int T, int K>
__device__ __forceinline__ void gs2_sort_bits(GsLds &L, const int n, const uint32_t (&key)[K], const int bits)
{
	constexpr int NW = T / WAVE, HIST = NW * 256;
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const int span = (((n + NW - 1) / NW) + 63) & ~63;
	const int lo = w * span, hi = lo + span < n ? lo + span : n;
	const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
	for (int shift = 0; shift < bits; shift += 8) {
		const int b = bits - shift < 8 ? bits - shift : 8;
		const uint32_t mask = (1u << b) - 1u;
#pragma unroll
		for (int u = 0; u < K; ++u) { const int i = tid + u * T; if (i < n) L.dig[i] = (uint8_t)((key[u] >> shift) & mask); }
		for (int k = tid; k < HIST; k += T) L.whist[k] = 0;
		gs_bar();
		uint32_t pk[K], ret[K]; // per step: item | lanes before me with my digit << 16 | first lane with my digit << 22; what that lane's atomic returned
#pragma unroll
		for (int s = 0; s < K; ++s) { const int i = lo + s * WAVE + lane; pk[s] = i < hi ? L.cur[i] : 0u; }
#pragma unroll
		for (int s = 0; s < K; ++s) {
			ret[s] = 0;
			if (lo + s * WAVE >= hi) continue; // wave-uniform
			const bool v = lo + s * WAVE + lane < hi;
			const uint32_t d = L.dig[pk[s]];
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
		gs_bar();
		{ // exclusive scan of the counters in (digit, wave) order: thread t = digit t / G, waves 4 (t % G) ... (G = T / 256 groups of four waves)
			constexpr int G = T / 256;
			const int d = tid / G, w0 = (tid % G) * 4;
			const uint32_t c0 = L.whist[(w0 + 0) * 256 + d], c1 = L.whist[(w0 + 1) * 256 + d], c2 = L.whist[(w0 + 2) * 256 + d], c3 = L.whist[(w0 + 3) * 256 + d];
			const uint32_t ex = gs_block_excl(c0 + c1 + c2 + c3, L.wtot);
			L.whist[(w0 + 0) * 256 + d] = ex, L.whist[(w0 + 1) * 256 + d] = ex + c0, L.whist[(w0 + 2) * 256 + d] = ex + c0 + c1, L.whist[(w0 + 3) * 256 + d] = ex + c0 + c1 + c2;
		}
		gs_bar();
#pragma unroll
		for (int s = 0; s < K; ++s) { // (every lane takes part in the shuffles)
			const uint32_t within = (uint32_t)__shfl((int)ret[s], (int)((pk[s] >> 22) & 63u), WAVE);
			const uint32_t pos = within + ((pk[s] >> 16) & 63u) + L.whist[w * 256 + L.dig[pk[s] & 0xffffu]];
			if (lo + s * WAVE + lane < hi) L.alt[pos] = (uint16_t)(pk[s] & 0xffffu);
		}
		gs_bar();
		uint16_t *t = L.cur; L.cur = L.alt; L.alt = t;
	}
}

static inline size_t gs2_lds_bytes(int np)
{
	const size_t s = std::max<size_t>(3 * (size_t)np + sizeof(uint32_t) * (size_t)(GS2_T / WAVE) * 256, 4 * (size_t)np);
	return 2 * (size_t)np + s + (size_t)np / 4 + 256;
}

// the transposition passes of the cm order (gs2_body): true = a double pass moved nothing, the order stands
template <int T>
__device__ __forceinline__ bool gs2_fixup(uint32_t *yk, uint16_t *cur, volatile uint32_t *flag, const int n)
{
	const int tid = threadIdx.x;
	bool settled = false;
	for (int it = 0; it < GS2_FIX_IT && !settled; ++it) {
		bool moved = false;
		for (int ph = 0; ph < 2; ++ph) {
			for (int i = 2 * tid + ph; i + 1 < n; i += 2 * T) {
				const uint32_t ka = yk[i], kb = yk[i + 1];
				if (ka > kb) { const uint16_t ia = cur[i], ib = cur[i + 1]; yk[i] = kb, yk[i + 1] = ka, cur[i] = ib, cur[i + 1] = ia, moved = true; }
			}
			gs_bar();
		}
		if (moved) flag[it % 3] = 1;
		if (tid == 0) flag[(it + 1) % 3] = 0; // (the word of the pass after next: nobody reads it now -- its last readers left two barriers ago)
		gs_bar();
		settled = flag[it % 3] == 0;
	}
	return settled;
}

// FIX: the cm order by transpositions out of the cs order where the host asks for it (GenomeSort::y_fixup).  Only the 14-items form is built with it:
// in the 10-items form -- 64 VGPRs, two workgroups a CU -- the mere presence of the loop cost the kernel 30 spilled registers inlined and a stack
// frame as a call, 327 -> 535-570 us at 12.1 M hits either way (round 6, profiles/r06e_*).
template <int T, int K, bool FIX>
__device__ __forceinline__ void gs2_body(const GenomeSort &a, unsigned char *gs_mem)
{
	constexpr int NW = T / WAVE;
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, np = a.np;
	int g, gb, n, c0 = 0; bool first = true; // the unit: a genome, or (contig bins) consecutive contigs c0 ... of genome g -- `first`: from the genome's first hit on
	if (a.bins) { const int4 b = a.bins[blockIdx.x]; gb = b.x, n = b.y, g = b.z & 0x7fffffff, first = b.z < 0, c0 = b.w; } // (headpos: the host copies goff)
	else {
		g = a.glist ? a.glist[blockIdx.x] : (int)blockIdx.x, gb = a.goff[g], n = a.goff[g + 1] - gb;
		if (tid == 0) { a.headpos[g] = gb; if (g == a.n_genome - 1) a.headpos[g + 1] = gb + n; }
	}
	if (n == 0) return;
	const size_t s_bytes = 3 * (size_t)np + sizeof(uint32_t) * NW * 256 > 4 * (size_t)np ? 3 * (size_t)np + sizeof(uint32_t) * NW * 256 : 4 * (size_t)np;
	uint16_t *const idx0 = (uint16_t *)gs_mem;
	unsigned char *const S = gs_mem + 2 * (size_t)np;
	GsLds L;
	L.cur = idx0, L.alt = (uint16_t *)S, L.dig = S + 2 * (size_t)np, L.whist = (uint32_t *)(S + 3 * (size_t)np), L.stage = (uint32_t *)S;
	L.head = (unsigned long long *)(S + s_bytes), L.tie = L.head + np / 64;
	L.wtot = (uint32_t *)(L.tie + np / 64), L.wagg = (int2 *)(L.wtot + NW);
	const int64_t N = a.N;
	const int32_t *const up = a.up + gb; // plane f of the genome at up + f * N (0 pid, 1 contig, 2 rank, 3 score_ori, 4 score_adj, 5 n_exon, 6 off_exon, 7 cs, 8 ce, 9 cm; 12 gene, 13 CDS length, 15 score key, 16 rev / multi-exon bits)
	const int cb = a.ctg_base[g];
	uint32_t *const st = L.stage;

	// plane f, file order -> registers (item tid + u * T in element u)
#define GS2_LOAD(R, f) do { _Pragma("unroll") for (int u = 0; u < K; ++u) { const int i = tid + u * T; (R)[u] = i < n ? (uint32_t)up[(int64_t)(f) * N + i] : 0u; } } while (0)
	// registers (file order) -> staging area; then the plane in X order: V[u] = value of the hit at X position tid + u * T
#define GS2_STAGE(R) do { _Pragma("unroll") for (int u = 0; u < K; ++u) { const int i = tid + u * T; if (i < n) st[i] = (R)[u]; } gs_bar(); } while (0)
#ifndef GS2_PREFETCH
#define GS2_PREFETCH 1
#endif
#if GS2_PREFETCH // the next plane's loads fly while this one is gathered (ten more live registers)
#define GS2_BEGIN(cur, next) do { GS2_STAGE(R); if ((next) >= 0) GS2_LOAD(R, next); } while (0)
#else
#define GS2_BEGIN(cur, next) do { GS2_LOAD(R, cur); GS2_STAGE(R); } while (0)
#endif
#define GS2_GET(V) do { uint32_t j_[K]; _Pragma("unroll") for (int u = 0; u < K; ++u) { const int x = tid + u * T; j_[u] = x < n ? idx0[x] : 0u; } \
		_Pragma("unroll") for (int u = 0; u < K; ++u) (V)[u] = st[j_[u]]; } while (0)
#define GS2_OUT(dst, V) do { _Pragma("unroll") for (int u = 0; u < K; ++u) { const int x = tid + u * T; if (x < n) (dst)[gb + x] = (int32_t)(V)[u]; } } while (0)

	// ---- X order: pg_hit_sort(g, 0) = by (contig, cs), ties in file order (the reference's own tie order is replayed later where it matters) ----
	{
		uint32_t key[K];
		GS2_LOAD(key, 7);
#pragma unroll
		for (int u = 0; u < K; ++u) { const int i = tid + u * T; if (i < n) L.cur[i] = (uint16_t)i; }
		if (a.ctg_bits + a.cs_bits <= 32) { // one composite key
			uint32_t cg[K];
			GS2_LOAD(cg, 1);
#pragma unroll
			for (int u = 0; u < K; ++u) key[u] |= a.cs_bits < 32 ? (cg[u] - (uint32_t)c0) << a.cs_bits : 0u;
			gs_bar();
			gs2_sort_bits<T, K>(L, n, key, a.cs_bits + a.ctg_bits);
		} else {
			gs_bar();
			gs2_sort_bits<T, K>(L, n, key, a.cs_bits);
			GS2_LOAD(key, 1);
#pragma unroll
			for (int u = 0; u < K; ++u) key[u] -= (uint32_t)c0;
			gs2_sort_bits<T, K>(L, n, key, a.ctg_bits);
		}
	}
	if (L.cur != idx0) { // the permutation phase wants the order in the first array (S becomes the staging area)
		uint32_t t[K];
#pragma unroll
		for (int u = 0; u < K; ++u) { const int i = tid + u * T; t[u] = i < n ? L.cur[i] : 0u; }
		gs_bar(); // (nobody reads S any more)
#pragma unroll
		for (int u = 0; u < K; ++u) { const int i = tid + u * T; if (i < n) idx0[i] = (uint16_t)t[u]; }
	}
	gs_bar();

	// ---- the planes through LDS, one at a time: coalesced read in file order, gather in LDS, coalesced write in X order; the words of a
	// 16-byte record are collected in registers over consecutive planes and written whole ----
	uint32_t W0[K], W1[K], W2[K], V[K], R[K];
	// record A = {cs, seg, ce, pm} (k_sweep.hpp).  Plane 1, contig: segment ids, and where a contig starts in X order (bit array)
#if GS2_PREFETCH
	GS2_LOAD(R, 1);
#endif
	GS2_BEGIN(1, 7);
	GS2_GET(V);
#pragma unroll
	for (int u = 0; u < K; ++u) {
		const int x = tid + u * T;
		if (x - lane >= n) break; // wave-uniform
		const bool v = x < n;
		const uint32_t cp = (v && x > 0) ? st[idx0[x - 1]] : ~0u;
		W1[u] = (uint32_t)cb + V[u];
		if (v) a.o.seg[gb + x] = (int32_t)W1[u];
		const unsigned long long hb = __ballot(v && V[u] != cp);
		if (lane == 0) L.head[x >> 6] = hb;
	}
	gs_bar();
	// plane 7, cs: the static marks of the cs sort's tie groups (hazard H2b, see k_rep_fill)
	GS2_BEGIN(7, 8);
	GS2_GET(W0);
#pragma unroll
	for (int u = 0; u < K; ++u) {
		const int x = tid + u * T;
		if (x - lane >= n) continue;
		const bool v = x < n;
		bool tie = false;
		if (v) {
			const bool hd = (L.head[x >> 6] >> (x & 63)) & 1ull, hn = x + 1 < n ? (bool)((L.head[(x + 1) >> 6] >> ((x + 1) & 63)) & 1ull) : true;
			tie = (!hd && st[idx0[x - 1]] == W0[u]) || (!hn && st[idx0[x + 1]] == W0[u]);
		}
		const unsigned long long tb = __ballot(tie);
		if (lane == 0) L.tie[x >> 6] = tb;
	}
	gs_bar();
	// plane 8, ce, and pm = its running maximum inside a contig (bounds the sweep's look-back): a segmented inclusive max scan over the X
	// order, wave w over a contiguous span, two sweeps (span aggregates, then the scan proper); the result goes back into the staging area
	GS2_BEGIN(8, 15);
	GS2_GET(W2); // ce in X order
	gs_bar();
	{
		const int span = (((n + NW - 1) / NW) + 63) & ~63;
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
	GS2_GET(V); // pm
#pragma unroll
	for (int u = 0; u < K; ++u) {
		const int x = tid + u * T;
		if (x < n) a.A[gb + x] = make_int4((int)W0[u], (int)W1[u], (int)W2[u], (int)V[u]);
	}
	gs_bar();
	// record B = {rk, gid, cds, pid}: planes 15 (score key), 12 (gene), 13 (CDS length), 0 (protein)
	GS2_BEGIN(15, 12); GS2_GET(W0); gs_bar();
	GS2_BEGIN(12, 13); GS2_GET(W1); GS2_OUT(a.o.gid, W1); gs_bar();
	GS2_BEGIN(13, 0); GS2_GET(W2); gs_bar();
	GS2_BEGIN(0, 2); GS2_GET(V); GS2_OUT(a.o.pid, V);
#pragma unroll
	for (int u = 0; u < K; ++u) {
		const int x = tid + u * T;
		if (x < n) a.B[gb + x] = make_int4((int)W0[u], (int)W1[u], (int)W2[u], (int)V[u]);
	}
	gs_bar();
	// record C = {rank, n_exon, off_exon, score_ori}: planes 2, 5, 6, 3
	GS2_BEGIN(2, 5); GS2_GET(W0); GS2_OUT(a.o.rank, W0); gs_bar();
	GS2_BEGIN(5, 6); GS2_GET(W1); GS2_OUT(a.o.nex, W1); gs_bar();
	GS2_BEGIN(6, 3); GS2_GET(W2); gs_bar();
	GS2_BEGIN(3, 4); GS2_GET(V); GS2_OUT(a.o.sori, V);
#pragma unroll
	for (int u = 0; u < K; ++u) {
		const int x = tid + u * T;
		if (x < n) a.C[gb + x] = make_int4((int)W0[u], (int)W1[u], (int)W2[u], (int)V[u]);
	}
	gs_bar();
	// plane 4, score_adj; plane 16, the static flag bits (+ head of the genome, + member of a cs tie group); file index, genome
	GS2_BEGIN(4, 16); GS2_GET(V); GS2_OUT(a.o.sadj, V); gs_bar();
	GS2_BEGIN(16, a.bins ? 17 : 9); GS2_GET(V);
#pragma unroll
	for (int u = 0; u < K; ++u) {
		const int x = tid + u * T;
		if (x >= n) break;
		a.o.flags[gb + x] = V[u] | ((x == 0 && first) ? F_HEAD : 0u) | (((L.tie[x >> 6] >> (x & 63)) & 1ull) ? F_CSTIE : 0u);
		a.o.gnm[gb + x] = g;
		if (!a.bins) a.o.fidx[gb + x] = (int32_t)idx0[x];
	}
	gs_bar();
	if (a.bins) { GS2_BEGIN(17, 9); GS2_GET(V); GS2_OUT(a.o.fidx, V); gs_bar(); } // the file index travels as a plane of its own (the input is grouped by contig)
	// ---- Y order: pg_hit_sort(g, 1) = by (contig, cm), ties in X order; the items are X positions now, the keys cm and contig in X order ----
	GS2_BEGIN(9, 1); GS2_GET(W0); GS2_OUT(a.o.cm, W0); gs_bar();
	GS2_BEGIN(1, -1); GS2_GET(W1); gs_bar(); // contig, once more (cheaper than ten registers held since the first plane)
#undef GS2_LOAD
#undef GS2_BEGIN
#undef GS2_STAGE
#undef GS2_GET
#undef GS2_OUT
	L.cur = idx0, L.alt = (uint16_t *)S;
#pragma unroll
	for (int u = 0; u < K; ++u) { const int x = tid + u * T; if (x < n) L.cur[x] = (uint16_t)x; }
#pragma unroll
	for (int u = 0; u < K; ++u) W1[u] -= (uint32_t)c0;
	if (a.ctg_bits + a.cm_bits <= 32) {
#pragma unroll
		for (int u = 0; u < K; ++u) W0[u] |= a.cm_bits < 32 ? W1[u] << a.cm_bits : 0u;
		gs_bar();
		// Round 6: the cm order is the cs order up to inversions between OVERLAPPING hits (cs_i <= cs_j and cm_i > cm_j: hit i reaches past the start
		// of j), so it is a few transpositions away from the order the unit is in -- odd-even transposition passes over (key, position) in LDS, stable
		// (equal keys are never exchanged, so ties keep the X order), until a double pass moves nothing; piles deeper than GS2_FIX_IT double passes
		// (synth.dense) go on with the radix passes from where the transpositions left them (any permutation is a valid start of a stable LSD sort).
		bool settled = false;
		if (FIX && a.y_fixup) {
			uint32_t *const yk = (uint32_t *)S; // (the staging area is free: 4 np bytes)
			volatile uint32_t *const flag = L.wtot; // [3] used in turn
#pragma unroll
			for (int u = 0; u < K; ++u) { const int x = tid + u * T; if (x < n) yk[x] = W0[u]; }
			if (tid < 3) flag[tid] = 0;
			gs_bar();
			// (the keys need not stay in registers through the passes -- 64 VGPRs are all this kernel has: should the radix passes be needed after
			// all, the keys come back out of the planes this workgroup has just written: cm and the contig segment in X order)
			settled = gs2_fixup<T>(yk, L.cur, flag, n);
			if (!settled) {
				__syncthreads(); // (the stores of the two planes have to have landed)
#pragma unroll
				for (int u = 0; u < K; ++u) {
					const int x = tid + u * T;
					W0[u] = x < n ? ((uint32_t)a.o.cm[gb + x] | (a.cm_bits < 32 ? ((uint32_t)(a.o.seg[gb + x] - cb) - (uint32_t)c0) << a.cm_bits : 0u)) : 0u;
				}
			}
		}
		if (!settled) gs2_sort_bits<T, K>(L, n, W0, a.cm_bits + a.ctg_bits);
	} else {
		gs_bar();
		gs2_sort_bits<T, K>(L, n, W0, a.cm_bits);
		gs2_sort_bits<T, K>(L, n, W1, a.ctg_bits);
	}
#pragma unroll
	for (int u = 0; u < K; ++u) { const int y = tid + u * T; if (y < n) a.yperm[gb + y] = gb + (int32_t)L.cur[y]; }
}

// up to 10 items per thread, 64 VGPRs: two workgroups per CU
__global__ __launch_bounds__(GS2_T, 8) void k_genome_sort2(GenomeSort a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char gs2_mem[];
	gs2_body<GS2_T, GS2_K, false>(a, gs2_mem);
}
// up to 14 items per thread (np <= 14 336: what the largest genomes of the bacterial sets need), 128 VGPRs: one workgroup per CU like the
// round-3 kernel, but 6 radix passes instead of 8 at these sizes and no patch-up pass
__global__ __launch_bounds__(GS2_T, 4) void k_genome_sort2d(GenomeSort a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char gs2d_mem[];
	gs2_body<GS2_T, GS2_K_BIG, true>(a, gs2d_mem);
}
