// k_stage_b.hpp -- stage B kernels (hit.c:153-247) and PG_SET_FILTER.
// Included by pga_backend.hip (one translation unit); uses the context types, BLOCK / WAVE and dev_prims.hpp from there.
#pragma once

// ------------------------------------------------------------------------------------------------
// stage B (hit.c:153-247)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_post_part(const uint32_t *flags, const int32_t *pid, const int32_t *rank, const int32_t *sori, const int32_t *sadj,
                                                       const int32_t *nex, int n, int P, int32_t *max_ori, unsigned long long *sums)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	int p = pid[h];
	// pg_cap_score_dom's table (hit.c:230-238).  A protein has thousands of hits in a shard and one maximum: look before asking the L2 for
	// an atomic -- the value only grows, so a stale (smaller) reading costs a redundant atomic, never a missed one
	const int so = sori[h];
	if (max_ori[p] < so) atomicMax(&max_ori[p], so);
	if (rank[h] == 0 && !(flags[h] & PGA_F_FLT)) {
		int w = nex[h] == 1 ? 0 : 1;
		atomicAdd(&sums[p], (unsigned long long)(long long)sadj[h]);
		atomicAdd(&sums[(int64_t)(2 + w) * P + p], 1ull); // (the count of hit.c:201 is c[0] + c[1] of hit.c:165: k_post_count adds them up -- one atomic a hit less)
		atomicAdd(&sums[(int64_t)(4 + w) * P + p], (unsigned long long)(long long)sori[h]);
	}
}
// The same sums with the atomics in LDS (round 6).  The device executes global atomics at ~30 G/s whatever their addresses, and a hit contributes three: 0.74 ms at
// 12.1 M hits, 6.7 ms at configs[3]'s 96.6 M.  What there is to reduce is one contribution per (genome, protein): a workgroup that reads the hits of MANY genomes holds the
// proteins' sums of its stretch in LDS -- 28 bytes a protein (three 64-bit sums, the two counts in the halves of one word, emptied into the global table when half full), the
// proteins in p_tiles ranges when they do not fit (workgroup b: range b % p_tiles, stretch b / p_tiles) -- and sends what is not zero to the global tables at the end.
// max_ori stays as it was (a look, rarely an atomic).  Same integers, another order of additions.
constexpr int PP_T = 1024;
__global__ __launch_bounds__(PP_T) void k_post_part_lds(const uint32_t *flags, const int32_t *pid, const int32_t *rank, const int32_t *sori, const int32_t *sadj,
                                                          const int32_t *nex, int n, int P, int p_tiles, int PT /* proteins a range */, int32_t *max_ori, unsigned long long *sums)
{
	extern __shared__ unsigned long long pp_lds[];
	unsigned long long *s_adj = pp_lds, *s_o0 = pp_lds + PT, *s_o1 = pp_lds + 2 * (size_t)PT;
	unsigned int *s_cnt = (unsigned int *)(pp_lds + 3 * (size_t)PT);
	const int t = blockIdx.x % p_tiles, chunk = blockIdx.x / p_tiles, n_chunk = gridDim.x / p_tiles;
	const int p0 = t * PT, p1 = p0 + PT < P ? p0 + PT : P;
	for (int i = threadIdx.x; i < PT; i += PP_T) s_adj[i] = 0, s_o0[i] = 0, s_o1[i] = 0, s_cnt[i] = 0;
	__syncthreads();
	const int64_t per = ((int64_t)n + n_chunk - 1) / n_chunk, h0 = (int64_t)chunk * per, h1 = h0 + per < n ? h0 + per : n;
	for (int64_t h = h0 + threadIdx.x; h < h1; h += PP_T) {
		const int p = pid[h], so = sori[h];
		if (t == 0 && max_ori[p] < so) atomicMax(&max_ori[p], so); // pg_cap_score_dom's table (hit.c:230-238), as k_post_part
		if (p < p0 || p >= p1 || rank[h] != 0 || (flags[h] & PGA_F_FLT)) continue;
		const int w = nex[h] == 1 ? 0 : 1, i = p - p0;
		atomicAdd(&s_adj[i], (unsigned long long)(long long)sadj[h]);
		{ // (a count that fills half of its sixteen bits moves on to the global table at once: any number of rank-0 hits of one protein is counted exactly --
		  // at most 1 023 other additions are on their way while the 2^15 are taken out again)
			const unsigned old = atomicAdd(&s_cnt[i], w ? 65536u : 1u);
			if ((w ? old >> 16 : old & 0xffffu) == 0x7fffu) { atomicAdd(&sums[(int64_t)(2 + w) * P + p], 0x8000ull); atomicSub(&s_cnt[i], w ? 0x80000000u : 0x8000u); }
		}
		atomicAdd(w ? &s_o1[i] : &s_o0[i], (unsigned long long)(long long)so);
	}
	__syncthreads();
	for (int i = threadIdx.x; i < p1 - p0; i += PP_T) {
		const unsigned int cw = s_cnt[i];
		const unsigned long long a = s_adj[i], o0 = s_o0[i], o1 = s_o1[i];
		const int p = p0 + i;
		if (a) atomicAdd(&sums[p], a); // (what is zero adds nothing: most proteins of most stretches)
		if (cw & 0xffffu) atomicAdd(&sums[(int64_t)2 * P + p], (unsigned long long)(cw & 0xffffu));
		if (cw >> 16) atomicAdd(&sums[(int64_t)3 * P + p], (unsigned long long)(cw >> 16));
		if (o0) atomicAdd(&sums[(int64_t)4 * P + p], o0);
		if (o1) atomicAdd(&sums[(int64_t)5 * P + p], o1);
	}
}
__global__ __launch_bounds__(BLOCK) void k_post_count(unsigned long long *sums, int P)
{
	const int p = blockIdx.x * BLOCK + threadIdx.x;
	if (p < P) sums[(int64_t)P + p] = sums[(int64_t)2 * P + p] + sums[(int64_t)3 * P + p];
}

__global__ __launch_bounds__(BLOCK) void k_post_apply(uint32_t *flags, const int32_t *pid, const int32_t *nex, int32_t *sdom, int n,
                                                        const int32_t *max_ori, const uint8_t *rep, const uint8_t *pj, int64_t *cnt)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	int p = pid[h];
	int mo = max_ori[p];
	if (sdom[h] > mo) sdom[h] = mo; // hit.c:243-244
	uint32_t f = flags[h], nf = rep[p] ? (f | PGA_F_REP) : (f & ~PGA_F_REP);
	if (!(f & (PGA_F_FLT | PGA_F_PSEUDO)) && nex[h] == 1 && pj[p]) { // hit.c:175-182
		nf |= PGA_F_PSEUDO;
		if (cnt) atomicAdd((unsigned long long *)cnt, 1ull); // log only
	}
	if (nf != f) flags[h] = nf;
}

__global__ __launch_bounds__(BLOCK) void k_set_filter(uint32_t *flags, int n, int which) // pgpriv.h:109-116
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	uint32_t f = flags[h];
	bool hit = which == PGA_FLT_PSEUDO ? (f & PGA_F_PSEUDO) != 0
	         : which == PGA_FLT_VTX0 ? (f & PGA_F_VTX) == 0
	         : which == PGA_FLT_WEAK2 ? ((f & PGA_F_WEAK_MASK) >> PGA_F_WEAK_SHIFT) == 2
	         : (f & PGA_F_SHADOW) != 0;
	if (hit && !(f & PGA_F_FLT)) flags[h] = f | PGA_F_FLT;
}
