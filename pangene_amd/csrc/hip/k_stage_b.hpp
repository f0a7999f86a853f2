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
