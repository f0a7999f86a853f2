// k_vertex.hpp -- pg_gen_vtx per-genome part and pg_graph_flag_vtx.
// Included by pga_backend.hip (one translation unit); uses the context types, BLOCK / WAVE and dev_prims.hpp from there.
#pragma once

// ------------------------------------------------------------------------------------------------
// pg_gen_vtx, per-genome part (vertex.c:28-51)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_vtx1(const uint32_t *flags, const int32_t *gnm, const int32_t *gid, const int32_t *rank, const int32_t *pdom,
                                                  int n, int Q, int32_t *cnt, uint32_t *dombits, int64_t words_per_genome, int64_t *dcnt, int64_t *live_cnt)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	const uint32_t f = h < n ? flags[h] : (uint32_t)PGA_F_FLT;
	if (live_cnt) { // the hits that are not filtered (one atomic a wave, spread over LIVE_CNT_N words) -- what the rounds behind this step still have to look at (ensure_z)
		const unsigned long long live = __ballot(!(f & PGA_F_FLT));
		if (live && (threadIdx.x & 63) == 0) atomicAdd((unsigned long long *)&live_cnt[(blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6)) & (LIVE_CNT_N - 1)], (unsigned long long)__popcll(live));
	}
	if (h >= n) return;
	if ((f & PGA_F_FLT) || rank[h] != 0) return;
	int g = gid[h];
	if (f & PGA_F_SHADOW) {
		if (pdom[h] < 0) atomicAdd((unsigned long long *)&dcnt[3], 1ull); // vertex.c:38
		atomicAdd(&cnt[Q + g], 1);
	} else {
		atomicAdd(&cnt[g], 1);
		uint32_t old = atomicOr(&dombits[(int64_t)gnm[h] * words_per_genome + (g >> 5)], 1u << (g & 31));
		if (old & (1u << (g & 31))) atomicAdd((unsigned long long *)&dcnt[3], 1ull); // two rank-0 hits of one gene: cannot happen after hit.c:107-128
	}
}

// The same with the genes' counts through LDS (round 6, as k_post_part_lds: one contribution per (genome, gene) is what there is to reduce -- a workgroup reads a stretch
// of many genomes and sends what is not zero at the end; the (genome, gene) bits have nothing to reduce and stay global atomics): 8 bytes a gene, shards from 0.5 M hits.
constexpr int VX_T = 1024;
__global__ __launch_bounds__(VX_T) void k_vtx1_lds(const uint32_t *flags, const int32_t *gnm, const int32_t *gid, const int32_t *rank, const int32_t *pdom,
                                                     int n, int Q, int32_t *cnt, uint32_t *dombits, int64_t words_per_genome, int64_t *dcnt, int64_t *live_cnt)
{
	extern __shared__ int vx_cnt[]; // [2 Q]
	for (int i = threadIdx.x; i < 2 * Q; i += VX_T) vx_cnt[i] = 0;
	__syncthreads();
	const int64_t per = (((int64_t)n + gridDim.x - 1) / gridDim.x + 63) & ~(int64_t)63, h0 = (int64_t)blockIdx.x * per, h1 = h0 + per < n ? h0 + per : n; // (whole waves: the ballots)
	long long n_live = 0;
	for (int64_t b = h0; b < h1; b += VX_T) {
		const int64_t h = b + threadIdx.x;
		const uint32_t f = h < h1 ? flags[h] : (uint32_t)PGA_F_FLT;
		if (live_cnt) n_live += __popcll(__ballot(!(f & PGA_F_FLT)));
		if (h >= h1 || (f & PGA_F_FLT) || rank[h] != 0) continue;
		const int g = gid[h];
		if (f & PGA_F_SHADOW) {
			if (pdom[h] < 0) atomicAdd((unsigned long long *)&dcnt[3], 1ull); // vertex.c:38
			atomicAdd(&vx_cnt[Q + g], 1);
		} else {
			atomicAdd(&vx_cnt[g], 1);
			const uint32_t old = atomicOr(&dombits[(int64_t)gnm[h] * words_per_genome + (g >> 5)], 1u << (g & 31));
			if (old & (1u << (g & 31))) atomicAdd((unsigned long long *)&dcnt[3], 1ull); // two rank-0 hits of one gene: cannot happen after hit.c:107-128
		}
	}
	if (live_cnt && (threadIdx.x & 63) == 0 && n_live) atomicAdd((unsigned long long *)&live_cnt[(blockIdx.x * (VX_T / WAVE) + (threadIdx.x >> 6)) & (LIVE_CNT_N - 1)], (unsigned long long)n_live);
	__syncthreads();
	for (int i = threadIdx.x; i < 2 * Q; i += VX_T) { const int v = vx_cnt[i]; if (v) atomicAdd(&cnt[i], v); }
}

// Fold of the (genome, sub gene, dom gene) relation into one genome bitset per (sub, dom) pair, the form the host greedy
// consumes (vertex.c:60-80 marks cell (genome, dom) for every genome of the pair).  A sub gene has very few distinct dom
// genes, so each gene owns VTX_K slots: a slot is claimed for a dom gene with atomicCAS, the genome bit is an atomicOr.
// A gene with more than VTX_K dom genes spills single-genome records into an overflow area.
constexpr int VTX_K = 8;

__global__ __launch_bounds__(BLOCK) void k_vtx_fold(const uint32_t *flags, const int32_t *gnm, const int32_t *gid, const int32_t *rank, const int32_t *pdom,
                                                      const int32_t *prot_gid, const int32_t *ggl, int n, const uint32_t *dombits, int64_t words_per_genome,
                                                      int32_t *dom_tab, unsigned long long *bits, int nw, unsigned long long *ovf, long long ovf_cap, int64_t *dcnt)
{
	const int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	const uint32_t f = flags[h];
	if ((f & PGA_F_FLT) || rank[h] != 0 || !(f & PGA_F_SHADOW) || pdom[h] < 0) return;
	const int j = gnm[h], D = prot_gid[pdom[h]], g = gid[h];
	if (!(dombits[(int64_t)j * words_per_genome + (D >> 5)] >> (D & 31) & 1u)) return; // dom is not dominant in this genome: the greedy never looks
	const int jg = ggl[j];
	int k = 0;
	for (; k < VTX_K; ++k) {
		int32_t *p = &dom_tab[(int64_t)g * VTX_K + k];
		int cur = *(volatile int32_t *)p;
		if (cur < 0) cur = atomicCAS(p, -1, D), cur = cur < 0 ? D : cur;
		if (cur == D) break;
	}
	if (k < VTX_K) {
		atomicOr(&bits[((int64_t)g * VTX_K + k) * nw + (jg >> 6)], 1ull << (jg & 63));
	} else {
		const long long at = (long long)atomicAdd((unsigned long long *)&dcnt[0], 1ull);
		if (at < ovf_cap) {
			unsigned long long *r = ovf + at * (1 + nw);
			r[0] = (unsigned long long)g << 20 | (unsigned long long)D;
			for (int w = 0; w < nw; ++w) r[1 + w] = w == (jg >> 6) ? 1ull << (jg & 63) : 0ull;
		}
	}
}

struct InDomSet { const int32_t *tab; __device__ __forceinline__ I32 operator()(int64_t i) const { return I32{tab[i] >= 0 ? 1 : 0}; } };

// slot -> record: key (sub << 20 | dom), then the genome words; also mails the record count (dcnt[10]) and the counters to the host
__global__ __launch_bounds__(BLOCK) void k_vtx_compact(const int32_t *dom_tab, const int32_t *slot, int64_t n_slot, const unsigned long long *bits, int nw,
                                                         unsigned long long *out, int64_t *dcnt, int64_t *host_box)
{
	const int64_t s = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (s >= n_slot) return;
	const int D = dom_tab[s];
	if (D >= 0) {
		unsigned long long *r = out + (int64_t)slot[s] * (1 + nw);
		r[0] = (unsigned long long)(s / VTX_K) << 20 | (unsigned long long)D;
		for (int w = 0; w < nw; ++w) r[1 + w] = bits[s * nw + w];
	}
	if (s == n_slot - 1) {
		dcnt[10] = slot[s] + (D >= 0 ? 1 : 0);
		__threadfence();
		for (int t = 0; t < 16; ++t) sys_store(&host_box[t], dcnt[t]);
	}
}

// live_cnt (or NULL): += the hits that are not filtered afterwards (one atomic a wave): pga_branch_loop asks now and then whether the live lists are worth building again
__global__ __launch_bounds__(BLOCK) void k_flag_vtx(uint32_t *flags, const int32_t *gid, int n, const int32_t *g2s, int then_filter, Gate gate = Gate{nullptr, 0}, int64_t *live_cnt = nullptr) // graph.c:61-69 (+ PG_SET_FILTER(vtx == 0))
{
	if (gate_closed(gate)) return;
	int h = blockIdx.x * BLOCK + threadIdx.x;
	uint32_t nf = PGA_F_FLT;
	if (h < n) {
		const uint32_t f = flags[h];
		nf = g2s[gid[h]] >= 0 ? (f | PGA_F_VTX) : (f & ~PGA_F_VTX);
		if (then_filter && !(nf & PGA_F_VTX)) nf |= PGA_F_FLT;
		if (nf != f) flags[h] = nf;
	}
	if (live_cnt) {
		const unsigned long long live = __ballot(!(nf & PGA_F_FLT));
		if (live && (threadIdx.x & 63) == 0) atomicAdd((unsigned long long *)&live_cnt[(blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6)) & (LIVE_CNT_N - 1)], (unsigned long long)__popcll(live));
	}
}
// The same for the members of the live lists alone, along the gene-major index (pga_branch_loop, when few hits are left: at configs[3]'s 96.6 M hits the
// fifteen rounds read 8 bytes of every hit to filter the hits of a few deleted genes among the 0.7-13 % that are not filtered already).  Hits outside the lists
// are filtered, and their vtx bit is set again from the final g2s by the flag_vtx behind the loop.
__global__ __launch_bounds__(BLOCK) void k_flag_vtx_z(uint32_t *flags, const int32_t *zx, const int32_t *zg, int nz, const int32_t *g2s, Gate gate, int64_t *live_cnt)
{
	if (gate_closed(gate)) return;
	const int z = blockIdx.x * BLOCK + threadIdx.x;
	bool live = false;
	if (z < nz) {
		const int x = zx[z];
		const uint32_t f = flags[x];
		uint32_t nf = g2s[zg[z]] >= 0 ? (f | PGA_F_VTX) : ((f & ~PGA_F_VTX) | PGA_F_FLT);
		if (nf != f) flags[x] = nf;
		live = !(nf & PGA_F_FLT);
	}
	if (live_cnt) {
		const unsigned long long m = __ballot(live);
		if (m && (threadIdx.x & 63) == 0) atomicAdd((unsigned long long *)&live_cnt[(blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6)) & (LIVE_CNT_N - 1)], (unsigned long long)__popcll(m));
	}
}
// dcnt[8] = the sum of the partial counts
__global__ __launch_bounds__(BLOCK) void k_live_sum(const int64_t *live_cnt, int64_t *dcnt)
{
	__shared__ long long part[BLOCK / WAVE];
	long long s = 0;
	for (int i = threadIdx.x; i < LIVE_CNT_N; i += BLOCK) s += live_cnt[i];
	s = (long long)wave_sum64((unsigned long long)s);
	if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0) { long long t = 0; for (int k = 0; k < BLOCK / WAVE; ++k) t += part[k]; dcnt[8] = t; }
}
