// k_common.hpp -- small device-side helpers shared by every kernel file.
// Included by pga_backend.hip (one translation unit); uses the context types, BLOCK / WAVE and dev_prims.hpp from there.
#pragma once

// (int32_t)double as the reference's build does it (x86-64 cvttsd2si: truncation, and 0x80000000 for anything that does not fit).  The
// hardware conversion here saturates instead.  It only matters for the averaged arc distance of graph.c:141 once contig coordinates
// pass 32 bits: graph.c:73 narrows a distance to int32_t, a negative one goes into the uint64_t sum of graph.c:131-133 sign-extended,
// and the average of THAT no longer fits (the reference prints ad:i:-2147483647 for such arcs; tests/test_abi.py holds one).
__device__ __forceinline__ int32_t cvt_i32_x86(double v) { return (v > -2147483649.0 && v < 2147483648.0) ? (int32_t)v : (int32_t)0x80000000; }

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t hash_u32(uint32_t key) // pg_hash_uint32, pgpriv.h:88-97
{
	key += ~(key << 15);
	key ^=  (key >> 10);
	key +=  (key << 3);
	key ^=  (key >> 6);
	key += ~(key << 11);
	key ^=  (key >> 16);
	return key;
}

// Pinned host memory seen from a kernel.  The caches between a CU and the host are not part of the in-stream ordering the kernels
// rely on among themselves: a plain store may rest in the L2 of the XCD that issued it until a system-scope release -- the one the
// runtime issues when the host asks about the stream (sync_st), which is why the bulk results (segment counters, degrees,
// n_dist_loci: thousands of words a round) can be plain stores.  The few words of the mailboxes and the staging-area reads of
// k_copy_in are accessed at system scope explicitly (a system-scope store per word is slow: doing it for the bulk results made
// the gene kernels 30 us slower).
__device__ __forceinline__ void sys_store(int32_t *p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void sys_store(int64_t *p, int64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ uint32_t sys_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// small uploads (a g2s table, per-protein marks, head indices) straight out of the pinned staging area: one short kernel in
// stream order instead of a DMA command, whose set-up latency sits between two dependent kernels.  src is 4-byte aligned and
// padded to whole words (stage_upload); dst may be any byte address.
struct CopyIn { void *dst[2]; const uint32_t *src[2]; unsigned long long n[2]; };
__global__ __launch_bounds__(BLOCK) void k_copy_in(CopyIn l)
{
	const size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x, step = (size_t)gridDim.x * BLOCK;
#pragma unroll
	for (int k = 0; k < 2; ++k) {
		const size_t n = l.n[k];
		if (n == 0) continue;
		if (((size_t)l.dst[k] & 3) == 0) {
			uint32_t *d = (uint32_t *)l.dst[k];
			for (size_t i = t; i < (n >> 2); i += step) d[i] = sys_load(l.src[k] + i);
			for (size_t i = (n & ~(size_t)3) + t; i < n; i += step) ((uint8_t *)l.dst[k])[i] = (uint8_t)(sys_load(l.src[k] + (i >> 2)) >> (8 * (i & 3)));
		} else {
			for (size_t i = t; i < n; i += step) ((uint8_t *)l.dst[k])[i] = (uint8_t)(sys_load(l.src[k] + (i >> 2)) >> (8 * (i & 3)));
		}
	}
}

__global__ void k_fill_i32(int32_t *p, int64_t n, int32_t v)
{
	int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i < n) p[i] = v;
}

// mailbox[k] = a[0] + b[0]: totals of a scan land in the device mailbox so that one 128-byte copy brings every
// size the host needs (one round trip instead of one per value)
// dcnt[10] = a[0] + b[0] (element count after a compaction scan), then all 16 device counters go straight into the pinned
// host mirror: the host reads them after the stream sync without a separate copy command
__global__ void k_mail_sum(const int32_t *a, const int32_t *b, int64_t *dcnt, int64_t *host_box)
{
	if (threadIdx.x == 0) dcnt[10] = (int64_t)a[0] + b[0];
	__syncthreads();
	if (threadIdx.x < 16) sys_store(&host_box[threadIdx.x], dcnt[threadIdx.x]);
}

// the same for a run count: dcnt[10] = number of distinct keys of a sorted array, from the exclusive count of run heads before
// its last element
__global__ void k_mail_runs(const uint64_t *key, const int32_t *slot, int64_t m, int64_t *dcnt, int64_t *host_box)
{
	if (threadIdx.x == 0) dcnt[10] = (int64_t)slot[m - 1] + ((m == 1 || key[m - 1] != key[m - 2]) ? 1 : 0);
	__syncthreads();
	if (threadIdx.x < 16) sys_store(&host_box[threadIdx.x], dcnt[threadIdx.x]);
}

// the number of gene pairs of a branch round (dcnt[15] = a[0] + b[0]) for kernels and host alike
__global__ void k_mail_pairs(const int32_t *a, const int32_t *b, int64_t *dcnt, int64_t *host_box)
{
	if (threadIdx.x == 0) dcnt[15] = (int64_t)a[0] + b[0];
	__syncthreads();
	if (threadIdx.x < 16) sys_store(&host_box[threadIdx.x], dcnt[threadIdx.x]);
}

// end of an arc round on the gene-major index: the counters for the host, then the round's overflow counter starts again
__global__ void k_mail_round(int64_t *dcnt, int64_t *host_box, int32_t *tail /* pinned, or NULL: {overflowed genes, invariant violations} of this round */)
{
	if (threadIdx.x < 16) sys_store(&host_box[threadIdx.x], dcnt[threadIdx.x]);
	if (threadIdx.x == 0 && tail) sys_store(&tail[0], (int32_t)dcnt[9]), sys_store(&tail[1], (int32_t)dcnt[3]);
	__syncthreads();
	if (threadIdx.x == 0) { if (dcnt[9] || dcnt[3]) dcnt[11] = 1; dcnt[9] = 0; } // [11]: sticky, for rounds nobody looks at one by one (pga_branch_loop)
}

__global__ void k_mail_flush(const int64_t *dcnt, int64_t *host_box)
{
	if (threadIdx.x < 16) sys_store(&host_box[threadIdx.x], dcnt[threadIdx.x]);
}

struct ZeroList { void *p[4]; unsigned long long dwords[4]; };
// several small clears in one launch (each hipMemsetAsync is a launch of its own; a round needs a dozen of them)
__global__ __launch_bounds__(BLOCK) void k_zero_multi(ZeroList z)
{
	unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		if (i < z.dwords[k]) { ((uint32_t *)z.p[k])[i] = 0; return; }
		i -= z.dwords[k];
	}
}

__device__ __forceinline__ int genome_of(const int32_t *goff, int n_genome, int i) // last g with goff[g] <= i
{
	int lo = 0, hi = n_genome;
	while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (goff[mid] <= i) lo = mid; else hi = mid; }
	return lo;
}

// Wave-wide reductions without the LDS crossbar (__shfl_xor compiles to ds_bpermute_b32): four DPP row shifts leave every row's
// total in its last lane, four v_readlane collect them.  The result is wave-uniform.  All 64 lanes must be active.
#define PGA_DPP_SHR(x, n) __builtin_amdgcn_update_dpp(0, (x), 0x110 | (n), 0xf, 0xf, false) /* row_shr:n, 0 where the row ends */

__device__ __forceinline__ int wave_sum(int v)
{
	v += PGA_DPP_SHR(v, 1); v += PGA_DPP_SHR(v, 2); v += PGA_DPP_SHR(v, 4); v += PGA_DPP_SHR(v, 8);
	return __builtin_amdgcn_readlane(v, 15) + __builtin_amdgcn_readlane(v, 31) + __builtin_amdgcn_readlane(v, 47) + __builtin_amdgcn_readlane(v, 63);
}

#define PGA_DPP_SHR64(v, n) do { \
	const unsigned lo_ = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v), 0x110 | (n), 0xf, 0xf, false); \
	const unsigned hi_ = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)((v) >> 32), 0x110 | (n), 0xf, 0xf, false); \
	(v) += (unsigned long long)hi_ << 32 | lo_; } while (0)

__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long v)
{
	PGA_DPP_SHR64(v, 1); PGA_DPP_SHR64(v, 2); PGA_DPP_SHR64(v, 4); PGA_DPP_SHR64(v, 8);
	unsigned long long s = 0;
#pragma unroll
	for (int l = 15; l < 64; l += 16)
		s += (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l) << 32 | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
	return s;
}

#define PGA_DPP_MAX(v, n) do { const int t_ = __builtin_amdgcn_update_dpp((v), (v), 0x110 | (n), 0xf, 0xf, false); (v) = (v) > t_ ? (v) : t_; } while (0) /* own value where the row ends */

__device__ __forceinline__ int wave_max(int v)
{
	PGA_DPP_MAX(v, 1); PGA_DPP_MAX(v, 2); PGA_DPP_MAX(v, 4); PGA_DPP_MAX(v, 8);
	const int a = __builtin_amdgcn_readlane(v, 15), b = __builtin_amdgcn_readlane(v, 31), c = __builtin_amdgcn_readlane(v, 47), e = __builtin_amdgcn_readlane(v, 63);
	const int ab = a > b ? a : b, ce = c > e ? c : e;
	return ab > ce ? ab : ce;
}

// where a tie-order hazard (h2_cm_tie / h2_cs_tie / h3_dom_tie) happened: contig-segment ids, at most PGA_HAZARD_CAP of them (counter: dcnt[14])
__device__ __forceinline__ void hz_note(int64_t *cnt14, int32_t *list, int seg)
{
	const unsigned long long at = atomicAdd((unsigned long long *)cnt14, 1ull);
	if (at < (unsigned long long)PGA_HAZARD_CAP) list[at] = seg;
}

// Half-arc records (k_genes.hpp): per hit, in gene-major order, its successor (hf*) and predecessor (hb*) adjacency of the current
// walk: key word = round tag << 21 | target vertex (gene << 1 | rev) in hfk / hbk, payload {distance, score of this hit, score
// of the other hit, 0} in hfp / hbp.
constexpr uint32_t HA_NONE = 0x1fffffu;  // "no adjacency" target (gene ids stay below 2^20 - 1)
constexpr int HA_TAG_SHIFT = 21;
constexpr uint32_t HA_TAG_MAX = 0x7ffu;
__device__ __forceinline__ bool hx_valid(uint32_t x, uint32_t tag) { return (x >> HA_TAG_SHIFT) == tag && (x & HA_NONE) != HA_NONE; }
// a hit is walkable in this round exactly when the walk wrote its predecessor key in this round
__device__ __forceinline__ bool hx_walk(uint32_t x, uint32_t tag) { return (x >> HA_TAG_SHIFT) == tag; }
