// k_common.hpp -- small device-side helpers shared by every kernel file.
// Included by pga_backend.hip (one translation unit); uses the context types, BLOCK / WAVE and dev_prims.hpp from there.
#pragma once

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
	if (threadIdx.x < 16) host_box[threadIdx.x] = dcnt[threadIdx.x];
}

__global__ void k_mail_flush(const int64_t *dcnt, int64_t *host_box)
{
	if (threadIdx.x < 16) host_box[threadIdx.x] = dcnt[threadIdx.x];
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
