// dev_prims.hpp -- device-wide building blocks written for gfx950 (wave64): prefix scans and a stable
// LSD radix sort (8-bit digits, ballot-based in-wave ranking).  No library calls (no rocPRIM/hipCUB):
// these are the "contig-segmented radix sort" and "wavefront __ballot/prefix-sum" pieces of the path.
//
// Conventions: 256-thread workgroups (4 waves, one per SIMD), every launch >> 256 workgroups at the
// sizes that matter (1 M hits = 1 k tiles of 1 k for the scans, 512 tiles of 2 k for the sort), all
// global accesses by consecutive lanes are to consecutive addresses (16 B/lane where a thread owns 4
// consecutive 32-bit items).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pgd {

constexpr int WAVE = 64;
constexpr int BLOCK = 256;

// A launch that may turn out to have nothing to do.  pga_branch_loop queues the fifteen rounds of graph.c:301-314 without asking the
// host in between; once a round marks no hit and deletes no segment, the state is a fixed point of the round function -- every later
// round would recompute what is there already (the thresholds of pg_flt_high_occ still tighten: that test runs every round, and when
// it deletes something the rounds are live again).  w[0] = the last round in which a segment was deleted, w[1] = the last round in
// which a hit's weak_br was raised (-1: none); a launch of round r passes min = r - 1 (branch steps: nothing changed in the round
// before) or r (the arc round: nothing changed in this one).  w == NULL: always open.
struct Gate { const int32_t *w; int32_t min; };
__device__ __forceinline__ bool gate_closed(const Gate g) { return g.w != nullptr && (g.w[0] > g.w[1] ? g.w[0] : g.w[1]) < g.min; }

// ------------------------------------------------------------------------------------------------
// wave-level inclusive scan with an arbitrary associative operator
// ------------------------------------------------------------------------------------------------
template <class T, class Op>
__device__ __forceinline__ T wave_scan_incl(T v, Op op, int lane)
{
#pragma unroll
	for (int d = 1; d < WAVE; d <<= 1) {
		T u = v.shfl_up(d);
		if (lane >= d) v = op(u, v);
	}
	return v;
}

// scan element types carry their own shuffle so that structs (segment, value) travel as two registers
struct I32 {
	int32_t v;
	__device__ __forceinline__ I32 shfl_up(int d) const { return I32{__shfl_up(v, d, WAVE)}; }
	__device__ __forceinline__ I32 shfl(int l) const { return I32{__shfl(v, l, WAVE)}; }
};
struct SegMax { // (segment id, running max) for a segmented inclusive max
	int32_t seg, v;
	__device__ __forceinline__ SegMax shfl_up(int d) const { return SegMax{__shfl_up(seg, d, WAVE), __shfl_up(v, d, WAVE)}; }
	__device__ __forceinline__ SegMax shfl(int l) const { return SegMax{__shfl(seg, l, WAVE), __shfl(v, l, WAVE)}; }
};
struct OpSum { __device__ __forceinline__ I32 operator()(I32 a, I32 b) const { return I32{a.v + b.v}; } };
struct OpMax { __device__ __forceinline__ I32 operator()(I32 a, I32 b) const { return I32{a.v > b.v ? a.v : b.v}; } };
constexpr int32_t SEG_EMPTY = INT32_MIN; // two-sided identity of OpSegMax
struct OpSegMax { // a is to the left of b
	__device__ __forceinline__ SegMax operator()(SegMax a, SegMax b) const
	{
		if (b.seg == SEG_EMPTY) return a;
		return SegMax{b.seg, (a.seg == b.seg && a.v > b.v) ? a.v : b.v};
	}
};

// ------------------------------------------------------------------------------------------------
// device-wide scan: tile reduce -> (one-block scan of the tile sums, folded into the next step while tiles are few) ->
// tile scan with carry-in.  A tile is BLOCK * IPT consecutive elements; thread t owns IPT consecutive ones.
// In/Out are functors: T In::operator()(int64 i), void Out::operator()(int64 i, T incl, T excl_or_identity)
// ------------------------------------------------------------------------------------------------
constexpr int IPT = 4;
constexpr int TILE = BLOCK * IPT;

template <class T, class Op>
__device__ __forceinline__ T block_scan_excl(T thread_total, Op op, T identity, T *wave_tot /* LDS[4] */, T *block_total)
{
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	T incl = wave_scan_incl(thread_total, op, lane);
	if (lane == 63) wave_tot[w] = incl;
	__syncthreads();
	T carry = identity;
	for (int k = 0; k < w; ++k) carry = op(carry, wave_tot[k]);
	T prev = incl.shfl_up(1);
	T excl = lane == 0 ? carry : op(carry, prev);
	if (block_total) {
		T tot = wave_tot[0];
		for (int k = 1; k < BLOCK / WAVE; ++k) tot = op(tot, wave_tot[k]);
		*block_total = tot;
	}
	__syncthreads();
	return excl;
}

template <class T, class Op, class In>
__global__ __launch_bounds__(BLOCK) void scan_tile_reduce(In in, int64_t n, T *tile_sum, Op op, T identity, Gate gate)
{
	__shared__ T wave_tot[BLOCK / WAVE];
	if (gate_closed(gate)) return;
	const int64_t base = (int64_t)blockIdx.x * TILE + (int64_t)threadIdx.x * IPT;
	T acc = identity;
#pragma unroll
	for (int k = 0; k < IPT; ++k)
		if (base + k < n) acc = op(acc, in(base + k));
	T tot;
	block_scan_excl(acc, op, identity, wave_tot, &tot);
	if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
}

// exclusive scan of the tile sums, one workgroup, in place.  Every thread takes a contiguous chunk (its loads are independent of each
// other, unlike a loop of 256-wide steps with a carried sum: 36 us -> a few us for the 6000 tiles of a 12 M-hit shard); chunks are
// in order, so Op need not commute.
template <class T, class Op>
__global__ __launch_bounds__(BLOCK) void scan_tile_sums(T *tile_sum, int64_t n_tile, Op op, T identity, Gate gate)
{
	__shared__ T wave_tot[BLOCK / WAVE];
	if (gate_closed(gate)) return;
	const int64_t chunk = (n_tile + BLOCK - 1) / BLOCK, lo = (int64_t)threadIdx.x * chunk, hi = lo + chunk < n_tile ? lo + chunk : n_tile;
	T a = identity;
	for (int64_t i = lo; i < hi; ++i) a = op(a, tile_sum[i]);
	T tot;
	T excl = block_scan_excl(a, op, identity, wave_tot, &tot);
	for (int64_t i = lo; i < hi; ++i) {
		const T v = tile_sum[i];
		tile_sum[i] = excl;
		excl = op(excl, v);
	}
}

// FUSED: tile_excl holds the raw tile sums and every workgroup reduces the ones before its tile itself (in order: Op need
// not commute) -- one launch less, worth it while there are few tiles
template <bool FUSED, class T, class Op, class In, class Out>
__global__ __launch_bounds__(BLOCK) void scan_tile_apply(In in, Out out, int64_t n, const T *tile_excl, Op op, T identity, Gate gate)
{
	__shared__ T wave_tot[BLOCK / WAVE];
	if (gate_closed(gate)) return;
	T carry_in;
	if (FUSED) {
		const int64_t nt = blockIdx.x, chunk = (nt + BLOCK - 1) / BLOCK, lo = (int64_t)threadIdx.x * chunk, hi = lo + chunk < nt ? lo + chunk : nt;
		T a = identity;
		for (int64_t i = lo; i < hi; ++i) a = op(a, tile_excl[i]);
		block_scan_excl(a, op, identity, wave_tot, &carry_in);
	} else carry_in = tile_excl[blockIdx.x];
	const int64_t base = (int64_t)blockIdx.x * TILE + (int64_t)threadIdx.x * IPT;
	T v[IPT];
	T acc = identity;
#pragma unroll
	for (int k = 0; k < IPT; ++k) {
		v[k] = base + k < n ? in(base + k) : identity;
		acc = op(acc, v[k]);
	}
	T excl = block_scan_excl(acc, op, identity, wave_tot, (T *)nullptr);
	T run = op(carry_in, excl);
#pragma unroll
	for (int k = 0; k < IPT; ++k) {
		if (base + k < n) {
			T before = run;
			run = op(run, v[k]);
			out(base + k, run, before);
		}
	}
}

// A short scan in ONE launch (round 6): one workgroup, thread t over a contiguous chunk -- reduce, scan of the 256 partial results, apply.  The lists of an
// order override (a few thousand entries, three scans an override, sixty-six overrides a pass of the isoform-rich 200-assembly set) and the per-segment
// scans took two launches each through the tiled form.
constexpr int SCAN_ONE_IPT = 16;
constexpr int64_t SCAN_ONE_MAX = (int64_t)BLOCK * SCAN_ONE_IPT; // 4 096
template <class T, class Op, class In, class Out>
__global__ __launch_bounds__(BLOCK) void scan_one_block(In in, Out out, int64_t n, Op op, T identity, Gate gate)
{
	__shared__ T wave_tot[BLOCK / WAVE];
	if (gate_closed(gate)) return;
	const int64_t lo = (int64_t)threadIdx.x * SCAN_ONE_IPT;
	T v[SCAN_ONE_IPT]; // (every input of the thread asked for before any is combined: the inputs are gathers as a rule)
#pragma unroll
	for (int k = 0; k < SCAN_ONE_IPT; ++k) v[k] = lo + k < n ? in(lo + k) : identity;
	T acc = identity;
#pragma unroll
	for (int k = 0; k < SCAN_ONE_IPT; ++k) acc = op(acc, v[k]);
	T run = block_scan_excl(acc, op, identity, wave_tot, (T *)nullptr);
#pragma unroll
	for (int k = 0; k < SCAN_ONE_IPT; ++k) {
		if (lo + k >= n) break;
		const T before = run;
		run = op(run, v[k]);
		out(lo + k, run, before);
	}
}

// host-side driver; tile_buf must hold ceil(n/TILE) elements of T
template <class T, class Op, class In, class Out>
static inline void device_scan(In in, Out out, int64_t n, T *tile_buf, Op op, T identity, hipStream_t st, Gate gate = Gate{nullptr, 0})
{
	if (n <= 0) return;
	if (n <= SCAN_ONE_MAX) { hipLaunchKernelGGL((scan_one_block<T, Op, In, Out>), dim3(1), dim3(BLOCK), 0, st, in, out, n, op, identity, gate); return; }
	const int64_t n_tile = (n + TILE - 1) / TILE;
	hipLaunchKernelGGL((scan_tile_reduce<T, Op, In>), dim3((unsigned)n_tile), dim3(BLOCK), 0, st, in, n, tile_buf, op, identity, gate);
	if (n_tile <= 2048) {
		hipLaunchKernelGGL((scan_tile_apply<true, T, Op, In, Out>), dim3((unsigned)n_tile), dim3(BLOCK), 0, st, in, out, n, tile_buf, op, identity, gate);
	} else {
		hipLaunchKernelGGL((scan_tile_sums<T, Op>), dim3(1), dim3(BLOCK), 0, st, tile_buf, n_tile, op, identity, gate);
		hipLaunchKernelGGL((scan_tile_apply<false, T, Op, In, Out>), dim3((unsigned)n_tile), dim3(BLOCK), 0, st, in, out, n, tile_buf, op, identity, gate);
	}
}
static inline int64_t scan_tiles(int64_t n) { return (n + TILE - 1) / TILE + 1; }

// common functors
struct InI32 { const int32_t *p; __device__ __forceinline__ I32 operator()(int64_t i) const { return I32{p[i]}; } };
struct InU32 { const uint32_t *p; __device__ __forceinline__ I32 operator()(int64_t i) const { return I32{(int32_t)p[i]}; } };
struct OutExclI32 { int32_t *p; __device__ __forceinline__ void operator()(int64_t i, I32, I32 ex) const { p[i] = ex.v; } };
struct OutExclU32 { uint32_t *p; __device__ __forceinline__ void operator()(int64_t i, I32, I32 ex) const { p[i] = (uint32_t)ex.v; } };
struct OutInclI32 { int32_t *p; __device__ __forceinline__ void operator()(int64_t i, I32 in, I32) const { p[i] = in.v; } };
struct InSegMax { const int32_t *seg, *val; __device__ __forceinline__ SegMax operator()(int64_t i) const { return SegMax{seg[i], val[i]}; } };
struct OutSegMax { int32_t *p; __device__ __forceinline__ void operator()(int64_t i, SegMax in, SegMax) const { p[i] = in.v; } };

// ------------------------------------------------------------------------------------------------
// stable LSD radix sort, 64-bit keys + 32-bit values, 8 bits per pass, only the bit range that varies.
// Per pass: (1) per-tile digit histogram; (2) exclusive scan of the digit-major [256][n_tile] table;
// (3) scatter with a stable rank: inside a wave the lanes holding the same digit are found with eight
// __ballot()s and ranked with a popcount of the lower lanes, waves and tiles are ordered by the
// scanned tables.  Equal keys therefore keep their input order (needed: file order breaks cs ties,
// emission order = genome order groups equal arcs by genome).
// ------------------------------------------------------------------------------------------------
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = BLOCK * RS_ITEMS; // 2048 keys per workgroup

__global__ __launch_bounds__(BLOCK) void rs_hist(const uint64_t *__restrict__ keys, int64_t n, int shift, uint32_t *__restrict__ table, int n_tile)
{
	__shared__ uint32_t h[256];
	h[threadIdx.x] = 0;
	__syncthreads();
	const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
	for (int k = 0; k < RS_ITEMS; ++k) {
		const int64_t i = base + k * BLOCK + threadIdx.x;
		if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
	}
	__syncthreads();
	table[(int64_t)threadIdx.x * n_tile + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of each digit's row of per-tile counts (one workgroup per digit) + the digit totals.  Together with the
// 256-entry prefix every scatter workgroup computes for itself this replaces a general scan of the 256 x n_tile table
// (three launches) by one launch.
__global__ __launch_bounds__(BLOCK) void rs_rowscan(uint32_t *__restrict__ table, int n_tile, uint32_t *__restrict__ digit_total)
{
	__shared__ I32 wave_tot[BLOCK / WAVE];
	__shared__ uint32_t carry_s;
	uint32_t *row = table + (int64_t)blockIdx.x * n_tile;
	if (threadIdx.x == 0) carry_s = 0;
	__syncthreads();
	for (int base = 0; base < n_tile; base += BLOCK) {
		const int i = base + threadIdx.x;
		const I32 v{i < n_tile ? (int32_t)row[i] : 0};
		I32 tot;
		const I32 excl = block_scan_excl(v, OpSum{}, I32{0}, wave_tot, &tot);
		const uint32_t carry = carry_s;
		if (i < n_tile) row[i] = carry + (uint32_t)excl.v;
		__syncthreads();
		if (threadIdx.x == 0) carry_s = carry + (uint32_t)tot.v;
		__syncthreads();
	}
	if (threadIdx.x == 0) digit_total[blockIdx.x] = carry_s;
}

__global__ __launch_bounds__(BLOCK) void rs_scatter(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                     uint64_t *__restrict__ kout, uint32_t *__restrict__ vout, int64_t n, int shift,
                                                     const uint32_t *__restrict__ table, int n_tile, const uint32_t *__restrict__ digit_total)
{
	__shared__ uint32_t whist[BLOCK / WAVE][256];
	__shared__ uint32_t gbase[256];
	__shared__ I32 wave_tot[BLOCK / WAVE];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
	for (int k = 0; k < BLOCK / WAVE; ++k) whist[k][threadIdx.x] = 0;
	{ // base of digit d = keys with a smaller digit (prefix over the 256 totals) + keys with digit d in earlier tiles
		const I32 excl = block_scan_excl(I32{(int32_t)digit_total[threadIdx.x]}, OpSum{}, I32{0}, wave_tot, (I32 *)nullptr);
		gbase[threadIdx.x] = (uint32_t)excl.v + table[(int64_t)threadIdx.x * n_tile + blockIdx.x];
	}
	__syncthreads();
	const int64_t wbase = (int64_t)blockIdx.x * RS_TILE + (int64_t)w * (WAVE * RS_ITEMS);
	const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
	uint64_t key[RS_ITEMS];
	uint32_t rnk[RS_ITEMS];
#pragma unroll
	for (int k = 0; k < RS_ITEMS; ++k) {
		const int64_t i = wbase + k * WAVE + lane;
		key[k] = i < n ? kin[i] : ~0ull;
		const uint32_t d = (uint32_t)(key[k] >> shift) & 255u;
		uint64_t peers = ~0ull;
#pragma unroll
		for (int b = 0; b < 8; ++b) {
			const uint64_t bal = __ballot((d >> b) & 1u);
			peers &= ((d >> b) & 1u) ? bal : ~bal;
		}
		const uint32_t before = whist[w][d];
		const uint32_t r = (uint32_t)__popcll(peers & lt);
		__builtin_amdgcn_wave_barrier();
		if (r == 0) whist[w][d] = before + (uint32_t)__popcll(peers);
		__builtin_amdgcn_wave_barrier();
		rnk[k] = before + r;
	}
	__syncthreads();
	{ // exclusive scan over the 4 waves for digit = threadIdx.x
		uint32_t run = 0;
#pragma unroll
		for (int k = 0; k < BLOCK / WAVE; ++k) {
			const uint32_t t = whist[k][threadIdx.x];
			whist[k][threadIdx.x] = run;
			run += t;
		}
	}
	__syncthreads();
#pragma unroll
	for (int k = 0; k < RS_ITEMS; ++k) {
		const int64_t i = wbase + k * WAVE + lane;
		if (i < n) {
			const uint32_t d = (uint32_t)(key[k] >> shift) & 255u;
			const uint32_t pos = gbase[d] + whist[w][d] + rnk[k];
			kout[pos] = key[k];
			vout[pos] = vin[i];
		}
	}
}

struct RadixBufs { // caller-provided temporaries
	uint64_t *k_alt; uint32_t *v_alt; uint32_t *table; int32_t *tile_buf;
};
static inline int64_t rs_tiles(int64_t n) { return (n + RS_TILE - 1) / RS_TILE; }
static inline int64_t rs_table_len(int64_t n) { return 256 * rs_tiles(n); }

// Sorts (keys, vals) by bits [0, n_bits) of the key.  Returns through *k_res / *v_res which of the two
// buffer pairs holds the result (ping-pong).
static inline void device_radix_sort(uint64_t *keys, uint32_t *vals, int64_t n, int n_bits, const RadixBufs &b,
                                     uint64_t **k_res, uint32_t **v_res, hipStream_t st)
{
	uint64_t *ki = keys, *ko = b.k_alt;
	uint32_t *vi = vals, *vo = b.v_alt;
	if (n > 0) {
		const int n_tile = (int)rs_tiles(n);
		for (int shift = 0; shift < n_bits; shift += 8) {
			hipLaunchKernelGGL(rs_hist, dim3((unsigned)n_tile), dim3(BLOCK), 0, st, ki, n, shift, b.table, n_tile);
			hipLaunchKernelGGL(rs_rowscan, dim3(256), dim3(BLOCK), 0, st, b.table, n_tile, (uint32_t *)b.tile_buf);
			hipLaunchKernelGGL(rs_scatter, dim3((unsigned)n_tile), dim3(BLOCK), 0, st, ki, vi, ko, vo, n, shift, b.table, n_tile, (const uint32_t *)b.tile_buf);
			uint64_t *tk = ki; ki = ko; ko = tk;
			uint32_t *tv = vi; vi = vo; vo = tv;
		}
	}
	*k_res = ki, *v_res = vi;
}

} // namespace pgd
