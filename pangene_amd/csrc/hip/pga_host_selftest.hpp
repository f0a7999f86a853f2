// pga_host_selftest.hpp -- self-tests of the device primitives and the counter calibration kernels (tests/, profiles/tools/calibrate.py).
// Host side of the device ABI (include/pangene_hip.h); included by pga_backend.hip (one translation unit), in this order.
#pragma once


// ------------------------------------------------------------------------------------------------
// self-test hooks for the device primitives (tests/test_prims_gpu.py): sort / scan arbitrary host data
// ------------------------------------------------------------------------------------------------
extern "C" int pga_selftest_sort(uint64_t *keys, uint32_t *vals, int64_t n, int32_t n_bits)
{
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return PGA_ERR_NO_DEVICE;
	uint64_t *ka, *kb; uint32_t *va, *vb, *table; int32_t *tile;
	HIPCHK(hipMalloc((void **)&ka, sizeof(uint64_t) * (size_t)(n + 1))); HIPCHK(hipMalloc((void **)&kb, sizeof(uint64_t) * (size_t)(n + 1)));
	HIPCHK(hipMalloc((void **)&va, sizeof(uint32_t) * (size_t)(n + 1))); HIPCHK(hipMalloc((void **)&vb, sizeof(uint32_t) * (size_t)(n + 1)));
	HIPCHK(hipMalloc((void **)&table, sizeof(uint32_t) * (size_t)(rs_table_len(n) + 1)));
	HIPCHK(hipMalloc((void **)&tile, tile_buf_bytes(n)));
	HIPCHK(hipMemcpy(ka, keys, sizeof(uint64_t) * (size_t)n, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(va, vals, sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice));
	RadixBufs b = { kb, vb, table, tile };
	uint64_t *kr; uint32_t *vr;
	device_radix_sort(ka, va, n, n_bits, b, &kr, &vr, 0);
	HIPCHK(hipDeviceSynchronize());
	HIPCHK(hipMemcpy(keys, kr, sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(vals, vr, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost));
	(void)hipFree(ka); (void)hipFree(kb); (void)hipFree(va); (void)hipFree(vb); (void)hipFree(table); (void)hipFree(tile);
	return 0;
}

// cross-shard arc merge on host data: `gathered` holds W slots of slot_sz entries (count[r] valid, sorted by x, unique keys)
extern "C" int pga_selftest_merge(const pga_arc_part_t *gathered, const int64_t *count, int32_t W, int64_t slot_sz, pga_arc_part_t *out, int64_t *n_out)
{
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return PGA_ERR_NO_DEVICE;
	pga_ctx c; // a bare context: stream, counters, pool
	HIPCHK(hipStreamCreateWithFlags(&c.st, hipStreamNonBlocking));
	TRY(dalloc(&c, &c.dcnt, 16)); TRY(dalloc_commit(&c));
	HIPCHK(hipHostMalloc((void **)&c.h_cnt, 16 * sizeof(int64_t), hipHostMallocDefault));
	pga_arc_part_t *dg = nullptr, *res = nullptr;
	HIPCHK(hipMalloc((void **)&dg, sizeof(pga_arc_part_t) * (size_t)(W * slot_sz + 1)));
	HIPCHK(hipMemcpy(dg, gathered, sizeof(pga_arc_part_t) * (size_t)(W * slot_sz), hipMemcpyHostToDevice));
	int rc = pga_arc_merge(&c, dg, count, W, slot_sz, &res, n_out);
	if (rc == 0 && *n_out) rc = hipMemcpyAsync(out, res, sizeof(pga_arc_part_t) * (size_t)*n_out, hipMemcpyDeviceToHost, c.st) == hipSuccess ? 0 : PGA_ERR_NO_DEVICE;
	(void)hipStreamSynchronize(c.st);
	(void)hipFree(dg); (void)hipFree(c.dcnt); (void)hipHostFree(c.h_cnt);
	c.pool.release();
	(void)hipStreamDestroy(c.st);
	return rc;
}

// ------------------------------------------------------------------------------------------------
// Calibration of the rocprofv3 memory counters (profiles/tools/calibrate.py): kernels with KNOWN byte counts in the access patterns the
// path's kernels use -- coalesced streams of 4 and 16 bytes per lane, 4- and 16-byte gathers / scatters through a permutation (inside
// windows of `window` items, or over the whole array) -- so that FETCH_SIZE / WRITE_SIZE can be turned into bytes per pattern instead
// of by one factor for everything (MI355X_MICROARCH.md calibrates the factor 2 of FETCH_SIZE for wide coalesced reads only).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t cal_perm(int64_t i, int64_t n, int64_t window) // a bijection of [0, n) that permutes inside windows (a power of two)
{
	const int64_t base = i & ~(window - 1), span = base + window <= n ? window : 0; // (the last, partial window stays in place)
	return span ? base + (((i - base) * 40503 + 12345) & (window - 1)) : i;
}
__global__ __launch_bounds__(BLOCK) void k_cal_read16(const int4 *__restrict__ src, int64_t n, int32_t *sink)
{
	int acc = 0;
	for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) { const int4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
	if (acc == 0x7fffffff) sink[0] = acc;
}
__global__ __launch_bounds__(BLOCK) void k_cal_read4(const int32_t *__restrict__ src, int64_t n, int32_t *sink)
{
	int acc = 0;
	for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) acc ^= src[i];
	if (acc == 0x7fffffff) sink[0] = acc;
}
__global__ __launch_bounds__(BLOCK) void k_cal_gather4(const int32_t *__restrict__ src, int64_t n, int64_t window, int32_t *sink)
{
	int acc = 0;
	for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) acc ^= src[cal_perm(i, n, window)];
	if (acc == 0x7fffffff) sink[0] = acc;
}
__global__ __launch_bounds__(BLOCK) void k_cal_gather16(const int4 *__restrict__ src, int64_t n, int64_t window, int32_t *sink)
{
	int acc = 0;
	for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) { const int4 v = src[cal_perm(i, n, window)]; acc ^= v.x ^ v.w; }
	if (acc == 0x7fffffff) sink[0] = acc;
}
__global__ __launch_bounds__(BLOCK) void k_cal_write16(int4 *__restrict__ dst, int64_t n)
{
	for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) dst[i] = make_int4((int)i, 1, 2, 3);
}
__global__ __launch_bounds__(BLOCK) void k_cal_write4(int32_t *__restrict__ dst, int64_t n)
{
	for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) dst[i] = (int)i;
}
__global__ __launch_bounds__(BLOCK) void k_cal_scatter4(int32_t *__restrict__ dst, int64_t n, int64_t window)
{
	for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) dst[cal_perm(i, n, window)] = (int)i;
}
__global__ __launch_bounds__(BLOCK) void k_cal_scatter16(int4 *__restrict__ dst, int64_t n, int64_t window)
{
	for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) dst[cal_perm(i, n, window)] = make_int4((int)i, 1, 2, 3);
}

// runs every pattern once over n items (n * 16 bytes must be past the 256 MiB Infinity Cache to mean anything); the names of the
// kernels carry the pattern, the caller knows the bytes: read16 16 n, read4 4 n, gather4 4 n, gather16 16 n, write16 16 n,
// write4 4 n, scatter4 4 n, scatter16 16 n.  window: a power of two (a genome's worth of items), or 0 = the whole array (rounded down).
extern "C" int pga_selftest_traffic(int64_t n, int64_t window)
{
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return PGA_ERR_NO_DEVICE;
	if (n < 1024) return PGA_ERR_ARG;
	if (window <= 0) { window = 1; while (window * 2 <= n) window *= 2; }
	if (window & (window - 1)) return PGA_ERR_ARG;
	int4 *a = nullptr, *b = nullptr; int32_t *sink = nullptr;
	HIPCHK(hipMalloc((void **)&a, sizeof(int4) * (size_t)n)); HIPCHK(hipMalloc((void **)&b, sizeof(int4) * (size_t)n)); HIPCHK(hipMalloc((void **)&sink, 256));
	HIPCHK(hipMemset(a, 1, sizeof(int4) * (size_t)n)); HIPCHK(hipMemset(b, 2, sizeof(int4) * (size_t)n));
	HIPCHK(hipDeviceSynchronize());
	const unsigned grid = (unsigned)std::min<int64_t>((n + BLOCK - 1) / BLOCK, (int64_t)256 * 64);
	hipLaunchKernelGGL(k_cal_read16, dim3(grid), dim3(BLOCK), 0, 0, (const int4 *)a, n, sink);
	hipLaunchKernelGGL(k_cal_read4, dim3(grid), dim3(BLOCK), 0, 0, (const int32_t *)b, n, sink);
	hipLaunchKernelGGL(k_cal_gather4, dim3(grid), dim3(BLOCK), 0, 0, (const int32_t *)a, n, window, sink);
	hipLaunchKernelGGL(k_cal_gather16, dim3(grid), dim3(BLOCK), 0, 0, (const int4 *)b, n, window, sink);
	hipLaunchKernelGGL(k_cal_write16, dim3(grid), dim3(BLOCK), 0, 0, a, n);
	hipLaunchKernelGGL(k_cal_write4, dim3(grid), dim3(BLOCK), 0, 0, (int32_t *)b, n);
	hipLaunchKernelGGL(k_cal_scatter4, dim3(grid), dim3(BLOCK), 0, 0, (int32_t *)a, n, window);
	hipLaunchKernelGGL(k_cal_scatter16, dim3(grid), dim3(BLOCK), 0, 0, b, n, window);
	HIPCHK(hipDeviceSynchronize());
	(void)hipFree(a); (void)hipFree(b); (void)hipFree(sink);
	return 0;
}

// mode 0: exclusive sum; 1: exclusive max (identity -1); 2: segmented inclusive max with seg[]
extern "C" int pga_selftest_scan(const int32_t *in, const int32_t *seg, int32_t *out, int64_t n, int32_t mode)
{
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return PGA_ERR_NO_DEVICE;
	int32_t *di, *ds, *dout; int64_t *tile;
	HIPCHK(hipMalloc((void **)&di, sizeof(int32_t) * (size_t)(n + 1))); HIPCHK(hipMalloc((void **)&ds, sizeof(int32_t) * (size_t)(n + 1)));
	HIPCHK(hipMalloc((void **)&dout, sizeof(int32_t) * (size_t)(n + 1))); HIPCHK(hipMalloc((void **)&tile, sizeof(int64_t) * (size_t)(scan_tiles(n) + 8)));
	HIPCHK(hipMemcpy(di, in, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice));
	if (seg) HIPCHK(hipMemcpy(ds, seg, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice));
	if (mode == 0) device_scan<I32>(InI32{di}, OutExclI32{dout}, n, (I32 *)tile, OpSum{}, I32{0}, 0);
	else if (mode == 1) device_scan<I32>(InI32{di}, OutExclI32{dout}, n, (I32 *)tile, OpMax{}, I32{-1}, 0);
	else device_scan<SegMax>(InSegMax{ds, di}, OutSegMax{dout}, n, (SegMax *)tile, OpSegMax{}, SegMax{SEG_EMPTY, 0}, 0);
	HIPCHK(hipDeviceSynchronize());
	HIPCHK(hipMemcpy(out, dout, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost));
	(void)hipFree(di); (void)hipFree(ds); (void)hipFree(dout); (void)hipFree(tile);
	return 0;
}
