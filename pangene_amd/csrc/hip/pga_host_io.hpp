// pga_host_io.hpp -- fetch / put / download, the gene matrix, timing hooks, the copy-kernel calibration, pga_reserve.
// Host side of the device ABI (include/pangene_hip.h); included by pga_backend.hip (one translation unit), in this order.
#pragma once


extern "C" int pga_sync(pga_ctx_t *c)
{
	TRY(sync_st(c));
	if (c->z_early) TRY(early_index(c)); // (the host computes next: the device may as well)
	return 0;
}

extern "C" int pga_fetch_later(pga_ctx_t *c, const void *src_backend, size_t nbytes, const void **host_view)
{
	if (c->h_stage_cap < nbytes) {
		if (c->h_stage) HIPCHK(hipStreamSynchronize(c->st));
		c->h_stage = c->pin.get(nbytes + nbytes / 2 + 256);
		if (!c->h_stage) return PGA_ERR_NOMEM;
		c->h_stage_cap = nbytes + nbytes / 2 + 256;
	}
	*host_view = c->h_stage;
	if (nbytes) HIPCHK(hipMemcpyAsync(c->h_stage, src_backend, nbytes, hipMemcpyDeviceToHost, c->st));
	return 0;
}

// the wait of a fetch.  Armed by pga_vtx_partials (pga_ctx::z_early): the gene-major index is queued behind the copy, and the wait is for the copy alone
static int fetch_wait(pga_ctx *c)
{
	if (!c->z_early) return sync_st(c);
	if (!c->z_ev) HIPCHK(hipEventCreateWithFlags(&c->z_ev, hipEventDisableTiming));
	HIPCHK(hipEventRecord(c->z_ev, c->st));
	TRY(early_index(c)); // (sync_epoch stays: nobody has waited for the STREAM)
	for (unsigned long long it = 1;; ++it) { // (as sync_st: poll for a while, then park)
		const hipError_t e = hipEventQuery(c->z_ev);
		if (e == hipSuccess) return 0;
		if (e != hipErrorNotReady) HIPCHK(e);
		__builtin_ia32_pause();
		if ((it & 0x3ff) == 0) {
			static thread_local timespec t0 = { 0, 0 }; timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
			if (it == 0x400) t0 = t;
			if ((t.tv_sec - t0.tv_sec) + (t.tv_nsec - t0.tv_nsec) * 1e-9 > 200e-6) break;
		}
	}
	HIPCHK(hipEventSynchronize(c->z_ev));
	return 0;
}

extern "C" int pga_fetch(pga_ctx_t *c, void *dst_host, const void *src_backend, size_t nbytes)
{
	if (nbytes == 0) return 0;
	// dst_host is caller memory, as a rule pageable: a copy straight into it makes the runtime stage it, or pin and unpin the
	// pages (megabytes: milliseconds, part of them charged to whatever runtime call comes next).  Up to a few megabytes the data
	// lands in a pinned buffer of the context first.
	if (nbytes <= ((size_t)2 << 20)) {
		if (c->h_fetch_cap < nbytes) {
			c->h_fetch = c->pin.get(nbytes + nbytes / 2 + 256);
			if (!c->h_fetch) return PGA_ERR_NOMEM;
			c->h_fetch_cap = nbytes + nbytes / 2 + 256;
		}
		HIPCHK(hipMemcpyAsync(c->h_fetch, src_backend, nbytes, hipMemcpyDeviceToHost, c->st));
		TRY(fetch_wait(c));
		memcpy(dst_host, c->h_fetch, nbytes);
		return 0;
	}
	HIPCHK(hipMemcpyAsync(dst_host, src_backend, nbytes, hipMemcpyDeviceToHost, c->st));
	return fetch_wait(c);
}

extern "C" int pga_put(pga_ctx_t *c, void *dst_backend, const void *src_host, size_t nbytes)
{
	if (nbytes == 0) return 0;
	// src_host is caller memory.  Small pieces (the votes, counts and sizes a sharded run puts in front of its collectives) go through
	// the pinned staging area: the bytes are the library's when the call returns, the copy runs in stream order, nobody waits
	if (nbytes <= ((size_t)64 << 10)) return stage_upload(c, dst_backend, src_host, nbytes);
	HIPCHK(hipMemcpyAsync(dst_backend, src_host, nbytes, hipMemcpyHostToDevice, c->st));
	return sync_st(c);
}

extern "C" int pga_copy(pga_ctx_t *c, void *dst_backend, const void *src_backend, size_t nbytes)
{
	if (nbytes == 0) return 0;
	HIPCHK(hipMemcpyAsync(dst_backend, src_backend, nbytes, hipMemcpyDeviceToDevice, c->st));
	return sync_st(c);
}

extern "C" int pga_scratch(pga_ctx_t *c, size_t nbytes, void **ptr)
{
	*ptr = c->pool.get(S_SCRATCH, nbytes);
	return *ptr ? 0 : PGA_ERR_NOMEM;
}

extern "C" int pga_download(pga_ctx_t *c, const pga_hit_state_t *o)
{
	const int N = c->N;
	if (N == 0) return 0;
	if (o->flt_x_bits) {
		unsigned long long *bits = (unsigned long long *)c->pool.get(S_MISC, sizeof(uint64_t) * (size_t)((N + 63) / 64) + 16);
		if (!bits) return PGA_ERR_NOMEM;
		hipLaunchKernelGGL(k_flt_bits, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->flags, N, bits);
		HIPCHK(hipMemcpyAsync(o->flt_x_bits, bits, sizeof(uint64_t) * (size_t)((N + 63) / 64), hipMemcpyDeviceToHost, c->st));
		if (!o->flags && !o->rank && !o->score_dom && !o->pid_dom && !o->pid_dom0 && !o->pos_x && !o->pos_y) return sync_st(c);
	}
	int32_t *dl = (int32_t *)c->pool.get(S_DL, sizeof(int32_t) * 7 * (size_t)N);
	if (!dl) return PGA_ERR_NOMEM;
	hipLaunchKernelGGL(k_to_file, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->fidx, c->gnm, c->goff, c->flags, c->rank, c->sdom, c->pdom, c->pdom0, c->yperm, N,
	                   (uint32_t *)dl, dl + (size_t)N, dl + 2 * (size_t)N, dl + 3 * (size_t)N, dl + 4 * (size_t)N, dl + 5 * (size_t)N, dl + 6 * (size_t)N);
	void *dst[7] = { o->flags, o->rank, o->score_dom, o->pid_dom, o->pid_dom0, o->pos_x, o->pos_y };
	for (int k = 0; k < 7; ++k)
		if (dst[k]) HIPCHK(hipMemcpyAsync(dst[k], dl + (size_t)k * N, sizeof(int32_t) * (size_t)N, hipMemcpyDeviceToHost, c->st));
	return sync_st(c);
}

extern "C" int pga_ctg_counts(pga_ctx_t *c, int32_t *cnt)
{
	const size_t nb = sizeof(int32_t) * (size_t)std::max(1, c->n_seg_ctg);
	int32_t *d = (int32_t *)c->pool.get(S_MISC, nb + 16);
	if (!d) return PGA_ERR_NOMEM;
	HIPCHK(hipMemsetAsync(d, 0, nb, c->st));
	if (c->N) hipLaunchKernelGGL(k_ctg_counts, dim3(nblk(c->N)), dim3(BLOCK), 0, c->st, c->flags, c->seg, c->N, d);
	return c->n_seg_ctg ? pga_fetch(c, cnt, d, sizeof(int32_t) * (size_t)c->n_seg_ctg) : sync_st(c);
}

extern "C" int pga_gene_matrix(pga_ctx_t *c, const int32_t *asm_of_ctg, int32_t n_asm, int32_t n_seg, int32_t *mat)
{
	if (n_seg != c->n_seg || n_asm < 0) return PGA_ERR_ARG;
	const size_t nm = (size_t)n_seg * (size_t)n_asm, nc = (size_t)std::max(1, c->n_seg_ctg);
	int32_t *d = (int32_t *)c->pool.get(S_MISC, sizeof(int32_t) * (nm + nc) + 64);
	if (!d) return PGA_ERR_NOMEM;
	if (nm == 0) return 0;
	HIPCHK(hipMemsetAsync(d, 0, sizeof(int32_t) * nm, c->st));
	TRY(upload(c, d + nm, asm_of_ctg, (size_t)c->n_seg_ctg));
	if (c->N) hipLaunchKernelGGL(k_gene_matrix, dim3(nblk(c->N)), dim3(BLOCK), 0, c->st, c->flags, c->seg, c->gid, c->g2s, c->N, d + nm, n_asm, d);
	return pga_fetch(c, mat, d, sizeof(int32_t) * nm);
}

extern "C" int pga_hazards(pga_ctx_t *c, pga_hazard_t *out)
{
	HIPCHK(hipMemcpyAsync(c->h_cnt, c->dcnt, 16 * sizeof(int64_t), hipMemcpyDeviceToHost, c->st));
	TRY(sync_st(c));
	out->h1_head_tie = c->h_cnt[4], out->h2_cm_tie = c->h_cnt[5], out->h2_cs_tie = c->h_cnt[6], out->h3_dom_tie = c->h_cnt[7];
	return 0;
}

extern "C" int pga_timing_reset(pga_ctx_t *c)
{
	TRY(sync_st(c));
	c->sync_epoch_reset = c->sync_epoch;
	for (auto &t : c->timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
	c->timed.clear();
	c->timing_on = true;
	{ const char *e = getenv("PANGENE_TIME_ROUNDS"); c->timing_rounds = e && *e == '1'; }
	return 0;
}

extern "C" int pga_timing_get(pga_ctx_t *c, int32_t which, double *total_ms, int64_t *n_launch, int64_t *units)
{
	if (which == 4) { // host waits on the stream since pga_timing_reset (not a kernel class: nothing to wait for)
		if (total_ms) *total_ms = 0;
		if (n_launch) *n_launch = (int64_t)(c->sync_epoch - c->sync_epoch_reset);
		if (units) *units = 0;
		return 0;
	}
	if (which == 7) { // which K1 the sweeps of stage A / pg_post_process run on this upload (measure_list_density): not a timing either
		if (total_ms) *total_ms = c->list_density;
		if (n_launch) *n_launch = c->density_known ? (c->lists_in_lds ? 1 : 0) : -1;
		if (units) *units = c->density_tiles;
		return 0;
	}
	TRY(sync_st(c));
	const int only = which >> 8; // (class | (k + 1) << 8: the k-th timed launch of the class alone)
	which &= 255;
	double ms = 0; int64_t n = 0, u = 0; int k = 0;
	for (auto &t : c->timed) {
		if (t.which != which) continue;
		if (only && ++k != only) continue;
		float f = 0;
		HIPCHK(hipEventElapsedTime(&f, t.a, t.b));
		ms += f, ++n, u += t.units;
	}
	if (total_ms) *total_ms = ms;
	if (n_launch) *n_launch = n;
	if (units) *units = u;
	return 0;
}

// The HBM bandwidth a plain copy reaches on THIS device in THIS process ("calibrate with a copy kernel in the same
// run"): 16 bytes per lane and U of them in flight per lane (the loads of a step are all issued before its stores), `bytes` read and
// `bytes` written per repetition, timed with HIP events; GB/s of read + write.  Round 4's form (one 16-byte item per lane per step,
// grid-stride, at most 8192 workgroups) reached 4.7 TB/s where the guide measured 6.3 with a float4 copy: a calibration that
// undersells the device makes every fraction "of measured" look better than it is, so the best of a few shapes is what is reported.
typedef int pga_v4i __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(BLOCK) void k_copy16(const int4 *__restrict__ src_, int4 *__restrict__ dst_, size_t n16)
{
	const pga_v4i *__restrict__ src = reinterpret_cast<const pga_v4i *>(src_);
	pga_v4i *__restrict__ dst = reinterpret_cast<pga_v4i *>(dst_);
	const size_t step = (size_t)gridDim.x * BLOCK * U;
	for (size_t i0 = (size_t)blockIdx.x * BLOCK * U + threadIdx.x; i0 < n16; i0 += step) {
		pga_v4i v[U];
#pragma unroll
		for (int u = 0; u < U; ++u) if (i0 + (size_t)u * BLOCK < n16) v[u] = NT ? __builtin_nontemporal_load(&src[i0 + (size_t)u * BLOCK]) : src[i0 + (size_t)u * BLOCK];
#pragma unroll
		for (int u = 0; u < U; ++u) if (i0 + (size_t)u * BLOCK < n16) { if (NT) __builtin_nontemporal_store(v[u], &dst[i0 + (size_t)u * BLOCK]); else dst[i0 + (size_t)u * BLOCK] = v[u]; }
	}
}

extern "C" int pga_copy_gbps(size_t bytes, int32_t reps, double *gbps)
{
	int ndev = 0;
	if (gbps == nullptr || hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return PGA_ERR_NO_DEVICE;
	bytes = std::max<size_t>(bytes & ~(size_t)15, (size_t)1 << 20);
	reps = std::max(1, reps);
	void *a = nullptr, *b = nullptr;
	if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) { (void)hipGetLastError(); if (a) (void)hipFree(a); return PGA_ERR_NOMEM; }
	hipStream_t st; hipEvent_t e0, e1;
	HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
	HIPCHK(hipMemsetAsync(a, 1, bytes, st));
	const size_t n16 = bytes / 16;
	int ncu = 256;
	{ int dev = 0, v = 0; if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v; }
	double best = 0;
	static const bool verbose = getenv("PANGENE_TIMING") != nullptr;
	for (int shape = 0; shape < 8; ++shape) {
		const int U = shape & 1 ? 8 : 4, per_cu = shape & 2 ? 16 : 8; const bool nt = (shape & 4) != 0;
		const unsigned grid = (unsigned)std::min<size_t>((n16 + (size_t)BLOCK * U - 1) / ((size_t)BLOCK * U), (size_t)ncu * per_cu);
		double top = 0;
		for (int r = 0; r < reps + 1; ++r) { // (the first one warms)
			HIPCHK(hipEventRecord(e0, st));
			if (U == 4 && !nt) hipLaunchKernelGGL((k_copy16<4, false>), dim3(grid), dim3(BLOCK), 0, st, (const int4 *)a, (int4 *)b, n16);
			else if (U == 8 && !nt) hipLaunchKernelGGL((k_copy16<8, false>), dim3(grid), dim3(BLOCK), 0, st, (const int4 *)a, (int4 *)b, n16);
			else if (U == 4) hipLaunchKernelGGL((k_copy16<4, true>), dim3(grid), dim3(BLOCK), 0, st, (const int4 *)a, (int4 *)b, n16);
			else hipLaunchKernelGGL((k_copy16<8, true>), dim3(grid), dim3(BLOCK), 0, st, (const int4 *)a, (int4 *)b, n16);
			HIPCHK(hipEventRecord(e1, st));
			HIPCHK(hipEventSynchronize(e1));
			float ms = 0;
			HIPCHK(hipEventElapsedTime(&ms, e0, e1));
			if (r > 0 && ms > 0) top = std::max(top, 2.0 * (double)bytes / (ms * 1e-3) / 1e9);
		}
		if (verbose) fprintf(stderr, "[pga_copy_gbps] %d items per lane, %d workgroups per CU, %s stores: %.0f GB/s\n", U, per_cu, nt ? "nontemporal" : "plain", top);
		best = std::max(best, top);
	}
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(st);
	(void)hipFree(a); (void)hipFree(b);
	*gbps = best;
	return 0;
}

extern "C" int pga_reserve(int64_t n_hit, int64_t n_exon, int32_t n_prot, int32_t n_gene, int32_t n_genome, int64_t raw_words)
{
	int ndev = 0;
	if (!dev_cache_on() || n_hit <= 0 || n_hit >= (1 << 30) || n_exon < 0 || n_exon >= INT32_MAX || hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return PGA_ERR_ARG;
	if (g_last_dev.load() >= 0) (void)hipSetDevice(g_last_dev.load()); // (the current device is a property of the thread)
	size_t want[2];
	{
		pga_ctx tmp;
		tmp.N = (int32_t)n_hit, tmp.E = (int32_t)n_exon, tmp.P = n_prot, tmp.Q = n_gene, tmp.n_genome = n_genome;
		(void)plan_persistent(&tmp);
		size_t tot = 0;
		for (auto &e : tmp.plan) tot += e.second;
		want[0] = tot, want[1] = pool_want(n_hit, n_genome, n_prot, n_gene, raw_words);
	}
	static const bool timing = getenv("PANGENE_TIMING") != nullptr;
	timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
	int made = 0;
	void *mine[2] = { nullptr, nullptr }; // what this call has put into the cache, or found there fitting: not to be evicted by its own second block
	{ std::lock_guard<std::mutex> lk(g_dev_mu); ++g_dev_reserving; }
	struct Done { ~Done() { { std::lock_guard<std::mutex> lk(g_dev_mu); --g_dev_reserving; } g_dev_cv.notify_all(); } } done;
	for (int k = 1; k >= 0; --k) { // (the larger one first)
		{
			std::lock_guard<std::mutex> lk(g_dev_mu);
			bool have = false;
			for (const DevBlock &b : g_dev_cache)
				if (!have && b.p != mine[1 - k] && b.dev == cur_dev() && b.cap >= want[k] && b.cap <= 2 * want[k] + ((size_t)64 << 20)) have = true, mine[k] = b.p;
			if (have) continue;
		}
		const size_t padded = (want[k] + want[k] / 8 + ((size_t)64 << 20) - 1) & ~(((size_t)64 << 20) - 1);
		void *q = nullptr;
		if (hipMalloc(&q, padded) != hipSuccess) { (void)hipGetLastError(); continue; }
		++made;
		std::lock_guard<std::mutex> lk(g_dev_mu);
		if (g_dev_cache.size() >= 2) { // the cache holds one context's worth: the smallest block that is not one of this call's makes room
			size_t small = (size_t)-1;
			for (size_t i = 0; i < g_dev_cache.size(); ++i)
				if (g_dev_cache[i].p != mine[0] && g_dev_cache[i].p != mine[1] && (small == (size_t)-1 || g_dev_cache[i].cap < g_dev_cache[small].cap)) small = i;
			if (small != (size_t)-1) { (void)hipFree(g_dev_cache[small].p); g_dev_cache.erase(g_dev_cache.begin() + (long)small); }
		}
		g_dev_cache.push_back(DevBlock{q, padded, cur_dev()});
		mine[k] = q;
	}
	if (timing) { timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1); fprintf(stderr, "[pga_reserve] %lld hits: %.1f + %.1f GB asked for, %d block(s) allocated in %.1f ms\n", (long long)n_hit, want[0] / 1073741824.0, want[1] / 1073741824.0, made, ((t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9) * 1e3); }
	return 0;
}

// ------------------------------------------------------------------------------------------------
// blocks staged before there is a context (include/pangene_hip.h: pga_stage_h2d)
// ------------------------------------------------------------------------------------------------
struct StageEnt { const char *host; size_t bytes; char *dev; size_t cap; hipEvent_t ev; int devno; };
static std::mutex g_stage_mu;
static std::vector<StageEnt> g_stage;      // slabs whose copy stands
static std::vector<StageEnt> g_stage_idle; // device buffers (and their events) waiting for the next slab
static hipStream_t g_stage_st = nullptr; static int g_stage_st_dev = -1;
static const size_t STAGE_IDLE_MAX = 24;   // buffers kept between data sets (64 MiB each as a rule)

extern "C" int pga_stage_h2d(const void *host, size_t bytes)
{
	static const bool off = env_has("PANGENE_STAGE", "0");
	if (off || host == nullptr || bytes == 0) return PGA_ERR_ARG;
	if (g_last_dev.load() >= 0) (void)hipSetDevice(g_last_dev.load()); // (the current device is a property of the thread)
	StageEnt e = { (const char *)host, bytes, nullptr, 0, nullptr, cur_dev() };
	{
		std::lock_guard<std::mutex> lk(g_stage_mu);
		if (g_stage_st == nullptr || g_stage_st_dev != e.devno) {
			if (g_stage_st) (void)hipStreamDestroy(g_stage_st);
			g_stage_st = nullptr;
			if (hipStreamCreateWithFlags(&g_stage_st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); g_stage_st = nullptr; return PGA_ERR_NO_DEVICE; }
			g_stage_st_dev = e.devno;
		}
		for (size_t i = 0; i < g_stage_idle.size(); ++i)
			if (g_stage_idle[i].devno == e.devno && g_stage_idle[i].cap >= bytes) { e.dev = g_stage_idle[i].dev, e.cap = g_stage_idle[i].cap, e.ev = g_stage_idle[i].ev; g_stage_idle.erase(g_stage_idle.begin() + (long)i); break; }
	}
	if (e.dev == nullptr) {
		const size_t cap = (bytes + ((size_t)64 << 20) - 1) & ~(((size_t)64 << 20) - 1);
		if (hipMalloc((void **)&e.dev, cap) != hipSuccess) { (void)hipGetLastError(); return PGA_ERR_NOMEM; }
		e.cap = cap;
		if (hipEventCreateWithFlags(&e.ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(e.dev); return PGA_ERR_NO_DEVICE; }
	}
	std::lock_guard<std::mutex> lk(g_stage_mu); // (one stream: the copies and their events are queued in one order)
	if (hipMemcpyAsync(e.dev, host, bytes, hipMemcpyHostToDevice, g_stage_st) != hipSuccess || hipEventRecord(e.ev, g_stage_st) != hipSuccess) {
		(void)hipGetLastError(); (void)hipStreamSynchronize(g_stage_st); (void)hipEventDestroy(e.ev); (void)hipFree(e.dev);
		return PGA_ERR_NO_DEVICE;
	}
	g_stage.push_back(e);
	return 0;
}

extern "C" void pga_stage_drop(const void *host)
{
	std::vector<StageEnt> gone;
	{
		std::lock_guard<std::mutex> lk(g_stage_mu);
		for (size_t i = 0; i < g_stage.size();)
			if (g_stage[i].host == (const char *)host) { gone.push_back(g_stage[i]); g_stage.erase(g_stage.begin() + (long)i); } else ++i;
	}
	for (StageEnt &e : gone) {
		(void)hipEventSynchronize(e.ev); // (a copy still reading the slab, or a context's copy out of the buffer queued behind it: over before either is used again)
		std::lock_guard<std::mutex> lk(g_stage_mu);
		if (g_stage_idle.size() < STAGE_IDLE_MAX) g_stage_idle.push_back(e);
		else { (void)hipEventDestroy(e.ev); (void)hipFree(e.dev); }
	}
}

// the staged copy that holds [p, p + n), if any (pga_create)
static bool stage_lookup(const char *p, size_t n, const char **dev, hipEvent_t *ev)
{
	std::lock_guard<std::mutex> lk(g_stage_mu);
	for (const StageEnt &e : g_stage)
		if (e.devno == cur_dev() && p >= e.host && p + n <= e.host + e.bytes) { *dev = e.dev + (p - e.host), *ev = e.ev; return true; }
	return false;
}

extern "C" int pga_warm(void)
{
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return PGA_ERR_NO_DEVICE;
	int32_t *p = nullptr;
	HIPCHK(hipMalloc((void **)&p, 256));
	hipLaunchKernelGGL(k_fill_i32, dim3(1), dim3(BLOCK), 0, 0, p, (int64_t)16, 0); // the first launch loads the code object
	HIPCHK(hipDeviceSynchronize());
	(void)hipFree(p);
	return 0;
}
