// pga_host_common.hpp -- waits, sweep launches, small uploads: what every stage of the host side uses.
// Host side of the device ABI (include/pangene_hip.h); included by pga_backend.hip (one translation unit), in this order.
#pragma once


// ================================================================================================
// host side of the ABI
// ================================================================================================
// Waiting for the stream.  hipStreamSynchronize parks the thread (tens of microseconds to come back); the waits of a pass are
// short and many, so the thread polls hipStreamQuery instead.  The runtime's own completion tracking is what makes the results
// visible: kernels in the middle of a stream release their writes at agent scope only, and it is the runtime's end-of-stream
// marker that releases them at system scope -- data a kernel (or a copy kernel) stored into pinned host memory may otherwise
// still sit in the L2 of the XCD that wrote it.  (A doorbell written by a last tiny kernel and polled by the host was faster
// still and WRONG for exactly that reason: its fence covers the L2 of one XCD; one run in a few hundred read stale counters.)
// PANGENE_WAIT=sync selects the plain blocking call.
static int sync_st(pga_ctx *c)
{
	static const bool poll = [] { const char *e = getenv("PANGENE_WAIT"); return !(e && strcmp(e, "sync") == 0); }();
	++c->sync_epoch;
	if (poll) { // poll for up to ~200 us (the waits of a pass are 20-30 us as a rule), then let the runtime park the thread:
		// a rank must not burn a core through a wait of milliseconds (the queued branch rounds; several ranks share a node)
		timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
		for (unsigned long long it = 1;; ++it) {
			const hipError_t e = hipStreamQuery(c->st);
			if (e == hipSuccess) return 0;
			if (e != hipErrorNotReady) HIPCHK(e);
			__builtin_ia32_pause();
			if ((it & 0x3f) == 0) {
				timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
				if ((t.tv_sec - t0.tv_sec) + (t.tv_nsec - t0.tv_nsec) * 1e-9 > 200e-6) break;
			}
		}
	}
	HIPCHK(hipStreamSynchronize(c->st));
	return 0;
}

static int bits_for(uint32_t maxv) { int b = 1; while (b < 32 && (maxv >> b)) ++b; return b; }

static int make_sweep_view(pga_ctx *c, SweepView *v)
{
	v->A = c->recA, v->B = c->recB, v->C = c->recC, v->sori = c->sori, v->exon = c->exon, v->flags = c->flags, v->pdom = c->pdom, v->sdom = c->sdom, v->pdom0 = c->pdom0;
	v->n = c->N, v->min_ov = c->par.min_ov_ratio, v->check_strand = c->par.check_strand, v->hz = c->dcnt + 4, v->stage_c = c->any_multi;
	v->init_dom = c->sweep_init ? 1 : 0;
	v->literal = c->exon_regular && getenv("PANGENE_MERGE_LITERAL") == nullptr ? 0 : 1;
	v->gate = c->gate;
	v->xmap = nullptr;
	v->slow_cnt = nullptr, v->slow_list = (int32_t *)c->pool.get(S_SLOW, sizeof(int32_t) * (size_t)c->N);
	v->hz_list = (int32_t *)c->pool.get(S_HZLIST, sizeof(int32_t) * PGA_HAZARD_CAP);
	if (!v->slow_list || !v->hz_list) return PGA_ERR_NOMEM;
	return 0;
}

// Which K1 the sweeps of stage A and pg_post_process run on this upload: k_list_density adds up, over a sample of ~256 tiles, the
// exons k_sweep would copy into LDS.  Fewer than SW_LEAN_BELOW a tile on average: the lists stay in global memory (k_sweep_lean, twice the
// workgroups on a CU).  The keys of an upload never change, so this is asked once (one host wait, ~20 us); the results are the same either way.
// PANGENE_SWEEP_LISTS=lds|global fixes the choice (tests run every case both ways).
constexpr int SW_LEAN_BELOW = 1024;
static int measure_list_density(pga_ctx *c)
{
	c->density_known = true;
	if (c->N == 0) return 0;
	const char *e = getenv("PANGENE_SWEEP_LISTS");
	if (e && (strcmp(e, "lds") == 0 || strcmp(e, "global") == 0)) { c->lists_in_lds = e[0] == 'l'; return 0; }
	const int nt = (int)nblk(c->N, SW_TILE), stride = std::max(1, nt / 256), ns = (nt + stride - 1) / stride;
	HIPCHK(hipMemsetAsync(c->dcnt + 16, 0, sizeof(int64_t), c->st));
	hipLaunchKernelGGL(k_list_density, dim3(ns), dim3(SW_TILE), 0, c->st, c->recA, c->recC, c->N, stride, c->dcnt + 16);
	HIPCHK(hipMemcpyAsync(c->h_cnt + 16, c->dcnt + 16, sizeof(int64_t), hipMemcpyDeviceToHost, c->st));
	{ const int rc = sync_st(c); if (rc) return rc; }
	c->lists_in_lds = c->h_cnt[16] >= (int64_t)SW_LEAN_BELOW * ns, c->list_density = (double)c->h_cnt[16] / ns, c->density_tiles = ns;
	if (getenv("PANGENE_TIMING")) fprintf(stderr, "[pga] exon lists of the sweeps: %.0f a tile to stage (%d tiles sampled) -> %s\n", (double)c->h_cnt[16] / ns, ns, c->lists_in_lds ? "LDS (k_sweep)" : "global memory (k_sweep_lean)");
	return 0;
}

static void pack_records(pga_ctx *c)
{
	if (c->N) hipLaunchKernelGGL(k_pack_rec, dim3(nblk(c->N)), dim3(BLOCK), 0, c->st, c->seg, c->cs, c->ce, c->pm, c->rk, c->gid, c->cds, c->rank, c->nex, c->offx,
	                             c->pid, c->sori, c->N, c->recA, c->recB, c->recC, c->flags);
}

template <int MODE> static int launch_sweep(pga_ctx *c, int timed_which)
{
	SweepView v;
	if (c->N == 0) return 0;
	{ const int rc = make_sweep_view(c, &v); if (rc) return rc; }
	TimedLaunch t; t.which = timed_which; t.units = c->N;
	constexpr int reps = 1;
	static_assert(MODE == 0 || MODE == 1 || MODE == 3, "sweep modes");
	static const bool no_compact = env_has("PANGENE_LIVE", "fullsweep"); // (tests: live lists, but the sweeps over every record)
	if (MODE == 0 && c->live_on && c->z_valid && !no_compact) v.A = c->cA, v.B = c->cB, v.C = c->cC, v.n = c->NL, v.xmap = c->cx; // the sweeps of the rounds: the members' records (see SweepView::xmap; z_valid: a cs override under -S drops the lists -- until they are built again the compact records are not what the X order holds)
	const int n_sw = v.n; // (0: one workgroup of sentinels -- the k_sweep_slow behind it still has counters to clear)
	const bool timed = (timed_which == 0 || timed_which == 1) && c->timing_on; // (the stage-C sweeps are not timed one by one: two events per launch cost ~10 us of queue time)
	if (timed) {
		HIPCHK(hipEventCreate(&t.a)); HIPCHK(hipEventCreate(&t.b));
		if (reps != 1) HIPCHK(hipEventRecord(t.a, c->st));
	}
	c->walk_valid = false, c->ha_valid = false;
	const int nt = std::max(1, (int)nblk(n_sw, SW_TILE));
	v.prof = nullptr; v.dbg = 0;
#ifdef PGA_SW_PROFILE
	{ const char *e = getenv("PGA_SW_DBG"); v.dbg = e ? atoi(e) : 0; }
	HIPCHK(hipMalloc((void **)&v.prof, sizeof(long long) * SW_NSTAMP * SW_NW * (size_t)nt)); HIPCHK(hipMemset(v.prof, 0, sizeof(long long) * SW_NSTAMP * SW_NW * (size_t)nt));
#endif
	for (int rep = 0; rep < reps; ++rep) {
		v.slow_cnt = c->dcnt + 12 + (c->sweep_seq & 1);
		// a timed launch carries its own start/stop events: they take the dispatch's begin and end time stamps, i.e. the
		// duration of k_sweep itself, the figure rocprofv3 --kernel-trace reports for it
		hipEvent_t ea = timed && reps == 1 ? t.a : nullptr, eb = timed && reps == 1 ? t.b : nullptr;
		if (c->any_multi && MODE != 0 && !c->lists_in_lds) hipExtLaunchKernelGGL((k_sweep_lean<MODE == 0 ? 1 : MODE>), dim3(nt), dim3(SW_TILE), 0, c->st, ea, eb, 0, v);
		else if (c->any_multi) hipExtLaunchKernelGGL((k_sweep<MODE, true>), dim3(nt), dim3(SW_TILE), 0, c->st, ea, eb, 0, v);
		else hipExtLaunchKernelGGL((k_sweep<MODE, false>), dim3(nt), dim3(SW_TILE), 0, c->st, ea, eb, 0, v);
		hipLaunchKernelGGL((k_sweep_slow<MODE>), dim3((unsigned)std::min<int64_t>(2 * c->n_cu, std::max<int64_t>(64, nblk(n_sw)))), dim3(BLOCK), 0, c->st, v, (long long *)(c->dcnt + 12 + ((c->sweep_seq + 1) & 1)), MODE == 0 ? c->ga_ctl : (int32_t *)nullptr); // (grid-stride over a list whose length only the device knows)
		++c->sweep_seq;
	}
#ifdef PGA_SW_PROFILE
	{
		std::vector<long long> hp((size_t)SW_NSTAMP * SW_NW * nt);
		HIPCHK(hipStreamSynchronize(c->st));
		HIPCHK(hipMemcpy(hp.data(), v.prof, hp.size() * sizeof(long long), hipMemcpyDeviceToHost));
		(void)hipFree(v.prof);
		double d[SW_NSTAMP] = { 0 };
		for (size_t w = 0; w < (size_t)SW_NW * nt; ++w)
			for (int k = 1; k < 10; ++k) { const long long x = hp[w * SW_NSTAMP + k], y = hp[w * SW_NSTAMP + k - 1]; if (x && y) d[k] += (double)(x - y); }
		const double q = 1.0 / ((double)SW_NW * nt);
		{ double mx = 0, sm = 0; for (size_t w = 0; w < (size_t)SW_NW * nt; ++w) mx += (double)hp[w * SW_NSTAMP + 10], sm += (double)hp[w * SW_NSTAMP + 11]; fprintf(stderr, "[sweep<%d> epilogue merges: steps of the longest lane %.1f, of all lanes %.1f per wave]\n", MODE, mx * q, sm * q); }
		fprintf(stderr, "[sweep<%d> profile, n %d, ticks/wave] records->LDS %.0f | barrier %.0f | want+scan %.0f | barrier+offsets+sources %.0f | barrier+gather %.0f | barrier %.0f | runs %.0f | list+eval %.0f | finish %.0f\n", MODE, c->N,
		        d[1] * q, d[2] * q, d[3] * q, d[4] * q, d[5] * q, d[6] * q, d[7] * q, d[8] * q, d[9] * q);
	}
#endif
	if (timed) {
		if (reps != 1) HIPCHK(hipEventRecord(t.b, c->st));
		c->timed.push_back(t);
	}
	return 0;
}

static int radix_sort_pool(pga_ctx *c, uint64_t *keys, uint32_t *vals, int64_t n, int n_bits, uint64_t **kres, uint32_t **vres)
{
	RadixBufs b;
	if (n > std::max<int64_t>(2 * (int64_t)c->N + 2, (int64_t)c->P + 2)) return PGA_ERR_ARG; // work buffers are sized once, in create
	b.k_alt = (uint64_t *)c->pool.get(S_KEY_B, 0);
	b.v_alt = (uint32_t *)c->pool.get(S_VAL_B, 0);
	b.table = (uint32_t *)c->pool.get(S_TABLE, 0);
	b.tile_buf = (int32_t *)c->pool.get(S_TILE, tile_buf_bytes(n));
	if (!b.k_alt || !b.v_alt || !b.table || !b.tile_buf) return PGA_ERR_NOMEM;
	device_radix_sort(keys, vals, n, n_bits, b, kres, vres, c->st);
	return 0;
}

extern "C" void pga_destroy(pga_ctx_t *c)
{
	if (c == nullptr) return;
	if (c->st) (void)hipStreamSynchronize(c->st);
	if (getenv("PANGENE_TIMING")) { // how well the one-allocation plan of create_impl fitted the run
		size_t n_own = 0, b_own = 0;
		for (size_t i = 0; i < c->pool.p.size(); ++i) if (c->pool.p[i] && c->pool.own[i]) ++n_own, b_own += c->pool.cap[i];
		fprintf(stderr, "[pga_destroy] %d hits: temporaries used %.1f of %.1f MB of their arena, %zu slots (%.1f MB) had to be allocated on their own\n",
		        c->N, c->pool.arena_off / 1048576.0, c->pool.arena_cap / 1048576.0, n_own, b_own / 1048576.0);
	}
	for (auto &t : c->timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
	if (c->span_a) (void)hipEventDestroy(c->span_a);
	for (void *q : c->owned) (void)hipFree(q);
	dev_big_free(c->arena, c->arena_cap), c->arena = nullptr;
	c->pool.release();
	for (int k = 0; k < 2; ++k) if (c->ov_ev[k]) { (void)hipEventDestroy(c->ov_ev[k]); c->ov_ev[k] = nullptr; }
	c->pin.release(); // h_cnt, h_stage, h_g2s, h_round, h_ndl live there
	if (c->g2s_done) (void)hipEventDestroy(c->g2s_done);
	if (c->z_ev) (void)hipEventDestroy(c->z_ev);
	if (c->own_stream && c->st) (void)hipStreamDestroy(c->st);
	delete c;
}

static hipStream_t g_active_stream = nullptr; // stream of the live context: collectives of a sharded run are enqueued here

extern "C" void *pga_active_stream(void) { return (void *)g_active_stream; }

extern "C" int pga_set_stream(pga_ctx_t *c, void *hip_stream)
{
	if (c == nullptr) return PGA_ERR_ARG;
	if (c->st) HIPCHK(hipStreamSynchronize(c->st));
	if (c->own_stream && c->st) (void)hipStreamDestroy(c->st);
	c->st = (hipStream_t)hip_stream, c->own_stream = false;
	g_active_stream = c->st;
	return 0;
}

// clears up to four buffers (byte counts are rounded up to whole dwords; every pool buffer has that slack) in one launch
static void zero_multi(pga_ctx *c, void *p0, size_t b0, void *p1 = nullptr, size_t b1 = 0, void *p2 = nullptr, size_t b2 = 0, void *p3 = nullptr, size_t b3 = 0)
{
	ZeroList z = { { p0, p1, p2, p3 }, { (b0 + 3) / 4, (b1 + 3) / 4, (b2 + 3) / 4, (b3 + 3) / 4 } };
	const unsigned long long tot = z.dwords[0] + z.dwords[1] + z.dwords[2] + z.dwords[3];
	if (tot) hipLaunchKernelGGL(k_zero_multi, dim3((unsigned)((tot + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, c->st, z);
}

template <class T> static int upload(pga_ctx *c, T *dst, const T *src, size_t n)
{
	if (n == 0) return 0;
	HIPCHK(hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyHostToDevice, c->st));
	return 0;
}

#define TRY(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

static int stage_upload(pga_ctx *c, void *d0, const void *s0, size_t n0, void *d1 = nullptr, const void *s1 = nullptr, size_t n1 = 0);
