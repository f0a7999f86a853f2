// pga_host_stage_b.hpp -- pg_post_process (graph.c:7-32) and the vertex partials (vertex.c:28-51).
// Host side of the device ABI (include/pangene_hip.h); included by pga_backend.hip (one translation unit), in this order.
#pragma once


extern "C" int pga_post_partials(pga_ctx_t *c, int32_t **max_ori, int64_t **sums)
{
	zero_multi(c, c->max_ori, sizeof(int32_t) * (size_t)std::max(1, c->P), c->sums, sizeof(int64_t) * 6 * (size_t)std::max(1, c->P));
	// (round 6) the sums through LDS where the proteins fit in one or two ranges and a workgroup's stretch is long enough to hold several genomes
	static const bool pp_atomics = env_has("PANGENE_POST", "atomics"); // (tests: the one-atomic-a-contribution form on every shard)
	static const bool pp_force = env_has("PANGENE_POST", "lds");       // (tests: the LDS form on small shards too)
	bool lds_done = false;
	if (c->N && c->P && !pp_atomics && (c->N >= (1 << 19) || pp_force)) {
		const size_t room = (size_t)144 << 10;
		const int p_tiles = (size_t)c->P * 28 <= room ? 1 : (size_t)((c->P + 1) / 2) * 28 <= room ? 2 : 0;
		if (p_tiles) {
			const int PT = (c->P + p_tiles - 1) / p_tiles;
			const size_t lds = (size_t)PT * 28 + 16;
			static size_t lds_set = 0;
			bool ok = true;
			if (lds > lds_set) { ok = hipFuncSetAttribute(reinterpret_cast<const void *>(k_post_part_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess; if (ok) lds_set = lds; else (void)hipGetLastError(); }
			if (ok) {
				const int n_chunk = (int)std::max<int64_t>(1, std::min<int64_t>(c->n_cu, c->N / 32768)); // (a stretch of >= 32 k hits: the reduction is what the launch is for)
				hipLaunchKernelGGL(k_post_part_lds, dim3((unsigned)(n_chunk * p_tiles)), dim3(PP_T), lds, c->st, c->flags, c->pid, c->rank, c->sori, c->sadj, c->nex, c->N, c->P, p_tiles, PT,
				                   c->max_ori, (unsigned long long *)c->sums);
				lds_done = true;
			}
		}
	}
	if (c->N && !lds_done) hipLaunchKernelGGL(k_post_part, dim3(nblk(c->N)), dim3(BLOCK), 0, c->st, c->flags, c->pid, c->rank, c->sori, c->sadj, c->nex, c->N, c->P,
	                             c->max_ori, (unsigned long long *)c->sums);
	if (c->N && c->P) hipLaunchKernelGGL(k_post_count, dim3(nblk(c->P)), dim3(BLOCK), 0, c->st, (unsigned long long *)c->sums, c->P);
	*max_ori = c->max_ori, *sums = c->sums;
	return 0; // no wait: a consumer that is not on this stream calls pga_sync first
}

extern "C" int pga_post_apply(pga_ctx_t *c, const uint8_t *prot_rep, const uint8_t *prot_pj, int64_t *n_pseudo)
{
	c->yrec_valid = false, c->wrec_valid = false;
	uint8_t *d = (uint8_t *)c->pool.get(S_MISC, 2 * (size_t)c->P + 16);
	if (!d) return PGA_ERR_NOMEM;
	c->walk_valid = false, c->ha_valid = false;
	TRY(stage_upload(c, d, prot_rep, (size_t)c->P, d + c->P, prot_pj, (size_t)c->P)); // caller memory
	if (n_pseudo) HIPCHK(hipMemsetAsync(c->dcnt + 2, 0, sizeof(int64_t), c->st));
	if (c->N) hipLaunchKernelGGL(k_post_apply, dim3(nblk(c->N)), dim3(BLOCK), 0, c->st, c->flags, c->pid, c->nex, c->sdom, c->N, c->max_ori, d, d + c->P, n_pseudo ? c->dcnt + 2 : (int64_t *)nullptr);
	if (n_pseudo) { // the count only feeds a log line: nobody waits for it otherwise
		HIPCHK(hipMemcpyAsync(c->h_cnt, c->dcnt, 16 * sizeof(int64_t), hipMemcpyDeviceToHost, c->st));
		TRY(sync_st(c));
		*n_pseudo = c->h_cnt[2];
	}
	return 0;
}

extern "C" int pga_shadow(pga_ctx_t *c, int32_t cal_dom_sc, int32_t *stats)
{
	c->yrec_valid = false, c->wrec_valid = false;
	if (cal_dom_sc) TRY(launch_sweep<1>(c, -1)); else TRY(launch_sweep<0>(c, 2));
	if (stats) {
		int32_t *d_stats = (int32_t *)c->pool.get(S_STATS, sizeof(int32_t) * 4 * (size_t)c->n_genome + 16);
		if (!d_stats) return PGA_ERR_NOMEM;
		HIPCHK(hipMemsetAsync(d_stats, 0, sizeof(int32_t) * 2 * (size_t)c->n_genome + 16, c->st));
		if (c->N) hipLaunchKernelGGL(k_count_shadow, dim3(nblk(c->N)), dim3(BLOCK), 0, c->st, c->flags, c->gnm, c->N, d_stats);
		HIPCHK(hipMemcpyAsync(stats, d_stats, sizeof(int32_t) * 2 * (size_t)c->n_genome, hipMemcpyDeviceToHost, c->st));
		return sync_st(c);
	}
	return 0;
}

extern "C" int pga_set_filter(pga_ctx_t *c, int32_t which)
{
	if (which < 0 || which > 3) return PGA_ERR_ARG;
	c->walk_valid = false, c->ha_valid = false;
	if (c->N) hipLaunchKernelGGL(k_set_filter, dim3(nblk(c->N)), dim3(BLOCK), 0, c->st, c->flags, c->N, which);
	return 0;
}

static int check_invariant(pga_ctx *c, bool flushed = false) // flushed: a k_mail_sum just sent the counters to the host mirror
{
	if (!flushed) hipLaunchKernelGGL(k_mail_flush, dim3(1), dim3(64), 0, c->st, c->dcnt, c->h_box);
	TRY(sync_st(c));
	return c->h_cnt[3] ? PGA_ERR_INVARIANT : 0;
}

extern "C" int pga_vtx_partials(pga_ctx_t *c, int32_t **cnt, uint64_t **records, int64_t *n_records)
{
	const int N = c->N, Q = c->Q, GL = c->n_genome;
	const int64_t wpg = (Q + 31) / 32, n_slot = (int64_t)std::max(1, Q) * VTX_K;
	const int nw = (c->n_genome_global + 63) / 64;
	static const long long first_cap = [] { const char *e = getenv("PANGENE_VTX_SPILL_CAP"); return e && atoll(e) > 0 ? atoll(e) : 65536ll; }(); // (tests shrink it to reach the second attempt)
	long long ovf_cap = first_cap;
	uint32_t *bits = (uint32_t *)c->pool.get(S_BITS, sizeof(uint32_t) * (size_t)(wpg * GL) + 16);
	if (!bits) return PGA_ERR_NOMEM;
	*n_records = 0, *cnt = c->vtx_cnt;
	for (int attempt = 0;; ++attempt) { // the spill area beyond the VTX_K dominator slots per gene grows to what the first attempt counted
		// [dom_tab: n_slot i32][slot: n_slot i32][pair bits: n_slot * nw u64][records: (n_slot + ovf_cap) * (1 + nw) u64]
		const size_t b_tab = sizeof(int32_t) * (size_t)n_slot, b_bits = sizeof(uint64_t) * (size_t)n_slot * (size_t)nw, b_rec = sizeof(uint64_t) * (size_t)(n_slot + ovf_cap) * (size_t)(1 + nw);
		char *blk = (char *)c->pool.get(S_TRIPLES, 2 * b_tab + b_bits + b_rec + 64);
		if (!blk) return PGA_ERR_NOMEM;
		int32_t *dom_tab = (int32_t *)blk, *slot = (int32_t *)(blk + b_tab);
		unsigned long long *pbits = (unsigned long long *)(blk + 2 * b_tab), *rec = (unsigned long long *)(blk + 2 * b_tab + b_bits);
		zero_multi(c, bits, sizeof(uint32_t) * (size_t)(wpg * GL) + 16, c->vtx_cnt, sizeof(int32_t) * 2 * (size_t)std::max(1, Q), c->dcnt, sizeof(int64_t), pbits, b_bits);
		HIPCHK(hipMemsetAsync(dom_tab, 0xff, b_tab, c->st)); // every slot empty (-1)
		HIPCHK(hipMemsetAsync(c->live_cnt, 0, sizeof(int64_t) * LIVE_CNT_N, c->st)); // k_vtx1 counts the hits without flt there
		*records = (uint64_t *)rec;
		if (N == 0) return sync_st(c);
		{ // (round 6) the genes' counts through LDS on shards whose stretches hold several genomes (PANGENE_POST=atomics / lds as for k_post_part_lds)
			static const bool vx_atomics = env_has("PANGENE_POST", "atomics"), vx_force = env_has("PANGENE_POST", "lds");
			const size_t lds = sizeof(int) * 2 * (size_t)std::max(1, Q);
			static size_t lds_set = 0;
			bool ok = !vx_atomics && (N >= (1 << 19) || vx_force) && lds <= ((size_t)144 << 10);
			if (ok && lds > lds_set) { ok = hipFuncSetAttribute(reinterpret_cast<const void *>(k_vtx1_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess; if (ok) lds_set = lds; else (void)hipGetLastError(); }
			if (ok) {
				const int n_chunk = (int)std::max<int64_t>(1, std::min<int64_t>(2 * c->n_cu, N / 32768));
				hipLaunchKernelGGL(k_vtx1_lds, dim3((unsigned)n_chunk), dim3(VX_T), lds, c->st, c->flags, c->gnm, c->gid, c->rank, c->pdom, N, Q, c->vtx_cnt, bits, wpg, c->dcnt, c->live_cnt);
			} else
			hipLaunchKernelGGL(k_vtx1, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->flags, c->gnm, c->gid, c->rank, c->pdom, N, Q, c->vtx_cnt, bits, wpg, c->dcnt, c->live_cnt);
		}
		hipLaunchKernelGGL(k_live_sum, dim3(1), dim3(BLOCK), 0, c->st, (const int64_t *)c->live_cnt, c->dcnt);
		hipLaunchKernelGGL(k_vtx_fold, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->flags, c->gnm, c->gid, c->rank, c->pdom, c->prot_gid, c->ggl, N, bits, wpg,
		                   dom_tab, pbits, nw, rec + n_slot * (1 + nw), ovf_cap, c->dcnt);
		I32 *tile = (I32 *)c->pool.get(S_TILE, tile_buf_bytes(n_slot));
		if (!tile) return PGA_ERR_NOMEM;
		device_scan<I32>(InDomSet{dom_tab}, OutExclI32{slot}, n_slot, tile, OpSum{}, I32{0}, c->st);
		hipLaunchKernelGGL(k_vtx_compact, dim3(nblk(n_slot)), dim3(BLOCK), 0, c->st, dom_tab, slot, n_slot, pbits, nw, rec, c->dcnt, c->h_box);
		TRY(sync_st(c));
		if (c->h_cnt[3]) return PGA_ERR_INVARIANT;
		c->live_hint = c->h_cnt[8];
		const int64_t n_rec = c->h_cnt[10], n_ovf = c->h_cnt[0];
		if (n_ovf > ovf_cap) { // more spilled (genome, gene) cells than there was room for: once more, with room (the reference has no such limit)
			// which dominators win a gene's slots is a race, so the number of spilled cells may differ a little between attempts:
			// the second attempt gets a margin, a third one the upper bound (a spilled cell is at least one hit)
			if (ovf_cap >= (long long)N) return PGA_ERR_RANGE;
			ovf_cap = attempt == 0 ? std::min<long long>(N, 2 * n_ovf + 64) : (long long)N;
			continue;
		}
		if (n_ovf) { // the spilled single-genome records follow the folded ones
			HIPCHK(hipMemcpyAsync(rec + n_rec * (1 + nw), rec + n_slot * (1 + nw), sizeof(uint64_t) * (size_t)n_ovf * (size_t)(1 + nw), hipMemcpyDeviceToDevice, c->st));
			TRY(sync_st(c));
		}
		*n_records = n_rec + n_ovf;
		c->z_early = !c->z_valid; // (the caller fetches the records next and then computes for a while: pga_fetch)
		return 0;
	}
}

// Upload out of CALLER memory without waiting for it: the bytes (up to three pieces) are copied into a pinned staging area first,
// so the caller's buffers are free when the call returns and the DMA runs in stream order.
static int stage_upload(pga_ctx *c, void *d0, const void *s0, size_t n0, void *d1, const void *s1, size_t n1)
{
	const size_t a0 = (n0 + 15) & ~(size_t)15, nb = a0 + n1;
	if (nb == 0) return 0;
	if (c->h_g2s_cap < nb) {
		if (c->h_g2s) HIPCHK(hipStreamSynchronize(c->st)); // (the old piece stays in the arena)
		c->h_g2s = (int32_t *)c->pin.get(nb + nb / 2 + 64);
		if (!c->h_g2s) return PGA_ERR_NOMEM;
		c->h_g2s_cap = nb + nb / 2;
	}
	if (!c->g2s_done) HIPCHK(hipEventCreateWithFlags(&c->g2s_done, hipEventDisableTiming));
	else HIPCHK(hipEventSynchronize(c->g2s_done)); // the previous upload out of the staging area (long finished in practice)
	char *h = (char *)c->h_g2s;
	if (n0) memcpy(h, s0, n0);
	if (n1) memcpy(h + a0, s1, n1);
	if (nb <= ((size_t)1 << 20)) { // small: a copy kernel reads the staging area itself (see k_copy_in)
		char *hd = nullptr;
		HIPCHK(hipHostGetDevicePointer((void **)&hd, h, 0));
		CopyIn l = { { d0, d1 }, { (const uint32_t *)hd, (const uint32_t *)(hd + a0) }, { n0, n1 } };
		hipLaunchKernelGGL(k_copy_in, dim3(nblk((std::max(n0, n1) + 3) / 4)), dim3(BLOCK), 0, c->st, l);
	} else {
		if (n0) HIPCHK(hipMemcpyAsync(d0, h, n0, hipMemcpyHostToDevice, c->st));
		if (n1) HIPCHK(hipMemcpyAsync(d1, h + a0, n1, hipMemcpyHostToDevice, c->st));
	}
	HIPCHK(hipEventRecord(c->g2s_done, c->st));
	return 0;
}
