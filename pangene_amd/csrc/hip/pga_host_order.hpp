// pga_host_order.hpp -- exact-order overrides, the index-0 channel, hazard lists.
// Host side of the device ABI (include/pangene_hip.h); included by pga_backend.hip (one translation unit), in this order.
#pragma once



extern "C" int pga_override_order(pga_ctx_t *c, int32_t which, int32_t n_seg, const int32_t *seg_genome, const int32_t *seg_start,
                                  const int64_t *seg_off, const int32_t *file_idx)
{
	const int N = c->N;
	// The gene-major index (hits by (gene, genome, X position), k_genes.hpp) survives an override: a cm override leaves it alone (only
	// zposy, the index by cm position, is derived again); a cs override moves hits inside (contig, cs) tie groups, so the X positions the
	// index stores are renumbered and the ORDER of two hits of one (gene, genome) that share their start may go stale -- which nothing
	// can see unless both are walkable, i.e. on opposite strands under -S (one gene's overlapping hits are filtered down to one
	// otherwise): with -S the index is rebuilt.  (The full-size configs[4] run rebuilt it -- a sort and seven gathers over 22 M hits --
	// 49 times per pass.)
	const bool z_keep = c->z_valid && (which == 1 || !c->par.check_strand);
	// The half-arc records of the walk that stands survive too when the override is small: only the overridden contigs are walked again
	// (k_walk_list), with the tag that stands.  Not with virtual contigs (a piece's neighbours in the walk may lie in the piece next to it).
	// The walk's 32-byte records (k_pack_wrec) stand as well except for the overridden hits: those are packed again here, whether a walk stands or not
	// (round 5 packed the whole shard again after every override that found no walk standing: 13 of 145 us per pass of the full-size configs[4] set).
	const bool wrec_ok = c->wrec_valid && z_keep && !c->zposy_stale && n_seg > 0;
	const bool partial = c->ha_valid && wrec_ok && c->vfirst == nullptr && seg_off[n_seg] * 8 <= (int64_t)N;
	c->walk_valid = false, c->ha_valid = false, c->yrec_valid = false, c->wrec_valid = false;
	if (!z_keep) c->z_valid = false;
	if (n_seg <= 0 || N == 0) return 0;
	const int64_t T = seg_off[n_seg];
	if (T == 0) { if (wrec_ok) c->wrec_valid = true; if (partial) c->ha_valid = true; return 0; }
	auto walk_again = [&](const int32_t *d_pos) -> int { // (after the override's own kernels, on the same stream; d_pos: places in the walk's list -- cm positions, or places in the members' list)
		int32_t *hzl = (int32_t *)c->pool.get(S_HZLIST, sizeof(int32_t) * PGA_HAZARD_CAP);
		if (!hzl) return PGA_ERR_NOMEM;
		if (wrec_ok) {
			hipLaunchKernelGGL(k_pack_wrec_list, dim3(nblk(T)), dim3(BLOCK), 0, c->st, WrecSrc{c->ylist, c->seg, c->gid, c->cm, c->sori, c->sdom, c->pdom0, c->prot_gid, c->flags, c->zpos, c->vfirst, c->vbase}, d_pos, T, c->wrec);
			c->wrec_valid = true, c->zposy_stale = false;
		}
		if (partial) {
			hipLaunchKernelGGL(k_walk_list, dim3(nblk(T)), dim3(BLOCK), 0, c->st, Walk{c->flags, c->ylist, c->wrec, c->g2s, c->hfk, c->hbk, c->hfp, c->hbp, c->round_tag, c->ha_ori, c->NL, c->dcnt, hzl, Gate{nullptr, 0}}, d_pos, T);
			c->ha_valid = true;
		}
		return 0;
	};
	const bool live = c->live_on && z_keep; // the lists stand and have to follow the override (an index that is dropped is built again, lists and all)
	// positions and file indices of the overridden hits, built in page-locked memory (a real DMA; from a std::vector the runtime stages)
	// (nothing waits at the end of an override any more -- sixty-six of them per pass each found the device still at the round queued
	// before -- so the lists must not be overwritten while their copy is under way: two halves, an event each)
	const size_t ov_bytes = (sizeof(int32_t) * 3 * (size_t)T + 255) & ~(size_t)255;
	if (c->h_ov_cap < ov_bytes) {
		if (c->h_ov) HIPCHK(hipStreamSynchronize(c->st));
		const size_t cap = ov_bytes + ov_bytes / 2 + 256;
		c->h_ov = (int32_t *)c->pin.get(2 * cap);
		if (!c->h_ov) return PGA_ERR_NOMEM;
		c->h_ov_cap = cap, c->ov_ev_used[0] = c->ov_ev_used[1] = false;
	}
	const int half = (int)(c->ov_seq++ & 1u);
	if (!c->ov_ev[half]) HIPCHK(hipEventCreateWithFlags(&c->ov_ev[half], hipEventDisableTiming));
	if (c->ov_ev_used[half]) HIPCHK(hipEventSynchronize(c->ov_ev[half]));
	int32_t *pos = (int32_t *)((char *)c->h_ov + (size_t)half * c->h_ov_cap), *fil = pos + T, *fst = fil + T;
	for (int32_t s = 0; s < n_seg; ++s) {
		const int32_t g = seg_genome[s], base = c->h_goff[(size_t)g];
		for (int64_t k = seg_off[s]; k < seg_off[s + 1]; ++k)
			pos[(size_t)k] = base + seg_start[s] + (int32_t)(k - seg_off[s]), fil[(size_t)k] = base + file_idx[k], fst[(size_t)k] = (int32_t)seg_off[s];
	}
	int32_t *d_pos = (int32_t *)c->pool.get(S_OVPOS, sizeof(int32_t) * 4 * (size_t)T + 64), *d_fil = (int32_t *)c->pool.get(S_OVFILE, sizeof(int32_t) * (size_t)T);
	int32_t *remap = (int32_t *)c->pool.get(S_I32_B, sizeof(int32_t) * (size_t)N);
	if (!d_pos || !d_fil || !remap) return PGA_ERR_NOMEM;
	int32_t *d_fst = d_pos + (size_t)T, *d_ex = d_pos + 2 * (size_t)T, *d_lpos = d_pos + 3 * (size_t)T; // (live lists) first entry of the entry's contig, members listed before it, its place in the lists
	TRY(upload(c, d_pos, pos, (size_t)T)); TRY(upload(c, d_fil, fil, (size_t)T));
	if (live) TRY(upload(c, d_fst, fst, (size_t)T));
	HIPCHK(hipEventRecord(c->ov_ev[half], c->st)); c->ov_ev_used[half] = true;
	if (!c->inv_valid) { hipLaunchKernelGGL(k_inv_only, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->fidx, c->gnm, c->goff, N, c->inv); c->inv_valid = true; } // (then kept current by the overrides themselves)
	if (live) { // the override in the lists' coordinates (k_order.hpp); before anything moves: inv and the flag words are those of the order that is being replaced
		I32 *tl = (I32 *)c->pool.get(S_TILE, tile_buf_bytes(T));
		if (!tl) return PGA_ERR_NOMEM;
		device_scan<I32>(InOvMember{c->flags, c->inv, d_fil}, OutExclI32{d_ex}, T, tl, OpSum{}, I32{0}, c->st);
		hipLaunchKernelGGL(k_ovl_pos, dim3(nblk(T)), dim3(BLOCK), 0, c->st, (const int32_t *)d_ex, (const uint32_t *)c->flags, (const int32_t *)c->inv, (const int32_t *)d_pos, (const int32_t *)d_fil, (const int32_t *)d_fst, T, (const int32_t *)c->lx, d_lpos);
	}
	if (which == 1) {
		hipLaunchKernelGGL(k_ov_sety, dim3(nblk(T)), dim3(BLOCK), 0, c->st, d_pos, d_fil, T, c->inv, c->yperm, live ? (const int32_t *)d_lpos : (const int32_t *)nullptr, c->ylist_buf);
		if (z_keep) c->zposy_stale = true;
		if (wrec_ok) TRY(walk_again(live ? d_lpos : d_pos));
		return 0;
	}
	int32_t *tmp = (int32_t *)c->pool.get(S_PERM, sizeof(int32_t) * (OV_PLANES + 13) * (size_t)T + 64);
	if (!tmp) return PGA_ERR_NOMEM;
	PermArrays p = { { c->fidx, c->pid, c->gid, c->cm, c->nex, c->sadj, c->rank, c->sdom, c->pdom, c->pdom0, (int32_t *)c->flags, c->sori }, { c->recA, c->recB, c->recC } };
	static_assert(OV_FLAGS == 10, "the flag word's place in PermArrays");
	hipLaunchKernelGGL(k_ov_gather, dim3(nblk(T)), dim3(BLOCK), 0, c->st, p, d_pos, d_fil, T, c->inv, tmp, remap, z_keep ? (const int32_t *)c->zpos : (const int32_t *)nullptr);
	hipLaunchKernelGGL(k_ov_scatter, dim3(nblk(T)), dim3(BLOCK), 0, c->st, p, d_pos, d_fil, T, tmp, c->gnm, c->goff, c->inv, c->zx, z_keep ? c->zpos : (int32_t *)nullptr);
	hipLaunchKernelGGL(k_ov_remap_y, dim3(nblk(T)), dim3(BLOCK), 0, c->st, c->yperm, d_pos, T, remap);
	if (live) hipLaunchKernelGGL(k_ovl_remap_ylist, dim3(nblk(T)), dim3(BLOCK), 0, c->st, c->ylist_buf, (const int32_t *)d_lpos, T, (const int32_t *)remap);
	if (z_keep) c->zposy_stale = true;
	SegMax *tile = (SegMax *)c->pool.get(S_TILE, tile_buf_bytes(T));
	if (!tile) return PGA_ERR_NOMEM;
	device_scan<SegMax>(InSegMaxList{c->recA, d_pos}, OutSegMaxList{c->recA, d_pos}, T, tile, OpSegMax{}, SegMax{SEG_EMPTY, 0}, c->st); // pm follows the new order
	hipLaunchKernelGGL(k_cstie_list, dim3(nblk(T)), dim3(BLOCK), 0, c->st, c->recA, d_pos, T, N, c->flags);
	if (live) { // the members' compact records follow (SweepView::xmap), and their pm
		hipLaunchKernelGGL(k_ovl_records, dim3(nblk(T)), dim3(BLOCK), 0, c->st, (const int32_t *)d_pos, (const int32_t *)d_lpos, T, (const int4 *)c->recA, (const int4 *)c->recB, (const int4 *)c->recC, c->cA, c->cB, c->cC, c->cx);
		device_scan<SegMax>(InSegMaxListL{c->cA, d_lpos}, OutSegMaxListL{c->cA, d_lpos}, T, tile, OpSegMax{}, SegMax{SEG_EMPTY, 0}, c->st);
	}
	if (wrec_ok) TRY(walk_again(live ? d_lpos : d_pos));
	return 0;
}

extern "C" int pga_set_head(pga_ctx_t *c, const int32_t *head_file)
{
	const int GL = c->n_genome;
	if (GL == 0 || c->N == 0) return 0;
	int32_t *d = (int32_t *)c->pool.get(S_OVFILE, sizeof(int32_t) * (size_t)GL);
	if (!d) return PGA_ERR_NOMEM;
	TRY(stage_upload(c, d, head_file, sizeof(int32_t) * (size_t)GL)); // head_file is caller memory
	if (!c->inv_valid) { hipLaunchKernelGGL(k_inv_only, dim3(nblk(c->N)), dim3(BLOCK), 0, c->st, c->fidx, c->gnm, c->goff, c->N, c->inv); c->inv_valid = true; }
	hipLaunchKernelGGL(k_set_head, dim3(nblk(GL)), dim3(BLOCK), 0, c->st, d, c->goff, c->inv, GL, c->headpos, c->flags);
	return 0;
}

extern "C" int pga_hazard_segs(pga_ctx_t *c, int32_t *segs, int32_t cap, int64_t *n_total)
{
	hipLaunchKernelGGL(k_mail_flush, dim3(1), dim3(64), 0, c->st, c->dcnt, c->h_box);
	TRY(sync_st(c));
	*n_total = c->h_cnt[14];
	int64_t n = std::min<int64_t>(std::min<int64_t>(*n_total, PGA_HAZARD_CAP), cap);
	const int32_t *list = (const int32_t *)c->pool.get(S_HZLIST, sizeof(int32_t) * PGA_HAZARD_CAP);
	if (n > 0 && list) { HIPCHK(hipMemcpyAsync(segs, list, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->st)); TRY(sync_st(c)); }
	return 0;
}
