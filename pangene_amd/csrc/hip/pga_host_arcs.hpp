// pga_host_arcs.hpp -- pg_graph_flag_vtx, the walk, pg_gen_arc in its two formulations, the cross-shard merge (graph.c:61-177).
// Host side of the device ABI (include/pangene_hip.h); included by pga_backend.hip (one translation unit), in this order.
#pragma once


extern "C" int pga_flag_vtx(pga_ctx_t *c, const int32_t *g2s, int32_t n_seg, int32_t then_filter)
{
	TRY(stage_upload(c, c->g2s, g2s, sizeof(int32_t) * (size_t)c->Q)); // g2s is caller memory
	c->n_seg = n_seg;
	if (then_filter) c->walk_valid = false, c->ha_valid = false;
	if (c->N) hipLaunchKernelGGL(k_flag_vtx, dim3(nblk(c->N)), dim3(BLOCK), 0, c->st, c->flags, c->gid, c->N, c->g2s, then_filter);
	return 0;
}

static void ensure_yrec(pga_ctx *c)
{
	if (c->yrec_valid || c->N == 0) return;
	hipLaunchKernelGGL(k_pack_yrec, dim3(nblk(c->N)), dim3(BLOCK), 0, c->st, c->yperm, c->seg, c->gid, c->gnm, c->cm, c->sori, c->sdom, c->pdom0, c->prot_gid, c->flags,
	                   c->N, c->yrecA, c->yrecB, c->vfirst, c->vbase);
	c->yrec_valid = true;
}

// walkable marks in cm order + predecessor; shared by arc_round and mark_hits
static int walk_prev(pga_ctx *c, int32_t **val_out, int32_t **prev_out)
{
	const int N = c->N;
	int32_t *val = (int32_t *)c->pool.get(S_WALK_VAL, sizeof(int32_t) * (size_t)N);
	int32_t *prev = (int32_t *)c->pool.get(S_WALK_PREV, sizeof(int32_t) * (size_t)N);
	I32 *tile = (I32 *)c->pool.get(S_TILE, tile_buf_bytes(N));
	if (!val || !prev || !tile) return PGA_ERR_NOMEM;
	*val_out = val, *prev_out = prev;
	if (c->walk_valid) return 0; // pg_mark_branch_flt_hit walks exactly what the pg_gen_arc before it walked: nothing changed in between
	device_scan<I32>(InWalk{c->flags, c->yperm}, OutPrev{val, prev}, N, tile, OpMax{}, I32{-1}, c->st); // marks + exclusive running max = previous walkable
	c->walk_valid = true;
	return 0;
}

// gene-major index: hits sorted by (gene, X position) -- X order is genome-major, so a gene's hits are grouped by genome
// known_live >= 0: the caller has just counted the hits without flt (and nothing was filtered since): the live lists are built, without a wait
static int build_z(pga_ctx *c, int64_t known_live, bool may_wait = true);
static int ensure_z(pga_ctx *c)
{
	if (c->N == 0) return 0;
	if (c->z_valid) {
		if (c->zposy_stale) c->wrec_valid = false, c->zposy_stale = false, c->ha_valid = false;
		return 0;
	}
	return build_z(c, -1);
}
static int build_z(pga_ctx *c, int64_t known_live, bool may_wait) // may_wait = false: nothing is built when building would mean a host wait (the live lists' count)
{
	const int N = c->N;
	uint64_t *key = (uint64_t *)c->pool.get(S_KEY_A, 0);
	uint32_t *val = (uint32_t *)c->pool.get(S_VAL_A, 0);
	if (!key || !val) return PGA_ERR_NOMEM;
	// Live lists (pga_host_context.hpp): worth their two scans and one host wait when at most three hits in four are left -- the vertex step
	// counted them (live_hint) before graph.c:285-288 filtered some more.  PANGENE_LIVE_LISTS=0 never, =1 whenever the count is known.
	static const int live_env = [] { const char *e = getenv("PANGENE_LIVE_LISTS"); return e ? atoi(e) : -1; }();
	const bool want_live = live_env != 0 && (known_live >= 0 || (c->live_hint >= 0 && (live_env == 1 || c->live_hint * 4 <= (int64_t)N * 3)));
	int n = N;
	if (want_live && known_live < 0 && !may_wait) return 0;
	if (want_live) {
		I32 *tile = (I32 *)c->pool.get(S_TILE, tile_buf_bytes(N));
		if (!tile) return PGA_ERR_NOMEM;
		device_scan<I32>(InLiveX{c->flags}, OutLiveX{c->flags, c->lx, c->gid, key, val, (int64_t)N}, N, tile, OpSum{}, I32{0}, c->st);
		hipLaunchKernelGGL(k_mail_live, dim3(1), dim3(64), 0, c->st, c->lx, (int64_t)N, c->dcnt, c->h_box);
		device_scan<I32>(InLiveY{c->flags, c->yperm}, OutLiveY{c->flags, c->yperm, c->ylist_buf}, N, tile, OpSum{}, I32{0}, c->st);
		HIPCHK(hipMemsetAsync(c->zpos, 0xff, sizeof(int32_t) * (size_t)N, c->st)); // -1: not in the index
		if (known_live < 0) TRY(sync_st(c)); // the grids of everything that follows are sized by the members' number
		n = known_live < 0 ? (int)c->h_cnt[10] : (int)known_live;
		c->live_on = true, c->NL = n, c->ylist = c->ylist_buf;
		// the sweeps of the rounds run over the members' records (SweepView::xmap): compact copies, pm over the members
		hipLaunchKernelGGL(k_live_records, dim3(nblk(N)), dim3(BLOCK), 0, c->st, (const uint32_t *)c->flags, (const int32_t *)c->lx, (const int4 *)c->recA, (const int4 *)c->recB, (const int4 *)c->recC, N, c->cA, c->cB, c->cC, c->cx);
		if (n) device_scan<SegMax>(InSegMaxA{c->cA}, OutSegMaxA{c->cA}, n, (SegMax *)c->pool.get(S_TILE, tile_buf_bytes(N)), OpSegMax{}, SegMax{SEG_EMPTY, 0}, c->st);
		if (getenv("PANGENE_TIMING")) fprintf(stderr, "[pga] live lists: %d of %d hits (%.3f) are not filtered; the walk and the gene-major index hold those\n", n, N, (double)n / N);
	} else {
		c->live_on = false, c->NL = N, c->ylist = c->yperm;
		hipLaunchKernelGGL(k_zkey, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->gid, N, key, val);
	}
	uint64_t *ks = key; uint32_t *vs = val;
	if (n) TRY(radix_sort_pool(c, key, val, n, bits_for((uint32_t)std::max(1, c->Q)), &ks, &vs));
	if (n) hipLaunchKernelGGL(k_zrec, dim3(nblk(n)), dim3(BLOCK), 0, c->st, vs, ks, c->ctg_base, c->n_genome, c->flags, c->cm, c->seg, n, ZIndex{c->zx, c->zy, c->zg, c->zst, c->zpos});
	hipLaunchKernelGGL(k_zoff, dim3(nblk(c->Q + 1)), dim3(BLOCK), 0, c->st, ks, n, c->Q, c->zoff);
	c->z_valid = true, c->ha_valid = false, c->zposy_stale = false, c->wrec_valid = false;
	return 0;
}

// the index (and the walk's records) queued now, behind whatever the stream holds: see pga_ctx::z_early
static int early_index(pga_ctx *c)
{
	c->z_early = false;
	if (c->N == 0 || c->z_valid) return 0;
	TRY(build_z(c, -1, false));
	if (c->z_valid && !c->wrec_valid && c->NL) {
		hipLaunchKernelGGL(k_pack_wrec, dim3(nblk(c->NL)), dim3(BLOCK), 0, c->st, WrecSrc{c->ylist, c->seg, c->gid, c->cm, c->sori, c->sdom, c->pdom0, c->prot_gid, c->flags, c->zpos, c->vfirst, c->vbase}, c->NL, c->wrec);
		c->wrec_valid = true;
	}
	return 0;
}

static void time_mark(pga_ctx *c, TimedLaunch *t, int which, bool end)
{
	if (!end) { t->which = which, t->units = c->N; (void)hipEventCreate(&t->a); (void)hipEventCreate(&t->b); (void)hipEventRecord(t->a, c->st); }
	else { (void)hipEventRecord(t->b, c->st); c->timed.push_back(*t); }
}

// (A) of k_genes.hpp: walk the cm order once, leave every walkable hit's two half-arc records
static int ensure_half_arcs(pga_ctx *c, int use_ori)
{
	if (c->N == 0) return 0;
	TRY(ensure_z(c));
	if (!c->wrec_valid) {
		if (c->NL) hipLaunchKernelGGL(k_pack_wrec, dim3(nblk(c->NL)), dim3(BLOCK), 0, c->st, WrecSrc{c->ylist, c->seg, c->gid, c->cm, c->sori, c->sdom, c->pdom0, c->prot_gid, c->flags, c->zpos, c->vfirst, c->vbase}, c->NL, c->wrec);
		c->wrec_valid = true;
	}
	if (c->ha_valid && c->ha_ori == use_ori) return 0;
	if (++c->round_tag >= HA_TAG_MAX) { // tags wrap: forget every old record (HA_TAG_MAX itself is never a tag: it is what the cleared key words carry)
		HIPCHK(hipMemsetAsync(c->hfk, 0xff, sizeof(uint32_t) * (size_t)c->N, c->st));
		HIPCHK(hipMemsetAsync(c->hbk, 0xff, sizeof(uint32_t) * (size_t)c->N, c->st));
		c->round_tag = 1;
	}
	int32_t *hzl = (int32_t *)c->pool.get(S_HZLIST, sizeof(int32_t) * PGA_HAZARD_CAP);
	if (!hzl) return PGA_ERR_NOMEM;
	TimedLaunch tw; if (c->timing_rounds) time_mark(c, &tw, 6, false);
	const Walk wk = {c->flags, c->ylist, c->wrec, c->g2s, c->hfk, c->hbk, c->hfp, c->hbp, c->round_tag, use_ori, c->NL, c->dcnt, hzl, c->gate};
	static const int ipt_env = [] { const char *e = getenv("PANGENE_WALK_IPT"); return e ? atoi(e) : 0; }(); // (measurements: 1 / 4 positions a thread whatever the size)
	if (ipt_env == 4 || (ipt_env != 1 && c->NL >= WK_FEW_FROM)) hipLaunchKernelGGL(k_walk<4>, dim3(nblk(c->NL, BLOCK * 4)), dim3(BLOCK), 0, c->st, wk);
	else if (c->NL) hipLaunchKernelGGL(k_walk<1>, dim3(nblk(c->NL, BLOCK)), dim3(BLOCK), 0, c->st, wk);
	if (c->timing_rounds) time_mark(c, &tw, 6, true);
	c->ha_valid = true, c->ha_ori = use_ori;
	return 0;
}

struct CurTable { uint64_t *ax; uint8_t *aw, *vwk; int32_t *s1, *agid, *vs, *ve, *sg, *dg; };
static int cur_table(pga_ctx *c, int64_t n_arc, int n_seg, CurTable *t)
{
	const int n_vtx = 2 * n_seg;
	t->ax = (uint64_t *)c->pool.get(S_ARCX, sizeof(uint64_t) * (size_t)n_arc + 16);
	t->aw = (uint8_t *)c->pool.get(S_ARCW, (size_t)n_arc + 16);
	t->s1 = (int32_t *)c->pool.get(S_BR_S1, sizeof(int32_t) * (size_t)n_arc + 16), t->agid = (int32_t *)c->pool.get(S_BR_GID, sizeof(int32_t) * (size_t)n_arc + 16);
	t->vs = (int32_t *)c->pool.get(S_BR_VS, sizeof(int32_t) * (size_t)n_vtx + 16), t->ve = (int32_t *)c->pool.get(S_BR_VE, sizeof(int32_t) * (size_t)n_vtx + 16);
	t->sg = (int32_t *)c->pool.get(S_BR_SEGGID, sizeof(int32_t) * (size_t)n_seg + 16), t->dg = (int32_t *)c->pool.get(S_DEG, sizeof(int32_t) * (size_t)n_vtx + 16);
	t->vwk = (uint8_t *)c->pool.get(S_VWK, (size_t)n_vtx + 16);
	return (t->ax && t->aw && t->s1 && t->agid && t->vs && t->ve && t->sg && t->dg && t->vwk) ? 0 : PGA_ERR_NOMEM;
}

// pg_gen_arc on the gene-major index (k_genes.hpp).  Leaves the round's arcs, every gene's in its own stretch of the table arrays,
// and everything the branch steps read (what pga_arc_set_current would derive) in place; seg_cnt[2S] and the degrees go to the
// pinned buffer h_round_dev when one is given.  The counters travel to the pinned mirror with the last kernel; nothing waits here.
static int arc_round_genes(pga_ctx *c, int use_ori, int32_t **seg_cnt_out, int32_t **deg_out, int32_t *h_round_dev, bool mail = true)
{
	const int N = c->N, S = c->n_seg;
	const int64_t cap = 2 * (int64_t)N + 2; // distinct arcs <= half-arcs <= 2 (N - 1)
	int32_t *seg_cnt = (int32_t *)c->pool.get(S_SEGCNT, sizeof(int32_t) * 2 * (size_t)std::max(1, S) * SEGCNT_COPIES);
	pga_arc_part_t *stage = (pga_arc_part_t *)c->pool.get(S_ARC_STAGE, sizeof(pga_arc_part_t) * (size_t)cap);
	int4 *gmeta = (int4 *)c->pool.get(S_GMETA, sizeof(int4) * (size_t)std::max(1, S));
	CurTable t;
	if (!seg_cnt || !stage || !gmeta) return PGA_ERR_NOMEM;
	TRY(cur_table(c, cap, S, &t));
	*seg_cnt_out = seg_cnt, *deg_out = t.dg;
	c->table_sparse = true;
	TimedLaunch tr; if (c->timing_rounds) time_mark(c, &tr, 5, false);
	TRY(launch_sweep<0>(c, 2)); // graph.c:102
	TRY(ensure_half_arcs(c, use_ori));
	if (S == 0) { // nothing to build; the round's tail still has to be written (the pinned buffer is recycled memory)
		if (c->timing_rounds) time_mark(c, &tr, 5, true);
		hipLaunchKernelGGL(k_mail_round, dim3(1), dim3(64), 0, c->st, c->dcnt, c->h_box, h_round_dev);
		return 0;
	}
	int32_t *big = (int32_t *)c->pool.get(S_BIGLIST, sizeof(int32_t) * (size_t)std::max(1, c->Q));
	if (!big) return PGA_ERR_NOMEM;
	static const int cap_log2 = [] { const char *e = getenv("PANGENE_GENE_TABLE_LOG2"); const int v = e ? atoi(e) : 9; return v < 1 ? 1 : v > 9 ? 9 : v; }();
	GeneArcs ga = { c->zy, c->zoff, c->hfk, c->hbk, c->hfp, c->hbp, c->g2s, c->Q, S, c->round_tag, cap_log2, seg_cnt, t.sg, stage, gmeta,
	                t.ax, t.s1, t.agid, t.aw, t.vs, t.ve, t.dg, t.vwk, h_round_dev, big, c->ga_ctl, c->dcnt, c->gate, (c->gate.w || c->loop_gated) ? c->loopctl + 2 : (int32_t *)nullptr };
	// (round 6, measured side by side at configs[1] / human 47 x 20 k, ms per pass: <128 threads, 128 keys, 512 hits> 5.26 / 5.11 -- kept; <64, 64, 256> 5.83 / 4.98;
	// <128, 64, 256> 5.65 / 5.08; <64, 128, 512> 5.49 / 5.13; <64, 32, 256> 5.90 / 5.05: smaller tables put more genes on a CU and send more of them to the second kernel)
	hipLaunchKernelGGL((k_gene_arcs_wave_t<GA_WAVE_NT, GA_CAP_WAVE, GA_WAVE_HITS, 7>), dim3((unsigned)c->Q), dim3(GA_WAVE_NT), 0, c->st, ga);
	hipLaunchKernelGGL(k_gene_arcs_big, dim3((unsigned)std::min(c->Q, 8 * c->n_cu)), dim3(GA_BIG_NT), 0, c->st, ga);
	if (c->timing_rounds) time_mark(c, &tr, 5, true);
	if (mail) hipLaunchKernelGGL(k_mail_round, dim3(1), dim3(64), 0, c->st, c->dcnt, c->h_box, h_round_dev ? h_round_dev + 4 * (size_t)S : (int32_t *)nullptr); // invariant / overflow counters for the host; the overflow counter starts again
	return 0;
}

// the round's table as one array sorted by x (see k_genes.hpp (C)); waits, returns the size
static int arc_table_compact(pga_ctx *c, pga_arc_part_t **arcs_out, int64_t *n_out)
{
	const int S = c->n_seg;
	*n_out = 0, *arcs_out = nullptr;
	if (S == 0 || c->N == 0) return 0;
	const int64_t cap = 2 * (int64_t)c->N + 2;
	pga_arc_part_t *stage = (pga_arc_part_t *)c->pool.get(S_ARC_STAGE, 0), *arcs = (pga_arc_part_t *)c->pool.get(S_ARCS, sizeof(pga_arc_part_t) * (size_t)cap);
	int4 *gmeta = (int4 *)c->pool.get(S_GMETA, 0);
	int32_t *off = (int32_t *)c->pool.get(S_GOFF, sizeof(int32_t) * (size_t)S);
	I32 *tile = (I32 *)c->pool.get(S_TILE, tile_buf_bytes(S));
	if (!stage || !arcs || !gmeta || !off || !tile) return PGA_ERR_NOMEM;
	device_scan<I32>(InGmeta{gmeta}, OutExclI32{off}, S, tile, OpSum{}, I32{0}, c->st);
	hipLaunchKernelGGL(k_arc_compact, dim3(nblk(S, BLOCK / WAVE)), dim3(BLOCK), 0, c->st, gmeta, off, S, stage, arcs, c->dcnt, c->h_box);
	TRY(sync_st(c));
	*arcs_out = arcs, *n_out = c->h_cnt[10];
	return 0;
}

// The reference's formulation -- every temp arc through one global sort (graph.c:127,151): kept as the path of rounds in which a
// hub gene overflows the LDS table of k_gene_arcs, and (PANGENE_ARC_SORT_PATH=1) as an independent check of the gene path.
// sweep_done: the round's pg_shadow (graph.c:102) has run already (a gene-path attempt that overflowed): it must not run again --
// by the time a deferred round is repeated the hits may follow the NEXT cs order (graph.c:123), and ties would fall differently.
static int arc_round_sorted(pga_ctx_t *c, int32_t use_ori, int32_t **seg_cnt_out, pga_arc_part_t **arcs_out, int64_t *n_arcs_out, bool sweep_done)
{
	const int N = c->N, S = c->n_seg, GL = c->n_genome;
	int32_t *seg_cnt = (int32_t *)c->pool.get(S_SEGCNT, sizeof(int32_t) * 2 * (size_t)std::max(1, S) * SEGCNT_COPIES);
	if (!seg_cnt) return PGA_ERR_NOMEM;
	*seg_cnt_out = seg_cnt, *arcs_out = nullptr, *n_arcs_out = 0;
	if (N == 0) {
		HIPCHK(hipMemsetAsync(seg_cnt, 0, sizeof(int32_t) * 2 * (size_t)std::max(1, S) * SEGCNT_COPIES, c->st));
		return sync_st(c);
	}
	if (!sweep_done) TRY(launch_sweep<0>(c, 2)); // graph.c:102
	int32_t *val, *prev;
	TRY(walk_prev(c, &val, &prev));
	const int64_t wpg = (S + 31) / 32;
	uint32_t *seen = (uint32_t *)c->pool.get(S_BITS, sizeof(uint32_t) * (size_t)(wpg * GL) + 16);
	int32_t *has = (int32_t *)c->pool.get(S_I32_C, sizeof(int32_t) * (size_t)N);
	int32_t *slot = (int32_t *)c->pool.get(S_SLOT, sizeof(int32_t) * (size_t)(2 * (int64_t)N + 2));
	if (!seen || !has || !slot) return PGA_ERR_NOMEM;
	zero_multi(c, seen, sizeof(uint32_t) * (size_t)(wpg * GL) + 16, seg_cnt, sizeof(int32_t) * 2 * (size_t)std::max(1, S) * SEGCNT_COPIES);
	ensure_yrec(c);
	hipLaunchKernelGGL(k_arc_flag, dim3(nblk(N)), dim3(BLOCK), 0, c->st, val, prev, c->yrecA, c->g2s, N, S, has, seg_cnt, seen, wpg, c->dcnt, (int32_t *)c->pool.get(S_HZLIST, sizeof(int32_t) * PGA_HAZARD_CAP));
	if (S) hipLaunchKernelGGL(k_segcnt_sum, dim3(nblk(2 * S)), dim3(BLOCK), 0, c->st, seg_cnt, 2 * S);
	I32 *tile = (I32 *)c->pool.get(S_TILE, tile_buf_bytes(N));
	device_scan<I32>(InI32{has}, OutExclI32{slot}, N, tile, OpSum{}, I32{0}, c->st);
	// number of adjacencies = slot[N-1] + has[N-1]
	hipLaunchKernelGGL(k_mail_sum, dim3(1), dim3(64), 0, c->st, slot + (N - 1), has + (N - 1), c->dcnt, c->h_box);
	TRY(check_invariant(c, true));
	const int64_t M = 2 * c->h_cnt[10];
	if (M == 0) return sync_st(c);
	const int vbits = bits_for((uint32_t)(2 * std::max(1, S)));
	uint64_t *key = (uint64_t *)c->pool.get(S_KEY_A, sizeof(uint64_t) * (size_t)M);
	uint32_t *idx = (uint32_t *)c->pool.get(S_VAL_A, sizeof(uint32_t) * (size_t)M);
	int4 *tpay = (int4 *)c->pool.get(S_TDIST, sizeof(int4) * (size_t)M), *spay = (int4 *)c->pool.get(S_SDIST, sizeof(int4) * (size_t)M);
	if (!key || !idx || !tpay || !spay) return PGA_ERR_NOMEM;
	ArcEmit e = { has, slot, prev, c->yrecA, c->yrecB, c->g2s, key, idx, tpay, N, use_ori, vbits };
	hipLaunchKernelGGL(k_arc_emit, dim3(nblk(N)), dim3(BLOCK), 0, c->st, e);
	uint64_t *ks; uint32_t *vs;
	TRY(radix_sort_pool(c, key, idx, M, 2 * vbits, &ks, &vs)); // graph.c:127 and :151 in one stable sort
	hipLaunchKernelGGL(k_arc_gather, dim3(nblk(M)), dim3(BLOCK), 0, c->st, vs, M, tpay, spay);
	tile = (I32 *)c->pool.get(S_TILE, tile_buf_bytes(M));
	device_scan<I32>(InKeyHead{ks}, OutExclI32{slot}, M, tile, OpSum{}, I32{0}, c->st); // run heads straight from the sorted keys
	hipLaunchKernelGGL(k_mail_runs, dim3(1), dim3(64), 0, c->st, ks, slot, M, c->dcnt, c->h_box);
	TRY(sync_st(c));
	const int64_t A = c->h_cnt[10];
	pga_arc_part_t *arcs = (pga_arc_part_t *)c->pool.get(S_ARCS, sizeof(pga_arc_part_t) * (size_t)A);
	if (!arcs) return PGA_ERR_NOMEM;
	{
		int32_t *run_start = (int32_t *)c->pool.get(S_RUNSTART, sizeof(int32_t) * (size_t)A + 16);
		int32_t *c_n = (int32_t *)tpay, *c_s1 = c_n + (size_t)M, *c_s2 = c_n + 2 * (size_t)M; // the unsorted payload is free again: reuse it
		uint64_t *c_dn = (uint64_t *)c->pool.get(S_CDN, sizeof(uint64_t) * (size_t)M);
		if (!run_start || !c_dn) return PGA_ERR_NOMEM;
		hipLaunchKernelGGL(k_arc_l1, dim3(nblk(M)), dim3(BLOCK), 0, c->st, ks, M, spay, slot, run_start, c_n, c_dn, c_s1, c_s2);
		hipLaunchKernelGGL(k_arc_l2, dim3(nblk(A, BLOCK / WAVE)), dim3(BLOCK), 0, c->st, ks, M, A, run_start, c_n, c_dn, c_s1, c_s2, vbits, arcs);
	}
	*arcs_out = arcs, *n_arcs_out = A;
	return sync_st(c);
}

static bool arc_sort_path_forced() { static const bool f = getenv("PANGENE_ARC_SORT_PATH") != nullptr; return f; }

extern "C" int pga_arc_round(pga_ctx_t *c, int32_t use_ori, int32_t **seg_cnt_out, pga_arc_part_t **arcs_out, int64_t *n_arcs_out)
{
	if (c->x_redo) { // the round pga_arc_round_x gave up (its sweep has run)
		c->x_redo = false, c->table_sparse = false;
		return arc_round_sorted(c, use_ori, seg_cnt_out, arcs_out, n_arcs_out, c->N != 0);
	}
	if (c->N && !arc_sort_path_forced()) {
		int32_t *deg;
		*n_arcs_out = 0;
		TRY(arc_round_genes(c, use_ori, seg_cnt_out, &deg, nullptr));
		TRY(sync_st(c));
		if (c->h_cnt[3]) return PGA_ERR_INVARIANT;
		if (c->h_cnt[9] == 0) return arc_table_compact(c, arcs_out, n_arcs_out); // the exchange wants one sorted array
		c->table_sparse = false;
		return arc_round_sorted(c, use_ori, seg_cnt_out, arcs_out, n_arcs_out, true);
	}
	c->table_sparse = false;
	return arc_round_sorted(c, use_ori, seg_cnt_out, arcs_out, n_arcs_out, false);
}

// pg_gen_arc for a run that is not sharded: the round's table is the graph's table at once (what pga_arc_set_current would
// derive is produced by the same kernels), and the only things the host needs -- segment counters, out-degrees, table size --
// arrive with ONE wait at the end.
static int arc_round_check(pga_ctx *c, int S, int32_t *seg_cnt_host, int32_t *deg_host) // after a wait: 0 ok, 1 = a gene overflowed its table, < 0 error
{
	const int n_vtx = 2 * S;
	const int32_t *tail = c->h_round + 2 * (size_t)n_vtx; // {overflowed genes, invariant violations} of THIS round (the mailbox may have moved on)
	if (tail[1]) return PGA_ERR_INVARIANT;
	if (tail[0]) return 1;
	if (n_vtx && seg_cnt_host) memcpy(seg_cnt_host, c->h_round, sizeof(int32_t) * (size_t)n_vtx), memcpy(deg_host, c->h_round + n_vtx, sizeof(int32_t) * (size_t)n_vtx);
	return 0;
}

extern "C" int pga_arc_round_local(pga_ctx_t *c, int32_t use_ori, int32_t n_seg, int32_t *seg_cnt_host, int32_t *deg_host)
{
	const int S = n_seg, n_vtx = 2 * S;
	if (S != c->n_seg) return PGA_ERR_ARG;
	c->arc_deferred = false, c->arc_done = false;
	if (c->N && !arc_sort_path_forced() && !c->force_sort_once) {
		int32_t *seg_cnt, *deg;
		const size_t need = sizeof(int32_t) * (2 * (size_t)n_vtx + 2) + 64;
		if (c->h_round_cap < need) {
			if (c->h_round) HIPCHK(hipStreamSynchronize(c->st));
			c->h_round = (int32_t *)c->pin.get(need + need / 2);
			if (!c->h_round) return PGA_ERR_NOMEM;
			c->h_round_cap = need + need / 2;
		}
		int32_t *h_dev = nullptr;
		HIPCHK(hipHostGetDevicePointer((void **)&h_dev, c->h_round, 0));
		if (seg_cnt_host == nullptr) HIPCHK(hipMemsetAsync(c->dcnt + 11, 0, sizeof(int64_t), c->st)); // the sticky flag of pga_branch_loop covers this round and what follows it
		TRY(arc_round_genes(c, use_ori, &seg_cnt, &deg, h_dev)); // the gene kernels write the counters and degrees into the pinned buffer
		c->br_n = 2 * (int64_t)c->N + 2, c->br_S = S, c->br_np = 0; // (br_n: extent of the table arrays; the arcs are counted when somebody asks, pga_arc_table)
		if (seg_cnt_host == nullptr) { c->arc_deferred = true, c->arc_epoch = c->sync_epoch; return 0; } // the caller collects the results later (pga_arc_round_finish)
		TRY(sync_st(c));
		const int rc = arc_round_check(c, S, seg_cnt_host, deg_host);
		if (rc <= 0) return rc;
		c->sweep_done = true;
	}
	if (seg_cnt_host == nullptr) { // deferred call on the sort path: done at once, the results wait in host memory for pga_arc_round_finish
		c->def_host.assign(2 * (size_t)n_vtx + 1, 0);
		TRY(pga_arc_round_local(c, use_ori, n_seg, c->def_host.data(), c->def_host.data() + n_vtx));
		c->arc_deferred = true, c->arc_done = true;
		return 0;
	}
	int32_t *seg_cnt; pga_arc_part_t *arcs; int64_t n = 0;
	const bool sweep_done = c->sweep_done; // set by a gene-path attempt of this very round (just above, or the deferred one being repeated)
	c->table_sparse = false, c->force_sort_once = false, c->sweep_done = false;
	TRY(arc_round_sorted(c, use_ori, &seg_cnt, &arcs, &n, sweep_done));
	TRY(pga_arc_set_current(c, arcs, n, S, deg_host));
	if (n_vtx) TRY(pga_fetch(c, seg_cnt_host, seg_cnt, sizeof(int32_t) * (size_t)n_vtx));
	return 0;
}

extern "C" int pga_arc_round_finish(pga_ctx_t *c, int32_t n_seg, int32_t *seg_cnt_host, int32_t *deg_host)
{
	if (!c->arc_deferred) return PGA_ERR_ARG;
	c->arc_deferred = false;
	if (c->arc_done) { // the round took the sort path and is complete
		const size_t n_vtx = 2 * (size_t)n_seg;
		c->arc_done = false;
		if (n_vtx && seg_cnt_host) memcpy(seg_cnt_host, c->def_host.data(), sizeof(int32_t) * n_vtx), memcpy(deg_host, c->def_host.data() + n_vtx, sizeof(int32_t) * n_vtx);
		return 0;
	}
	if (c->sync_epoch == c->arc_epoch) TRY(sync_st(c)); // nobody has waited since the round was queued
	const int rc = arc_round_check(c, n_seg, seg_cnt_host, deg_host);
	if (rc == 1) c->force_sort_once = true, c->sweep_done = true; // the caller repeats the round (without deferring): it takes the sort path, without a second sweep
	return rc;
}

extern "C" int pga_arc_table(pga_ctx_t *c, const pga_arc_part_t **arcs, int64_t *n_arc)
{
	if (c->table_sparse) {
		pga_arc_part_t *a;
		TRY(arc_table_compact(c, &a, n_arc));
		*arcs = a;
		return 0;
	}
	*arcs = c->cur_tab, *n_arc = c->cur_tab_n;
	return 0;
}

extern "C" int pga_arc_merge(pga_ctx_t *c, const pga_arc_part_t *gathered, const int64_t *count, int32_t W, int64_t slot_sz,
                             pga_arc_part_t **out, int64_t *n_out)
{
	int64_t tot = 0;
	for (int r = 0; r < W; ++r) tot += count[r];
	*out = nullptr, *n_out = 0;
	if (tot == 0) return 0;
	std::vector<int64_t> off((size_t)W + 1, 0);
	for (int r = 0; r < W; ++r) off[(size_t)r + 1] = off[(size_t)r] + count[r];
	int64_t *d_off = (int64_t *)c->pool.get(S_MG_SRC, sizeof(int64_t) * ((size_t)W + 1));
	uint64_t *key = (uint64_t *)c->pool.get(S_MG_KEY, sizeof(uint64_t) * (size_t)tot + 64);
	uint32_t *val = (uint32_t *)c->pool.get(S_MG_VAL, sizeof(uint32_t) * (size_t)tot + 64);
	int32_t *slot = (int32_t *)c->pool.get(S_MG_SLOT, sizeof(int32_t) * (size_t)tot);
	int32_t *tile = (int32_t *)c->pool.get(S_TILE, 0);
	if (!d_off || !key || !val || !slot || !tile) return PGA_ERR_NOMEM;
	if (tot > 2 * (int64_t)c->N + 2) { // the scan buffer is sized for 2N items
		tile = (int32_t *)c->pool.get(S_TILE, tile_buf_bytes(tot));
		if (!tile) return PGA_ERR_NOMEM;
	}
	TRY(upload(c, d_off, off.data(), (size_t)W + 1));
	MergeLists L = { W, slot_sz, d_off };
	hipLaunchKernelGGL(k_mg_rank, dim3(nblk(tot)), dim3(BLOCK), 0, c->st, gathered, L, key, val);
	device_scan<I32>(InMgHead{key}, OutExclI32{slot}, tot, (I32 *)tile, OpSum{}, I32{0}, c->st);
	int64_t *box = nullptr;
	HIPCHK(hipHostGetDevicePointer((void **)&box, c->h_cnt, 0)); // the count goes straight into the pinned mirror
	hipLaunchKernelGGL(k_mg_count, dim3(1), dim3(64), 0, c->st, key, slot, tot, box + 10);
	TRY(sync_st(c));
	const uint64_t *ks = key; const uint32_t *vs = val;
	const int64_t A = c->h_cnt[10];
	int32_t *run_start = (int32_t *)c->pool.get(S_MG_RUN, sizeof(int32_t) * (size_t)A + 16);
	pga_arc_part_t *res = (pga_arc_part_t *)c->pool.get(S_MG_OUT, sizeof(pga_arc_part_t) * (size_t)A + 16);
	if (!run_start || !res) return PGA_ERR_NOMEM;
	hipLaunchKernelGGL(k_mg_runstart, dim3(nblk(tot)), dim3(BLOCK), 0, c->st, ks, slot, tot, run_start);
	hipLaunchKernelGGL(k_mg_sum, dim3(nblk(A, BLOCK / WAVE)), dim3(BLOCK), 0, c->st, gathered, vs, tot, A, run_start, res);
	*out = res, *n_out = A;
	return 0;
}


extern "C" int pga_arc_set_current(pga_ctx_t *c, const pga_arc_part_t *arcs, int64_t n_arc, int32_t n_seg, int32_t *deg)
{
	const int n_vtx = 2 * n_seg;
	c->br_n = n_arc, c->br_S = n_seg, c->br_np = 0;
	c->cur_tab = arcs, c->cur_tab_n = n_arc, c->table_sparse = false;
	if (n_vtx) memset(deg, 0, sizeof(int32_t) * (size_t)n_vtx);
	CurTable t;
	TRY(cur_table(c, n_arc, n_seg, &t));
	if (n_vtx == 0) return 0;
	zero_multi(c, t.vs, sizeof(int32_t) * (size_t)n_vtx, t.ve, sizeof(int32_t) * (size_t)n_vtx, t.aw, (size_t)n_arc, t.vwk, (size_t)n_vtx);
	if (n_arc) {
		hipLaunchKernelGGL(k_seg_gid, dim3(nblk(c->Q)), dim3(BLOCK), 0, c->st, c->g2s, c->Q, n_seg, t.sg);
		hipLaunchKernelGGL(k_cur_prep, dim3(nblk(n_arc)), dim3(BLOCK), 0, c->st, arcs, n_arc, t.sg, t.ax, t.s1, t.agid, t.vs, t.ve);
	}
	hipLaunchKernelGGL(k_deg, dim3(nblk(n_vtx)), dim3(BLOCK), 0, c->st, t.vs, t.ve, n_vtx, t.dg);
	HIPCHK(hipMemcpyAsync(deg, t.dg, sizeof(int32_t) * (size_t)n_vtx, hipMemcpyDeviceToHost, c->st));
	return sync_st(c);
}
