// pga_host_branch.hpp -- branch.c on the device: representative positions, pg_n_local, pair enumeration, the decision, the queued rounds, hit marking.
// Host side of the device ABI (include/pangene_hip.h); included by pga_backend.hip (one translation unit), in this order.
#pragma once


// (pga_branch_loop) the launches of pg_gen_rep_pos left to the fused launches of the round's front (k_loop_front1 / 2, k_branch.hpp): what they need
struct RepDefer { bool on; RepFill rf; int32_t *rx; unsigned nb; bool cleared; };

static int rep_pos_impl(pga_ctx *c, RepDefer *df)
{
	const int N = c->N, GL = c->n_genome, Q = c->Q;
	if (df) df->on = false;
	const int64_t n_ent = (int64_t)Q * GL;
	int4 *rp = (int4 *)c->pool.get(S_RP_SEG, sizeof(int4) * (size_t)n_ent);
	int32_t *iv = (int32_t *)c->pool.get(S_RP_IV, sizeof(int32_t) * (size_t)n_ent);
	int32_t *hzl = (int32_t *)c->pool.get(S_HZLIST, sizeof(int32_t) * PGA_HAZARD_CAP);
	if (!rp || !iv || !hzl) return PGA_ERR_NOMEM;
	if (N) {
		int32_t *rx = (int32_t *)c->pool.get(S_I32_B, sizeof(int32_t) * (size_t)N);
		I32 *tile = (I32 *)c->pool.get(S_TILE, tile_buf_bytes(N));
		if (!rx || !tile) return PGA_ERR_NOMEM;
		TRY(ensure_half_arcs(c, c->ha_ori < 0 ? 0 : c->ha_ori)); // which hits are walkable, gene-major (normally left by the arc round just before)
		static const bool rank_scan = env_has("PANGENE_RANK", "scan"); // (tests: the general scan on shards of short genomes too)
		const bool defer = df && c->gs_np <= (1 << 15) && !rank_scan;
		if (defer) ; // (k_loop_front1)
		else if (c->gs_np <= (1 << 15) && !rank_scan) hipLaunchKernelGGL(k_rank_genome, dim3((unsigned)GL), dim3(RK_T), 0, c->st, (const uint32_t *)c->flags, (const int32_t *)c->goff, rx, c->gate); // rank among the walkable hits of the genome, cs order
		else device_scan<I32>(InWalkX{c->flags}, OutRank{rx, c->flags}, N, tile, OpSum{}, I32{0}, c->st, c->gate); // rank among the walkable hits, cs order
		if (!c->tg_valid) { hipLaunchKernelGGL(k_tie_bounds, dim3(nblk(N)), dim3(BLOCK), 0, c->st, (const int4 *)c->recA, (const uint32_t *)c->flags, N, c->tg); c->tg_valid = true; }
		RepFill rf = { c->tg, n_ent, GL, Q, N, c->NL, c->zx, c->zy, c->zg, c->zst, c->zoff, c->hbk, c->round_tag, c->recA, c->gid, c->flags, rx, c->goff, c->ctg_base, (void *)rp, iv, c->dcnt, hzl, c->vfirst, c->vbase, c->gate };
		if (defer) { df->on = true, df->rf = rf, df->rx = rx, df->cleared = c->live_on, df->nb = nblk(c->live_on ? std::max(c->NL, 1) : std::max(c->NL, Q)); return 0; }
		if (c->live_on) { // the index holds the live hits only: genes and (gene, genome) groups without an entry are many -- their records by one coalesced fill
			const unsigned nb = nblk(std::max(c->NL, 1));
			if (c->rp_form == RP_COMPACT) {
				if (n_ent) hipLaunchKernelGGL((k_rep_clear<RP_COMPACT>), dim3(nblk(n_ent)), dim3(BLOCK), 0, c->st, (void *)rp, n_ent, c->gate);
				hipLaunchKernelGGL((k_rep_fill<RP_COMPACT, true>), dim3(nb), dim3(BLOCK), 0, c->st, rf);
			} else {
				if (n_ent) hipLaunchKernelGGL((k_rep_clear<RP_FULL>), dim3(nblk(n_ent)), dim3(BLOCK), 0, c->st, (void *)rp, n_ent, c->gate);
				if (c->rp_form == RP_WIDE) hipLaunchKernelGGL((k_rep_fill<RP_WIDE, true>), dim3(nb), dim3(BLOCK), 0, c->st, rf);
				else hipLaunchKernelGGL((k_rep_fill<RP_FULL, true>), dim3(nb), dim3(BLOCK), 0, c->st, rf);
			}
		} else {
			const unsigned nb = nblk(std::max(c->NL, Q));
			if (c->rp_form == RP_COMPACT) hipLaunchKernelGGL((k_rep_fill<RP_COMPACT, false>), dim3(nb), dim3(BLOCK), 0, c->st, rf);
			else if (c->rp_form == RP_WIDE) hipLaunchKernelGGL((k_rep_fill<RP_WIDE, false>), dim3(nb), dim3(BLOCK), 0, c->st, rf);
			else hipLaunchKernelGGL((k_rep_fill<RP_FULL, false>), dim3(nb), dim3(BLOCK), 0, c->st, rf);
		}
	} else if (n_ent) {
		hipLaunchKernelGGL(k_fill_i32, dim3(nblk(4 * n_ent)), dim3(BLOCK), 0, c->st, (int32_t *)rp, 4 * n_ent, -1); // "absent" in either record form
	}
	return 0;
}

extern "C" int pga_rep_pos(pga_ctx_t *c) { return rep_pos_impl(c, nullptr); }

// n = number of pairs, or (np_dev != NULL) the capacity of d_pairs with the actual number in device memory
static int n_local_dev(pga_ctx *c, const int32_t *d_pairs, int64_t n, const int64_t *np_dev, int32_t local_dist, int32_t local_count, int32_t frag_mode, int32_t **cnt)
{
	int32_t *d_cnt = (int32_t *)c->pool.get(S_NLCNT, sizeof(int32_t) * (size_t)n + 16);
	int4 *rp = (int4 *)c->pool.get(S_RP_SEG, 0);
	if (!d_cnt || !rp) return PGA_ERR_NOMEM;
	*cnt = d_cnt;
	NLocalHz hz = { (const int32_t *)c->pool.get(S_RP_IV, 0), c->ctg_base, c->dcnt, (int32_t *)c->pool.get(S_HZLIST, sizeof(int32_t) * PGA_HAZARD_CAP) };
	if (!hz.iv || !hz.list) return PGA_ERR_NOMEM;
	// the grid follows the last known number of pairs (the kernel strides over whatever there is)
	const int64_t est = np_dev ? (c->br_np_seen > 0 ? c->br_np_seen : std::min<int64_t>(n, 1 << 18)) : n;
	static const int lanes_env = [] { const char *e = getenv("PANGENE_NL_LANES"); return e ? atoi(e) : 0; }(); // (measurements: 16 / 32 / 64 whatever the number of genomes)
	const int lanes = (lanes_env == 16 || lanes_env == 32 || lanes_env == 64) ? lanes_env : nl_lanes_for(c->n_genome);
	const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(nblk(est, BLOCK / WAVE * (WAVE / lanes)), 1 << 20));
#define NL_LAUNCH(FORM, LANES) hipLaunchKernelGGL((k_n_local<FORM, LANES>), dim3(grid), dim3(BLOCK), 0, c->st, d_pairs, n, np_dev, c->n_genome, (const void *)rp, local_dist, local_count, frag_mode, d_cnt, hz, c->gate)
#define NL_FORM(FORM) do { if (lanes == 16) NL_LAUNCH(FORM, 16); else if (lanes == 32) NL_LAUNCH(FORM, 32); else NL_LAUNCH(FORM, 64); } while (0)
	if (n && c->rp_form == RP_COMPACT) NL_FORM(RP_COMPACT);
	else if (n && c->rp_form == RP_WIDE) NL_FORM(RP_WIDE);
	else if (n) NL_FORM(RP_FULL);
#undef NL_FORM
#undef NL_LAUNCH
	return 0;
}

extern "C" int pga_n_local(pga_ctx_t *c, const int32_t *pairs, int64_t n, int32_t local_dist, int32_t local_count, int32_t frag_mode, int32_t **cnt)
{
	int32_t *d_pairs = (int32_t *)c->pool.get(S_PAIRS, sizeof(int32_t) * 2 * (size_t)n + 16);
	if (!d_pairs) return PGA_ERR_NOMEM;
	if (n) TRY(upload(c, d_pairs, pairs, 2 * (size_t)n));
	TRY(n_local_dev(c, d_pairs, n, nullptr, local_dist, local_count, frag_mode, cnt));
	return sync_st(c); // pairs is caller memory; the exchange may run on another stream
}

static inline bool loop_cap_forced(const pga_ctx *c) { static const bool on = getenv("PANGENE_LOOP_PAIR_CAP") != nullptr; return on && c->in_loop && !c->loop_room_given; }
static inline size_t loop_btot_at(int n_vtx) { return ((size_t)n_vtx + 7) & ~(size_t)3; } // the totals of k_loop_front1's workgroups, behind the vertices' counts in S_BR_PC

// pg_gen_rep_pos unfused after all (the launches rep_pos_impl left out)
static void rep_launch_deferred(pga_ctx *c, const RepDefer &d)
{
	const RepFill &rf = d.rf;
	hipLaunchKernelGGL(k_rank_genome, dim3((unsigned)rf.GL), dim3(RK_T), 0, c->st, rf.flags, rf.goff, d.rx, rf.gate);
	const int form = c->rp_form;
	if (d.cleared) {
		if (rf.n_ent && form == RP_COMPACT) hipLaunchKernelGGL((k_rep_clear<RP_COMPACT>), dim3(nblk(rf.n_ent)), dim3(BLOCK), 0, c->st, rf.rp_out, rf.n_ent, rf.gate);
		else if (rf.n_ent) hipLaunchKernelGGL((k_rep_clear<RP_FULL>), dim3(nblk(rf.n_ent)), dim3(BLOCK), 0, c->st, rf.rp_out, rf.n_ent, rf.gate);
		if (form == RP_COMPACT) hipLaunchKernelGGL((k_rep_fill<RP_COMPACT, true>), dim3(d.nb), dim3(BLOCK), 0, c->st, rf);
		else if (form == RP_WIDE) hipLaunchKernelGGL((k_rep_fill<RP_WIDE, true>), dim3(d.nb), dim3(BLOCK), 0, c->st, rf);
		else hipLaunchKernelGGL((k_rep_fill<RP_FULL, true>), dim3(d.nb), dim3(BLOCK), 0, c->st, rf);
	} else {
		if (form == RP_COMPACT) hipLaunchKernelGGL((k_rep_fill<RP_COMPACT, false>), dim3(d.nb), dim3(BLOCK), 0, c->st, rf);
		else if (form == RP_WIDE) hipLaunchKernelGGL((k_rep_fill<RP_WIDE, false>), dim3(d.nb), dim3(BLOCK), 0, c->st, rf);
		else hipLaunchKernelGGL((k_rep_fill<RP_FULL, false>), dim3(d.nb), dim3(BLOCK), 0, c->st, rf);
	}
}

// enumerate the pairs and count them (k_br_wave<1>, k_n_local) for the pair count in dcnt[15] (capacity c->br_cap)
static int branch_enumerate(pga_ctx *c, int32_t **cnt, const RepDefer *df = nullptr)
{
	const int n_vtx = 2 * c->br_S;
	int32_t *s1 = (int32_t *)c->pool.get(S_BR_S1, 0), *agid = (int32_t *)c->pool.get(S_BR_GID, 0), *vs = (int32_t *)c->pool.get(S_BR_VS, 0), *ve = (int32_t *)c->pool.get(S_BR_VE, 0);
	int32_t *poff = (int32_t *)c->pool.get(S_BR_POFF, 0);
	int32_t *pairs = (int32_t *)c->pool.get(S_PAIRS, sizeof(int32_t) * 2 * (size_t)c->br_cap + 16);
	if (!pairs || !s1 || !agid || !vs || !ve || !poff) return PGA_ERR_NOMEM;
	if (df) { // the records of pg_gen_rep_pos and the pair list in one launch (k_loop_front1 ran: ranks, counts, offsets inside each workgroup of 1 024 vertices)
		const int32_t *pc = (const int32_t *)c->pool.get(S_BR_PC, 0);
		const LoopFront2 a = { (int)df->nb, n_vtx, (n_vtx + RK_T - 1) / RK_T, vs, ve, s1, agid, c->br_par.diff, poff, pc + loop_btot_at(n_vtx), pc, pairs, c->br_cap, c->dcnt };
		const dim3 grid(df->nb + nblk(n_vtx, BLOCK / WAVE));
		const int form = c->rp_form;
		if (df->cleared) {
			if (form == RP_COMPACT) hipLaunchKernelGGL((k_loop_front2<RP_COMPACT, true>), grid, dim3(BLOCK), 0, c->st, df->rf, a);
			else if (form == RP_WIDE) hipLaunchKernelGGL((k_loop_front2<RP_WIDE, true>), grid, dim3(BLOCK), 0, c->st, df->rf, a);
			else hipLaunchKernelGGL((k_loop_front2<RP_FULL, true>), grid, dim3(BLOCK), 0, c->st, df->rf, a);
		} else {
			if (form == RP_COMPACT) hipLaunchKernelGGL((k_loop_front2<RP_COMPACT, false>), grid, dim3(BLOCK), 0, c->st, df->rf, a);
			else if (form == RP_WIDE) hipLaunchKernelGGL((k_loop_front2<RP_WIDE, false>), grid, dim3(BLOCK), 0, c->st, df->rf, a);
			else hipLaunchKernelGGL((k_loop_front2<RP_FULL, false>), grid, dim3(BLOCK), 0, c->st, df->rf, a);
		}
	} else
	hipLaunchKernelGGL((k_br_wave<1>), dim3(nblk(n_vtx, BLOCK / WAVE)), dim3(BLOCK), 0, c->st, n_vtx, vs, ve, s1, agid, c->br_par.diff, poff, pairs, c->br_cap, (const int32_t *)c->pool.get(S_BR_PC, 0),
	                   (const int32_t *)nullptr, 0.0, 0.0, (uint8_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, c->dcnt, (uint8_t *)nullptr, (const int64_t *)nullptr, c->gate);
	return n_local_dev(c, pairs, c->br_cap, c->dcnt + 15, c->br_par.local_dist, c->br_par.local_count, c->br_par.frag_mode, cnt);
}

static int branch_pairs_impl(pga_ctx *c, const uint64_t *arc_x, const int32_t *arc_s1, int64_t n_arc, const int32_t *seg_gid, int32_t n_seg,
                             double branch_diff, int32_t local_dist, int32_t local_count, int32_t frag_mode, int32_t **cnt, int64_t *n_pairs, const RepDefer *df)
{
	if (arc_x == nullptr) n_arc = c->br_n, n_seg = c->br_S; // the table of pga_arc_set_current
	const int n_vtx = 2 * n_seg;
	uint64_t *ax = (uint64_t *)c->pool.get(S_ARCX, sizeof(uint64_t) * (size_t)n_arc + 16);
	uint8_t *aw = (uint8_t *)c->pool.get(S_ARCW, (size_t)n_arc + 16);
	int32_t *s1 = (int32_t *)c->pool.get(S_BR_S1, sizeof(int32_t) * (size_t)n_arc + 16), *agid = (int32_t *)c->pool.get(S_BR_GID, sizeof(int32_t) * (size_t)n_arc + 16);
	int32_t *vs = (int32_t *)c->pool.get(S_BR_VS, sizeof(int32_t) * (size_t)n_vtx + 16), *ve = (int32_t *)c->pool.get(S_BR_VE, sizeof(int32_t) * (size_t)n_vtx + 16);
	int32_t *pc = (int32_t *)c->pool.get(S_BR_PC, sizeof(int32_t) * (loop_btot_at(n_vtx) + 64) + 16), *poff = (int32_t *)c->pool.get(S_BR_POFF, sizeof(int32_t) * (size_t)n_vtx + 16);
	int32_t *sg = (int32_t *)c->pool.get(S_BR_SEGGID, sizeof(int32_t) * (size_t)n_seg + 16);
	if (!ax || !aw || !s1 || !agid || !vs || !ve || !pc || !poff || !sg) return PGA_ERR_NOMEM;
	c->br_n = n_arc, c->br_S = n_seg, c->br_np = -1;
	c->br_par.diff = branch_diff, c->br_par.local_dist = local_dist, c->br_par.local_count = local_count, c->br_par.frag_mode = frag_mode;
	if (n_pairs) *n_pairs = 0;
	*cnt = (int32_t *)c->pool.get(S_NLCNT, 16);
	if (df && !df->on) df = nullptr;
	if (n_arc == 0 || n_vtx == 0) { if (df) rep_launch_deferred(c, *df); c->br_np = 0; return sync_st(c); }
	if (arc_x) {
		TRY(upload(c, ax, arc_x, (size_t)n_arc)); TRY(upload(c, s1, arc_s1, (size_t)n_arc)); TRY(upload(c, sg, seg_gid, (size_t)n_seg));
		HIPCHK(hipMemsetAsync(vs, 0, sizeof(int32_t) * (size_t)n_vtx, c->st)); HIPCHK(hipMemsetAsync(ve, 0, sizeof(int32_t) * (size_t)n_vtx, c->st));
		hipLaunchKernelGGL(k_br_prep, dim3(nblk(n_arc)), dim3(BLOCK), 0, c->st, ax, n_arc, sg, agid, vs, ve);
		HIPCHK(hipMemsetAsync(aw, 0, (size_t)n_arc, c->st)); // (the tables of arc_round_local / arc_set_current arrive with weak_br = 0)
	}
	static const bool general_scan = getenv("PANGENE_PAIR_SCAN_GENERAL") != nullptr; // (tests: the path of graphs with more than 65536 vertices)
	const bool one_wg = n_vtx <= PO_THREADS * PO_MAX_ITEMS && !general_scan;
	if (df && !(one_wg && n_pairs == nullptr)) { rep_launch_deferred(c, *df); df = nullptr; }
	if (df) { // (pga_branch_loop) the counts, their offsets inside workgroups of 1 024 vertices, the ranks of pg_gen_rep_pos and its clear in ONE launch
		const int nbc = (n_vtx + RK_T - 1) / RK_T; // <= 64 (one_wg)
		const RepFill &rf = df->rf;
		const int64_t n_clear = df->cleared ? rf.n_ent : 0;
		const LoopFront a = { n_vtx, nbc, rf.GL, vs, ve, s1, branch_diff, pc, poff, pc + loop_btot_at(n_vtx), rf.flags, rf.goff, df->rx, rf.rp_out, n_clear, c->rp_form == RP_COMPACT ? 8 : 16, c->gate };
		hipLaunchKernelGGL(k_loop_front1, dim3((unsigned)(nbc + rf.GL + (n_clear + RK_T - 1) / RK_T)), dim3(RK_T), 0, c->st, a);
		if (c->br_cap < 4 * (int64_t)n_vtx && !loop_cap_forced(c)) c->br_cap = 4 * (int64_t)n_vtx;
		return branch_enumerate(c, cnt, df);
	}
	hipLaunchKernelGGL(k_br_count, dim3(nblk(n_vtx)), dim3(BLOCK), 0, c->st, n_vtx, vs, ve, s1, branch_diff, pc, c->gate);
	if (one_wg) hipLaunchKernelGGL(k_pair_offsets, dim3(1), dim3(PO_THREADS), 0, c->st, (const int32_t *)pc, n_vtx, poff, c->dcnt, c->h_box, n_pairs ? -1ll : loop_cap_forced(c) ? (long long)c->br_cap : (long long)std::max<int64_t>(c->br_cap, 4 * (int64_t)n_vtx), c->gate); // offsets, and dcnt[15] = number of pairs
	else {
		I32 *tile = (I32 *)c->pool.get(S_TILE, tile_buf_bytes(n_vtx));
		device_scan<I32>(InI32{pc}, OutExclI32{poff}, n_vtx, tile, OpSum{}, I32{0}, c->st, c->gate);
		hipLaunchKernelGGL(k_mail_pairs, dim3(1), dim3(64), 0, c->st, poff + (n_vtx - 1), pc + (n_vtx - 1), c->dcnt, c->h_box);
	}
	if (n_pairs) { // somebody outside needs the count (the all-reduce of a sharded run): wait for it and size the buffers exactly
		TRY(sync_st(c));
		c->br_np = c->h_cnt[15], *n_pairs = c->br_np;
		c->br_cap = std::max<int64_t>(c->br_cap, std::max<int64_t>(c->br_np, 16)); // (never shrinks: lists queued earlier may still be in use)
		return c->br_np ? branch_enumerate(c, cnt) : 0;
	}
	// otherwise nothing waits: the buffers keep the capacity that was enough so far, pga_branch_decide checks the count when
	// it has to wait for its own results anyway and repeats the enumeration in the (first-round) case that it was not
	if (c->br_cap < 4 * (int64_t)n_vtx && !loop_cap_forced(c)) c->br_cap = 4 * (int64_t)n_vtx;
	return branch_enumerate(c, cnt);
}

extern "C" int pga_branch_pairs(pga_ctx_t *c, const uint64_t *arc_x, const int32_t *arc_s1, int64_t n_arc, const int32_t *seg_gid, int32_t n_seg,
                                double branch_diff, int32_t local_dist, int32_t local_count, int32_t frag_mode, int32_t **cnt, int64_t *n_pairs)
{
	return branch_pairs_impl(c, arc_x, arc_s1, n_arc, seg_gid, n_seg, branch_diff, local_dist, local_count, frag_mode, cnt, n_pairs, nullptr);
}

// pg_flt_high_occ's three tests (graph.c:226-258) on the device, so that a branch round's bulk results need not travel
struct RoundFilter { int on; int32_t max_tot_cnt, max_degree, max_dist_loci; uint8_t *del_host; };

static int decide_impl(pga_ctx *c, double branch_diff, double branch_diff_dist, double branch_diff_cut, uint8_t *arc_weak,
                       int32_t *n_dist_loci, int64_t *n_flt1, int64_t *n_flt2, const RoundFilter *rf)
{
	const int n_vtx = 2 * c->br_S, S = c->br_S;
	const int64_t n_arc = c->br_n;
	if (n_flt1) *n_flt1 = 0;
	if (n_flt2) *n_flt2 = 0;
	if (n_arc == 0 || n_vtx == 0) {
		if (n_vtx && n_dist_loci) memset(n_dist_loci, 0, sizeof(int32_t) * (size_t)n_vtx);
		if (rf && rf->on && S) memset(rf->del_host, 0, (size_t)S);
		return rf ? sync_st(c) : 0;
	}
	uint8_t *aw = (uint8_t *)c->pool.get(S_ARCW, 0);
	int32_t *s1 = (int32_t *)c->pool.get(S_BR_S1, 0), *agid = (int32_t *)c->pool.get(S_BR_GID, 0), *vs = (int32_t *)c->pool.get(S_BR_VS, 0), *ve = (int32_t *)c->pool.get(S_BR_VE, 0);
	int32_t *poff = (int32_t *)c->pool.get(S_BR_POFF, 0);
	int32_t *grp = (int32_t *)c->pool.get(S_BR_GRP, sizeof(int32_t) * (size_t)n_arc + 16);
	uint8_t *vwk = (uint8_t *)c->pool.get(S_VWK, (size_t)n_vtx + 16);
	const size_t need = sizeof(int32_t) * (size_t)n_vtx + 64;
	if (c->h_ndl_cap < need) {
		if (c->h_ndl) HIPCHK(hipStreamSynchronize(c->st));
		c->h_ndl = (int32_t *)c->pin.get(need + need / 2);
		if (!c->h_ndl) return PGA_ERR_NOMEM;
		c->h_ndl_cap = need + need / 2;
	}
	int32_t *ndl_dev = nullptr;
	HIPCHK(hipHostGetDevicePointer((void **)&ndl_dev, c->h_ndl, 0)); // n_dist_loci goes straight into pinned host memory ...
	int32_t *ndl_out = ndl_dev;
	if (rf) { // ... unless only the device looks at it: then the pinned buffer carries the per-segment verdicts instead
		ndl_out = (int32_t *)c->pool.get(S_BR_NDL, sizeof(int32_t) * (size_t)n_vtx + 16);
		if (!ndl_out) return PGA_ERR_NOMEM;
	}
	if (!grp || !vwk) return PGA_ERR_NOMEM;
	for (int attempt = 0;; ++attempt) {
		int32_t *cnt = (int32_t *)c->pool.get(S_NLCNT, 0);
		if (n_flt1 || n_flt2) HIPCHK(hipMemsetAsync(c->dcnt, 0, 2 * sizeof(int64_t), c->st)); // [0], [1]: arcs marked 1 / 2 (log only)
		hipLaunchKernelGGL((k_br_wave<2>), dim3(nblk(n_vtx, BLOCK / WAVE)), dim3(BLOCK), 0, c->st, n_vtx, vs, ve, s1, agid, branch_diff, poff, (int32_t *)nullptr, (int64_t)c->br_cap, (const int32_t *)nullptr, cnt,
		                   branch_diff_dist, branch_diff_cut, aw, grp, ndl_out, (n_flt1 || n_flt2) ? c->dcnt : (int64_t *)nullptr, vwk, c->br_np < 0 ? c->dcnt + 15 : (const int64_t *)nullptr);
		if (rf && rf->on)
			hipLaunchKernelGGL(k_round_filter, dim3(nblk(S)), dim3(BLOCK), 0, c->st, S, (const int32_t *)c->pool.get(S_SEGCNT, 0), (const int32_t *)c->pool.get(S_DEG, 0), (const int32_t *)ndl_out,
			                   rf->max_tot_cnt, rf->max_degree, rf->max_dist_loci, (uint8_t *)ndl_dev);
		if (arc_weak && !c->table_sparse) HIPCHK(hipMemcpyAsync(arc_weak, aw, (size_t)n_arc, hipMemcpyDeviceToHost, c->st));
		if (n_flt1 || n_flt2) hipLaunchKernelGGL(k_mail_flush, dim3(1), dim3(64), 0, c->st, c->dcnt, c->h_box);
		TRY(sync_st(c));
		if (c->br_np >= 0 || c->h_cnt[15] <= c->br_cap || attempt) { if (c->br_np < 0) c->br_np = c->h_cnt[15]; c->br_np_seen = c->br_np; break; }
		// more pairs than the buffers held (pairs beyond the capacity were neither listed nor counted): enumerate again, with room
		c->br_cap = c->h_cnt[15] + c->h_cnt[15] / 2;
		HIPCHK(hipMemsetAsync(aw, 0, (size_t)n_arc, c->st)); HIPCHK(hipMemsetAsync(vwk, 0, (size_t)n_vtx, c->st));
		int32_t *dummy;
		TRY(branch_enumerate(c, &dummy));
	}
	if (n_dist_loci && !rf) memcpy(n_dist_loci, c->h_ndl, sizeof(int32_t) * (size_t)n_vtx);
	if (rf && rf->on) memcpy(rf->del_host, c->h_ndl, (size_t)S);
	if (n_flt1) *n_flt1 = c->h_cnt[0];
	if (n_flt2) *n_flt2 = c->h_cnt[1];
	return 0;
}

extern "C" int pga_branch_decide(pga_ctx_t *c, double branch_diff, double branch_diff_dist, double branch_diff_cut, uint8_t *arc_weak,
                                 int32_t *n_dist_loci, int64_t *n_flt1, int64_t *n_flt2)
{
	if (n_dist_loci == nullptr) return PGA_ERR_ARG;
	return decide_impl(c, branch_diff, branch_diff_dist, branch_diff_cut, arc_weak, n_dist_loci, n_flt1, n_flt2, nullptr);
}

extern "C" int pga_branch_decide_filter(pga_ctx_t *c, double branch_diff, double branch_diff_dist, double branch_diff_cut, int32_t do_filter,
                                        int32_t max_tot_cnt, int32_t max_degree, int32_t max_dist_loci, uint8_t *del)
{
	// only behind a deferred round on the gene-major path: its segment counters and degrees are then where k_round_filter looks
	if (!(c->arc_deferred && !c->arc_done && c->table_sparse) || c->br_S != c->n_seg || (do_filter && del == nullptr)) return 2;
	RoundFilter rf = { do_filter, max_tot_cnt, max_degree, max_dist_loci, del };
	return decide_impl(c, branch_diff, branch_diff_dist, branch_diff_cut, nullptr, nullptr, nullptr, nullptr, &rf);
}

// entries of a rank's slot: what the previous run over the shard needed (the largest local table of any rank in any round) with a margin,
// never less than the largest table the host-driven rounds have seen; before there is a previous run, that with a wide margin
static int64_t x_arc_cap(const pga_ctx *c, const pga_loop_xchg_t *x)
{
	const int64_t m = std::max<int64_t>(c->x_arcs_seen, x->arc_cap_hint);
	// (before there is a previous run: the tables of the branch rounds grow to a multiple of the first graphs' -- 3.5x at configs[1])
	return std::max<int64_t>(c->x_arc_floor, c->x_arcs_seen > 0 ? m + m / 8 + 1024 : 5 * m + 4096);
}

// (sharded form) the round's local table -> every rank's slot -> the merged table as the current one: pga_arc_round's compaction, the
// all-gather, pga_arc_merge and pga_arc_set_current with every count left in device memory
struct LoopX { const pga_loop_xchg_t *x; int64_t arc_cap, pair_cap, ecap; int32_t *gbuf; int64_t slot_words; pga_arc_part_t *merged; int64_t *xstat, *d_off; };

// local_gate: the gate of this rank's own arc round (it may have found nothing to do: the slot then stands as it is); stamp / round: the loop's fixed-point words
static int loop_exchange_table(pga_ctx *c, const LoopX &L, Gate local_gate = Gate{nullptr, 0}, int32_t *stamp = nullptr, int round = 0)
{
	const int S = c->n_seg, n_vtx = 2 * S, W = L.x->world;
	const int64_t mcap = (int64_t)W * L.arc_cap;
	pga_arc_part_t *stage = (pga_arc_part_t *)c->pool.get(S_ARC_STAGE, 0);
	int4 *gmeta = (int4 *)c->pool.get(S_GMETA, 0);
	int32_t *goff = (int32_t *)c->pool.get(S_GOFF, sizeof(int32_t) * (size_t)S);
	int32_t *seg_cnt = (int32_t *)c->pool.get(S_SEGCNT, sizeof(int32_t) * 2 * (size_t)std::max(1, S) * SEGCNT_COPIES);
	I32 *tile = (I32 *)c->pool.get(S_TILE, tile_buf_bytes(std::max<int64_t>(std::max<int64_t>(S, mcap), 2 * (int64_t)c->N + 2)));
	uint64_t *key = (uint64_t *)c->pool.get(S_MG_KEY, sizeof(uint64_t) * (size_t)mcap + 64);
	uint32_t *val = (uint32_t *)c->pool.get(S_MG_VAL, sizeof(uint32_t) * (size_t)mcap + 64);
	int32_t *slot = (int32_t *)c->pool.get(S_MG_SLOT, sizeof(int32_t) * (size_t)mcap + 64);
	if (!goff || !tile || !key || !val || !slot || (c->N && (!stage || !gmeta || !seg_cnt))) return PGA_ERR_NOMEM;
	if (c->N) {
		device_scan<I32>(InGmeta{gmeta}, OutExclI32{goff}, S, tile, OpSum{}, I32{0}, c->st, local_gate);
		hipLaunchKernelGGL(k_xs_compact, dim3(nblk(S, BLOCK / WAVE)), dim3(BLOCK), 0, c->st, gmeta, goff, S, stage, seg_cnt, L.gbuf, L.arc_cap, (const int64_t *)c->dcnt, local_gate, (const int32_t *)stamp, round);
	}
	else HIPCHK(hipMemsetAsync(L.gbuf, 0, sizeof(int32_t) * (size_t)(XS_HDR + xs_seg_words(S)), c->st)); // a rank without hits: an empty table, no counts
	{ const int rc = L.x->allgather(L.x->user, L.gbuf, L.gbuf + L.slot_words, L.slot_words * (int64_t)sizeof(int32_t)); if (rc) return rc; }
	XSlots X = { L.gbuf + L.slot_words, L.slot_words, L.arc_cap, W, S };
	hipLaunchKernelGGL(k_xs_sum_rank, dim3(nblk(std::max<int64_t>(mcap, n_vtx))), dim3(BLOCK), 0, c->st, X, seg_cnt, L.d_off, c->dcnt, L.xstat, key, val, stamp, round);
	device_scan<I32>(InMgHeadN{key, L.d_off + W}, OutExclI32{slot}, mcap, tile, OpSum{}, I32{0}, c->st);
	hipLaunchKernelGGL(k_mgx_heads_sum, dim3(nblk(mcap)), dim3(BLOCK), 0, c->st, X, (const uint64_t *)key, (const uint32_t *)val, (const int32_t *)slot, (const int64_t *)(L.d_off + W), L.merged, c->dcnt + 10);
	CurTable t;
	TRY(cur_table(c, L.ecap, S, &t));
	// (t.sg, the gene of every segment, stands: the gene kernels write it for every live segment, and a deleted one keeps its number)
	hipLaunchKernelGGL(k_curx_table, dim3(nblk(mcap + n_vtx)), dim3(BLOCK), 0, c->st, (const pga_arc_part_t *)L.merged, (const int64_t *)(c->dcnt + 10), mcap, (const int32_t *)t.sg, n_vtx, t.ax, t.s1, t.agid, t.aw, t.vs, t.ve, t.dg, t.vwk);
	c->table_sparse = false, c->cur_tab = L.merged, c->cur_tab_n = 0; // (the size stays on the device: nobody may ask for this table -- the loop's caller runs a round of its own next)
	return 0;
}

// pg_gen_arc of a sharded run with ONE wait: pga_arc_round + the exchange + pga_arc_merge + pga_arc_set_current, the table sizes left
// in device memory (the ranks' tables travel in slots of a capacity all ranks share, see pga_loop_xchg_t).  seg_cnt_host[2S], deg_host[2S]
// and *n_arc are the global results.  1 = the round is void on some rank (a hub gene beyond its LDS table, a table beyond the slot):
// every rank gets 1 and repeats the round through pga_arc_round (which then takes the sort path, without a second sweep);
// 2 = not applicable (no capacity known yet: the first round of a shard is host-driven).
extern "C" int pga_arc_round_x(pga_ctx_t *c, int32_t use_ori, int32_t n_seg, const pga_loop_xchg_t *x, int32_t *seg_cnt_host, int32_t *deg_host, int64_t *n_arc)
{
	const int S = n_seg, n_vtx = 2 * S, N = c->N;
	if (x == nullptr || x->world < 1 || x->allgather == nullptr || S != c->n_seg || S == 0 || arc_sort_path_forced()) return 2;
	if (c->x_arcs_seen <= 0 && x->arc_cap_hint <= 0) return 2;
	LoopX L = { x, 0, 0, 0, nullptr, 0, nullptr, nullptr, nullptr };
	L.arc_cap = x_arc_cap(c, x);
	{ const long long ea = xloop_cap(1); if (ea && c->x_arcs_seen == 0) L.arc_cap = std::max<int64_t>(ea, 1); }
	if ((int64_t)x->world * L.arc_cap >= ((int64_t)1 << 31)) return 2; // (entries of the gathered tables are numbered in 32 bits)
	L.ecap = std::max<int64_t>(2 * (int64_t)N + 2, (int64_t)x->world * L.arc_cap);
	L.slot_words = xs_slot_words(S, L.arc_cap);
	L.gbuf = (int32_t *)c->pool.get(S_XG_BUF, sizeof(int32_t) * (size_t)L.slot_words * ((size_t)x->world + 1) + 64);
	L.merged = (pga_arc_part_t *)c->pool.get(S_XG_OUT, sizeof(pga_arc_part_t) * (size_t)x->world * (size_t)L.arc_cap + 64);
	L.xstat = (int64_t *)c->pool.get(S_XSTAT, sizeof(int64_t) * 8);
	L.d_off = (int64_t *)c->pool.get(S_MG_SRC, sizeof(int64_t) * ((size_t)x->world + 2));
	if (!L.gbuf || !L.merged || !L.xstat || !L.d_off) return PGA_ERR_NOMEM;
	const size_t need = sizeof(int32_t) * (2 * (size_t)n_vtx) + 8 * sizeof(int64_t) + 64;
	if (c->h_round_cap < need) {
		if (c->h_round) HIPCHK(hipStreamSynchronize(c->st));
		c->h_round = (int32_t *)c->pin.get(need + need / 2);
		if (!c->h_round) return PGA_ERR_NOMEM;
		c->h_round_cap = need + need / 2;
	}
	c->arc_deferred = false, c->arc_done = false;
	HIPCHK(hipMemsetAsync(L.xstat, 0, sizeof(int64_t) * 8, c->st));
	HIPCHK(hipMemsetAsync(c->dcnt + 9, 0, sizeof(int64_t), c->st));
	HIPCHK(hipMemsetAsync(c->dcnt + 11, 0, sizeof(int64_t), c->st));
	CurTable t;
	TRY(cur_table(c, L.ecap, S, &t));
	if (N) { int32_t *seg_cnt, *deg; TRY(arc_round_genes(c, use_ori, &seg_cnt, &deg, nullptr, false)); }
	hipLaunchKernelGGL(k_seg_gid, dim3(nblk(c->Q)), dim3(BLOCK), 0, c->st, c->g2s, c->Q, S, t.sg); // (a rank without hits ran no gene kernel)
	TRY(loop_exchange_table(c, L));
	int32_t *seg_cnt = (int32_t *)c->pool.get(S_SEGCNT, 0);
	int64_t *tail = (int64_t *)(c->h_round + 2 * (size_t)n_vtx + ((2 * (size_t)n_vtx) & 1)); // 8-byte aligned, behind the two vectors
	HIPCHK(hipMemcpyAsync(c->h_round, seg_cnt, sizeof(int32_t) * (size_t)n_vtx, hipMemcpyDeviceToHost, c->st));
	HIPCHK(hipMemcpyAsync(c->h_round + n_vtx, t.dg, sizeof(int32_t) * (size_t)n_vtx, hipMemcpyDeviceToHost, c->st));
	HIPCHK(hipMemcpyAsync(tail, L.xstat, 8 * sizeof(int64_t), hipMemcpyDeviceToHost, c->st));
	hipLaunchKernelGGL(k_mail_flush, dim3(1), dim3(64), 0, c->st, c->dcnt, c->h_box);
	TRY(sync_st(c));
	c->br_n = L.ecap, c->br_S = S, c->br_np = 0;
	if (tail[4] || c->h_cnt[3]) return PGA_ERR_INVARIANT;
	c->x_arcs_run = std::max<int64_t>(c->x_arcs_run, tail[1]);
	if (tail[2]) c->x_arcs_seen = std::max<int64_t>(c->x_arcs_seen, tail[1]), c->x_arc_floor = std::max<int64_t>(c->x_arc_floor, tail[1] + tail[1] / 4 + 1024); // a table beyond its slot: the next round of this run already knows
	if (tail[2] || tail[3]) { c->x_redo = true; return 1; } // (from the gathered slots alone -- every local cause is in the rank's header: the same verdict on every rank)
	memcpy(seg_cnt_host, c->h_round, sizeof(int32_t) * (size_t)n_vtx), memcpy(deg_host, c->h_round + n_vtx, sizeof(int32_t) * (size_t)n_vtx);
	c->cur_tab = L.merged, c->cur_tab_n = c->h_cnt[10], c->table_sparse = false;
	*n_arc = c->h_cnt[10];
	return 0;
}

extern "C" int pga_branch_loop(pga_ctx_t *c, int32_t n_round, const pga_branch_par_t *par, const int32_t *max_tot_cnt, const int32_t *max_degree,
                               const int32_t *max_dist_loci, uint8_t *seg_alive, const pga_loop_xchg_t *x, int32_t *seg_cnt_host, int32_t *ndl_host)
{
	static const bool off = getenv("PANGENE_BRANCH_LOOP_HOST") != nullptr; // (tests: keep the host-driven rounds exercised)
	const int S = c->n_seg, n_vtx = 2 * S, N = c->N;
	if (off || n_round <= 0 || par == nullptr || seg_alive == nullptr || N == 0 || S == 0 || c->br_S != S || n_vtx > PO_THREADS * PO_MAX_ITEMS) return 2;
	if (par->final_on && (seg_cnt_host == nullptr || ndl_host == nullptr)) return 2;
	if (x == nullptr ? !(c->arc_deferred && !c->arc_done && c->table_sparse) : (c->table_sparse || c->arc_deferred || x->world < 1 || x->allgather == nullptr || x->allreduce_i32_sum == nullptr)) return 2;
	uint8_t *alive = (uint8_t *)c->pool.get(S_MISC, (size_t)S + 64);
	int32_t *ndl = (int32_t *)c->pool.get(S_BR_NDL, sizeof(int32_t) * (size_t)n_vtx + 16);
	if (!alive || !ndl) return PGA_ERR_NOMEM;
	HIPCHK(hipMemsetAsync(alive, 1, (size_t)S, c->st));
	const int64_t br_cap_before = c->br_cap;
	LoopX L = { x, 0, 0, 0, nullptr, 0, nullptr, nullptr, nullptr };
	if (x == nullptr) { // Room for the pair lists of every round (nobody can ask for more on the way): a vertex with n out-arcs lists at most n^2
	  // pairs (branch.c:70-88), and pg_flt_high_occ keeps n near max_degree (graph.c:243-250) -- the lists grow over the rounds,
	  // so the first round's length says little.  A list that still overflows costs a repeated run (sticky flag), not a wrong one.
		int64_t dmax = 8;
		for (int r = 1; r < n_round; ++r) dmax = std::max<int64_t>(dmax, max_degree[r]);
		c->br_cap = std::max<int64_t>(c->br_cap, std::min<int64_t>((int64_t)n_vtx * dmax * dmax, (int64_t)1 << 26));
		static const long long cap_env = [] { const char *e = getenv("PANGENE_LOOP_PAIR_CAP"); return e ? atoll(e) : 0ll; }(); // (tests: a first attempt with this little room -> status 4 -> a second one with what was needed)
		if (cap_env > 0 && !c->loop_room_given) c->br_cap = cap_env;
	} else {
		// Capacities all ranks share: they follow from the merged tables (identical everywhere) and from the slots of earlier all-gathers.
		// The pair list's worst case (above) is too much to all-reduce every round: what earlier runs over this shard saw, with a
		// margin, or a million pairs on the first run -- a list beyond that costs a repeated run (status 3), and the next one knows.
		int64_t dmax = 8;
		for (int r = 1; r < n_round; ++r) dmax = std::max<int64_t>(dmax, max_degree[r]);
		const int64_t worst = std::min<int64_t>((int64_t)n_vtx * dmax * dmax, (int64_t)1 << 26);
		L.pair_cap = std::min<int64_t>(worst, c->x_pairs_seen > 0 ? c->x_pairs_seen + c->x_pairs_seen / 8 + 4096 : std::max<int64_t>((int64_t)1 << 20, 4 * (int64_t)n_vtx));
		L.pair_cap = std::max<int64_t>(std::max<int64_t>(L.pair_cap, std::min<int64_t>(worst, c->x_pair_floor)), 4 * (int64_t)n_vtx); // (pga_branch_pairs never works with less)
		L.arc_cap = x_arc_cap(c, x); // (every slot travels at its capacity)
		if (c->x_pairs_seen == 0) { // (tests: start with buffers that are too small, to reach status 3 and the learned capacities)
			const long long ep = xloop_cap(0), ea = xloop_cap(1);
			if (ep) L.pair_cap = std::max<int64_t>(ep, 4 * (int64_t)n_vtx);
			if (ea && c->x_arcs_seen == 0) L.arc_cap = std::max<int64_t>(ea, 1);
		}
		c->br_cap = L.pair_cap; // what k_pair_offsets tests and k_br_wave / k_n_local stride over
		if ((int64_t)x->world * L.arc_cap >= ((int64_t)1 << 31)) return 2; // (entries of the gathered tables are numbered in 32 bits)
		L.ecap = std::max<int64_t>(2 * (int64_t)N + 2, (int64_t)x->world * L.arc_cap);
		L.slot_words = xs_slot_words(S, L.arc_cap);
		L.gbuf = (int32_t *)c->pool.get(S_XG_BUF, sizeof(int32_t) * (size_t)L.slot_words * ((size_t)x->world + 1) + 64);
		L.merged = (pga_arc_part_t *)c->pool.get(S_XG_OUT2, sizeof(pga_arc_part_t) * (size_t)x->world * (size_t)L.arc_cap + 64); // (not the slot the current table may live in: it is read below)
		L.xstat = (int64_t *)c->pool.get(S_XSTAT, sizeof(int64_t) * 8);
		L.d_off = (int64_t *)c->pool.get(S_MG_SRC, sizeof(int64_t) * ((size_t)x->world + 2));
		if (!L.gbuf || !L.merged || !L.xstat || !L.d_off) return PGA_ERR_NOMEM;
		HIPCHK(hipMemsetAsync(L.xstat, 0, sizeof(int64_t) * 8, c->st));
		HIPCHK(hipMemsetAsync(c->dcnt + 11, 0, sizeof(int64_t), c->st)); // the sticky flag covers the queued rounds
		HIPCHK(hipMemsetAsync(c->dcnt + 9, 0, sizeof(int64_t), c->st));
		// the table the caller made current (pga_arc_set_current), once more into arrays that also hold every later round's
		CurTable t;
		const pga_arc_part_t *tab = c->cur_tab; const int64_t n0 = c->cur_tab_n;
		if (n0 > L.ecap) return 2;
		TRY(cur_table(c, L.ecap, S, &t));
		if (!c->pool.get(S_BR_GRP, sizeof(int32_t) * (size_t)L.ecap + 16)) return PGA_ERR_NOMEM;
		zero_multi(c, t.vs, sizeof(int32_t) * (size_t)n_vtx, t.ve, sizeof(int32_t) * (size_t)n_vtx, t.aw, (size_t)L.ecap, t.vwk, (size_t)n_vtx);
		if (n0) {
			hipLaunchKernelGGL(k_seg_gid, dim3(nblk(c->Q)), dim3(BLOCK), 0, c->st, c->g2s, c->Q, S, t.sg);
			hipLaunchKernelGGL(k_cur_prep, dim3(nblk(n0)), dim3(BLOCK), 0, c->st, tab, n0, t.sg, t.ax, t.s1, t.agid, t.vs, t.ve);
		}
		hipLaunchKernelGGL(k_deg, dim3(nblk(n_vtx)), dim3(BLOCK), 0, c->st, t.vs, t.ve, n_vtx, t.dg);
		c->br_n = L.ecap;
	}
	bool first_x = true; // the first exchange of this call writes the rank's slot whatever the gates say (what the buffer holds from an earlier call is no slot of this layout)
	if (par->pre_on) { // graph 2 (graph.c:293-296): pg_flt_high_occ on graph 1's table (no branch step has run: n_dist_loci = 0), PG_SET_FILTER(vtx == 0), pg_gen_arc
		// (sharded, round 6: the caller's table came through pga_arc_round_x, so the segment counters in device memory are the global ones and the
		// degrees those of the merged table -- every rank deletes the same segments)
		HIPCHK(hipMemsetAsync(ndl, 0, sizeof(int32_t) * (size_t)n_vtx, c->st));
		int32_t *vs = (int32_t *)c->pool.get(S_BR_VS, 0), *ve = (int32_t *)c->pool.get(S_BR_VE, 0), *sg = (int32_t *)c->pool.get(S_BR_SEGGID, 0), *dg = (int32_t *)c->pool.get(S_DEG, 0), *seg_cnt = (int32_t *)c->pool.get(S_SEGCNT, 0);
		uint8_t *vwk = (uint8_t *)c->pool.get(S_VWK, (size_t)n_vtx + 16);
		if (!vs || !ve || !sg || !dg || !seg_cnt || !vwk) return PGA_ERR_NOMEM;
		hipLaunchKernelGGL(k_round_del, dim3(nblk(S)), dim3(BLOCK), 0, c->st, S, (const int32_t *)ndl, par->pre_max_tot_cnt, par->pre_max_degree, par->pre_max_dist_loci, (const int32_t *)sg, c->g2s, vs, ve, dg, seg_cnt, vwk, alive,
		                   (x || par->final_on) ? (int4 *)c->pool.get(S_GMETA, 0) : (int4 *)nullptr);
		hipLaunchKernelGGL(k_flag_vtx, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->flags, c->gid, N, c->g2s, 1);
		c->walk_valid = false, c->ha_valid = false;
		int32_t *sc2, *deg2;
		TRY(arc_round_genes(c, par->use_ori, &sc2, &deg2, nullptr, false));
		c->br_n = 2 * (int64_t)N + 2, c->br_S = S, c->br_np = 0;
		if (x) { TRY(loop_exchange_table(c, L)); c->br_n = L.ecap; first_x = false; }
	}
	// The fixed point (dev_prims.hpp: Gate).  Inside this loop the tie orders stand still (the caller asked exact_quiet), so a round that
	// marks no hit and deletes no segment leaves a state every later round reproduces: their kernels are queued all the same -- the
	// host does not look -- and leave at once.  Human-shaped shards reach it after three or four of the fifteen rounds; bacterial ones
	// as a rule do not.  Sharded runs keep every round: their collectives are queued by the host, and "nothing changed" would have to
	// hold on every rank.
	static const bool no_skip = env_has("PANGENE_LOOP", "noskip");
	// Sharded (round 6): the stamps become a property of all ranks -- deletions are global anyway (every rank holds the merged table), the marks travel
	// in the header of the round's all-gather (k_xs_compact writes the word whether its gate is open or not) and k_xs_sum_rank stamps the round on every rank when any rank marked; a rank's own arc
	// round follows its own hits (nothing changed here: its slot stands as it is, k_xs_compact leaves), the collectives are queued all the same.
	const bool gated = !no_skip && (int64_t)c->round_tag + n_round + 4 < (int64_t)HA_TAG_MAX;
	struct GateScope { pga_ctx *c; ~GateScope() { c->gate = Gate{nullptr, 0}, c->loop_gated = false, c->in_loop = false; } } gate_scope{c}; // (every way out of this function leaves the launches open)
	c->loop_gated = gated, c->in_loop = true;
	const uint32_t tag_before = c->round_tag;
	HIPCHK(hipMemsetAsync(c->dcnt + 16, 0, sizeof(int64_t), c->st)); // the longest pair list that did not fit (k_loop_front2 / k_pair_offsets)
	if (gated) HIPCHK(hipMemsetAsync(c->loopctl, 0xff, 4 * sizeof(int32_t), c->st)); // -1: nothing has happened yet; round 0 runs (its branch steps ask for a change in round -1 or later)
	// Live lists inside the queue.  pg_flt_high_occ's first rounds delete segments wholesale on many-genome shards (1 250 bacterial
	// genomes: 98.5 % of the hits are without flt before the test of round 1, 38.5 % after it), so on shards where a wait is small beside a round
	// the loop asks after the deletions of rounds 1, 2, 4 and 8 how many hits are left (k_flag_vtx counts them) and, when a quarter of what the
	// lists hold has gone, builds them again before the round's pg_gen_arc.  PANGENE_LIVE_LISTS=0: never; =2: ask in every round, whatever the size.
	static const int live_env = [] { const char *e = getenv("PANGENE_LIVE_LISTS"); return e ? atoi(e) : -1; }();
	const bool live_ask = live_env != 0 && (live_env == 2 || N >= (1 << 21));
	for (int r = 0; r < n_round; ++r) {
		c->loop_round = r;
		bool rebuilt = false;
		c->gate = gated ? Gate{c->loopctl, r - 1} : Gate{nullptr, 0}; // the branch steps of round r: something changed in round r - 1
		// pg_mark_branch_flt_arc (branch.c:48-106)
		// (round 6: the five launches of these two steps' fronts as two, k_loop_front1 / 2 in k_branch.hpp; PANGENE_LOOP_FUSE=0: one by one as before)
		static const bool fuse = !env_has("PANGENE_LOOP_FUSE", "0");
		RepDefer df;
		TRY(rep_pos_impl(c, fuse ? &df : nullptr));
		int32_t *cnt;
		TRY(branch_pairs_impl(c, nullptr, nullptr, 0, nullptr, S, par->branch_diff, par->local_dist, par->local_count, par->frag_mode, &cnt, nullptr, fuse ? &df : nullptr));
		if (x) { const int rc = x->allreduce_i32_sum(x->user, cnt, L.pair_cap); if (rc) return rc; } // n_local over every rank's genomes (entries beyond the list's end: whatever they were)
		{
			const int64_t n_arc = c->br_n;
			uint8_t *aw = (uint8_t *)c->pool.get(S_ARCW, 0), *vwk = (uint8_t *)c->pool.get(S_VWK, (size_t)n_vtx + 16);
			int32_t *s1 = (int32_t *)c->pool.get(S_BR_S1, 0), *agid = (int32_t *)c->pool.get(S_BR_GID, 0), *vs = (int32_t *)c->pool.get(S_BR_VS, 0), *ve = (int32_t *)c->pool.get(S_BR_VE, 0);
			int32_t *poff = (int32_t *)c->pool.get(S_BR_POFF, 0), *grp = (int32_t *)c->pool.get(S_BR_GRP, sizeof(int32_t) * (size_t)n_arc + 16);
			int32_t *sg = (int32_t *)c->pool.get(S_BR_SEGGID, 0), *dg = (int32_t *)c->pool.get(S_DEG, 0), *seg_cnt = (int32_t *)c->pool.get(S_SEGCNT, 0);
			if (!aw || !vwk || !s1 || !agid || !vs || !ve || !poff || !grp || !sg || !dg || !seg_cnt) return PGA_ERR_NOMEM;
			hipLaunchKernelGGL((k_br_wave<2>), dim3(nblk(n_vtx, BLOCK / WAVE)), dim3(BLOCK), 0, c->st, n_vtx, vs, ve, s1, agid, par->branch_diff, poff, (int32_t *)nullptr, (int64_t)c->br_cap, (const int32_t *)nullptr, (const int32_t *)c->pool.get(S_NLCNT, 0),
			                   par->branch_diff_dist, par->branch_diff_cut, aw, grp, ndl, (int64_t *)nullptr, vwk, c->dcnt + 15, c->gate);
			// pg_mark_branch_flt_hit + PG_SET_FILTER(weak_br == 2) (branch.c:108-145, graph.c:309): with the numbering the arcs were made with
			TRY(pga_mark_hits(c, nullptr, nullptr, 0, nullptr, 1));
			c->gate = gated ? Gate{c->loopctl, r} : Gate{nullptr, 0}; // from here on: something changed in THIS round
			if (r > 0) { // pg_flt_high_occ + pg_hard_delete + PG_SET_FILTER(vtx == 0) (graph.c:219-263, 312): the thresholds tighten every round, so this test always runs
				hipLaunchKernelGGL(k_round_del, dim3(nblk(S)), dim3(BLOCK), 0, c->st, S, (const int32_t *)ndl, max_tot_cnt[r], max_degree[r], max_dist_loci[r], (const int32_t *)sg, c->g2s, vs, ve, dg, seg_cnt, vwk, alive,
				                   (x || par->final_on) ? (int4 *)c->pool.get(S_GMETA, 0) : (int4 *)nullptr, // (whoever compacts the genes' stretches afterwards must find a deleted one empty)
				                   gated ? c->loopctl : (int32_t *)nullptr, r);
				const bool ask = live_ask && (live_env == 2 || r == 1 || r == 2 || r == 4 || r == 8) && (r + 1 < n_round || par->final_on);
				if (ask) HIPCHK(hipMemsetAsync(c->live_cnt, 0, sizeof(int64_t) * LIVE_CNT_N, c->st));
				static const bool fv_x = env_has("PANGENE_FLAG_VTX", "x"); // (tests: every hit in every round, as before)
				if (c->live_on && c->z_valid && !fv_x && (int64_t)c->NL * 5 < (int64_t)N) // few members: along the index (40 bytes a member against 8 a hit)
					hipLaunchKernelGGL(k_flag_vtx_z, dim3(nblk(std::max(c->NL, 1))), dim3(BLOCK), 0, c->st, c->flags, (const int32_t *)c->zx, (const int32_t *)c->zg, c->NL, (const int32_t *)c->g2s, c->gate, ask ? c->live_cnt : (int64_t *)nullptr);
				else
				hipLaunchKernelGGL(k_flag_vtx, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->flags, c->gid, N, c->g2s, 1, c->gate, ask ? c->live_cnt : (int64_t *)nullptr);
				c->walk_valid = false, c->ha_valid = false;
				if (ask) {
					hipLaunchKernelGGL(k_live_sum, dim3(1), dim3(BLOCK), 0, c->st, (const int64_t *)c->live_cnt, c->dcnt);
					hipLaunchKernelGGL(k_mail_flush, dim3(1), dim3(64), 0, c->st, c->dcnt, c->h_box);
					TRY(sync_st(c));
					const int64_t left = c->h_cnt[8]; // (0: the round's kernels found their gate closed -- nothing was counted, nothing has changed)
					if (left > 0 && left * 4 <= (int64_t)c->NL * 3) {
						if (getenv("PANGENE_TIMING")) fprintf(stderr, "[pga_branch_loop] round %d: %lld of %d hits are without flt, the lists hold %d: built again\n", r, (long long)left, N, c->NL);
						TRY(build_z(c, left));
						rebuilt = true;
					}
				}
			}
		}
		if (r + 1 < n_round || par->final_on) { // pg_gen_arc (graph.c:313)
			int32_t *seg_cnt, *deg;
			// (new lists renumber the index: the half-arc records of the last walk mean nothing any more, so this arc round runs whatever the gate says --
			// it is open anyway when hits were filtered in this round, but the quarter may have gone over several rounds)
			// (sharded: so does the first arc round of the call -- the slot it fills must hold THIS rank's counters, and what the caller left in device
			// memory are the global ones)
			if (rebuilt || (x && first_x)) c->gate = Gate{nullptr, 0};
			TRY(arc_round_genes(c, par->use_ori, &seg_cnt, &deg, nullptr, false)); // (no mail: the kernels raise the sticky flag themselves)
			c->br_n = 2 * (int64_t)N + 2, c->br_S = S, c->br_np = 0;
			if (x) { TRY(loop_exchange_table(c, L, first_x ? Gate{nullptr, 0} : c->gate, gated ? c->loopctl : (int32_t *)nullptr, r)); c->br_n = L.ecap; first_x = false; }
		}
	}
	c->gate = Gate{nullptr, 0};
	c->arc_deferred = false, c->arc_done = false;
	int32_t *h_ctl = nullptr;
	if (gated) {
		if (!c->h_loopctl) c->h_loopctl = (int32_t *)c->pin.get(64);
		h_ctl = c->h_loopctl;
		if (!h_ctl) return PGA_ERR_NOMEM;
		HIPCHK(hipMemcpyAsync(h_ctl, c->loopctl, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, c->st));
	}
	const size_t fetch_need = (((size_t)S + 128 + (par->final_on ? 4 * sizeof(int32_t) * (size_t)S + 64 : 0) + 15) & ~(size_t)15) + 64 + 16;
	if (c->h_fetch_cap < fetch_need) {
		c->h_fetch = c->pin.get(fetch_need + fetch_need / 2 + 512);
		if (!c->h_fetch) return PGA_ERR_NOMEM;
		c->h_fetch_cap = fetch_need + fetch_need / 2 + 512;
	}
	HIPCHK(hipMemcpyAsync(c->h_fetch, alive, (size_t)S, hipMemcpyDeviceToHost, c->st));
	int64_t *h_need = (int64_t *)((char *)c->h_fetch + fetch_need - 16); // (the last 16 bytes of what was asked for)
	HIPCHK(hipMemcpyAsync(h_need, c->dcnt + 16, sizeof(int64_t), hipMemcpyDeviceToHost, c->st));
	int32_t *h_fin = (int32_t *)((char *)c->h_fetch + (((size_t)S + 127) & ~(size_t)63)); // (final_on) the last arc round's segment counters, the last branch step's n_dist_loci
	if (par->final_on) {
		HIPCHK(hipMemcpyAsync(h_fin, (const int32_t *)c->pool.get(S_SEGCNT, 0), sizeof(int32_t) * (size_t)n_vtx, hipMemcpyDeviceToHost, c->st));
		HIPCHK(hipMemcpyAsync(h_fin + n_vtx, ndl, sizeof(int32_t) * (size_t)n_vtx, hipMemcpyDeviceToHost, c->st));
	}
	int64_t *h_x = nullptr; // (sharded) behind the bytes, 8-byte aligned: the 4 collective flags (as int32) and the run's statistics
	if (x) {
		int32_t *flags4 = L.gbuf; // (the gather buffer is free again)
		hipLaunchKernelGGL(k_xs_flags, dim3(1), dim3(64), 0, c->st, c->dcnt, L.xstat, (long long)L.pair_cap, flags4);
		{ const int rc = x->allreduce_i32_sum(x->user, flags4, 4); if (rc) return rc; }
		h_x = (int64_t *)((char *)c->h_fetch + (((size_t)S + 7) & ~(size_t)7));
		HIPCHK(hipMemcpyAsync(h_x, flags4, 16, hipMemcpyDeviceToHost, c->st));
		HIPCHK(hipMemcpyAsync(h_x + 2, L.xstat, 4 * sizeof(int64_t), hipMemcpyDeviceToHost, c->st));
	}
	hipLaunchKernelGGL(k_mail_flush, dim3(1), dim3(64), 0, c->st, c->dcnt, c->h_box);
	TRY(sync_st(c));
	if (gated) { // the half-arc records carry the tag of the last arc round that RAN (a round that found nothing to do wrote none)
		c->round_tag = h_ctl[2] >= 0 ? (uint32_t)h_ctl[2] : tag_before;
		if (getenv("PANGENE_TIMING")) fprintf(stderr, "[pga_branch_loop] %d rounds queued; the last round that deleted a segment: %d, that marked a hit: %d (-1: none) -- the rounds after both found their kernels closed\n", n_round, h_ctl[0], h_ctl[1]);
	}
	c->br_np_seen = std::max<int64_t>(c->br_np_seen, c->h_cnt[15]);
	if (getenv("PANGENE_TIMING")) fprintf(stderr, "[pga_branch_loop] counters after %d rounds: invariant %lld, table overflows (last round) %lld, sticky %lld, pairs (last round) %lld of capacity %lld\n", n_round, (long long)c->h_cnt[3], (long long)c->h_cnt[9], (long long)c->h_cnt[11], (long long)c->h_cnt[15], (long long)c->br_cap);
	if (x && par->final_on) c->cur_tab = L.merged, c->cur_tab_n = c->h_cnt[10], c->table_sparse = false; // the merged table of the last queued arc round is the graph's (pga_arc_table)
	if (x) {
		const int32_t *f = (const int32_t *)h_x;
		c->x_pairs_run = std::max<int64_t>(c->x_pairs_run, h_x[2]), c->x_arcs_run = std::max<int64_t>(c->x_arcs_run, h_x[3]); // the next run's capacities (every round's list travels at its capacity: a margin above what was needed, not more)
		if (f[0] && f[2]) { // status 3: what the void run counted is worth little (it ran on empty tables from the overflow on) -- double what was too small, keep what was not
			c->x_pairs_seen = std::max<int64_t>(c->x_pairs_seen, c->x_pairs_run), c->x_arcs_seen = std::max<int64_t>(c->x_arcs_seen, c->x_arcs_run);
			c->x_pair_floor = std::max<int64_t>(c->x_pair_floor, h_x[2] > L.pair_cap ? std::max<int64_t>(2 * L.pair_cap, h_x[2] + h_x[2] / 4) : L.pair_cap);
			c->x_arc_floor = std::max<int64_t>(c->x_arc_floor, h_x[3] > L.arc_cap ? std::max<int64_t>(2 * L.arc_cap, h_x[3] + h_x[3] / 4) : L.arc_cap);
		}
		else if (!f[0]) c->x_pair_floor = 0, c->x_arc_floor = 0; // a run that went through: its statistics are the next run's capacities
		c->br_np_seen = std::max<int64_t>(c->br_np_seen, h_x[2]);
		c->br_cap = std::max<int64_t>(br_cap_before, c->br_cap);
		if (getenv("PANGENE_TIMING")) fprintf(stderr, "[pga_branch_loop] sharded over %d ranks: flags (summed) void %d invariant %d capacity %d hub %d; longest pair list %lld of %lld, largest local table %lld of %lld\n", x->world, f[0], f[1], f[2], f[3],
		                                          (long long)h_x[2], (long long)L.pair_cap, (long long)h_x[3], (long long)L.arc_cap);
		if (f[1]) return PGA_ERR_INVARIANT;
		if (f[0]) return (f[2] && !f[3]) ? 3 : 1;
	}
	else if (c->h_cnt[11]) {
		if (c->h_cnt[3]) return PGA_ERR_INVARIANT;
		// A pair list beyond its capacity and nothing else (no hub gene beyond its LDS table): the next attempt gets the room -- status 4, the caller runs the
		// queue again (round 6: a shard of 10 000 genomes lists 10^7 pairs in round 0, before pg_flt_high_occ has seen the graph; the queue gave
		// up for good and every pass of configs[3] at full size ran its fifteen rounds host-driven).  Lists beyond 2^26 pairs: status 1 as before.
		const int64_t need = *h_need;
		if (getenv("PANGENE_TIMING")) fprintf(stderr, "[pga_branch_loop] the longest pair list that did not fit: %lld (capacity %lld)\n", (long long)need, (long long)c->br_cap);
		if (need > c->br_cap && c->h_cnt[9] == 0 && need + need / 4 <= ((int64_t)1 << 27)) { c->br_cap = need + need / 4, c->loop_room_given = true; return 4; }
		return 1;
	}
	memcpy(seg_alive, c->h_fetch, (size_t)S);
	if (par->final_on) memcpy(seg_cnt_host, h_fin, sizeof(int32_t) * (size_t)n_vtx), memcpy(ndl_host, h_fin + n_vtx, sizeof(int32_t) * (size_t)n_vtx);
	return 0;
}

extern "C" int pga_mark_hits(pga_ctx_t *c, const uint64_t *arc_x, const uint8_t *arc_weak, int64_t n_arc, int64_t *n_marked, int32_t then_filter)
{
	const int N = c->N;
	if (n_marked) *n_marked = 0;
	if (N == 0) return 0;
	if (arc_x == nullptr) { // the arcs (and their weak_br) left resident by the round: every hit looks at its own two half-arcs (k_genes.hpp)
		TRY(ensure_half_arcs(c, c->ha_ori < 0 ? 0 : c->ha_ori));
		const uint64_t *ax = (const uint64_t *)c->pool.get(S_ARCX, 0); const uint8_t *aw = (const uint8_t *)c->pool.get(S_ARCW, 0);
		const int32_t *vs = (const int32_t *)c->pool.get(S_BR_VS, 0), *ve = (const int32_t *)c->pool.get(S_BR_VE, 0);
		const uint8_t *vwk = (const uint8_t *)c->pool.get(S_VWK, 0);
		if (!ax || !aw || !vs || !ve || !vwk) return PGA_ERR_NOMEM;
		if (n_marked) HIPCHK(hipMemsetAsync(c->dcnt + 2, 0, sizeof(int64_t), c->st));
		if (c->NL) hipLaunchKernelGGL(k_mark_hits_z, dim3(nblk(c->NL)), dim3(BLOCK), 0, c->st, c->zx, c->zy, c->zg, c->hfk, c->hbk, c->round_tag, c->NL, c->g2s, ax, aw, vs, ve, vwk, c->flags,
		                   n_marked ? c->dcnt + 2 : (int64_t *)nullptr, then_filter, n_marked ? Gate{nullptr, 0} : c->gate, c->gate.w ? c->loopctl : (int32_t *)nullptr, c->loop_round);
		if (then_filter) c->walk_valid = false, c->ha_valid = false; // else: weak_br does not enter the walkable test, the half-arcs stay valid
	} else {
		uint64_t *ax = (uint64_t *)c->pool.get(S_ARCX, sizeof(uint64_t) * (size_t)n_arc + 16);
		uint8_t *aw = (uint8_t *)c->pool.get(S_ARCW, (size_t)n_arc + 16);
		int32_t *wn = (int32_t *)c->pool.get(S_WEAKNEW, sizeof(int32_t) * (size_t)N);
		if (!ax || !aw || !wn) return PGA_ERR_NOMEM;
		TRY(upload(c, ax, arc_x, (size_t)n_arc)); TRY(upload(c, aw, arc_weak, (size_t)n_arc));
		zero_multi(c, wn, sizeof(int32_t) * (size_t)N, c->dcnt + 2, sizeof(int64_t));
		int32_t *val, *prev;
		TRY(walk_prev(c, &val, &prev));
		ensure_yrec(c);
		hipLaunchKernelGGL(k_mark_hits, dim3(nblk(N)), dim3(BLOCK), 0, c->st, val, prev, c->yrecA, c->yrecB, c->g2s, N, ax, aw, n_arc, (const int32_t *)nullptr, (const int32_t *)nullptr, (const uint8_t *)nullptr, wn);
		hipLaunchKernelGGL(k_weak_merge, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->flags, wn, N, n_marked ? c->dcnt + 2 : (int64_t *)nullptr);
		if (then_filter) TRY(pga_set_filter(c, PGA_FLT_WEAK2));
	}
	if (n_marked) {
		HIPCHK(hipMemcpyAsync(c->h_cnt, c->dcnt, 16 * sizeof(int64_t), hipMemcpyDeviceToHost, c->st));
		TRY(sync_st(c));
		*n_marked = c->h_cnt[2];
	}
	return 0;
}
