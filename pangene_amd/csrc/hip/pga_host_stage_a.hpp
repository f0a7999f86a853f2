// pga_host_stage_a.hpp -- pga_create / pga_begin / pga_ingest: upload, per-hit constants, stage A (read.c:243-260).
// Host side of the device ABI (include/pangene_hip.h); included by pga_backend.hip (one translation unit), in this order.
#pragma once
static bool stage_lookup(const char *p, size_t n, const char **dev, hipEvent_t *ev); // (pga_host_io.hpp)


static size_t pool_want(int64_t N, int64_t GL, int64_t P, int64_t Q, int64_t raw_words)
{
	const size_t per_hit = 568 /* measured: 530-540 B/hit at 1 M and 12 M hits (PANGENE_TIMING reports the fit at destroy) */, tables = (size_t)GL * ((size_t)P * 12 + (size_t)Q * 36) + (size_t)Q * 512 + (size_t)P * 64;
	return ((size_t)N * per_hit + tables + (64u << 20) + (size_t)raw_words * 4 + 255) & ~(size_t)255;
}

// the persistent arrays of a context (one allocation: dalloc_commit); also what pga_reserve sizes its first block by
static int plan_persistent(pga_ctx *c)
{
	const int N = c->N, E = c->E, GL = c->n_genome;
	TRY(dalloc(c, &c->dcnt, 24)); TRY(dalloc(c, &c->loopctl, 4));
	// persistent arrays
	TRY(dalloc(c, &c->fidx, N)); TRY(dalloc(c, &c->gnm, N)); TRY(dalloc(c, &c->seg, N)); TRY(dalloc(c, &c->pid, N)); TRY(dalloc(c, &c->gid, N));
	TRY(dalloc(c, &c->cs, N)); TRY(dalloc(c, &c->ce, N)); TRY(dalloc(c, &c->cm, N)); TRY(dalloc(c, &c->cds, N)); TRY(dalloc(c, &c->nex, N));
	TRY(dalloc(c, &c->offx, N)); TRY(dalloc(c, &c->sori, N)); TRY(dalloc(c, &c->sadj, N)); TRY(dalloc(c, &c->pm, N)); TRY(dalloc(c, &c->rk, N)); TRY(dalloc(c, &c->recA, N)); TRY(dalloc(c, &c->recB, N)); TRY(dalloc(c, &c->recC, N)); TRY(dalloc(c, &c->yrecA, N)); TRY(dalloc(c, &c->yrecB, N));
	TRY(dalloc(c, &c->rank, N)); TRY(dalloc(c, &c->sdom, N)); TRY(dalloc(c, &c->pdom, N)); TRY(dalloc(c, &c->pdom0, N)); TRY(dalloc(c, &c->flags, N));
	TRY(dalloc(c, &c->yperm, N)); TRY(dalloc(c, &c->goff, GL + 1)); TRY(dalloc(c, &c->ggl, GL)); TRY(dalloc(c, &c->ctg_base, GL + 1)); TRY(dalloc(c, &c->inv, N)); TRY(dalloc(c, &c->headpos, GL + 1)); TRY(dalloc(c, &c->exon, E));
	TRY(dalloc(c, &c->eoff, GL + 1)); TRY(dalloc(c, &c->woff, GL + 1));
	TRY(dalloc(c, &c->zx, N)); TRY(dalloc(c, &c->zy, N)); TRY(dalloc(c, &c->zg, N)); TRY(dalloc(c, &c->zst, N)); TRY(dalloc(c, &c->zpos, N)); TRY(dalloc(c, &c->wrec, 2 * (size_t)N)); TRY(dalloc(c, &c->zoff, (size_t)c->Q + 2));
	TRY(dalloc(c, &c->hfk, N)); TRY(dalloc(c, &c->hbk, N)); TRY(dalloc(c, &c->hfp, N)); TRY(dalloc(c, &c->hbp, N));
	TRY(dalloc(c, &c->lx, (size_t)N + 1)); TRY(dalloc(c, &c->ylist_buf, N)); TRY(dalloc(c, &c->live_cnt, LIVE_CNT_N)); TRY(dalloc(c, &c->tg, N));
	TRY(dalloc(c, &c->ga_ctl, 4));
	TRY(dalloc(c, &c->cA, N)); TRY(dalloc(c, &c->cB, N)); TRY(dalloc(c, &c->cC, N)); TRY(dalloc(c, &c->cx, N));
	TRY(dalloc(c, &c->prot_gid, c->P)); TRY(dalloc(c, &c->gene_pref, c->Q)); TRY(dalloc(c, &c->hrank, c->P));
	TRY(dalloc(c, &c->max_ori, c->P)); TRY(dalloc(c, &c->sums, 6 * (size_t)c->P)); TRY(dalloc(c, &c->vtx_cnt, 2 * (size_t)c->Q)); TRY(dalloc(c, &c->g2s, c->Q));
	return 0;
}

static int create_impl(pga_ctx *c, const pga_shard_t *sh)
{
	const int N = c->N, E = c->E, GL = c->n_genome;
	static const bool timing = getenv("PANGENE_TIMING") != nullptr;
	auto now = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; };
	const double t0 = now();
	HIPCHK(hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking));
	g_active_stream = c->st;
	{
		int dev = 0, ncu = 0;
		if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && ncu > 0) c->n_cu = ncu;
		g_last_dev.store(dev);
	}
	c->own_stream = true;
	c->h_cnt = (int64_t *)c->pin.get(24 * sizeof(int64_t));
	if (!c->h_cnt) return PGA_ERR_NOMEM;
	memset(c->h_cnt, 0, 24 * sizeof(int64_t));
	HIPCHK(hipHostGetDevicePointer((void **)&c->h_box, c->h_cnt, 0));
	TRY(plan_persistent(c));
	bool vsplit = false; // some genome arrives with virtual contigs (64-bit coordinates)
	int64_t n_vseg = 0;
	for (int g = 0; g < GL; ++g) {
		const pga_genome_block_t &b = sh->block[g];
		if (b.n_ctg < 0 || (b.vfirst == nullptr) != (b.vbase == nullptr)) return PGA_ERR_ARG;
		vsplit = vsplit || b.vfirst != nullptr, n_vseg += b.n_ctg;
	}
	if (n_vseg >= INT32_MAX) return PGA_ERR_RANGE;
	if (vsplit) { TRY(dalloc(c, &c->vfirst, (size_t)n_vseg + 1)); TRY(dalloc(c, &c->vbase, (size_t)n_vseg + 1)); }
	TRY(dalloc_commit(c));

	// host-side small tables (genome-sized)
	std::vector<int32_t> ctg_base((size_t)GL + 1, 0), eoff((size_t)GL + 1, 0);
	std::vector<int64_t> woff((size_t)GL + 1, 0);
	c->h_goff.assign((size_t)GL + 1, 0);
	c->rp_form = vsplit ? RP_WIDE : RP_COMPACT;
	std::vector<int32_t> h_vfirst; std::vector<int64_t> h_vbase;
	if (vsplit) h_vfirst.assign((size_t)n_vseg + 1, 0), h_vbase.assign((size_t)n_vseg + 1, 0);
	uint32_t max_cs = 0, max_cm = 0, max_sadj = 0;
	int32_t max_hit = 0, max_ctg = 1;
	bool neg_sadj = false, multi = false;
	for (int g = 0; g < GL; ++g) {
		const pga_genome_block_t &b = sh->block[g];
		if (b.n_hit < 0 || b.n_exon < 0 || b.n_ctg < 0 || b.n_words != (size_t)PGA_BLOCK_PLANES * b.n_hit + ((size_t)b.n_hit + 3) / 4 + 2 * (size_t)b.n_exon) return PGA_ERR_ARG;
		c->h_goff[(size_t)g + 1] = c->h_goff[(size_t)g] + b.n_hit, eoff[(size_t)g + 1] = eoff[(size_t)g] + b.n_exon;
		ctg_base[(size_t)g + 1] = ctg_base[(size_t)g] + b.n_ctg;
		if ((b.n_ctg >= 4096 || b.n_hit >= (1 << 20)) && c->rp_form == RP_COMPACT) c->rp_form = RP_FULL;
		if (vsplit) { // the shard-wide tables; a genome without its own: every contig is its own first piece, base 0
			const int32_t cb = ctg_base[(size_t)g];
			for (int32_t v = 0; v < b.n_ctg; ++v) {
				const int32_t f = b.vfirst ? b.vfirst[v] : v;
				const int64_t base = b.vbase ? b.vbase[v] : 0;
				// the pieces of a contig are consecutive and in coordinate order (pangene_hip.h): the (contig, cs) and (contig, cm) orders of
				// the pieces are then the orders of the contig
				if (f < 0 || f > v || base < 0 || (f != v && (b.vfirst[v - 1] != f || base < b.vbase[v - 1]))) return PGA_ERR_ARG;
				h_vfirst[(size_t)cb + (size_t)v] = cb + f, h_vbase[(size_t)cb + (size_t)v] = base;
			}
		}
		max_cs = std::max(max_cs, (uint32_t)b.max_cs), max_cm = std::max(max_cm, (uint32_t)b.max_cm), max_sadj = std::max(max_sadj, (uint32_t)b.max_score_adj);
		neg_sadj = neg_sadj || b.any_neg_score_adj, multi = multi || b.any_multi_exon;
		max_hit = std::max(max_hit, b.n_hit), max_ctg = std::max(max_ctg, b.n_ctg);
	}
	if (c->h_goff[(size_t)GL] != N || eoff[(size_t)GL] != E) return PGA_ERR_ARG;
	// The blocks of a shard usually lie side by side in a few slabs of host memory (the reader carves them out of page-locked slabs, 256
	// bytes apart at most): neighbours travel as ONE DMA.  One copy command per genome -- 0.5 MB each for a bacterial genome -- ran at
	// 24 GB/s on a link that does 56: the set-up of a command costs as much as its transfer.  A run's padding is copied along, so
	// the device image of a run mirrors its host addresses: woff[g] = where block g starts in the raw area.
	struct Run { const char *base; size_t bytes; int64_t dev_word; };
	std::vector<Run> runs;
	{
		std::vector<int32_t> by_addr;
		for (int g = 0; g < GL; ++g) if (sh->block[g].n_words) by_addr.push_back(g);
		std::sort(by_addr.begin(), by_addr.end(), [&](int32_t x, int32_t y) { return (uintptr_t)sh->block[x].data < (uintptr_t)sh->block[y].data; });
		int64_t dev_word = 0;
		for (int32_t g : by_addr) {
			const char *p = (const char *)sh->block[g].data;
			const size_t nb = sizeof(int32_t) * sh->block[g].n_words;
			if (!runs.empty() && p >= runs.back().base + runs.back().bytes && (size_t)(p - (runs.back().base + runs.back().bytes)) <= 1024 && (size_t)(p - runs.back().base) % 4 == 0) { // (a gap this small cannot hold an unmapped page)
				runs.back().bytes = (size_t)(p - runs.back().base) + nb;
			} else {
				if (!runs.empty()) dev_word += (int64_t)((runs.back().bytes + 255) / 256 * 64);
				runs.push_back(Run{p, nb, dev_word});
			}
			woff[(size_t)g] = runs.back().dev_word + (int64_t)((size_t)(p - runs.back().base) / 4);
		}
		if (!runs.empty()) dev_word += (int64_t)((runs.back().bytes + 255) / 256 * 64);
		woff[(size_t)GL] = dev_word; // the size of the raw area, in words
	}
	c->n_seg_ctg = ctg_base[(size_t)GL];
	c->h_ggl.assign(sh->genome_global, sh->genome_global + GL);
	c->cs_bits = bits_for(max_cs), c->cm_bits = bits_for(max_cm), c->seg_bits = bits_for((uint32_t)std::max(1, c->n_seg_ctg));
	c->sc_bits = neg_sadj ? 64 : std::min(64, 33 + bits_for(max_sadj)); // score key = score_adj << 33 | preferred << 32 | hash(pid)
	// pg_hash_uint32 is a bijection, so its rank among the P proteins orders them as the hash does: when score_adj, the preferred bit
	// and that rank fit 32 bits together, the hits' comparison keys need no sort of their own (a P-sized sort instead of an N-sized one)
	{
		const int pb = bits_for((uint32_t)std::max(1, c->P)), sb = bits_for(max_sadj);
		c->rk_shift = (!neg_sadj && sb + 1 + pb <= 32 && getenv("PANGENE_RANK_BY_SORT") == nullptr) ? pb + 1 : -1;
	}
	c->any_multi = multi, c->density_known = false, c->lists_in_lds = true;
	c->ctg_bits = bits_for((uint32_t)(max_ctg - 1));
	c->gs_np = std::max(64, (max_hit + 63) & ~63);
	c->gs_ok = c->gs_np <= GS_NP_MAX && getenv("PANGENE_GLOBAL_SORT") == nullptr; // (plane 15 holds a 32-bit comparison key either way: rk_shift < 0 = its dense rank, made below)
	c->gs2 = 0;
	if (c->gs_ok && c->gs_np <= GS2_NP_BIG) {
		c->gs2 = 1;
		std::vector<int32_t> small, big;
		int np_small = 64;
		for (int g = 0; g < GL; ++g) {
			const int nh = sh->block[g].n_hit;
			if (nh <= GS2_NP_MAX && c->gs2 == 1) small.push_back(g), np_small = std::max(np_small, (nh + 63) & ~63);
			else big.push_back(g);
		}
		// (a) a shard that cannot even fill the CUs once gains nothing from two workgroups per CU, and two half-empty launches in a row
		// cost more than one: everything by the 14-items form then; (b) the largest genomes first: the tail of a launch is then made
		// of the short ones
		if ((int)small.size() < 2 * c->n_cu) { big.insert(big.end(), small.begin(), small.end()); small.clear(); np_small = 64; }
		auto by_size = [&](int32_t x, int32_t y) { return sh->block[x].n_hit != sh->block[y].n_hit ? sh->block[x].n_hit > sh->block[y].n_hit : x < y; };
		std::sort(small.begin(), small.end(), by_size), std::sort(big.begin(), big.end(), by_size);
		c->gs2_n_small = (int)small.size(), c->gs2_n_big = (int)big.size(), c->gs2_np_small = np_small;
		small.insert(small.end(), big.begin(), big.end());
		if (c->gs2 && hipFuncSetAttribute(reinterpret_cast<const void *>(k_genome_sort2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gs2_lds_bytes(GS2_NP_MAX)) != hipSuccess) { (void)hipGetLastError(); c->gs2 = 0; }
		if (c->gs2 && hipFuncSetAttribute(reinterpret_cast<const void *>(k_genome_sort2d), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gs2_lds_bytes(GS2_NP_BIG)) != hipSuccess) { (void)hipGetLastError(); c->gs2 = 0; }
		if (c->gs2) {
			c->gs2_list = (int32_t *)c->pool.get(S_GS2LIST, sizeof(int32_t) * (size_t)std::max(1, GL));
			if (!c->gs2_list) return PGA_ERR_NOMEM;
			if (GL) HIPCHK(hipMemcpyAsync(c->gs2_list, small.data(), sizeof(int32_t) * (size_t)GL, hipMemcpyHostToDevice, c->st));
			HIPCHK(hipStreamSynchronize(c->st)); // (the list is a local)
		}
	}
	if (c->gs_ok) {
		const void *kf = c->gs_np <= GS_K_SMALL * GS_T ? reinterpret_cast<const void *>(k_genome_sort) : reinterpret_cast<const void *>(k_genome_sort_big);
		if (hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gs_lds_bytes(c->gs_np)) != hipSuccess) { (void)hipGetLastError(); c->gs_ok = false; }
	}

	c->gf_pos_bits = bits_for((uint32_t)std::max(1, max_hit - 1));
	c->gf_k32 = !neg_sadj && bits_for(max_sadj) + c->gf_pos_bits <= 32 && !env_has("PANGENE_FILTERS", "k64") && (gf_lds_bytes(c->P, c->Q) > (size_t)64 << 10 || env_has("PANGENE_FILTERS", "k32")); // (small tables: the 8-byte form, as before; tests force the other)
	c->gf_ok = gf_lds_bytes(c->P, c->Q, c->gf_k32) <= (size_t)150 << 10 && !env_has("PANGENE_FILTERS", "global");
	if (c->gf_ok && hipFuncSetAttribute(c->gf_k32 ? reinterpret_cast<const void *>(k_genome_filters<true>) : reinterpret_cast<const void *>(k_genome_filters<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
	                                    (int)gf_lds_bytes(c->P, c->Q, c->gf_k32)) != hipSuccess) { (void)hipGetLastError(); c->gf_ok = false; }

	{ // every temporary of a run comes out of one allocation: sorts and scans of 2N temp arcs, (genome x protein / gene) tables, ...
		const size_t want = pool_want(N, GL, c->P, c->Q, woff[(size_t)GL]) + ((max_hit > GS2_NP_BIG && max_ctg > 1) ? (size_t)N * 76 : 0); // (+ the planes once more, grouped by contig: contig bins)
		size_t got = 0;
		void *a = getenv("PANGENE_NO_ARENA") == nullptr ? dev_big_alloc(want, &got) : nullptr;
		if (a) { // else: slot by slot
			c->pool.arena = (char *)a, c->pool.arena_cap = got, c->pool.arena_off = 0;
			if (poison_on()) (void)hipMemset(a, 0x5a, got);
		}
	}
	const double t1 = now();
	// the blocks as they are (one DMA per genome out of pinned memory), then one kernel spreads them into flat file-order arrays
	int32_t *raw = (int32_t *)c->pool.get(S_RAW, sizeof(int32_t) * (size_t)woff[(size_t)GL] + 64);
	int32_t *up = (int32_t *)c->pool.get(S_UPLOAD, sizeof(int32_t) * (size_t)N * 18 + 64); // stays resident: begin() restarts a run without PCIe traffic
	if (!raw || !up) return PGA_ERR_NOMEM;
	size_t n_staged = 0, b_staged = 0;
	for (const Run &r : runs) { // (a run inside a slab the reader has staged already -- pga_stage_h2d -- comes out of that copy, behind its event)
		const char *dv = nullptr; hipEvent_t ev = nullptr;
		if (stage_lookup(r.base, r.bytes, &dv, &ev)) {
			HIPCHK(hipStreamWaitEvent(c->st, ev, 0));
			HIPCHK(hipMemcpyAsync(raw + r.dev_word, dv, r.bytes, hipMemcpyDeviceToDevice, c->st));
			++n_staged, b_staged += r.bytes;
		} else HIPCHK(hipMemcpyAsync(raw + r.dev_word, r.base, r.bytes, hipMemcpyHostToDevice, c->st));
	}
	TRY(upload(c, c->goff, c->h_goff.data(), (size_t)GL + 1)); TRY(upload(c, c->ggl, c->h_ggl.data(), GL));
	TRY(upload(c, c->ctg_base, ctg_base.data(), (size_t)GL + 1)); TRY(upload(c, c->eoff, eoff.data(), (size_t)GL + 1)); TRY(upload(c, c->woff, woff.data(), (size_t)GL + 1));
	TRY(upload(c, c->prot_gid, sh->prot_gid, c->P)); TRY(upload(c, c->gene_pref, sh->gene_pref, c->Q));
	if (vsplit) { TRY(upload(c, c->vfirst, h_vfirst.data(), (size_t)n_vseg + 1)); TRY(upload(c, c->vbase, h_vbase.data(), (size_t)n_vseg + 1)); } // (locals: the sync at the end of this function comes before they go)
	// half-arc records are validated by a round tag: none may survive from an earlier context whose memory this one inherited
	if (N) { HIPCHK(hipMemsetAsync(c->hfk, 0xff, sizeof(uint32_t) * (size_t)N, c->st)); HIPCHK(hipMemsetAsync(c->hbk, 0xff, sizeof(uint32_t) * (size_t)N, c->st)); }
	HIPCHK(hipMemsetAsync(c->dcnt, 0, 16 * sizeof(int64_t), c->st));
	if (N) hipLaunchKernelGGL(k_unblock, dim3(nblk(N)), dim3(BLOCK), 0, c->st, raw, c->woff, c->goff, c->eoff, GL, N, up, c->ctg_base, (int32_t)max_cs, (int32_t)max_cm, (int32_t)max_sadj, neg_sadj ? 1 : 0, multi ? 1 : 0, c->P, c->dcnt);
	if (E) hipLaunchKernelGGL(k_unblock_exons, dim3(nblk(E)), dim3(BLOCK), 0, c->st, raw, c->woff, c->goff, c->eoff, GL, E, c->exon);
	{ // work buffers shared by every sort / scan of the run: sized for the largest input (2N temp arcs)
		const int64_t W = std::max<int64_t>(2 * (int64_t)N + 2, (int64_t)c->P + 2);
		if (!c->pool.get(S_KEY_A, sizeof(uint64_t) * (size_t)W) || !c->pool.get(S_VAL_A, sizeof(uint32_t) * (size_t)W) ||
		    !c->pool.get(S_KEY_B, sizeof(uint64_t) * (size_t)W) || !c->pool.get(S_VAL_B, sizeof(uint32_t) * (size_t)W) ||
		    !c->pool.get(S_TABLE, sizeof(uint32_t) * (size_t)rs_table_len(W)) ||
		    !c->pool.get(S_TILE, tile_buf_bytes(W))) return PGA_ERR_NOMEM;
	}
	if (c->P && c->rk_shift >= 0) { // rank of hash(pid) over the proteins
		uint64_t *key = (uint64_t *)c->pool.get(S_KEY_A, 0); uint32_t *val = (uint32_t *)c->pool.get(S_VAL_A, 0);
		hipLaunchKernelGGL(k_hkey, dim3(nblk(c->P)), dim3(BLOCK), 0, c->st, c->P, key, val);
		RadixBufs b = { (uint64_t *)c->pool.get(S_KEY_B, 0), (uint32_t *)c->pool.get(S_VAL_B, 0), (uint32_t *)c->pool.get(S_TABLE, 0), (int32_t *)c->pool.get(S_TILE, 0) };
		uint64_t *ks; uint32_t *vs;
		device_radix_sort(key, val, c->P, 32, b, &ks, &vs, c->st);
		hipLaunchKernelGGL(k_hrank, dim3(nblk(c->P)), dim3(BLOCK), 0, c->st, ks, vs, c->P, c->hrank);
	}
	if (N) { // per-hit constants that depend on the input alone (gene, CDS length, score key, static flag bits): once per upload, file order
		FileHits f = { up, up + (size_t)N, up + 2 * (size_t)N, up + 3 * (size_t)N, up + 4 * (size_t)N, up + 5 * (size_t)N, up + 6 * (size_t)N, up + 7 * (size_t)N, up + 8 * (size_t)N,
		               up + 9 * (size_t)N, (const uint8_t *)(up + 14 * (size_t)N) };
		hipLaunchKernelGGL(k_prepare, dim3(nblk(N)), dim3(BLOCK), 0, c->st, f, N, c->goff, GL, c->ctg_base, c->exon, c->prot_gid, c->gene_pref,
		                   up + 10 * (size_t)N, up + 11 * (size_t)N, up + 12 * (size_t)N, up + 13 * (size_t)N, (uint64_t *)c->pool.get(S_KEY_A, 0), (uint32_t *)c->pool.get(S_VAL_A, 0),
		                   c->rk_shift, c->hrank, up + 15 * (size_t)N, up + 16 * (size_t)N, c->dcnt + 9);
		if (c->rk_shift < 0) {
			// The comparison key of overlap.c:137 (score_adj, preferred, hash(pid)) does not fit 32 bits here: the sweeps compare its dense RANK over the
			// shard instead (k_rank_scatter).  The key is made of what the upload brought -- nothing a pass changes --, so like the gene, the CDS length
			// and the static flag bits it is ranked once per upload (rounds 1-5 sorted the N 64-bit keys again in every pga_begin: 8 radix passes, 3.4 ms
			// of the 14.4 ms of stage A at the 21.9 M hits of configs[4]).
			int32_t *head = (int32_t *)c->pool.get(S_HEAD, sizeof(int32_t) * ((size_t)N + 1)), *incl = (int32_t *)c->pool.get(S_SLOT, sizeof(int32_t) * ((size_t)N + 1));
			I32 *tile = (I32 *)c->pool.get(S_TILE, tile_buf_bytes(N));
			if (!head || !incl || !tile) return PGA_ERR_NOMEM;
			uint64_t *ks; uint32_t *vs;
			TRY(radix_sort_pool(c, (uint64_t *)c->pool.get(S_KEY_A, 0), (uint32_t *)c->pool.get(S_VAL_A, 0), N, c->sc_bits, &ks, &vs));
			hipLaunchKernelGGL(k_arc_head, dim3(nblk(N)), dim3(BLOCK), 0, c->st, ks, (int64_t)N, head);
			device_scan<I32>(InI32{head}, OutInclI32{incl}, N, tile, OpSum{}, I32{0}, c->st);
			hipLaunchKernelGGL(k_rank_scatter, dim3(nblk(N)), dim3(BLOCK), 0, c->st, ks, vs, incl, N, up + 15 * (size_t)N);
		}
	}
	// CONTIG BINS (k_segsort.hpp).  A genome beyond what one workgroup's LDS sorts (14 336 hits) whose contigs all fit is sorted bin by bin -- a bin
	// = consecutive contigs of the genome, as many as fit.  For that the planes are grouped by contig here, once per upload: a stable sort of the
	// file-order indices by contig segment, one gather, and the contigs' sizes fetched for the host to cut the bins.  PANGENE_BINS=0: never;
	// PANGENE_BIN_CAP=n (tests): bins of at most n hits even where a genome would fit whole.
	c->bin_on = false;
	{
		static const int bins_env = [] { const char *e = getenv("PANGENE_BINS"); return e ? atoi(e) : -1; }();
		static const int cap_env = [] { const char *e = getenv("PANGENE_BIN_CAP"); return e ? atoi(e) : 0; }();
		const int cap_small = cap_env > 0 ? std::min(cap_env, GS2_NP_MAX) : GS2_NP_MAX, cap_big = cap_env > 0 ? cap_small : GS2_NP_BIG;
		if (N && bins_env != 0 && getenv("PANGENE_GLOBAL_SORT") == nullptr && (cap_env > 0 || (max_hit > GS2_NP_BIG && max_ctg > 1))) {
			uint64_t *key = (uint64_t *)c->pool.get(S_KEY_A, 0); uint32_t *val = (uint32_t *)c->pool.get(S_VAL_A, 0);
			const int n_sc = c->n_seg_ctg;
			int32_t *d_coff = (int32_t *)c->pool.get(S_BINS, sizeof(int4) * ((size_t)n_sc + (size_t)GL + 4)); // (first the contigs' offsets, then the bins: never more bins than contigs)
			if (!key || !val || !d_coff) return PGA_ERR_NOMEM;
			hipLaunchKernelGGL(k_zkey, dim3(nblk(N)), dim3(BLOCK), 0, c->st, (const int32_t *)(up + 11 * (size_t)N), N, key, val); // key = contig segment, val = file-order index
			uint64_t *ks; uint32_t *vs;
			TRY(radix_sort_pool(c, key, val, N, c->seg_bits, &ks, &vs));
			hipLaunchKernelGGL(k_zoff, dim3(nblk(n_sc + 1)), dim3(BLOCK), 0, c->st, (const uint64_t *)ks, N, n_sc, d_coff);
			std::vector<int32_t> coff((size_t)n_sc + 1);
			HIPCHK(hipMemcpyAsync(coff.data(), d_coff, sizeof(int32_t) * ((size_t)n_sc + 1), hipMemcpyDeviceToHost, c->st));
			TRY(sync_st(c));
			int32_t max_c = 0;
			for (int s = 0; s < n_sc; ++s) max_c = std::max(max_c, coff[(size_t)s + 1] - coff[(size_t)s]);
			if (max_c <= cap_big) {
				struct Bin { int32_t start, n, gz, c0, nc; };
				std::vector<Bin> small, big;
				int max_nc = 1, np_small = 64, np_big = 64;
				for (int g = 0; g < GL; ++g) {
					const int s0 = ctg_base[(size_t)g], s1 = ctg_base[(size_t)g + 1];
					bool first = true;
					for (int s = s0; s < s1;) { // as many consecutive contigs as fit a lean bin; a contig beyond that alone in a wide one
						int e = s, tot = 0;
						while (e < s1 && (tot + (coff[(size_t)e + 1] - coff[(size_t)e]) <= cap_small || e == s)) tot += coff[(size_t)e + 1] - coff[(size_t)e], ++e;
						if (tot > 0) {
							const Bin b = { coff[(size_t)s], tot, (int32_t)((uint32_t)g | (first ? 0x80000000u : 0u)), s - s0, e - s };
							(tot <= cap_small ? small : big).push_back(b);
							(tot <= cap_small ? np_small : np_big) = std::max(tot <= cap_small ? np_small : np_big, (tot + 63) & ~63);
							max_nc = std::max(max_nc, e - s), first = false;
						}
						s = e;
					}
				}
				auto by_size = [](const Bin &x, const Bin &y) { return x.n != y.n ? x.n > y.n : x.start < y.start; };
				std::sort(small.begin(), small.end(), by_size), std::sort(big.begin(), big.end(), by_size);
				bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(k_genome_sort2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gs2_lds_bytes(GS2_NP_MAX)) == hipSuccess &&
				          hipFuncSetAttribute(reinterpret_cast<const void *>(k_genome_sort2d), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gs2_lds_bytes(GS2_NP_BIG)) == hipSuccess;
				if (!ok) (void)hipGetLastError();
				int32_t *upc = ok ? (int32_t *)c->pool.get(S_UPLOAD2, sizeof(int32_t) * (size_t)N * 18 + 64) : nullptr;
				if (ok && upc) {
					hipLaunchKernelGGL(k_cgroup, dim3(nblk(N)), dim3(BLOCK), 0, c->st, (const int32_t *)up, (const uint32_t *)vs, (int64_t)N, (const int32_t *)c->goff, upc);
					std::vector<int4> hb;
					for (const Bin &b : small) hb.push_back(make_int4(b.start, b.n, b.gz, b.c0));
					for (const Bin &b : big) hb.push_back(make_int4(b.start, b.n, b.gz, b.c0));
					c->bins = (int4 *)d_coff; // (the offsets have been fetched)
					if (!hb.empty()) HIPCHK(hipMemcpyAsync(c->bins, hb.data(), sizeof(int4) * hb.size(), hipMemcpyHostToDevice, c->st));
					HIPCHK(hipStreamSynchronize(c->st)); // (hb is a local)
					c->bin_on = true, c->up_grouped = upc, c->bin_n_small = (int)small.size(), c->bin_n_big = (int)big.size(), c->bin_np_small = np_small, c->bin_np_big = np_big;
					c->bin_ctg_bits = bits_for((uint32_t)(max_nc - 1));
					if (timing) fprintf(stderr, "[pga_create] contig bins: %zu of up to %d hits + %zu of up to %d (largest contig %d hits, at most %d contigs a bin)\n", small.size(), np_small, big.size(), np_big, max_c, max_nc);
				}
			}
		}
	}
	HIPCHK(hipMemcpyAsync(c->h_cnt, c->dcnt, 16 * sizeof(int64_t), hipMemcpyDeviceToHost, c->st));
	int rc = sync_st(c); // the caller's blocks and tables have been read
	if (rc == 0 && c->h_cnt[8]) { // k_unblock: a hit outside the device layout, or beyond what its block declared (direct users of this ABI: the host driver checks while it packs)
		fprintf(stderr, "[E::pga_create] %lld hit(s) with coordinates, contig ids or scores outside what their genome block declares\n", (long long)c->h_cnt[8]);
		rc = PGA_ERR_RANGE;
	}
	c->exon_regular = rc != 0 || c->h_cnt[9] == 0; // (dcnt[9] is the rounds' overflow counter later on; pga_begin clears it)
	if (timing) fprintf(stderr, "[pga_create] allocations %.3f ms, upload of %.1f MB in %zu copy command(s) (%zu of them, %.1f MB, out of slabs staged while the files were read) + unpack %.3f ms\n", (t1 - t0) * 1e3, woff[(size_t)GL] * 4e-6, runs.size(), n_staged, b_staged * 1e-6, (now() - t1) * 1e3);
	return rc;
}

static int y_fixup_on() { static const int on = [] { const char *e = getenv("PANGENE_Y_FIXUP"); return (e && *e == '0') ? 0 : 1; }(); return on; } // (k_segsort2.hpp: the cm order out of the cs order by transpositions; 0 = by radix passes, as rounds 3-5)
// per-hit constants in file order, X order (sort + gather), running max of ce, Y order; resets all state
extern "C" int pga_begin(pga_ctx_t *c)
{
	c->yrec_valid = false, c->wrec_valid = false, c->z_valid = false, c->zposy_stale = false;
	c->tg_valid = false, c->z_early = false;
	c->live_on = false, c->NL = c->N, c->ylist = c->yperm, c->live_hint = -1; // (the flag words are written afresh: no F_MEMBER survives)
	const int N = c->N, GL = c->n_genome;
	c->walk_valid = false, c->ha_valid = false;
	if (c->x_arcs_run > 0) c->x_arcs_seen = c->x_arcs_run; // sharded rounds: what the run that just ended needed is what this one's exchange buffers hold
	if (c->x_pairs_run > 0) c->x_pairs_seen = c->x_pairs_run;
	c->x_arcs_run = 0, c->x_pairs_run = 0, c->x_redo = false;
	if (c->timing_on) { // class 3: the whole of stage A (sorts, per-hit constants, pg_flag_pseudo, sweeps, filters) = pga_begin + pga_ingest
		if (c->span_a) (void)hipEventDestroy(c->span_a);
		HIPCHK(hipEventCreate(&c->span_a));
		HIPCHK(hipEventRecord(c->span_a, c->st));
	}
	HIPCHK(hipMemsetAsync(c->dcnt, 0, 16 * sizeof(int64_t), c->st));
	if (c->Q) hipLaunchKernelGGL(k_fill_i32, dim3(nblk(c->Q)), dim3(BLOCK), 0, c->st, c->g2s, (int64_t)c->Q, -1);
	c->n_seg = 0;
	if (N == 0) return sync_st(c);
	int32_t *up = (int32_t *)c->pool.get(S_UPLOAD, 0);
	int32_t *f_pid = up, *f_cid = up + (size_t)N, *f_rank = up + 2 * (size_t)N, *f_sori = up + 3 * (size_t)N, *f_sadj = up + 4 * (size_t)N, *f_nex = up + 5 * (size_t)N,
		*f_offx = up + 6 * (size_t)N, *f_cs = up + 7 * (size_t)N, *f_ce = up + 8 * (size_t)N, *f_cm = up + 9 * (size_t)N,
		*f_gnm = up + 10 * (size_t)N, *f_seg = up + 11 * (size_t)N, *f_gid = up + 12 * (size_t)N, *f_cds = up + 13 * (size_t)N;
	uint8_t *f_rev = (uint8_t *)(up + 14 * (size_t)N);
	if (!up) return PGA_ERR_NOMEM;
	if (c->bin_on) { // contig bins (k_segsort.hpp): both orders, every per-hit constant, the packed records -- a workgroup per bin, its keys in LDS
		HitArrays o = { c->fidx, c->gnm, c->seg, c->pid, c->gid, c->cs, c->ce, c->cm, c->cds, c->nex, c->offx, c->sori, c->sadj, c->rank, c->sdom, c->pdom, c->pdom0, c->rk, c->flags };
		GenomeSort gs = { c->up_grouped, (int64_t)N, c->goff, c->ctg_base, c->cs_bits, c->cm_bits, c->bin_ctg_bits, c->bin_np_small, GL,
		                  o, c->yperm, c->headpos, c->recA, c->recB, c->recC, nullptr, nullptr, c->bins, y_fixup_on() };
		if (c->bin_n_small) hipLaunchKernelGGL(k_genome_sort2, dim3((unsigned)c->bin_n_small), dim3(GS2_T), gs2_lds_bytes(c->bin_np_small), c->st, gs);
		if (c->bin_n_big) { GenomeSort g2 = gs; g2.bins = c->bins + c->bin_n_small, g2.np = c->bin_np_big; hipLaunchKernelGGL(k_genome_sort2d, dim3((unsigned)c->bin_n_big), dim3(GS2_T), gs2_lds_bytes(c->bin_np_big), c->st, g2); }
		HIPCHK(hipMemcpyAsync(c->headpos, c->goff, sizeof(int32_t) * ((size_t)GL + 1), hipMemcpyDeviceToDevice, c->st));
		c->inv_valid = false;
		return 0;
	}
	if (c->gs_ok) { // one launch: both orders, every per-hit constant, the packed records (k_segsort.hpp)
		HitArrays o = { c->fidx, c->gnm, c->seg, c->pid, c->gid, c->cs, c->ce, c->cm, c->cds, c->nex, c->offx, c->sori, c->sadj, c->rank, c->sdom, c->pdom, c->pdom0, c->rk, c->flags };
		GenomeSort gs = { up, (int64_t)N, c->goff, c->ctg_base, c->cs_bits, c->cm_bits, c->ctg_bits, c->gs_np, GL,
		                  o, c->yperm, c->headpos, c->recA, c->recB, c->recC, nullptr, nullptr, nullptr, y_fixup_on() };
		static const bool gs_prof = getenv("PANGENE_GS_PROF") != nullptr;
		if (gs_prof) { gs.prof = (long long *)c->pool.get(S_SCRATCH, sizeof(long long) * 32 * (size_t)GL); if (gs.prof) HIPCHK(hipMemsetAsync(gs.prof, 0, sizeof(long long) * 32 * (size_t)GL, c->st)); }
		if (c->gs2 && !gs_prof) {
			if (c->gs2_n_small) { GenomeSort g1 = gs; g1.glist = c->gs2_list, g1.np = c->gs2_np_small; hipLaunchKernelGGL(k_genome_sort2, dim3((unsigned)c->gs2_n_small), dim3(GS2_T), gs2_lds_bytes(c->gs2_np_small), c->st, g1); }
			if (c->gs2_n_big) { GenomeSort g2 = gs; g2.glist = c->gs2_list + c->gs2_n_small; hipLaunchKernelGGL(k_genome_sort2d, dim3((unsigned)c->gs2_n_big), dim3(GS2_T), gs2_lds_bytes(c->gs_np), c->st, g2); }
		}
		else if (c->gs_np <= GS_K_SMALL * GS_T) hipLaunchKernelGGL(k_genome_sort, dim3((unsigned)GL), dim3(GS_T), gs_lds_bytes(c->gs_np), c->st, gs);
		else hipLaunchKernelGGL(k_genome_sort_big, dim3((unsigned)GL), dim3(GS_T), gs_lds_bytes(c->gs_np), c->st, gs);
		c->inv_valid = false;
		if (gs.prof) { // mean cycles per phase over the workgroups (100 MHz constant counter: 10 ns per tick)
			std::vector<long long> hp((size_t)32 * GL);
			HIPCHK(hipStreamSynchronize(c->st));
			HIPCHK(hipMemcpy(hp.data(), gs.prof, hp.size() * sizeof(long long), hipMemcpyDeviceToHost));
			double d[20] = { 0 }; long long t_min = INT64_MAX, t_max = 0;
			for (int g2 = 0; g2 < GL; ++g2) { for (int k = 1; k <= 12; ++k) d[k] += (double)(hp[(size_t)g2 * 32 + k] - hp[(size_t)g2 * 32 + k - 1]); for (int k = 17; k <= 21; ++k) d[k - 4] += (double)(hp[(size_t)g2 * 32 + k] - hp[(size_t)g2 * 32 + k - 1]); d[0] += (double)(hp[(size_t)g2 * 32 + 16] - hp[(size_t)g2 * 32]); t_min = std::min(t_min, hp[(size_t)g2 * 32]), t_max = std::max(t_max, hp[(size_t)g2 * 32 + 12]); }
			fprintf(stderr, "[k_genome_sort profile, np %d, ticks/workgroup]", c->gs_np);
			for (int k = 1; k <= 12; ++k) fprintf(stderr, " %d:%.0f", k, d[k] / GL);
			fprintf(stderr, " | first radix pass: until the byte plane is staged %.0f, histogram %.0f (wave 0) + %.0f (barrier), scan %.0f, scatter %.0f (wave 0) + %.0f (barrier)", d[0] / GL, d[13] / GL, d[14] / GL, d[15] / GL, d[16] / GL, d[17] / GL);
			fprintf(stderr, " | kernel span %lld ticks\n", t_max - t_min);
		}
		return 0;
	}
	// (the per-hit constants -- genome, segment, gene, CDS length, static flag bits and, when it fits 32 bits, the comparison key -- were
	// computed once, at the upload: create_impl's k_prepare.  Round 4 ran it again every pass here: 1.2 ms of exon-list walks at 21.9 M hits.)
	int32_t *rk_f = up + 15 * (size_t)N; // the comparison key in 32 bits, or its dense rank over the shard: either way made once per upload (create_impl)
	uint64_t *key = (uint64_t *)c->pool.get(S_KEY_A, 0);
	uint32_t *val = (uint32_t *)c->pool.get(S_VAL_A, 0);
	if (!up || !key || !val) return PGA_ERR_NOMEM;
	FileHits f = { f_pid, f_cid, f_rank, f_sori, f_sadj, f_nex, f_offx, f_cs, f_ce, f_cm, f_rev };
	uint64_t *ks; uint32_t *vs;
	// X order: pg_hit_sort(g, 0), hit.c:29-64, for every genome at once; stable => ties keep file order
	key = (uint64_t *)c->pool.get(S_KEY_A, 0), val = (uint32_t *)c->pool.get(S_VAL_A, 0);
	hipLaunchKernelGGL(k_xkey, dim3(nblk(N)), dim3(BLOCK), 0, c->st, f_seg, f_cs, N, c->cs_bits, key, val);
	TRY(radix_sort_pool(c, key, val, N, c->cs_bits + c->seg_bits, &ks, &vs));
	HitArrays o = { c->fidx, c->gnm, c->seg, c->pid, c->gid, c->cs, c->ce, c->cm, c->cds, c->nex, c->offx, c->sori, c->sadj, c->rank, c->sdom, c->pdom, c->pdom0, c->rk, c->flags };
	hipLaunchKernelGGL(k_gather, dim3(nblk(N)), dim3(BLOCK), 0, c->st, f, f_gnm, f_seg, f_gid, f_cds, rk_f, vs, N, c->goff, o);
	hipLaunchKernelGGL(k_inv_only, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->fidx, c->gnm, c->goff, N, c->inv);
	c->inv_valid = true;
	HIPCHK(hipMemcpyAsync(c->headpos, c->goff, sizeof(int32_t) * ((size_t)GL + 1), hipMemcpyDeviceToDevice, c->st));
	// running max of ce per contig
	SegMax *tile = (SegMax *)c->pool.get(S_TILE, tile_buf_bytes(N));
	device_scan<SegMax>(InSegMax{c->seg, c->ce}, OutSegMax{c->pm}, N, tile, OpSegMax{}, SegMax{SEG_EMPTY, 0}, c->st);
	pack_records(c);
	// Y order: pg_hit_sort(g, 1); ties keep X order
	key = (uint64_t *)c->pool.get(S_KEY_A, 0), val = (uint32_t *)c->pool.get(S_VAL_A, 0);
	hipLaunchKernelGGL(k_ykey, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->seg, c->cm, N, c->cm_bits, key, val);
	TRY(radix_sort_pool(c, key, val, N, c->cm_bits + c->seg_bits, &ks, &vs));
	HIPCHK(hipMemcpyAsync(c->yperm, vs, sizeof(int32_t) * (size_t)N, hipMemcpyDeviceToDevice, c->st));
	return 0;
}

extern "C" int pga_create(pga_ctx_t **out, const pga_shard_t *sh, const pga_params_t *par)
{
	if (out == nullptr || sh == nullptr || par == nullptr) return PGA_ERR_ARG;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
		fprintf(stderr, "[E::pga_create] no HIP device is visible; libpangene_amd has no CPU fallback\n");
		return PGA_ERR_NO_DEVICE;
	}
	if (sh->abi_version != PGA_ABI_VERSION) { fprintf(stderr, "[E::pga_create] the caller was built against ABI version %u of pangene_hip.h, this library implements %u\n", sh->abi_version, (unsigned)PGA_ABI_VERSION); return PGA_ERR_ARG; }
	if (sh->n_hit >= (1 << 30) /* arc table positions are 2 * (gene-major index) in 32 bits */ || sh->n_exon >= INT32_MAX || sh->n_gene >= (1 << 20) || sh->n_genome_global >= (1 << 24)) return PGA_ERR_RANGE;
	pga_ctx *c = new pga_ctx();
	c->n_genome = sh->n_genome, c->n_genome_global = sh->n_genome_global, c->P = sh->n_prot, c->Q = sh->n_gene;
	c->N = (int32_t)sh->n_hit, c->E = (int32_t)sh->n_exon, c->par = *par;
	if (sh->n_genome > 0 && sh->block == nullptr) { delete c; return PGA_ERR_ARG; }
	int rc = create_impl(c, sh);
	if (rc) { pga_destroy(c); return rc; }
	*out = c;
	return 0;
}

// stage A (read.c:243-260) for all genomes of the shard
extern "C" int pga_ingest(pga_ctx_t *c, int32_t *stats)
{
	c->yrec_valid = false, c->wrec_valid = false;
	const int N = c->N, GL = c->n_genome, P = c->P, Q = c->Q;
	int32_t *d_stats = (int32_t *)c->pool.get(S_STATS, sizeof(int32_t) * 4 * (size_t)GL + 16);
	if (!d_stats) return PGA_ERR_NOMEM;
	HIPCHK(hipMemsetAsync(d_stats, 0, sizeof(int32_t) * 4 * (size_t)GL + 16, c->st));
	// the per-genome counts only feed a log line: the four-kernel form of the filters counts with one global atomic per filtered hit
	// (17 M of them onto 200 addresses on the full-size configs[4] set: 33 ms), so they are only kept when somebody asked for them
	int32_t *k_stats = stats ? d_stats : nullptr;
	if (N) {
		const int64_t TP = (int64_t)GL * P, TQ = (int64_t)GL * Q;
		if (c->any_multi) { // pg_flag_pseudo (hit.c:66-105) only ever marks a protein that has a multi-exon hit (max_n > 1, hit.c:84)
			int32_t *tmax = (int32_t *)c->pool.get(S_TAB_A, sizeof(int32_t) * (size_t)TP);
			int32_t *tmin = (int32_t *)c->pool.get(S_TAB_B, sizeof(int32_t) * (size_t)TP);
			int32_t *tr1 = (int32_t *)c->pool.get(S_TAB_C, sizeof(int32_t) * (size_t)TP);
			uint32_t *pbits = (uint32_t *)c->pool.get(S_TAB_D, sizeof(uint32_t) * (size_t)((TP + 31) / 32) + 64); // (S_TAB_D: the filters' table of the four-kernel form, which comes later)
			if (!tmax || !tmin || !tr1 || !pbits) return PGA_ERR_NOMEM;
			HIPCHK(hipMemsetAsync(pbits, 0, sizeof(uint32_t) * (size_t)((TP + 31) / 32), c->st));
			hipLaunchKernelGGL(k_pseudo0, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->gnm, c->pid, c->rank, N, P, pbits, tmax, tmin, tr1); // the cells that matter, initialised by the hits that name them
			hipLaunchKernelGGL(k_pseudo1, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->gnm, c->pid, c->nex, N, P, (const uint32_t *)pbits, tmax, tmin);
			hipLaunchKernelGGL(k_pseudo2, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->gnm, c->pid, c->nex, c->rank, c->flags, N, P, (const uint32_t *)pbits, tmax, tmin, tr1, k_stats);
			hipLaunchKernelGGL(k_pseudo3, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->gnm, c->pid, c->rank, N, P, (const uint32_t *)pbits, tmax, tmin, tr1, c->recC); // (the ranks that change are patched in record C too)
		}
		unsigned long long *tbest = c->gf_ok ? nullptr : (unsigned long long *)c->pool.get(S_TAB_D, sizeof(uint64_t) * (size_t)TQ);
		uint8_t *noiso = c->gf_ok ? nullptr : (uint8_t *)c->pool.get(S_TAB_A, (size_t)TP + 16); // byte (genome, protein): the protein has a hit there without flt_iso_ov
		if (!c->gf_ok && (!tbest || !noiso)) return PGA_ERR_NOMEM;
		// read.c:248-254: ONE sweep for pg_shadow(cal_dom_sc=1), the reset behind it and pg_flt_ov_isoform (k_sweep<3>: they walk the same pairs)
		if (c->any_multi && !c->density_known) TRY(measure_list_density(c));
		c->sweep_init = true;
		const int rc_sw = launch_sweep<3>(c, 0); // "K1", the hit-filter+overlap kernel
		c->sweep_init = false;
		TRY(rc_sw);
		if (c->gf_ok) { // read.c:249-256 per genome, the tables in LDS
			GenomeFilters gf = { c->flags, c->pid, c->gid, c->rank, c->sadj, c->pdom, c->pdom0, c->goff, c->recA, P, Q, d_stats, c->dcnt, (int32_t *)c->pool.get(S_HZLIST, sizeof(int32_t) * PGA_HAZARD_CAP), c->gf_pos_bits };
			if (!gf.hz_list) return PGA_ERR_NOMEM;
			if (c->gf_k32) hipLaunchKernelGGL(k_genome_filters<true>, dim3((unsigned)GL), dim3(GF_T), gf_lds_bytes(P, Q, true), c->st, gf);
			else hipLaunchKernelGGL(k_genome_filters<false>, dim3((unsigned)GL), dim3(GF_T), gf_lds_bytes(P, Q, false), c->st, gf);
		} else {
		HIPCHK(hipMemsetAsync(noiso, 0, (size_t)TP, c->st));
		hipLaunchKernelGGL(k_iso_apply, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->flags, c->gnm, c->pid, c->pdom, c->pdom0, N, P, noiso, k_stats);
		hipLaunchKernelGGL(k_chain, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->flags, c->gnm, c->pdom0, N, P, noiso, k_stats);
		HIPCHK(hipMemsetAsync(tbest, 0, sizeof(uint64_t) * (size_t)TQ, c->st));
		hipLaunchKernelGGL(k_subopt1, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->flags, c->gnm, c->gid, c->rank, c->sadj, c->goff, N, Q, tbest);
		hipLaunchKernelGGL(k_subopt2, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->flags, c->gnm, c->gid, c->pid, c->goff, N, Q, tbest, k_stats, c->rank, c->sadj, c->recA, c->dcnt,
		                   (int32_t *)c->pool.get(S_HZLIST, sizeof(int32_t) * PGA_HAZARD_CAP));
		}
	}
	if (c->timing_on && c->span_a) {
		TimedLaunch t; t.which = 3, t.units = N, t.a = c->span_a, c->span_a = nullptr;
		HIPCHK(hipEventCreate(&t.b));
		HIPCHK(hipEventRecord(t.b, c->st));
		c->timed.push_back(t);
	}
	if (stats) {
		HIPCHK(hipMemcpyAsync(stats, d_stats, sizeof(int32_t) * 4 * (size_t)GL, hipMemcpyDeviceToHost, c->st));
		return sync_st(c);
	}
	return 0;
}
