// pga_host_context.hpp -- the big device blocks and their cache, the pools, the context (pga_ctx) and its one-allocation plan.
// Host side of the device ABI (include/pangene_hip.h); included by pga_backend.hip (one translation unit), in this order.
#pragma once


// the shared scan / sort work buffer: tile sums of a scan over n items (8 bytes each) or the 256 digit totals of a radix pass
static inline size_t tile_buf_bytes(int64_t n) { return std::max<size_t>(sizeof(int64_t) * (size_t)(scan_tiles(std::max<int64_t>(rs_table_len(n), n)) + 8), 256 * sizeof(uint32_t) + 64); }

static inline unsigned nblk(int64_t n, int per = BLOCK) { return (unsigned)((n + per - 1) / per); }

// ------------------------------------------------------------------------------------------------
// The two big device allocations of a context (the arena of the persistent arrays, the arena of the temporaries) outlive it in a
// small process-wide cache: hipMalloc / hipFree of gigabytes take anything from 0.4 to 350 ms on this pool's boxes, which made the
// upload-inclusive pass of the SAME shard range from 10 to 47 ms.  A process that runs one data set after another (a service, the
// bench's cold passes) pays for the memory once.  Bounded: two blocks are kept (one context's worth); a block is reused for a
// request it fits without wasting more than half of it.  pga_host_trim(0) (pg_trim_host_cache) gives them back.
// ------------------------------------------------------------------------------------------------
struct DevBlock { void *p; size_t cap; int dev; };
static std::atomic<int> g_last_dev{-1}; // the device of the last context (or pga_set_device): where a pga_reserve on another thread allocates
static int cur_dev() { int d = 0; return hipGetDevice(&d) == hipSuccess ? d : 0; }
static std::mutex g_dev_mu;
static std::vector<DevBlock> g_dev_cache;
static std::condition_variable g_dev_cv; static int g_dev_reserving = 0; // pga_reserve calls under way: whoever wants a big block waits for them first (the block is probably theirs)
static bool dev_cache_on() { static const bool on = [] { const char *e = getenv("PANGENE_DEV_CACHE"); return !(e && *e == '0'); }(); return on; }

static void *dev_big_alloc(size_t want, size_t *got)
{
	{
		std::unique_lock<std::mutex> lk(g_dev_mu);
		g_dev_cv.wait(lk, [] { return g_dev_reserving == 0; });
		size_t best = (size_t)-1;
		const int dev = cur_dev();
		for (size_t i = 0; i < g_dev_cache.size(); ++i)
			if (g_dev_cache[i].dev == dev && g_dev_cache[i].cap >= want && g_dev_cache[i].cap <= 2 * want + ((size_t)64 << 20) && (best == (size_t)-1 || g_dev_cache[i].cap < g_dev_cache[best].cap)) best = i;
		if (best != (size_t)-1) {
			DevBlock b = g_dev_cache[best];
			g_dev_cache.erase(g_dev_cache.begin() + (long)best);
			*got = b.cap;
			return b.p;
		}
	}
	void *q = nullptr;
	// A block that will be kept is asked for with room to spare (an eighth, to the next 64 MiB): the next data set of a series is a few
	// per cent larger or smaller than this one, and a block that is a megabyte short means hipFree + hipMalloc -- 15 ms of a 9 ms pass
	// (two of five data sets of a bench run showed it).
	if (dev_cache_on() && want >= ((size_t)1 << 20)) {
		const size_t padded = (want + want / 8 + ((size_t)64 << 20) - 1) & ~(((size_t)64 << 20) - 1);
		if (hipMalloc(&q, padded) == hipSuccess) { *got = padded; return q; }
		(void)hipGetLastError(), q = nullptr;
	}
	if (hipMalloc(&q, want) != hipSuccess) {
		(void)hipGetLastError();
		{ // the cache may be what stands in the way
			std::lock_guard<std::mutex> lk(g_dev_mu);
			for (DevBlock &b : g_dev_cache) (void)hipFree(b.p);
			g_dev_cache.clear();
		}
		if (hipMalloc(&q, want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
	}
	*got = want;
	return q;
}

static void dev_big_free(void *p, size_t cap)
{
	if (p == nullptr) return;
	if (dev_cache_on() && cap >= ((size_t)1 << 20)) {
		std::lock_guard<std::mutex> lk(g_dev_mu);
		if (g_dev_cache.size() >= 2) { // keep the two largest
			size_t small = 0;
			for (size_t i = 1; i < g_dev_cache.size(); ++i) if (g_dev_cache[i].cap < g_dev_cache[small].cap) small = i;
			if (g_dev_cache[small].cap >= cap) { (void)hipFree(p); return; }
			(void)hipFree(g_dev_cache[small].p);
			g_dev_cache.erase(g_dev_cache.begin() + (long)small);
		}
		g_dev_cache.push_back(DevBlock{p, cap, cur_dev()});
		return;
	}
	(void)hipFree(p);
}

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct DevPool { // persistent, grow-only device temporaries keyed by slot
	std::vector<void *> p; std::vector<size_t> cap; std::vector<char> own; // own: the slot has a hipMalloc of its own
	// One big allocation made in create() from which the slots are carved (a bump allocator: a slot that outgrows its piece
	// takes a new one): the first pass of a run then needs two hipMalloc calls instead of ~70.
	char *arena = nullptr; size_t arena_cap = 0, arena_off = 0;
	void *get(int slot, size_t bytes)
	{
		if ((int)p.size() <= slot) p.resize(slot + 1, nullptr), cap.resize(slot + 1, 0), own.resize(slot + 1, 0);
		if (bytes == 0) bytes = 16;
		if (cap[slot] < bytes) {
			if (p[slot] && own[slot]) (void)hipFree(p[slot]);
			size_t want = (bytes + bytes / 4 + 256 + 255) & ~(size_t)255;
			if (arena && arena_off + want <= arena_cap) { p[slot] = arena + arena_off, arena_off += want, own[slot] = 0; }
			else if (hipMalloc(&p[slot], want) == hipSuccess) { own[slot] = 1; if (poison_on()) (void)hipMemset(p[slot], 0x5a, want); }
			else { p[slot] = nullptr; cap[slot] = 0; own[slot] = 0; return nullptr; }
			cap[slot] = want;
		}
		return p[slot];
	}
	void release()
	{
		for (size_t i = 0; i < p.size(); ++i) if (p[i] && own[i]) (void)hipFree(p[i]);
		dev_big_free(arena, arena_cap);
		p.clear(); cap.clear(); own.clear(); arena = nullptr; arena_cap = arena_off = 0;
	}
};

// Small pinned host buffers (mailboxes, staging areas, per-round results): carved out of a few pinned blocks that outlive the
// context in a process-wide cache -- hipHostMalloc costs milliseconds and would otherwise be paid several times in the first pass
// over every data set.
struct PinBlock { char *p; size_t cap; };
static std::mutex g_pin_mu;
static std::vector<PinBlock> g_pin_cache;
struct PinArena {
	std::vector<PinBlock> blocks; size_t off = 0;
	void *get(size_t bytes)
	{
		bytes = (bytes + 255) & ~(size_t)255;
		if (blocks.empty() || off + bytes > blocks.back().cap) {
			PinBlock b = { nullptr, 0 };
			{
				std::lock_guard<std::mutex> lk(g_pin_mu);
				for (size_t i = 0; i < g_pin_cache.size(); ++i)
					if (g_pin_cache[i].cap >= bytes) { b = g_pin_cache[i]; g_pin_cache.erase(g_pin_cache.begin() + (long)i); break; }
			}
			if (b.p == nullptr) {
				b.cap = std::max<size_t>(bytes, (size_t)8 << 20);
				if (hipHostMalloc((void **)&b.p, b.cap, hipHostMallocDefault) != hipSuccess) return nullptr;
			}
			blocks.push_back(b), off = 0;
		}
		void *r = blocks.back().p + off;
		off += bytes;
		if (poison_on()) memset(r, 0x5a, bytes);
		return r;
	}
	void release() // back to the cache (a handful of blocks per process)
	{
		std::lock_guard<std::mutex> lk(g_pin_mu);
		for (PinBlock &b : blocks) { if (g_pin_cache.size() < 16) g_pin_cache.push_back(b); else (void)hipHostFree(b.p); }
		blocks.clear(), off = 0;
	}
};

enum { // pool slots
	S_KEY_A, S_KEY_B, S_VAL_A, S_VAL_B, S_TABLE, S_TILE, S_I32_A, S_I32_B, S_I32_C, S_TAB_A, S_TAB_B, S_TAB_C, S_TAB_D,
	S_TDIST, S_TS1, S_TS2, S_TGEN, S_SDIST, S_SS1, S_SS2, S_SGEN, S_HEAD, S_SLOT, S_ARCS, S_SEGCNT, S_BITS, S_TRIPLES,
	S_WALK_VAL, S_WALK_PREV, S_PERM, S_OVPOS, S_OVFILE, S_RUNSTART, S_CDN, S_MG_KEY, S_MG_VAL, S_MG_SRC, S_MG_OUT, S_MG_HEAD, S_MG_SLOT, S_MG_RUN, S_BR_S1, S_BR_GID, S_BR_VS, S_BR_VE, S_BR_PC, S_BR_POFF, S_BR_GRP, S_BR_NDL, S_BR_SEGGID, S_PAIRS, S_NLCNT, S_ARCX, S_ARCW, S_WEAKNEW, S_RP_SEG, S_RP_R, S_RP_CM, S_RP_POS, S_RP_IV, S_DL, S_SCRATCH, S_UPLOAD, S_RAW, S_ARC_STAGE, S_GMETA, S_GOFF, S_DEG, S_BIGLIST, S_STAGE_SID, S_STATS, S_G2S, S_MISC, S_SLOW, S_HZLIST, S_VWK, S_XG_BUF, S_XG_OUT, S_XG_OUT2, S_XSTAT, S_GS2LIST, S_UPLOAD2, S_BINS,
	S_COUNT
};

constexpr int LIVE_CNT_N = 1024;
struct TimedLaunch { hipEvent_t a, b; int which; int64_t units; };

struct pga_ctx {
	hipStream_t st = nullptr; bool own_stream = false;
	int32_t n_genome = 0, n_genome_global = 0, P = 0, Q = 0, n_seg_ctg = 0;
	int32_t N = 0, E = 0;
	int n_cu = 256;
	uint32_t sweep_seq = 0; // parity selects the slow-list counter (dcnt[12] / dcnt[13])
	pga_params_t par;
	std::vector<int32_t> h_goff, h_ggl;
	// static per hit (X order)
	int32_t *fidx = 0, *gnm = 0, *seg = 0, *pid = 0, *gid = 0, *cs = 0, *ce = 0, *cm = 0, *cds = 0, *nex = 0, *offx = 0, *sori = 0, *sadj = 0, *pm = 0;
	int32_t *rk = 0;        // dense rank of the score key (score_adj, preferred, hash(pid)) of overlap.c:137 over the shard; 0 = key 0
	int sc_bits = 64;       // significant bits of that key
	int rk_shift = -1;      // >= 0: the key fits 32 bits as score_adj << rk_shift | preferred << (rk_shift - 1) | (rank of hash(pid) among the proteins): no sort
	int32_t *hrank = 0;     // [P] rank of hash(pid) + 1 (0 for a hash of 0)
	bool any_multi = true;  // some hit has more than one exon
	bool exon_regular = true; // every exon list is sorted and disjoint (k_prepare): the sweeps may take the shortcuts of cds_inter_t
	double list_density = 0; int64_t density_tiles = 0; // (what k_list_density found: exons a tile, tiles sampled)
	bool lists_in_lds = true, density_known = false; // K1 of stage A / pg_post_process stages the exon lists (k_sweep) or leaves them where they are (k_sweep_lean): by k_list_density, once per upload
	int rp_form = 0;         // form of the (gene, genome) position records (see k_rep_fill)
	int32_t *vfirst = 0; int64_t *vbase = 0; // virtual contigs (pga_genome_block_t), per contig segment of the shard: segment of the contig's first piece, the piece's base; NULL = no genome has any
	int4 *recA = 0, *recB = 0, *recC = 0; // packed sweep records (derived from the arrays above, see k_pack_rec)
	// dynamic per hit
	int32_t *rank = 0, *sdom = 0, *pdom = 0, *pdom0 = 0; uint32_t *flags = 0;
	int32_t *yperm = 0, *goff = 0, *ggl = 0, *ctg_base = 0, *inv = 0, *headpos = 0, *eoff = 0; int64_t *woff = 0;
	int cs_bits = 1, cm_bits = 1, seg_bits = 1, ctg_bits = 1;
	bool inv_valid = false;  // inv[] (file index -> X position) matches the current order: built on demand (pga_set_head)
	bool sweep_init = false; // the next pg_shadow(cal_dom_sc=1) also initialises pid_dom / score_dom of the filtered hits (pga_ingest)
	int gs2 = 0; // stage A's orders by the kernels of k_segsort2.hpp: genomes of up to 10 240 hits by k_genome_sort2 (two workgroups per CU), the others (up to 14 336) by k_genome_sort2d
	int32_t *gs2_list = nullptr; int gs2_n_small = 0, gs2_n_big = 0, gs2_np_small = 64; // the two lists of genomes: [small..., big...]
	// contig bins (k_segsort.hpp: GenomeSort::bins): genomes too large for one workgroup's LDS, every contig of which fits, are sorted bin by bin
	bool bin_on = false; int32_t *up_grouped = nullptr; int4 *bins = nullptr; int bin_n_small = 0, bin_n_big = 0, bin_np_small = 64, bin_np_big = 64, bin_ctg_bits = 1;
	bool gs_ok = false; int gs_np = 64; // stage A's orders by k_genome_sort (one workgroup per genome, keys in LDS): every genome fits
	int2 *exon = 0; int32_t *prot_gid = 0; uint8_t *gene_pref = 0;
	// exchange vectors
	int32_t *max_ori = 0; int64_t *sums = 0; int32_t *vtx_cnt = 0; int32_t *g2s = 0; int32_t n_seg = 0;
	uint64_t sync_epoch_reset = 0;
	bool gf_ok = false; // k_genome_filters: the per-genome tables of read.c:254-256 fit the LDS
	bool gf_k32 = false; int gf_pos_bits = 0; // ... with 4-byte `best` entries (score_adj and a position inside a genome in 32 bits)
	bool x_redo = false; // pga_arc_round_x gave the round up: the next pga_arc_round repeats it on the sort path
	int64_t x_pairs_seen = 0, x_arcs_seen = 0; // sharded rounds: the longest pair list / the largest local arc table of any rank in the PREVIOUS run over this context (pga_begin shifts)
	int64_t x_pairs_run = 0, x_arcs_run = 0;   // ... and in the run under way
	int64_t x_pair_floor = 0, x_arc_floor = 0; // after a run that was void for want of room (its statistics are worth little): capacities not to go below
	int64_t *dcnt = 0;      // device counters: [0] triples [1] arcs-temp [2] misc [3] invariant flag, [4..7] hazards
	int64_t *h_cnt = 0;     // pinned mirror
	int64_t *h_box = 0;     // the same memory as the device sees it
	void *h_stage = nullptr; size_t h_stage_cap = 0; // pinned landing area of fetch_later
	void *h_fetch = nullptr; size_t h_fetch_cap = 0; // pinned landing area of pga_fetch
	int32_t *h_g2s = nullptr; size_t h_g2s_cap = 0; hipEvent_t g2s_done = nullptr; // pinned staging of flag_vtx's gene -> segment map
	DevPool pool; PinArena pin;
	bool walk_valid = false; // S_WALK_VAL / S_WALK_PREV match the current flags and cm order
	// gene-major index (k_genes.hpp): hits by (gene, genome, X position); half-arc records of the current walk
	int32_t *zx = 0, *zy = 0, *zg = 0; int2 *zst = 0; int32_t *zpos = 0, *zoff = 0; // gene-major planes (k_genes.hpp)
	// LIVE LISTS: flt is monotone and everything but the BED writers skips filtered hits, so when few hits are left the structures the
	// rounds iterate over -- the cm-order list of the walk and the gene-major index with its half-arc records -- are built over the hits without
	// flt only (ensure_z, at the first pg_gen_arc of a run: stage A, pg_post_process and graph.c:285-288 have filtered by then).  The X order
	// itself (records, flags, every plane) is not compacted: members carry F_MEMBER, ylist / zx hold X positions.
	bool live_on = false;    // the lists of this run hold members only; false: every hit (ylist == yperm, NL == N)
	int32_t NL = 0;          // entries of the cm-order list and of the gene-major index
	int32_t *lx = 0;         // [N + 1] members before X position x when the lists were built (a contig keeps its range and its count through order overrides: valid at contig starts)
	int32_t *ylist_buf = 0;  // [N] storage of the members' list in cm order
	const int32_t *ylist = 0;// what the walk reads: ylist_buf, or yperm when the lists hold every hit
	// The gene-major index behind the vertex greedy (round 6): pg_gen_vtx's host part (the greedy over genes, ~0.3-0.7 ms) leaves the device idle, and the
	// index of the first pg_gen_arc (a sort and three gathers: 0.1 ms at configs[1], 1.2 ms at 12.1 M hits) only needs what the upload and stage A left.
	// pga_vtx_partials arms this; the host's next fetch queues the index behind its copy and waits for the COPY alone (an event), not for the stream.
	bool z_early = false; hipEvent_t z_ev = nullptr;
	int32_t *ga_ctl = 0; // [2] k_gene_arcs_big's hand-out counters (cleared by the k_sweep_slow of the arc round's sweep)
	int4 *cA = 0, *cB = 0, *cC = 0; int32_t *cx = 0; // [N] live lists: the sweep's records of the members, compact, in X order (pm over the members), and each member's X position
	int2 *tg = 0; bool tg_valid = false; // [N] where the (contig, cs) tie group of a hit begins and ends in the cs order (k_tie_bounds: once per pass, members only)
	int64_t *live_cnt = 0;   // [LIVE_CNT_N] partial counts of the hits without flt (k_vtx1 / k_flag_vtx spread their atomics: 190 000 waves onto ONE word cost 1.9 ms at 12.1 M hits), summed into dcnt[8] by k_live_sum
	int64_t live_hint = -1;  // hits without flt as the vertex step counted them (k_vtx1): decides whether the lists are worth building; -1 not known
	int4 *wrec = 0; bool wrec_valid = false; // the walk's 32-byte records in cm order (k_pack_wrec): they carry the gene-major position, so a new index or a new cm order makes them stale
	uint32_t *hfk = 0, *hbk = 0; int4 *hfp = 0, *hbp = 0; // half-arc key words and payloads
	bool z_valid = false, ha_valid = false; uint32_t round_tag = 0; int ha_ori = -1;
	Gate gate = Gate{nullptr, 0};     // what the launches of the moment carry (pga_branch_loop sets it per phase; open everywhere else)
	int32_t *loopctl = nullptr;       // [4] device: Gate::w[0..1], [2] = tag of the last arc round of the loop that ran
	int loop_round = 0;               // the round the launches of the moment belong to (stamps)
	bool in_loop = false;             // inside pga_branch_loop
	bool loop_room_given = false;     // pga_branch_loop returned status 4 once on this context (tests: PANGENE_LOOP_PAIR_CAP applies until then)
	bool loop_gated = false;          // pga_branch_loop runs with gates: an arc round that runs leaves its tag in loopctl[2], whether its own gate is open or not
	int32_t *h_loopctl = nullptr;     // pinned mirror of loopctl (bump-allocated once per context)
	int32_t *h_ov = nullptr; size_t h_ov_cap = 0; // pinned: position / file-index lists of an order override, two halves used in turn
	hipEvent_t ov_ev[2] = { nullptr, nullptr }; bool ov_ev_used[2] = { false, false }; unsigned ov_seq = 0; // a half is free again when the copy out of it has happened
	bool zposy_stale = false; // the gene-major index stands but the cm order (or the X numbering) changed: the walk's records have to be packed again (ensure_z)
	const pga_arc_part_t *cur_tab = nullptr; int64_t cur_tab_n = 0; // the table of pga_arc_set_current
	bool table_sparse = false; // the current arc table lives in the genes' stretches (arc_round_genes) and has not been compacted
	int32_t *h_round = nullptr; size_t h_round_cap = 0; // pinned: segment counters + degrees of a round
	unsigned long long sync_epoch = 0, arc_epoch = 0; bool arc_deferred = false, arc_done = false, force_sort_once = false, sweep_done = false; std::vector<int32_t> def_host; // a round whose results nobody has waited for yet (pga_arc_round_finish)
	int4 *yrecA = 0, *yrecB = 0; bool yrec_valid = false; // Y-order static records (k_pack_yrec), rebuilt after anything that changes their sources
	int64_t br_np_seen = 0; // the last pair count the host got to know (sizes the next grid)
	int64_t br_n = 0, br_np = 0, br_cap = 0; int32_t br_S = 0; // arcs / pairs (-1: not known on the host yet) / pair capacity / segments of the last branch_pairs
	struct { double diff; int32_t local_dist, local_count, frag_mode; } br_par = { 0, 0, 0, 0 };
	int32_t *h_ndl = nullptr; size_t h_ndl_cap = 0; // pinned: n_dist_loci of a round
	std::vector<TimedLaunch> timed; bool timing_on = false; // HIP-event timing of kernel classes, switched on by pga_timing_reset (bench.py)
	bool timing_rounds = false; // ... also every pg_gen_arc round (class 5: sweep + walk scan + gene kernels = SURVEY 8(d)'s K2) and its walk scan alone (class 6); PANGENE_TIME_ROUNDS=1 at pga_timing_reset: two more events per round, so only for a pass that is not itself timed
	hipEvent_t span_a = nullptr; // start of stage A (pga_begin), paired with an event at the end of pga_ingest
	std::vector<void *> owned; void *arena = nullptr; size_t arena_cap = 0; // owned: allocations of their own (PANGENE_NO_ARENA); arena: the one block the persistent arrays are carved from
	std::vector<std::pair<void **, size_t>> plan; // persistent arrays waiting for the arena (create)
};

// persistent arrays are carved from ONE allocation: dalloc() only records the request, dalloc_commit() allocates and hands out
template <class T> static int dalloc(pga_ctx *c, T **p, size_t n)
{
	c->plan.emplace_back((void **)p, (((n ? n : 1) * sizeof(T)) + 255) & ~(size_t)255);
	return 0;
}

static int dalloc_commit(pga_ctx *c)
{
	size_t tot = 0;
	if (getenv("PANGENE_NO_ARENA")) { // debugging aid: one allocation per array (out-of-bounds accesses then land in padding)
		for (auto &e : c->plan) { void *q = nullptr; if (hipMalloc(&q, e.second) != hipSuccess) return PGA_ERR_NOMEM; *e.first = q; c->owned.push_back(q); if (poison_on()) (void)hipMemset(q, 0x5a, e.second); }
		c->plan.clear();
		return 0;
	}
	for (auto &e : c->plan) tot += e.second;
	size_t got = 0;
	void *base = dev_big_alloc(tot ? tot : 256, &got);
	if (base == nullptr) return PGA_ERR_NOMEM;
	c->arena = base, c->arena_cap = got;
	if (poison_on()) (void)hipMemset(base, 0x5a, tot ? tot : 256);
	size_t off = 0;
	for (auto &e : c->plan) *e.first = (char *)base + off, off += e.second;
	c->plan.clear();
	return 0;
}
