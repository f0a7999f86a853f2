// pga_backend.hip -- the MI355X (gfx950) implementation of the thin device ABI in
// include/pangene_hip.h.  All per-hit work of the pangene graph-construction path runs here as
// hand-written HIP kernels over a structure-of-arrays shard that stays resident in HBM for the whole
// run (upload once, 19 interval-dominance sweeps, 17 arc rounds, 15 branch rounds, one download).
//
// Data layout (DESIGN.md "HBM layout"): hits are physically stored in X order = (genome, contig, cs,
// file index), one 32-bit array per field, so a wave reads 256 B contiguous per field and the sweep's
// neighbours are adjacent in memory.  `seg` is the dense (genome, contig) id, `pm` the per-contig
// running maximum of ce (bounds the look-back of the sweep), `yperm` the cm order as a permutation of
// X positions.  Keys never change, so the two sorts the reference repeats 67 times per genome
// (hit.c:29-64) are done exactly once.
//
// Everything is integer work except three IEEE-double expressions (overlap.c:134,170) -- compile with
// -ffp-contract=off.  No MFMA: this path is HBM/latency bound.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <time.h>
#include <vector>
#include <mutex>
#include <atomic>
#include <condition_variable>
#include <algorithm>
#include "pangene_hip.h"
#include "dev_prims.hpp"

using namespace pgd;

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
	fprintf(stderr, "[E::pga] %s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return PGA_ERR_NO_DEVICE; } } while (0)

// debugging aid: PANGENE_POISON=1 fills every fresh device / pinned allocation with a pattern, so that a read of memory nobody
// wrote shows up the same way in every run (recycled memory otherwise holds whatever the previous context left there)
// a switch that takes a comma-separated list of words (PANGENE_FILTERS=k32,global  PANGENE_LOOP=nopre,nofinal,noskip)
static bool env_has(const char *name, const char *word)
{
	const char *e = getenv(name);
	const size_t n = strlen(word);
	for (; e && *e; ) { const char *c = strchr(e, ','); const size_t len = c ? (size_t)(c - e) : strlen(e); if (len == n && strncmp(e, word, n) == 0) return true; e = c ? c + 1 : nullptr; }
	return false;
}
// PANGENE_XLOOP_CAP=pairs[,arcs] (tests: exchange buffers of the sharded queued rounds that are too small at first; 0 or absent = as learned)
static long long xloop_cap(int which)
{
	const char *e = getenv("PANGENE_XLOOP_CAP");
	if (!e) return 0;
	if (which == 0) return atoll(e);
	const char *c = strchr(e, ',');
	return c ? atoll(c + 1) : 0;
}
static bool poison_on() { static const bool f = getenv("PANGENE_POISON") != nullptr; return f; }

#define F_HEAD 0x80000000u   // static: first hit of its genome in X order (index-0 quirk, overlap.c:108)
#define F_MULTI 0x40000000u  // static: the hit has more than one exon (lets the sweep skip the exon records)
#define F_CSTIE 0x20000000u  // static: an X-order neighbour shares (contig, cs) -- member of a tie group of the cs sort (hazard H2b; set by k_pack_rec)
#define F_PUBLIC 0x7ffu

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
	S_WALK_VAL, S_WALK_PREV, S_PERM, S_OVPOS, S_OVFILE, S_RUNSTART, S_CDN, S_MG_KEY, S_MG_VAL, S_MG_SRC, S_MG_OUT, S_MG_HEAD, S_MG_SLOT, S_MG_RUN, S_BR_S1, S_BR_GID, S_BR_VS, S_BR_VE, S_BR_PC, S_BR_POFF, S_BR_GRP, S_BR_NDL, S_BR_SEGGID, S_PAIRS, S_NLCNT, S_ARCX, S_ARCW, S_WEAKNEW, S_RP_SEG, S_RP_R, S_RP_CM, S_RP_POS, S_RP_IV, S_DL, S_SCRATCH, S_UPLOAD, S_RAW, S_ARC_STAGE, S_GMETA, S_GOFF, S_DEG, S_BIGLIST, S_STAGE_SID, S_STATS, S_G2S, S_MISC, S_SLOW, S_HZLIST, S_VWK, S_XG_BUF, S_XG_OUT, S_XG_OUT2, S_XSTAT, S_GS2LIST,
	S_COUNT
};

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
	int4 *wrec = 0; bool wrec_valid = false; // the walk's 32-byte records in cm order (k_pack_wrec): they carry the gene-major position, so a new index or a new cm order makes them stale
	uint32_t *hfk = 0, *hbk = 0; int4 *hfp = 0, *hbp = 0; // half-arc key words and payloads
	bool z_valid = false, ha_valid = false; uint32_t round_tag = 0; int ha_ori = -1;
	Gate gate = Gate{nullptr, 0};     // what the launches of the moment carry (pga_branch_loop sets it per phase; open everywhere else)
	int32_t *loopctl = nullptr;       // [4] device: Gate::w[0..1], [2] = tag of the last arc round of the loop that ran
	int loop_round = 0;               // the round the launches of the moment belong to (stamps)
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

extern "C" int pga_is_device(void) { return 1; }

// pinned host memory: what the reader packs the genomes into, so that the upload is plain DMA
extern "C" int pga_host_alloc(size_t nbytes, void **ptr)
{
	*ptr = nullptr;
	return hipHostMalloc(ptr, nbytes ? nbytes : 1, hipHostMallocDefault) == hipSuccess ? 0 : PGA_ERR_NOMEM;
}
extern "C" void pga_host_free(void *ptr) { if (ptr) (void)hipHostFree(ptr); }

extern "C" int pga_set_device(int32_t device) { if (hipSetDevice(device) != hipSuccess) return PGA_ERR_NO_DEVICE; g_last_dev.store(device); return 0; }
extern "C" int pga_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }

extern "C" void pga_host_trim(size_t keep_bytes)
{
	if (keep_bytes == 0) { // "give everything back": the cached device blocks too
		std::lock_guard<std::mutex> lk(g_dev_mu);
		for (DevBlock &b : g_dev_cache) (void)hipFree(b.p);
		g_dev_cache.clear();
	}
	std::lock_guard<std::mutex> lk(g_pin_mu);
	size_t kept = 0, n_keep = 0;
	for (; n_keep < g_pin_cache.size() && kept + g_pin_cache[n_keep].cap <= keep_bytes; ++n_keep) kept += g_pin_cache[n_keep].cap;
	for (size_t i = n_keep; i < g_pin_cache.size(); ++i) (void)hipHostFree(g_pin_cache[i].p);
	g_pin_cache.resize(n_keep);
}

extern "C" const char *pga_strerror(int code)
{
	switch (code) {
	case PGA_OK: return "ok";
	case PGA_ERR_NO_DEVICE: return "no usable HIP device / HIP runtime error (this library has no CPU fallback)";
	case PGA_ERR_RANGE: return "value out of range for the device layout";
	case PGA_ERR_ARG: return "bad argument";
	case PGA_ERR_NOMEM: return "out of device memory";
	case PGA_ERR_INVARIANT: return "reference invariant violated";
	}
	return "unknown";
}

// The kernels live in one header per part of the path; this file holds the context and the host side of the C ABI.
#include "k_common.hpp"
#include "k_ingest.hpp"
#include "k_sweep.hpp"
#include "k_segsort.hpp"
#include "k_segsort2.hpp"
#include "k_stage_b.hpp"
#include "k_vertex.hpp"
#include "k_arcs.hpp"
#include "k_branch.hpp"
#include "k_genes.hpp"
#include "k_order.hpp"

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
	v->slow_cnt = nullptr, v->slow_list = (int32_t *)c->pool.get(S_SLOW, sizeof(int32_t) * (size_t)c->N);
	v->hz_list = (int32_t *)c->pool.get(S_HZLIST, sizeof(int32_t) * PGA_HAZARD_CAP);
	if (!v->slow_list || !v->hz_list) return PGA_ERR_NOMEM;
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
	const bool timed = (timed_which == 0 || timed_which == 1) && c->timing_on; // (the stage-C sweeps are not timed one by one: two events per launch cost ~10 us of queue time)
	if (timed) {
		HIPCHK(hipEventCreate(&t.a)); HIPCHK(hipEventCreate(&t.b));
		if (reps != 1) HIPCHK(hipEventRecord(t.a, c->st));
	}
	c->walk_valid = false, c->ha_valid = false;
	const int nt = (int)nblk(c->N, SW_TILE);
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
		if (c->any_multi) hipExtLaunchKernelGGL((k_sweep<MODE, true>), dim3(nt), dim3(SW_TILE), 0, c->st, ea, eb, 0, v);
		else hipExtLaunchKernelGGL((k_sweep<MODE, false>), dim3(nt), dim3(SW_TILE), 0, c->st, ea, eb, 0, v);
		hipLaunchKernelGGL((k_sweep_slow<MODE>), dim3((unsigned)std::min<int64_t>(2 * c->n_cu, std::max<int64_t>(64, nblk(c->N)))), dim3(BLOCK), 0, c->st, v, (long long *)(c->dcnt + 12 + ((c->sweep_seq + 1) & 1))); // (grid-stride over a list whose length only the device knows)
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

static size_t pool_want(int64_t N, int64_t GL, int64_t P, int64_t Q, int64_t raw_words)
{
	const size_t per_hit = 568 /* measured: 530-540 B/hit at 1 M and 12 M hits (PANGENE_TIMING reports the fit at destroy) */, tables = (size_t)GL * ((size_t)P * 12 + (size_t)Q * 36) + (size_t)Q * 512 + (size_t)P * 64;
	return ((size_t)N * per_hit + tables + (64u << 20) + (size_t)raw_words * 4 + 255) & ~(size_t)255;
}

// the persistent arrays of a context (one allocation: dalloc_commit); also what pga_reserve sizes its first block by
static int plan_persistent(pga_ctx *c)
{
	const int N = c->N, E = c->E, GL = c->n_genome;
	TRY(dalloc(c, &c->dcnt, 16)); TRY(dalloc(c, &c->loopctl, 4));
	// persistent arrays
	TRY(dalloc(c, &c->fidx, N)); TRY(dalloc(c, &c->gnm, N)); TRY(dalloc(c, &c->seg, N)); TRY(dalloc(c, &c->pid, N)); TRY(dalloc(c, &c->gid, N));
	TRY(dalloc(c, &c->cs, N)); TRY(dalloc(c, &c->ce, N)); TRY(dalloc(c, &c->cm, N)); TRY(dalloc(c, &c->cds, N)); TRY(dalloc(c, &c->nex, N));
	TRY(dalloc(c, &c->offx, N)); TRY(dalloc(c, &c->sori, N)); TRY(dalloc(c, &c->sadj, N)); TRY(dalloc(c, &c->pm, N)); TRY(dalloc(c, &c->rk, N)); TRY(dalloc(c, &c->recA, N)); TRY(dalloc(c, &c->recB, N)); TRY(dalloc(c, &c->recC, N)); TRY(dalloc(c, &c->yrecA, N)); TRY(dalloc(c, &c->yrecB, N));
	TRY(dalloc(c, &c->rank, N)); TRY(dalloc(c, &c->sdom, N)); TRY(dalloc(c, &c->pdom, N)); TRY(dalloc(c, &c->pdom0, N)); TRY(dalloc(c, &c->flags, N));
	TRY(dalloc(c, &c->yperm, N)); TRY(dalloc(c, &c->goff, GL + 1)); TRY(dalloc(c, &c->ggl, GL)); TRY(dalloc(c, &c->ctg_base, GL + 1)); TRY(dalloc(c, &c->inv, N)); TRY(dalloc(c, &c->headpos, GL + 1)); TRY(dalloc(c, &c->exon, E));
	TRY(dalloc(c, &c->eoff, GL + 1)); TRY(dalloc(c, &c->woff, GL + 1));
	TRY(dalloc(c, &c->zx, N)); TRY(dalloc(c, &c->zy, N)); TRY(dalloc(c, &c->zg, N)); TRY(dalloc(c, &c->zst, N)); TRY(dalloc(c, &c->zpos, N)); TRY(dalloc(c, &c->wrec, 2 * (size_t)N)); TRY(dalloc(c, &c->zoff, (size_t)c->Q + 2));
	TRY(dalloc(c, &c->hfk, N)); TRY(dalloc(c, &c->hbk, N)); TRY(dalloc(c, &c->hfp, N)); TRY(dalloc(c, &c->hbp, N));
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
	c->h_cnt = (int64_t *)c->pin.get(16 * sizeof(int64_t));
	if (!c->h_cnt) return PGA_ERR_NOMEM;
	memset(c->h_cnt, 0, 16 * sizeof(int64_t));
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
	c->any_multi = multi;
	c->ctg_bits = bits_for((uint32_t)(max_ctg - 1));
	c->gs_np = std::max(64, (max_hit + 63) & ~63);
	c->gs_ok = c->gs_np <= GS_NP_MAX && c->rk_shift >= 0 && getenv("PANGENE_GLOBAL_SORT") == nullptr;
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
		const size_t want = pool_want(N, GL, c->P, c->Q, woff[(size_t)GL]);
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
	for (const Run &r : runs) HIPCHK(hipMemcpyAsync(raw + r.dev_word, r.base, r.bytes, hipMemcpyHostToDevice, c->st));
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
	}
	HIPCHK(hipMemcpyAsync(c->h_cnt, c->dcnt, 16 * sizeof(int64_t), hipMemcpyDeviceToHost, c->st));
	int rc = sync_st(c); // the caller's blocks and tables have been read
	if (rc == 0 && c->h_cnt[8]) { // k_unblock: a hit outside the device layout, or beyond what its block declared (direct users of this ABI: the host driver checks while it packs)
		fprintf(stderr, "[E::pga_create] %lld hit(s) with coordinates, contig ids or scores outside what their genome block declares\n", (long long)c->h_cnt[8]);
		rc = PGA_ERR_RANGE;
	}
	c->exon_regular = rc != 0 || c->h_cnt[9] == 0; // (dcnt[9] is the rounds' overflow counter later on; pga_begin clears it)
	if (timing) fprintf(stderr, "[pga_create] allocations %.3f ms, upload of %.1f MB in %zu copy command(s) + unpack %.3f ms\n", (t1 - t0) * 1e3, woff[(size_t)GL] * 4e-6, runs.size(), (now() - t1) * 1e3);
	return rc;
}

// per-hit constants in file order, X order (sort + gather), running max of ce, Y order; resets all state
extern "C" int pga_begin(pga_ctx_t *c)
{
	c->yrec_valid = false, c->wrec_valid = false, c->z_valid = false, c->zposy_stale = false;
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
	if (c->gs_ok) { // one launch: both orders, every per-hit constant, the packed records (k_segsort.hpp)
		HitArrays o = { c->fidx, c->gnm, c->seg, c->pid, c->gid, c->cs, c->ce, c->cm, c->cds, c->nex, c->offx, c->sori, c->sadj, c->rank, c->sdom, c->pdom, c->pdom0, c->rk, c->flags };
		GenomeSort gs = { up, (int64_t)N, c->goff, c->ctg_base, c->cs_bits, c->cm_bits, c->ctg_bits, c->gs_np, GL,
		                  o, c->yperm, c->headpos, c->recA, c->recB, c->recC, nullptr, nullptr };
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
	int32_t *rk_f = c->rk_shift >= 0 ? up + 15 * (size_t)N : (int32_t *)c->pool.get(S_TAB_A, sizeof(uint64_t) * (size_t)N);
	int32_t *head = (int32_t *)c->pool.get(S_HEAD, sizeof(int32_t) * ((size_t)N + 1)), *incl = (int32_t *)c->pool.get(S_SLOT, sizeof(int32_t) * ((size_t)N + 1));
	uint64_t *key = (uint64_t *)c->pool.get(S_KEY_A, 0);
	uint32_t *val = (uint32_t *)c->pool.get(S_VAL_A, 0);
	if (!up || !rk_f || !head || !incl || !key || !val) return PGA_ERR_NOMEM;
	FileHits f = { f_pid, f_cid, f_rank, f_sori, f_sadj, f_nex, f_offx, f_cs, f_ce, f_cm, f_rev };
	if (c->rk_shift < 0) hipLaunchKernelGGL(k_score_key, dim3(nblk(N)), dim3(BLOCK), 0, c->st, f_pid, f_sadj, f_gid, c->gene_pref, N, key, val);
	uint64_t *ks; uint32_t *vs;
	if (c->rk_shift < 0) { // dense rank of the 64-bit score keys (see k_rank_scatter)
		TRY(radix_sort_pool(c, key, val, N, c->sc_bits, &ks, &vs));
		hipLaunchKernelGGL(k_arc_head, dim3(nblk(N)), dim3(BLOCK), 0, c->st, ks, (int64_t)N, head);
		I32 *tile = (I32 *)c->pool.get(S_TILE, tile_buf_bytes(N));
		device_scan<I32>(InI32{head}, OutInclI32{incl}, N, tile, OpSum{}, I32{0}, c->st);
		hipLaunchKernelGGL(k_rank_scatter, dim3(nblk(N)), dim3(BLOCK), 0, c->st, ks, vs, incl, N, rk_f);
	}
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
			if (!tmax || !tmin || !tr1) return PGA_ERR_NOMEM;
			HIPCHK(hipMemsetAsync(tmax, 0, sizeof(int32_t) * (size_t)TP, c->st));
			hipLaunchKernelGGL(k_fill_i32, dim3(nblk(TP)), dim3(BLOCK), 0, c->st, tmin, TP, INT32_MAX);
			hipLaunchKernelGGL(k_fill_i32, dim3(nblk(TP)), dim3(BLOCK), 0, c->st, tr1, TP, INT32_MAX);
			hipLaunchKernelGGL(k_pseudo1, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->gnm, c->pid, c->nex, N, P, tmax, tmin);
			hipLaunchKernelGGL(k_pseudo2, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->gnm, c->pid, c->nex, c->rank, c->flags, N, P, tmax, tmin, tr1, k_stats);
			hipLaunchKernelGGL(k_pseudo3, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->gnm, c->pid, c->rank, N, P, tmax, tmin, tr1);
			hipLaunchKernelGGL(k_pack_rank, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->rank, N, c->recC); // rank changed
		}
		unsigned long long *tbest = c->gf_ok ? nullptr : (unsigned long long *)c->pool.get(S_TAB_D, sizeof(uint64_t) * (size_t)TQ);
		uint8_t *noiso = c->gf_ok ? nullptr : (uint8_t *)c->pool.get(S_TAB_A, (size_t)TP + 16); // byte (genome, protein): the protein has a hit there without flt_iso_ov
		if (!c->gf_ok && (!tbest || !noiso)) return PGA_ERR_NOMEM;
		// read.c:248-254: ONE sweep for pg_shadow(cal_dom_sc=1), the reset behind it and pg_flt_ov_isoform (k_sweep<3>: they walk the same pairs)
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

extern "C" int pga_post_partials(pga_ctx_t *c, int32_t **max_ori, int64_t **sums)
{
	zero_multi(c, c->max_ori, sizeof(int32_t) * (size_t)std::max(1, c->P), c->sums, sizeof(int64_t) * 6 * (size_t)std::max(1, c->P));
	if (c->N) hipLaunchKernelGGL(k_post_part, dim3(nblk(c->N)), dim3(BLOCK), 0, c->st, c->flags, c->pid, c->rank, c->sori, c->sadj, c->nex, c->N, c->P,
	                             c->max_ori, (unsigned long long *)c->sums);
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
		*records = (uint64_t *)rec;
		if (N == 0) return sync_st(c);
		hipLaunchKernelGGL(k_vtx1, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->flags, c->gnm, c->gid, c->rank, c->pdom, N, Q, c->vtx_cnt, bits, wpg, c->dcnt);
		hipLaunchKernelGGL(k_vtx_fold, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->flags, c->gnm, c->gid, c->rank, c->pdom, c->prot_gid, c->ggl, N, bits, wpg,
		                   dom_tab, pbits, nw, rec + n_slot * (1 + nw), ovf_cap, c->dcnt);
		I32 *tile = (I32 *)c->pool.get(S_TILE, tile_buf_bytes(n_slot));
		if (!tile) return PGA_ERR_NOMEM;
		device_scan<I32>(InDomSet{dom_tab}, OutExclI32{slot}, n_slot, tile, OpSum{}, I32{0}, c->st);
		hipLaunchKernelGGL(k_vtx_compact, dim3(nblk(n_slot)), dim3(BLOCK), 0, c->st, dom_tab, slot, n_slot, pbits, nw, rec, c->dcnt, c->h_box);
		TRY(sync_st(c));
		if (c->h_cnt[3]) return PGA_ERR_INVARIANT;
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
static int ensure_z(pga_ctx *c)
{
	if (c->N == 0) return 0;
	if (c->z_valid) {
		if (c->zposy_stale) c->wrec_valid = false, c->zposy_stale = false, c->ha_valid = false;
		return 0;
	}
	const int N = c->N;
	uint64_t *key = (uint64_t *)c->pool.get(S_KEY_A, 0);
	uint32_t *val = (uint32_t *)c->pool.get(S_VAL_A, 0);
	if (!key || !val) return PGA_ERR_NOMEM;
	hipLaunchKernelGGL(k_zkey, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->gid, N, key, val);
	uint64_t *ks; uint32_t *vs;
	TRY(radix_sort_pool(c, key, val, N, bits_for((uint32_t)std::max(1, c->Q)), &ks, &vs));
	hipLaunchKernelGGL(k_zrec, dim3(nblk(N)), dim3(BLOCK), 0, c->st, vs, ks, c->ctg_base, c->n_genome, c->flags, c->cm, c->seg, N, ZIndex{c->zx, c->zy, c->zg, c->zst, c->zpos});
	hipLaunchKernelGGL(k_zoff, dim3(nblk(c->Q + 1)), dim3(BLOCK), 0, c->st, ks, N, c->Q, c->zoff);
	c->z_valid = true, c->ha_valid = false, c->zposy_stale = false, c->wrec_valid = false;
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
		hipLaunchKernelGGL(k_pack_wrec, dim3(nblk(c->N)), dim3(BLOCK), 0, c->st, WrecSrc{c->yperm, c->seg, c->gid, c->cm, c->sori, c->sdom, c->pdom0, c->prot_gid, c->flags, c->zpos, c->vfirst, c->vbase}, c->N, c->wrec);
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
	hipLaunchKernelGGL(k_walk, dim3(nblk(c->N, WK_TILE)), dim3(BLOCK), 0, c->st, Walk{c->flags, c->yperm, c->wrec, c->g2s, c->hfk, c->hbk, c->hfp, c->hbp, c->round_tag, use_ori, c->N, c->dcnt, hzl, c->gate});
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
	                t.ax, t.s1, t.agid, t.aw, t.vs, t.ve, t.dg, t.vwk, h_round_dev, big, c->dcnt, c->gate, c->gate.w ? c->loopctl + 2 : (int32_t *)nullptr };
	hipLaunchKernelGGL(k_gene_arcs_wave, dim3((unsigned)c->Q), dim3(BLOCK), 0, c->st, ga);
	hipLaunchKernelGGL(k_gene_arcs_big, dim3((unsigned)std::min(c->Q, 8 * c->n_cu)), dim3(BLOCK), 0, c->st, ga);
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

extern "C" int pga_rep_pos(pga_ctx_t *c)
{
	const int N = c->N, GL = c->n_genome, Q = c->Q;
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
		device_scan<I32>(InWalkX{c->flags}, OutRank{rx, c->flags}, N, tile, OpSum{}, I32{0}, c->st, c->gate); // rank among the walkable hits, cs order
		RepFill rf = { n_ent, GL, Q, N, c->zx, c->zy, c->zg, c->zst, c->zoff, c->hbk, c->round_tag, c->recA, c->gid, c->flags, rx, c->goff, c->ctg_base, (void *)rp, iv, c->dcnt, hzl, c->vfirst, c->vbase, c->gate };
		const unsigned nb = nblk(std::max(N, Q));
		if (c->rp_form == RP_COMPACT) hipLaunchKernelGGL((k_rep_fill<RP_COMPACT>), dim3(nb), dim3(BLOCK), 0, c->st, rf);
		else if (c->rp_form == RP_WIDE) hipLaunchKernelGGL((k_rep_fill<RP_WIDE>), dim3(nb), dim3(BLOCK), 0, c->st, rf);
		else hipLaunchKernelGGL((k_rep_fill<RP_FULL>), dim3(nb), dim3(BLOCK), 0, c->st, rf);
	} else if (n_ent) {
		hipLaunchKernelGGL(k_fill_i32, dim3(nblk(4 * n_ent)), dim3(BLOCK), 0, c->st, (int32_t *)rp, 4 * n_ent, -1); // "absent" in either record form
	}
	return 0;
}

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
	const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(nblk(est, BLOCK / WAVE * NL_PAIRS), 1 << 20));
	if (n && c->rp_form == RP_COMPACT) hipLaunchKernelGGL((k_n_local<RP_COMPACT>), dim3(grid), dim3(BLOCK), 0, c->st, d_pairs, n, np_dev, c->n_genome, (const void *)rp, local_dist, local_count, frag_mode, d_cnt, hz, c->gate);
	else if (n && c->rp_form == RP_WIDE) hipLaunchKernelGGL((k_n_local<RP_WIDE>), dim3(grid), dim3(BLOCK), 0, c->st, d_pairs, n, np_dev, c->n_genome, (const void *)rp, local_dist, local_count, frag_mode, d_cnt, hz, c->gate);
	else if (n) hipLaunchKernelGGL((k_n_local<RP_FULL>), dim3(grid), dim3(BLOCK), 0, c->st, d_pairs, n, np_dev, c->n_genome, (const void *)rp, local_dist, local_count, frag_mode, d_cnt, hz, c->gate);
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

// enumerate the pairs and count them (k_br_wave<1>, k_n_local) for the pair count in dcnt[15] (capacity c->br_cap)
static int branch_enumerate(pga_ctx *c, int32_t **cnt)
{
	const int n_vtx = 2 * c->br_S;
	int32_t *s1 = (int32_t *)c->pool.get(S_BR_S1, 0), *agid = (int32_t *)c->pool.get(S_BR_GID, 0), *vs = (int32_t *)c->pool.get(S_BR_VS, 0), *ve = (int32_t *)c->pool.get(S_BR_VE, 0);
	int32_t *poff = (int32_t *)c->pool.get(S_BR_POFF, 0);
	int32_t *pairs = (int32_t *)c->pool.get(S_PAIRS, sizeof(int32_t) * 2 * (size_t)c->br_cap + 16);
	if (!pairs || !s1 || !agid || !vs || !ve || !poff) return PGA_ERR_NOMEM;
	hipLaunchKernelGGL((k_br_wave<1>), dim3(nblk(n_vtx, BLOCK / WAVE)), dim3(BLOCK), 0, c->st, n_vtx, vs, ve, s1, agid, c->br_par.diff, poff, pairs, c->br_cap, (const int32_t *)c->pool.get(S_BR_PC, 0),
	                   (const int32_t *)nullptr, 0.0, 0.0, (uint8_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, c->dcnt, (uint8_t *)nullptr, (const int64_t *)nullptr, c->gate);
	return n_local_dev(c, pairs, c->br_cap, c->dcnt + 15, c->br_par.local_dist, c->br_par.local_count, c->br_par.frag_mode, cnt);
}

extern "C" int pga_branch_pairs(pga_ctx_t *c, const uint64_t *arc_x, const int32_t *arc_s1, int64_t n_arc, const int32_t *seg_gid, int32_t n_seg,
                                double branch_diff, int32_t local_dist, int32_t local_count, int32_t frag_mode, int32_t **cnt, int64_t *n_pairs)
{
	if (arc_x == nullptr) n_arc = c->br_n, n_seg = c->br_S; // the table of pga_arc_set_current
	const int n_vtx = 2 * n_seg;
	uint64_t *ax = (uint64_t *)c->pool.get(S_ARCX, sizeof(uint64_t) * (size_t)n_arc + 16);
	uint8_t *aw = (uint8_t *)c->pool.get(S_ARCW, (size_t)n_arc + 16);
	int32_t *s1 = (int32_t *)c->pool.get(S_BR_S1, sizeof(int32_t) * (size_t)n_arc + 16), *agid = (int32_t *)c->pool.get(S_BR_GID, sizeof(int32_t) * (size_t)n_arc + 16);
	int32_t *vs = (int32_t *)c->pool.get(S_BR_VS, sizeof(int32_t) * (size_t)n_vtx + 16), *ve = (int32_t *)c->pool.get(S_BR_VE, sizeof(int32_t) * (size_t)n_vtx + 16);
	int32_t *pc = (int32_t *)c->pool.get(S_BR_PC, sizeof(int32_t) * (size_t)n_vtx + 16), *poff = (int32_t *)c->pool.get(S_BR_POFF, sizeof(int32_t) * (size_t)n_vtx + 16);
	int32_t *sg = (int32_t *)c->pool.get(S_BR_SEGGID, sizeof(int32_t) * (size_t)n_seg + 16);
	if (!ax || !aw || !s1 || !agid || !vs || !ve || !pc || !poff || !sg) return PGA_ERR_NOMEM;
	c->br_n = n_arc, c->br_S = n_seg, c->br_np = -1;
	c->br_par.diff = branch_diff, c->br_par.local_dist = local_dist, c->br_par.local_count = local_count, c->br_par.frag_mode = frag_mode;
	if (n_pairs) *n_pairs = 0;
	*cnt = (int32_t *)c->pool.get(S_NLCNT, 16);
	if (n_arc == 0 || n_vtx == 0) { c->br_np = 0; return sync_st(c); }
	if (arc_x) {
		TRY(upload(c, ax, arc_x, (size_t)n_arc)); TRY(upload(c, s1, arc_s1, (size_t)n_arc)); TRY(upload(c, sg, seg_gid, (size_t)n_seg));
		HIPCHK(hipMemsetAsync(vs, 0, sizeof(int32_t) * (size_t)n_vtx, c->st)); HIPCHK(hipMemsetAsync(ve, 0, sizeof(int32_t) * (size_t)n_vtx, c->st));
		hipLaunchKernelGGL(k_br_prep, dim3(nblk(n_arc)), dim3(BLOCK), 0, c->st, ax, n_arc, sg, agid, vs, ve);
		HIPCHK(hipMemsetAsync(aw, 0, (size_t)n_arc, c->st)); // (the tables of arc_round_local / arc_set_current arrive with weak_br = 0)
	}
	static const bool general_scan = getenv("PANGENE_PAIR_SCAN_GENERAL") != nullptr; // (tests: the path of graphs with more than 65536 vertices)
	const bool one_wg = n_vtx <= PO_THREADS * PO_MAX_ITEMS && !general_scan;
	hipLaunchKernelGGL(k_br_count, dim3(nblk(n_vtx)), dim3(BLOCK), 0, c->st, n_vtx, vs, ve, s1, branch_diff, pc, c->gate);
	if (one_wg) hipLaunchKernelGGL(k_pair_offsets, dim3(1), dim3(PO_THREADS), 0, c->st, (const int32_t *)pc, n_vtx, poff, c->dcnt, c->h_box, n_pairs ? -1ll : (long long)std::max<int64_t>(c->br_cap, 4 * (int64_t)n_vtx), c->gate); // offsets, and dcnt[15] = number of pairs
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
	if (c->br_cap < 4 * (int64_t)n_vtx) c->br_cap = 4 * (int64_t)n_vtx;
	return branch_enumerate(c, cnt);
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

static int loop_exchange_table(pga_ctx *c, const LoopX &L)
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
		device_scan<I32>(InGmeta{gmeta}, OutExclI32{goff}, S, tile, OpSum{}, I32{0}, c->st);
		hipLaunchKernelGGL(k_xs_compact, dim3(nblk(S, BLOCK / WAVE)), dim3(BLOCK), 0, c->st, gmeta, goff, S, stage, seg_cnt, L.gbuf, L.arc_cap, (const int64_t *)c->dcnt);
	}
	else HIPCHK(hipMemsetAsync(L.gbuf, 0, sizeof(int32_t) * (size_t)(XS_HDR + xs_seg_words(S)), c->st)); // a rank without hits: an empty table, no counts
	{ const int rc = L.x->allgather(L.x->user, L.gbuf, L.gbuf + L.slot_words, L.slot_words * (int64_t)sizeof(int32_t)); if (rc) return rc; }
	XSlots X = { L.gbuf + L.slot_words, L.slot_words, L.arc_cap, W, S };
	hipLaunchKernelGGL(k_xs_sum_rank, dim3(nblk(std::max<int64_t>(mcap, n_vtx))), dim3(BLOCK), 0, c->st, X, seg_cnt, L.d_off, c->dcnt, L.xstat, key, val);
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
	if (par->final_on && (x != nullptr || seg_cnt_host == nullptr || ndl_host == nullptr)) return 2;
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
	if (par->pre_on) { // graph 2 (graph.c:293-296): pg_flt_high_occ on graph 1's table (no branch step has run: n_dist_loci = 0), PG_SET_FILTER(vtx == 0), pg_gen_arc
		if (x) return 2;
		HIPCHK(hipMemsetAsync(ndl, 0, sizeof(int32_t) * (size_t)n_vtx, c->st));
		int32_t *vs = (int32_t *)c->pool.get(S_BR_VS, 0), *ve = (int32_t *)c->pool.get(S_BR_VE, 0), *sg = (int32_t *)c->pool.get(S_BR_SEGGID, 0), *dg = (int32_t *)c->pool.get(S_DEG, 0), *seg_cnt = (int32_t *)c->pool.get(S_SEGCNT, 0);
		uint8_t *vwk = (uint8_t *)c->pool.get(S_VWK, (size_t)n_vtx + 16);
		if (!vs || !ve || !sg || !dg || !seg_cnt || !vwk) return PGA_ERR_NOMEM;
		hipLaunchKernelGGL(k_round_del, dim3(nblk(S)), dim3(BLOCK), 0, c->st, S, (const int32_t *)ndl, par->pre_max_tot_cnt, par->pre_max_degree, par->pre_max_dist_loci, (const int32_t *)sg, c->g2s, vs, ve, dg, seg_cnt, vwk, alive,
		                   par->final_on ? (int4 *)c->pool.get(S_GMETA, 0) : (int4 *)nullptr);
		hipLaunchKernelGGL(k_flag_vtx, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->flags, c->gid, N, c->g2s, 1);
		c->walk_valid = false, c->ha_valid = false;
		int32_t *sc2, *deg2;
		TRY(arc_round_genes(c, par->use_ori, &sc2, &deg2, nullptr, false));
		c->br_n = 2 * (int64_t)N + 2, c->br_S = S, c->br_np = 0;
	}
	// The fixed point (dev_prims.hpp: Gate).  Inside this loop the tie orders stand still (the caller asked exact_quiet), so a round that
	// marks no hit and deletes no segment leaves a state every later round reproduces: their kernels are queued all the same -- the
	// host does not look -- and leave at once.  Human-shaped shards reach it after three or four of the fifteen rounds; bacterial ones
	// as a rule do not.  Sharded runs keep every round: their collectives are queued by the host, and "nothing changed" would have to
	// hold on every rank.
	static const bool no_skip = env_has("PANGENE_LOOP", "noskip");
	const bool gated = x == nullptr && !no_skip && (int64_t)c->round_tag + n_round + 4 < (int64_t)HA_TAG_MAX;
	struct GateScope { pga_ctx *c; ~GateScope() { c->gate = Gate{nullptr, 0}; } } gate_scope{c}; // (every way out of this function leaves the launches open)
	const uint32_t tag_before = c->round_tag;
	if (gated) HIPCHK(hipMemsetAsync(c->loopctl, 0xff, 4 * sizeof(int32_t), c->st)); // -1: nothing has happened yet; round 0 runs (its branch steps ask for a change in round -1 or later)
	for (int r = 0; r < n_round; ++r) {
		c->loop_round = r;
		c->gate = gated ? Gate{c->loopctl, r - 1} : Gate{nullptr, 0}; // the branch steps of round r: something changed in round r - 1
		// pg_mark_branch_flt_arc (branch.c:48-106)
		TRY(pga_rep_pos(c));
		int32_t *cnt;
		TRY(pga_branch_pairs(c, nullptr, nullptr, 0, nullptr, S, par->branch_diff, par->local_dist, par->local_count, par->frag_mode, &cnt, nullptr));
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
				hipLaunchKernelGGL(k_flag_vtx, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->flags, c->gid, N, c->g2s, 1, c->gate);
				c->walk_valid = false, c->ha_valid = false;
			}
		}
		if (r + 1 < n_round || par->final_on) { // pg_gen_arc (graph.c:313)
			int32_t *seg_cnt, *deg;
			TRY(arc_round_genes(c, par->use_ori, &seg_cnt, &deg, nullptr, false)); // (no mail: the kernels raise the sticky flag themselves)
			c->br_n = 2 * (int64_t)N + 2, c->br_S = S, c->br_np = 0;
			if (x) { TRY(loop_exchange_table(c, L)); c->br_n = L.ecap; }
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
	const size_t fetch_need = (size_t)S + 128 + (par->final_on ? 4 * sizeof(int32_t) * (size_t)S + 64 : 0);
	if (c->h_fetch_cap < fetch_need) {
		c->h_fetch = c->pin.get(fetch_need + fetch_need / 2 + 512);
		if (!c->h_fetch) return PGA_ERR_NOMEM;
		c->h_fetch_cap = fetch_need + fetch_need / 2 + 512;
	}
	HIPCHK(hipMemcpyAsync(c->h_fetch, alive, (size_t)S, hipMemcpyDeviceToHost, c->st));
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
	else if (c->h_cnt[11]) return c->h_cnt[3] ? PGA_ERR_INVARIANT : 1;
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
		hipLaunchKernelGGL(k_mark_hits_z, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->zx, c->zy, c->zg, c->hfk, c->hbk, c->round_tag, N, c->g2s, ax, aw, vs, ve, vwk, c->flags,
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
	const bool partial = c->ha_valid && c->wrec_valid && z_keep && !c->zposy_stale && c->vfirst == nullptr && n_seg > 0 && seg_off[n_seg] * 8 <= (int64_t)N;
	c->walk_valid = false, c->ha_valid = false, c->yrec_valid = false, c->wrec_valid = false;
	if (!z_keep) c->z_valid = false;
	if (n_seg <= 0 || N == 0) return 0;
	const int64_t T = seg_off[n_seg];
	if (T == 0) { if (partial) c->ha_valid = true, c->wrec_valid = true; return 0; }
	auto walk_again = [&](const int32_t *d_pos) -> int { // (after the override's own kernels, on the same stream)
		int32_t *hzl = (int32_t *)c->pool.get(S_HZLIST, sizeof(int32_t) * PGA_HAZARD_CAP);
		if (!hzl) return PGA_ERR_NOMEM;
		hipLaunchKernelGGL(k_pack_wrec_list, dim3(nblk(T)), dim3(BLOCK), 0, c->st, WrecSrc{c->yperm, c->seg, c->gid, c->cm, c->sori, c->sdom, c->pdom0, c->prot_gid, c->flags, c->zpos, c->vfirst, c->vbase}, d_pos, T, c->wrec);
		hipLaunchKernelGGL(k_walk_list, dim3(nblk(T)), dim3(BLOCK), 0, c->st, Walk{c->flags, c->yperm, c->wrec, c->g2s, c->hfk, c->hbk, c->hfp, c->hbp, c->round_tag, c->ha_ori, c->N, c->dcnt, hzl, Gate{nullptr, 0}}, d_pos, T);
		c->ha_valid = true, c->wrec_valid = true, c->zposy_stale = false;
		return 0;
	};
	// positions and file indices of the overridden hits, built in page-locked memory (a real DMA; from a std::vector the runtime stages)
	// (nothing waits at the end of an override any more -- sixty-six of them per pass each found the device still at the round queued
	// before -- so the lists must not be overwritten while their copy is under way: two halves, an event each)
	const size_t ov_bytes = (sizeof(int32_t) * 2 * (size_t)T + 255) & ~(size_t)255;
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
	int32_t *pos = (int32_t *)((char *)c->h_ov + (size_t)half * c->h_ov_cap), *fil = pos + T;
	for (int32_t s = 0; s < n_seg; ++s) {
		const int32_t g = seg_genome[s], base = c->h_goff[(size_t)g];
		for (int64_t k = seg_off[s]; k < seg_off[s + 1]; ++k)
			pos[(size_t)k] = base + seg_start[s] + (int32_t)(k - seg_off[s]), fil[(size_t)k] = base + file_idx[k];
	}
	int32_t *d_pos = (int32_t *)c->pool.get(S_OVPOS, sizeof(int32_t) * (size_t)T), *d_fil = (int32_t *)c->pool.get(S_OVFILE, sizeof(int32_t) * (size_t)T);
	int32_t *remap = (int32_t *)c->pool.get(S_I32_B, sizeof(int32_t) * (size_t)N);
	if (!d_pos || !d_fil || !remap) return PGA_ERR_NOMEM;
	TRY(upload(c, d_pos, pos, (size_t)T)); TRY(upload(c, d_fil, fil, (size_t)T));
	HIPCHK(hipEventRecord(c->ov_ev[half], c->st)); c->ov_ev_used[half] = true;
	if (!c->inv_valid) { hipLaunchKernelGGL(k_inv_only, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->fidx, c->gnm, c->goff, N, c->inv); c->inv_valid = true; } // (then kept current by the overrides themselves)
	if (which == 1) {
		hipLaunchKernelGGL(k_ov_sety, dim3(nblk(T)), dim3(BLOCK), 0, c->st, d_pos, d_fil, T, c->inv, c->yperm);
		if (z_keep) c->zposy_stale = true;
		if (partial) TRY(walk_again(d_pos));
		return 0;
	}
	int32_t *tmp = (int32_t *)c->pool.get(S_PERM, sizeof(int32_t) * (OV_PLANES + 13) * (size_t)T + 64);
	if (!tmp) return PGA_ERR_NOMEM;
	PermArrays p = { { c->fidx, c->pid, c->gid, c->cm, c->nex, c->sadj, c->rank, c->sdom, c->pdom, c->pdom0, (int32_t *)c->flags, c->sori }, { c->recA, c->recB, c->recC } };
	static_assert(OV_FLAGS == 10, "the flag word's place in PermArrays");
	hipLaunchKernelGGL(k_ov_gather, dim3(nblk(T)), dim3(BLOCK), 0, c->st, p, d_pos, d_fil, T, c->inv, tmp, remap, z_keep ? (const int32_t *)c->zpos : (const int32_t *)nullptr);
	hipLaunchKernelGGL(k_ov_scatter, dim3(nblk(T)), dim3(BLOCK), 0, c->st, p, d_pos, d_fil, T, tmp, c->gnm, c->goff, c->inv, c->zx, z_keep ? c->zpos : (int32_t *)nullptr);
	hipLaunchKernelGGL(k_ov_remap_y, dim3(nblk(T)), dim3(BLOCK), 0, c->st, c->yperm, d_pos, T, remap);
	if (z_keep) c->zposy_stale = true;
	SegMax *tile = (SegMax *)c->pool.get(S_TILE, tile_buf_bytes(T));
	if (!tile) return PGA_ERR_NOMEM;
	device_scan<SegMax>(InSegMaxList{c->recA, d_pos}, OutSegMaxList{c->recA, d_pos}, T, tile, OpSegMax{}, SegMax{SEG_EMPTY, 0}, c->st); // pm follows the new order
	hipLaunchKernelGGL(k_cstie_list, dim3(nblk(T)), dim3(BLOCK), 0, c->st, c->recA, d_pos, T, N, c->flags);
	if (partial) TRY(walk_again(d_pos));
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

extern "C" int pga_sync(pga_ctx_t *c) { return sync_st(c); }

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
		TRY(sync_st(c));
		memcpy(dst_host, c->h_fetch, nbytes);
		return 0;
	}
	HIPCHK(hipMemcpyAsync(dst_host, src_backend, nbytes, hipMemcpyDeviceToHost, c->st));
	return sync_st(c);
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
	{ std::lock_guard<std::mutex> lk(g_dev_mu); ++g_dev_reserving; }
	struct Done { ~Done() { { std::lock_guard<std::mutex> lk(g_dev_mu); --g_dev_reserving; } g_dev_cv.notify_all(); } } done;
	for (int k = 1; k >= 0; --k) { // (the larger one first)
		{
			std::lock_guard<std::mutex> lk(g_dev_mu);
			bool have = false;
			for (const DevBlock &b : g_dev_cache) have = have || (b.dev == cur_dev() && b.cap >= want[k] && b.cap <= 2 * want[k] + ((size_t)64 << 20));
			if (have) continue;
		}
		const size_t padded = (want[k] + want[k] / 8 + ((size_t)64 << 20) - 1) & ~(((size_t)64 << 20) - 1);
		void *q = nullptr;
		if (hipMalloc(&q, padded) != hipSuccess) { (void)hipGetLastError(); continue; }
		++made;
		std::lock_guard<std::mutex> lk(g_dev_mu);
		if (g_dev_cache.size() >= 2) { // the cache holds one context's worth: the smallest block that is not the one just asked for makes room
			size_t small = 0;
			for (size_t i = 1; i < g_dev_cache.size(); ++i) if (g_dev_cache[i].cap < g_dev_cache[small].cap) small = i;
			(void)hipFree(g_dev_cache[small].p);
			g_dev_cache.erase(g_dev_cache.begin() + (long)small);
		}
		g_dev_cache.push_back(DevBlock{q, padded, cur_dev()});
	}
	if (timing) { timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1); fprintf(stderr, "[pga_reserve] %lld hits: %.1f + %.1f GB asked for, %d block(s) allocated in %.1f ms\n", (long long)n_hit, want[0] / 1073741824.0, want[1] / 1073741824.0, made, ((t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9) * 1e3); }
	return 0;
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

extern "C" const pga_backend_t *pga_backend(void)
{
	static const pga_backend_t b = {
		"hip-gfx950", pga_create, pga_destroy, pga_begin, pga_ingest, pga_post_partials, pga_post_apply, pga_shadow, pga_set_filter,
		pga_vtx_partials, pga_flag_vtx, pga_arc_round, pga_arc_merge, pga_arc_set_current, pga_rep_pos, pga_n_local, pga_branch_pairs, pga_branch_decide, pga_mark_hits, pga_override_order, pga_set_head, pga_fetch, pga_put, pga_copy, pga_scratch,
		pga_download, pga_hazards, pga_is_device, pga_strerror, pga_timing_reset, pga_timing_get, pga_sync, pga_fetch_later, pga_hazard_segs, pga_host_alloc, pga_host_free, pga_arc_round_local, pga_ctg_counts, pga_gene_matrix, pga_arc_table, pga_arc_round_finish, pga_branch_decide_filter, pga_branch_loop, pga_host_trim, pga_set_device, pga_device_count, pga_arc_round_x, pga_copy_gbps, pga_warm, pga_reserve
	};
	return &b;
}

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
