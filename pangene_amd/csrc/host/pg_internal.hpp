// pg_internal.hpp -- host-side internals shared by the reader, the writers and the round driver.
#pragma once
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <atomic>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>
#include "pangene_amd.h"
#include "pangene_hip.h"

namespace pgx {

// name -> first-seen id map that owns its strings (the reference's dict.c:29-91; ids are insertion
// order, which is what read.c:151-168 relies on).  c_str() pointers stay valid for the dict's life.
// How many threads are worth starting: the hardware threads this process may use, capped by the CPU bandwidth its control group grants.
// (The GPU boxes of this project show 256 hardware threads and grant 16 cores of CPU time -- cpu.max "1600000 100000": a pool of 64
// threads runs no faster than one of 16, it only burns the quota in a quarter of every 100 ms period and then ALL threads of the process,
// the one feeding the GPU included, stand still until the next period; profiles/r04_cpu_scaling_box.txt.)
unsigned host_threads(unsigned cap);
// a switch that takes a comma-separated list of words (PANGENE_LOOP=nopre,nofinal)
inline bool env_word(const char *name, const char *word)
{
	const char *e = std::getenv(name);
	const size_t n = std::strlen(word);
	while (e && *e) { const char *c = std::strchr(e, ','); const size_t len = c ? (size_t)(c - e) : std::strlen(e); if (len == n && std::strncmp(e, word, n) == 0) return true; e = c ? c + 1 : nullptr; }
	return false;
}
bool device_is_up(); // the backend's runtime has been started in this process (graph_driver.cpp)

// Name -> id.  Open addressing over (32-bit hash, id) slots, names in blocks that never move (pg_gene_t / pg_prot_t / pg_ctg_t keep
// `const char *` into them).  A batch read builds three of these per FILE (10 000 names each for a bacterial genome): with
// std::unordered_map + std::deque<std::string> that was a node and often a string allocation per name -- a sixth of the parse.
class FlatIndex { // the table alone: names live elsewhere (NameDict's blocks, or another dictionary: the snapshots of the batch reader)
public:
	static inline uint64_t hash(std::string_view s) {
		uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)s.size();
		const char *p = s.data();
		size_t n = s.size();
		for (; n >= 8; p += 8, n -= 8) { uint64_t w; std::memcpy(&w, p, 8); h = (h ^ w) * 0xff51afd7ed558ccdull; h ^= h >> 29; }
		if (n) { uint64_t w = 0; std::memcpy(&w, p, n); h = (h ^ w) * 0xff51afd7ed558ccdull; h ^= h >> 29; }
		return h ^ (h >> 32);
	}
	int32_t size() const { return (int32_t)ptr_.size(); }
	void reserve(size_t n) { ptr_.reserve(n), len_.reserve(n); size_t cap = 16; while (cap < 2 * n + 2) cap <<= 1; if (cap > tab_.size()) rehash(cap); }
	int32_t find(std::string_view s, uint64_t h) const {
		if (tab_.empty()) return -1;
		const size_t mask = tab_.size() - 1;
		for (size_t k = (size_t)h & mask;; k = (k + 1) & mask) {
			const Slot &t = tab_[k];
			if (t.id < 0) return -1;
			if (t.h == (uint32_t)(h >> 32) && len_[(size_t)t.id] == (uint32_t)s.size() && std::memcmp(ptr_[(size_t)t.id], s.data(), s.size()) == 0) return t.id;
		}
	}
	int32_t find(std::string_view s) const { return find(s, hash(s)); }
	// the caller has looked (find() < 0) and has put the name where it stays: the next id
	int32_t add(const char *stable, size_t len, uint64_t h) {
		if (2 * (ptr_.size() + 1) > tab_.size()) rehash(tab_.empty() ? 16 : tab_.size() * 2);
		const int32_t id = (int32_t)ptr_.size();
		ptr_.push_back(stable), len_.push_back((uint32_t)len), hs_.push_back(h);
		place(h, id);
		return id;
	}
	const char *name(int32_t id) const { return ptr_[(size_t)id]; }
	std::string_view view(int32_t id) const { return std::string_view(ptr_[(size_t)id], len_[(size_t)id]); }
private:
	struct Slot { uint32_t h; int32_t id; };
	void place(uint64_t h, int32_t id) { const size_t mask = tab_.size() - 1; size_t k = (size_t)h & mask; while (tab_[k].id >= 0) k = (k + 1) & mask; tab_[k] = Slot{(uint32_t)(h >> 32), id}; }
	void rehash(size_t cap) { tab_.assign(cap, Slot{0, -1}); for (size_t i = 0; i < ptr_.size(); ++i) place(hs_[i], (int32_t)i); }
	std::vector<Slot> tab_;
	std::vector<const char *> ptr_; std::vector<uint32_t> len_; std::vector<uint64_t> hs_;
};

class NameDict {
public:
	int32_t size() const { return ix_.size(); }
	int32_t get(std::string_view s) const { return ix_.find(s); }
	// returns id; *absent tells whether the name was new
	int32_t put(std::string_view s, bool *absent) {
		const uint64_t h = FlatIndex::hash(s);
		const int32_t id = ix_.find(s, h);
		if (id >= 0) { if (absent) *absent = false; return id; }
		if (absent) *absent = true;
		return ix_.add(keep(s), s.size(), h);
	}
	const char *name(int32_t id) const { return ix_.name(id); }          // NUL-terminated, never moves
	std::string_view view(int32_t id) const { return ix_.view(id); }
private:
	const char *keep(std::string_view s) {
		if (s.size() + 1 > left_) {
			const size_t b = std::max<size_t>(s.size() + 1, (size_t)64 << 10);
			blocks_.emplace_back(new char[b]);
			at_ = blocks_.back().get(), left_ = b;
		}
		char *p = at_;
		std::memcpy(p, s.data(), s.size()), p[s.size()] = 0;
		at_ += s.size() + 1, left_ -= s.size() + 1;
		return p;
	}
	FlatIndex ix_;
	std::vector<std::unique_ptr<char[]>> blocks_;
	char *at_ = nullptr; size_t left_ = 0;
};

// one contig segment whose exact (reference, unstable-sort) order is replayed on the host, exact_order.cpp
struct ExactSeg {
	int32_t k = 0, start = 0;             // local genome index; offset of the contig inside the genome
	std::vector<int32_t> file;            // file index of each hit of the contig, in file order
	std::vector<uint64_t> cs, cm;         // sort keys, aligned with `file`
	std::vector<std::vector<int32_t>> hx, hy; // mode all: the cs / cm orders X_1.., Y_1.. until the sequence becomes periodic
	std::vector<int32_t> heads;           // file index of the hit at array index 0 of X_1, X_2, ...
	int cyc_start = -1, period = 0;       // X_t == X_{cyc_start + (t - cyc_start) % period} for t >= cyc_start (1-based)
	// Which order the backend currently holds for cs (0) and cm (1): the IDENTITY of a stored order, -1 = none of them (the backend's
	// own).  xid[i] / yid[i] = the first stored order equal to hx[i] / hy[i] (replay() compares them once), so that "has this order
	// been handed over already?" is a comparison of two numbers.  Comparing the orders themselves at each of the 67 sorts of a pass
	// cost 110 of the 163 ms of a pass over 200 isoform-rich assemblies (a million hits on tracked contigs) and 4.7 of 22.5 ms on the
	// human-shaped shard.
	int32_t pushed_id[2] = { -1, -1 };
	std::vector<int32_t> xid, yid;
	bool full = false;                    // every order of the contig is replayed and handed over (mode all, or a contig on which a tie hazard was seen)
};

// one genome packed for the backend (pga_genome_block_t) as soon as its PAF has been parsed.  The block is carved out of a
// slab of pinned host memory (DataExt::slabs); slabs go back to a process-wide cache after the upload.
struct GenomePack {
	pga_genome_block_t blk{};
	void *buf = nullptr;
	int err = 0;                              // PGA_ERR_RANGE: a coordinate does not fit the device layout
	uint64_t sig = 0;                         // what the genome looked like when it was packed (graph_driver.cpp: genome_signature)
	// virtual contigs (pga_genome_block_t): empty unless a contig of the genome had to be cut
	std::vector<int32_t> vfirst, vreal;       // per piece: first piece of its contig; the contig's own id (pg_hit_t::cid)
	std::vector<int64_t> vbase;               // per piece: the base its coordinates are relative to
};
struct HostSlab { char *p = nullptr; size_t cap = 0, off = 0; bool pinned = false, fresh = false; // fresh: page-locked for this very read (not taken from the cache)
	int pending = 0; bool closed = false, staged = false; }; // blocks handed out and not yet written; no more blocks will come; its bytes are on their way to the device (pga_stage_h2d)

// host-private companion of a pg_data_t (struct layout of pg_data_t itself must not change)
struct DataExt {
	std::vector<uint8_t> is_local;     // per genome: hits live in this process
	// Memory of a batch read that the genomes' hit / exon arrays point into: ONE mapping on huge pages, carved by the parser threads
	// (paf_reader.cpp).  A hundred threads filling malloc'ed arrays take 4 KiB page faults at the rate ONE address space sustains
	// (measured on the 256-thread GPU box: 15 GB/s of fresh memory whatever the thread count, 120-140 GB/s on huge pages;
	// profiles/r04_pagefault_box.txt).  Arrays inside are not free()'d one by one: arena_owns() tells, ext_drop unmaps.
	struct HostArena { char *map = nullptr; size_t map_bytes = 0; char *base = nullptr; size_t bytes = 0; std::atomic<size_t> used{0}; };
	std::deque<HostArena> arenas;
	bool arena_owns(const void *p) const { for (const HostArena &a : arenas) if ((const char *)p >= a.base && (const char *)p < a.base + a.bytes) return true; return false; }
	std::vector<uint8_t> hits_sorted;  // per genome: host AoS already in X (cs) order
	const pga_backend_t *be = nullptr;
	pga_ctx_t *ctx = nullptr;          // backend context (owns the HBM-resident shard)
	std::vector<int32_t> local_genomes; // global index of each genome in the shard
	std::vector<int64_t> hit_off;      // shard hit offsets
	std::vector<std::vector<int32_t>> vreal; // per local genome: piece -> contig id, for genomes with virtual contigs (empty: the backend's contigs are the genome's)
	std::vector<int32_t> n_vctg;       // per local genome: contigs as the backend counts them (pieces)
	std::vector<std::vector<int32_t>> y_file;  // per genome: FILE index of the k-th hit in cm order
	std::vector<ExactSeg> xsegs;
	std::vector<int32_t> deg;          // out-degree of every oriented vertex of the round's arc table
	std::vector<int32_t> sc_buf;       // landing area of a round's segment counters (arc_collect)
	std::vector<uint8_t> del_buf;      // per-segment verdicts of branch_decide_filter
	const pga_arc_part_t *cur_arcs = nullptr; // the round's arc table, in backend memory
	std::string vtx_sel_text;          // -G output of the current run (printed when pg_graph_gen returns)
	int exact_mode_of_segs = -1;       // mode xsegs was built for
	const pg_data_t *q_d = nullptr;    // the data set the context belongs to
	std::vector<std::pair<int32_t, int32_t>> extra_ctgs; // (local genome, contig) pairs that get the full exact order although the mode is auto (sorted)
	std::vector<std::pair<int32_t, int32_t>> static_ctgs; // (local genome, contig) pairs tracked in full because the static keys predict a tie channel there (exact_order.cpp: static_tie_contigs); sorted
	bool check_strand = false;         // PG_F_CHECK_STRAND and min_ov_ratio of the options the tracked contigs were chosen with (they enter the static prediction)
	double min_ov_ratio = 0.5;
	std::vector<std::thread> xworkers; // background replay of the reference's sort sequence
	std::atomic<size_t> xnext{0};
	bool xreplayed = false;            // the segments' sort sequences have been (or are being) replayed
	int32_t xsegs_n_genome = -1;       // number of genomes the segments were built for
	uint64_t xsegs_gen = 0;            // counts the sets of segments built (exact_init): order identities mean something inside one set only
	int x_sorts[2] = {0, 0};           // cs / cm sorts of the reference seen so far in this run
	std::vector<int32_t> head_file;    // per local genome: file index of the hit at array index 0 (-1 canonical)
	std::vector<int32_t> seg_renumber; // branch rounds queued to the end: old segment number -> number in the graph that is written (-1: deleted); empty otherwise
	int route_verbose = -1;            // sharded runs: the log level every rank's ROUTING decisions follow (max over the ranks, agreed at upload time); -1 = pg_verbose
	bool skip_loop_once = false;       // sharded pga_branch_loop asked for a repeated run because an exchange buffer was too small (status 3): that run is host-driven, later ones queue again
	int64_t x_arc_slot = 0;            // sharded runs: the largest local arc table of any host-driven round so far, over all ranks
	int loop_room_retries = 0;         // status 4 of pga_branch_loop on this data set (a pair list beyond its capacity: the run is repeated with room; after a few of them the rounds are host-driven)
	bool no_branch_loop = false;       // the queued branch rounds (pga_branch_loop) met something they cannot handle on this data set: host-driven rounds from now on
	bool arc_via_x = false;            // (sharded) the last pg_gen_arc came through pga_arc_round_x: the backend holds the GLOBAL segment counters and the merged table's degrees
	bool arc_pending = false;          // a deferred arc round whose host results have not been collected (arc_collect)
	bool rerun = false;                // pg_rerun_resident(): keep the backend context, skip pack + upload
	bool read_failed = false;          // a pg_read_paf ran out of memory half way through a file: the data set is not what the files hold
	bool host_full = false;            // the last sync also fetched rank / score_dom / dominators
	bool pos_valid = false;            // pos_x / y_order on the host match the backend's current orders
	bool order_touched = false;        // an order override has been handed to the backend since the last sync (exact_sort)
	int64_t ov_calls = 0, ov_hits = 0; double ov_list_s = 0, ov_backend_s = 0; // (PANGENE_TIMING) order overrides of the run: how many, how many hits, where the time went
	uint64_t pos_sig = 0;              // order_signature() of the orders pos_x / y_file were fetched from
	std::vector<uint64_t> flt_bits;    // bit (shard hit offset of the genome + host index) = flt, refreshed by every sync
	std::vector<int32_t> pos_x;        // per local hit (file order): position inside its genome in cs order
	std::vector<std::vector<int32_t>> file_of_host, host_of_file; // per genome whose records were moved into cs order (hits_sorted): host array index <-> file index
	bool host_order_valid = false;     // the moved records follow the backend's CURRENT cs order
	bool host_stale = false;           // per-hit flags on the host are older than the backend's
	std::vector<GenomePack> packs;     // per genome (global index); empty buf = not packed (any more)
	double pack_sec = 0.0;             // wall seconds spent packing (reader threads) since the last upload
	std::vector<HostSlab> slabs;       // pinned memory the blocks live in (guarded by slab_mu while reader threads pack)
	std::mutex slab_mu;
	int64_t n_hit_local = 0;
};

DataExt *ext_of(const pg_data_t *d, bool create);
// pack the genomes [j0, j1) that have no pack yet (host threads); called by the reader after the commit and by the driver as a fallback
void pack_genomes(const pg_data_t *d, DataExt *ext, int32_t j0, int32_t j1, double time_share = 1.0, bool verify = true); // time_share: fraction of the call's wall time booked as packing time (calls that run side by side on n threads: 1 / n)
void trim_host_caches(size_t keep_bytes);
void slab_prefetch(size_t bytes, std::thread *helper); // page-lock about `bytes` of block memory ahead, on a helper thread the caller joins
void free_packs(DataExt *ext, bool wait);
void ext_drop(const pg_data_t *d);

const pga_backend_t *backend_default();   // link-time selected (HIP in the product)

extern pg_exchange_t g_xchg; extern bool g_has_xchg;
void set_error(int code, const char *where);

FILE *out_stream();

double now_sec();
const char *stamp();

// bring the host AoS (flags, rank, dominators, order) up to date with the backend
int sync_host(pg_data_t *d, bool full);

int exact_mode();
void exact_override(int m);
void exact_init(const pg_data_t *d, DataExt *ext);
void exact_begin(DataExt *ext);
void exact_prefetch(const pg_data_t *d, DataExt *ext);
int exact_sort(DataExt *ext, int by_cm);
void exact_shutdown(DataExt *ext);
bool exact_quiet(DataExt *ext, int n);
void exact_skip(DataExt *ext, int n);
uint64_t order_signature(const DataExt *ext);

// phase accounting of the host driver (seconds, accumulated over the last run)
enum { PH_BEGIN, PH_EXACT, PH_INGEST, PH_POST, PH_VTX, PH_ARC_DEV, PH_ARC_HOST, PH_BRANCH_HOST, PH_NLOCAL, PH_MARK_HITS, PH_FLT, PH_SYNC_HOST, PH_COUNT };
extern double g_phase[PH_COUNT];
struct Phase { int id; double t0; explicit Phase(int i) : id(i), t0(now_sec()) {} ~Phase() { g_phase[id] += now_sec() - t0; } };

} // namespace pgx
