// graph_driver.cpp -- host side of the graph-construction path: packs the parsed hits into one
// structure-of-arrays shard, hands it to the backend (HIP kernels; include/pangene_hip.h) and drives
// the rounds of the reference's pg_post_process (graph.c:7-32) and pg_graph_gen (graph.c:280-322).
// Everything hit-sized happens behind the backend ABI; this file only touches P-, Q-, S- and A-sized
// data (proteins, genes, segments, arcs): the representative-isoform sort (hit.c:205-217), the greedy
// vertex selection (vertex.c:54-97), branch marking (branch.c:48-106), pruning (graph.c:179-263).
// With an exchange hook installed (pg_set_exchange) the same code runs on every rank of a sharded run:
// partial vectors are all-reduced / all-gathered in backend memory (RCCL) and every rank then takes
// the identical host-side decisions, so no broadcast is needed.
#include <sys/mman.h>
#include <sys/time.h>
#include <sys/resource.h>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <climits>
#include <condition_variable>
#include <mutex>
#include <thread>
#include "pg_internal.hpp"
#include "ksort_exact.hpp"

int pg_verbose = 3;

namespace pgx {

pg_exchange_t g_xchg; bool g_has_xchg = false;
static int64_t g_n_coll = 0; // collectives issued so far (pg_collective_count)
double g_phase[PH_COUNT];
static int g_err = 0; static char g_errstr[256] = "";
static double g_path_sec = 0.0, g_upload_sec = 0.0, g_pack_sec = 0.0, g_t_path0 = 0.0; static int64_t g_path_hits = 0; static int g_attempts = 0;

void set_error(int code, const char *where)
{
	g_err = code;
	std::snprintf(g_errstr, sizeof(g_errstr), "%s: backend status %d", where, code);
	std::fprintf(stderr, "[E::pangene_amd] %s\n", g_errstr);
}

double now_sec()
{
	struct timeval tp;
	gettimeofday(&tp, nullptr);
	return tp.tv_sec + tp.tv_usec * 1e-6;
}

const char *stamp() // "<real>*<cpu/real>", the reference's pg_timestamp (sys.c:131-138)
{
	static char buf[64];
	static double t0 = -1.0;
	double t = now_sec();
	if (t0 < 0) t0 = t;
	struct rusage r;
	getrusage(RUSAGE_SELF, &r);
	double cpu = r.ru_utime.tv_sec + r.ru_stime.tv_sec + 1e-6 * (r.ru_utime.tv_usec + r.ru_stime.tv_usec);
	std::snprintf(buf, sizeof(buf), "%.3f*%.2f", t - t0, (cpu + 1e-6) / (t - t0 + 1e-6));
	return buf;
}

// PANGENE_FORCE_EXCHANGE=1 routes a single-process run through the exchange callbacks too (lets one GPU
// exercise the RCCL plumbing; results are unchanged because every collective is then an identity)
static inline bool sharded()
{
	static int force = -1;
	if (force < 0) { const char *e = std::getenv("PANGENE_FORCE_EXCHANGE"); force = e && *e == '1'; }
	return g_has_xchg && (g_xchg.world > 1 || force);
}

// Which ROUTE a step takes (queued rounds or host-driven ones, one slot all-gather or the three-step exchange, ...) must never
// depend on a per-rank setting: the collectives of the ranks would not match.  The log level picks routes (the verbose ones fetch
// counts for their log lines), so a sharded run agrees on ONE level -- the maximum over the ranks, all-reduced once per upload --
// and every routing decision reads that; what a rank PRINTS still follows its own pg_verbose.
static inline int route_v(const DataExt *ext) { return (sharded() && ext && ext->route_verbose >= 0) ? ext->route_verbose : pg_verbose; }

#define BE_CALL(expr, where) do { int rc__ = (expr); if (rc__ != 0) { set_error(rc__, where); return rc__; } } while (0)

// the exchange callbacks run on the backend's stream (built-in RCCL) or somewhere else (torch.distributed, gloo): in the
// second case the backend has to finish what it was asked to do first
static inline int xready(const pga_backend_t *be, pga_ctx_t *ctx) { return g_xchg.stream_ordered ? 0 : be->sync(ctx); }

static int xreduce(const pga_backend_t *be, pga_ctx_t *ctx, void *buf, int64_t count, int32_t dtype, int32_t op)
{
	if (!sharded() || count == 0) return 0;
	BE_CALL(xready(be, ctx), "sync");
	++g_n_coll;
	return g_xchg.allreduce(g_xchg.user, buf, count, dtype, op, be->is_device());
}

// all-gather of a variable-length array living in backend memory -> host vector (rank order)
template <class T>
static int xgather(const pga_backend_t *be, pga_ctx_t *ctx, const T *local, int64_t n, std::vector<T> &out, bool local_on_host = false)
{
	if (!sharded()) {
		out.resize((size_t)n);
		if (local_on_host) { if (n) std::memcpy(out.data(), local, (size_t)n * sizeof(T)); return 0; }
		return n ? be->fetch(ctx, out.data(), local, (size_t)n * sizeof(T)) : 0;
	}
	const int W = g_xchg.world;
	void *scr;
	std::vector<int64_t> cnt((size_t)W);
	BE_CALL(be->scratch(ctx, sizeof(int64_t) * (size_t)(W + 1), &scr), "scratch");
	BE_CALL(be->put(ctx, scr, &n, sizeof(int64_t)), "put");
	BE_CALL(xready(be, ctx), "sync");
	g_n_coll += 2;
	BE_CALL(g_xchg.allgather(g_xchg.user, scr, (char *)scr + sizeof(int64_t), sizeof(int64_t), be->is_device()), "allgather(count)");
	BE_CALL(be->fetch(ctx, cnt.data(), (char *)scr + sizeof(int64_t), sizeof(int64_t) * (size_t)W), "fetch");
	int64_t mx = *std::max_element(cnt.begin(), cnt.end()), tot = 0;
	for (int64_t c : cnt) tot += c;
	out.clear();
	if (mx == 0) return 0;
	size_t slot = (size_t)mx * sizeof(T);
	BE_CALL(be->scratch(ctx, slot * (size_t)(W + 1), &scr), "scratch");
	if (n) BE_CALL(local_on_host ? be->put(ctx, scr, local, (size_t)n * sizeof(T)) : be->copy(ctx, scr, local, (size_t)n * sizeof(T)), "copy");
	BE_CALL(xready(be, ctx), "sync");
	BE_CALL(g_xchg.allgather(g_xchg.user, scr, (char *)scr + slot, (int64_t)slot, be->is_device()), "allgather(data)");
	std::vector<T> all((size_t)mx * (size_t)W);
	BE_CALL(be->fetch(ctx, all.data(), (char *)scr + slot, slot * (size_t)W), "fetch");
	out.reserve((size_t)tot);
	for (int r = 0; r < W; ++r)
		out.insert(out.end(), all.begin() + (size_t)r * (size_t)mx, all.begin() + (size_t)r * (size_t)mx + (size_t)cnt[(size_t)r]);
	return 0;
}

// ---------------------------------------------------------------------------------------------
// shard packing: host AoS (pg_hit_t, 88 B) -> one structure-of-arrays block per genome (pga_genome_block_t), FILE order.
// Done by the reader as soon as a genome has been parsed (host threads, pinned memory), so that pg_post_process only has
// to hand the blocks to the backend: the upload is then plain DMA out of pinned memory.
// ---------------------------------------------------------------------------------------------
static void *block_alloc(DataExt *ext, size_t bytes);
static void block_done(DataExt *ext, const void *buf);
// (the signature of a genome's records: what it is for is said above stale_packs)
struct SigState { uint64_t h[4]; };
static inline void sig_begin(SigState &s, const pg_genome_t *g)
{
	s.h[0] = 1469598103934665603ull ^ (uint64_t)(uint32_t)g->n_hit, s.h[1] = 0x9e3779b97f4a7c15ull ^ (uint64_t)(uint32_t)g->n_exon, s.h[2] = 0xc2b2ae3d27d4eb4full ^ (uint64_t)(uint32_t)g->n_ctg, s.h[3] = 0x165667b19e3779f9ull;
}
static inline void sig_hit(SigState &s, int64_t i, const pg_hit_t &a)
{
	uint64_t &x = s.h[i & 3];
	x = (x ^ ((uint64_t)(uint32_t)a.pid << 32 | (uint32_t)a.cid)) * 1099511628211ull;
	x = (x ^ (uint64_t)a.cs) * 1099511628211ull; // (each 64-bit coordinate through a multiply of its own: shifted into one word they aliased -- cs bit 21 with ce bit 0)
	x = (x ^ (uint64_t)a.ce) * 0x9e3779b97f4a7c15ull;
	x = (x ^ (uint64_t)a.cm) * 1099511628211ull;
	x = (x ^ ((uint64_t)(uint32_t)a.score_adj << 32 | (uint32_t)a.score_ori)) * 1099511628211ull;
	x = (x ^ ((uint64_t)(uint32_t)a.off_exon << 32 | (uint32_t)a.n_exon << 8 | (uint32_t)a.rev << 7 | ((uint32_t)(a.rank & 0x7f) ^ (uint64_t)(uint32_t)a.rank << 40))) * 1099511628211ull;
}
static inline void sig_exon(SigState &s, int64_t i, const pg_exon_t &e) { uint64_t &x = s.h[i & 3]; x = (x ^ ((uint64_t)(uint32_t)e.os << 32 | (uint32_t)e.oe)) * 0x100000001b3ull; }
static inline uint64_t sig_end(const SigState &s) { return (s.h[0] * 31 + s.h[1]) * 31 + (s.h[2] * 31 + s.h[3]); }
static uint64_t genome_signature(const pg_genome_t *g)
{
	SigState s;
	sig_begin(s, g);
	for (int32_t i = 0; i < g->n_hit; ++i) sig_hit(s, i, g->hit[i]);
	for (int32_t i = 0; i < g->n_exon; ++i) sig_exon(s, i, g->exon[i]);
	return sig_end(s);
}


// Contigs whose coordinates do not fit 31 bits (pangene.h:71 has int64_t; the reference handles any length) reach the backend as VIRTUAL
// contigs (pga_genome_block_t): cut at hit-free gaps into pieces of < 2^30 bp (+ the cluster of overlapping hits the cut has to wait
// for), coordinates relative to the piece.  piece_of[h] = piece of hit h; returns false when a single cluster spans too much.
static bool virtual_contigs(const pg_genome_t *g, std::vector<int32_t> &piece_of, std::vector<int32_t> &vfirst, std::vector<int32_t> &vreal, std::vector<int64_t> &vbase)
{
	static const int64_t PIECE = [] { const char *e = std::getenv("PANGENE_VCTG_PIECE"); return e && std::atoll(e) > 0 ? std::atoll(e) : (int64_t)1 << 30; }(); // (tests shrink it to cut ordinary contigs)
	const int32_t n = g->n_hit;
	std::vector<int32_t> ord((size_t)n);
	for (int32_t i = 0; i < n; ++i) ord[(size_t)i] = i;
	std::sort(ord.begin(), ord.end(), [g](int32_t x, int32_t y) { const pg_hit_t &a = g->hit[x], &b = g->hit[y]; return a.cid != b.cid ? a.cid < b.cid : a.cs != b.cs ? a.cs < b.cs : x < y; });
	piece_of.assign((size_t)n, 0), vfirst.clear(), vreal.clear(), vbase.clear();
	int32_t k = 0;
	for (int32_t c = 0; c < g->n_ctg; ++c) { // (contigs without hits keep one piece: the numbering of the others must not depend on them... it does not: pieces are numbered in contig order)
		const int32_t first = (int32_t)vfirst.size();
		int64_t base = 0, reach = -1; // reach: the largest ce so far = where the current cluster of overlapping hits ends
		bool open = false;
		vfirst.push_back(first), vreal.push_back(c), vbase.push_back(0);
		for (; k < n && g->hit[ord[(size_t)k]].cid == c; ++k) {
			const pg_hit_t &a = g->hit[ord[(size_t)k]];
			if (!open) { base = a.cs >= INT32_MAX / 2 ? a.cs : 0, vbase.back() = base, open = true; } // (a contig that fits keeps its coordinates)
			else if (a.cs > reach && a.cs - base >= PIECE) { base = a.cs; vfirst.push_back(first), vreal.push_back(c), vbase.push_back(base); } // a gap (strictly: no cm of the next piece can equal one of this piece), and the piece is long enough: cut
			if (a.ce - base >= INT32_MAX - 1) return false;
			reach = std::max(reach, a.ce);
			piece_of[(size_t)ord[(size_t)k]] = (int32_t)vfirst.size() - 1;
		}
	}
	return true;
}

static void pack_one(const pg_data_t *d, DataExt *ext, int32_t j)
{
	GenomePack &pk = ext->packs[(size_t)j];
	const pg_genome_t *g = &d->genome[j];
	const int64_t n = g->n_hit, ne = g->n_exon;
	const size_t nw = (size_t)(PGA_BLOCK_PLANES * n + (n + 3) / 4 + 2 * ne);
	pk = GenomePack();
	pk.buf = block_alloc(ext, (nw ? nw : 1) * sizeof(int32_t));
	int32_t *w = (int32_t *)pk.buf;
	uint8_t *rev = (uint8_t *)(w + PGA_BLOCK_PLANES * n);
	int32_t *ex = w + PGA_BLOCK_PLANES * n + (n + 3) / 4;
	// the host array is in file order until the first sync_host and in cs order afterwards (file_of_host: host index -> file index)
	const bool sorted = (size_t)j < ext->hits_sorted.size() && ext->hits_sorted[(size_t)j] && (size_t)j < ext->file_of_host.size();
	const int32_t *fof = sorted ? ext->file_of_host[(size_t)j].data() : nullptr;
	int32_t max_cs = 0, max_cm = 0, max_sadj = 0, neg = 0, multi = 0;
	// 64-bit coordinates: does any hit of the genome need a virtual contig?
	// (round 6: ONE pass over the records -- the test for coordinates beyond 31 bits and the signature ride with the copy; a genome that does
	// need virtual contigs is found out on the way and starts again as such)
	static const bool force_v = std::getenv("PANGENE_VCTG_PIECE") != nullptr;
	bool wide = force_v;
	std::vector<int32_t> piece_of;
	SigState sig;
again:
	sig_begin(sig, g);
	max_cs = 0, max_cm = 0, max_sadj = 0, neg = 0, multi = 0;
	if (wide) {
		bool ok = true;
		for (int64_t h = 0; h < n; ++h) { const pg_hit_t *a = &g->hit[h]; if (a->cs < 0 || a->cm < a->cs || a->ce < a->cs || a->cid < 0 || a->cid >= g->n_ctg) ok = false; }
		if (!ok || !virtual_contigs(g, piece_of, pk.vfirst, pk.vreal, pk.vbase)) { pk.err = PGA_ERR_RANGE; wide = false; }
	}
	for (int64_t h = 0; h < n; ++h) {
		const pg_hit_t *a = &g->hit[h];
		const int64_t f = fof ? fof[h] : h;
		if (!sorted) sig_hit(sig, h, *a);
		if (pk.err) continue;
		if (!wide && (a->ce >= INT32_MAX - 1 || a->cm >= INT32_MAX - 1)) { wide = true; goto again; }
		int32_t cid = a->cid;
		int64_t cs = a->cs, ce = a->ce, cm = a->cm;
		if (wide) { cid = piece_of[(size_t)h]; const int64_t base = pk.vbase[(size_t)cid]; cs -= base, ce -= base, cm -= base; }
		if (cs < 0 || ce >= INT32_MAX || cm < 0 || cm >= INT32_MAX || ce < cs || cid < 0) { pk.err = PGA_ERR_RANGE; continue; }
		w[f] = a->pid, w[n + f] = cid, w[2 * n + f] = a->rank, w[3 * n + f] = a->score_ori, w[4 * n + f] = a->score_adj;
		w[5 * n + f] = a->n_exon, w[6 * n + f] = a->off_exon, w[7 * n + f] = (int32_t)cs, w[8 * n + f] = (int32_t)ce, w[9 * n + f] = (int32_t)cm;
		rev[f] = a->rev;
		max_cs = std::max(max_cs, (int32_t)cs), max_cm = std::max(max_cm, (int32_t)cm);
		if (a->score_adj < 0) neg = 1; else max_sadj = std::max(max_sadj, a->score_adj);
		multi |= a->n_exon != 1;
	}
	for (int64_t e = 0; e < ne; ++e) { ex[2 * e] = g->exon[e].os, ex[2 * e + 1] = g->exon[e].oe; if (!sorted) sig_exon(sig, e, g->exon[e]); }
	pga_genome_block_t &b = pk.blk;
	b.n_hit = (int32_t)n, b.n_exon = (int32_t)ne, b.n_ctg = wide ? (int32_t)pk.vfirst.size() : g->n_ctg;
	b.max_cs = max_cs, b.max_cm = max_cm, b.max_score_adj = max_sadj, b.any_neg_score_adj = neg, b.any_multi_exon = multi;
	b.data = w, b.n_words = nw;
	b.vfirst = wide ? pk.vfirst.data() : nullptr, b.vbase = wide ? pk.vbase.data() : nullptr;
	if (!wide) pk.vfirst.clear(), pk.vreal.clear(), pk.vbase.clear();
	pk.sig = sorted ? 0 : sig_end(sig);
	block_done(ext, pk.buf);
}

// A pack made when the genome was read is only good while the genome still is what it was then.  The public pg_data_t may be
// edited between pg_read_paf and pg_post_process (the reference reads g->hit at post-process time, graph.c:7-32): a changed hit or
// exon count, or a change in ANY field of ANY record that the pack carries, makes the block stale and it is packed again.  (Round 4
// sampled every 257th record; one pass over the records costs a fraction of the pack it may save -- four independent multiply chains
// keep it at memory speed.  A 64-bit hash, not a proof: an edit is missed with probability ~2^-64; genomes whose records a sync_host has
// already moved into cs order are not compared -- their pack was made from the order the signature was taken in.)
// which packs of [j0, j1) no longer match their genome?  The signatures read every record once (88 bytes a hit: 0.1 s for 12 M hits on one core), so the genomes are shared out
static void stale_packs(const pg_data_t *d, const DataExt *ext, int32_t j0, int32_t j1, std::vector<int32_t> &out, const std::vector<uint8_t> *only = nullptr)
{
	std::vector<int32_t> have;
	int64_t hh = 0;
	for (int32_t j = j0; j < j1 && (size_t)j < ext->packs.size(); ++j) {
		if (only && !(*only)[(size_t)j]) continue;
		if ((size_t)j < ext->is_local.size() && !ext->is_local[(size_t)j]) continue;
		if (ext->packs[(size_t)j].buf != nullptr) have.push_back(j), hh += d->genome[j].n_hit;
	}
	std::vector<uint8_t> stale(have.size(), 0);
	auto check = [&](size_t i) {
		const int32_t j = have[i];
		const GenomePack &pk = ext->packs[(size_t)j];
		const bool sorted = (size_t)j < ext->hits_sorted.size() && ext->hits_sorted[(size_t)j]; // (records moved into cs order by a sync_host: the signature was taken in file order)
		stale[i] = pk.blk.n_hit != d->genome[j].n_hit || pk.blk.n_exon != d->genome[j].n_exon || (!sorted && pk.sig != genome_signature(&d->genome[j]));
	};
	unsigned nc = hh > 200000 ? std::min<unsigned>(host_threads(64u), (unsigned)(hh / 40000)) : 1u; // (a thread costs ~15 us to start and hashes a hit in ~5 ns: 64 of them for a million hits spend their time being started)
	if (nc > have.size()) nc = (unsigned)have.size();
	std::vector<std::thread> th;
	std::atomic<size_t> next{0};
	auto work = [&]() { for (;;) { const size_t i = next.fetch_add(1); if (i >= have.size()) break; check(i); } };
	try { for (unsigned t = 1; t < nc; ++t) th.emplace_back(work); } catch (const std::system_error &) {} // (at the thread limit: with the threads there are)
	work();
	for (auto &x : th) x.join();
	out.clear();
	for (size_t i = 0; i < have.size(); ++i) if (stale[i]) out.push_back(have[i]);
}

void pack_genomes(const pg_data_t *d, DataExt *ext, int32_t j0, int32_t j1, double time_share, bool verify)
{
	const double t0 = now_sec();
	if (ext->packs.size() < (size_t)d->n_genome) ext->packs.resize((size_t)d->n_genome); // (batch reads reserve the entries before their threads start)
	std::vector<int32_t> todo;
	int64_t hits = 0;
	if (verify) { // which packs still stand?
		std::vector<int32_t> st;
		stale_packs(d, ext, j0, j1, st);
		for (int32_t j : st) ext->packs[(size_t)j] = GenomePack(); // stale: the slab keeps the old bytes until the upload is over
	}
	for (int32_t j = j0; j < j1; ++j) {
		if ((size_t)j < ext->is_local.size() && !ext->is_local[(size_t)j]) continue;
		if (ext->packs[(size_t)j].buf == nullptr) todo.push_back(j), hits += d->genome[j].n_hit;
	}
	unsigned nt = hits > 100000 ? host_threads(64u) : 1u;
	if (nt > todo.size()) nt = (unsigned)todo.size();
	if (nt <= 1) { for (int32_t j : todo) pack_one(d, ext, j); }
	else {
		std::atomic<size_t> next{0};
		std::vector<std::thread> th;
		for (unsigned t = 0; t < nt; ++t)
			th.emplace_back([&]() { for (;;) { const size_t i = next.fetch_add(1); if (i >= todo.size()) break; pack_one(d, ext, todo[i]); } });
		for (auto &x : th) x.join();
	}
	std::lock_guard<std::mutex> lk(ext->slab_mu);
	ext->pack_sec += (now_sec() - t0) * time_share;
}

// Pinned host memory is expensive to get and to give back (every hipHostMalloc / hipHostFree maps or unmaps pages and takes the
// runtime's lock: 0.1-0.3 ms a call): the blocks are carved out of a few large slabs, and the slabs of a finished upload go into
// a process-wide cache for the next data set instead of back to the driver (bounded: what exceeds the bound is released).
static std::mutex g_slab_mu;
static std::vector<HostSlab> g_slab_cache;
static size_t g_slab_cached = 0;
static const size_t SLAB_BYTES = [] { const char *e = std::getenv("PANGENE_SLAB_MB"); return (size_t)(e && std::atoi(e) > 0 ? std::atoi(e) : 64) << 20; }(); // (tests shrink it: slabs that fill up -- and are staged -- on small data sets)
static const size_t SLAB_CACHE_MAX = (size_t)4 << 30;   // while a data set is alive (its upload and its downloads reuse the slabs)
// (when no data set is left a process keeps 256 MiB page-locked: ext_drop, paf_reader.cpp; pg_trim_host_cache(0) releases that too)

// Page-locking memory is slow (a few GB/s) and serial: a batch read knows roughly how much block memory its files will need, so a
// helper thread locks the slabs ahead while the files are parsed (slab_prefetch); a packer that finds the cache empty while
// the helper is still at it waits for the next slab instead of locking one more itself.
static std::condition_variable g_slab_cv;
static int g_prefetch_left = 0; // slabs the helper has still to deliver (guarded by g_slab_mu)

// Page-locked block memory only once the device runtime is up in this process (after the first upload): the runtime's start-up,
// triggered by a first hipHostMalloc in the middle of a batch read, maps and registers memory for ~0.3 s and stalls the page faults
// of every parser thread meanwhile (a `pangene` command parsed its 100 files in 0.30 s instead of 0.05 s).
static std::atomic<bool> g_device_up{false};
bool device_is_up() { return g_device_up.load(); }

static HostSlab slab_get(size_t min_bytes, bool allow_pin = true)
{
	allow_pin = allow_pin && g_device_up.load();
	{
		std::unique_lock<std::mutex> lk(g_slab_mu);
		for (;;) {
			for (size_t i = 0; i < g_slab_cache.size(); ++i)
				if (g_slab_cache[i].cap >= min_bytes) {
					HostSlab s = g_slab_cache[i];
					g_slab_cache.erase(g_slab_cache.begin() + (long)i);
					g_slab_cached -= s.cap, s.off = 0, s.fresh = false, s.pending = 0, s.closed = false, s.staged = false;
					return s;
				}
			if (g_prefetch_left <= 0 || min_bytes > SLAB_BYTES) break;
			g_slab_cv.wait(lk);
		}
	}
	HostSlab s;
	s.cap = std::max(min_bytes, SLAB_BYTES);
	const pga_backend_t *be = backend_default();
	void *q = nullptr;
	if (allow_pin && be->host_alloc && be->host_alloc(s.cap, &q) == 0) s.pinned = true;
	else { // beyond the budget of freshly page-locked memory (block_alloc), or no device (yet): plain pages -- huge ones where the kernel hands them out on request (64 MiB = 32 faults instead of 16 384 while the packer threads fill the slab)
		void *m = nullptr;
		if (posix_memalign(&m, (size_t)2 << 20, s.cap) == 0) { (void)madvise(m, s.cap, MADV_HUGEPAGE); q = m; }
		else q = std::malloc(s.cap);
		s.pinned = false;
	}
	s.p = (char *)q;
	s.fresh = true;
	return s;
}

void slab_prefetch(size_t bytes, std::thread *helper) // *helper is joined by the caller when its reads are done
{
	const pga_backend_t *be = backend_default();
	if (be->host_alloc == nullptr || !be->is_device() || !g_device_up.load()) return;
	{ const char *e = std::getenv("PANGENE_PIN_BUDGET_MB"); if (e && std::atoi(e) <= 0) return; }
	size_t have = 0;
	int n = 0;
	{
		std::lock_guard<std::mutex> lk(g_slab_mu);
		for (const HostSlab &c : g_slab_cache) have += c.cap;
		if (bytes > have) n = (int)std::min<size_t>((bytes - have + SLAB_BYTES - 1) / SLAB_BYTES, SLAB_CACHE_MAX / SLAB_BYTES);
		g_prefetch_left += n;
	}
	if (n == 0) return;
	*helper = std::thread([be, n]() { // two lockers side by side (the driver serialises part of the work, not all of it)
		std::atomic<int> next{0};
		auto lock_slabs = [&]() {
			while (next.fetch_add(1) < n) {
				void *q = nullptr;
				const bool ok = be->host_alloc(SLAB_BYTES, &q) == 0;
				std::lock_guard<std::mutex> lk(g_slab_mu);
				--g_prefetch_left;
				if (ok) { HostSlab s; s.p = (char *)q, s.cap = SLAB_BYTES, s.pinned = true; g_slab_cache.push_back(s), g_slab_cached += s.cap; }
				g_slab_cv.notify_all();
			}
		};
		std::thread second(lock_slabs);
		lock_slabs();
		second.join();
	});
}

// A slab that no more blocks will be carved out of and whose blocks have all been written goes on its way to the device at once (round 6: the DMA
// runs while later files are still parsed; pga_create() takes its blocks out of that copy).  Page-locked slabs only, and only once the device is up.
static void stage_slab(DataExt *ext, char *p, size_t n)
{
	const pga_backend_t *be = backend_default();
	static const bool off = [] { const char *e = std::getenv("PANGENE_STAGE"); return e && *e == '0'; }();
	bool ok = !off && be->stage != nullptr && be->is_device() && g_device_up.load() && n > 0 && be->stage(p, n) == 0;
	if (ok) return;
	std::lock_guard<std::mutex> lk(ext->slab_mu);
	for (HostSlab &x : ext->slabs) if (x.p == p) x.staged = false; // (uploaded by pga_create as before)
}

static void *block_alloc(DataExt *ext, size_t bytes)
{
	bytes = (bytes + 255) & ~(size_t)255;
	char *st_p = nullptr; size_t st_n = 0;
	void *r;
	{
		std::lock_guard<std::mutex> lk(ext->slab_mu);
		if (ext->slabs.empty() || ext->slabs.back().off + bytes > ext->slabs.back().cap) {
			if (!ext->slabs.empty()) { // the slab in use so far is closed: staged now if nobody is still writing into it, else by the last writer (block_done)
				HostSlab &prev = ext->slabs.back();
				prev.closed = true;
				if (prev.pending == 0 && prev.pinned && !prev.staged && prev.off) prev.staged = true, st_p = prev.p, st_n = prev.off;
			}
			// Page-locking costs ~1 ms per MB (measured: 580 MB of blocks for 12 M hits = 0.5 s, twice the parsing itself) and the DMA
			// it buys saves ~0.1 ms per MB on the one upload.  What the cache holds is used; beyond PIN_BUDGET of freshly locked
			// memory a read takes plain pages (the runtime stages those uploads).
			static const size_t PIN_BUDGET = [] { const char *e = std::getenv("PANGENE_PIN_BUDGET_MB"); return (size_t)(e ? std::max(0, std::atoi(e)) : 192) << 20; }(); // (tuning: 0 = no page-locking while files are read)
			size_t fresh = 0;
			for (const HostSlab &x : ext->slabs) if (x.pinned && x.fresh) fresh += x.cap;
			HostSlab ns = slab_get(bytes, fresh + SLAB_BYTES <= PIN_BUDGET);
			ext->slabs.push_back(ns);
		}
		HostSlab &s = ext->slabs.back();
		r = s.p + s.off;
		s.off += bytes;
		++s.pending;
	}
	if (st_p) stage_slab(ext, st_p, st_n);
	return r;
}

// the block at `buf` has been written completely (pack_one)
static void block_done(DataExt *ext, const void *buf)
{
	char *st_p = nullptr; size_t st_n = 0;
	{
		std::lock_guard<std::mutex> lk(ext->slab_mu);
		for (HostSlab &x : ext->slabs)
			if (x.p && (const char *)buf >= x.p && (const char *)buf < x.p + x.cap) {
				if (x.pending > 0) --x.pending;
				if (x.closed && x.pending == 0 && x.pinned && !x.staged && x.off) x.staged = true, st_p = x.p, st_n = x.off;
				break;
			}
	}
	if (st_p) stage_slab(ext, st_p, st_n);
}

static void slab_put(HostSlab &s) // back into the process-wide cache (or to the system)
{
	if (s.p == nullptr) return;
	std::lock_guard<std::mutex> lk(g_slab_mu);
	if (s.pinned && g_slab_cached + s.cap <= SLAB_CACHE_MAX) g_slab_cache.push_back(s), g_slab_cached += s.cap;
	else if (s.pinned) backend_default()->host_free(s.p);
	else std::free(s.p);
	s.p = nullptr;
}

// release cached pinned memory beyond `keep` bytes (largest slabs first stay: they are the ones the next upload asks for)
void trim_host_caches(size_t keep)
{
	const pga_backend_t *be = backend_default();
	{
		std::lock_guard<std::mutex> lk(g_slab_mu);
		std::sort(g_slab_cache.begin(), g_slab_cache.end(), [](const HostSlab &a, const HostSlab &b) { return a.cap > b.cap; });
		size_t kept = 0, n_keep = 0;
		for (; n_keep < g_slab_cache.size() && kept + g_slab_cache[n_keep].cap <= keep; ++n_keep) kept += g_slab_cache[n_keep].cap;
		for (size_t i = n_keep; i < g_slab_cache.size(); ++i) {
			if (g_slab_cache[i].pinned) be->host_free(g_slab_cache[i].p); else std::free(g_slab_cache[i].p);
		}
		g_slab_cache.resize(n_keep);
		g_slab_cached = kept;
	}
	if (be->host_trim) be->host_trim(keep / 4);
}

// The blocks of an upload that is over.  Page-locked slabs go back into the process-wide cache at once; what has to be given back to
// the system (plain pages: free() of gigabytes is munmap work, 7-9 ms for the 0.6 GB of a 12 M-hit shard, 110 ms for configs[3]) goes
// on a thread of its own unless the caller wants it done now: nobody waits for memory to be unmapped.
void free_packs(DataExt *ext, bool wait)
{
	std::vector<GenomePack> old_packs;
	old_packs.swap(ext->packs); // (their vectors of virtual-contig tables: freed with the rest)
	ext->packs.resize(old_packs.size());
	std::vector<HostSlab> plain;
	for (HostSlab &s : ext->slabs) // (a staged copy of a slab ends here: before the slab can be written again)
		if (s.p && s.staged) { if (backend_default()->stage_drop) backend_default()->stage_drop(s.p); s.staged = false; }
	{
		std::lock_guard<std::mutex> lk(g_slab_mu);
		const pga_backend_t *be = backend_default();
		for (HostSlab &s : ext->slabs) {
			if (s.p == nullptr) continue;
			if (s.pinned && g_slab_cached + s.cap <= SLAB_CACHE_MAX) g_slab_cache.push_back(s), g_slab_cached += s.cap;
			else if (s.pinned) be->host_free(s.p);
			else plain.push_back(s);
		}
		ext->slabs.clear();
	}
	auto drop = [](std::vector<HostSlab> pl, std::vector<GenomePack> pk) { for (HostSlab &s : pl) std::free(s.p); pk.clear(); };
	if (wait || (plain.empty() && old_packs.size() < 64)) { drop(std::move(plain), std::move(old_packs)); return; }
	// (a helper thread owns what it frees; nothing else refers to it.  At the thread limit std::thread throws: free here then -- an exception must
	// not cross the extern "C" entry points above this)
	try { std::thread(drop, plain, old_packs).detach(); }
	catch (const std::system_error &) { drop(std::move(plain), std::move(old_packs)); }
}

static int build_backend(const pg_opt_t *opt, pg_data_t *d, DataExt *ext)
{
	const double t_bb0 = now_sec();
	ext->be = backend_default();
	ext->local_genomes.clear();
	ext->is_local.resize((size_t)d->n_genome, 1);
	ext->hits_sorted.resize((size_t)d->n_genome, 0); // genomes a previous sync_host put into cs order stay marked: their file order is in file_of_host
	ext->pos_valid = false, ext->host_full = false, ext->order_touched = false, ext->pos_sig = 0;
	for (int32_t j = 0; j < d->n_genome; ++j)
		if (ext->is_local[(size_t)j]) ext->local_genomes.push_back(j);
	const int32_t nl = (int32_t)ext->local_genomes.size();
	// Are the packs the reader made still what the genomes are (a caller may edit the public pg_data_t between pg_read_paf and here)?  The check reads
	// every record once (0.8 ms of a 9 ms upload-inclusive pass at configs[1], 5-7 ms of 80 at 12.1 M hits) and nothing in the upload needs its answer
	// before the end: it runs on threads of its own WHILE the blocks are copied to the device (round 6), and a pack that turns out stale -- no caller
	// of the reference's CLI ever makes one -- costs a second upload.  PANGENE_PACK_CHECK=first: the check in front of the upload, as before.
	static const bool check_first = [] { const char *e = std::getenv("PANGENE_PACK_CHECK"); return e && std::strcmp(e, "first") == 0; }();
	if (ext->packs.size() < (size_t)d->n_genome) ext->packs.resize((size_t)d->n_genome);
	std::vector<uint8_t> had((size_t)d->n_genome, 0); // packs that exist already: the ones to check (those made below are new)
	for (int32_t j = 0; j < d->n_genome; ++j) had[(size_t)j] = ext->packs[(size_t)j].buf != nullptr;
	std::vector<int32_t> stale;
	std::thread checker;
	struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{checker}; // (every way out of this function waits for the check)
	if (check_first) pack_genomes(d, ext, 0, d->n_genome); // check, then pack what is missing or stale
	else {
		try { checker = std::thread([&]() { stale_packs(d, ext, 0, d->n_genome, stale, &had); }); }
		catch (const std::system_error &) { stale_packs(d, ext, 0, d->n_genome, stale, &had); }
		pack_genomes(d, ext, 0, d->n_genome, 1.0, false); // only those the reader has not packed already (or whose pack was released after an earlier upload)
	}
	static const bool timing = std::getenv("PANGENE_TIMING") != nullptr;
	double t1 = t_bb0, t2 = t_bb0;
	int rc = 0;
	for (int attempt = 0;; ++attempt) {
		// (attempt 0 may work with a stale pack: whatever goes wrong with it is looked at again once the check has spoken)
		auto verdict = [&]() -> bool { // true: some pack was stale and has been made again -- go round once more
			if (attempt > 0 || check_first) return false;
			if (checker.joinable()) checker.join();
			if (stale.empty()) return false;
			if (ext->ctx) ext->be->destroy(ext->ctx), ext->ctx = nullptr;
			for (int32_t j : stale) ext->packs[(size_t)j] = GenomePack(); // (the slab keeps the old bytes until the upload is over)
			pack_genomes(d, ext, 0, d->n_genome, 1.0, false);
			if (timing) std::fprintf(stderr, "[build_backend] %zu pack(s) no longer matched their genome: packed and uploaded again\n", stale.size());
			return true;
		};
		int64_t N = 0, E = 0;
		ext->hit_off.assign((size_t)nl + 1, 0);
		std::vector<pga_genome_block_t> blk((size_t)nl);
		int err = 0;
		bool limit = false;
		for (int32_t k = 0; k < nl && !err; ++k) {
			const GenomePack &pk = ext->packs[(size_t)ext->local_genomes[(size_t)k]];
			if (pk.err) { err = pk.err; break; }
			blk[(size_t)k] = pk.blk;
			N += pk.blk.n_hit, E += pk.blk.n_exon;
			ext->hit_off[(size_t)k + 1] = N;
		}
		if (!err && (d->n_gene >= (1 << 20) || d->n_genome >= (1 << 24) || E >= INT32_MAX || N >= INT32_MAX)) err = PGA_ERR_RANGE, limit = true;
		if (err) {
			if (verdict()) continue;
			if (!limit) free_packs(ext, false); // (a pack's own error: its blocks go; the limits of the shard leave them, as before)
			return err;
		}
		ext->vreal.assign((size_t)nl, std::vector<int32_t>()), ext->n_vctg.assign((size_t)nl, 0);
		for (int32_t k = 0; k < nl; ++k) { // contigs as the backend counts them (virtual contigs: the pieces of the long ones)
			const GenomePack &pk = ext->packs[(size_t)ext->local_genomes[(size_t)k]];
			ext->n_vctg[(size_t)k] = pk.blk.n_ctg;
			if (!pk.vreal.empty()) ext->vreal[(size_t)k] = pk.vreal;
		}
		if (pg_verbose >= 3) {
			int64_t n_cut = 0, n_piece = 0, n_g = 0;
			for (int32_t k = 0; k < nl; ++k) {
				const GenomePack &pk = ext->packs[(size_t)ext->local_genomes[(size_t)k]];
				if (pk.vreal.empty()) continue;
				++n_g;
				for (size_t v = 0; v < pk.vfirst.size(); ++v) n_piece += pk.vfirst[v] != (int32_t)v, n_cut += v + 1 < pk.vfirst.size() && pk.vfirst[v] == (int32_t)v && pk.vfirst[v + 1] == (int32_t)v;
			}
			if (n_g) std::fprintf(stderr, "[M::%s] 64-bit coordinates: %lld genome(s) with virtual contigs, %lld contig(s) cut, %lld extra piece(s)\n", __func__, (long long)n_g, (long long)n_cut, (long long)n_piece);
		}
		std::vector<int32_t> pgid((size_t)d->n_prot);
		std::vector<uint8_t> gpref((size_t)d->n_gene);
		for (int32_t p = 0; p < d->n_prot; ++p) pgid[(size_t)p] = d->prot[p].gid;
		for (int32_t q = 0; q < d->n_gene; ++q) gpref[(size_t)q] = d->gene[q].preferred;
		pga_shard_t sh;
		std::memset(&sh, 0, sizeof(sh));
		sh.abi_version = PGA_ABI_VERSION;
		sh.n_genome = nl, sh.n_genome_global = d->n_genome, sh.genome_global = ext->local_genomes.data();
		sh.n_prot = d->n_prot, sh.n_gene = d->n_gene, sh.n_hit = N, sh.n_exon = E;
		sh.block = blk.data(), sh.prot_gid = pgid.data(), sh.gene_pref = gpref.data();
		pga_params_t par;
		std::memset(&par, 0, sizeof(par));
		par.min_ov_ratio = opt->min_ov_ratio;
		par.check_strand = !!(opt->flag & PG_F_CHECK_STRAND);
		par.drop_sgl_exon = !!(opt->flag & PG_F_DROP_SGL_EXON);
		if (ext->ctx) ext->be->destroy(ext->ctx), ext->ctx = nullptr;
		ext->n_hit_local = N;
		t1 = now_sec();
		rc = ext->be->create(&ext->ctx, &sh, &par); // returns when the blocks have been read
		if (rc == 0) g_device_up.store(true);
		t2 = now_sec();
		if (verdict()) continue;
		break;
	}
	size_t n_pin = 0, n_plain = 0, n_fresh = 0;
	for (const HostSlab &x : ext->slabs) { if (x.pinned) ++n_pin; else ++n_plain; if (x.fresh) ++n_fresh; }
	free_packs(ext, false);
	if (timing) {
		std::fprintf(stderr, "[build_backend] tables %.3f ms, create %.3f ms, release of the host blocks %.3f ms; block slabs: %zu page-locked, %zu plain, %zu new in this read\n", (t1 - t_bb0) * 1e3, (t2 - t1) * 1e3, (now_sec() - t2) * 1e3, n_pin, n_plain, n_fresh);
	}
	return rc;
}

// Pull per-hit state back.
//   full = false: what the GFA writers need -- one flt bit per hit (DataExt::flt_bits, indexed by X position) and, once per order
//                 epoch, the two orders (pos_x: file index -> X position; y_file: k-th hit in cm order -> file index).  The host
//                 records are neither moved nor refreshed: pg_write_walk reaches them through the file index.
//   full = true : every per-hit field of the host records, and the records themselves put into cs order, which is how the
//                 reference leaves them (hit.c:57-63 replaces g->hit on every sort; the BED writers print in array order).
int sync_host(pg_data_t *d, bool full)
{
	DataExt *ext = ext_of(d, false);
	if (ext == nullptr || ext->ctx == nullptr) return g_err;
	if (!ext->host_stale && !(full && !ext->host_full)) return g_err;
	Phase ph(PH_SYNC_HOST);
	const int64_t N = ext->n_hit_local;
	uint64_t sig_now = ext->pos_sig;
	if (ext->order_touched) { // overrides were handed over since the last sync: do the tracked contigs hold other orders than the copy was taken from?
		exact_shutdown(ext); // (a replay nobody has asked for yet)
		sig_now = order_signature(ext);
		if (sig_now != ext->pos_sig) ext->pos_valid = false;
		ext->order_touched = false;
	}
	const bool need_pos = !ext->pos_valid;
	// The arrays land in one of the pinned slabs the genome blocks were uploaded from (idle by now, kept in the process-wide
	// cache): megabytes copied into pageable memory make the runtime pin and unpin the destination, which costs milliseconds and
	// was seen to stall the NEXT runtime call of a process's first big run by 10-25 ms.
	const size_t nbits = (size_t)((N + 63) / 64) + 1, n_arr = (full ? 5 : 0) + (need_pos ? 2 : 0);
	HostSlab land = slab_get(sizeof(int32_t) * (size_t)N * n_arr + sizeof(uint64_t) * nbits + 64);
	if (land.p == nullptr) { set_error(PGA_ERR_NOMEM, "sync_host"); return g_err; }
	int32_t *w = (int32_t *)land.p;
	uint32_t *flags = nullptr;
	int32_t *rank = nullptr, *sdom = nullptr, *pdom = nullptr, *pdom0 = nullptr, *pxl = nullptr, *py = nullptr;
	if (full) flags = (uint32_t *)w, rank = w + (size_t)N, sdom = w + 2 * (size_t)N, pdom = w + 3 * (size_t)N, pdom0 = w + 4 * (size_t)N, w += 5 * (size_t)N;
	if (need_pos) pxl = w, py = w + (size_t)N, w += 2 * (size_t)N;
	uint64_t *bits = (uint64_t *)(land.p + ((((char *)w - land.p) + 7) & ~(size_t)7));
	pga_hit_state_t st = { flags, rank, sdom, pdom, pdom0, pxl, py, bits };
	{
		const int rc = ext->be->download(ext->ctx, &st);
		if (rc) { slab_put(land); set_error(rc, "download"); return rc; }
	}
	ext->flt_bits.assign(bits, bits + nbits);
	if (need_pos && ext->pos_x.size() != (size_t)N) ext->pos_x.resize((size_t)N); // (filled genome by genome on the host threads below: one thread copying 48 MB was a third of this step at 12 M hits)
	struct Land { HostSlab &s; ~Land() { slab_put(s); } } land_guard{land}; // released when the records have been updated
	const int32_t *px = ext->pos_x.data();
	ext->y_file.resize((size_t)d->n_genome);
	ext->file_of_host.resize((size_t)d->n_genome);
	ext->host_of_file.resize((size_t)d->n_genome);
	if (need_pos) ext->host_order_valid = false; // the records (if they were ever moved) follow an older cs order
	const bool move = full && !ext->host_order_valid;
	auto do_genome = [&](size_t k) {
		int32_t j = ext->local_genomes[k];
		pg_genome_t *g = &d->genome[j];
		const int64_t off = ext->hit_off[k];
		if (need_pos) {
			if (g->n_hit > 0) std::memcpy(ext->pos_x.data() + off, pxl + off, sizeof(int32_t) * (size_t)g->n_hit);
			ext->y_file[(size_t)j].assign((size_t)g->n_hit, 0);
			for (int32_t f = 0; f < g->n_hit; ++f) ext->y_file[(size_t)j][(size_t)py[(size_t)(off + f)]] = f;
		}
		if (move) { // the host array is in file order until the first move and in the PREVIOUS cs order afterwards
			pg_hit_t *a = (pg_hit_t *)std::malloc(sizeof(pg_hit_t) * (size_t)(g->n_hit > 0 ? g->n_hit : 1));
			if (!ext->hits_sorted[(size_t)j]) {
				for (int32_t f = 0; f < g->n_hit; ++f) a[px[(size_t)(off + f)]] = g->hit[f];
			} else {
				const int32_t *old = ext->file_of_host[(size_t)j].data(); // host index -> file index
				for (int32_t h = 0; h < g->n_hit; ++h) a[px[(size_t)(off + old[h])]] = g->hit[h];
			}
			if (!ext->arena_owns(g->hit)) std::free(g->hit);
			else { // a batch read's arrays lie in its arena: the pages of this slice go back now (the mapping itself at ext_drop), or a full sync would hold every hit twice
				const uintptr_t a0 = ((uintptr_t)g->hit + 4095) & ~(uintptr_t)4095, a1 = ((uintptr_t)g->hit + sizeof(pg_hit_t) * (size_t)g->m_hit) & ~(uintptr_t)4095;
				if (a1 > a0) (void)madvise((void *)a0, a1 - a0, MADV_DONTNEED);
			}
			g->hit = a, g->m_hit = g->n_hit;
			ext->hits_sorted[(size_t)j] = 1;
			ext->file_of_host[(size_t)j].assign((size_t)g->n_hit, 0);
			ext->host_of_file[(size_t)j].assign((size_t)g->n_hit, 0);
			for (int32_t f = 0; f < g->n_hit; ++f) {
				ext->file_of_host[(size_t)j][(size_t)px[(size_t)(off + f)]] = f;
				ext->host_of_file[(size_t)j][(size_t)f] = px[(size_t)(off + f)];
			}
		}
		if (!full) return;
		for (int32_t f = 0; f < g->n_hit; ++f) {
			const size_t s = (size_t)(off + f);
			pg_hit_t *h = &g->hit[px[s]];
			const uint32_t fl = flags[s];
			h->flt = !!(fl & PGA_F_FLT), h->flt_iso_sub_self = !!(fl & PGA_F_ISO_SUB), h->flt_iso_ov = !!(fl & PGA_F_ISO_OV);
			h->flt_chain = !!(fl & PGA_F_CHAIN), h->pseudo = !!(fl & PGA_F_PSEUDO), h->vtx = !!(fl & PGA_F_VTX);
			h->shadow = !!(fl & PGA_F_SHADOW), h->rep = !!(fl & PGA_F_REP), h->weak_br = (fl & PGA_F_WEAK_MASK) >> PGA_F_WEAK_SHIFT;
			h->rank = rank[s], h->score_dom = sdom[s], h->pid_dom = pdom[s], h->pid_dom0 = pdom0[s];
		}
	};
	if (need_pos || full) { // genomes are independent and the 88-byte records are scattered: spread them over host threads
		const size_t ng = ext->local_genomes.size();
		unsigned nt = N > 200000 ? host_threads(16u) : 1u;
		if (nt > ng) nt = (unsigned)ng;
		if (nt <= 1) { for (size_t k = 0; k < ng; ++k) do_genome(k); }
		else {
			std::vector<std::thread> th;
			for (unsigned t = 0; t < nt; ++t)
				th.emplace_back([&, t]() { for (size_t k = ng * t / nt; k < ng * (t + 1) / nt; ++k) do_genome(k); });
			for (auto &x : th) x.join();
		}
	}
	if (move) ext->host_order_valid = true;
	if (need_pos) ext->pos_sig = sig_now;
	ext->pos_valid = true;
	ext->host_stale = false;
	ext->host_full = full;
	return 0;
}

// Fully tracked contigs need their exact S1 order from stage A on (first-wins ties); the index-0 channel alone only
// matters from stage C on, so its hand-over (and the wait for the background replay) is deferred to pg_graph_gen then.
// PANGENE_TRACE=<file>: after every step of the path, one line with a hash of each per-hit state array as the backend holds it
// (file order, so independent of the backend's internal orders).  Two backends driven over the same input write the same
// lines; the first line that differs names the step -- and so the group of kernels -- that went wrong (tests/test_trace.py).
static const char *trace_path() { static const char *p = std::getenv("PANGENE_TRACE"); return (p && *p) ? p : nullptr; }
static uint64_t fnv1a(const void *data, size_t n, uint32_t mask)
{
	uint64_t h = 1469598103934665603ull;
	const uint32_t *w = (const uint32_t *)data;
	for (size_t i = 0; i < n; ++i) {
		uint32_t v = w[i] & mask;
		for (int b = 0; b < 4; ++b) h = (h ^ (v & 0xffu)) * 1099511628211ull, v >>= 8;
	}
	return h;
}
static int trace_state(DataExt *ext, const char *step, int round, bool first = false)
{
	if (trace_path() == nullptr || ext == nullptr || ext->ctx == nullptr) return 0;
	const size_t N = (size_t)ext->n_hit_local;
	std::vector<uint32_t> flags(N + 1);
	std::vector<int32_t> rank(N + 1), sdom(N + 1), pdom(N + 1), pdom0(N + 1), px(N + 1), py(N + 1);
	pga_hit_state_t st = { flags.data(), rank.data(), sdom.data(), pdom.data(), pdom0.data(), px.data(), py.data(), nullptr };
	BE_CALL(ext->be->download(ext->ctx, &st), "download(trace)");
	// PANGENE_TRACE_DIFF=1: how much of the per-hit state a step changed (hits whose walkable test -- flt or shadow --, whose weak_br,
	// whose flag word changed; the 256-hit tiles of the cs order that hold such a hit): what "rounds that cost what changed" would have to touch
	static const bool diff_on = std::getenv("PANGENE_TRACE_DIFF") != nullptr;
	if (diff_on) {
		static std::vector<uint32_t> prev;
		if (prev.size() == N && !first) {
			int64_t n_any = 0, n_walk = 0, n_weak = 0, n_flt = 0;
			std::vector<uint8_t> tile((N + 255) / 256 + 1, 0);
			size_t k = 0;
			for (size_t f = 0; f < N; ++f) {
				const uint32_t x = (flags[f] ^ prev[f]) & 0x7ffu;
				if (!x) continue;
				++n_any, n_walk += (x & (PGA_F_FLT | PGA_F_SHADOW)) != 0, n_weak += (x & PGA_F_WEAK_MASK) != 0, n_flt += (x & PGA_F_FLT) != 0;
				while (k + 1 < ext->hit_off.size() && (int64_t)f >= ext->hit_off[k + 1]) ++k;
				tile[(size_t)(ext->hit_off[k] + px[f]) >> 8] = 1;
			}
			int64_t n_tile = 0;
			for (uint8_t t : tile) n_tile += t;
			int64_t n_live = 0, n_wk = 0; // what a step that skips filtered hits still has to look at: hits without flt, and the walkable ones among them
			for (size_t f = 0; f < N; ++f) n_live += !(flags[f] & PGA_F_FLT), n_wk += !(flags[f] & (PGA_F_FLT | PGA_F_SHADOW));
			std::fprintf(stderr, "[trace_diff] %-18s %3d: %lld of %zu hits changed their flag word (walkable test %lld, flt %lld, weak_br %lld); %lld of %zu tiles of 256 hits dirty; live (flt == 0) %lld = %.3f, walkable %lld = %.3f\n", step, round,
			             (long long)n_any, N, (long long)n_walk, (long long)n_flt, (long long)n_weak, (long long)n_tile, tile.size(), (long long)n_live, (double)n_live / (double)(N ? N : 1), (long long)n_wk, (double)n_wk / (double)(N ? N : 1));
		}
		else { // the first step of a run: nothing to compare with
			int64_t n_live = 0, n_wk = 0;
			for (size_t f = 0; f < N; ++f) n_live += !(flags[f] & PGA_F_FLT), n_wk += !(flags[f] & (PGA_F_FLT | PGA_F_SHADOW));
			std::fprintf(stderr, "[trace_diff] %-18s %3d: %zu hits; live (flt == 0) %lld = %.3f, walkable %lld = %.3f\n", step, round, N, (long long)n_live, (double)n_live / (double)(N ? N : 1), (long long)n_wk, (double)n_wk / (double)(N ? N : 1));
		}
		prev.assign(flags.begin(), flags.begin() + (long)N);
	}
	std::FILE *fp = std::fopen(trace_path(), first ? "w" : "a");
	if (fp == nullptr) return 0;
	std::fprintf(fp, "%s\t%d\tflags=%016llx\trank=%016llx\tscore_dom=%016llx\tpid_dom=%016llx\tpid_dom0=%016llx\tpos_x=%016llx\tpos_y=%016llx\n", step, round,
	             (unsigned long long)fnv1a(flags.data(), N, 0x7ffu), (unsigned long long)fnv1a(rank.data(), N, ~0u), (unsigned long long)fnv1a(sdom.data(), N, ~0u),
	             (unsigned long long)fnv1a(pdom.data(), N, ~0u), (unsigned long long)fnv1a(pdom0.data(), N, ~0u), (unsigned long long)fnv1a(px.data(), N, ~0u),
	             (unsigned long long)fnv1a(py.data(), N, ~0u));
	std::fclose(fp);
	return 0;
}

static bool exact_early(const DataExt *ext)
{
	if (exact_mode() == 2) return true;
	for (const ExactSeg &s : ext->xsegs) if (s.full) return true;
	return false;
}

// ---------------------------------------------------------------------------------------------
// stage A + B
// ---------------------------------------------------------------------------------------------
static int post_process_impl(const pg_opt_t *opt, pg_data_t *d)
{
	DataExt *ext = ext_of(d, true);
	double t0 = now_sec();
	if (ext->read_failed) { set_error(PGA_ERR_NOMEM, "pg_post_process: a pg_read_paf ran out of memory, the data set is incomplete"); return PGA_ERR_NOMEM; }
	if (!(ext->rerun && ext->ctx)) {
		g_pack_sec = ext->pack_sec, ext->pack_sec = 0.0; // what the reader spent on the blocks of this upload
		BE_CALL(build_backend(opt, d, ext), "create"); // (rest of the) pack + allocation + H2D
		g_pack_sec += ext->pack_sec, ext->pack_sec = 0.0;
		ext->route_verbose = -1;
		if (sharded()) { // the level the routes of this data set follow on EVERY rank (route_v): the most verbose rank's
			int32_t v = pg_verbose;
			void *scr;
			BE_CALL(ext->be->scratch(ext->ctx, 16, &scr), "scratch");
			BE_CALL(ext->be->put(ext->ctx, scr, &v, sizeof(v)), "put");
			BE_CALL(xreduce(ext->be, ext->ctx, scr, 1, PG_X_I32, PG_X_MAX), "allreduce(log level)");
			BE_CALL(ext->be->fetch(ext->ctx, &v, scr, sizeof(v)), "fetch");
			ext->route_verbose = v;
		}
		const double tx0 = now_sec();
		const bool cs_opt = !!(opt->flag & PG_F_CHECK_STRAND); // (enters the static tie prediction of exact_init)
		if (!(ext->xsegs_n_genome == d->n_genome && ext->exact_mode_of_segs == exact_mode() && ext->extra_ctgs.empty() && ext->xreplayed && ext->check_strand == cs_opt && ext->min_ov_ratio == opt->min_ov_ratio)) { // else: the reader did it
			ext->check_strand = cs_opt, ext->min_ov_ratio = opt->min_ov_ratio;
			exact_init(d, ext);
			ext->exact_mode_of_segs = exact_mode();
		}
		if (std::getenv("PANGENE_TIMING")) std::fprintf(stderr, "[post_process] exact_init %.3f ms\n", (now_sec() - tx0) * 1e3);
	} else if (ext->exact_mode_of_segs != exact_mode() || ext->check_strand != !!(opt->flag & PG_F_CHECK_STRAND) || ext->min_ov_ratio != opt->min_ov_ratio) {
		exact_shutdown(ext);
		ext->check_strand = !!(opt->flag & PG_F_CHECK_STRAND), ext->min_ov_ratio = opt->min_ov_ratio;
		exact_init(d, ext);
		ext->exact_mode_of_segs = exact_mode();
	}
	ext->rerun = false;
	g_upload_sec = now_sec() - t0;
	const pga_backend_t *be = ext->be;
	pga_ctx_t *ctx = ext->ctx;
	g_t_path0 = now_sec();
	for (int i = 0; i < PH_COUNT; ++i) g_phase[i] = 0;
	exact_begin(ext); // background replay of the reference's sort sequence (keys only), overlaps stages A and B
	{ Phase ph(PH_BEGIN); BE_CALL(be->begin(ctx), "begin"); }
	// pg_hit_sort(g, 0), read.c:247.  Mode "all" needs the exact S1 order for stage A (first-wins ties); in mode "auto"
	// only array index 0 matters and it is inert while every shadow flag is 0 (read.c:252, i.e. until the sweep of
	// round 1), so the hand-over waits until pg_graph_gen and the replay overlaps stages A and B.
	if (exact_early(ext)) { Phase ph(PH_EXACT); BE_CALL(exact_sort(ext, 0), "override_order"); }
	const int32_t nl = (int32_t)ext->local_genomes.size(), P = d->n_prot;
	if (pg_verbose >= 3)
		std::fprintf(stderr, "[M::%s::%s] %d genes and %d proteins; %ld hits of %d genomes on backend '%s'\n", __func__, stamp(),
		             d->n_gene, d->n_prot, (long)ext->n_hit_local, nl, be->name);

	std::vector<int32_t> st4((size_t)nl * 4);
	{ Phase ph(PH_INGEST); BE_CALL(be->ingest(ctx, pg_verbose >= 3 ? st4.data() : nullptr), "ingest"); } // read.c:243-260 for every local genome
	BE_CALL(trace_state(ext, "ingest", 0, true), "trace");
	Phase ph_post(PH_POST);
	if (pg_verbose >= 3)
		for (int32_t k = 0; k < nl; ++k) {
			const pg_genome_t *g = &d->genome[ext->local_genomes[(size_t)k]];
			std::fprintf(stderr, "[M::pg_read_paf::%s] [%d] %s: %d kept and %d+%d+%d+%d filtered\n", stamp(), ext->local_genomes[(size_t)k],
			             g->label ? g->label : "-", g->n_hit, st4[(size_t)k*4], st4[(size_t)k*4+1], st4[(size_t)k*4+2], st4[(size_t)k*4+3]);
		}

	int32_t *b_max; int64_t *b_sum;
	BE_CALL(be->post_partials(ctx, &b_max, &b_sum), "post_partials");
	BE_CALL(xreduce(be, ext->ctx, b_max, P, PG_X_I32, PG_X_MAX), "allreduce(max_score_ori)");
	BE_CALL(xreduce(be, ext->ctx, b_sum, 6 * (int64_t)P, PG_X_I64, PG_X_SUM), "allreduce(protein sums)");
	static thread_local std::vector<int32_t> mx; // (reused from pass to pass, as gen_vtx's)
	static thread_local std::vector<int64_t> sm;
	mx.resize((size_t)P), sm.resize((size_t)P * 6);
	if (P) { // one wait for both vectors: the first copy rides with the second one's
		const void *mx_view = nullptr;
		BE_CALL(be->fetch_later(ctx, b_max, sizeof(int32_t) * (size_t)P, &mx_view), "fetch_later");
		BE_CALL(be->fetch(ctx, sm.data(), b_sum, sizeof(int64_t) * (size_t)P * 6), "fetch");
		std::memcpy(mx.data(), mx_view, sizeof(int32_t) * (size_t)P);
	}
	const double tp0 = now_sec();
	// pg_cap_score_dom's table (hit.c:230-238) and pg_flag_representative's protein part (hit.c:205-217)
	static thread_local std::vector<pg128_t> z;
	z.resize((size_t)P);
	for (int32_t i = 0; i < d->n_gene; ++i) d->gene[i].rep_pid = -1;
	for (int32_t i = 0; i < P; ++i) {
		d->prot[i].max_score_ori = mx[(size_t)i];
		d->prot[i].rep = 0;
		z[(size_t)i].x = ((uint64_t)sm[(size_t)i] << 32) + (uint64_t)sm[(size_t)P + (size_t)i];
		z[(size_t)i].y = (uint64_t)i;
		d->prot[i].n = (int32_t)(uint32_t)z[(size_t)i].x;
		d->prot[i].avg_score_adj = d->prot[i].n ? (int32_t)((double)(z[(size_t)i].x >> 32) / d->prot[i].n + .499) : 0;
	}
	const double tp1 = now_sec();
	ksort_exact(z.data(), z.size(), [](const pg128_t &a) { return a.x; }); // unstable in the reference; ties reach LN/pp
	const double tp2 = now_sec();
	for (int32_t i = P - 1; i >= 0; --i) {
		int32_t pid = (int32_t)z[(size_t)i].y, gid = d->prot[pid].gid;
		if (d->gene[gid].rep_pid < 0) d->gene[gid].rep_pid = pid, d->prot[pid].rep = 1;
	}
	static thread_local std::vector<uint8_t> rep, pj;
	rep.resize((size_t)P), pj.assign((size_t)P, 0);
	for (int32_t i = 0; i < P; ++i) rep[(size_t)i] = (uint8_t)d->prot[i].rep;
	if (!(opt->flag & PG_F_NO_JOINT_PSEUDO)) { // per-protein predicate of hit.c:176-181 (doubles, no contraction)
		for (int32_t i = 0; i < P; ++i) {
			const int64_t c0 = sm[2 * (size_t)P + (size_t)i], c1 = sm[3 * (size_t)P + (size_t)i];
			const int64_t s0 = sm[4 * (size_t)P + (size_t)i], s1 = sm[5 * (size_t)P + (size_t)i];
			const int32_t ic1 = (int32_t)c1, ic0 = (int32_t)c0;
			bool a = ic1 > 0 && ic1 >= d->n_genome * opt->min_vertex_ratio && ((double)s1 / ic1) / ((double)s0 / ic0) >= 0.99;
			bool b = (ic1 == 0 || ic1 <= d->n_genome * opt->min_vertex_ratio) && (opt->flag & PG_F_DROP_SGL_EXON);
			pj[(size_t)i] = a || b;
		}
	}
	if (std::getenv("PANGENE_TIMING")) std::fprintf(stderr, "[post] host step between the fetch of the protein sums and post_apply: %.3f ms (%d proteins; sums -> keys %.3f, sort of the proteins %.3f, representatives + joint-pseudo predicate %.3f)\n", (now_sec() - tp0) * 1e3, P, (tp1 - tp0) * 1e3, (tp2 - tp1) * 1e3, (now_sec() - tp2) * 1e3);
	int64_t n_pj = 0;
	BE_CALL(be->post_apply(ctx, rep.data(), pj.data(), (!(opt->flag & PG_F_NO_JOINT_PSEUDO) && pg_verbose >= 3) ? &n_pj : nullptr), "post_apply");
	if (!(opt->flag & PG_F_NO_JOINT_PSEUDO) && pg_verbose >= 3)
		std::fprintf(stderr, "[M::%s::%s] %ld pseudogene hits identified jointly\n", __func__, stamp(), (long)n_pj);
	std::vector<int32_t> st2((size_t)nl * 2);
	BE_CALL(be->shadow(ctx, 0, pg_verbose >= 3 ? st2.data() : nullptr), "shadow"); // graph.c:20-28
	if (pg_verbose >= 3)
		for (int32_t k = 0; k < nl; ++k) {
			const pg_genome_t *g = &d->genome[ext->local_genomes[(size_t)k]];
			std::fprintf(stderr, "[M::%s::%s] genome[%d]: %s; %d hits remain, of which %d are shadowed\n", __func__, stamp(),
			             ext->local_genomes[(size_t)k], g->label ? g->label : "-", st2[(size_t)k*2], st2[(size_t)k*2+1]);
		}
	ext->host_stale = true;
	BE_CALL(trace_state(ext, "post_process", 0), "trace");
	return 0;
}

// ---------------------------------------------------------------------------------------------
// stage C helpers (S/A-sized, host)
// ---------------------------------------------------------------------------------------------
static void gen_g2s(pg_graph_t *q) // graph.c:49-59
{
	const pg_data_t *d = q->d;
	std::free(q->g2s);
	q->g2s = (int32_t *)std::malloc(sizeof(int32_t) * (size_t)(d->n_gene > 0 ? d->n_gene : 1));
	for (int32_t i = 0; i < d->n_gene; ++i) q->g2s[i] = -1;
	for (int32_t i = 0; i < q->n_seg; ++i) q->g2s[q->seg[i].gid] = i;
}

static void arc_index(pg_graph_t *q) // graph.c:202-217
{
	std::free(q->idx);
	q->idx = (uint64_t *)std::calloc((size_t)(q->n_seg > 0 ? q->n_seg * 2 : 1), sizeof(uint64_t));
	for (int32_t i0 = 0, i = 1; i <= q->n_arc; ++i)
		if (i == q->n_arc || q->arc[i].x >> 32 != q->arc[i0].x >> 32)
			q->idx[q->arc[i0].x >> 32] = (uint64_t)i0 << 32 | (uint32_t)(i - i0), i0 = i;
}

// pg_gen_vtx (vertex.c:6-100): counts and sub->dom triples come from the backend, the order-sensitive
// greedy stays here
static int gen_vtx(const pg_opt_t *opt, pg_graph_t *q, DataExt *ext)
{
	pg_data_t *d = q->d;
	const pga_backend_t *be = ext->be;
	const int32_t Q = d->n_gene, G = d->n_genome;
	Phase ph_vtx(PH_VTX);
	int32_t *b_cnt; uint64_t *b_tri; int64_t n_tri;
	const double tv0 = now_sec();
	BE_CALL(be->vtx_partials(ext->ctx, &b_cnt, &b_tri, &n_tri), "vtx_partials");
	const double tv1 = now_sec();
	BE_CALL(xreduce(be, ext->ctx, b_cnt, 2 * (int64_t)Q, PG_X_I32, PG_X_SUM), "allreduce(n_dom,n_sub)");
	static thread_local std::vector<int32_t> cntv; // (these buffers are reused from pass to pass: fresh vectors of a few hundred KB are mmap'ed and page-faulted every time, ~0.1 ms of a 4.6 ms pass)
	cntv.assign((size_t)Q * 2, 0);
	const void *cntv_view = nullptr; // copied out behind the next wait (the fetch of the pair records, or the explicit one below)
	if (Q) BE_CALL(be->fetch_later(ext->ctx, b_cnt, sizeof(int32_t) * (size_t)Q * 2, &cntv_view), "fetch_later");
	// What the greedy needs is, per (sub, dom) gene pair, the SET of genomes in which sub is sub-ordinate to a dominant dom: a
	// selected sub gene marks cell (genome, dom) in each of them (vertex.c:73-77).  The backend hands over one genome
	// bitset per pair of its shard and only those travel between ranks: the host work is O(pairs x G/64), independent of
	// the number of hits.
	const uint64_t m20 = (1u << 20) - 1;
	const int64_t nw = ((int64_t)G + 63) / 64; // words of a genome bitset
	static thread_local std::vector<uint64_t> pairs; // records of 1 + nw words: key = sub << 20 | dom, then the genome bits
	pairs.resize((size_t)(n_tri * (1 + nw)));
	if (n_tri && !sharded()) BE_CALL(be->fetch(ext->ctx, pairs.data(), b_tri, sizeof(uint64_t) * pairs.size()), "fetch");
	const double tv2 = now_sec();
	if (sharded()) { // every rank's records, concatenated in rank order; a pair may come from several ranks (disjoint genome bits)
		BE_CALL(xgather(be, ext->ctx, b_tri, n_tri * (1 + nw), pairs), "allgather(vertex pairs)");
	}
	if (Q) {
		if (n_tri == 0 || sharded()) BE_CALL(be->sync(ext->ctx), "sync"); // (no fetch above waited)
		std::memcpy(cntv.data(), cntv_view, sizeof(int32_t) * (size_t)Q * 2);
	}
	// group the records by sub gene (counting sort on the key's sub field keeps it linear)
	const int64_t n_rec = (int64_t)(pairs.size() / (size_t)(1 + nw));
	static thread_local std::vector<int64_t> sub_off, sub_rec, cur;
	sub_off.assign((size_t)Q + 1, 0);
	for (int64_t r = 0; r < n_rec; ++r) ++sub_off[(size_t)(pairs[(size_t)(r * (1 + nw))] >> 20) + 1];
	for (int32_t g = 0; g < Q; ++g) sub_off[(size_t)g + 1] += sub_off[(size_t)g];
	sub_rec.resize((size_t)n_rec);
	{
		cur.assign(sub_off.begin(), sub_off.end() - 1);
		for (int64_t r = 0; r < n_rec; ++r) sub_rec[(size_t)cur[(size_t)(pairs[(size_t)(r * (1 + nw))] >> 20)]++] = r;
	}
	static thread_local std::vector<uint64_t> marked; // genome bitset per dom gene, allocated on first use
	static thread_local std::vector<int32_t> mark_slot, ycnt; // ycnt: #genomes where the gene is dominant and already marked
	static thread_local std::vector<pg128_t> cnt;
	marked.clear(), mark_slot.assign((size_t)Q, -1), ycnt.assign((size_t)Q, 0), cnt.resize((size_t)Q);
	for (int32_t i = 0; i < Q; ++i) { // vertex.c:18-19,47-56
		cnt[(size_t)i].x = (uint64_t)(int64_t)d->prot[d->gene[i].rep_pid].avg_score_adj;
		cnt[(size_t)i].y = (uint64_t)i;
		cnt[(size_t)i].x += (uint64_t)cntv[(size_t)i] << 32;
		cnt[(size_t)i].y += (uint64_t)cntv[(size_t)Q + (size_t)i] << 32;
		if (d->gene[i].preferred) cnt[(size_t)i].x |= 1ULL << 63;
	}
	ksort_exact(cnt.data(), cnt.size(), [](const pg128_t &a) { return a.x; }); // vertex.c:59, tie order matters
	q->n_seg = 0;
	if (q->m_seg < Q) { // room for every gene at once (the loop below would otherwise grow the array a dozen times)
		const int32_t old_m = q->m_seg;
		q->m_seg = Q + 16;
		q->seg = (pg_seg_t *)std::realloc(q->seg, sizeof(pg_seg_t) * (size_t)q->m_seg);
		std::memset((void *)(q->seg + old_m), 0, sizeof(pg_seg_t) * (size_t)(q->m_seg - old_m));
	}
	marked.reserve((size_t)nw * 1024);
	ext->vtx_sel_text.clear();
	for (int32_t i = Q - 1; i >= 0; --i) { // vertex.c:60-80
		const int32_t n_dom = (int32_t)(cnt[(size_t)i].x << 1 >> 33), n_sub = (int32_t)(cnt[(size_t)i].y >> 32);
		const int32_t gid = (int32_t)cnt[(size_t)i].y;
		const int32_t x = cntv[(size_t)gid], y = ycnt[(size_t)gid];
		if (opt->flag & PG_F_WRITE_VTX_SEL) { // -G (vertex.c:66-67); buffered: a hazard escalation repeats the run
			char line[512];
			std::snprintf(line, sizeof(line), "g\t%s\t%d\t%d\t%d\t%d\t%c\t%c\n", d->gene[gid].name, (int32_t)cnt[(size_t)i].x, x, y, n_sub,
			              "NY"[d->gene[gid].included], "NY"[d->gene[gid].preferred]);
			ext->vtx_sel_text += line;
		}
		if (d->gene[gid].included || (n_dom >= G * opt->min_vertex_ratio && y < x)) {
			if (q->n_seg >= q->m_seg) {
				int32_t old = q->m_seg;
				q->m_seg = q->n_seg + 1; q->m_seg += (q->m_seg >> 1) + 16;
				q->seg = (pg_seg_t *)std::realloc(q->seg, sizeof(pg_seg_t) * (size_t)q->m_seg);
				std::memset((void *)(q->seg + old), 0, sizeof(pg_seg_t) * (size_t)(q->m_seg - old));
			}
			pg_seg_t *p = &q->seg[q->n_seg++];
			p->gid = gid, p->n_dom = n_dom, p->n_sub = n_sub;
			if (x > 0)
				for (int64_t k = sub_off[(size_t)gid]; k < sub_off[(size_t)gid + 1]; ++k) {
					const uint64_t *rec = &pairs[(size_t)(sub_rec[(size_t)k] * (1 + nw))];
					const int32_t dom = (int32_t)(rec[0] & m20);
					if (mark_slot[(size_t)dom] < 0) mark_slot[(size_t)dom] = (int32_t)(marked.size() / (size_t)nw), marked.resize(marked.size() + (size_t)nw, 0);
					uint64_t *mk = &marked[(size_t)mark_slot[(size_t)dom] * (size_t)nw];
					for (int64_t w = 0; w < nw; ++w) {
						const uint64_t fresh = rec[1 + w] & ~mk[w];
						mk[w] |= fresh, ycnt[(size_t)dom] += __builtin_popcountll(fresh);
					}
				}
		}
	}
	if (std::getenv("PANGENE_TIMING")) std::fprintf(stderr, "[vtx] partials %.3f ms, fetch %.3f ms (records %ld / %ld), greedy %.3f ms\n", (tv1 - tv0) * 1e3, (tv2 - tv1) * 1e3, (long)n_tri, (long)n_rec, (now_sec() - tv2) * 1e3);
	// segments by gene id (vertex.c:85-94; keys unique, any sort gives the reference's order)
	{ // (gene ids are unique and < Q: one placement pass instead of a comparison sort)
		std::vector<int32_t> at((size_t)Q, -1);
		for (int32_t i = 0; i < q->n_seg; ++i) at[(size_t)q->seg[i].gid] = i;
		std::vector<pg_seg_t> tmp(q->seg, q->seg + q->n_seg);
		int32_t k = 0;
		for (int32_t g = 0; g < Q; ++g) if (at[(size_t)g] >= 0) q->seg[k++] = tmp[(size_t)at[(size_t)g]];
	}
	gen_g2s(q);
	if (pg_verbose >= 3)
		std::fprintf(stderr, "[M::%s::%s] selected %d vertices out of %d genes\n", "pg_gen_vtx", stamp(), q->n_seg, Q);
	return 0;
}

static int loop_allreduce(void *user, void *buf, int64_t count);
static int loop_allgather(void *user, const void *in, void *out, int64_t bytes);

// pg_gen_arc (graph.c:87-177): per-genome work + local reduce on the backend, cross-shard merge and
// the three double roundings of graph.c:170-172 here
// defer: the next thing is a branch step, which reads the table on the backend and waits for its own results anyway: the round's
// host results (segment counters, degrees) are collected there (arc_collect) instead of being waited for here
static int gen_arc(const pg_opt_t *opt, pg_graph_t *q, DataExt *ext, bool defer = false)
{
	const pga_backend_t *be = ext->be;
	const int32_t S = q->n_seg;
	int32_t *b_seg; pga_arc_part_t *b_arc; int64_t n_loc;
	{ Phase ph(PH_EXACT); BE_CALL(exact_sort(ext, 1), "override_order"); } // graph.c:103
	if (!sharded() && defer && route_v(ext) < 3 && be->arc_round_finish) { // the host results are collected later (arc_collect): nothing to prepare here
		{ Phase ph(PH_ARC_DEV); BE_CALL(be->arc_round_local(ext->ctx, !!(opt->flag & PG_F_ORI_FOR_BRANCH), S, nullptr, nullptr), "arc_round"); }
		{ Phase ph(PH_EXACT); BE_CALL(exact_sort(ext, 0), "override_order"); } // graph.c:123
		ext->arc_pending = true, ext->cur_arcs = nullptr, q->n_arc = 0;
		return 0;
	}
	std::vector<int32_t> sc((size_t)S * 2 + 1);
	ext->deg.assign((size_t)S * 2 + 1, 0);
	ext->arc_via_x = false;
	static const bool no_x = std::getenv("PANGENE_SHARDED_LOOP_HOST") != nullptr; // (tests: the host-driven exchange of a sharded run)
	if (sharded() && !no_x && be->arc_round_x && be->is_device() && ext->x_arc_slot > 0 && S > 0 && route_v(ext) < 3) {
		// every rank's table in a slot of a capacity all ranks share (no size exchange), merged on the backend, ONE wait
		pga_loop_xchg_t lx;
		lx.user = ext, lx.rank = g_xchg.rank, lx.world = g_xchg.world, lx.arc_cap_hint = ext->x_arc_slot, lx.allreduce_i32_sum = loop_allreduce, lx.allgather = loop_allgather;
		int64_t n_arc = 0;
		int rc;
		{ Phase ph(PH_ARC_DEV); rc = be->arc_round_x(ext->ctx, !!(opt->flag & PG_F_ORI_FOR_BRANCH), S, &lx, sc.data(), ext->deg.data(), &n_arc); }
		if (rc < 0) { set_error(rc, "arc_round_x"); return rc; }
		if (rc == 0) {
			{ Phase ph(PH_EXACT); BE_CALL(exact_sort(ext, 0), "override_order"); } // graph.c:123
			for (int32_t i = 0; i < S; ++i) q->seg[i].n_genome = sc[(size_t)i], q->seg[i].tot_cnt = sc[(size_t)S + (size_t)i];
			ext->cur_arcs = nullptr, q->n_arc = (int32_t)n_arc; // (fetch_arcs asks the backend for the table)
			ext->arc_via_x = true;
			return 0;
		}
		// 1: void on some rank (all ranks were told), 2: not applicable -- the host-driven exchange below
	}
	if (!sharded()) {
		// one call, one wait: the backend keeps the table (and what branch marking, hit marking and the degree filter read from
		// it) resident; it travels to the host once, after the last round (fetch_arcs)
		int64_t n_arc = 0;
		const pga_arc_part_t *tab = nullptr;
		{ Phase ph(PH_ARC_DEV); BE_CALL(be->arc_round_local(ext->ctx, !!(opt->flag & PG_F_ORI_FOR_BRANCH), S, sc.data(), ext->deg.data()), "arc_round"); }
		if (pg_verbose >= 3) BE_CALL(be->arc_table(ext->ctx, &tab, &n_arc), "arc_table"); // only the log lines want the number of arcs of every round
		{ Phase ph(PH_EXACT); BE_CALL(exact_sort(ext, 0), "override_order"); } // graph.c:123
		for (int32_t i = 0; i < S; ++i) q->seg[i].n_genome = sc[(size_t)i], q->seg[i].tot_cnt = sc[(size_t)S + (size_t)i];
		ext->cur_arcs = tab, q->n_arc = (int32_t)n_arc;
		return 0;
	}
	{ Phase ph(PH_ARC_DEV); BE_CALL(be->arc_round(ext->ctx, !!(opt->flag & PG_F_ORI_FOR_BRANCH), &b_seg, &b_arc, &n_loc), "arc_round"); }
	{ Phase ph(PH_EXACT); BE_CALL(exact_sort(ext, 0), "override_order"); } // graph.c:123
	Phase ph_host(PH_ARC_HOST);
	const pga_arc_part_t *cur = b_arc;
	int64_t n_cur = n_loc;
	{
		// one all-reduce carries the segment counters and, in W extra slots, every rank's arc-table size (each rank adds its
		// own into its slot), so the all-gather of the tables below needs no size exchange of its own
		const int W = g_xchg.world;
		std::vector<int32_t> mine((size_t)W, 0), all((size_t)S * 2 + (size_t)W);
		mine[(size_t)g_xchg.rank] = (int32_t)n_loc;
		void *scr;
		BE_CALL(be->scratch(ext->ctx, sizeof(int32_t) * ((size_t)S * 2 + (size_t)W), &scr), "scratch");
		if (S) BE_CALL(be->copy(ext->ctx, scr, b_seg, sizeof(int32_t) * (size_t)S * 2), "copy");
		BE_CALL(be->put(ext->ctx, (int32_t *)scr + (size_t)S * 2, mine.data(), sizeof(int32_t) * (size_t)W), "put");
		BE_CALL(xreduce(be, ext->ctx, scr, 2 * (int64_t)S + W, PG_X_I32, PG_X_SUM), "allreduce(seg counts, table sizes)");
		BE_CALL(be->fetch(ext->ctx, all.data(), scr, sizeof(int32_t) * all.size()), "fetch");
		std::copy(all.begin(), all.begin() + (size_t)S * 2, sc.begin());
		// all-gather the local tables (RCCL) and reduce by key on the backend; integer sums => order-independent
		std::vector<int64_t> cnt((size_t)W);
		int64_t slot = 0, n_mg = 0;
		for (int r = 0; r < W; ++r) cnt[(size_t)r] = all[(size_t)S * 2 + (size_t)r], slot = std::max(slot, cnt[(size_t)r]);
		ext->x_arc_slot = std::max(ext->x_arc_slot, slot); // the largest local table of any host-driven round of this data set, over all ranks (every rank has the same number)
		pga_arc_part_t *merged = nullptr;
		if (slot) {
			const size_t bytes = (size_t)slot * sizeof(pga_arc_part_t);
			BE_CALL(be->scratch(ext->ctx, bytes * (size_t)(W + 1), &scr), "scratch");
			if (n_loc) BE_CALL(be->copy(ext->ctx, scr, b_arc, (size_t)n_loc * sizeof(pga_arc_part_t)), "copy");
			BE_CALL(xready(be, ext->ctx), "sync");
			++g_n_coll;
			BE_CALL(g_xchg.allgather(g_xchg.user, scr, (char *)scr + bytes, (int64_t)bytes, be->is_device()), "allgather(arcs)");
			BE_CALL(be->arc_merge(ext->ctx, (pga_arc_part_t *)((char *)scr + bytes), cnt.data(), W, slot, &merged, &n_mg), "arc_merge");
		}
		cur = merged, n_cur = n_mg;
	}
	BE_CALL(be->arc_set_current(ext->ctx, cur, n_cur, S, ext->deg.data()), "arc_set_current");
	for (int32_t i = 0; i < S; ++i) q->seg[i].n_genome = sc[(size_t)i], q->seg[i].tot_cnt = sc[(size_t)S + (size_t)i];
	ext->cur_arcs = cur, q->n_arc = (int32_t)n_cur;
	return 0;
}

// collect the host results of a deferred round.  1: the round had to be repeated (a hub gene overflowed the per-gene table), so
// whatever was computed from its table on the backend in the meantime has to be repeated as well
static int arc_collect(const pg_opt_t *opt, pg_graph_t *q, DataExt *ext)
{
	if (!ext->arc_pending) return 0;
	ext->arc_pending = false;
	const int32_t S = q->n_seg;
	std::vector<int32_t> &sc = ext->sc_buf; // (both buffers are overwritten in full: no clearing, no allocation per round)
	if (sc.size() < (size_t)S * 2 + 1) sc.resize((size_t)S * 2 + 1);
	if (ext->deg.size() < (size_t)S * 2 + 1) ext->deg.resize((size_t)S * 2 + 1);
	int rc = ext->be->arc_round_finish(ext->ctx, S, sc.data(), ext->deg.data());
	if (rc < 0) { set_error(rc, "arc_round_finish"); return rc; }
	if (rc == 1) BE_CALL(ext->be->arc_round_local(ext->ctx, !!(opt->flag & PG_F_ORI_FOR_BRANCH), S, sc.data(), ext->deg.data()), "arc_round");
	for (int32_t i = 0; i < S; ++i) q->seg[i].n_genome = sc[(size_t)i], q->seg[i].tot_cnt = sc[(size_t)S + (size_t)i];
	return rc;
}

// bring the round's arc table to the host and apply the three double roundings of graph.c:170-172
static int fetch_arcs(pg_graph_t *q, DataExt *ext)
{
	Phase ph_host(PH_ARC_HOST);
	{ // the table as one array sorted by x, wherever the last round left it
		int64_t n = 0;
		BE_CALL(ext->be->arc_table(ext->ctx, &ext->cur_arcs, &n), "arc_table");
		q->n_arc = (int32_t)n;
	}
	// read where the backend lands it (its pinned staging area: fetch_later + one wait), not out of a zeroed vector the table was copied
	// into -- configs[1]'s 46 k arcs are 2.2 MB, just beyond what pga_fetch stages, and went to pageable memory by the runtime's slow path
	const pga_arc_part_t *part = nullptr;
	const size_t n_part = (size_t)q->n_arc;
	if (n_part) {
		const void *view = nullptr;
		BE_CALL(ext->be->fetch_later(ext->ctx, ext->cur_arcs, sizeof(pga_arc_part_t) * n_part, &view), "fetch_later");
		BE_CALL(ext->be->sync(ext->ctx), "sync");
		part = (const pga_arc_part_t *)view; // LIFETIME: the backend's staging area -- valid until the next fetch_later (pangene_hip.h).  The conversion loop below
		// makes NO backend call; whoever adds one has to copy the table out first.
	}
	if ((int64_t)n_part > q->m_arc) {
		q->m_arc = (int32_t)n_part + ((int32_t)n_part >> 1) + 16;
		q->arc = (pg_arc_t *)std::realloc(q->arc, sizeof(pg_arc_t) * (size_t)q->m_arc);
	}
	for (size_t i = 0; i < n_part; ++i) {
		pg_arc_t *p = &q->arc[i];
		std::memset(p, 0, sizeof(*p));
		p->x = part[i].x, p->n_genome = part[i].n_genome, p->tot_cnt = part[i].tot_cnt;
		if (!ext->seg_renumber.empty()) { // the table of rounds that were queued to the end: segment numbers of before the deletions (monotone map: the order stands)
			const uint32_t v = (uint32_t)(p->x >> 32), w = (uint32_t)p->x;
			p->x = (uint64_t)((uint32_t)ext->seg_renumber[v >> 1] << 1 | (v & 1u)) << 32 | ((uint32_t)ext->seg_renumber[w >> 1] << 1 | (w & 1u));
		}
		p->avg_dist = (int32_t)(int64_t)((double)(int64_t)part[i].sum_dist / part[i].tot_cnt + .499);
		p->s1 = (int32_t)((double)part[i].sum_s1 / part[i].n_genome + .499);
		p->s2 = (int32_t)((double)part[i].sum_s2 / part[i].n_genome + .499);
	}
	return 0;
}

// pg_graph_flag_vtx + PG_SET_FILTER(vtx == 0): they always come as a pair (graph.c:287-288,295,312)
static int flag_vtx(pg_graph_t *q, DataExt *ext) { return ext->be->flag_vtx(ext->ctx, q->g2s, q->n_seg, 1); }

// pg_flt_high_occ + pg_hard_delete (graph.c:219-263)
static int flt_high_occ(int32_t max_avg_occ, int32_t max_degree, int32_t max_dist_loci, pg_graph_t *q, DataExt *ext)
{
	Phase ph(PH_FLT);
	int32_t n_high_occ = 0, n_high_deg = 0, n_high_loci = 0;
	for (int32_t i = 0; i < q->n_seg; ++i)
		if (q->seg[i].tot_cnt > max_avg_occ * q->d->n_genome) q->seg[i].del = 1, ++n_high_occ;
	for (int32_t v = 0; v < 2 * q->n_seg; ++v) // out-degree of every oriented vertex of the round's arc table (graph.c:243-250)
		if (ext->deg[(size_t)v] > max_degree && !q->seg[v >> 1].del) q->seg[v >> 1].del = 1, ++n_high_deg;
	for (int32_t i = 0; i < q->n_seg; ++i) {
		pg_seg_t *s = &q->seg[i];
		int32_t m = s->n_dist_loci[0] > s->n_dist_loci[1] ? s->n_dist_loci[0] : s->n_dist_loci[1];
		if (m > max_dist_loci && !s->del) s->del = 1, ++n_high_loci;
	}
	if (pg_verbose >= 3)
		std::fprintf(stderr, "[M::%s::%s] filtered %d high-occurrence segments, %d high-degree segments and %d segments connecting distant loci\n",
		             "pg_flt_high_occ", stamp(), n_high_occ, n_high_deg, n_high_loci);
	int32_t k = 0;
	for (int32_t i = 0; i < q->n_seg; ++i) {
		if (!q->seg[i].del) q->seg[k++] = q->seg[i];
		else if (pg_verbose >= 3)
			std::fprintf(stderr, "#del\t%s\tavg_occ=%.1f\tdist_deg=%d,%d\n", q->d->gene[q->seg[i].gid].name,
			             (double)q->seg[i].tot_cnt / q->d->n_genome, q->seg[i].n_dist_loci[0], q->seg[i].n_dist_loci[1]);
	}
	q->n_seg = k;
	gen_g2s(q);
	return flag_vtx(q, ext);
}

// pg_mark_branch_flt_arc (branch.c:48-106).  The reference calls pg_n_local (O(#genomes)) once per candidate gene
// pair; here the backend enumerates all pairs of the round from the arc table, counts them over the local genomes
// in one launch, the counts are all-reduced, and a second launch applies the marking logic per vertex.
static int mark_branch_flt_arc(const pg_opt_t *opt, pg_graph_t *q, DataExt *ext)
{
	const pga_backend_t *be = ext->be;
	Phase ph_all(PH_BRANCH_HOST);
	std::vector<int32_t> ndl((size_t)q->n_seg * 2 + 1);
	int64_t n_flt1 = 0, n_flt2 = 0;
	for (int attempt = 0;; ++attempt) {
		BE_CALL(be->rep_pos(ext->ctx), "rep_pos");
		int32_t *b_cnt; int64_t np = 0;
		Phase ph(PH_NLOCAL);
		BE_CALL(be->branch_pairs(ext->ctx, nullptr, nullptr, 0, nullptr, q->n_seg, opt->branch_diff, opt->local_dist, opt->local_count,
		                         !!(opt->flag & PG_F_FRAG_MODE), &b_cnt, sharded() ? &np : nullptr), "branch_pairs"); // the pair count only matters to the all-reduce
		BE_CALL(xreduce(be, ext->ctx, b_cnt, np, PG_X_I32, PG_X_SUM), "allreduce(n_local)");
		// per-arc weak_br stays resident for mark_hits; the two totals only feed the log line
		BE_CALL(be->branch_decide(ext->ctx, opt->branch_diff, opt->branch_diff_dist, opt->branch_diff_cut, nullptr, ndl.data(), pg_verbose >= 3 ? &n_flt1 : nullptr, pg_verbose >= 3 ? &n_flt2 : nullptr), "branch_decide");
		g_phase[PH_BRANCH_HOST] -= now_sec() - ph.t0; // counted under PH_NLOCAL
		// branch_decide has waited: a round that was left running behind this step is over too -- its results are collected now,
		// before anything touches the hits.  Had it to be repeated, so has this step (it read the discarded table).
		const int rc = arc_collect(opt, q, ext);
		if (rc < 0) return rc;
		if (rc == 0 || attempt) break;
	}
	for (int32_t j = 0; j < q->n_seg; ++j) q->seg[j].n_dist_loci[0] = ndl[(size_t)j * 2], q->seg[j].n_dist_loci[1] = ndl[(size_t)j * 2 + 1];
	if (pg_verbose >= 3)
		std::fprintf(stderr, "[M::%s::%s] marked %ld locally diverged branches and %ld distantly diverged branches\n", "pg_mark_branch_flt_arc", stamp(), (long)n_flt1, (long)n_flt2);
	return 0;
}

// The same step for the common case -- not sharded, no log lines, the round's arc table still waiting to be collected -- with
// pg_flt_high_occ's tests made on the backend as well (branch_decide_filter): what comes back is one byte per segment instead
// of the round's counters, degrees and n_dist_loci.  *done = false: the preconditions did not hold, or the arc round had to be
// repeated -- the caller takes the general route (mark_branch_flt_arc + flt_high_occ) for this round.
static int mark_branch_flt_arc_fast(const pg_opt_t *opt, pg_graph_t *q, DataExt *ext, bool do_filter, int32_t max_tot_cnt, int32_t max_degree, int32_t max_dist_loci,
                                    std::vector<uint8_t> &del, bool *done)
{
	const pga_backend_t *be = ext->be;
	*done = false;
	static const bool host_only = std::getenv("PANGENE_ROUND_FILTER_HOST") != nullptr; // (tests: keep the general route exercised)
	if (host_only || sharded() || route_v(ext) >= 3 || be->branch_decide_filter == nullptr || !ext->arc_pending) return 0;
	const int32_t S = q->n_seg;
	{
		Phase ph(PH_NLOCAL);
		BE_CALL(be->rep_pos(ext->ctx), "rep_pos");
		int32_t *b_cnt;
		BE_CALL(be->branch_pairs(ext->ctx, nullptr, nullptr, 0, nullptr, S, opt->branch_diff, opt->local_dist, opt->local_count, !!(opt->flag & PG_F_FRAG_MODE), &b_cnt, nullptr), "branch_pairs");
		del.resize((size_t)S + 1);
		const int rc = be->branch_decide_filter(ext->ctx, opt->branch_diff, opt->branch_diff_dist, opt->branch_diff_cut, do_filter ? 1 : 0, max_tot_cnt, max_degree, max_dist_loci, del.data());
		if (rc == 2) return 0; // not behind a deferred round of the gene-major path
		if (rc != 0) { set_error(rc, "branch_decide_filter"); return rc; }
	}
	Phase ph(PH_BRANCH_HOST);
	ext->arc_pending = false;
	const int rc = be->arc_round_finish(ext->ctx, S, nullptr, nullptr); // the wait is over: does the round stand?
	if (rc < 0) { set_error(rc, "arc_round_finish"); return rc; }
	if (rc == 1) { // a hub gene overflowed the per-gene table: the round is repeated on the sort path, the branch step on the general route
		std::vector<int32_t> sc((size_t)S * 2 + 1);
		ext->deg.assign((size_t)S * 2 + 1, 0);
		BE_CALL(be->arc_round_local(ext->ctx, !!(opt->flag & PG_F_ORI_FOR_BRANCH), S, sc.data(), ext->deg.data()), "arc_round");
		for (int32_t i = 0; i < S; ++i) q->seg[i].n_genome = sc[(size_t)i], q->seg[i].tot_cnt = sc[(size_t)S + (size_t)i];
		return 0;
	}
	*done = true;
	return 0;
}

// pg_flt_high_occ + pg_hard_delete with the verdicts of branch_decide_filter
static int apply_round_filter(pg_graph_t *q, DataExt *ext, const std::vector<uint8_t> &del)
{
	Phase ph(PH_FLT);
	int32_t k = 0;
	for (int32_t i = 0; i < q->n_seg; ++i)
		if (!del[(size_t)i]) q->seg[k++] = q->seg[i];
	q->n_seg = k;
	gen_g2s(q);
	return flag_vtx(q, ext);
}

// Rounds 0 .. R-1 of the branch filter (graph.c:300-314) queued on the backend in one go, one wait at the end (pga_branch_loop):
// the common case -- not sharded, no log lines, no contig whose order is replayed in full.  *done = false: not applicable
// (nothing happened).  Returns RC_REDO when the queued rounds met something only the host-driven rounds can handle: the
// shard's state is undefined then and pg_graph_gen repeats the run.
enum { RC_REDO = 1000 };
// the two collectives of the sharded form: ordered with the backend's stream when they return (see pga_loop_xchg_t)
static int loop_allreduce(void *user, void *buf, int64_t count)
{
	DataExt *ext = (DataExt *)user;
	if (count == 0) return 0;
	const int rc = xready(ext->be, ext->ctx);
	++g_n_coll;
	return rc ? rc : g_xchg.allreduce(g_xchg.user, buf, count, PG_X_I32, PG_X_SUM, ext->be->is_device());
}
static int loop_allgather(void *user, const void *in, void *out, int64_t bytes)
{
	DataExt *ext = (DataExt *)user;
	const int rc = xready(ext->be, ext->ctx);
	++g_n_coll;
	return rc ? rc : g_xchg.allgather(g_xchg.user, in, out, bytes, ext->be->is_device());
}

// pre: behind graph 1's deferred arc round, graph 2 included (pga_branch_par_t::pre_on); fin: R = all the rounds, and the arc round of the
// graph that is written is queued too (final_on) -- the segments' public fields and the renumbering map for fetch_arcs are left here
static int branch_loop_fast(const pg_opt_t *opt, pg_graph_t *q, DataExt *ext, int32_t R, bool *done, bool pre = false, bool fin = false)
{
	*done = false;
	const pga_backend_t *be = ext->be;
	const bool shd = sharded();
	if (be->branch_loop == nullptr || route_v(ext) >= 3 || trace_path() != nullptr || (!shd && !ext->arc_pending) || R < 1 || ext->no_branch_loop) return 0;
	static const bool no_x = std::getenv("PANGENE_SHARDED_LOOP_HOST") != nullptr; // (tests: the host-driven rounds of a sharded run)
	if (shd && (no_x || !be->is_device())) return 0;
	if (shd && ext->skip_loop_once) { ext->skip_loop_once = false; return 0; } // the repeated run after status 3
	if (pre && shd && !ext->arc_via_x) return 0; // (sharded: graph 2's tests read the GLOBAL segment counters in device memory: only pga_arc_round_x leaves those)
	const int n_sorts = 2 * R - 1 + (pre ? 1 : 0) + (fin ? 1 : 0); // of each kind: one pair per pg_mark_branch_flt_hit (branch.c:116,140), one per pg_gen_arc (graph.c:103,123)
	static const bool dbg = std::getenv("PANGENE_TIMING") != nullptr;
	bool quiet;
	{ Phase ph(PH_EXACT); quiet = exact_quiet(ext, n_sorts); }
	if (shd) { // every rank queues the rounds, or none does: the ranks vote (a rank without hits or segments cannot; one whose hit order needs the host neither)
		int32_t ok = (quiet && ext->n_hit_local > 0 && q->n_seg > 0) ? 1 : 0;
		void *scr;
		BE_CALL(be->scratch(ext->ctx, 16, &scr), "scratch");
		BE_CALL(be->put(ext->ctx, scr, &ok, sizeof(ok)), "put");
		BE_CALL(xreduce(be, ext->ctx, scr, 1, PG_X_I32, PG_X_SUM), "allreduce(vote)");
		BE_CALL(be->fetch(ext->ctx, &ok, scr, sizeof(ok)), "fetch");
		if (ok != g_xchg.world) { if (dbg) std::fprintf(stderr, "[branch_loop] %d of %d ranks can queue their rounds: host-driven rounds\n", ok, g_xchg.world); return 0; }
	}
	else if (!quiet) { if (dbg) std::fprintf(stderr, "[branch_loop] not quiet: the hit at array index 0 of some genome changes within the next %d sorts\n", n_sorts); return 0; }
	const int32_t S = q->n_seg, n = opt->n_branch_flt;
	std::vector<int32_t> m_tot((size_t)R), m_deg((size_t)R), m_loci((size_t)R);
	for (int32_t i = 0; i < R; ++i) { // graph.c:303-306
		const double r = 1.0 + (double)(n - 1 - i) / n;
		m_tot[(size_t)i] = (int32_t)(opt->max_avg_occ * r + .499) * q->d->n_genome;
		m_deg[(size_t)i] = (int32_t)(opt->max_degree * r + .499);
		m_loci[(size_t)i] = (int32_t)(opt->max_dist_loci * r + .499);
	}
	pga_branch_par_t par;
	par.branch_diff = opt->branch_diff, par.branch_diff_dist = opt->branch_diff_dist, par.branch_diff_cut = opt->branch_diff_cut;
	par.local_dist = opt->local_dist, par.local_count = opt->local_count, par.frag_mode = !!(opt->flag & PG_F_FRAG_MODE), par.use_ori = !!(opt->flag & PG_F_ORI_FOR_BRANCH);
	par.pre_on = pre ? 1 : 0; // graph.c:294: pg_flt_high_occ(q, max_avg_occ * 2, max_degree * 2, max_dist_loci)
	par.pre_max_tot_cnt = opt->max_avg_occ * 2 * q->d->n_genome, par.pre_max_degree = opt->max_degree * 2, par.pre_max_dist_loci = opt->max_dist_loci;
	par.final_on = fin ? 1 : 0;
	std::vector<uint8_t> &alive = ext->del_buf;
	alive.assign((size_t)S + 1, 1);
	std::vector<int32_t> fin_sc, fin_ndl;
	if (fin) fin_sc.assign((size_t)S * 2 + 1, 0), fin_ndl.assign((size_t)S * 2 + 1, 0);
	pga_loop_xchg_t lx;
	lx.user = ext, lx.rank = g_xchg.rank, lx.world = g_xchg.world, lx.arc_cap_hint = ext->x_arc_slot, lx.allreduce_i32_sum = loop_allreduce, lx.allgather = loop_allgather;
	int rc;
	{ Phase ph(PH_NLOCAL); rc = be->branch_loop(ext->ctx, R, &par, m_tot.data(), m_deg.data(), m_loci.data(), alive.data(), shd ? &lx : nullptr, fin ? fin_sc.data() : nullptr, fin ? fin_ndl.data() : nullptr); }
	if (dbg) std::fprintf(stderr, "[branch_loop] %d rounds queued%s: backend status %d\n", R, shd ? " (sharded)" : "", rc);
	if (rc == 2) return 0;
	if (rc == 1) { ext->no_branch_loop = true; return RC_REDO; }
	if (rc == 3) { ext->skip_loop_once = true; return RC_REDO; }
	if (rc == 4) { if (++ext->loop_room_retries > 2) ext->no_branch_loop = true; return RC_REDO; } // a pair list beyond its capacity: the backend has made room, the queue runs again
	if (rc != 0) { set_error(rc, "branch_loop"); return rc; }
	ext->arc_pending = false;
	exact_skip(ext, n_sorts);
	Phase ph(PH_FLT);
	int32_t k = 0;
	if (fin) ext->seg_renumber.assign((size_t)S, -1);
	for (int32_t i = 0; i < S; ++i)
		if (alive[(size_t)i]) {
			if (fin) { // the public fields of the graph that is written (graph.c:125-126 of its arc round, branch.c:90 of the last branch step)
				pg_seg_t *sg = &q->seg[i];
				sg->n_genome = fin_sc[(size_t)i], sg->tot_cnt = fin_sc[(size_t)S + (size_t)i];
				sg->n_dist_loci[0] = fin_ndl[(size_t)i * 2], sg->n_dist_loci[1] = fin_ndl[(size_t)i * 2 + 1];
				ext->seg_renumber[(size_t)i] = k;
			}
			q->seg[k++] = q->seg[i];
		}
	q->n_seg = k;
	gen_g2s(q);
	if (!fin) BE_CALL(flag_vtx(q, ext), "flag_vtx"); // the renumbered g2s (the hits' flags do not change: the loop filtered them already)
	// (fin: the backend's table is in the old numbering -- it gets the new g2s when the table has been fetched: graph_gen_impl)
	*done = true;
	return 0;
}

static int mark_branch_flt_hit(pg_graph_t *q, DataExt *ext) // branch.c:108-145; the arcs and their weak_br are already resident
{
	int64_t n = 0;
	{ Phase ph(PH_EXACT); BE_CALL(exact_sort(ext, 1), "override_order"); } // branch.c:116
	{ Phase ph(PH_MARK_HITS); BE_CALL(ext->be->mark_hits(ext->ctx, nullptr, nullptr, q->n_arc, pg_verbose >= 3 ? &n : nullptr, 1), "mark_hits"); }
	{ Phase ph(PH_EXACT); BE_CALL(exact_sort(ext, 0), "override_order"); } // branch.c:140
	if (pg_verbose >= 3)
		std::fprintf(stderr, "[M::%s::%s] marked %ld diverged hits\n", "pg_mark_branch_flt_hit", stamp(), (long)n);
	return 0;
}

static int graph_gen_impl(const pg_opt_t *opt, pg_graph_t *q)
{
	DataExt *ext = ext_of(q->d, false);
	if (ext == nullptr || ext->ctx == nullptr) { set_error(PGA_ERR_ARG, "pg_graph_gen: pg_post_process has not run"); return PGA_ERR_ARG; }
	const pga_backend_t *be = ext->be;
	pga_ctx_t *ctx = ext->ctx;
	ext->arc_pending = false;
	// graph 1: initial vertices (graph.c:284-291)
	BE_CALL(be->set_filter(ctx, PGA_FLT_PSEUDO), "set_filter");
	BE_CALL(gen_vtx(opt, q, ext), "gen_vtx");
	if (!exact_early(ext)) { Phase ph(PH_EXACT); BE_CALL(exact_sort(ext, 0), "set_head"); } // index 0 of the S1 order, needed from the first sweep of stage C on (see post_process_impl)
	BE_CALL(flag_vtx(q, ext), "flag_vtx");
	BE_CALL(trace_state(ext, "gen_vtx+flag_vtx", 0), "trace");
	// Graphs 2 and 3 as ONE queue when the backend can (pga_branch_loop with its pre-step): graph 1's arc round is left running, the
	// loop starts with graph 2's pg_flt_high_occ + pg_gen_arc and goes on with the branch rounds -- no wait between graph 1 and round n-2.
	static const bool no_pre = env_word("PANGENE_LOOP", "nopre"); // (tests: graph 2 host-driven in front of the queued rounds)
	static const bool no_fin = env_word("PANGENE_LOOP", "nofinal"); // (tests: the last round host-driven behind the queued ones)
	// (sharded, round 6: the same one queue, behind a graph-1 round that came through pga_arc_round_x -- from the second run over a data set on, when the slot capacity is known)
	const bool try_pre = !no_pre && opt->n_branch_flt >= 2 && ext->be->branch_loop != nullptr && route_v(ext) < 3 && trace_path() == nullptr && !ext->no_branch_loop &&
	                     (!sharded() || (ext->be->arc_round_x != nullptr && ext->be->is_device() && ext->x_arc_slot > 0 && std::getenv("PANGENE_SHARDED_LOOP_HOST") == nullptr));
	BE_CALL(gen_arc(opt, q, ext, try_pre), "gen_arc");
	BE_CALL(trace_state(ext, "gen_arc", 1), "trace");
	if (pg_verbose >= 3) std::fprintf(stderr, "[M::%s::%s] round-1 graph: %d genes and %d arcs\n", "pg_graph_gen", stamp(), q->n_seg, q->n_arc);
	int32_t i_first = 0;
	bool queued = false;
	bool queued_all = false;
	ext->seg_renumber.clear();
	if (try_pre) {
		if (!no_fin) { // every round and the arc round of the graph that is written
			const int rc = branch_loop_fast(opt, q, ext, opt->n_branch_flt, &queued_all, true, true);
			if (rc) return rc;
			queued = queued_all;
		}
		const int rc = queued ? 0 : branch_loop_fast(opt, q, ext, opt->n_branch_flt - 1, &queued, true);
		if (rc) return rc;
		if (!queued) { // not applicable after all (the order replay needs the host, ...): graph 1's results the usual way
			const int rc2 = arc_collect(opt, q, ext);
			if (rc2 < 0) return rc2;
		}
	}
	if (!queued) {
		// graph 2: after removing high-occurrence vertices (graph.c:293-298)
		BE_CALL(flt_high_occ(opt->max_avg_occ * 2, opt->max_degree * 2, opt->max_dist_loci, q, ext), "flt_high_occ");
		BE_CALL(trace_state(ext, "flt_high_occ", 1), "trace");
		BE_CALL(gen_arc(opt, q, ext, opt->n_branch_flt > 0), "gen_arc");
		BE_CALL(trace_state(ext, "gen_arc", 2), "trace");
		if (pg_verbose >= 3) std::fprintf(stderr, "[M::%s::%s] round-2 graph: %d genes and %d arcs\n", "pg_graph_gen", stamp(), q->n_seg, q->n_arc);
		// graph 3: branch filtering (graph.c:300-315)
		if (opt->n_branch_flt >= 2) { // every round in one go when the backend can (round 6: also behind a host-driven graph 2), or all but the last one
			if (!no_fin && !no_pre) {
				const int rc = branch_loop_fast(opt, q, ext, opt->n_branch_flt, &queued_all, false, true);
				if (rc) return rc;
				queued = queued_all;
			}
			if (!queued) {
				const int rc = branch_loop_fast(opt, q, ext, opt->n_branch_flt - 1, &queued);
				if (rc) return rc;
			}
		}
	}
	if (queued_all) i_first = opt->n_branch_flt; // nothing left to drive
	else if (queued) {
		i_first = opt->n_branch_flt - 1;
		BE_CALL(gen_arc(opt, q, ext, true), "gen_arc"); // the arc round of round n-2, with the renumbered segments
	}
	for (int32_t i = i_first; i < opt->n_branch_flt; ++i) {
		double r = 1.0 + (double)(opt->n_branch_flt - 1 - i) / opt->n_branch_flt;
		int32_t max_avg_occ = (int32_t)(opt->max_avg_occ * r + .499);
		int32_t max_degree = (int32_t)(opt->max_degree * r + .499);
		int32_t max_dist_loci = (int32_t)(opt->max_dist_loci * r + .499);
		// every round but the last one leaves its bulk results on the backend when it can (the last one fills the public fields of
		// pg_seg_t: n_genome, tot_cnt, n_dist_loci)
		bool on_backend = false;
		if (i + 1 < opt->n_branch_flt)
			BE_CALL(mark_branch_flt_arc_fast(opt, q, ext, i > 0, max_avg_occ * q->d->n_genome, max_degree, max_dist_loci, ext->del_buf, &on_backend), "mark_branch_flt_arc");
		if (!on_backend) BE_CALL(mark_branch_flt_arc(opt, q, ext), "mark_branch_flt_arc");
		BE_CALL(mark_branch_flt_hit(q, ext), "mark_branch_flt_hit"); // with PG_SET_FILTER(weak_br == 2), graph.c:309
		BE_CALL(trace_state(ext, "mark_branch", i + 3), "trace");
		if (i > 0 && on_backend) BE_CALL(apply_round_filter(q, ext, ext->del_buf), "flt_high_occ");
		else if (i > 0) BE_CALL(flt_high_occ(max_avg_occ, max_degree, max_dist_loci, q, ext), "flt_high_occ"); // with PG_SET_FILTER(vtx == 0), graph.c:312
		if (i > 0) BE_CALL(trace_state(ext, "flt_high_occ", i + 3), "trace");
		BE_CALL(gen_arc(opt, q, ext, i + 1 < opt->n_branch_flt), "gen_arc");
		BE_CALL(trace_state(ext, "gen_arc", i + 3), "trace");
	}
	BE_CALL(be->set_filter(ctx, PGA_FLT_SHADOW), "set_filter"); // graph.c:316
	BE_CALL(fetch_arcs(q, ext), "fetch_arcs");
	if (queued_all) { // the table is on the host: now the backend may learn the new numbering (gene matrix, a later run)
		ext->seg_renumber.clear();
		BE_CALL(flag_vtx(q, ext), "flag_vtx");
	}
	if (opt->min_arc_cnt > 1) { // graph.c:191-200
		int32_t k = 0, n_aflt = 0;
		for (int32_t i = 0; i < q->n_arc; ++i) {
			if (q->arc[i].n_genome < opt->min_arc_cnt) { ++n_aflt; continue; }
			q->arc[k++] = q->arc[i];
		}
		q->n_arc = k;
		if (pg_verbose >= 3) std::fprintf(stderr, "[M::%s::%s] filtered %d low-occurrence arcs\n", "pg_graph_cut_low_arc", stamp(), n_aflt);
	}
	arc_index(q);
	if (pg_verbose >= 3) std::fprintf(stderr, "[M::%s::%s] round-3 graph: %d genes and %d arcs\n", "pg_graph_gen", stamp(), q->n_seg, q->n_arc);
	BE_CALL(exact_sort(ext, 1), "override_order"); // the cm order pg_write_walk will see (format.c:190)
	ext->host_stale = true;
	pga_hazard_t hz;
	if (exact_mode() == 0 && be->hazards(ctx, &hz) == 0 && (hz.h1_head_tie | hz.h2_cm_tie | hz.h3_dom_tie | hz.h2_cs_tie) && pg_verbose >= 2)
		std::fprintf(stderr, "[W::%s] tie-order hazards seen (head-tie %ld, cm-tie %ld, dominator-tie %ld, cs-tie at the local_count boundary %ld): output may differ from the reference's unstable sort order\n",
		             "pg_graph_gen", (long)hz.h1_head_tie, (long)hz.h2_cm_tie, (long)hz.h3_dom_tie, (long)hz.h2_cs_tie);
	// (What the writers need of the per-hit state -- one flt bit per hit and, once per order epoch, the two orders -- is fetched by the
	// writer that asks first, pg_write_walk / pg_write_bed / pg_write_matrix through sync_host: it is the device-side counterpart of the two
	// sorts per genome the reference does INSIDE pg_write_walk, format.c:183-225, not a step of pg_graph_gen.  Rounds 2-4 fetched it here,
	// eagerly: 16-21 ms of a 12 M-hit first pass.)
	return g_err;
}

} // namespace pgx

using namespace pgx;

extern "C" {

void pg_opt_init(pg_opt_t *opt) // defaults of option.c:9-25
{
	std::memset(opt, 0, sizeof(*opt));
	opt->gene_delim = ':';
	opt->min_prot_iden = 0.5, opt->min_prot_ratio = 0.5, opt->score_adj_coef = 2.0;
	opt->min_ov_ratio = 0.5, opt->min_vertex_ratio = 0.05;
	opt->max_avg_occ = 10, opt->max_degree = 15, opt->max_dist_loci = 3;
	opt->n_branch_flt = 15, opt->min_arc_cnt = 1;
	opt->local_dist = 2000000, opt->local_count = 10;
	opt->branch_diff = 0.02, opt->branch_diff_dist = 0.05, opt->branch_diff_cut = 0.5;
}

void pg_post_process(const pg_opt_t *opt, pg_data_t *d)
{
	g_err = 0;
	g_t_path0 = now_sec();
	post_process_impl(opt, d);
	g_path_sec = now_sec() - g_t_path0;
	DataExt *ext = ext_of(d, false);
	g_path_hits = ext ? ext->n_hit_local : 0;
	int32_t n_pref = 0;
	for (int32_t i = 0; i < d->n_gene; ++i) n_pref += d->gene[i].preferred;
	if (pg_verbose >= 3) std::fprintf(stderr, "[M::%s] there are %d preferred genes\n", __func__, n_pref);
}

pg_graph_t *pg_graph_init(pg_data_t *d) // graph.c:34-41
{
	pg_graph_t *g = (pg_graph_t *)std::calloc(1, sizeof(pg_graph_t));
	g->d = d;
	g->m_seg = d->n_gene > 0 ? d->n_gene : 1;
	g->seg = (pg_seg_t *)std::calloc((size_t)g->m_seg, sizeof(pg_seg_t));
	return g;
}

// Did a tie-order channel other than array index 0 open during the run (two walkable hits sharing
// (contig, cm); two equal-score dominators)?  Collective: every rank gets the same answer.
// Tie-order hazards of the run that just finished (mode auto).  *need: some event happened on a contig that does not follow
// the reference's exact order yet (collective answer when sharded); *give_up: the event list is incomplete somewhere.  The
// contigs concerned are added to ext->extra_ctgs.
static int hazards_review(DataExt *ext, bool *need, bool *give_up)
{
	pga_hazard_t hz;
	const double th0 = now_sec();
	BE_CALL(ext->be->hazards(ext->ctx, &hz), "hazards");
	const double th1 = now_sec();
	double th2 = th1, th3 = th1;
	int32_t flags[2] = { 0, 0 }; // {need, give up}
	if (hz.h2_cm_tie + hz.h3_dom_tie + hz.h2_cs_tie > 0) {
		std::vector<int32_t> segs(PGA_HAZARD_CAP);
		int64_t n_total = 0;
		th2 = now_sec();
		BE_CALL(ext->be->hazard_segs(ext->ctx, segs.data(), (int32_t)segs.size(), &n_total), "hazard_segs");
		th3 = now_sec();
		const int64_t n_got = std::min<int64_t>(n_total, (int64_t)segs.size());
		if (n_total > n_got) flags[0] = flags[1] = 1;
		// contig-segment id -> (local genome, contig): the shard lists the contigs genome-major
		std::vector<int32_t> base(ext->local_genomes.size() + 1, 0);
		for (size_t k = 0; k < ext->local_genomes.size(); ++k) base[k + 1] = base[k] + (k < ext->n_vctg.size() ? ext->n_vctg[k] : ext->q_d->genome[ext->local_genomes[k]].n_ctg);
		std::sort(segs.begin(), segs.begin() + n_got);
		int64_t n_new = 0;
		for (int64_t i = 0; i < n_got; ++i) {
			if (i && segs[(size_t)i] == segs[(size_t)i - 1]) continue;
			const size_t k = (size_t)(std::upper_bound(base.begin(), base.end(), segs[(size_t)i]) - base.begin()) - 1;
			int32_t ctg_local = segs[(size_t)i] - base[k];
			if (k < ext->vreal.size() && !ext->vreal[k].empty() && (size_t)ctg_local < ext->vreal[k].size()) ctg_local = ext->vreal[k][(size_t)ctg_local]; // a piece of a virtual contig -> the contig
			const std::pair<int32_t, int32_t> gc((int32_t)k, ctg_local);
			if (std::binary_search(ext->static_ctgs.begin(), ext->static_ctgs.end(), gc)) continue; // follows the exact order already (static prediction)
			auto it = std::lower_bound(ext->extra_ctgs.begin(), ext->extra_ctgs.end(), gc);
			if (it == ext->extra_ctgs.end() || *it != gc) ext->extra_ctgs.insert(it, gc), ++n_new;
			if (std::getenv("PANGENE_TIMING")) std::fprintf(stderr, "[hazard] local genome %d contig %d\n", gc.first, gc.second);
		}
		if (n_new) flags[0] = 1;
		if (pg_verbose >= 2)
			std::fprintf(stderr, "[M::%s::%s] tie-order hazards on this rank: %ld equal-cm neighbours, %ld equal-key dominators, %ld order-dependent local_count tests, on %ld contig(s) not yet following the exact order\n",
			             "pg_graph_gen", stamp(), (long)hz.h2_cm_tie, (long)hz.h3_dom_tie, (long)hz.h2_cs_tie, (long)n_new);
	}
	if (sharded()) {
		void *scr;
		BE_CALL(ext->be->scratch(ext->ctx, 16, &scr), "scratch");
		BE_CALL(ext->be->put(ext->ctx, scr, flags, sizeof(flags)), "put");
		BE_CALL(xreduce(ext->be, ext->ctx, scr, 2, PG_X_I32, PG_X_MAX), "allreduce(hazard)");
		BE_CALL(ext->be->fetch(ext->ctx, flags, scr, sizeof(flags)), "fetch");
	}
	*need = flags[0] != 0, *give_up = flags[1] != 0;
	if (std::getenv("PANGENE_TIMING")) std::fprintf(stderr, "[hazards_review] counters %.3f ms, list set-up %.3f ms, list fetch %.3f ms, rest %.3f ms (cm %ld dom %ld cs %ld)\n",
	                                               (th1 - th0) * 1e3, (th2 - th1) * 1e3, (th3 - th2) * 1e3, (now_sec() - th3) * 1e3, (long)hz.h2_cm_tie, (long)hz.h3_dom_tie, (long)hz.h2_cs_tie);
	return 0;
}

void pg_graph_gen(const pg_opt_t *opt, pg_graph_t *q)
{
	double t = now_sec();
	DataExt *ext = ext_of(q->d, false);
	if (ext) ext->q_d = q->d;
	auto run_graph = [&]() { // graph_gen_impl; when the queued branch rounds gave up, once more from stage A with host-driven rounds
		int rc = graph_gen_impl(opt, q);
		if (rc == RC_REDO && ext) {
			if (pg_verbose >= 2) std::fprintf(stderr, "[M::%s::%s] the queued branch rounds met a case they leave to the host: repeating stages A-C with host-driven rounds\n", "pg_graph_gen", stamp());
			q->n_seg = 0, q->n_arc = 0;
			std::memset((void *)q->seg, 0, sizeof(pg_seg_t) * (size_t)q->m_seg);
			ext->rerun = true;
			rc = post_process_impl(opt, q->d);
			if (rc == 0) rc = graph_gen_impl(opt, q);
		}
		return rc;
	};
	if (g_err == 0 && run_graph() != 0) q->n_arc = 0;
	const double t_first = now_sec() - t;
	int n_attempt = 1;
	// Mode auto: the canonical order provably gives the reference's result unless a tie-order hazard occurred.  Where one
	// did, the contigs concerned get the reference's exact order (replayed on the host) and stages A-C are repeated on the
	// resident shard; hazards that then only occur on such contigs are harmless.  After three attempts, or when the event
	// list overflowed, every contig is tracked (mode "all").  The decision is collective when sharded.
	double t_review = 0.0;
	const double path_before = g_path_sec; // what pg_post_process measured
	for (int attempt = 0; g_err == 0 && ext && exact_mode() == 1; ++attempt) {
		bool need = false, give_up = false;
		const double tr0 = now_sec();
		const int rc_review = hazards_review(ext, &need, &give_up);
		t_review += now_sec() - tr0;
		if (rc_review != 0 || !need) break;
		// Only the contigs on which the ties occurred get the reference's exact order (the event list covers every channel of
		// equal cm of walkable neighbours, equal-key dominators / sub-optimal isoforms, cs ties that decide a
		// local_count test); the repeated run is reviewed again, and ties that then only occur on tracked contigs are harmless.
		// PANGENE_ESCALATE_ALL=1 tracks every contig at once instead (the round-1 behaviour).
		const bool k_selective = std::getenv("PANGENE_ESCALATE_ALL") == nullptr;
		const bool all = !k_selective || give_up || attempt >= 2;
		if (pg_verbose >= 2)
			std::fprintf(stderr, "[M::%s::%s] repeating stages A-C with the reference's exact hit order on %s\n", __func__, stamp(),
			             all ? "every contig" : "the contigs where the ties occurred");
		if (all) exact_override(2);
		exact_shutdown(ext);
		exact_init(q->d, ext), ext->exact_mode_of_segs = exact_mode();
		q->n_seg = 0, q->n_arc = 0;
		std::memset((void *)q->seg, 0, sizeof(pg_seg_t) * (size_t)q->m_seg);
		ext->rerun = true;
		++n_attempt;
		if (post_process_impl(opt, q->d) != 0 || run_graph() != 0) q->n_arc = 0;
		if (all) { exact_override(-1); break; }
	}
	if (ext && (!ext->extra_ctgs.empty() || ext->exact_mode_of_segs != exact_mode())) { // back to the cheap tracking for a later rerun
		ext->extra_ctgs.clear();
		exact_shutdown(ext);
		exact_init(q->d, ext), ext->exact_mode_of_segs = exact_mode();
	}
	if (ext && !ext->vtx_sel_text.empty()) {
		std::fwrite(ext->vtx_sel_text.data(), 1, ext->vtx_sel_text.size(), out_stream());
		ext->vtx_sel_text.clear();
	}
	g_path_sec += now_sec() - t;
	g_attempts = n_attempt;
	if (std::getenv("PANGENE_TIMING")) {
		std::fprintf(stderr, "[phases]");
		for (int i = 0; i < PH_COUNT; ++i) std::fprintf(stderr, " %s %.2f", pg_phase_name(i), g_phase[i] * 1e3);
		std::fprintf(stderr, " | path %.2f ms (%d attempt%s; pg_post_process %.2f, pg_graph_gen's first attempt %.2f, hazard review %.2f ms)\n", g_path_sec * 1e3, n_attempt, n_attempt > 1 ? "s: tie-order hazards" : "", path_before * 1e3, t_first * 1e3, t_review * 1e3);
		if (ext && ext->ov_calls) std::fprintf(stderr, "[exact_sort] %lld order override(s) of %lld hits in all: lists %.2f ms, backend (copy + kernels + wait) %.2f ms\n", (long long)ext->ov_calls, (long long)ext->ov_hits, ext->ov_list_s * 1e3, ext->ov_backend_s * 1e3);
		if (ext) ext->ov_calls = ext->ov_hits = 0, ext->ov_list_s = ext->ov_backend_s = 0;
	}
}

void pg_graph_destroy(pg_graph_t *q) // graph.c:43-47
{
	if (q == nullptr) return;
	std::free(q->g2s); std::free(q->seg); std::free(q->arc); std::free(q->idx);
	std::free(q);
}

int pg_last_error(void) { return g_err; }
const char *pg_last_error_str(void) { return g_errstr; }
double pg_last_path_seconds(void) { return g_path_sec; }
double pg_last_upload_seconds(void) { return g_upload_sec; }
double pg_last_pack_seconds(void) { return g_pack_sec; }

int pg_backend_is_device(void) { return backend_default()->is_device(); }
int pg_set_device(int32_t device) { const pga_backend_t *be = backend_default(); return be->set_device ? be->set_device(device) : 0; }
int pg_device_count(void) { const pga_backend_t *be = backend_default(); return be->device_count ? be->device_count() : 0; }

int pg_device_warm(void) { const pga_backend_t *be = backend_default(); return be->warm ? be->warm() : 0; }

double pg_device_copy_gbps(size_t bytes, int32_t reps) // < 0: no device / not supported by the backend
{
	const pga_backend_t *be = backend_default();
	double g = -1.0;
	if (be->copy_gbps == nullptr || be->copy_gbps(bytes, reps, &g) != 0) return -1.0;
	return g;
}

void pg_trim_host_cache(size_t keep_bytes) { trim_host_caches(keep_bytes); } // page-locked memory the library keeps for its next upload

int pg_shard_counts(const pg_data_t *d, int64_t *n_hit, int64_t *n_exon)
{
	DataExt *ext = ext_of(d, false);
	if (ext == nullptr) return PGA_ERR_ARG;
	int64_t h = 0, e = 0;
	for (int32_t j : ext->local_genomes) h += d->genome[j].n_hit, e += d->genome[j].n_exon;
	if (n_hit) *n_hit = h;
	if (n_exon) *n_exon = e;
	return 0;
}

int pg_sync_host(pg_data_t *d) { return sync_host(d, true); } // refresh every per-hit field of the host records

int pg_phase_times(double *out, int n) // seconds per driver phase of the last run; returns the number of phases
{
	for (int i = 0; i < n && i < PH_COUNT; ++i) out[i] = g_phase[i];
	return PH_COUNT;
}

const char *pg_phase_name(int i)
{
	static const char *nm[PH_COUNT] = { "begin(sort)", "exact_order(host)", "ingest", "post", "vtx", "arc_round(device)", "arc_merge(host)",
	                                    "branch_mark(host)", "n_local", "mark_hits", "flt_high_occ", "sync_host(download)" };
	return i >= 0 && i < PH_COUNT ? nm[i] : "?";
}

int pg_rerun_resident(pg_data_t *d) // next pg_post_process restarts on the shard already in HBM
{
	DataExt *ext = ext_of(d, false);
	if (ext == nullptr || ext->ctx == nullptr) return PGA_ERR_ARG;
	ext->rerun = true;
	return 0;
}

int pg_kernel_timing(pg_data_t *d, int32_t which, double *total_ms, int64_t *n_launch, int64_t *units)
{
	DataExt *ext = ext_of(d, false);
	if (ext == nullptr || ext->ctx == nullptr || ext->be->timing_get == nullptr) return PGA_ERR_ARG;
	return ext->be->timing_get(ext->ctx, which, total_ms, n_launch, units);
}

int64_t pg_collective_count(void) { return g_n_coll; }

int pg_kernel_timing_reset(pg_data_t *d)
{
	DataExt *ext = ext_of(d, false);
	if (ext == nullptr || ext->ctx == nullptr || ext->be->timing_reset == nullptr) return PGA_ERR_ARG;
	return ext->be->timing_reset(ext->ctx);
}
int64_t pg_last_path_hits(void) { return g_path_hits; }
int pg_last_attempts(void) { return g_attempts; }

// the reference's timers (sys.c:117-140; pgpriv.h): main.c:117,149 calls them, so a main.c built on top of this library links
double pg_realtime(void)
{
	static double t0 = -1.0;
	const double t = now_sec();
	if (t0 < 0) t0 = t;
	return t - t0;
}

double pg_cputime(void)
{
	struct rusage r;
	getrusage(RUSAGE_SELF, &r);
	return r.ru_utime.tv_sec + r.ru_stime.tv_sec + 1e-6 * (r.ru_utime.tv_usec + r.ru_stime.tv_usec);
}

long pg_peakrss(void)
{
	struct rusage r;
	getrusage(RUSAGE_SELF, &r);
	return r.ru_maxrss * 1024L; // Linux reports kilobytes
}

void pg_set_exchange(const pg_exchange_t *x)
{
	if (x) g_xchg = *x, g_has_xchg = true;
	else g_has_xchg = false;
}

} // extern "C"
