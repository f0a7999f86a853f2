// exact_order.cpp -- reproduces, on the host, the array order the reference's UNSTABLE radix sort
// (ksort.h:52-87 via pg_hit_sort, hit.c:29-64) leaves inside contig segments, for the segments where
// that order can reach the output, and hands it to the backend as an order override.
//
// Why: pg_shadow never resets hit[0] of a genome (overlap.c:108), so WHICH hit of
// the first tie group sits at index 0 decides that hit's shadow flag in every later round; with paralog
// families (several proteins aligned to identical coordinates) this changes the GFA.  The hit order is a
// function of the static keys only -- file order --cs--> S1 --cm--> C1 --cs--> S2 ... -- so it can be
// replayed exactly without knowing any flag: one O(n) flag pass per level and sort, per tracked segment,
// until the sequence reaches its fixed point (normally after two or three sorts; the reference repeats
// all 67).  The device keeps doing all the per-hit work; only the tie order inside tracked segments comes
// from here.
//
// Modes (env PANGENE_EXACT or pg_set_exact_mode): "auto" (default) tracks the first non-empty contig of
// each genome when its leading cs tie group has >= 2 hits (the index-0 channel); "all" tracks every contig
// (then even --bed line order equals the reference's); "off" keeps the canonical stable order everywhere.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <climits>
#include <thread>
#include "pg_internal.hpp"
#include "ksort_exact.hpp"

namespace pgx {

static int g_exact_mode = -1; // -1: read the environment
static int g_exact_override = -1; // set while a run is being repeated with every contig tracked (hazard escalation)

void exact_override(int m) { g_exact_override = m; }

int exact_mode()
{
	if (g_exact_override >= 0) return g_exact_override;
	if (g_exact_mode < 0) {
		const char *e = std::getenv("PANGENE_EXACT");
		g_exact_mode = (e == nullptr || std::strcmp(e, "auto") == 0) ? 1 : std::strcmp(e, "all") == 0 ? 2 : std::strcmp(e, "off") == 0 ? 0 : 1;
	}
	return g_exact_mode;
}
void set_exact_mode(int m) { g_exact_mode = m; }

// ---------------------------------------------------------------------------------------------
// Static tie prediction (mode auto).  Apart from array index 0 (always tracked) the reference's unstable sort can reach the
// output through three channels, and for each the PARSED KEYS already name the contigs on which it can open:
//   H2a  two walkable hits with one (contig, cm): the order of the cm sort decides the arc and the W-line.  Two hits that
//        share cm share a CDS base, so a pair of ONE gene (on one strand under -S) never stays walkable -- pg_flt_ov_isoform
//        (overlap.c:58-93) filters one of them in stage A, unless something filtered it before.  Superset: a (contig, cm)
//        group with two hits of different genes (or, under -S, of different strands).
//   H3   first-wins ties: the dominator arg-max (overlap.c:150,153: equal 64-bit scores = one protein, one score_adj) and
//        pg_flt_subopt_isoform (hit.c:116: equal score_adj of two proteins of one gene); array order only decides between hits
//        of one (contig, cs) tie group.  Superset: two hits with one (contig, cs, score_adj) that are one protein, or -- under -S
//        only -- two proteins of one gene on opposite strands (same-strand isoforms sharing their start share a CDS base: one of
//        them is filtered by pg_flt_ov_isoform before hit.c:116 runs).
// Such contigs follow the reference's exact order from stage A on (ExactSeg::full), so a run needs ONE attempt where the
// round-3 design ran stages A-C, collected the hazard events and ran them again.  The dynamic detection on the backend stays:
// it covers H2b (the local_count boundary, not predictable from the keys) and anything this prediction would miss; an event
// on a contig that is tracked in full is harmless.  PANGENE_STATIC_TIES=0 switches the prediction off (tests: the repeated run).
// ---------------------------------------------------------------------------------------------
static bool static_predict() { static const bool on = [] { const char *e = std::getenv("PANGENE_STATIC_TIES"); return !(e && *e == '0'); }(); return on; }

// CDS intersection of two hits of one contig (what pg_hit_overlap, overlap.c:6-42, returns in its upper word): the exon lists are
// ascending and disjoint, so a merge of the two lists meets every intersecting pair once
static int64_t cds_intersection(const pg_genome_t *g, const pg_hit_t *a, const pg_hit_t *b)
{
	if (!(a->cs < b->ce && a->ce > b->cs)) return 0;
	const pg_exon_t *ea = g->exon + a->off_exon, *eb = g->exon + b->off_exon;
	int32_t ia = 0, ib = 0;
	int64_t inter = 0;
	while (ia < a->n_exon && ib < b->n_exon) {
		const int64_t s0 = a->cs + ea[ia].os, e0 = a->cs + ea[ia].oe, s1 = b->cs + eb[ib].os, e1 = b->cs + eb[ib].oe;
		const int64_t lo = s0 > s1 ? s0 : s1, hi = e0 < e1 ? e0 : e1;
		if (hi > lo) inter += hi - lo;
		if (e0 < e1) ++ia; else ++ib;
	}
	return inter;
}

static int64_t cds_length(const pg_genome_t *g, const pg_hit_t *a)
{
	int64_t len = 0;
	for (int32_t i = 0; i < a->n_exon; ++i) len += g->exon[a->off_exon + i].oe - g->exon[a->off_exon + i].os;
	return len;
}

static void static_tie_contigs(const pg_data_t *d, const pg_genome_t *g, bool check_strand, double min_ov_ratio, std::vector<int32_t> &out)
{
	out.clear();
	const int32_t n = g->n_hit;
	if (n < 2) return;
	size_t cap = 64;
	while (cap < (size_t)n * 2) cap <<= 1;
	// open addressing, one table for both kinds of key: slot = index of the group's first hit + 1; the members of a group are chained
	static thread_local std::vector<int32_t> tab, nxt;
	std::vector<uint8_t> marked((size_t)g->n_ctg, 0);
	static const bool dbg = std::getenv("PANGENE_TIMING") != nullptr && std::getenv("PANGENE_TIMING")[0] == '2'; // (PANGENE_TIMING=2: also which tie marked which contig -- a line per contig)
	auto mix = [](uint64_t a, uint64_t b) { uint64_t h = (a + 0x9e3779b97f4a7c15ull) * 0xbf58476d1ce4e5b9ull; h ^= h >> 29; h = (h + b) * 0x94d049bb133111ebull; h ^= h >> 32; return h; };
	for (int kind = 0; kind < 2; ++kind) {
		tab.assign(cap, 0);
		if (kind == 0) nxt.assign((size_t)n, -1);
		for (int32_t i = 0; i < n; ++i) {
			const pg_hit_t *a = &g->hit[i];
			if (a->cid < 0 || a->cid >= g->n_ctg || marked[(size_t)a->cid]) continue;
			const int32_t ga = d->prot[a->pid].gid;
			const uint64_t k0 = (uint64_t)(uint32_t)a->cid << 32 | (kind == 0 ? 0u : (uint32_t)ga), k1 = kind == 0 ? (uint64_t)a->cm : ((uint64_t)a->cs << 1 ^ (uint64_t)(uint32_t)a->score_adj << 40);
			for (size_t p = (size_t)mix(k0, k1) & (cap - 1);; p = (p + 1) & (cap - 1)) {
				if (tab[p] == 0) { tab[p] = i + 1; break; }
				const int32_t first = tab[p] - 1;
				const pg_hit_t *b = &g->hit[first];
				if (b->cid != a->cid) continue;
				if (kind == 0) {
					if (b->cm != a->cm) continue;
					// H2a: could a and a member of its (contig, cm) group both be walkable (!flt && !shadow)?  They share a CDS base.  One gene
					// (one strand under -S): pg_flt_ov_isoform filters one.  Different genes: pg_shadow marks one of every pair it compares
					// (overlap.c:126-154) -- it skips the pair when the strands differ under -S, or when the shorter CDS is covered by less
					// than min_ov_ratio (overlap.c:134-136): only then can both stay walkable
					int32_t last = first;
					for (int32_t m = first; m >= 0 && !marked[(size_t)a->cid]; last = m, m = nxt[(size_t)m]) {
						const pg_hit_t *c = &g->hit[m];
						bool hazard = check_strand && c->rev != a->rev;
						if (!hazard && d->prot[c->pid].gid != ga) {
							const int64_t la = cds_length(g, a), lc = cds_length(g, c), mn = la < lc ? la : lc;
							hazard = (double)cds_intersection(g, a, c) / (double)(mn > 0 ? mn : 1) < min_ov_ratio;
						}
						if (hazard) {
							marked[(size_t)a->cid] = 1;
							if (dbg) std::fprintf(stderr, "[static] cm tie: contig %d cm %ld: proteins %d and %d\n", a->cid, (long)a->cm, a->pid, c->pid);
						}
					}
					nxt[(size_t)last] = i; // (i joins the chain; nxt[i] is -1)
					break;
				}
				// H3: first-wins ties between two hits of one (contig, cs) group -- equal 64-bit scores (one protein, one score_adj: the
				// dominator arg-max, overlap.c:150,153) or equal score_adj of two proteins of one gene (hit.c:116)
				if (b->cs != a->cs || b->score_adj != a->score_adj || d->prot[b->pid].gid != ga) continue;
				// one protein twice (duplicate alignments): both can be the dominator of a third hit in stage A's pg_shadow, which runs
				// before anything filters one of them.  Two proteins of the gene: they share their first CDS base, so pg_flt_ov_isoform
				// (read.c:254) has filtered one before pg_flt_subopt_isoform (read.c:256) looks -- unless -S keeps pairs of opposite
				// strands apart (overlap.c:77).  (Isoform-rich sets are full of such pairs: marking them all made every pass of the
				// 200 x 110 k-isoform set replay thousands of contigs, 2.8 s of host time.)
				if (b->pid != a->pid && !(check_strand && b->rev != a->rev)) continue;
				marked[(size_t)a->cid] = 1;
				if (dbg) std::fprintf(stderr, "[static] cs tie: contig %d cs %ld gene %d: proteins %d and %d, score_adj %d\n", a->cid, (long)a->cs, ga, a->pid, b->pid, a->score_adj);
				break;
			}
		}
	}
	for (int32_t c = 0; c < g->n_ctg; ++c) if (marked[(size_t)c]) out.push_back(c);
}

// build the tracked segments from the host arrays, which must still be in FILE order
void exact_init(const pg_data_t *d, DataExt *ext)
{
	exact_shutdown(ext); // nobody may still be replaying the segments that go away
	ext->xsegs.clear();
	ext->xreplayed = false, ext->xsegs_n_genome = d->n_genome;
	++ext->xsegs_gen;
	const int mode = exact_mode();
	if (mode == 0) return;
	// genomes are independent: host threads collect their segments, which are then appended in genome order
	const size_t ng = ext->local_genomes.size();
	std::vector<std::vector<ExactSeg>> per((size_t)ng);
	std::vector<std::vector<int32_t>> per_static((size_t)ng);
	ext->static_ctgs.clear();
	auto do_genome = [&](size_t k) {
		std::vector<ExactSeg> &out = per[k];
		const pg_genome_t *g = &d->genome[ext->local_genomes[k]];
		if (g->n_hit < 2) return;
		std::vector<int32_t> cnt((size_t)g->n_ctg + 1, 0);
		for (int32_t i = 0; i < g->n_hit; ++i) ++cnt[(size_t)g->hit[i].cid + 1];
		for (int32_t c = 0; c < g->n_ctg; ++c) cnt[(size_t)c + 1] += cnt[(size_t)c];
		int32_t c0 = 0;
		while (c0 < g->n_ctg && cnt[(size_t)c0 + 1] == cnt[(size_t)c0]) ++c0;
		// host index of every hit in FILE order (the host array is in cs order once a sync has happened)
		const int32_t j = ext->local_genomes[k];
		std::vector<int32_t> host_of_file((size_t)g->n_hit);
		if ((size_t)j < ext->hits_sorted.size() && ext->hits_sorted[(size_t)j])
			for (int32_t h = 0; h < g->n_hit; ++h) host_of_file[(size_t)ext->file_of_host[(size_t)j][(size_t)h]] = h;
		else
			for (int32_t h = 0; h < g->n_hit; ++h) host_of_file[(size_t)h] = h;
		// contigs that get the full treatment although the mode is auto: tie hazards seen there in a previous attempt, and the
		// contigs on which the STATIC keys already say that a tie channel other than array index 0 can open (static_tie_contigs)
		std::vector<int32_t> extra;
		if (mode == 1) {
			auto lo = std::lower_bound(ext->extra_ctgs.begin(), ext->extra_ctgs.end(), std::make_pair((int32_t)k, (int32_t)INT32_MIN));
			for (; lo != ext->extra_ctgs.end() && lo->first == (int32_t)k; ++lo) extra.push_back(lo->second);
			if (static_predict()) {
				static_tie_contigs(d, g, ext->check_strand, ext->min_ov_ratio, per_static[k]);
				for (int32_t c : per_static[k]) extra.push_back(c);
				std::sort(extra.begin(), extra.end());
				extra.erase(std::unique(extra.begin(), extra.end()), extra.end());
			}
		}
		// file indices grouped by contig, file order inside a contig (one pass; a contig's hits are then contiguous)
		const bool every = mode == 2 || !extra.empty();
		std::vector<int32_t> by_ctg((size_t)g->n_hit), cur(cnt.begin(), cnt.end() - 1);
		for (int32_t i = 0; i < g->n_hit; ++i) {
			const int32_t c = g->hit[host_of_file[(size_t)i]].cid;
			if (every || c == c0) by_ctg[(size_t)cur[(size_t)c]++] = i;
		}
		for (int32_t c = c0; c < g->n_ctg; ++c) {
			const bool full = mode == 2 || std::binary_search(extra.begin(), extra.end(), c);
			if (!full && c != c0) { if (extra.empty()) break; else continue; } // auto only ever looks at the first non-empty contig
			const int32_t n = cnt[(size_t)c + 1] - cnt[(size_t)c];
			if (n < 2) continue;
			ExactSeg s;
			s.k = (int32_t)k, s.start = cnt[(size_t)c], s.full = full;
			s.file.reserve((size_t)n), s.cs.reserve((size_t)n), s.cm.reserve((size_t)n);
			int64_t min_cs = INT64_MAX; int32_t n_min = 0;
			for (int32_t t = cnt[(size_t)c]; t < cnt[(size_t)c + 1]; ++t) {
				const int32_t i = by_ctg[(size_t)t];
				const pg_hit_t *a = &g->hit[host_of_file[(size_t)i]];
				s.file.push_back(i), s.cs.push_back((uint64_t)a->cs), s.cm.push_back((uint64_t)a->cm);
				if (a->cs < min_cs) min_cs = a->cs, n_min = 1; else if (a->cs == min_cs) ++n_min;
			}
			if (!full && n_min < 2) continue;  // auto: only the index-0 channel, i.e. a leading tie group on the first contig
			out.push_back(std::move(s));
		}
	};
	int64_t tot = 0;
	for (size_t k = 0; k < ng; ++k) tot += d->genome[ext->local_genomes[k]].n_hit;
	unsigned nt = tot > 200000 ? host_threads(16u) : 1u;
	if (nt > ng) nt = (unsigned)ng;
	if (nt <= 1) { for (size_t k = 0; k < ng; ++k) do_genome(k); }
	else {
		std::atomic<size_t> next{0};
		std::vector<std::thread> th;
		for (unsigned t = 0; t < nt; ++t) th.emplace_back([&]() { for (;;) { const size_t k = next.fetch_add(1); if (k >= ng) break; do_genome(k); } });
		for (auto &x : th) x.join();
	}
	for (size_t k = 0; k < ng; ++k) {
		for (ExactSeg &s : per[k]) ext->xsegs.push_back(std::move(s));
		for (int32_t c : per_static[k]) ext->static_ctgs.emplace_back((int32_t)k, c); // (sorted: genomes ascending, contigs ascending inside)
	}
	if (std::getenv("PANGENE_TIMING")) {
		size_t n_full = 0, n_full_hits = 0;
		for (const ExactSeg &s : ext->xsegs) if (s.full) ++n_full, n_full_hits += s.file.size();
		std::fprintf(stderr, "[exact_init] mode %d: %zu tracked contig(s), %zu of them in full (%zu hits); %zu by the static tie prediction\n", mode, ext->xsegs.size(), n_full, n_full_hits, ext->static_ctgs.size());
	}
}

static void emulate(ExactSeg &s, std::vector<int32_t> &curv, int by_cm) // one pg_hit_sort of this contig segment
{
	const std::vector<uint64_t> &key = by_cm ? s.cm : s.cs;
	const size_t n = curv.size();
	const uint64_t *kp = key.data();
	int32_t *cur = curv.data();
	// (round 6) keys below 2^32 -- every contig without 64-bit coordinates -- travel with their index in ONE 64-bit word: the sort's element moves depend on the keys
	// alone (ksort_exact.hpp), and elements of 8 bytes instead of 16 halve what its passes and insertion sorts move
	bool narrow = true;
	for (size_t i = 0; i < n && narrow; ++i) narrow = kp[i] >> 32 == 0;
	if (narrow) {
		static thread_local std::vector<uint64_t> tls8;
		if (tls8.size() < n) tls8.resize(n);
		uint64_t *t8 = tls8.data();
		for (size_t i = 0; i < n; ++i) t8[i] = kp[(size_t)cur[i]] << 32 | (uint32_t)cur[i];
		ksort_exact(t8, n, [](const uint64_t &a) { return a >> 32; });
		for (size_t i = 0; i < n; ++i) cur[i] = (int32_t)(uint32_t)t8[i];
		return;
	}
	static thread_local std::vector<pg128_t> tls; // reused: a fresh vector per call would be mmap'ed and page-faulted each time
	if (tls.size() < curv.size()) tls.resize(curv.size());
	pg128_t *t = tls.data();
	for (size_t i = 0; i < n; ++i) t[i].x = kp[(size_t)cur[i]], t[i].y = (uint64_t)cur[i];
	ksort_exact(t, n, [](const pg128_t &a) { return a.x; });
	for (size_t i = 0; i < n; ++i) cur[i] = (int32_t)t[i].y;
}

// Replay the whole sort sequence of one segment: file order -cs-> X1 -cm-> Y1 -cs-> X2 ...  It is a deterministic
// map on a finite set, so it becomes periodic; the replay stops when a cs order repeats (normally after 2-4
// sorts; the reference performs 67) or after MAX_SORTS.  keep_orders: store every order (mode "all"), else only
// the hit at array index 0 of each cs order (mode "auto").
static const int MAX_SORTS = 40;
static void replay(ExactSeg &s, bool keep_orders)
{
	std::vector<int32_t> cur(s.file.size());
	for (size_t i = 0; i < cur.size(); ++i) cur[i] = (int32_t)i;
	std::vector<std::vector<int32_t>> hist;
	s.hx.clear(), s.hy.clear(), s.heads.clear();
	s.cyc_start = -1, s.period = 0;
	for (int t = 1; t <= MAX_SORTS; ++t) {
		emulate(s, cur, 0);
		for (size_t i = 0; i < hist.size(); ++i)
			if (hist[i] == cur) { s.cyc_start = (int)i + 1, s.period = t - s.cyc_start; break; }
		if (s.cyc_start > 0) break;
		hist.push_back(cur);
		s.heads.push_back(s.file[(size_t)cur[0]]);
		emulate(s, cur, 1);
		if (keep_orders) s.hy.push_back(cur);
	}
	if (keep_orders) s.hx.swap(hist);
	// identities of the stored orders (a contig's X orders are distinct up to the cycle by construction, its Y orders need not be)
	auto ident = [](const std::vector<std::vector<int32_t>> &h, std::vector<int32_t> &id) {
		id.assign(h.size(), 0);
		for (size_t i = 0; i < h.size(); ++i) {
			id[i] = (int32_t)i;
			for (size_t j = 0; j < i; ++j) if (h[j] == h[i]) { id[i] = (int32_t)j; break; }
		}
	};
	ident(s.hx, s.xid), ident(s.hy, s.yid);
	s.pushed_id[0] = s.pushed_id[1] = -1;
}

static inline size_t order_index(const ExactSeg &s, int t, size_t n_stored) // which stored order the t-th sort (1-based) produced
{
	size_t i = (s.cyc_start > 0 && t >= s.cyc_start) ? (size_t)(s.cyc_start - 1 + (t - s.cyc_start) % s.period) : (size_t)(t - 1);
	return i < n_stored ? i : n_stored - 1;
}

static void exact_wait(DataExt *ext)
{
	for (std::thread &t : ext->xworkers) t.join();
	ext->xworkers.clear();
}

// start of a run: the arrays are in file order (read.c:232-234).  The replay depends on the keys only, so it runs
// on background threads while the GPU does stage A and B; the first consumer joins them.
static void spawn_replay(DataExt *ext)
{
	if (ext->xreplayed) return;
	ext->xreplayed = true; // (valid once the workers have been joined: exact_wait)
	if (ext->xsegs.empty()) return;
	const bool all = exact_mode() == 2;
	size_t tot = 0;
	for (const ExactSeg &s : ext->xsegs) tot += s.file.size();
	unsigned nt = tot > 50000 ? host_threads(16u) : 1u;
	if (nt > ext->xsegs.size()) nt = (unsigned)ext->xsegs.size();
	ext->xnext.store(0);
	for (unsigned t = 0; t < nt; ++t)
		ext->xworkers.emplace_back([ext, all]() {
			for (;;) {
				size_t i = ext->xnext.fetch_add(1);
				if (i >= ext->xsegs.size()) break;
				replay(ext->xsegs[i], all || ext->xsegs[i].full);
			}
		});
}

// start of a run: the arrays are in file order (read.c:232-234).  The replay depends on the keys only: it runs on background
// threads, started as early as the reader (exact_prefetch) or here, while the GPU does stages A and B; the first consumer joins
// them.  A repeated run on the same shard (pg_rerun_resident) finds the replay done.
// (Round 6: a replay the reader started is NOT waited for here -- on a 12.1 M-hit shard it was 9-33 ms of an upload-inclusive pass of 60-80, spent
// in front of stage A, which needs nothing of it (mode "auto": the hand-over is in pg_graph_gen; mode "all": exact_sort waits as every consumer
// does).  The workers leave every segment as this function would -- pushed_id = -1, replay()'s last line -- and touch nothing else of the data set.)
void exact_begin(DataExt *ext)
{
	const bool in_flight = !ext->xworkers.empty();
	static const bool wait_here = std::getenv("PANGENE_EXACT_WAIT") != nullptr; // (tests / timing: the old order of things)
	const double t0 = now_sec();
	if (wait_here) exact_wait(ext);
	if (std::getenv("PANGENE_TIMING")) std::fprintf(stderr, "[exact_begin] waited %.3f ms for the order replay the reader started%s\n", (now_sec() - t0) * 1e3, in_flight && !wait_here ? " (it goes on beside stages A and B: its first consumer waits)" : "");
	ext->head_file.assign(ext->local_genomes.size(), -1);
	ext->x_sorts[0] = ext->x_sorts[1] = 0;
	if (!in_flight || wait_here) for (ExactSeg &s : ext->xsegs) s.pushed_id[0] = s.pushed_id[1] = -1;
	spawn_replay(ext);
}

// called by the reader when a batch of files has been committed: the tracked segments and their replay need nothing but the
// parsed keys, so they are out of the way before pg_post_process even starts
void exact_prefetch(const pg_data_t *d, DataExt *ext)
{
	ext->local_genomes.clear();
	ext->is_local.resize((size_t)d->n_genome, 1);
	for (int32_t j = 0; j < d->n_genome; ++j)
		if (ext->is_local[(size_t)j]) ext->local_genomes.push_back(j);
	for (int32_t j : ext->local_genomes) if ((size_t)j < ext->hits_sorted.size() && ext->hits_sorted[(size_t)j]) return; // records already moved: pg_post_process sorts it out
	ext->extra_ctgs.clear();
	exact_init(d, ext);
	ext->exact_mode_of_segs = exact_mode();
	spawn_replay(ext);
}

// One pg_hit_sort(g, by_cm) of the reference happened: hand the orders that changed to the backend.
int exact_sort(DataExt *ext, int by_cm)
{
	if (ext->xsegs.empty()) return 0;
	exact_wait(ext);
	const int t = ++ext->x_sorts[by_cm];
	// contigs tracked for the index-0 channel only (mode auto): the identity of the hit at array index 0, cs order only
	if (!by_cm) {
		bool changed = false;
		for (ExactSeg &s : ext->xsegs) {
			if (s.full) continue;
			const int32_t h = s.heads[order_index(s, t, s.heads.size())];
			if (ext->head_file[(size_t)s.k] != h) ext->head_file[(size_t)s.k] = h, changed = true;
		}
		if (changed) { const int rc = ext->be->set_head(ext->ctx, ext->head_file.data()); if (rc) return rc; }
	}
	// fully tracked contigs: the whole order
	const double t_list0 = now_sec();
	static thread_local std::vector<int32_t> sg, ss, fi; // (reused: a million entries per call on isoform-rich shards)
	static thread_local std::vector<int64_t> so;
	sg.clear(), ss.clear(), fi.clear(), so.assign(1, 0);
	for (ExactSeg &s : ext->xsegs) {
		if (!s.full) continue;
		const std::vector<std::vector<int32_t>> &h = by_cm ? s.hy : s.hx;
		if (h.empty()) continue;
		const size_t oi = order_index(s, t, h.size());
		const int32_t id = (by_cm ? s.yid : s.xid)[oi];
		if (id == s.pushed_id[by_cm]) continue;
		s.pushed_id[by_cm] = id;
		const std::vector<int32_t> &ord = h[oi];
		sg.push_back(s.k), ss.push_back(s.start);
		const size_t at = fi.size();
		fi.resize(at + ord.size());
		const int32_t *fl = s.file.data();
		for (size_t k = 0; k < ord.size(); ++k) fi[at + k] = fl[(size_t)ord[k]];
		so.push_back((int64_t)fi.size());
	}
	if (sg.empty()) return 0;
	ext->order_touched = true; // the backend's orders change: sync_host compares what the tracked contigs hold now with what its copy was taken from (order_signature)
	const double t_be0 = now_sec();
	const int rc = ext->be->override_order(ext->ctx, by_cm, (int32_t)sg.size(), sg.data(), ss.data(), so.data(), fi.data());
	ext->ov_calls += 1, ext->ov_hits += (int64_t)fi.size(), ext->ov_list_s += t_be0 - t_list0, ext->ov_backend_s += now_sec() - t_be0;
	return rc;
}

void exact_shutdown(DataExt *ext) { exact_wait(ext); }

// What the orders of the fully tracked contigs are right now, as one number: a repeated run over a resident shard goes through
// the same overrides and ends with the same orders, and the host's copy of the two orders (pos_x, y_file: 8 bytes a hit to
// fetch and to turn around) is still good then.
uint64_t order_signature(const DataExt *ext)
{
	uint64_t h = 1469598103934665603ull;
	auto mix = [&h](uint64_t x) { h = (h ^ x) * 1099511628211ull; };
	mix(ext->xsegs_gen);
	for (const ExactSeg &s : ext->xsegs) {
		if (!s.full) continue;
		mix((uint64_t)(uint32_t)s.k << 32 | (uint32_t)s.start);
		mix((uint64_t)(uint32_t)s.pushed_id[0] << 32 | (uint32_t)s.pushed_id[1]); // (identities inside this set of segments: xsegs_gen tells the sets apart)
	}
	return h;
}

// Would the next n sorts of each kind (hit.c:29-64) need nothing from the host?  True when the hit at array index 0 of every
// genome stays the one the backend holds through the next n cs sorts, and every contig tracked in full keeps the two orders the
// backend holds (the sort sequence of a contig is periodic -- as a rule it has reached its fixed point after two sorts, and a
// sequence of stable sorts (<= 64 hits, ksort.h:79-80) always has: X_2 = X_3 = ..., Y_1 = Y_2 = ...).
bool exact_quiet(DataExt *ext, int n)
{
	if (ext->xsegs.empty()) return true;
	exact_wait(ext);
	for (const ExactSeg &s : ext->xsegs) {
		if (s.full) {
			for (int by_cm = 0; by_cm < 2; ++by_cm) {
				const std::vector<std::vector<int32_t>> &h = by_cm ? s.hy : s.hx;
				if (h.empty()) continue;
				size_t last = (size_t)-1;
				for (int t = ext->x_sorts[by_cm] + 1; t <= ext->x_sorts[by_cm] + n; ++t) {
					const size_t i = order_index(s, t, h.size());
					if (i == last) continue; // (compared already)
					if ((by_cm ? s.yid : s.xid)[i] != s.pushed_id[by_cm]) return false;
					last = i;
				}
			}
			continue;
		}
		for (int t = ext->x_sorts[0] + 1; t <= ext->x_sorts[0] + n; ++t)
			if (s.heads[order_index(s, t, s.heads.size())] != ext->head_file[(size_t)s.k]) return false;
	}
	return true;
}

// the backend made n sorts of each kind on its own (after exact_quiet said it could)
void exact_skip(DataExt *ext, int n) { if (!ext->xsegs.empty()) ext->x_sorts[0] += n, ext->x_sorts[1] += n; }

} // namespace pgx

extern "C" void pg_set_exact_mode(int mode) { pgx::set_exact_mode(mode); }
