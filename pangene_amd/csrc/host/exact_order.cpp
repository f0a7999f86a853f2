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
#include <thread>
#include "pg_internal.hpp"
#include "ksort_exact.hpp"

namespace pgx {

static int g_exact_mode = -1; // -1: read the environment

int exact_mode()
{
	if (g_exact_mode < 0) {
		const char *e = std::getenv("PANGENE_EXACT");
		g_exact_mode = (e == nullptr || std::strcmp(e, "auto") == 0) ? 1 : std::strcmp(e, "all") == 0 ? 2 : std::strcmp(e, "off") == 0 ? 0 : 1;
	}
	return g_exact_mode;
}
void set_exact_mode(int m) { g_exact_mode = m; }

// build the tracked segments from the host arrays, which must still be in FILE order
void exact_init(const pg_data_t *d, DataExt *ext)
{
	ext->xsegs.clear();
	const int mode = exact_mode();
	if (mode == 0) return;
	for (size_t k = 0; k < ext->local_genomes.size(); ++k) {
		const pg_genome_t *g = &d->genome[ext->local_genomes[k]];
		if (g->n_hit < 2) continue;
		std::vector<int32_t> cnt((size_t)g->n_ctg + 1, 0);
		for (int32_t i = 0; i < g->n_hit; ++i) ++cnt[(size_t)g->hit[i].cid + 1];
		for (int32_t c = 0; c < g->n_ctg; ++c) cnt[(size_t)c + 1] += cnt[(size_t)c];
		int32_t c0 = 0;
		while (c0 < g->n_ctg && cnt[(size_t)c0 + 1] == cnt[(size_t)c0]) ++c0;
		for (int32_t c = c0; c < g->n_ctg; ++c) {
			const int32_t n = cnt[(size_t)c + 1] - cnt[(size_t)c];
			if (n < 2) { if (mode == 1) break; else continue; }
			ExactSeg s;
			s.k = (int32_t)k, s.start = cnt[(size_t)c];
			s.file.reserve((size_t)n);
			int64_t min_cs = INT64_MAX; int32_t n_min = 0;
			for (int32_t i = 0; i < g->n_hit; ++i) {
				const pg_hit_t *a = &g->hit[i];
				if (a->cid != c) continue;
				s.file.push_back(i), s.cs.push_back((uint64_t)a->cs), s.cm.push_back((uint64_t)a->cm);
				if (a->cs < min_cs) min_cs = a->cs, n_min = 1; else if (a->cs == min_cs) ++n_min;
			}
			if (mode == 1 && n_min < 2) break;  // auto: only the index-0 channel, i.e. a leading tie group on the first contig
			ext->xsegs.push_back(std::move(s));
			if (mode == 1) break;
		}
	}
}

void exact_begin(DataExt *ext) // start of a run: arrays are in file order (read.c:232-234)
{
	for (ExactSeg &s : ext->xsegs) {
		s.cur.resize(s.file.size());
		for (size_t i = 0; i < s.file.size(); ++i) s.cur[i] = (int32_t)i; // indices into s.file / s.cs / s.cm
		s.hx.clear(), s.hy.clear(), s.pushed[0].clear(), s.pushed[1].clear();
		s.cyc_start = -1, s.period = 0, s.n_sort[0] = s.n_sort[1] = 0, s.view = &s.cur;
	}
	ext->head_file.assign(ext->local_genomes.size(), -1);
}

static void emulate(ExactSeg &s, int by_cm) // one pg_hit_sort of this contig segment
{
	const std::vector<uint64_t> &key = by_cm ? s.cm : s.cs;
	static thread_local std::vector<pg128_t> tls; // reused: a fresh 160 KB vector per call would be mmap'ed and page-faulted each time
	if (tls.size() < s.cur.size()) tls.resize(s.cur.size());
	pg128_t *t = tls.data();
	const size_t n = s.cur.size();
	const uint64_t *kp = key.data();
	int32_t *cur = s.cur.data();
	for (size_t i = 0; i < n; ++i) t[i].x = kp[(size_t)cur[i]], t[i].y = (uint64_t)cur[i];
	ksort_exact(t, n, [](const pg128_t &a) { return a.x; });
	for (size_t i = 0; i < n; ++i) cur[i] = (int32_t)t[i].y;
}

// advance one segment by one sort.  The sequence X1 -cm-> Y1 -cs-> X2 -cm-> Y2 ... is a deterministic map on a
// finite set, so it becomes periodic; once a cs order repeats, later orders are read from the history.
static void advance(ExactSeg &s, int by_cm, bool keep_y)
{
	const int t = ++s.n_sort[by_cm]; // 1-based index of this sort among the sorts of its kind
	if (s.cyc_start > 0) { // periodic: point at the stored order instead of copying it
		if (by_cm && !keep_y) return; // order not needed by the caller
		s.view = &(by_cm ? s.hy : s.hx)[(size_t)(s.cyc_start - 1 + (t - s.cyc_start) % s.period)];
		return;
	}
	s.view = &s.cur;
	emulate(s, by_cm);
	if (by_cm) { if (keep_y) s.hy.push_back(s.cur); return; }
	for (size_t i = 0; i < s.hx.size(); ++i)
		if (s.hx[i] == s.cur) { s.cyc_start = (int)i + 1, s.period = t - s.cyc_start; return; }
	s.hx.push_back(s.cur);
}

// Replay one pg_hit_sort(g, by_cm) of every tracked segment and hand what changed to the backend.
int exact_sort(DataExt *ext, int by_cm)
{
	if (ext->xsegs.empty()) return 0;
	const bool all = exact_mode() == 2;
	std::vector<ExactSeg *> todo;
	size_t tot = 0;
	for (ExactSeg &s : ext->xsegs) { todo.push_back(&s); if (s.cyc_start < 0) tot += s.cur.size(); }
	auto work = [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) advance(*todo[i], by_cm, all); };
	unsigned nt = tot > 200000 ? std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u) : 1u;
	if (nt > todo.size()) nt = (unsigned)todo.size();
	if (nt <= 1) work(0, todo.size());
	else {
		std::vector<std::thread> th;
		for (unsigned t = 0; t < nt; ++t) th.emplace_back(work, todo.size() * t / nt, todo.size() * (t + 1) / nt);
		for (auto &x : th) x.join();
	}
	if (!all) { // auto: only the identity of the hit at array index 0 matters, and only for the cs order
		if (by_cm) return 0;
		bool changed = false;
		for (ExactSeg *s : todo) {
			const int32_t h = s->file[(size_t)(*s->view)[0]];
			if (ext->head_file[(size_t)s->k] != h) ext->head_file[(size_t)s->k] = h, changed = true;
		}
		return changed ? ext->be->set_head(ext->ctx, ext->head_file.data()) : 0;
	}
	std::vector<int32_t> sg, ss, fi;
	std::vector<int64_t> so(1, 0);
	for (ExactSeg *s : todo) {
		if (*s->view == s->pushed[by_cm]) continue;
		s->pushed[by_cm] = *s->view;
		sg.push_back(s->k), ss.push_back(s->start);
		for (int32_t i : *s->view) fi.push_back(s->file[(size_t)i]);
		so.push_back((int64_t)fi.size());
	}
	if (sg.empty()) return 0;
	ext->pos_valid = false; // the backend's orders change: the host copy of them is stale
	return ext->be->override_order(ext->ctx, by_cm, (int32_t)sg.size(), sg.data(), ss.data(), so.data(), fi.data());
}

} // namespace pgx

extern "C" void pg_set_exact_mode(int mode) { pgx::set_exact_mode(mode); }
