// paf_reader.cpp -- host-side PAF ingest for the drop-in surface: pg_data_init/destroy, pg_read_paf,
// pg_scan_paf_ids, pg_read_list_dict.  Text parsing is outside the accelerated path (SURVEY.md section 2:
// "must be rebuilt on host"); what matters here is that ids, ranks, exon lists, cm and score_adj come
// out exactly as the reference's reader produces them (read.c:107-236, hit.c:14-27), because
// pg_hash_uint32(pid) and the first-seen numbering are score-relevant downstream.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include "pg_internal.hpp"

namespace pgx {

static std::unordered_map<const pg_data_t *, DataExt *> g_ext;
static std::mutex g_ext_mu;

DataExt *ext_of(const pg_data_t *d, bool create)
{
	std::lock_guard<std::mutex> lk(g_ext_mu);
	auto it = g_ext.find(d);
	if (it != g_ext.end()) return it->second;
	if (!create) return nullptr;
	DataExt *e = new DataExt();
	g_ext.emplace(d, e);
	return e;
}

void ext_drop(const pg_data_t *d)
{
	std::lock_guard<std::mutex> lk(g_ext_mu);
	auto it = g_ext.find(d);
	if (it == g_ext.end()) return;
	DataExt *e = it->second;
	exact_shutdown(e);
	if (e->ctx && e->be) e->be->destroy(e->ctx);
	free_packs(e, true);
	delete e;
	g_ext.erase(it);
	if (g_ext.empty()) trim_host_caches((size_t)256 << 20); // no data set left: most of the page-locked memory goes back
}

// gz-or-plain line source (zlib reads plain files transparently, as the reference's gzopen does)
class LineSource {
public:
	explicit LineSource(const char *fn) {
		fp_ = (fn && std::strcmp(fn, "-") != 0) ? gzopen(fn, "r") : gzdopen(0, "r");
		if (fp_) gzbuffer(fp_, 1 << 16); // (below the allocator's mmap threshold: a buffer that is mapped and unmapped per file costs every thread of the process a TLB shoot-down)
		buf_.resize(1 << 16);
	}
	~LineSource() { if (fp_) gzclose(fp_); }
	bool ok() const { return fp_ != nullptr; }
	// next line without the '\n'; a trailing '\r' is dropped when the line is longer than one char
	bool next(std::string &line) {
		line.clear();
		if (eof_ && beg_ >= end_) return false;
		bool got = false;
		for (;;) {
			if (beg_ >= end_) {
				if (eof_) break;
				int n = gzread(fp_, buf_.data(), (unsigned)buf_.size());
				beg_ = 0, end_ = n > 0 ? n : 0;
				if (end_ < (int)buf_.size()) eof_ = true;
				if (end_ == 0) break;
			}
			const char *p = (const char *)std::memchr(buf_.data() + beg_, '\n', end_ - beg_);
			int stop = p ? (int)(p - buf_.data()) : end_;
			line.append(buf_.data() + beg_, stop - beg_);
			got = true;
			beg_ = stop + 1;
			if (p) break;
		}
		if (!got) return false;
		if (line.size() > 1 && line.back() == '\r') line.pop_back();
		return true;
	}
private:
	gzFile fp_ = nullptr;
	std::vector<char> buf_;
	int beg_ = 0, end_ = 0;
	bool eof_ = false;
};

// A plain (not gzipped) PAF file wholly in memory: one read() into a buffer the thread keeps from file to file (no allocation, no
// mapping per file: in a process with a hundred parser threads every mmap / munmap is a TLB shoot-down for all of them), its lines
// parsed IN PLACE (the parser writes its terminators into the buffer).  Same line semantics as LineSource: a line ends at '\n', a last
// line without one counts when it is not empty, a trailing '\r' is dropped from lines longer than one character.
class WholeFile {
public:
	explicit WholeFile(const char *fn) {
		if (fn == nullptr || std::strcmp(fn, "-") == 0) return;
		const int fd = open(fn, O_RDONLY);
		if (fd < 0) return;
		unsigned char magic[2] = { 0, 0 };
		struct stat sb;
		if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || pread(fd, magic, 2, 0) < 0 || (magic[0] == 0x1f && magic[1] == 0x8b)) { close(fd); return; } // gzipped, a pipe, ...: LineSource
		static thread_local std::vector<char> tl_buf;
		const size_t n = (size_t)sb.st_size;
		if (tl_buf.size() < n + 2) tl_buf.resize(n + n / 4 + 4096);
		size_t got = 0;
		while (got < n) { const ssize_t k = read(fd, tl_buf.data() + got, n - got); if (k <= 0) break; got += (size_t)k; }
		close(fd);
		p_ = tl_buf.data(), end_ = p_ + got, ok_ = true;
	}
	bool ok() const { return ok_; }
	bool next(char *&line, size_t &len) { // the line is followed by a byte the caller may overwrite
		if (p_ >= end_) return false;
		char *nl = (char *)std::memchr(p_, '\n', (size_t)(end_ - p_));
		char *e = nl ? nl : end_;
		line = p_, len = (size_t)(e - p_);
		p_ = e + 1;
		if (len > 1 && line[len - 1] == '\r') --len;
		return true;
	}
private:
	char *p_ = nullptr, *end_ = nullptr;
	bool ok_ = false;
};

// strtol(q, 0, 10) for the digit strings of a PAF line: optional blanks and sign, then digits (anything else ends the number)
static inline int64_t parse_i64(const char *q)
{
	while (*q == ' ') ++q;
	bool neg = false;
	if (*q == '-') neg = true, ++q; else if (*q == '+') ++q;
	uint64_t v = 0;
	bool sat = false;
	while ((unsigned)(*q - '0') < 10u) {
		if (v > (uint64_t)INT64_MAX / 10) sat = true; // strtol saturates at LONG_MAX / LONG_MIN
		v = v * 10 + (uint64_t)(*q - '0'), ++q;
	}
	if (sat || v > (uint64_t)INT64_MAX) return neg ? INT64_MIN : INT64_MAX;
	return neg ? -(int64_t)v : (int64_t)v;
}

// grow helpers keeping the reference's malloc/realloc ownership (pg_data_destroy frees with free())
template <class T> static void grow0(T *&ptr, int32_t idx, int32_t &cap)
{
	if (idx < cap) return;
	int32_t old = cap;
	cap = idx + 1;
	cap += (cap >> 1) + 16;
	ptr = (T *)std::realloc(ptr, sizeof(T) * (size_t)cap);
	std::memset((void *)(ptr + old), 0, sizeof(T) * (size_t)(cap - old));
}

// coordinate of the middle CDS base (hit.c:14-27)
static int64_t middle_cds(int64_t cs, const pg_exon_t *e, int32_t n)
{
	int32_t tot = 0;
	for (int32_t i = 0; i < n; ++i) tot += e[i].oe - e[i].os;
	int32_t half = tot >> 1, acc = 0;
	for (int32_t i = 0; i < n; ++i) {
		int32_t l = e[i].oe - e[i].os;
		if (acc <= half && half < acc + l) return cs + e[i].os + half - acc;
		acc += l;
	}
	return -1; // zero-length CDS; the reference aborts here (hit.c:25)
}

// miniprot CIGAR -> exon list in ascending contig coordinates (read.c:47-90).  Returns false when the
// CIGAR does not span ce-cs (the reference asserts, read.c:75).
static bool cigar_to_exons(const char *cg, bool rev, int64_t span, std::vector<pg_exon_t> &ex, int32_t *n_fs)
{
	ex.clear();
	ex.push_back(pg_exon_t{0, 0});
	int64_t x = 0;
	int32_t fs = 0;
	const char *p = cg;
	while (*p) {
		char *r;
		int64_t l = std::strtol(p, &r, 10);
		char op = *r;
		if (op == 'N' || op == 'U' || op == 'V') {
			int64_t st, en;
			if (op == 'N') st = x, en = x + l;
			else if (op == 'U') st = x + 1, en = x + l - 2;
			else st = x + 2, en = x + l - 1;
			ex.back().oe = (int32_t)st;
			ex.push_back(pg_exon_t{(int32_t)en, (int32_t)en});
			x += l;
		} else if (op == 'M' || op == 'X' || op == '=' || op == 'D') {
			x += l * 3;
		} else if (op == 'F' || op == 'G') {
			x += l, ++fs;
		}
		if (op == 0) break;
		p = r + 1;
	}
	ex.back().oe = (int32_t)x;
	*n_fs = fs;
	if (x != span) return false;
	if (rev) { // flip to ascending contig coordinates
		std::vector<pg_exon_t> t(ex.size());
		for (size_t i = 0; i < ex.size(); ++i) {
			const pg_exon_t &s = ex[ex.size() - 1 - i];
			t[i].os = (int32_t)(x - s.oe), t[i].oe = (int32_t)(x - s.os);
		}
		ex.swap(t);
	}
	return true;
}

static char *file_label(const char *fn) // read.c:92-105
{
	if (fn == nullptr) return nullptr;
	int32_t len = (int32_t)std::strlen(fn), en = len, st;
	int32_t i = len - 1;
	while (i >= 0 && fn[i] != '/') --i;
	st = i + 1;
	if (en >= 3 && std::strncmp(fn + en - 3, ".gz", 3) == 0) en -= 3;
	if (en >= 4 && std::strncmp(fn + en - 4, ".paf", 4) == 0) en -= 4;
	if (st >= en) return nullptr;
	char *label = (char *)std::calloc((size_t)(en - st + 1), 1);
	std::memcpy(label, fn + st, (size_t)(en - st));
	return label;
}

// One parsed PAF file, self-contained (no global ids yet): parsing is thread-safe and files can be parsed in
// parallel; commit_file() then assigns the global ids sequentially in command-line order, which reproduces the
// reference's first-seen numbering (read.c:151-168) -- pg_hash_uint32(pid) makes the numbering score-relevant.
struct alignas(128) FileParse { // (files next to each other on the command line are parsed side by side: no cache line shared between two of these)
	bool opened = false, ids_only = false;
	char *label = nullptr;
	int32_t n_tot = 0;
	NameDict genes, prots, ctgs;              // local first-seen ids
	std::vector<uint8_t> g_pref, g_incl;      // per local gene (read.c:147-150,158-159)
	std::vector<int32_t> g_len, p_gene, p_len; // gene.len = max protein len (read.c:177); prot.gid, prot.len of the last line
	std::vector<int64_t> ctg_len;
	// pid / cid are LOCAL ids.  Plain malloc'ed arrays: they become the genome's own g->hit / g->exon (freed by pg_data_destroy)
	pg_hit_t *hits = nullptr; int32_t n_hit = 0, m_hit = 0;
	pg_exon_t *exons = nullptr; int32_t n_exon = 0, m_exon = 0;
	std::vector<int32_t> gmap, pmap;          // local -> global ids, -1 = not known yet (resolved at commit time)
	int32_t genome = -1;                      // index of the genome the commit appended
	~FileParse() { std::free(hits); std::free(exons); std::free(label); }
};

template <class T> static inline bool push_raw(T *&a, int32_t &n, int32_t &m, const T &v)
{
	if (n == m || a == nullptr) {
		const int32_t m2 = (m && a) ? m + (m >> 1) : 4096;
		T *b = (T *)std::realloc((void *)a, sizeof(T) * (size_t)m2);
		if (b == nullptr) return false; // out of memory: the element is dropped (the caller reports it)
		a = b, m = m2;
	}
	a[n++] = v;
	return true;
}

// big arrays a genome keeps: ask for huge pages where the kernel hands them out on request only (fewer page faults while the
// parser threads fill them side by side)
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23 /* Linux 5.14 */
#endif
static inline void *big_malloc(size_t bytes)
{
	void *p = std::malloc(bytes);
	if (p == nullptr) return nullptr;
	if (bytes >= ((size_t)4 << 20)) {
		const uintptr_t a = ((uintptr_t)p + 0x1fffff) & ~(uintptr_t)0x1fffff, e = ((uintptr_t)p + bytes) & ~(uintptr_t)0x1fffff;
		if (e > a) (void)madvise((void *)a, e - a, MADV_HUGEPAGE);
	}
	// A hundred parser threads filling fresh arrays take page faults at the rate ONE address space sustains (measured: the summed parse
	// time doubles from 64 to 128 threads, the wall time stays).  The whole array in one call instead of one trap per 4 KiB page
	// (kernels before 5.14 answer EINVAL: nothing lost).
	if (bytes >= ((size_t)128 << 10)) {
		const uintptr_t a = ((uintptr_t)p + 4095) & ~(uintptr_t)4095, e = ((uintptr_t)p + bytes) & ~(uintptr_t)4095;
		if (e > a) (void)madvise((void *)a, e - a, MADV_POPULATE_WRITE);
	}
	return p;
}

static void parse_file(const pg_opt_t *opt, const char *fn, bool ids_only, FileParse &fp)
{
	WholeFile whole(fn); // plain files: one read, lines parsed in place; everything else (gzip, stdin) through zlib
	std::unique_ptr<LineSource> src;
	if (!whole.ok()) src.reset(new LineSource(fn));
	fp.ids_only = ids_only;
	if (!whole.ok() && !src->ok()) return;
	fp.opened = true;
	fp.label = file_label(fn);
	if (!ids_only && fn && std::strcmp(fn, "-") != 0) { // room for the whole file at once (~150 bytes of text a line; a .gz holds 4-5 times its size)
		struct stat sb;
		if (stat(fn, &sb) == 0 && sb.st_size > 0) {
			const size_t len = std::strlen(fn);
			const size_t text = (size_t)sb.st_size * (len > 3 && std::strcmp(fn + len - 3, ".gz") == 0 ? 5 : 1);
			fp.m_hit = (int32_t)std::min<size_t>(text / 120 + 64, (size_t)1 << 30), fp.hits = (pg_hit_t *)big_malloc(sizeof(pg_hit_t) * (size_t)fp.m_hit);
			fp.m_exon = (int32_t)std::min<size_t>(text / 100 + 64, (size_t)1 << 30), fp.exons = (pg_exon_t *)big_malloc(sizeof(pg_exon_t) * (size_t)fp.m_exon);
			if (fp.hits == nullptr) fp.m_hit = 0; // (no room for the estimate: grow on demand)
			if (fp.exons == nullptr) fp.m_exon = 0;
		}
	}
	const NameDict *excl = (const NameDict *)opt->excl, *incl = (const NameDict *)opt->incl, *pref = (const NameDict *)opt->preferred;
	std::vector<int32_t> rank_of; // per local protein: lines seen in this file (read.c:170)
	std::vector<pg_exon_t> ex;
	std::string line, last_name;
	int32_t last_pid = -1, last_gid = -1;
	bool oom = false;
	// the growing arrays and their counters as locals while the file is parsed (written back once, at the end)
	pg_hit_t *hits = fp.hits; int32_t n_hit = fp.n_hit, m_hit = fp.m_hit;
	pg_exon_t *exons = fp.exons; int32_t n_exon = fp.n_exon, m_exon = fp.m_exon;
	int32_t n_tot = 0;
	for (;;) {
		char *s;
		size_t line_len;
		if (whole.ok()) { if (!whole.next(s, line_len)) break; }
		else { if (!src->next(line)) break; s = line.data(), line_len = line.size(); }
		s[line_len] = 0; // (the byte behind a line is the parser's: the '\n' of the file buffer, the terminator of the string)
		++n_tot;
		pg_hit_t hit;
		std::memset(&hit, 0, sizeof(hit));
		hit.pid = hit.pid_dom = hit.cid = hit.off_exon = hit.n_exon = -1;
		int32_t pid = -1, gid = -1, n_fs = -1, n_stop = -1, cig_fs = 0;
		bool have_exons = false;
		char *q = s;
		int32_t col = 0;
		bool dropped = false;
		char *const line_end = s + line_len;
		for (char *p = s;; ++p) {
			p = (char *)std::memchr(p, '\t', (size_t)(line_end - p));
			if (p == nullptr) p = line_end;
			char term = *p;
			*p = 0;
			if (col == 0 && last_pid >= 0 && (size_t)(p - q) == last_name.size() && std::memcmp(q, last_name.data(), last_name.size()) == 0) {
				// the same protein as the line before (PAF files are grouped by protein as a rule): the ids are known, and what the
				// dictionary calls would do again -- preferred / included marks, prot.gid, prot.len = 0 -- has the same outcome
				pid = last_pid, gid = last_gid;
				fp.p_len[(size_t)pid] = 0; // read.c:168
				hit.pid = pid;
				hit.rank = ++rank_of[(size_t)pid];
			} else if (col == 0) { // query name: gene<delim>protein (read.c:139-171)
				char *r = q;
				while (r < p && *r != opt->gene_delim) ++r;
				if (excl && excl->get(q) >= 0) { dropped = true; break; }
				bool has_delim = false;
				if (*r == opt->gene_delim && r < p) has_delim = true, *r = 0;
				if (excl && excl->get(q) >= 0) { dropped = true; break; }
				int32_t is_pref = pref && pref->get(q) >= 0, is_incl = incl && incl->get(q) >= 0;
				bool absent;
				gid = fp.genes.put(q, &absent);
				if (has_delim) *r = (char)opt->gene_delim;
				if (absent) fp.g_pref.push_back(0), fp.g_incl.push_back(0), fp.g_len.push_back(0);
				fp.g_pref[(size_t)gid] = (uint8_t)is_pref, fp.g_incl[(size_t)gid] = (uint8_t)is_incl;
				pid = fp.prots.put(q, &absent);
				if (absent) fp.p_gene.push_back(0), fp.p_len.push_back(0), rank_of.push_back(-1);
				fp.p_gene[(size_t)pid] = gid;
				fp.p_len[(size_t)pid] = 0; // read.c:168
				hit.pid = pid;
				hit.rank = ++rank_of[(size_t)pid];
				last_pid = pid, last_gid = gid, last_name.assign(q, (size_t)(p - q));
			} else if (col == 1) {
				int32_t len = (int32_t)parse_i64(q);
				fp.p_len[(size_t)pid] = len;
				if (fp.g_len[(size_t)gid] < len) fp.g_len[(size_t)gid] = len;
				if (ids_only) { dropped = true; break; }
			} else if (col == 2) hit.qs = (int32_t)parse_i64(q);
			else if (col == 3) {
				hit.qe = (int32_t)parse_i64(q);
				if (hit.qe - hit.qs < fp.p_len[(size_t)pid] * opt->min_prot_ratio) { dropped = true; break; }
			} else if (col == 4) {
				if (*q != '+' && *q != '-') { dropped = true; break; }
				hit.rev = *q == '+' ? 0 : 1;
			} else if (col == 5) { // contig ids are first-seen per file (read.c:190-198)
				bool a2;
				hit.cid = fp.ctgs.put(q, &a2);
				if (a2) fp.ctg_len.push_back(0);
			} else if (col == 6) fp.ctg_len[(size_t)hit.cid] = parse_i64(q);
			else if (col == 7) hit.cs = parse_i64(q);
			else if (col == 8) hit.ce = parse_i64(q);
			else if (col == 9) hit.mlen = (int32_t)parse_i64(q);
			else if (col == 10) {
				hit.blen = (int32_t)parse_i64(q);
				if (hit.mlen < hit.blen * opt->min_prot_iden) { dropped = true; break; }
			} else if (col >= 12) {
				if (std::strncmp(q, "ms:i:", 5) == 0) { // read.c:212-216: long double exp, then truncation
					double div = 1.0 - (double)hit.mlen / hit.blen;
					double uncov = 1.0 - (double)(hit.qe - hit.qs) / fp.p_len[(size_t)pid];
					hit.score_ori = (int32_t)parse_i64(q + 5);
					hit.score_adj = (int32_t)(hit.score_ori * expl(-opt->score_adj_coef * (div + uncov)) + .499);
				} else if (std::strncmp(q, "fs:i:", 5) == 0) n_fs = (int32_t)parse_i64(q + 5);
				else if (std::strncmp(q, "st:i:", 5) == 0) n_stop = (int32_t)parse_i64(q + 5);
				else if (std::strncmp(q, "cg:Z:", 5) == 0) {
					if (cigar_to_exons(q + 5, hit.rev, hit.ce - hit.cs, ex, &cig_fs)) {
						hit.n_exon = (int32_t)ex.size(), hit.off_exon = n_exon, hit.lof = cig_fs;
						for (const pg_exon_t &e : ex) if (!push_raw(exons, n_exon, m_exon, e)) oom = true;
						have_exons = true;
					} else if (pg_verbose >= 1) {
						std::fprintf(stderr, "[W::%s] CIGAR of line %d in '%s' does not span the alignment; hit dropped\n", __func__, n_tot, fn ? fn : "-");
					}
				}
			}
			q = p + 1, ++col;
			if (term == 0) break;
		}
		if (dropped || !have_exons || hit.n_exon < 1) continue;
		int32_t lof = (n_fs > 0 ? n_fs : 0) + (n_stop > 0 ? n_stop : 0); // read.c:230-231
		if (hit.lof < lof) hit.lof = lof;
		hit.cm = middle_cds(hit.cs, exons + hit.off_exon, hit.n_exon);
		if (hit.cm < 0 || oom) continue;
		if (!push_raw(hits, n_hit, m_hit, hit)) oom = true;
	}
	fp.hits = hits, fp.n_hit = n_hit, fp.m_hit = m_hit, fp.exons = exons, fp.n_exon = n_exon, fp.m_exon = m_exon, fp.n_tot = n_tot;
	if (oom && pg_verbose >= 1) std::fprintf(stderr, "[E::%s] out of memory while reading '%s': hits were dropped\n", __func__, fn ? fn : "-");
}

// Names that are in the global dictionaries already get their ids before the commit, from a frozen SNAPSHOT of the dictionaries
// (an immutable map published by the committing thread after the first file and whenever a thousand names have come since):
// look-ups need no lock and never wait for a commit, the commit never waits for a reader.  A pangenome's files share nearly all
// their names, so the sequential part of a batch read shrinks from "every name of every file" to the names a file is the first
// to bring (names missing from the snapshot are resolved by the commit itself).
struct DictSnap { std::unordered_map<std::string_view, int32_t> genes, prots; };
static std::shared_mutex g_dict_mu; // writers of pg_data_t's growing arrays (batch and single-file reads)

static void snap_refresh(const pg_data_t *d, std::shared_ptr<const DictSnap> &slot)
{
	const NameDict *dg = (const NameDict *)d->d_gene, *dp = (const NameDict *)d->d_prot;
	auto s = std::make_shared<DictSnap>();
	s->genes.reserve((size_t)dg->size() * 2), s->prots.reserve((size_t)dp->size() * 2);
	for (int32_t i = 0; i < dg->size(); ++i) s->genes.emplace(std::string_view(dg->name(i)), i);
	for (int32_t i = 0; i < dp->size(); ++i) s->prots.emplace(std::string_view(dp->name(i)), i);
	std::atomic_store(&slot, std::shared_ptr<const DictSnap>(s));
}

static void preresolve(const std::shared_ptr<const DictSnap> &slot, FileParse &fp)
{
	fp.gmap.assign((size_t)fp.genes.size(), -1), fp.pmap.assign((size_t)fp.prots.size(), -1);
	const std::shared_ptr<const DictSnap> s = std::atomic_load(&slot);
	if (!s) return;
	for (int32_t i = 0; i < fp.genes.size(); ++i) { auto it = s->genes.find(std::string_view(fp.genes.name(i))); if (it != s->genes.end()) fp.gmap[(size_t)i] = it->second; }
	for (int32_t i = 0; i < fp.prots.size(); ++i) { auto it = s->prots.find(std::string_view(fp.prots.name(i))); if (it != s->prots.end()) fp.pmap[(size_t)i] = it->second; }
}

// sequential part: global ids in first-seen order (the numbering of per-line dict_put calls, read.c:151-168), genome appended to `d`
static int32_t commit_ids(pg_data_t *d, FileParse &fp)
{
	if (!fp.opened) return -1;
	NameDict *dg = (NameDict *)d->d_gene, *dp = (NameDict *)d->d_prot, *dc = (NameDict *)d->d_ctg;
	DataExt *ext = ext_of(d, true);
	std::unique_lock<std::shared_mutex> lk(g_dict_mu);
	grow0(d->genome, d->n_genome, d->m_genome);
	fp.genome = d->n_genome;
	pg_genome_t *g = &d->genome[d->n_genome++];
	std::memset(g, 0, sizeof(*g));
	g->label = fp.label, fp.label = nullptr;
	if (ext->is_local.size() < (size_t)d->n_genome) ext->is_local.resize((size_t)d->n_genome, 0);
	if (ext->hits_sorted.size() < (size_t)d->n_genome) ext->hits_sorted.resize((size_t)d->n_genome, 0);
	ext->is_local[(size_t)d->n_genome - 1] = fp.ids_only ? 0 : 1;
	if (fp.gmap.size() != (size_t)fp.genes.size()) fp.gmap.assign((size_t)fp.genes.size(), -1);
	if (fp.pmap.size() != (size_t)fp.prots.size()) fp.pmap.assign((size_t)fp.prots.size(), -1);
	// genes then proteins, each in the order this file saw them first
	for (int32_t i = 0; i < fp.genes.size(); ++i) {
		int32_t gid = fp.gmap[(size_t)i];
		if (gid < 0) {
			bool absent;
			gid = dg->put(fp.genes.name(i), &absent);
			if (absent) { d->n_gene++; grow0(d->gene, gid, d->m_gene); }
			d->gene[gid].name = dg->name(gid);
			fp.gmap[(size_t)i] = gid;
		}
		d->gene[gid].preferred = fp.g_pref[(size_t)i], d->gene[gid].included = fp.g_incl[(size_t)i];
		if ((int32_t)d->gene[gid].len < fp.g_len[(size_t)i]) d->gene[gid].len = (uint32_t)fp.g_len[(size_t)i];
	}
	for (int32_t i = 0; i < fp.prots.size(); ++i) {
		int32_t pid = fp.pmap[(size_t)i];
		if (pid < 0) {
			bool absent;
			pid = dp->put(fp.prots.name(i), &absent);
			if (absent) { d->n_prot++; grow0(d->prot, pid, d->m_prot); }
			d->prot[pid].name = dp->name(pid);
			fp.pmap[(size_t)i] = pid;
		}
		d->prot[pid].gid = fp.gmap[(size_t)fp.p_gene[(size_t)i]];
		d->prot[pid].len = fp.p_len[(size_t)i];
	}
	if (!fp.ids_only) {
		g->n_ctg = g->m_ctg = fp.ctgs.size();
		g->ctg = (pg_ctg_t *)std::calloc((size_t)(g->n_ctg > 0 ? g->n_ctg : 1), sizeof(pg_ctg_t));
		for (int32_t c = 0; c < g->n_ctg; ++c) {
			bool a2;
			g->ctg[c].name = dc->name(dc->put(fp.ctgs.name(c), &a2));
			g->ctg[c].len = fp.ctg_len[(size_t)c];
		}
	}
	if (pg_verbose >= 3)
		std::fprintf(stderr, "[M::%s::%s] [%d] %s: %d lines parsed, %d hits kept%s\n", "pg_read_paf", stamp(), d->n_genome - 1,
		             g->label ? g->label : "-", fp.n_tot, fp.ids_only ? 0 : fp.n_hit, fp.ids_only ? " (ids only; hits owned by another shard)" : "");
	return 0;
}

// per-genome part, any thread (the genome's slot in d->genome exists and does not move: batch reads reserve the array first): the
// parsed arrays become the genome's own, the local protein ids are replaced by the global ones in place
static void finalize_genome(pg_data_t *d, FileParse &fp)
{
	if (!fp.opened || fp.ids_only || fp.genome < 0) return;
	pg_genome_t *g = &d->genome[fp.genome];
	for (int32_t i = 0; i < fp.n_hit; ++i) fp.hits[i].pid = fp.pmap[(size_t)fp.hits[i].pid];
	// (an estimate that was far too generous -- a .gz that compressed badly -- is given back: realloc in place, no copy)
	if (fp.hits && (size_t)fp.m_hit > (size_t)fp.n_hit * 2 + 4096) { pg_hit_t *t = (pg_hit_t *)std::realloc((void *)fp.hits, sizeof(pg_hit_t) * (size_t)(fp.n_hit + 1)); if (t) fp.hits = t, fp.m_hit = fp.n_hit + 1; }
	if (fp.exons && (size_t)fp.m_exon > (size_t)fp.n_exon * 2 + 4096) { pg_exon_t *t = (pg_exon_t *)std::realloc((void *)fp.exons, sizeof(pg_exon_t) * (size_t)(fp.n_exon + 1)); if (t) fp.exons = t, fp.m_exon = fp.n_exon + 1; }
	g->n_hit = fp.n_hit, g->m_hit = fp.m_hit > 0 ? fp.m_hit : 1;
	g->hit = fp.hits ? fp.hits : (pg_hit_t *)std::malloc(sizeof(pg_hit_t));
	g->n_exon = fp.n_exon, g->m_exon = fp.m_exon > 0 ? fp.m_exon : 1;
	g->exon = fp.exons ? fp.exons : (pg_exon_t *)std::malloc(sizeof(pg_exon_t));
	fp.hits = nullptr, fp.exons = nullptr, fp.n_hit = fp.m_hit = fp.n_exon = fp.m_exon = 0;
}

static int32_t read_paf_impl(const pg_opt_t *opt, pg_data_t *d, const char *fn, bool ids_only)
{
	FileParse fp;
	parse_file(opt, fn, ids_only, fp);
	int32_t rc = commit_ids(d, fp);
	if (rc == 0) finalize_genome(d, fp);
	if (rc == 0 && !ids_only) pack_genomes(d, ext_of(d, true), d->n_genome - 1, d->n_genome); // SoA block for the backend, while the next file is read
	return rc;
}

} // namespace pgx

using namespace pgx;

extern "C" {

pg_data_t *pg_data_init(void)
{
	pg_data_t *d = (pg_data_t *)std::calloc(1, sizeof(pg_data_t));
	d->d_ctg = new NameDict(), d->d_gene = new NameDict(), d->d_prot = new NameDict();
	return d;
}

void pg_data_destroy(pg_data_t *d)
{
	if (d == nullptr) return;
	ext_drop(d);
	for (int32_t i = 0; i < d->n_genome; ++i) {
		pg_genome_t *g = &d->genome[i];
		std::free(g->ctg); std::free(g->hit); std::free(g->exon); std::free(g->label);
	}
	std::free(d->genome); std::free(d->gene); std::free(d->prot);
	delete (NameDict *)d->d_ctg; delete (NameDict *)d->d_gene; delete (NameDict *)d->d_prot;
	std::free(d);
}

int32_t pg_read_paf(const pg_opt_t *opt, pg_data_t *d, const char *fn) { return read_paf_impl(opt, d, fn, false); }
int32_t pg_scan_paf_ids(const pg_opt_t *opt, pg_data_t *d, const char *fn) { return read_paf_impl(opt, d, fn, true); }

// parse many PAFs on host threads, commit them in command-line order (ids identical to n sequential
// pg_read_paf / pg_scan_paf_ids calls).  ids_only[i] != 0: register names only (the hits belong to another shard).
// A pipeline, not three barriers: every thread parses files (and looks the names it can up in the global dictionaries); whoever
// finds the next file of the command line parsed commits its ids (sequential, but short: only new names are inserted); a
// committed file's hits are then finished -- global protein ids, the SoA block in pinned memory for the backend -- by any
// thread.  Memory holds the files that are parsed but not finished yet, not the whole batch twice.
int32_t pg_read_paf_batch(const pg_opt_t *opt, pg_data_t *d, int32_t n, const char *const *fns, const uint8_t *ids_only, int32_t n_threads)
{
	if (n <= 0) return 0;
	if (n_threads <= 0) {
		const char *e = std::getenv("PANGENE_READ_THREADS");
		n_threads = e && std::atoi(e) > 0 ? std::atoi(e) : (int32_t)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 64u); // (beyond that the threads of one address space get in each other's way: page faults, allocator)
	}
	if (n_threads > n) n_threads = n;
	DataExt *ext = ext_of(d, true);
	const int32_t j0 = d->n_genome;
	{ // the arrays the per-genome part indexes must not move while it runs
		std::unique_lock<std::shared_mutex> lk(g_dict_mu);
		if (d->m_genome < j0 + n) {
			const int32_t old = d->m_genome;
			d->m_genome = j0 + n + 16;
			d->genome = (pg_genome_t *)std::realloc((void *)d->genome, sizeof(pg_genome_t) * (size_t)d->m_genome);
			std::memset((void *)(d->genome + old), 0, sizeof(pg_genome_t) * (size_t)(d->m_genome - old));
		}
		ext->is_local.resize((size_t)(j0 + n), 0), ext->hits_sorted.resize((size_t)(j0 + n), 0);
		ext->packs.resize((size_t)(j0 + n));
	}
	std::thread slab_helper;
	{ // the blocks of the local files take about 0.4 bytes per byte of PAF text (44 B a hit + 8 B an exon against ~150 B a line)
		size_t text = 0;
		for (int32_t i = 0; i < n; ++i) {
			struct stat sb;
			if (!(ids_only && ids_only[i]) && fns[i] && stat(fns[i], &sb) == 0) text += (size_t)sb.st_size;
		}
		if (text > ((size_t)64 << 20)) slab_prefetch(std::min<size_t>(text / 5 * 2, (size_t)192 << 20), &slab_helper); // (up to the budget of freshly locked memory: block_alloc)
	}
	std::vector<FileParse> fp((size_t)n);
	std::vector<std::atomic<uint8_t>> state((size_t)n); // 0 new, 1 parsed, 2 committed (ids final), 3 being / has been finished
	for (auto &x : state) x.store(0);
	std::atomic<int32_t> next_parse{0}, n_commit{0}, next_final{0}, n_fail{0};
	static const bool timing = std::getenv("PANGENE_TIMING") != nullptr;
	std::atomic<int64_t> us_parse{0}, us_resolve{0}, us_commit{0}, us_final{0};
	const double t_batch0 = now_sec();
	std::mutex commit_mu;
	std::shared_ptr<const DictSnap> snap; // published with atomic_store by whoever commits
	int32_t snap_names = 0;
	auto try_commit = [&]() {
		std::unique_lock<std::mutex> lk(commit_mu, std::try_to_lock);
		if (!lk.owns_lock()) return; // somebody else is at it (and will see what this thread just parsed: it re-checks before it leaves)
		for (;;) {
			const int32_t k = n_commit.load();
			if (k >= n || state[(size_t)k].load() != 1) break;
			const double tc0 = timing ? now_sec() : 0.0;
			if (commit_ids(d, fp[(size_t)k]) != 0) n_fail.fetch_add(1);
			// (a rebuild copies every name: geometrically spaced, so that files that all bring new names do not make the commits quadratic)
			if (d->n_gene + d->n_prot >= snap_names + std::max(1000, snap_names / 4) || (k == 0 && d->n_gene + d->n_prot > 0)) snap_refresh(d, snap), snap_names = d->n_gene + d->n_prot;
			if (timing) us_commit.fetch_add((int64_t)((now_sec() - tc0) * 1e6));
			state[(size_t)k].store(2);
			n_commit.store(k + 1);
		}
	};
	auto try_final = [&]() -> bool { // finish one committed file, if there is one
		for (;;) {
			int32_t k = next_final.load();
			if (k >= n || state[(size_t)k].load() < 2) return false;
			if (!next_final.compare_exchange_weak(k, k + 1)) continue;
			FileParse &f = fp[(size_t)k];
			const double tf0 = timing ? now_sec() : 0.0;
			finalize_genome(d, f);
			if (f.opened && !f.ids_only && f.genome >= 0) pack_genomes(d, ext, f.genome, f.genome + 1, 1.0 / n_threads); // (one genome: on this thread)
			{ FileParse done; std::swap(done.genes, f.genes), std::swap(done.prots, f.prots), std::swap(done.ctgs, f.ctgs); } // the names are not needed any more
			if (timing) us_final.fetch_add((int64_t)((now_sec() - tf0) * 1e6));
			state[(size_t)k].store(3);
			return true;
		}
	};
	auto work = [&]() {
		for (;;) {
			const int32_t i = next_parse.fetch_add(1);
			if (i < n) {
				const double tp0 = timing ? now_sec() : 0.0;
				parse_file(opt, fns[i], ids_only && ids_only[i], fp[(size_t)i]);
				const double tp1 = timing ? now_sec() : 0.0;
				if (fp[(size_t)i].opened) preresolve(snap, fp[(size_t)i]);
				if (timing) us_parse.fetch_add((int64_t)((tp1 - tp0) * 1e6)), us_resolve.fetch_add((int64_t)((now_sec() - tp1) * 1e6));
				state[(size_t)i].store(1);
				try_commit();
				// (a commit that was skipped because another thread held the lock: that thread may have left just before this
				// file's state changed -- look again once)
				if (n_commit.load() < n && state[(size_t)n_commit.load()].load() == 1) try_commit();
				while (try_final()) { }
				continue;
			}
			if (next_final.load() >= n) break;
			// nothing left to parse: help with what is ready, else SLEEP -- a hundred threads spinning on yield() here slowed the one
			// that commits (and everybody's page faults) several times over on a 256-thread host
			const int32_t kc = n_commit.load();
			if (kc < n && state[(size_t)kc].load() == 1) try_commit();
			if (!try_final()) std::this_thread::sleep_for(std::chrono::microseconds(100));
		}
	};
	std::vector<std::thread> th;
	for (int32_t t = 1; t < n_threads; ++t) th.emplace_back(work);
	work();
	for (auto &x : th) x.join();
	const double t_joined = now_sec();
	if (slab_helper.joinable()) slab_helper.join();
	if (timing) std::fprintf(stderr, "[pg_read_paf_batch] %d files on %d threads: %.1f ms wall (+ %.1f ms for the slab helper); summed over the threads: parsing %.1f ms, name look-ups %.1f ms, finishing (ids in place + SoA block) %.1f ms; the sequential commits %.1f ms\n",
	                         n, n_threads, (t_joined - t_batch0) * 1e3, (now_sec() - t_joined) * 1e3, us_parse.load() * 1e-3, us_resolve.load() * 1e-3, us_final.load() * 1e-3, us_commit.load() * 1e-3);
	ext->check_strand = !!(opt->flag & PG_F_CHECK_STRAND), ext->min_ov_ratio = opt->min_ov_ratio;
	exact_prefetch(d, ext); // the replay of the reference's tie order starts in the background
	return -n_fail.load();
}

// "-X a,b,c" or "-X @file" (first token of each line) -> name set (read.c:265-318)
void *pg_read_list_dict(const char *o)
{
	NameDict *nd = new NameDict();
	if (o == nullptr) return nd;
	if (*o != '@') {
		const char *q = o;
		for (const char *p = o;; ++p) {
			if (*p == ',' || *p == ' ' || *p == '\t' || *p == 0) {
				if (p > q) nd->put(std::string_view(q, (size_t)(p - q)), nullptr);
				if (*p == 0) break;
				q = p + 1;
			}
		}
	} else {
		LineSource src(o + 1);
		if (!src.ok()) { delete nd; return nullptr; }
		std::string line;
		while (src.next(line)) {
			size_t n = 0;
			while (n < line.size() && !std::isspace((unsigned char)line[n])) ++n;
			nd->put(std::string_view(line.data(), n), nullptr);
		}
	}
	return nd;
}

void pg_dict_destroy(void *h) { delete (NameDict *)h; }

} // extern "C"
