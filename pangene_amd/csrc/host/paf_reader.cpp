// paf_reader.cpp -- host-side PAF ingest for the drop-in surface: pg_data_init/destroy, pg_read_paf,
// pg_scan_paf_ids, pg_read_list_dict.  Text parsing is outside the accelerated path (SURVEY.md section 2:
// "must be rebuilt on host"); what matters here is that ids, ranks, exon lists, cm and score_adj come
// out exactly as the reference's reader produces them (read.c:107-236, hit.c:14-27), because
// pg_hash_uint32(pid) and the first-seen numbering are score-relevant downstream.
#include <fcntl.h>
#include <cerrno>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <sched.h>
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

unsigned host_threads(unsigned cap)
{
	static const unsigned budget = [] {
		unsigned n = std::max(1u, std::thread::hardware_concurrency());
		cpu_set_t set;
		if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) n = std::min<unsigned>(n, (unsigned)c); }
		long long quota = -1, period = -1;
		if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) { // cgroup v2: "<quota|max> <period>"
			char q[32] = {0};
			if (std::fscanf(f, "%31s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0) quota = std::atoll(q);
			std::fclose(f);
		} else { // cgroup v1
			if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(g, "%lld", &quota) != 1) quota = -1; std::fclose(g); }
			if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(g, "%lld", &period) != 1) period = -1; std::fclose(g); }
		}
		if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
		if (const char *e = std::getenv("PANGENE_HOST_THREADS")) if (std::atoi(e) > 0) n = (unsigned)std::atoi(e);
		return n;
	}();
	return std::max(1u, std::min(budget, cap));
}

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
	for (DataExt::HostArena &a : e->arenas) if (a.map) munmap(a.map, a.map_bytes);
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
		while (got < n) { const ssize_t k = read(fd, tl_buf.data() + got, n - got); if (k < 0 && errno == EINTR) continue; if (k <= 0) break; got += (size_t)k; }
		close(fd);
		if (got != n) return; // a short or failed read: not "the file" -- LineSource reads it line by line and reports what it finds
		p_ = tl_buf.data(), end_ = p_ + got, ok_ = true;
	}
	bool ok() const { return ok_; }
	size_t count_lines() const { // an upper bound of the hits the file holds
		size_t n = 0;
		for (const char *q = p_; q < end_;) { const char *nl = (const char *)std::memchr(q, '\n', (size_t)(end_ - q)); ++n; if (!nl) break; q = nl + 1; }
		return n;
	}
	size_t text_bytes() const { return (size_t)(end_ - p_); }
	// exons per line, from the head of the file: every intron operation of a CIGAR (N, U, V) opens one (names hold such letters
	// too: the estimate errs on the generous side, which is the side that costs nothing)
	double exons_per_line() const {
		const char *e = end_ - p_ > (64 << 10) ? p_ + (64 << 10) : end_;
		size_t ops = 0, lines = 1;
		for (const char *q = p_; q < e; ++q) { const char ch = *q; ops += ch == 'N' || ch == 'U' || ch == 'V'; lines += ch == '\n'; }
		return 1.0 + (double)ops / (double)lines;
	}
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

// read.c:216: score_adj = (int32_t)(score_ori * expl(x) + .499), long double arithmetic.  expl costs ~250 cycles a line -- a fifth of the
// whole parse -- and only the INTEGER part of the result is kept: exp() in double (error < 1 ulp) gives the same integer unless the
// value lies within a few 1e-16 (relative) of a whole number; only then (about once in 1e7 lines) the long double route decides.
static inline int32_t score_adjusted(int32_t score_ori, double x)
{
	const double t = (double)score_ori * std::exp(x) + .499, a = std::fabs(t);
	if (a < 2.0e9 && std::isfinite(t)) {
		const double nearest = std::nearbyint(t), band = a * 4.0e-15 + 1.0e-12; // (double rounding of exp, of the product and of the sum: < 4 ulp of t, with room)
		if (std::fabs(t - nearest) > band) return (int32_t)t;
	}
	return (int32_t)(score_ori * expl(x) + .499);
}

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
		const char *r = p;
		int64_t l = 0;
		if ((unsigned)(*r - '0') < 10u) { // the usual op: plain digits (strtol's other forms -- blanks, a sign -- below)
			uint64_t v = 0;
			int nd = 0;
			while ((unsigned)(*r - '0') < 10u && nd < 18) v = v * 10 + (uint64_t)(*r - '0'), ++r, ++nd;
			l = (int64_t)v;
			if ((unsigned)(*r - '0') < 10u) { char *e; l = std::strtol(p, &e, 10); r = e; } // 19+ digits: strtol's saturation
		} else { char *e; l = std::strtol(p, &e, 10); r = e; }
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
	if (rev) { // flip to ascending contig coordinates (in place: element i <-> n - 1 - i)
		const size_t n = ex.size();
		for (size_t i = 0; i < n / 2; ++i) {
			const pg_exon_t a = ex[i], b = ex[n - 1 - i];
			ex[i].os = (int32_t)(x - b.oe), ex[i].oe = (int32_t)(x - b.os);
			ex[n - 1 - i].os = (int32_t)(x - a.oe), ex[n - 1 - i].oe = (int32_t)(x - a.os);
		}
		if (n & 1) { const pg_exon_t a = ex[n / 2]; ex[n / 2].os = (int32_t)(x - a.oe), ex[n / 2].oe = (int32_t)(x - a.os); }
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

// Names that are in the global dictionaries already get their ids before the commit, from a frozen SNAPSHOT of the dictionaries
// (an immutable map published by the committing thread after the first file and whenever a thousand names have come since):
// look-ups need no lock and never wait for a commit, the commit never waits for a reader.  A pangenome's files share nearly all
// their names, so the sequential part of a batch read shrinks from "every name of every file" to the names a file is the first
// to bring (names missing from the snapshot are resolved by the commit itself).
// The snapshot also keeps the ATTRIBUTES the commit would assign (gene: preferred / included / len, protein: gene / len; read.c:147-177):
// an entry of a file that would assign exactly what is there already is left out of the commit altogether.  What the commits change
// after a snapshot goes into a change log (ChangeLog, appended under the commit lock); a file resolved against a snapshot that is
// `k` changes old re-applies its own entries for those k ids -- the sequential part of a batch read is then "the names and values a
// file is the first to bring", not "every name of every file" (0.16-0.47 s of a 0.5 s read of 1250 files before).
struct DictSnap {
	FlatIndex genes, prots; // (the names themselves stay in the global dictionaries: they never move)
	std::vector<uint8_t> g_pref, g_incl; std::vector<uint32_t> g_len; std::vector<int32_t> p_gid, p_len;
	int64_t log_pos = 0;
};
struct ChangeLog { std::vector<std::pair<uint8_t, int32_t>> v; }; // (0 gene | 1 protein, global id): an attribute of an id that existed before got another value
// A file's genes (or proteins) under LOCAL ids = the order the file saw them first.  A name the snapshot holds is never copied or hashed
// into a table of the file's own -- its local id hangs on its global id through an array -- only names new to the snapshot go into a
// small dictionary.  (A bacterial genome brings 10 000 names, all but a few known after the first file: building two dictionaries
// per file was a fifth of the parse.)
// global id -> local id, open addressing over 32-bit keys (the table a file needs is as large as the file's own list of names, whatever the
// size of the dictionary: an array over the whole snapshot per file was O(files x names) of memset and of memory for files in flight)
struct IdMap {
	std::vector<int32_t> key, val; size_t n = 0;
	void clear() { key.clear(), val.clear(), n = 0; }
	static size_t h(int32_t k, size_t mask) { return ((uint32_t)k * 2654435761u >> 7) & mask; }
	int32_t get(int32_t k) const {
		if (key.empty()) return -1;
		const size_t mask = key.size() - 1;
		for (size_t i = h(k, mask);; i = (i + 1) & mask) { if (key[i] == k) return val[i]; if (key[i] < 0) return -1; }
	}
	void put(int32_t k, int32_t v) { // k is not in the table
		if (2 * (n + 1) > key.size()) {
			std::vector<int32_t> ok, ov; ok.swap(key), ov.swap(val);
			const size_t cap = ok.empty() ? 1024 : 2 * ok.size();
			key.assign(cap, -1), val.assign(cap, -1);
			for (size_t i = 0; i < ok.size(); ++i) if (ok[i] >= 0) { size_t j = h(ok[i], cap - 1); while (key[j] >= 0) j = (j + 1) & (cap - 1); key[j] = ok[i], val[j] = ov[i]; }
		}
		const size_t mask = key.size() - 1;
		size_t i = h(k, mask);
		while (key[i] >= 0) i = (i + 1) & mask;
		key[i] = k, val[i] = v, ++n;
	}
};
struct LocalNames {
	const FlatIndex *known = nullptr;     // the snapshot's index at parse time (its ids are global ids), or NULL
	IdMap lid_of_global;                  // global id -> local id for the names of `known` this file has seen (sized by the file, not by the dictionary)
	NameDict fresh;                       // the names `known` does not hold, first-seen order
	std::vector<int32_t> lid_of_fresh;    // fresh id -> local id
	std::vector<int32_t> global;          // local id -> global id, -1 = not known yet (resolved by preresolve() or the commit)
	std::vector<int32_t> fresh_of;        // local id -> fresh id, -1 = a known name
	int32_t size() const { return (int32_t)global.size(); }
	void bind(const FlatIndex *k) { known = k; lid_of_global.clear(); }
	int32_t put(std::string_view s, int32_t known_id, bool *absent) { // known_id: what known->find(s) gave (the caller has looked)
		if (known_id >= 0) {
			int32_t l = lid_of_global.get(known_id);
			*absent = l < 0;
			if (l < 0) l = (int32_t)global.size(), lid_of_global.put(known_id, l), global.push_back(known_id), fresh_of.push_back(-1);
			return l;
		}
		const int32_t f = fresh.put(s, absent);
		if (*absent) lid_of_fresh.push_back((int32_t)global.size()), global.push_back(-1), fresh_of.push_back(f);
		return lid_of_fresh[(size_t)f];
	}
	std::string_view view(int32_t lid) const { return fresh_of[(size_t)lid] >= 0 ? fresh.view(fresh_of[(size_t)lid]) : known->view(global[(size_t)lid]); }
	const char *name(int32_t lid) const { return fresh_of[(size_t)lid] >= 0 ? fresh.name(fresh_of[(size_t)lid]) : known->name(global[(size_t)lid]); }
	int32_t local_of(int32_t gid, std::string_view nm) const { // the file's entry for global id `gid` (whose name is nm), -1 = none
		if (known && gid < known->size()) return lid_of_global.get(gid);
		const int32_t f = fresh.get(nm);
		return f < 0 ? -1 : lid_of_fresh[(size_t)f];
	}
};

// One parsed PAF file, self-contained (no global ids yet): parsing is thread-safe and files can be parsed in
// parallel; commit_file() then assigns the global ids sequentially in command-line order, which reproduces the
// reference's first-seen numbering (read.c:151-168) -- pg_hash_uint32(pid) makes the numbering score-relevant.
struct alignas(128) FileParse { // (files next to each other on the command line are parsed side by side: no cache line shared between two of these)
	bool opened = false, ids_only = false;
	bool failed = false; // out of memory while the arrays grew: hits are missing, the file must not be committed as if it were whole
	char *label = nullptr;
	int32_t n_tot = 0;
	LocalNames genes, prots;                   // local first-seen ids
	NameDict ctgs;
	std::shared_ptr<const DictSnap> snap;      // what genes / prots are bound to
	std::vector<uint8_t> g_pref, g_incl;      // per local gene (read.c:147-150,158-159)
	std::vector<int32_t> g_len, p_gene, p_len; // gene.len = max protein len (read.c:177); prot.gid, prot.len of the last line
	std::vector<int64_t> ctg_len;
	// pid / cid are LOCAL ids.  Plain malloc'ed arrays: they become the genome's own g->hit / g->exon (freed by pg_data_destroy)
	pg_hit_t *hits = nullptr; int32_t n_hit = 0, m_hit = 0;
	pg_exon_t *exons = nullptr; int32_t n_exon = 0, m_exon = 0;
	bool hits_arena = false, exons_arena = false; // the array lies in the batch read's arena (DataExt::arenas): never free()'d or realloc'ed
	DataExt::HostArena *arena = nullptr;       // where a batch read's files take their arrays from (NULL: malloc)
	// what preresolve() left for the commit: the local genes / proteins whose names or attributes the dictionary snapshot does not
	// already hold exactly as this file would set them; snap_log = position of the change log the snapshot was taken at (-1: no snapshot)
	std::vector<int32_t> todo_g, todo_p;
	int64_t snap_log = -1;
	int32_t genome = -1;                      // index of the genome the commit appended
	~FileParse() { if (!hits_arena) std::free(hits); if (!exons_arena) std::free(exons); std::free(label); }
};

template <class T> static inline bool push_raw(T *&a, int32_t &n, int32_t &m, const T &v, bool &in_arena)
{
	if (n == m || a == nullptr) {
		const int32_t m2 = (m && a) ? m + (m >> 1) : 4096;
		T *b;
		if (in_arena && a) { // an arena slice that turned out too small: the array moves to the heap (the slice is left behind)
			b = (T *)std::malloc(sizeof(T) * (size_t)m2);
			if (b) std::memcpy((void *)b, (const void *)a, sizeof(T) * (size_t)n), in_arena = false;
		} else b = (T *)std::realloc((void *)a, sizeof(T) * (size_t)m2);
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

static void parse_file(const pg_opt_t *opt, const char *fn, bool ids_only, FileParse &fp, const std::shared_ptr<const DictSnap> &snap_now = nullptr)
{
	fp.snap = snap_now;
	fp.genes.bind(fp.snap ? &fp.snap->genes : nullptr), fp.prots.bind(fp.snap ? &fp.snap->prots : nullptr);
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
			fp.m_hit = (int32_t)std::min<size_t>(text / 120 + 64, (size_t)1 << 30), fp.m_exon = (int32_t)std::min<size_t>(text / 100 + 64, (size_t)1 << 30);
			if (fp.arena && whole.ok()) { // a batch read: slices of its huge-page arena, the hits' by the exact number of lines
				fp.m_hit = (int32_t)std::min<size_t>(whole.count_lines() + 1, (size_t)1 << 30);
				fp.m_exon = (int32_t)std::min<size_t>((size_t)((double)fp.m_hit * whole.exons_per_line() * 1.15) + 1024, (size_t)1 << 30);
				const size_t hb = (sizeof(pg_hit_t) * (size_t)fp.m_hit + 63) & ~(size_t)63, eb = (sizeof(pg_exon_t) * (size_t)fp.m_exon + 63) & ~(size_t)63;
				const size_t at = fp.arena->used.fetch_add(hb + eb);
				if (at + hb + eb <= fp.arena->bytes) fp.hits = (pg_hit_t *)(fp.arena->base + at), fp.exons = (pg_exon_t *)(fp.arena->base + at + hb), fp.hits_arena = fp.exons_arena = true;
			}
			if (fp.hits == nullptr) fp.hits = (pg_hit_t *)big_malloc(sizeof(pg_hit_t) * (size_t)fp.m_hit);
			if (fp.exons == nullptr) fp.exons = (pg_exon_t *)big_malloc(sizeof(pg_exon_t) * (size_t)fp.m_exon);
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
	bool hits_arena = fp.hits_arena, exons_arena = fp.exons_arena;
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
			// the numeric columns (2-4, 7-11), as they nearly always are -- digits up to the tab: value and field end in one walk
			// (anything else -- blanks, a sign, 19 digits, digits followed by something -- takes the general way: strtol's rules)
			int64_t num = 0;
			bool have_num = false;
			if (col >= 1 && col <= 10 && col != 4 && col != 5 && (unsigned)(*p - '0') < 10u) {
				const char *r = p;
				uint64_t v = 0;
				int nd = 0;
				while ((unsigned)(*r - '0') < 10u && nd < 18) v = v * 10 + (uint64_t)(*r - '0'), ++r, ++nd;
				if (r == line_end || *r == '\t') p = (char *)r, num = (int64_t)v, have_num = true;
			}
			if (!have_num) while (p < line_end && *p != '\t') ++p; // (fields are a few bytes long: a loop beats a call)
			char term = *p;
			*p = 0;
#define PAF_NUM(q_) (have_num ? num : parse_i64(q_))
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
				if (has_delim) *r = (char)opt->gene_delim;
				const std::string_view gv(q, (size_t)(r - q)), pv(q, (size_t)(p - q));
				// the protein first: the snapshot knows its gene (the same name gives the same gene: checked against the prefix)
				int32_t kp = -1, kg = -1;
				if (fp.snap) {
					kp = fp.snap->prots.find(pv);
					if (kp >= 0) { const int32_t g0 = fp.snap->p_gid[(size_t)kp]; if (g0 >= 0 && g0 < fp.snap->genes.size() && fp.snap->genes.view(g0) == gv) kg = g0; }
					if (kg < 0) kg = fp.snap->genes.find(gv);
				}
				gid = fp.genes.put(gv, kg, &absent);
				if (absent) fp.g_pref.push_back(0), fp.g_incl.push_back(0), fp.g_len.push_back(0);
				fp.g_pref[(size_t)gid] = (uint8_t)is_pref, fp.g_incl[(size_t)gid] = (uint8_t)is_incl;
				pid = fp.prots.put(pv, kp, &absent);
				if (absent) fp.p_gene.push_back(0), fp.p_len.push_back(0), rank_of.push_back(-1);
				fp.p_gene[(size_t)pid] = gid;
				fp.p_len[(size_t)pid] = 0; // read.c:168
				hit.pid = pid;
				hit.rank = ++rank_of[(size_t)pid];
				last_pid = pid, last_gid = gid, last_name.assign(q, (size_t)(p - q));
			} else if (col == 1) {
				int32_t len = (int32_t)PAF_NUM(q);
				fp.p_len[(size_t)pid] = len;
				if (fp.g_len[(size_t)gid] < len) fp.g_len[(size_t)gid] = len;
				if (ids_only) { dropped = true; break; }
			} else if (col == 2) hit.qs = (int32_t)PAF_NUM(q);
			else if (col == 3) {
				hit.qe = (int32_t)PAF_NUM(q);
				if (hit.qe - hit.qs < fp.p_len[(size_t)pid] * opt->min_prot_ratio) { dropped = true; break; }
			} else if (col == 4) {
				if (*q != '+' && *q != '-') { dropped = true; break; }
				hit.rev = *q == '+' ? 0 : 1;
			} else if (col == 5) { // contig ids are first-seen per file (read.c:190-198)
				bool a2;
				hit.cid = fp.ctgs.put(q, &a2);
				if (a2) fp.ctg_len.push_back(0);
			} else if (col == 6) fp.ctg_len[(size_t)hit.cid] = PAF_NUM(q);
			else if (col == 7) hit.cs = PAF_NUM(q);
			else if (col == 8) hit.ce = PAF_NUM(q);
			else if (col == 9) hit.mlen = (int32_t)PAF_NUM(q);
			else if (col == 10) {
				hit.blen = (int32_t)PAF_NUM(q);
				if (hit.mlen < hit.blen * opt->min_prot_iden) { dropped = true; break; }
			} else if (col >= 12) {
				const bool tag5 = p - q >= 5 && q[2] == ':' && q[4] == ':';
				if (!tag5) { }
				else if (q[0] == 'm' && q[1] == 's' && q[3] == 'i') { // read.c:212-216: long double exp, then truncation
					double div = 1.0 - (double)hit.mlen / hit.blen;
					double uncov = 1.0 - (double)(hit.qe - hit.qs) / fp.p_len[(size_t)pid];
					hit.score_ori = (int32_t)parse_i64(q + 5);
					hit.score_adj = score_adjusted(hit.score_ori, -opt->score_adj_coef * (div + uncov));
				} else if (q[0] == 'f' && q[1] == 's' && q[3] == 'i') n_fs = (int32_t)parse_i64(q + 5);
				else if (q[0] == 's' && q[1] == 't' && q[3] == 'i') n_stop = (int32_t)parse_i64(q + 5);
				else if (q[0] == 'c' && q[1] == 'g' && q[3] == 'Z') {
					if (cigar_to_exons(q + 5, hit.rev, hit.ce - hit.cs, ex, &cig_fs)) {
						hit.n_exon = (int32_t)ex.size(), hit.off_exon = n_exon, hit.lof = cig_fs;
						for (const pg_exon_t &e : ex) if (!push_raw(exons, n_exon, m_exon, e, exons_arena)) oom = true;
						have_exons = true;
					} else if (pg_verbose >= 1) {
						std::fprintf(stderr, "[W::%s] CIGAR of line %d in '%s' does not span the alignment; hit dropped\n", __func__, n_tot, fn ? fn : "-");
					}
				}
			}
#undef PAF_NUM
			q = p + 1, ++col;
			if (term == 0) break;
		}
		if (dropped || !have_exons || hit.n_exon < 1) continue;
		int32_t lof = (n_fs > 0 ? n_fs : 0) + (n_stop > 0 ? n_stop : 0); // read.c:230-231
		if (hit.lof < lof) hit.lof = lof;
		hit.cm = middle_cds(hit.cs, exons + hit.off_exon, hit.n_exon);
		if (hit.cm < 0 || oom) continue;
		if (!push_raw(hits, n_hit, m_hit, hit, hits_arena)) oom = true;
	}
	fp.hits = hits, fp.n_hit = n_hit, fp.m_hit = m_hit, fp.exons = exons, fp.n_exon = n_exon, fp.m_exon = m_exon, fp.n_tot = n_tot;
	fp.hits_arena = hits_arena, fp.exons_arena = exons_arena;
	if (oom) {
		fp.failed = true;
		if (pg_verbose >= 1) std::fprintf(stderr, "[E::%s] out of memory while reading '%s'\n", __func__, fn ? fn : "-");
	}
}

static std::shared_mutex g_dict_mu; // writers of pg_data_t's growing arrays (batch and single-file reads)

static void snap_refresh(const pg_data_t *d, std::shared_ptr<const DictSnap> &slot, int64_t log_pos)
{
	const NameDict *dg = (const NameDict *)d->d_gene, *dp = (const NameDict *)d->d_prot;
	auto s = std::make_shared<DictSnap>();
	const size_t ng = (size_t)dg->size(), np = (size_t)dp->size();
	s->genes.reserve(ng), s->prots.reserve(np);
	s->g_pref.resize(ng), s->g_incl.resize(ng), s->g_len.resize(ng), s->p_gid.resize(np), s->p_len.resize(np);
	for (int32_t i = 0; i < dg->size(); ++i) s->genes.add(dg->name(i), dg->view(i).size(), FlatIndex::hash(dg->view(i))), s->g_pref[(size_t)i] = d->gene[i].preferred, s->g_incl[(size_t)i] = d->gene[i].included, s->g_len[(size_t)i] = d->gene[i].len;
	for (int32_t i = 0; i < dp->size(); ++i) s->prots.add(dp->name(i), dp->view(i).size(), FlatIndex::hash(dp->view(i))), s->p_gid[(size_t)i] = d->prot[i].gid, s->p_len[(size_t)i] = d->prot[i].len;
	s->log_pos = log_pos;
	std::atomic_store(&slot, std::shared_ptr<const DictSnap>(s));
}

static void preresolve(const std::shared_ptr<const DictSnap> &slot, FileParse &fp)
{
	fp.todo_g.clear(), fp.todo_p.clear(), fp.snap_log = -1;
	const std::shared_ptr<const DictSnap> s = std::atomic_load(&slot); // the newest one (the file was parsed against fp.snap, the same or an older one: ids never change)
	if (!s) return;
	for (int32_t i = 0; i < fp.genes.size(); ++i) {
		int32_t &g = fp.genes.global[(size_t)i];
		if (g < 0) g = s->genes.find(fp.genes.view(i));
		// what commit_ids would do to this gene changes nothing: preferred / included are assigned (read.c:158-159), len only grows (read.c:177)
		if (g < 0 || s->g_pref[(size_t)g] != fp.g_pref[(size_t)i] || s->g_incl[(size_t)g] != fp.g_incl[(size_t)i] || (int32_t)s->g_len[(size_t)g] < fp.g_len[(size_t)i]) fp.todo_g.push_back(i);
	}
	for (int32_t i = 0; i < fp.prots.size(); ++i) {
		int32_t &q = fp.prots.global[(size_t)i];
		if (q < 0) q = s->prots.find(fp.prots.view(i));
		const int32_t g = fp.genes.global[(size_t)fp.p_gene[(size_t)i]];
		if (q < 0 || g < 0 || s->p_gid[(size_t)q] != g || s->p_len[(size_t)q] != fp.p_len[(size_t)i]) fp.todo_p.push_back(i);
	}
	fp.snap_log = s->log_pos;
}

// sequential part: global ids in first-seen order (the numbering of per-line dict_put calls, read.c:151-168), genome appended to `d`
static int32_t commit_ids(pg_data_t *d, FileParse &fp, ChangeLog *log = nullptr)
{
	if (!fp.opened) return -1;
	if (fp.failed) { // a truncated genome would build a wrong graph with exit status 0: the read fails and pg_post_process refuses to run
		set_error(PGA_ERR_NOMEM, "pg_read_paf: out of memory");
		ext_of(d, true)->read_failed = true;
		return -1;
	}
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
	// genes then proteins, each in the order this file saw them first (an id's attributes depend on this file's entry for THAT id
	// only, so any subset of the entries can be applied on its own)
	auto one_gene = [&](int32_t i) {
		int32_t gid = fp.genes.global[(size_t)i];
		bool fresh = false;
		if (gid < 0) {
			bool absent;
			gid = dg->put(fp.genes.name(i), &absent);
			if (absent) { d->n_gene++; grow0(d->gene, gid, d->m_gene); fresh = true; }
			d->gene[gid].name = dg->name(gid);
			fp.genes.global[(size_t)i] = gid;
		}
		const bool longer = (int32_t)d->gene[gid].len < fp.g_len[(size_t)i];
		if (log && !fresh && (d->gene[gid].preferred != fp.g_pref[(size_t)i] || d->gene[gid].included != fp.g_incl[(size_t)i] || longer)) log->v.emplace_back((uint8_t)0, gid);
		d->gene[gid].preferred = fp.g_pref[(size_t)i], d->gene[gid].included = fp.g_incl[(size_t)i];
		if (longer) d->gene[gid].len = (uint32_t)fp.g_len[(size_t)i];
	};
	auto one_prot = [&](int32_t i) {
		int32_t pid = fp.prots.global[(size_t)i];
		bool fresh = false;
		if (pid < 0) {
			bool absent;
			pid = dp->put(fp.prots.name(i), &absent);
			if (absent) { d->n_prot++; grow0(d->prot, pid, d->m_prot); fresh = true; }
			d->prot[pid].name = dp->name(pid);
			fp.prots.global[(size_t)i] = pid;
		}
		const int32_t gid = fp.genes.global[(size_t)fp.p_gene[(size_t)i]];
		if (log && !fresh && (d->prot[pid].gid != gid || d->prot[pid].len != fp.p_len[(size_t)i])) log->v.emplace_back((uint8_t)1, pid);
		d->prot[pid].gid = gid;
		d->prot[pid].len = fp.p_len[(size_t)i];
	};
	const int64_t behind = (log && fp.snap_log >= 0) ? (int64_t)log->v.size() - fp.snap_log : -1;
	if (behind >= 0 && behind <= 256) {
		// the entries the snapshot did not settle, and this file's entries for the ids somebody changed since the snapshot
		const int64_t l0 = fp.snap_log, l1 = (int64_t)log->v.size(); // (the entries applied here may append to the log: not looked at again)
		for (int32_t i : fp.todo_g) one_gene(i);
		for (int64_t k = l0; k < l1; ++k) if (log->v[(size_t)k].first == 0) { const int32_t li = fp.genes.local_of(log->v[(size_t)k].second, dg->view(log->v[(size_t)k].second)); if (li >= 0) one_gene(li); }
		for (int32_t i : fp.todo_p) one_prot(i);
		for (int64_t k = l0; k < l1; ++k) {
			const std::pair<uint8_t, int32_t> c = log->v[(size_t)k];
			if (c.first == 1) { const int32_t li = fp.prots.local_of(c.second, dp->view(c.second)); if (li >= 0) one_prot(li); }
			else { // a gene whose id this file's proteins point at did not move (ids never change): nothing to do for them
			}
		}
	} else {
		for (int32_t i = 0; i < fp.genes.size(); ++i) one_gene(i);
		for (int32_t i = 0; i < fp.prots.size(); ++i) one_prot(i);
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
	for (int32_t i = 0; i < fp.n_hit; ++i) fp.hits[i].pid = fp.prots.global[(size_t)fp.hits[i].pid];
	// (an estimate that was far too generous -- a .gz that compressed badly -- is given back: realloc in place, no copy)
	if (fp.hits && !fp.hits_arena && (size_t)fp.m_hit > (size_t)fp.n_hit * 2 + 4096) { pg_hit_t *t = (pg_hit_t *)std::realloc((void *)fp.hits, sizeof(pg_hit_t) * (size_t)(fp.n_hit + 1)); if (t) fp.hits = t, fp.m_hit = fp.n_hit + 1; }
	if (fp.exons && !fp.exons_arena && (size_t)fp.m_exon > (size_t)fp.n_exon * 2 + 4096) { pg_exon_t *t = (pg_exon_t *)std::realloc((void *)fp.exons, sizeof(pg_exon_t) * (size_t)(fp.n_exon + 1)); if (t) fp.exons = t, fp.m_exon = fp.n_exon + 1; }
	g->n_hit = fp.n_hit, g->m_hit = fp.m_hit > 0 ? fp.m_hit : 1;
	g->hit = fp.hits ? fp.hits : (pg_hit_t *)std::malloc(sizeof(pg_hit_t));
	g->n_exon = fp.n_exon, g->m_exon = fp.m_exon > 0 ? fp.m_exon : 1;
	g->exon = fp.exons ? fp.exons : (pg_exon_t *)std::malloc(sizeof(pg_exon_t));
	fp.hits = nullptr, fp.exons = nullptr, fp.n_hit = fp.m_hit = fp.n_exon = fp.m_exon = 0, fp.hits_arena = fp.exons_arena = false;
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
	const DataExt *ext = ext_of(d, false);
	for (int32_t i = 0; i < d->n_genome; ++i) {
		pg_genome_t *g = &d->genome[i];
		std::free(g->ctg); std::free(g->label);
		if (!(ext && ext->arena_owns(g->hit))) std::free(g->hit);   // (arrays inside a batch read's arena go with the arena: ext_drop)
		if (!(ext && ext->arena_owns(g->exon))) std::free(g->exon);
	}
	ext_drop(d);
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
static double g_reserve_sec = 0.0; // what the last batch read spent reserving device memory before it parsed (pg_last_reserve_seconds)
double pg_last_reserve_seconds(void) { return g_reserve_sec; }

int32_t pg_read_paf_batch(const pg_opt_t *opt, pg_data_t *d, int32_t n, const char *const *fns, const uint8_t *ids_only, int32_t n_threads)
{
	if (n <= 0) return 0;
	g_reserve_sec = 0.0;
	if (n_threads <= 0) {
		const char *e = std::getenv("PANGENE_READ_THREADS");
		n_threads = e && std::atoi(e) > 0 ? std::atoi(e) : (int32_t)host_threads(64u); // (the CPU budget of the process, see host_threads; beyond 64 the threads of one address space get in each other's way: page faults, allocator)
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
	size_t local_text = 0; int32_t n_local = 0; // (what pga_reserve's estimate is made of)
	for (int32_t i = 0; i < n; ++i) {
		struct stat sb;
		const size_t len = fns[i] ? std::strlen(fns[i]) : 0;
		if (!(ids_only && ids_only[i]) && fns[i] && stat(fns[i], &sb) == 0) local_text += (size_t)sb.st_size * (len > 3 && std::strcmp(fns[i] + len - 3, ".gz") == 0 ? 5 : 1), ++n_local;
	}
	// A large data set has the device memory its upload will ask for allocated NOW, before a parser thread runs: hipMalloc of configs[3]'s
	// ~90 GB takes 1.9 s, and while it runs every page fault of the process waits (tried: on a helper thread beside the parsers the
	// parse took 3.3 s instead of 1.3 and the packing 2 s instead of 0.1 -- the time moved, it did not go away).  The estimate comes from
	// the files' sizes and the head of the first local plain file (lines per byte, introns per line); a wrong one costs the attempt.
	if (local_text >= ((size_t)1 << 30)) {
		const pga_backend_t *be = backend_default();
		for (int32_t i = 0; be && be->reserve && i < n; ++i) {
			const size_t len = fns[i] ? std::strlen(fns[i]) : 0;
			if ((ids_only && ids_only[i]) || !fns[i] || (len > 3 && std::strcmp(fns[i] + len - 3, ".gz") == 0)) continue;
			const int fd = open(fns[i], O_RDONLY);
			if (fd < 0) continue;
			std::vector<char> head((size_t)256 << 10);
			const ssize_t got = read(fd, head.data(), head.size());
			close(fd);
			if (got <= 0) break;
			int64_t lines = 0, introns = 0;
			bool in_cg = false;
			for (ssize_t q = 0; q < got; ++q) {
				const char ch = head[(size_t)q];
				if (ch == '\n') ++lines, in_cg = false;
				else if (ch == '\t') in_cg = q + 5 < got && std::memcmp(&head[(size_t)q + 1], "cg:Z:", 5) == 0;
				else if (in_cg && (ch == 'N' || ch == 'U' || ch == 'V')) ++introns;
			}
			if (lines < 8) break;
			const double per_byte = (double)lines / (double)got, ex_per_hit = 1.0 + (double)introns / (double)lines;
			const int64_t hits = (int64_t)((double)local_text * per_byte * 1.05) + 4096, exons = (int64_t)((double)hits * ex_per_hit * 1.05) + 4096;
			struct stat sb;
			const int64_t first_lines = stat(fns[i], &sb) == 0 ? (int64_t)((double)sb.st_size * per_byte) + 64 : lines;
			const double tr0 = now_sec();
			if (hits < ((int64_t)1 << 30) && exons < INT32_MAX)
				(void)be->reserve(hits, exons, (int32_t)std::min<int64_t>(first_lines, 1 << 24), (int32_t)std::min<int64_t>(first_lines, (1 << 20) - 1), n_local, hits * 11 + hits / 4 + exons * 2 + 64 * (int64_t)n_local);
			g_reserve_sec = now_sec() - tr0;
			break;
		}
	}
	std::vector<FileParse> fp((size_t)n);
	{ // the hit / exon arrays of the plain files come out of one mapping on huge pages (DataExt::HostArena says why); reserved for the
	  // worst case -- a PAF line of miniprot has >= 60 bytes, a hit record 88 -- and only touched where the parsers write
		size_t plain = 0;
		for (int32_t i = 0; i < n; ++i) {
			struct stat sb;
			const size_t len = fns[i] ? std::strlen(fns[i]) : 0;
			if (!(ids_only && ids_only[i]) && fns[i] && !(len > 3 && std::strcmp(fns[i] + len - 3, ".gz") == 0) && stat(fns[i], &sb) == 0 && S_ISREG(sb.st_mode)) plain += (size_t)sb.st_size;
		}
		static const bool no_arena = std::getenv("PANGENE_NO_ARENA") != nullptr;
		if (plain >= ((size_t)8 << 20) && !no_arena) {
			const size_t huge = (size_t)2 << 20, want = ((plain / 60 + (size_t)n) * sizeof(pg_hit_t) + (plain / 3 + 2048 * (size_t)n) * sizeof(pg_exon_t) + 128 * (size_t)n + huge - 1) & ~(huge - 1);
			void *m = mmap(nullptr, want + huge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
			if (m != MAP_FAILED) {
				ext->arenas.emplace_back();
				DataExt::HostArena &a = ext->arenas.back();
				a.map = (char *)m, a.map_bytes = want + huge;
				a.base = (char *)(((uintptr_t)m + huge - 1) & ~(uintptr_t)(huge - 1)), a.bytes = want;
				(void)madvise(a.base, a.bytes, MADV_HUGEPAGE);
				for (int32_t i = 0; i < n; ++i) fp[(size_t)i].arena = &a;
			}
		}
	}
	std::vector<std::atomic<uint8_t>> state((size_t)n); // 0 new, 1 parsed, 2 committed (ids final), 3 being / has been finished
	for (auto &x : state) x.store(0);
	std::atomic<int32_t> next_parse{0}, n_commit{0}, next_final{0}, n_fail{0};
	static const bool timing = std::getenv("PANGENE_TIMING") != nullptr;
	std::atomic<int64_t> us_parse{0}, us_resolve{0}, us_commit{0}, us_final{0};
	const double t_batch0 = now_sec();
	struct rusage ru0; getrusage(RUSAGE_SELF, &ru0);
	std::mutex commit_mu;
	std::shared_ptr<const DictSnap> snap; // published with atomic_store by whoever commits
	int32_t snap_names = 0;
	ChangeLog chg;                         // (commit lock)
	int64_t snap_chg = 0;
	std::atomic<int32_t> n_fast{0};
	auto try_commit = [&]() {
		std::unique_lock<std::mutex> lk(commit_mu, std::try_to_lock);
		if (!lk.owns_lock()) return; // somebody else is at it (and will see what this thread just parsed: it re-checks before it leaves)
		for (;;) {
			const int32_t k = n_commit.load();
			if (k >= n || state[(size_t)k].load() != 1) break;
			const double tc0 = timing ? now_sec() : 0.0;
			if (fp[(size_t)k].snap_log >= 0 && (int64_t)chg.v.size() - fp[(size_t)k].snap_log <= 256) n_fast.fetch_add(1);
			if (commit_ids(d, fp[(size_t)k], &chg) != 0) n_fail.fetch_add(1);
			// (a rebuild copies every name: geometrically spaced, so that files that all bring new names do not make the commits quadratic;
			// attribute changes: a new snapshot once 64 have come, or the files behind it would carry them along one by one)
			if (d->n_gene + d->n_prot >= snap_names + std::max(1000, snap_names / 4) || (k == 0 && d->n_gene + d->n_prot > 0) || (int64_t)chg.v.size() - snap_chg >= 64)
				snap_refresh(d, snap, (int64_t)chg.v.size()), snap_names = d->n_gene + d->n_prot, snap_chg = (int64_t)chg.v.size();
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
			{ FileParse done; std::swap(done.genes, f.genes), std::swap(done.prots, f.prots), std::swap(done.ctgs, f.ctgs), std::swap(done.snap, f.snap); } // the names are not needed any more
			if (timing) us_final.fetch_add((int64_t)((now_sec() - tf0) * 1e6));
			state[(size_t)k].store(3);
			return true;
		}
	};
	auto work = [&]() {
		for (;;) {
			const int32_t i = next_parse.fetch_add(1);
			if (i < n) {
				// the first wave of files would be parsed before the first commit has published a snapshot -- every name of theirs hashed into
				// dictionaries of their own and left to the sequential part: better wait for it (the time one file takes; asleep, not spinning)
				for (int spin = 0; i > 0 && spin < 400 && !std::atomic_load(&snap) && n_commit.load() < 1; ++spin) std::this_thread::sleep_for(std::chrono::microseconds(50));
				const double tp0 = timing ? now_sec() : 0.0;
				parse_file(opt, fns[i], ids_only && ids_only[i], fp[(size_t)i], std::atomic_load(&snap));
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
	if (timing) std::fprintf(stderr, "[pg_read_paf_batch] %d files on %d threads: %.1f ms wall (+ %.1f ms for the slab helper); summed over the threads: parsing %.1f ms, name look-ups %.1f ms, finishing (ids in place + SoA block) %.1f ms; the sequential commits %.1f ms (%d of them only the file's own news: %lld attribute change(s) logged); %.2f M minor page faults\n",
	                         n, n_threads, (t_joined - t_batch0) * 1e3, (now_sec() - t_joined) * 1e3, us_parse.load() * 1e-3, us_resolve.load() * 1e-3, us_final.load() * 1e-3, us_commit.load() * 1e-3, n_fast.load(), (long long)chg.v.size(), [&] { struct rusage r; getrusage(RUSAGE_SELF, &r); return (double)(r.ru_minflt - ru0.ru_minflt) * 1e-6; }());
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
