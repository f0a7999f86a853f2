// gfa_writer.cpp -- BED / GFA S-,L-,W-line emission (the reference's format.c:78-225).  Host-side;
// bytes must equal the reference's, so integer formatting follows its pg_sprintf_lite (format.c:24-76:
// "%ld" arguments are narrowed to int before printing) and the id:f tag its "%.4f".
#include <zlib.h>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <regex>
#include <atomic>
#include <string>
#include <thread>
#include <unordered_map>
#include "pg_internal.hpp"

namespace pgx {

static FILE *g_out = nullptr;
FILE *out_stream() { return g_out ? g_out : stdout; }

static inline void put_i32(std::string &s, int32_t c)
{
	char buf[16];
	int l = 0;
	uint32_t x = c >= 0 ? (uint32_t)c : (uint32_t)(-(int64_t)c);
	do { buf[l++] = (char)('0' + x % 10); x /= 10; } while (x > 0);
	if (c < 0) buf[l++] = '-';
	while (l > 0) s.push_back(buf[--l]);
}
static inline void put_long(std::string &s, int64_t v) { put_i32(s, (int32_t)v); } // format.c:44-46 narrows to int

static void bed_line(std::string &o, const pg_data_t *d, const pg_genome_t *g, const pg_hit_t *a)
{
	o += g->ctg[a->cid].name; o += '\t'; put_long(o, a->cs); o += '\t'; put_long(o, a->ce); o += '\t';
	o += d->prot[a->pid].name; o += '\t'; put_i32(o, a->score_ori); o += '\t'; o += "+-"[a->rev]; o += '\t';
	put_long(o, a->cs); o += '\t'; put_long(o, a->ce); o += "\t0\t"; put_i32(o, a->n_exon); o += '\t';
	for (int32_t i = 0; i < a->n_exon; ++i) { put_i32(o, g->exon[a->off_exon + i].oe - g->exon[a->off_exon + i].os); o += ','; }
	o += '\t';
	for (int32_t i = 0; i < a->n_exon; ++i) { put_i32(o, g->exon[a->off_exon + i].os); o += ','; }
	char idbuf[16];
	std::snprintf(idbuf, 15, "%.4f", (double)a->mlen / a->blen);
	o += "\tft:i:"; put_i32(o, a->flt);
	o += "\tpf:Z:"; put_i32(o, a->pseudo); put_i32(o, a->flt_iso_ov); put_i32(o, a->flt_chain); put_i32(o, a->flt_iso_sub_self);
	o += "\trk:i:"; put_i32(o, a->rank);
	o += "\trp:i:"; put_i32(o, a->rep);
	o += "\tsd:i:"; put_i32(o, a->shadow);
	o += "\tvt:i:"; put_i32(o, a->vtx);
	o += "\tbr:i:"; put_i32(o, a->weak_br);
	o += "\tcm:i:"; put_long(o, a->cm);
	o += "\tid:f:"; o += idbuf;
	o += "\tdm:Z:"; o += a->pid_dom0 < 0 ? "*" : d->prot[a->pid_dom0].name;
	o += '\n';
}

// "sample#hap#ctg" -> sample, hap (format.c:159-181): the decision is taken at the second field
static int32_t parse_sample(std::string &sample, const char *name)
{
	const char *h1 = std::strchr(name, '#');
	size_t l0 = h1 ? (size_t)(h1 - name) : std::strlen(name);
	if (l0 == 0) return -1;
	sample.assign(name, l0);
	if (h1 == nullptr) return -1;
	const char *q = h1 + 1;
	const char *h2 = std::strchr(q, '#');
	const char *p = h2 ? h2 : q + std::strlen(q);
	char *r;
	long hap = std::strtol(q, &r, 10);
	return (r == p && hap >= 0) ? (int32_t)hap : -1;
}

// every line of a plain or gzipped text file (without the line terminators)
static int read_lines(const char *fn, std::vector<std::string> &out)
{
	gzFile fp = (fn && std::strcmp(fn, "-") != 0) ? gzopen(fn, "r") : gzdopen(0, "r");
	if (fp == nullptr) return -1;
	std::string cur;
	char buf[1 << 16];
	int n;
	while ((n = gzread(fp, buf, sizeof(buf))) > 0)
		for (int i = 0; i < n; ++i) {
			if (buf[i] == '\n') { if (!cur.empty() && cur.back() == '\r') cur.pop_back(); out.push_back(cur); cur.clear(); }
			else cur.push_back(buf[i]);
		}
	if (!cur.empty()) out.push_back(cur);
	gzclose(fp);
	return 0;
}

} // namespace pgx

using namespace pgx;

extern "C" {

int pg_set_output(const char *path)
{
	if (g_out) { std::fclose(g_out); g_out = nullptr; }
	if (path == nullptr) return 0;
	g_out = std::fopen(path, "ab");
	return g_out ? 0 : -1;
}

void pg_write_bed(const pg_data_t *d, int32_t is_walk) // format.c:96-118
{
	if (sync_host(const_cast<pg_data_t *>(d), true) != 0) return;
	FILE *fp = out_stream();
	std::string o;
	for (int32_t j = 0; j < d->n_genome; ++j) {
		const pg_genome_t *g = &d->genome[j];
		for (int32_t i = 0; i < g->n_hit; ++i) {
			const pg_hit_t *a = &g->hit[i];
			if (is_walk && a->flt) continue;
			o.clear();
			bed_line(o, d, g, a);
			std::fwrite(o.data(), 1, o.size(), fp);
		}
	}
	std::fflush(fp);
}

void pg_write_graph(const pg_graph_t *q) // format.c:120-157
{
	const pg_data_t *d = q->d;
	FILE *fp = out_stream();
	std::string o;
	for (int32_t i = 0; i < q->n_seg; ++i) {
		const pg_seg_t *s = &q->seg[i];
		int32_t pid = d->gene[s->gid].rep_pid;
		o.clear();
		o += "S\t"; o += d->gene[s->gid].name; o += "\t*\tLN:i:"; put_i32(o, d->prot[pid].len);
		o += "\tng:i:"; put_i32(o, s->n_genome); o += "\tnc:i:"; put_i32(o, s->tot_cnt);
		o += "\tc1:i:"; put_i32(o, s->n_dom); o += "\tc2:i:"; put_i32(o, s->n_sub);
		o += "\tpp:Z:"; o += d->prot[pid].name; o += '\n';
		std::fwrite(o.data(), 1, o.size(), fp);
	}
	for (int32_t i = 0; i < q->n_arc; ++i) {
		const pg_arc_t *a = &q->arc[i];
		uint32_t v = (uint32_t)(a->x >> 32), w = (uint32_t)a->x;
		o.clear();
		o += "L\t"; o += d->gene[q->seg[v >> 1].gid].name; o += '\t'; o += "+-"[v & 1]; o += '\t';
		o += d->gene[q->seg[w >> 1].gid].name; o += '\t'; o += "+-"[w & 1]; o += "\t0M\tng:i:"; put_i32(o, a->n_genome);
		o += "\tnc:i:"; put_i32(o, a->tot_cnt); o += "\tad:i:"; put_i32(o, a->avg_dist);
		o += "\ts1:i:"; put_i32(o, a->s1); o += "\ts2:i:"; put_i32(o, a->s2); o += '\n';
		std::fwrite(o.data(), 1, o.size(), fp);
	}
	std::fflush(fp);
}

// W-lines (format.c:183-225): per genome, contigs in id order, surviving hits in cm order.  The host records stay where they
// are (file order, or cs order after a full sync); the cm order comes from the backend's Y permutation as file indices
// (DataExt::y_file) and flt from the bit vector of the last sync (indexed by X position).
// Genomes are independent: host threads format them side by side into buffers of their own (names copied with their known
// lengths, no per-line allocation), and the buffers go to the stream in genome order, a window of genomes at a time.
static inline char *cat_i32(char *p, int32_t c)
{
	char buf[16];
	int l = 0;
	uint32_t x = c >= 0 ? (uint32_t)c : (uint32_t)(-(int64_t)c);
	do { buf[l++] = (char)('0' + x % 10); x /= 10; } while (x > 0);
	if (c < 0) buf[l++] = '-';
	while (l > 0) *p++ = buf[--l];
	return p;
}

void pg_write_walk(pg_graph_t *q)
{
	pg_data_t *d = q->d;
	if (sync_host(d, false) != 0) return;
	DataExt *ext = ext_of(d, false);
	if (ext == nullptr || ext->ctx == nullptr) { set_error(PGA_ERR_ARG, "pg_write_walk: pg_graph_gen has not run on this data set"); return; }
	FILE *fp = out_stream();
	std::vector<int64_t> goff_of((size_t)d->n_genome, -1);
	for (size_t k = 0; k < ext->local_genomes.size(); ++k) goff_of[(size_t)ext->local_genomes[k]] = ext->hit_off[k];
	for (int32_t j = 0; j < d->n_genome; ++j) {
		const pg_genome_t *g = &d->genome[j];
		if (g->n_hit == 0) continue;
		if (goff_of[(size_t)j] < 0 || (size_t)j >= ext->y_file.size() || ext->y_file[(size_t)j].size() != (size_t)g->n_hit) { set_error(PGA_ERR_ARG, "pg_write_walk: genome without backend state"); return; }
	}
	std::vector<uint32_t> gene_len((size_t)d->n_gene);
	for (int32_t i = 0; i < d->n_gene; ++i) gene_len[(size_t)i] = (uint32_t)std::strlen(d->gene[i].name);
	const uint64_t *fb = ext->flt_bits.data();
	auto format_genome = [&](int32_t j, std::vector<char> &out) {
		out.clear();
		const pg_genome_t *g = &d->genome[j];
		if (g->n_hit == 0) return;
		const int64_t goff = goff_of[(size_t)j];
		const int32_t *yf = ext->y_file[(size_t)j].data();
		const int32_t *hof = ext->hits_sorted[(size_t)j] ? ext->host_of_file[(size_t)j].data() : nullptr;
		const int32_t *px = ext->pos_x.data() + goff;
		auto hit_of = [&](int32_t k) -> const pg_hit_t * { const int32_t f = yf[k]; return &g->hit[hof ? hof[f] : f]; };
		auto is_flt = [&](int32_t k) { const int64_t b = goff + px[yf[k]]; return (fb[b >> 6] >> (b & 63) & 1) != 0; };
		std::string sample;
		for (int32_t i0 = 0, i = 1; i <= g->n_hit; ++i) {
			if (i != g->n_hit && hit_of(i)->cid == hit_of(i0)->cid) continue;
			const int32_t cid = hit_of(i0)->cid;
			// the surviving hits of the contig, and the room their two lists take
			size_t need = 0;
			int32_t n = 0;
			for (int32_t k = i0; k < i; ++k) {
				if (is_flt(k)) continue;
				need += 1 + gene_len[(size_t)d->prot[hit_of(k)->pid].gid] + 12;
				++n;
			}
			if (n > 0) {
				const char *cname = g->ctg[cid].name;
				const size_t lc = std::strlen(cname), ll = g->label ? std::strlen(g->label) : 0;
				const int32_t hap = parse_sample(sample, cname);
				const size_t at = out.size();
				out.resize(at + need + lc + ll + sample.size() + 64);
				char *p = out.data() + at;
				*p++ = 'W', *p++ = '\t';
				if (hap >= 0) { std::memcpy(p, sample.data(), sample.size()), p += sample.size(); *p++ = '\t'; p = cat_i32(p, hap); }
				else if (g->label) { std::memcpy(p, g->label, ll), p += ll; *p++ = '\t', *p++ = '0'; }
				else { p = cat_i32(p, j); *p++ = '\t', *p++ = '0'; }
				*p++ = '\t';
				std::memcpy(p, cname, lc), p += lc;
				std::memcpy(p, "\t*\t*\t", 5), p += 5;
				for (int32_t k = i0; k < i; ++k) {
					if (is_flt(k)) continue;
					const pg_hit_t *a = hit_of(k);
					const int32_t gid = d->prot[a->pid].gid;
					*p++ = "><"[a->rev];
					std::memcpy(p, d->gene[gid].name, gene_len[(size_t)gid]), p += gene_len[(size_t)gid];
				}
				std::memcpy(p, "\tlf:B:i", 7), p += 7;
				for (int32_t k = i0; k < i; ++k) {
					if (is_flt(k)) continue;
					*p++ = ',';
					p = cat_i32(p, hit_of(k)->lof);
				}
				*p++ = '\n';
				out.resize((size_t)(p - out.data()));
			}
			i0 = i;
		}
	};
	unsigned nt = ext->n_hit_local > 200000 ? host_threads(32u) : 1u;
	if (nt > (unsigned)d->n_genome) nt = (unsigned)std::max(1, d->n_genome);
	if (nt <= 1) {
		std::vector<char> buf;
		for (int32_t j = 0; j < d->n_genome; ++j) { format_genome(j, buf); if (!buf.empty()) std::fwrite(buf.data(), 1, buf.size(), fp); }
	} else {
		const int32_t window = (int32_t)nt * 8; // genomes formatted side by side before their bytes are written (bounds the memory)
		std::vector<std::vector<char>> bufs((size_t)window);
		for (int32_t j0 = 0; j0 < d->n_genome; j0 += window) {
			const int32_t j1 = std::min(d->n_genome, j0 + window);
			std::atomic<int32_t> next{j0};
			std::vector<std::thread> th;
			for (unsigned t = 0; t < nt; ++t)
				th.emplace_back([&]() { for (;;) { const int32_t j = next.fetch_add(1); if (j >= j1) break; format_genome(j, bufs[(size_t)(j - j0)]); } });
			for (auto &x : th) x.join();
			for (int32_t j = j0; j < j1; ++j) if (!bufs[(size_t)(j - j0)].empty()) std::fwrite(bufs[(size_t)(j - j0)].data(), 1, bufs[(size_t)(j - j0)].size(), fp);
		}
	}
	std::fflush(fp);
}

// pangene.js gfa2matrix (pangene.js:1168-1247) straight from the graph in memory: rows = segments in S-line order, columns =
// sample#haplotype in the order the W-lines would introduce them, entry = presence (or, copy_number != 0, the number of
// occurrences) of the segment in the walks of that assembly.  The per-hit reduction runs on the backend (pga_gene_matrix).
void pg_write_matrix(pg_graph_t *q, int32_t copy_number)
{
	pg_data_t *d = q->d;
	DataExt *ext = ext_of(d, false);
	if (ext == nullptr || ext->ctx == nullptr) { set_error(PGA_ERR_ARG, "pg_write_matrix: pg_graph_gen has not run on this data set"); return; }
	// the backend counts per contig AS IT SEES THEM (pieces of virtual contigs, pga_genome_block_t): v_of[] = its numbering
	size_t n_ctg = 0, n_v = 0;
	for (size_t k = 0; k < ext->local_genomes.size(); ++k) n_ctg += (size_t)d->genome[ext->local_genomes[k]].n_ctg, n_v += (size_t)(k < ext->n_vctg.size() ? ext->n_vctg[k] : d->genome[ext->local_genomes[k]].n_ctg);
	std::vector<int32_t> vcnt(n_v + 1, 0), cnt(n_ctg + 1, 0), col(n_ctg + 1, -1), real_of(n_v + 1, 0);
	int rc = ext->be->ctg_counts(ext->ctx, vcnt.data());
	if (rc != 0) { set_error(rc, "ctg_counts"); return; }
	{
		size_t rb = 0, vb = 0;
		for (size_t kk = 0; kk < ext->local_genomes.size(); ++kk) {
			const int32_t nr = d->genome[ext->local_genomes[kk]].n_ctg, nv = kk < ext->n_vctg.size() ? ext->n_vctg[kk] : nr;
			const std::vector<int32_t> *vr = (kk < ext->vreal.size() && !ext->vreal[kk].empty()) ? &ext->vreal[kk] : nullptr;
			for (int32_t v = 0; v < nv; ++v) { const size_t r = rb + (size_t)(vr ? (*vr)[(size_t)v] : v); real_of[vb + (size_t)v] = (int32_t)r, cnt[r] += vcnt[vb + (size_t)v]; }
			rb += (size_t)nr, vb += (size_t)nv;
		}
	}
	std::vector<std::string> names;
	std::unordered_map<std::string, int32_t> idx;
	std::string sample, key;
	size_t k = 0;
	for (int32_t j : ext->local_genomes) { // W-lines: genomes in input order, contigs in id order, only contigs with a surviving hit (format.c:183-225)
		const pg_genome_t *g = &d->genome[j];
		for (int32_t c = 0; c < g->n_ctg; ++c, ++k) {
			if (cnt[k] == 0) continue;
			const int32_t hap = parse_sample(sample, g->ctg[c].name);
			char num[16];
			if (hap >= 0) std::snprintf(num, sizeof(num), "%d", hap), key = sample + "#" + num;
			else if (g->label) key = std::string(g->label) + "#0";
			else std::snprintf(num, sizeof(num), "%d", j), key = std::string(num) + "#0";
			auto it = idx.find(key);
			if (it == idx.end()) it = idx.emplace(key, (int32_t)names.size()).first, names.push_back(key);
			col[k] = it->second;
		}
	}
	const int32_t n_asm = (int32_t)names.size();
	std::vector<int32_t> mat((size_t)q->n_seg * (size_t)n_asm + 1, 0);
	std::vector<int32_t> vcol(n_v + 1, -1); // column of every contig as the backend numbers them
	for (size_t v = 0; v < n_v; ++v) vcol[v] = col[(size_t)real_of[v]];
	rc = ext->be->gene_matrix(ext->ctx, vcol.data(), n_asm, q->n_seg, mat.data());
	if (rc != 0) { set_error(rc, "gene_matrix"); return; }
	FILE *fp = out_stream();
	std::string o = "Gene\t";
	for (int32_t a = 0; a < n_asm; ++a) { if (a) o += '\t'; o += names[(size_t)a]; }
	o += '\n';
	std::fwrite(o.data(), 1, o.size(), fp);
	for (int32_t i = 0; i < q->n_seg; ++i) {
		o.assign(d->gene[q->seg[i].gid].name); o += '\t';
		for (int32_t a = 0; a < n_asm; ++a) {
			int32_t v = mat[(size_t)i * (size_t)n_asm + (size_t)a];
			if (!copy_number && v > 1) v = 1;
			if (a) o += '\t';
			put_i32(o, v);
		}
		o += '\n';
		std::fwrite(o.data(), 1, o.size(), fp);
	}
	std::fflush(fp);
}

// The same command on a GFA file, as pangene.js runs it (pangene.js:1168-1247 with the parser at 131-197): segments in the order
// S- and L-lines introduce them, walk steps whose name is not a segment yet are ignored, assemblies = "sample#hap" of the W-lines
// in first-seen order.  clstr_fn (may be NULL): CD-HIT cluster file; the members of a cluster are added to its representative
// ("*") and not printed themselves; print_cd prints the (member, representative) pairs instead of the matrix.  Plain or gzipped
// input.  Returns 0, or -1 when a file cannot be opened.
int pg_gfa2matrix_file(const char *gfa_fn, int32_t copy_number, const char *clstr_fn, int32_t print_cd)
{
	std::vector<std::string> lines;
	if (read_lines(gfa_fn, lines) != 0) return -1;
	std::vector<std::string> seg, asm_a;
	std::unordered_map<std::string, int32_t> seg_h, asm_h;
	std::vector<std::pair<int32_t, int32_t>> walk; // (assembly, segment) of every walk step
	auto seg_add = [&](const std::string &n) { auto it = seg_h.find(n); if (it == seg_h.end()) it = seg_h.emplace(n, (int32_t)seg.size()).first, seg.push_back(n); return it->second; };
	auto split = [](const std::string &l, std::vector<std::string> &t) { t.clear(); size_t b = 0; for (;;) { size_t e = l.find('\t', b); t.push_back(l.substr(b, e == std::string::npos ? e : e - b)); if (e == std::string::npos) break; b = e + 1; } };
	std::vector<std::string> t;
	for (const std::string &l : lines) {
		if (l.empty()) continue;
		if (l[0] == 'S') { split(l, t); if (t.size() >= 3) seg_add(t[1]); }
		else if (l[0] == 'L') { split(l, t); if (t.size() >= 5 && (t[2] == "+" || t[2] == "-") && (t[4] == "+" || t[4] == "-")) seg_add(t[1]), seg_add(t[3]); }
		else if (l[0] == 'W') {
			split(l, t);
			if (t.size() < 7) continue;
			const std::string a = t[1] + "#" + t[2];
			auto it = asm_h.find(a);
			if (it == asm_h.end()) it = asm_h.emplace(a, (int32_t)asm_a.size()).first, asm_a.push_back(a);
			const std::string &w = t[6];
			for (size_t i = 0; i < w.size();) { // ([><])([^\s><]+)
				if (w[i] != '>' && w[i] != '<') { ++i; continue; }
				size_t e = i + 1;
				while (e < w.size() && w[e] != '>' && w[e] != '<' && !std::isspace((unsigned char)w[e])) ++e;
				if (e > i + 1) { auto sit = seg_h.find(w.substr(i + 1, e - i - 1)); if (sit != seg_h.end()) walk.emplace_back(it->second, sit->second); }
				i = e;
			}
		}
	}
	const size_t n_asm = asm_a.size();
	std::vector<int32_t> mat(seg.size() * n_asm + 1, 0);
	for (const auto &st : walk) ++mat[(size_t)st.second * n_asm + (size_t)st.first];
	std::unordered_map<std::string, std::string> paralog;
	std::vector<std::string> paralog_order;
	FILE *fp = out_stream();
	if (clstr_fn) {
		std::vector<std::string> cl;
		if (read_lines(clstr_fn, cl) != 0) return -1;
		std::vector<std::pair<std::string, bool>> b;
		auto gene_of = [](const std::string &x) { return x.substr(0, x.find(':')); };
		auto process = [&]() {
			int sel = -1;
			for (size_t i = 0; i < b.size(); ++i) if (b[i].second) sel = (int)i;
			if (sel >= 0)
				for (size_t i = 0; i < b.size(); ++i) {
					if ((int)i == sel) continue;
					const std::string g = gene_of(b[i].first), p = gene_of(b[(size_t)sel].first);
					if (paralog.find(g) == paralog.end()) paralog_order.push_back(g);
					paralog[g] = p;
					if (print_cd) { const std::string o = g + "\t" + p + "\n"; std::fwrite(o.data(), 1, o.size(), fp); }
				}
			b.clear();
		};
		static const std::regex re_member("^\\d+\\s+\\S+,\\s+>(\\S+)\\.\\.\\.\\s+(\\S+)"); // the script's own pattern (pangene.js:1223), ECMAScript semantics
		for (const std::string &l : cl) {
			if (!l.empty() && l[0] == '>') { process(); continue; }
			std::smatch m;
			if (std::regex_search(l, m, re_member)) b.emplace_back(m[1].str(), m[2].str() == "*");
		}
		process();
		// `for (const g in paralog)`: JavaScript visits integer-like keys first, in ascending numeric order, then the others in
		// insertion order -- and the order matters when a paralog's target is itself a paralog
		std::vector<std::string> order;
		{
			std::vector<std::pair<uint64_t, std::string>> ints;
			for (const std::string &g : paralog_order) {
				bool idx = !g.empty() && g.size() <= 10 && (g.size() == 1 || g[0] != '0');
				uint64_t v = 0;
				for (char ch : g) { if (ch < '0' || ch > '9') { idx = false; break; } v = v * 10 + (uint64_t)(ch - '0'); }
				if (idx && v < 4294967295ull) ints.emplace_back(v, g);
			}
			std::sort(ints.begin(), ints.end());
			std::unordered_map<std::string, bool> is_int;
			for (auto &x : ints) order.push_back(x.second), is_int[x.second] = true;
			for (const std::string &g : paralog_order) if (!is_int.count(g)) order.push_back(g);
		}
		for (const std::string &g : order) {
			const std::string &pp = paralog[g];
			auto gi = seg_h.find(g), pi = seg_h.find(pp);
			if (gi == seg_h.end() || pi == seg_h.end()) continue;
			for (size_t a = 0; a < n_asm; ++a) mat[(size_t)pi->second * n_asm + a] += mat[(size_t)gi->second * n_asm + a];
		}
	}
	if (print_cd) { std::fflush(fp); return 0; }
	std::string o = "Gene\t";
	for (size_t a = 0; a < n_asm; ++a) { if (a) o += '\t'; o += asm_a[a]; }
	o += '\n';
	std::fwrite(o.data(), 1, o.size(), fp);
	for (size_t i = 0; i < seg.size(); ++i) {
		if (paralog.find(seg[i]) != paralog.end()) continue;
		o = seg[i]; o += '\t';
		for (size_t a = 0; a < n_asm; ++a) {
			int32_t v = mat[i * n_asm + a];
			if (!copy_number && v > 1) v = 1;
			if (a) o += '\t';
			put_i32(o, v);
		}
		o += '\n';
		std::fwrite(o.data(), 1, o.size(), fp);
	}
	std::fflush(fp);
	return 0;
}

} // extern "C"
