// gfa_writer.cpp -- BED / GFA S-,L-,W-line emission (the reference's format.c:78-225).  Host-side;
// bytes must equal the reference's, so integer formatting follows its pg_sprintf_lite (format.c:24-76:
// "%ld" arguments are narrowed to int before printing) and the id:f tag its "%.4f".
#include <cstdlib>
#include <cstring>
#include <string>
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
void pg_write_walk(pg_graph_t *q)
{
	pg_data_t *d = q->d;
	if (sync_host(d, false) != 0) return;
	DataExt *ext = ext_of(d, false);
	if (ext == nullptr || ext->ctx == nullptr) { set_error(PGA_ERR_ARG, "pg_write_walk: pg_graph_gen has not run on this data set"); return; }
	FILE *fp = out_stream();
	std::string o, sample;
	std::vector<int64_t> goff_of((size_t)d->n_genome, -1);
	for (size_t k = 0; k < ext->local_genomes.size(); ++k) goff_of[(size_t)ext->local_genomes[k]] = ext->hit_off[k];
	for (int32_t j = 0; j < d->n_genome; ++j) {
		const pg_genome_t *g = &d->genome[j];
		if (g->n_hit == 0) continue;
		const int64_t goff = goff_of[(size_t)j];
		if (goff < 0 || (size_t)j >= ext->y_file.size() || ext->y_file[(size_t)j].size() != (size_t)g->n_hit) { set_error(PGA_ERR_ARG, "pg_write_walk: genome without backend state"); return; }
		const int32_t *yf = ext->y_file[(size_t)j].data();
		const int32_t *hof = ext->hits_sorted[(size_t)j] ? ext->host_of_file[(size_t)j].data() : nullptr;
		const int32_t *px = ext->pos_x.data() + goff;
		const uint64_t *fb = ext->flt_bits.data();
		auto hit_of = [&](int32_t k) -> const pg_hit_t * { const int32_t f = yf[k]; return &g->hit[hof ? hof[f] : f]; };
		auto is_flt = [&](int32_t k) { const int64_t b = goff + px[yf[k]]; return (fb[b >> 6] >> (b & 63) & 1) != 0; };
		for (int32_t i0 = 0, i = 1; i <= g->n_hit; ++i) {
			if (i != g->n_hit && hit_of(i)->cid == hit_of(i0)->cid) continue;
			int32_t cid = hit_of(i0)->cid, n = 0;
			int32_t hap = parse_sample(sample, g->ctg[cid].name);
			o.clear();
			if (hap >= 0) { o += "W\t"; o += sample; o += '\t'; put_i32(o, hap); }
			else if (g->label) { o += "W\t"; o += g->label; o += "\t0"; }
			else { o += "W\t"; put_i32(o, j); o += "\t0"; }
			o += '\t'; o += g->ctg[cid].name; o += "\t*\t*\t";
			for (int32_t k = i0; k < i; ++k) {
				if (is_flt(k)) continue;
				const pg_hit_t *a = hit_of(k);
				o += "><"[a->rev]; o += d->gene[d->prot[a->pid].gid].name;
				++n;
			}
			if (n > 0) {
				o += "\tlf:B:i";
				for (int32_t k = i0; k < i; ++k) {
					if (is_flt(k)) continue;
					o += ','; put_i32(o, hit_of(k)->lof);
				}
				o += '\n';
				std::fwrite(o.data(), 1, o.size(), fp);
			}
			i0 = i;
		}
	}
	std::fflush(fp);
}

} // extern "C"
