/*
 * oracle.c -- TEST INFRASTRUCTURE ONLY.  Plain-C (C99) restatement of the per-hit part of
 * lh3/pangene v1.1-r231's graph-construction path, behind the same C ABI as the HIP product
 * (include/pangene_hip.h, prefix pgo_ instead of pga_).  It is the checker the parity tests compare
 * the kernels against and the partner of the host driver in the CPU-only tests; nothing in the product
 * links, imports or executes it.
 *
 * Pinning: tests/test_oracle_vs_ref.py runs the host driver on top of this file over every fixture in
 * tests/golden/ and requires the GFA / BED bytes to equal the outputs of the untouched reference
 * (oracle/_ref/pangene_ref, built by oracle/Makefile from the sources under /root/reference).
 *
 * Each function cites the reference lines it follows.  Order of hits: canonical X = (contig, cs, file
 * index) and Y = (contig, cm, X position) per genome, i.e. what the reference's pg_hit_sort
 * (hit.c:29-64) would give with a stable sort (its radix sort is unstable).
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "pangene_hip.h"

PGA_DECLARE(pgo)
const pga_backend_t *pgo_backend(void);

#define MALLOC(type, n) ((type*)malloc((size_t)((n) > 0 ? (n) : 1) * sizeof(type)))
#define CALLOC(type, n) ((type*)calloc((size_t)((n) > 0 ? (n) : 1), sizeof(type)))

struct pga_ctx {
	int32_t n_genome, n_genome_global, n_prot, n_gene;
	int64_t n_hit, n_exon;
	int64_t *off;             /* [n_genome+1] */
	int32_t *genome_global, *n_ctg;
	/* per hit, X order (genome-major) */
	int32_t *fidx;            /* file index inside the genome */
	int32_t *pid, *gid, *cid, *rank, *score_ori, *score_adj, *score_dom, *n_exon_of, *off_exon, *cds;
	int64_t *cs, *ce, *cm;  /* pangene.h:71 has int64_t; a block with virtual contigs (pga_genome_block_t) is put together again at load time */
	int32_t *pid_dom, *pid_dom0;
	uint32_t *flags;
	int32_t *yo;              /* Y order: yo[off[j]+k] = X position (global) of the k-th hit in cm order */
	int32_t *exon_os, *exon_oe;
	int32_t *prot_gid;
	uint8_t *gene_pref;
	pga_params_t par;
	/* exchange buffers */
	int32_t *max_ori; int64_t *sums;
	int32_t *vtx_cnt; uint64_t *triples; int64_t n_triples, m_triples;
	uint64_t *vtx_rec; /* folded (sub, dom) records handed to the driver */
	int32_t *ctg_base; int32_t hz_seg[PGA_HAZARD_CAP]; int64_t hz_n; /* contig-segment ids of the h2_cm / h3 hazard events */
	int32_t *g2s; int32_t n_seg;
	int32_t *seg_cnt; pga_arc_part_t *arcs; int64_t n_arcs, m_arcs;
	/* rep_pos: per local genome, per gene */
	int64_t *rp_x; int64_t *rp_y;  /* rp_x = cid<<32|r or -1 */
	int32_t *rp_iv;           /* H2b: (#walkable hits of the representative's (contig, cs) tie group before it) << 16 | (#after it) */
	int32_t *nl_cnt;
	pga_hazard_t hz;
	void *scratch; size_t m_scratch;
	/* branch state kept between branch_pairs and branch_decide / mark_hits */
	uint64_t *br_x; int32_t *br_s1, *br_gid, *br_pairs; uint8_t *br_weak; int64_t br_n, br_np; int32_t br_S;
	const pga_arc_part_t *cur_tab; int64_t cur_tab_n; /* the table of arc_set_current */
	int32_t *def_sc, *def_deg; /* results of a deferred arc_round_local */
	int64_t *head;            /* [n_genome] X position of the hit that plays "index 0" (never reset by pg_shadow) */
	/* raw shard, file order (kept so that begin() can restart the run) */
	int32_t *r_pid, *r_cid, *r_rank, *r_sori, *r_sadj, *r_nex, *r_offx; int64_t *r_cs, *r_ce, *r_cm; uint8_t *r_rev;
};

int pgo_is_device(void) { return 0; }
int pgo_host_alloc(size_t nbytes, void **ptr) { *ptr = malloc(nbytes ? nbytes : 1); return *ptr ? PGA_OK : PGA_ERR_NOMEM; }
void pgo_host_free(void *ptr) { free(ptr); }

const char *pgo_strerror(int code)
{
	switch (code) {
	case PGA_OK: return "ok";
	case PGA_ERR_NO_DEVICE: return "no device";
	case PGA_ERR_RANGE: return "value out of range for the device layout";
	case PGA_ERR_ARG: return "bad argument";
	case PGA_ERR_NOMEM: return "out of memory";
	case PGA_ERR_INVARIANT: return "reference invariant violated";
	}
	return "unknown";
}

/* ---- small helpers ---- */

/* pg_hash_uint32, pgpriv.h:88-97 (the score tie-breaker) */
static void hz_note(pga_ctx_t *c, int32_t j, int32_t cid) { if (c->hz_n < PGA_HAZARD_CAP) c->hz_seg[c->hz_n] = c->ctg_base[j] + cid; c->hz_n++; }

static inline uint32_t hash32(uint32_t key)
{
	key += ~(key << 15);
	key ^=  (key >> 10);
	key +=  (key << 3);
	key ^=  (key >> 6);
	key += ~(key << 11);
	key ^=  (key >> 16);
	return key;
}

typedef struct { int64_t k1, k2; int32_t k3, v; } skey_t;

static int skey_cmp(const void *a_, const void *b_)
{
	const skey_t *a = (const skey_t*)a_, *b = (const skey_t*)b_;
	if (a->k1 != b->k1) return a->k1 < b->k1 ? -1 : 1;
	if (a->k2 != b->k2) return a->k2 < b->k2 ? -1 : 1;
	if (a->k3 != b->k3) return a->k3 < b->k3 ? -1 : 1;
	return 0;
}

static inline int is_flt(const pga_ctx_t *c, int64_t i) { return (c->flags[i] & PGA_F_FLT) != 0; }
static inline int weak_of(const pga_ctx_t *c, int64_t i) { return (c->flags[i] & PGA_F_WEAK_MASK) >> PGA_F_WEAK_SHIFT; }

/* pg_hit_overlap, overlap.c:6-42: CDS intersection length of two hits on one contig (union unused) */
static int32_t cds_inter(const pga_ctx_t *c, int64_t a, int64_t b)
{
	int32_t ia = 0, ib = 0, na = c->n_exon_of[a], nb = c->n_exon_of[b];
	const int32_t *as = c->exon_os + c->off_exon[a], *ae = c->exon_oe + c->off_exon[a];
	const int32_t *bs = c->exon_os + c->off_exon[b], *be = c->exon_oe + c->off_exon[b];
	int64_t ca = c->cs[a], cb = c->cs[b], inter = 0;
	if (c->cid[a] != c->cid[b] || !(c->cs[a] < c->ce[b] && c->ce[a] > c->cs[b])) return 0; /* overlap.c:12 */
	while (ia < na && ib < nb) { /* two-pointer merge, overlap.c:17-33 */
		int64_t s0 = ca + as[ia], e0 = ca + ae[ia], s1 = cb + bs[ib], e1 = cb + be[ib];
		if (s0 < s1) { /* x = a */
			if (e0 < e1) { int64_t o = e0 - s1; inter += o > 0 ? o : 0; ++ia; }
			else { inter += e1 - s1; ++ib; }
		} else { /* x = b */
			if (e1 < e0) { int64_t o = e1 - s0; inter += o > 0 ? o : 0; ++ib; }
			else { inter += e0 - s0; ++ia; }
		}
	}
	return (int32_t)inter;
}

/* 64-bit comparison score, overlap.c:137-138 */
static inline uint64_t score64(const pga_ctx_t *c, int64_t i)
{
	return (uint64_t)c->score_adj[i] << 33 | (uint64_t)c->gene_pref[c->gid[i]] << 32 | hash32((uint32_t)c->pid[i]);
}

void pgo_destroy(pga_ctx_t *c)
{
	if (c == 0) return;
	free(c->off); free(c->genome_global); free(c->n_ctg); free(c->fidx);
	free(c->pid); free(c->gid); free(c->cid); free(c->rank); free(c->score_ori); free(c->score_adj); free(c->score_dom);
	free(c->n_exon_of); free(c->off_exon); free(c->cs); free(c->ce); free(c->cm); free(c->cds);
	free(c->pid_dom); free(c->pid_dom0); free(c->flags); free(c->yo); free(c->exon_os); free(c->exon_oe);
	free(c->prot_gid); free(c->gene_pref); free(c->max_ori); free(c->sums); free(c->vtx_cnt); free(c->triples); free(c->vtx_rec); free(c->ctg_base);
	free(c->g2s); free(c->seg_cnt); free(c->arcs); free(c->rp_x); free(c->rp_y); free(c->rp_iv); free(c->nl_cnt); free(c->scratch); free(c->head);
	free(c->br_x); free(c->br_s1); free(c->br_gid); free(c->br_pairs); free(c->br_weak); free(c->def_sc); free(c->def_deg);
	free(c->r_pid); free(c->r_cid); free(c->r_rank); free(c->r_sori); free(c->r_sadj); free(c->r_nex); free(c->r_offx); free(c->r_cs); free(c->r_ce); free(c->r_cm); free(c->r_rev);
	free(c);
}

#define DUP(type, dst, src, n) do { dst = MALLOC(type, n); memcpy(dst, src, (size_t)(n) * sizeof(type)); } while (0)

/* copy the shard (file order) */
int pgo_create(pga_ctx_t **out, const pga_shard_t *sh, const pga_params_t *par)
{
	pga_ctx_t *c;
	int64_t N = sh->n_hit, i;
	if (out == 0 || sh == 0 || par == 0 || sh->abi_version != PGA_ABI_VERSION) return PGA_ERR_ARG;
	c = CALLOC(pga_ctx_t, 1);
	c->n_genome = sh->n_genome, c->n_genome_global = sh->n_genome_global, c->n_prot = sh->n_prot, c->n_gene = sh->n_gene;
	c->n_hit = N, c->n_exon = sh->n_exon, c->par = *par;
	c->off = CALLOC(int64_t, sh->n_genome + 1);
	DUP(int32_t, c->genome_global, sh->genome_global, sh->n_genome);
	c->n_ctg = CALLOC(int32_t, sh->n_genome);
	{ int32_t g; c->ctg_base = CALLOC(int32_t, sh->n_genome + 1); for (g = 0; g < sh->n_genome; ++g) c->ctg_base[g + 1] = c->ctg_base[g] + sh->block[g].n_ctg; }
	c->exon_os = MALLOC(int32_t, sh->n_exon); c->exon_oe = MALLOC(int32_t, sh->n_exon);
	DUP(int32_t, c->prot_gid, sh->prot_gid, sh->n_prot); DUP(uint8_t, c->gene_pref, sh->gene_pref, sh->n_gene);
	c->r_pid = MALLOC(int32_t, N); c->r_cid = MALLOC(int32_t, N); c->r_rank = MALLOC(int32_t, N); c->r_sori = MALLOC(int32_t, N); c->r_sadj = MALLOC(int32_t, N);
	c->r_nex = MALLOC(int32_t, N); c->r_offx = MALLOC(int32_t, N); c->r_cs = MALLOC(int64_t, N); c->r_ce = MALLOC(int64_t, N); c->r_cm = MALLOC(int64_t, N);
	c->r_rev = MALLOC(uint8_t, N);
	{ /* unpack the per-genome blocks (pga_genome_block_t) into flat file-order arrays; exon offsets become shard-wide */
		int32_t g; int64_t hb = 0, eb = 0;
		for (g = 0; g < sh->n_genome; ++g) {
			const pga_genome_block_t *b = &sh->block[g];
			const int64_t n = b->n_hit;
			const int32_t *w = b->data, *ex = w + PGA_BLOCK_PLANES * n + (n + 3) / 4;
			const uint8_t *rev = (const uint8_t *)(w + PGA_BLOCK_PLANES * n);
			for (i = 0; i < n; ++i) {
				c->r_pid[hb + i] = w[i], c->r_cid[hb + i] = w[n + i], c->r_rank[hb + i] = w[2 * n + i], c->r_sori[hb + i] = w[3 * n + i];
				c->r_sadj[hb + i] = w[4 * n + i], c->r_nex[hb + i] = w[5 * n + i], c->r_offx[hb + i] = (int32_t)(eb + w[6 * n + i]);
				c->r_cs[hb + i] = w[7 * n + i], c->r_ce[hb + i] = w[8 * n + i], c->r_cm[hb + i] = w[9 * n + i], c->r_rev[hb + i] = rev[i];
				if (b->vfirst && b->vbase && w[n + i] >= 0 && w[n + i] < b->n_ctg) { /* a piece of a virtual contig: back to the contig and its own coordinates */
					const int64_t base = b->vbase[w[n + i]];
					c->r_cid[hb + i] = b->vfirst[w[n + i]], c->r_cs[hb + i] += base, c->r_ce[hb + i] += base, c->r_cm[hb + i] += base;
				}
			}
			for (i = 0; i < b->n_exon; ++i) c->exon_os[eb + i] = ex[2 * i], c->exon_oe[eb + i] = ex[2 * i + 1];
			c->off[g] = hb, c->n_ctg[g] = b->n_ctg;
			hb += n, eb += b->n_exon;
		}
		c->off[sh->n_genome] = hb;
		if (hb != N || eb != sh->n_exon) { pgo_destroy(c); return PGA_ERR_ARG; }
	}
	c->fidx = MALLOC(int32_t, N); c->pid = MALLOC(int32_t, N); c->gid = MALLOC(int32_t, N); c->cid = MALLOC(int32_t, N);
	c->rank = MALLOC(int32_t, N); c->score_ori = MALLOC(int32_t, N); c->score_adj = MALLOC(int32_t, N);
	c->score_dom = CALLOC(int32_t, N); c->n_exon_of = MALLOC(int32_t, N); c->off_exon = MALLOC(int32_t, N);
	c->cs = MALLOC(int64_t, N); c->ce = MALLOC(int64_t, N); c->cm = MALLOC(int64_t, N); c->cds = MALLOC(int32_t, N);
	c->pid_dom = MALLOC(int32_t, N); c->pid_dom0 = CALLOC(int32_t, N); c->flags = CALLOC(uint32_t, N);
	c->yo = MALLOC(int32_t, N);
	c->head = MALLOC(int64_t, sh->n_genome);
	c->max_ori = CALLOC(int32_t, c->n_prot);
	c->sums = CALLOC(int64_t, 6 * (int64_t)c->n_prot);
	c->vtx_cnt = CALLOC(int32_t, 2 * (int64_t)c->n_gene);
	c->g2s = MALLOC(int32_t, c->n_gene);
	for (i = 0; i < c->n_gene; ++i) c->g2s[i] = -1;
	*out = c;
	return PGA_OK;
}

/* pg_hit_sort (hit.c:29-64) in canonical order + pg_cds_len (overlap.c:45-51); state as read.c:133-134 */
int pgo_begin(pga_ctx_t *c)
{
	int64_t N = c->n_hit, i, k;
	int32_t j;
	skey_t *key = MALLOC(skey_t, N);
	memset(&c->hz, 0, sizeof(c->hz));
	c->hz_n = 0;
	for (j = 0; j < c->n_genome; ++j) c->head[j] = c->off[j];
	for (i = 0; i < c->n_gene; ++i) c->g2s[i] = -1;
	c->n_seg = 0;
	for (j = 0; j < c->n_genome; ++j) { /* X order */
		int64_t st = c->off[j], en = c->off[j + 1];
		for (i = st; i < en; ++i) key[i].k1 = c->r_cid[i], key[i].k2 = c->r_cs[i], key[i].k3 = (int32_t)(i - st), key[i].v = (int32_t)(i - st);
		qsort(key + st, (size_t)(en - st), sizeof(skey_t), skey_cmp);
		for (i = st; i < en; ++i) {
			int64_t s = st + key[i].v;
			int32_t e, len = 0;
			c->fidx[i] = key[i].v;
			c->pid[i] = c->r_pid[s], c->cid[i] = c->r_cid[s], c->rank[i] = c->r_rank[s];
			c->score_ori[i] = c->r_sori[s], c->score_adj[i] = c->r_sadj[s];
			c->n_exon_of[i] = c->r_nex[s], c->off_exon[i] = c->r_offx[s];
			c->cs[i] = c->r_cs[s], c->ce[i] = c->r_ce[s], c->cm[i] = c->r_cm[s];
			c->gid[i] = c->prot_gid[c->r_pid[s]];
			c->flags[i] = c->r_rev[s] ? PGA_F_REV : 0;
			c->pid_dom[i] = -1, c->pid_dom0[i] = 0, c->score_dom[i] = 0; /* read.c:133-134 */
			for (e = 0; e < c->r_nex[s]; ++e)
				len += c->exon_oe[c->r_offx[s] + e] - c->exon_os[c->r_offx[s] + e];
			c->cds[i] = len;
		}
	}
	for (j = 0; j < c->n_genome; ++j) { /* Y order */
		int64_t st = c->off[j], en = c->off[j + 1];
		for (i = st; i < en; ++i) key[i].k1 = c->cid[i], key[i].k2 = c->cm[i], key[i].k3 = (int32_t)(i - st), key[i].v = (int32_t)i;
		qsort(key + st, (size_t)(en - st), sizeof(skey_t), skey_cmp);
		for (k = st; k < en; ++k) c->yo[k] = key[k].v;
	}
	free(key);
	return PGA_OK;
}

/* pg_flag_pseudo, hit.c:66-105, one genome.  Keys (pid, rank) are unique inside a genome so the
 * result does not depend on the order of the hit array. */
static int32_t flag_pseudo(pga_ctx_t *c, int32_t j, int32_t *maxn, int32_t *minn, int32_t *r1)
{
	int64_t st = c->off[j], en = c->off[j + 1], i;
	int32_t n_pseudo = 0;
	for (i = st; i < en; ++i) maxn[c->pid[i]] = 0, minn[c->pid[i]] = INT32_MAX, r1[c->pid[i]] = INT32_MAX;
	for (i = st; i < en; ++i) { /* hit.c:78-83 */
		int32_t p = c->pid[i], ne = c->n_exon_of[i];
		if (ne > maxn[p]) maxn[p] = ne;
		if (ne < minn[p]) minn[p] = ne;
	}
	for (i = st; i < en; ++i) { /* hit.c:84-92 */
		int32_t p = c->pid[i], ne = c->n_exon_of[i];
		if (!(maxn[p] > 1 && (minn[p] == 1 || minn[p] * 2 <= maxn[p]))) continue;
		if (ne == 1 || ne * 2 <= maxn[p]) c->flags[i] |= PGA_F_PSEUDO, ++n_pseudo;
		else if (c->rank[i] < r1[p]) r1[p] = c->rank[i]; /* j1 = first non-pseudo in rank order */
	}
	for (i = st; i < en; ++i) { /* promote the first multi-exon hit to rank 0, hit.c:94-98 */
		int32_t p = c->pid[i];
		if (!(maxn[p] > 1 && (minn[p] == 1 || minn[p] * 2 <= maxn[p]))) continue;
		if (r1[p] == INT32_MAX || r1[p] == 0) continue;
		if (c->rank[i] < r1[p]) c->rank[i]++;
		else if (c->rank[i] == r1[p]) c->rank[i] = 0;
	}
	return n_pseudo;
}

typedef struct { uint64_t score; int64_t aid; int32_t ov_len; } shadow_aux_t;

/* pg_shadow, overlap.c:101-178, one genome (array in cs order).  Returns #shadowed non-flt hits. */
static int32_t shadow_genome(pga_ctx_t *c, int32_t j, int cal_dom_sc, int32_t *n_tot)
{
	int64_t st = c->off[j], en = c->off[j + 1], i, i0, jj;
	int32_t n_shadow = 0, tot = 0;
	shadow_aux_t *tmp = CALLOC(shadow_aux_t, en - st);
	for (i = st, i0 = st; i < en; ++i) { /* overlap.c:108: the loop starts at 1 => the hit at index 0 is never reset */
		int32_t li, gi;
		uint64_t si;
		if (is_flt(c, i)) continue;
		if (i != c->head[j]) c->flags[i] &= ~PGA_F_SHADOW;
		while (i0 < i && !(c->cid[i0] == c->cid[i] && c->ce[i0] > c->cs[i])) ++i0; /* overlap.c:114-115 */
		gi = c->gid[i], li = c->cds[i], si = score64(c, i);
		for (jj = i0; jj < i; ++jj) {
			int32_t x, lj, gj, sh;
			double cov_short;
			uint64_t sj;
			if (c->ce[jj] <= c->cs[i]) continue;
			if (is_flt(c, jj)) continue;
			if (c->par.check_strand && ((c->flags[i] ^ c->flags[jj]) & PGA_F_REV)) continue;
			gj = c->gid[jj];
			x = cds_inter(c, jj, i);
			if (x == 0) continue; /* overlap.c:132 */
			lj = c->cds[jj];
			cov_short = (double)x / (li < lj ? li : lj);
			if (gi != gj && cov_short < c->par.min_ov_ratio) continue; /* overlap.c:136 */
			sj = score64(c, jj);
			if (gi == gj || weak_of(c, i) == weak_of(c, jj)) /* overlap.c:139-142 */
				sh = (si < sj || (si == sj && c->rank[i] > c->rank[jj])) ? 0 : 1;
			else if (weak_of(c, i) > weak_of(c, jj)) sh = 0;
			else sh = 1;
			if (sh == 0) { /* overlap.c:148-154 */
				c->flags[i] |= PGA_F_SHADOW;
				if (tmp[i - st].score == sj && sj > 0) { c->hz.h3_dom_tie++; hz_note(c, j, c->cid[i]); }
				if (tmp[i - st].score < sj) tmp[i - st].score = sj, tmp[i - st].aid = jj, tmp[i - st].ov_len = x;
			} else {
				c->flags[jj] |= PGA_F_SHADOW;
				if (tmp[jj - st].score == si && si > 0) { c->hz.h3_dom_tie++; hz_note(c, j, c->cid[jj]); }
				if (tmp[jj - st].score < si) tmp[jj - st].score = si, tmp[jj - st].aid = i, tmp[jj - st].ov_len = x;
			}
		}
	}
	for (i = st; i < en; ++i) { /* overlap.c:157-175 */
		if (is_flt(c, i)) continue;
		++tot;
		c->pid_dom[i] = -1;
		if (cal_dom_sc) c->score_dom[i] = -1;
		if (tmp[i - st].score > 0) {
			int64_t a = tmp[i - st].aid;
			c->pid_dom[i] = c->pid[a];
			if (cal_dom_sc) {
				int32_t li = c->cds[i], lj = c->cds[a];
				c->score_dom[i] = (int32_t)(c->score_ori[i] * (1.0 - (double)tmp[i - st].ov_len / li)
				                            + c->score_ori[a] * ((double)tmp[i - st].ov_len / lj) + .499);
			}
		}
		if (c->flags[i] & PGA_F_SHADOW) ++n_shadow;
	}
	free(tmp);
	if (n_tot) *n_tot = tot;
	return n_shadow;
}

/* pg_flt_ov_isoform, overlap.c:58-93, one genome */
static int32_t flt_ov_isoform(pga_ctx_t *c, int32_t j)
{
	int64_t st = c->off[j], en = c->off[j + 1], i, i0, jj;
	int32_t n_flt = 0;
	for (i = st + 1, i0 = st; i < en; ++i) {
		uint64_t si;
		if (is_flt(c, i)) continue;
		while (i0 < i && !(c->cid[i0] == c->cid[i] && c->ce[i0] > c->cs[i])) ++i0;
		si = score64(c, i);
		for (jj = i0; jj < i; ++jj) {
			uint64_t sj;
			if (is_flt(c, jj) || c->ce[jj] <= c->cs[i]) continue;
			if (c->gid[i] != c->gid[jj]) continue;
			if (c->par.check_strand && ((c->flags[i] ^ c->flags[jj]) & PGA_F_REV)) continue;
			if (cds_inter(c, jj, i) == 0) continue;
			sj = score64(c, jj);
			if (si < sj || (si == sj && c->rank[i] > c->rank[jj])) c->flags[i] |= PGA_F_ISO_OV;
			else c->flags[jj] |= PGA_F_ISO_OV;
		}
	}
	for (i = st; i < en; ++i)
		if (c->flags[i] & PGA_F_ISO_OV) c->flags[i] |= PGA_F_FLT, ++n_flt;
	return n_flt;
}

/* pg_flt_chain_shadow, hit.c:130-146, one genome; flag[] has n_prot entries, all 1 on entry and exit */
static int32_t flt_chain_shadow(pga_ctx_t *c, int32_t j, int8_t *flag)
{
	int64_t st = c->off[j], en = c->off[j + 1], i;
	int32_t n_flt = 0;
	for (i = st; i < en; ++i)
		if (!(c->flags[i] & PGA_F_ISO_OV)) flag[c->pid[i]] = 0;
	for (i = st; i < en; ++i)
		if (c->pid_dom0[i] >= 0 && flag[c->pid_dom0[i]])
			c->flags[i] |= PGA_F_FLT | PGA_F_CHAIN, ++n_flt;
	for (i = st; i < en; ++i) flag[c->pid[i]] = 1;
	return n_flt;
}

/* pg_flt_subopt_isoform, hit.c:107-128, one genome; best[] / bidx[] have n_gene zeroed entries on entry and exit.
 * Tie order: the winner of a gene is the FIRST candidate with the maximal score_adj in array order (the last
 * one among negative scores); a candidate of another protein with the winner's score and the winner's (contig, cs) could sit
 * before it in the reference's unstable order: hazard. */
static int32_t flt_subopt_isoform(pga_ctx_t *c, int32_t j, uint64_t *best, int64_t *bidx)
{
	int64_t st = c->off[j], en = c->off[j + 1], i;
	int32_t n_flt = 0;
	for (i = st; i < en; ++i) {
		if (is_flt(c, i) || c->rank[i] > 0) continue;
		if (c->score_adj[i] > 0 && (uint64_t)c->score_adj[i] > best[c->gid[i]] >> 32) { /* hit.c:116; the cast there makes negatives huge */
			best[c->gid[i]] = (uint64_t)c->score_adj[i] << 32 | (uint32_t)c->pid[i], bidx[c->gid[i]] = i;
		} else if (c->score_adj[i] < 0) { /* (int32 > uint64) promotes to unsigned: a negative score_adj always wins */
			best[c->gid[i]] = (uint64_t)c->score_adj[i] << 32 | (uint32_t)c->pid[i], bidx[c->gid[i]] = i;
		}
	}
	for (i = st; i < en; ++i) {
		int64_t w;
		if (is_flt(c, i) || c->rank[i] > 0 || best[c->gid[i]] == 0) continue;
		w = bidx[c->gid[i]];
		if (c->pid[i] != c->pid[w] && c->cid[i] == c->cid[w] && c->cs[i] == c->cs[w] &&
		    (c->score_adj[w] < 0 ? c->score_adj[i] < 0 : c->score_adj[i] == c->score_adj[w])) { c->hz.h3_dom_tie++; hz_note(c, j, c->cid[i]); }
	}
	for (i = st; i < en; ++i) {
		if (is_flt(c, i)) continue;
		if (c->pid[i] != (int32_t)best[c->gid[i]])
			c->flags[i] |= PGA_F_FLT | PGA_F_ISO_SUB, ++n_flt;
	}
	for (i = st; i < en; ++i) best[c->gid[i]] = 0;
	return n_flt;
}

/* stage A, read.c:243-260 */
int pgo_ingest(pga_ctx_t *c, int32_t *stats)
{
	int32_t j, *maxn, *minn, *r1;
	int8_t *flag;
	uint64_t *best; int64_t *bidx;
	int64_t i;
	maxn = CALLOC(int32_t, c->n_prot), minn = CALLOC(int32_t, c->n_prot), r1 = CALLOC(int32_t, c->n_prot);
	flag = MALLOC(int8_t, c->n_prot);
	for (i = 0; i < c->n_prot; ++i) flag[i] = 1;
	best = CALLOC(uint64_t, c->n_gene); bidx = CALLOC(int64_t, c->n_gene);
	for (j = 0; j < c->n_genome; ++j) {
		int32_t n_pseudo, n_ov, n_chain, n_sub;
		n_pseudo = flag_pseudo(c, j, maxn, minn, r1);
		for (i = c->off[j]; i < c->off[j + 1]; ++i) /* PG_SET_FILTER(d, pseudo == 1), read.c:246 */
			if (c->flags[i] & PGA_F_PSEUDO) c->flags[i] |= PGA_F_FLT;
		shadow_genome(c, j, 1, 0); /* read.c:248 */
		for (i = c->off[j]; i < c->off[j + 1]; ++i) { /* read.c:249-253 */
			c->pid_dom0[i] = c->pid_dom[i];
			c->pid_dom[i] = -1, c->flags[i] &= ~PGA_F_SHADOW;
		}
		n_ov = flt_ov_isoform(c, j);
		n_chain = flt_chain_shadow(c, j, flag);
		n_sub = flt_subopt_isoform(c, j, best, bidx);
		if (stats) stats[j*4] = n_pseudo, stats[j*4+1] = n_ov, stats[j*4+2] = n_chain, stats[j*4+3] = n_sub;
	}
	free(maxn); free(minn); free(r1); free(flag); free(best); free(bidx);
	return PGA_OK;
}

/* partial reductions of pg_cap_score_dom (hit.c:230-238), pg_flag_representative (hit.c:196-204),
 * pg_flag_pseudo_joint (hit.c:158-169) */
int pgo_post_partials(pga_ctx_t *c, int32_t **max_ori, int64_t **sums)
{
	int64_t i, P = c->n_prot;
	memset(c->max_ori, 0, P * sizeof(int32_t));
	memset(c->sums, 0, 6 * P * sizeof(int64_t));
	for (i = 0; i < c->n_hit; ++i) {
		int32_t p = c->pid[i];
		if (c->score_ori[i] > c->max_ori[p]) c->max_ori[p] = c->score_ori[i];
		if (c->rank[i] == 0 && !is_flt(c, i)) {
			int w = c->n_exon_of[i] == 1 ? 0 : 1;
			c->sums[0*P + p] += c->score_adj[i];
			c->sums[1*P + p] += 1;
			c->sums[(2 + w)*P + p] += 1;
			c->sums[(4 + w)*P + p] += c->score_ori[i];
		}
	}
	*max_ori = c->max_ori, *sums = c->sums;
	return PGA_OK;
}

int pgo_post_apply(pga_ctx_t *c, const uint8_t *prot_rep, const uint8_t *prot_pj, int64_t *n_pseudo)
{
	int64_t i, n = 0;
	for (i = 0; i < c->n_hit; ++i) {
		int32_t p = c->pid[i];
		if (c->score_dom[i] > c->max_ori[p]) c->score_dom[i] = c->max_ori[p]; /* hit.c:243-244 */
		if (prot_rep[p]) c->flags[i] |= PGA_F_REP; else c->flags[i] &= ~PGA_F_REP; /* hit.c:202,222-223 */
		if (!is_flt(c, i) && !(c->flags[i] & PGA_F_PSEUDO) && c->n_exon_of[i] == 1 && prot_pj[p]) /* hit.c:175-182 */
			c->flags[i] |= PGA_F_PSEUDO, ++n;
	}
	if (n_pseudo) *n_pseudo = n;
	return PGA_OK;
}

int pgo_shadow(pga_ctx_t *c, int32_t cal_dom_sc, int32_t *stats)
{
	int32_t j;
	for (j = 0; j < c->n_genome; ++j) {
		int32_t tot, ns = shadow_genome(c, j, cal_dom_sc, &tot);
		if (stats) stats[2*j] = tot, stats[2*j+1] = ns;
	}
	return PGA_OK;
}

int pgo_set_filter(pga_ctx_t *c, int32_t which)
{
	int64_t i;
	for (i = 0; i < c->n_hit; ++i) {
		uint32_t f = c->flags[i];
		int hit = which == PGA_FLT_PSEUDO ? (f & PGA_F_PSEUDO) != 0
		        : which == PGA_FLT_VTX0 ? (f & PGA_F_VTX) == 0
		        : which == PGA_FLT_WEAK2 ? ((f & PGA_F_WEAK_MASK) >> PGA_F_WEAK_SHIFT) == 2
		        : which == PGA_FLT_SHADOW ? (f & PGA_F_SHADOW) != 0 : 0;
		if (hit) c->flags[i] |= PGA_F_FLT;
	}
	return PGA_OK;
}

static int cmp_u64(const void *a, const void *b) { uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return x < y ? -1 : x > y; }

/* per-genome part of pg_gen_vtx, vertex.c:21-51 */
int pgo_vtx_partials(pga_ctx_t *c, int32_t **cnt, uint64_t **records, int64_t *n_records)
{
	int32_t j, Q = c->n_gene;
	uint32_t *aux = MALLOC(uint32_t, Q);
	int8_t *flag = MALLOC(int8_t, Q);
	int64_t i;
	memset(c->vtx_cnt, 0, 2 * (size_t)Q * sizeof(int32_t));
	c->n_triples = 0;
	for (j = 0; j < c->n_genome; ++j) {
		int32_t g;
		for (g = 0; g < Q; ++g) aux[g] = (uint32_t)(Q + 1) << 1;
		memset(flag, 0, Q);
		for (i = c->off[j]; i < c->off[j + 1]; ++i) {
			int32_t gid;
			if (c->rank[i] != 0 || is_flt(c, i)) continue;
			gid = c->gid[i];
			if (c->flags[i] & PGA_F_SHADOW) {
				if (c->pid_dom[i] < 0) { free(aux); free(flag); return PGA_ERR_INVARIANT; } /* vertex.c:38 */
				flag[gid] |= 2;
				if (aux[gid] == (uint32_t)(Q + 1) << 1) aux[gid] = (uint32_t)c->prot_gid[c->pid_dom[i]] << 1;
			} else {
				flag[gid] |= 1;
				aux[gid] = (uint32_t)Q << 1;
			}
		}
		for (g = 0; g < Q; ++g) {
			if (flag[g] & 1) c->vtx_cnt[g]++;
			else if (flag[g] & 2) {
				uint32_t D = aux[g] >> 1;
				c->vtx_cnt[Q + g]++;
				if (aux[D] >> 1 == (uint32_t)Q) { /* only entries the greedy can observe */
					if (c->n_triples == c->m_triples) {
						c->m_triples = c->m_triples ? c->m_triples * 2 : 1024;
						c->triples = (uint64_t*)realloc(c->triples, c->m_triples * sizeof(uint64_t));
					}
					c->triples[c->n_triples++] = (uint64_t)c->genome_global[j] << 40 | (uint64_t)g << 20 | D;
				}
			}
		}
	}
	free(aux); free(flag);
	{ /* fold the (genome, sub, dom) triples into one record per (sub, dom): key, then the bit set of global genome indices */
		const int64_t nw = ((int64_t)c->n_genome_global + 63) / 64, stride = 1 + nw;
		const uint64_t m40 = (1ULL << 40) - 1;
		int64_t k, n_rec = 0;
		for (k = 0; k < c->n_triples; ++k) c->triples[k] = (c->triples[k] & m40) << 24 | c->triples[k] >> 40; /* (sub, dom) major, genome minor */
		if (c->n_triples) qsort(c->triples, (size_t)c->n_triples, sizeof(uint64_t), cmp_u64);
		free(c->vtx_rec);
		c->vtx_rec = (uint64_t*)calloc((size_t)(c->n_triples * stride + 1), sizeof(uint64_t));
		for (k = 0; k < c->n_triples; ++k) {
			const uint64_t key = c->triples[k] >> 24, g = c->triples[k] & 0xffffff;
			if (k == 0 || key != c->triples[k - 1] >> 24) c->vtx_rec[n_rec++ * stride] = key;
			c->vtx_rec[(n_rec - 1) * stride + 1 + (int64_t)(g >> 6)] |= 1ULL << (g & 63);
		}
		*cnt = c->vtx_cnt, *records = c->vtx_rec, *n_records = n_rec;
	}
	return PGA_OK;
}

/* pg_graph_flag_vtx, graph.c:61-69 */
int pgo_flag_vtx(pga_ctx_t *c, const int32_t *g2s, int32_t n_seg, int32_t then_filter)
{
	int64_t i;
	memcpy(c->g2s, g2s, c->n_gene * sizeof(int32_t));
	c->n_seg = n_seg;
	for (i = 0; i < c->n_hit; ++i) {
		if (g2s[c->gid[i]] >= 0) c->flags[i] |= PGA_F_VTX;
		else c->flags[i] &= ~PGA_F_VTX;
	}
	return then_filter ? pgo_set_filter(c, PGA_FLT_VTX0) : PGA_OK;
}

typedef struct { uint64_t x; int32_t n, dist, s1, s2; } tmparc_t;

static int tmparc_cmp(const void *a, const void *b)
{
	uint64_t x = ((const tmparc_t*)a)->x, y = ((const tmparc_t*)b)->x;
	return x < y ? -1 : x > y;
}

/* pg_get_score, graph.c:82-85 */
static inline int32_t get_score(const pga_ctx_t *c, int64_t i, int ori)
{
	return ori || c->score_ori[i] > c->score_dom[i] || c->pid_dom0[i] < 0 || c->g2s[c->prot_gid[c->pid_dom0[i]]] >= 0
		? c->score_ori[i] : c->score_dom[i];
}

/* pg_gen_arc, graph.c:87-177, over the local genomes, without the final roundings of 170-172 */
int pgo_arc_round(pga_ctx_t *c, int32_t use_ori, int32_t **seg_cnt_out, pga_arc_part_t **arcs_out, int64_t *n_arcs_out)
{
	int32_t j, S = c->n_seg, *cnt;
	int64_t n_arc = 0, m_arc = 0, n1, m1 = 0, i, i0, k;
	tmparc_t *arc = 0, *arc1 = 0;
	free(c->seg_cnt);
	c->seg_cnt = CALLOC(int32_t, 2 * (int64_t)S);
	cnt = MALLOC(int32_t, S);
	for (j = 0; j < c->n_genome; ++j) {
		uint32_t w, v = (uint32_t)-1;
		int64_t vpos = -1; int32_t vcid = -1, si = -1;
		shadow_genome(c, j, 0, 0); /* graph.c:102 */
		n1 = 0;
		memset(cnt, 0, S * sizeof(int32_t));
		for (k = c->off[j]; k < c->off[j + 1]; ++k) { /* cm order, graph.c:103-122 */
			int64_t a = c->yo[k];
			int32_t sid, sc;
			if (c->flags[a] & (PGA_F_FLT | PGA_F_SHADOW)) continue;
			sid = c->g2s[c->gid[a]];
			if (sid < 0) { free(cnt); free(arc); free(arc1); return PGA_ERR_INVARIANT; } /* graph.c:111 */
			w = (uint32_t)sid << 1 | (c->flags[a] & PGA_F_REV ? 1 : 0);
			++cnt[sid];
			if (c->cid[a] != vcid) v = (uint32_t)-1, vpos = -1;
			sc = get_score(c, a, use_ori);
			if (v != (uint32_t)-1) {
				if (c->cm[a] == vpos) { c->hz.h2_cm_tie++; hz_note(c, j, c->cid[a]); } /* hazard H2a: the order of the two decides the arc */
				if (n1 + 2 > m1) { m1 = m1 ? m1 * 2 : 1024; arc1 = (tmparc_t*)realloc(arc1, m1 * sizeof(tmparc_t)); }
				arc1[n1].x = (uint64_t)v << 32 | w, arc1[n1].dist = (int32_t)(c->cm[a] - vpos), arc1[n1].s1 = si, arc1[n1].s2 = sc, arc1[n1].n = 0, ++n1;
				arc1[n1].x = (uint64_t)(w^1) << 32 | (v^1), arc1[n1].dist = (int32_t)(c->cm[a] - vpos), arc1[n1].s1 = sc, arc1[n1].s2 = si, arc1[n1].n = 0, ++n1;
			}
			v = w, vpos = c->cm[a], vcid = c->cid[a], si = sc;
		}
		for (i = 0; i < S; ++i) /* graph.c:125-126 */
			c->seg_cnt[i] += cnt[i] > 0, c->seg_cnt[S + i] += cnt[i];
		if (n1) qsort(arc1, (size_t)n1, sizeof(tmparc_t), tmparc_cmp);
		for (i = 1, i0 = 0; i <= n1; ++i) { /* per-genome collapse, graph.c:128-145 */
			if (i == n1 || arc1[i0].x != arc1[i].x) {
				int32_t max_s1 = 0, max_s2 = 0;
				uint64_t dist = 0;
				for (k = i0; k < i; ++k) {
					dist += arc1[k].dist;
					max_s1 = max_s1 > arc1[k].s1 ? max_s1 : arc1[k].s1;
					max_s2 = max_s2 > arc1[k].s2 ? max_s2 : arc1[k].s2;
				}
				if (n_arc == m_arc) { m_arc = m_arc ? m_arc * 2 : 4096; arc = (tmparc_t*)realloc(arc, m_arc * sizeof(tmparc_t)); }
				arc[n_arc].x = arc1[i0].x, arc[n_arc].n = (int32_t)(i - i0);
				arc[n_arc].dist = (int32_t)((double)dist / (i - i0) + .499);
				arc[n_arc].s1 = max_s1, arc[n_arc].s2 = max_s2, ++n_arc;
				i0 = i;
			}
		}
	}
	free(arc1); free(cnt);
	if (n_arc) qsort(arc, (size_t)n_arc, sizeof(tmparc_t), tmparc_cmp); /* graph.c:151 */
	c->n_arcs = 0;
	for (i0 = 0, i = 1; i <= n_arc; ++i) { /* integer part of graph.c:153-169 */
		if (i == n_arc || arc[i].x != arc[i0].x) {
			pga_arc_part_t *p;
			if (c->n_arcs == c->m_arcs) { c->m_arcs = c->m_arcs ? c->m_arcs * 2 : 4096; c->arcs = (pga_arc_part_t*)realloc(c->arcs, c->m_arcs * sizeof(pga_arc_part_t)); }
			p = &c->arcs[c->n_arcs++];
			memset(p, 0, sizeof(*p));
			p->x = arc[i0].x, p->n_genome = (int32_t)(i - i0);
			for (k = i0; k < i; ++k) {
				p->tot_cnt += arc[k].n;
				p->sum_dist += (uint64_t)arc[k].dist * arc[k].n;
				p->sum_s1 += arc[k].s1, p->sum_s2 += arc[k].s2;
			}
			i0 = i;
		}
	}
	free(arc);
	*seg_cnt_out = c->seg_cnt, *arcs_out = c->arcs, *n_arcs_out = c->n_arcs;
	return PGA_OK;
}

static int arcpart_cmp(const void *a, const void *b)
{
	uint64_t x = ((const pga_arc_part_t*)a)->x, y = ((const pga_arc_part_t*)b)->x;
	return x < y ? -1 : x > y;
}

/* cross-shard reduce-by-key (the sums of graph.c:153-169 over the shards) */
int pgo_arc_merge(pga_ctx_t *c, const pga_arc_part_t *gathered, const int64_t *count, int32_t W, int64_t slot, pga_arc_part_t **out, int64_t *n_out)
{
	int64_t tot = 0, i, k = 0;
	int32_t r;
	pga_arc_part_t *t;
	for (r = 0; r < W; ++r) tot += count[r];
	t = MALLOC(pga_arc_part_t, tot);
	for (r = 0, i = 0; r < W; ++r) { memcpy(t + i, gathered + (int64_t)r * slot, count[r] * sizeof(pga_arc_part_t)); i += count[r]; }
	if (tot) qsort(t, (size_t)tot, sizeof(pga_arc_part_t), arcpart_cmp);
	for (i = 0; i < tot; ++i) {
		if (k > 0 && t[k-1].x == t[i].x) {
			t[k-1].n_genome += t[i].n_genome, t[k-1].tot_cnt += t[i].tot_cnt;
			t[k-1].sum_dist += t[i].sum_dist, t[k-1].sum_s1 += t[i].sum_s1, t[k-1].sum_s2 += t[i].sum_s2;
		} else t[k++] = t[i];
	}
	free(c->arcs);
	c->arcs = t, c->n_arcs = k, c->m_arcs = tot;
	*out = t, *n_out = k;
	return PGA_OK;
}

/* pg_gen_rep_pos, branch.c:6-29.
 * Tie order (hazard H2b): the reference's unstable sort may permute the hits that share (contig, cs).  The walkable
 * ones among them receive consecutive values of the running counter r in whatever order they end up, so the r of a representative
 * is only known up to the interval [r - nb, r + na] (nb / na = walkable members of its tie group before / after it in the canonical
 * order); pgo_n_local raises the hazard when that matters.  Two walkable hits of ONE gene inside a tie group (possible with -S:
 * opposite strands) make the representative itself order-dependent (branch.c:22-23: the last one wins): hazard at once. */
int pgo_rep_pos(pga_ctx_t *c)
{
	int64_t Q = c->n_gene, i, n = Q * c->n_genome;
	int32_t j;
	if (c->rp_x == 0) c->rp_x = MALLOC(int64_t, n), c->rp_y = MALLOC(int64_t, n), c->rp_iv = MALLOC(int32_t, n);
	for (i = 0; i < n; ++i) c->rp_x[i] = -1, c->rp_y[i] = 0, c->rp_iv[i] = 0;
	for (j = 0; j < c->n_genome; ++j) {
		int32_t r = 0;
		for (i = c->off[j]; i < c->off[j + 1]; ++i) {
			int64_t p;
			int32_t nb = 0, na = 0, same_gene = 0;
			if (c->flags[i] & (PGA_F_FLT | PGA_F_SHADOW)) continue;
			for (p = i - 1; p >= c->off[j] && c->cid[p] == c->cid[i] && c->cs[p] == c->cs[i]; --p)
				if (!(c->flags[p] & (PGA_F_FLT | PGA_F_SHADOW))) ++nb, same_gene |= c->gid[p] == c->gid[i];
			for (p = i + 1; p < c->off[j + 1] && c->cid[p] == c->cid[i] && c->cs[p] == c->cs[i]; ++p)
				if (!(c->flags[p] & (PGA_F_FLT | PGA_F_SHADOW))) ++na, same_gene |= c->gid[p] == c->gid[i];
			if (same_gene || nb > 0xffff || na > 0xffff) c->hz.h2_cs_tie++, hz_note(c, j, c->cid[i]), nb = na = 0;
			c->rp_x[j * Q + c->gid[i]] = (int64_t)c->cid[i] << 32 | r;
			c->rp_y[j * Q + c->gid[i]] = c->cm[i];
			c->rp_iv[j * Q + c->gid[i]] = nb << 16 | na;
			++r;
		}
	}
	return PGA_OK;
}

/* pg_n_local, branch.c:31-46, summed over the local genomes */
int pgo_n_local(pga_ctx_t *c, const int32_t *pairs, int64_t n, int32_t local_dist, int32_t local_count, int32_t frag_mode, int32_t **cnt)
{
	int64_t Q = c->n_gene, k;
	int32_t j;
	if (c->rp_x == 0) return PGA_ERR_ARG;
	free(c->nl_cnt);
	c->nl_cnt = CALLOC(int32_t, n);
	for (k = 0; k < n; ++k) {
		int32_t g1 = pairs[2*k], g2 = pairs[2*k+1], n_local = 0;
		for (j = 0; j < c->n_genome; ++j) {
			int64_t x1 = c->rp_x[j * Q + g1], x2 = c->rp_x[j * Q + g2], d;
			int32_t cc, iv1 = c->rp_iv[j * Q + g1], iv2 = c->rp_iv[j * Q + g2];
			if (x1 == -1 || x2 == -1) continue;
			if (!frag_mode && x1 >> 32 != x2 >> 32) continue;
			d = c->rp_y[j * Q + g1] - c->rp_y[j * Q + g2];
			cc = (int32_t)x1 - (int32_t)x2;
			if ((d >= -local_dist && d <= local_dist) || (cc >= -local_count && cc <= local_count)) ++n_local;
			if ((iv1 | iv2) && !(d >= -local_dist && d <= local_dist)) { /* H2b: is |r1 - r2| <= local_count the same for every tie order? */
				const int32_t lo = cc - (iv1 >> 16) - (iv2 & 0xffff), hi = cc + (iv1 & 0xffff) + (iv2 >> 16);
				const int all_in = lo >= -local_count && hi <= local_count, all_out = hi < -local_count || lo > local_count;
				if (!all_in && !all_out) {
					c->hz.h2_cs_tie++;
					if (iv1) hz_note(c, j, (int32_t)(x1 >> 32));
					if (iv2) hz_note(c, j, (int32_t)(x2 >> 32));
				}
			}
		}
		c->nl_cnt[k] = n_local;
	}
	*cnt = c->nl_cnt;
	return PGA_OK;
}


/* enumerate (and, with cnt != 0, consume) the pg_n_local calls of pg_mark_branch_flt_arc for one vertex, in the
 * reference's order (branch.c:64-90).  a0 = first arc, n = #arcs. */
static int64_t branch_vertex(pga_ctx_t *c, int64_t a0, int32_t n, double branch_diff, int32_t *pairs, const int32_t *cnt,
                             double bdist, double bcut, int32_t *n_group_out, int64_t *flt1, int64_t *flt2)
{
	int32_t i, j, max_s1 = 0, n_group = 0;
	int64_t k = 0;
	int32_t *tmp = CALLOC(int32_t, n);
	for (i = 0; i < n; ++i) max_s1 = max_s1 > c->br_s1[a0 + i] ? max_s1 : c->br_s1[a0 + i];
	for (i = 0; i < n; ++i) {
		double r = 1.0 - (double)c->br_s1[a0 + i] / max_s1; /* branch.c:71 */
		if (r > branch_diff) {
			int32_t n_local = 0;
			for (j = 0; j < n; ++j) {
				if (c->br_s1[a0 + j] != max_s1) continue; /* max_gid[], branch.c:66-68 */
				if (pairs) pairs[2*k] = c->br_gid[a0 + j], pairs[2*k+1] = c->br_gid[a0 + i];
				if (cnt) n_local += cnt[k];
				++k;
			}
			if (cnt) { /* branch.c:76-77 */
				if ((n_local == 0 && r > bdist) || r > bcut) c->br_weak[a0 + i] = 2, ++*flt2;
				else c->br_weak[a0 + i] = 1, ++*flt1;
			}
		}
	}
	for (i = 0; i < n; ++i) { /* branch.c:82-90: pg_n_local is evaluated before the tmp[j]==0 test */
		if (tmp[i] == 0) tmp[i] = ++n_group;
		for (j = i + 1; j < n; ++j) {
			if (pairs) pairs[2*k] = c->br_gid[a0 + i], pairs[2*k+1] = c->br_gid[a0 + j];
			if (cnt && cnt[k] > 0 && tmp[j] == 0) tmp[j] = tmp[i];
			++k;
		}
	}
	free(tmp);
	if (n_group_out) *n_group_out = n_group;
	return k;
}

/* the round's arc table: s1 (graph.c:171), target genes, out-degrees (graph.c:243-250) */
int pgo_sync(pga_ctx_t *c) { (void)c; return PGA_OK; }
int pgo_fetch_later(pga_ctx_t *c, const void *src, size_t nbytes, const void **host_view) { (void)c; (void)nbytes; *host_view = src; return PGA_OK; }

int pgo_arc_set_current(pga_ctx_t *c, const pga_arc_part_t *arcs, int64_t n_arc, int32_t n_seg, int32_t *deg)
{
	int64_t i;
	int32_t g, *seg_gid = CALLOC(int32_t, n_seg);
	for (g = 0; g < c->n_gene; ++g) if (c->g2s[g] >= 0 && c->g2s[g] < n_seg) seg_gid[c->g2s[g]] = g;
	free(c->br_x); free(c->br_s1); free(c->br_gid); free(c->br_weak);
	c->br_x = MALLOC(uint64_t, n_arc); c->br_s1 = MALLOC(int32_t, n_arc); c->br_gid = MALLOC(int32_t, n_arc); c->br_weak = CALLOC(uint8_t, n_arc);
	memset(deg, 0, 2 * (size_t)n_seg * sizeof(int32_t));
	for (i = 0; i < n_arc; ++i) {
		c->br_x[i] = arcs[i].x;
		c->br_s1[i] = (int32_t)((double)arcs[i].sum_s1 / arcs[i].n_genome + .499);
		c->br_gid[i] = seg_gid[(uint32_t)arcs[i].x >> 1];
		deg[arcs[i].x >> 32]++;
	}
	c->br_n = n_arc, c->br_S = n_seg;
	c->cur_tab = arcs, c->cur_tab_n = n_arc;
	free(seg_gid);
	return PGA_OK;
}

int pgo_arc_table(pga_ctx_t *c, const pga_arc_part_t **arcs, int64_t *n_arc) { *arcs = c->cur_tab, *n_arc = c->cur_tab_n; return PGA_OK; }

/* the deferred form (seg_cnt == NULL): the work is done at once, the results are handed over by arc_round_finish */
int pgo_arc_round_finish(pga_ctx_t *c, int32_t n_seg, int32_t *seg_cnt, int32_t *deg)
{
	if (c->def_sc == 0) return PGA_ERR_ARG;
	if (seg_cnt) { memcpy(seg_cnt, c->def_sc, 2 * (size_t)n_seg * sizeof(int32_t)); memcpy(deg, c->def_deg, 2 * (size_t)n_seg * sizeof(int32_t)); } /* NULL: status only */
	free(c->def_sc); free(c->def_deg); c->def_sc = c->def_deg = 0;
	return PGA_OK;
}

int pgo_arc_round_local(pga_ctx_t *c, int32_t use_ori, int32_t n_seg, int32_t *seg_cnt, int32_t *deg)
{
	if (seg_cnt == 0) {
		free(c->def_sc); free(c->def_deg);
		c->def_sc = CALLOC(int32_t, 2 * (int64_t)n_seg), c->def_deg = CALLOC(int32_t, 2 * (int64_t)n_seg);
		return pgo_arc_round_local(c, use_ori, n_seg, c->def_sc, c->def_deg);
	}
	int32_t *sc; pga_arc_part_t *arcs; int64_t n = 0, *n_arc = &n;
	int rc = pgo_arc_round(c, use_ori, &sc, &arcs, n_arc);
	if (rc != PGA_OK) return rc;
	if (n_seg != c->n_seg) return PGA_ERR_ARG;
	memcpy(seg_cnt, sc, 2 * (size_t)n_seg * sizeof(int32_t));
	return pgo_arc_set_current(c, arcs, *n_arc, n_seg, deg);
}

int pgo_branch_pairs(pga_ctx_t *c, const uint64_t *arc_x, const int32_t *arc_s1, int64_t n_arc, const int32_t *seg_gid, int32_t n_seg,
                     double branch_diff, int32_t local_dist, int32_t local_count, int32_t frag_mode, int32_t **cnt, int64_t *n_pairs)
{
	int64_t i, i0, np = 0, k;
	int pass, rc;
	if (arc_x) {
		free(c->br_x); free(c->br_s1); free(c->br_gid); free(c->br_weak);
		c->br_x = MALLOC(uint64_t, n_arc); c->br_s1 = MALLOC(int32_t, n_arc); c->br_gid = MALLOC(int32_t, n_arc); c->br_weak = CALLOC(uint8_t, n_arc);
		memcpy(c->br_x, arc_x, n_arc * sizeof(uint64_t)); memcpy(c->br_s1, arc_s1, n_arc * sizeof(int32_t));
		for (i = 0; i < n_arc; ++i) c->br_gid[i] = seg_gid[(uint32_t)arc_x[i] >> 1];
		c->br_n = n_arc, c->br_S = n_seg;
	} else { /* the table of arc_set_current */
		arc_x = c->br_x, n_arc = c->br_n, n_seg = c->br_S;
		memset(c->br_weak, 0, n_arc);
	}
	free(c->br_pairs);
	c->br_pairs = 0;
	for (pass = 0; pass < 2; ++pass) {
		k = 0;
		for (i0 = 0, i = 1; i <= n_arc; ++i)
			if (i == n_arc || arc_x[i] >> 32 != arc_x[i0] >> 32) {
				if (i - i0 >= 2) k += branch_vertex(c, i0, (int32_t)(i - i0), branch_diff, pass ? c->br_pairs + 2 * k : 0, 0, 0, 0, 0, 0, 0);
				i0 = i;
			}
		if (pass == 0) np = k, c->br_pairs = MALLOC(int32_t, 2 * np);
	}
	c->br_np = np;
	rc = pgo_n_local(c, c->br_pairs, np, local_dist, local_count, frag_mode, cnt);
	if (n_pairs) *n_pairs = np;
	return rc;
}

int pgo_branch_decide(pga_ctx_t *c, double branch_diff, double branch_diff_dist, double branch_diff_cut, uint8_t *arc_weak,
                      int32_t *n_dist_loci, int64_t *n_flt1, int64_t *n_flt2)
{
	int64_t i, i0, k = 0, f1 = 0, f2 = 0;
	memset(n_dist_loci, 0, 2 * (size_t)c->br_S * sizeof(int32_t));
	for (i0 = 0, i = 1; i <= c->br_n; ++i)
		if (i == c->br_n || c->br_x[i] >> 32 != c->br_x[i0] >> 32) {
			if (i - i0 >= 2) {
				int32_t ng;
				k += branch_vertex(c, i0, (int32_t)(i - i0), branch_diff, 0, c->nl_cnt + k, branch_diff_dist, branch_diff_cut, &ng, &f1, &f2);
				n_dist_loci[c->br_x[i0] >> 32] = ng;
			}
			i0 = i;
		}
	if (arc_weak) memcpy(arc_weak, c->br_weak, c->br_n);
	if (n_flt1) *n_flt1 = f1;
	if (n_flt2) *n_flt2 = f2;
	return PGA_OK;
}

/* branch_decide + the three tests of pg_flt_high_occ (graph.c:226-258) behind a deferred arc round */
int pgo_branch_decide_filter(pga_ctx_t *c, double branch_diff, double branch_diff_dist, double branch_diff_cut, int32_t do_filter,
                             int32_t max_tot_cnt, int32_t max_degree, int32_t max_dist_loci, uint8_t *del)
{
	int32_t s, S = c->br_S, *ndl;
	int rc;
	if (c->def_sc == 0 || (do_filter && del == 0)) return 2; /* no deferred round pending: the caller takes the classic calls */
	ndl = CALLOC(int32_t, 2 * (int64_t)S + 1);
	rc = pgo_branch_decide(c, branch_diff, branch_diff_dist, branch_diff_cut, 0, ndl, 0, 0);
	if (rc == PGA_OK && do_filter)
		for (s = 0; s < S; ++s) {
			int32_t m = ndl[2 * s] > ndl[2 * s + 1] ? ndl[2 * s] : ndl[2 * s + 1];
			del[s] = c->def_sc[S + s] > max_tot_cnt || c->def_deg[2 * s] > max_degree || c->def_deg[2 * s + 1] > max_degree || m > max_dist_loci;
		}
	free(ndl);
	return rc;
}

static int arc_weak(const uint64_t *ax, const uint8_t *aw, int64_t n, uint64_t x) /* pg_get_arc, pgpriv.h:99-107 */
{
	int64_t lo = 0, hi = n;
	while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (ax[mid] < x) lo = mid + 1; else hi = mid; }
	return lo < n && ax[lo] == x ? aw[lo] : 0;
}

/* pg_mark_branch_flt_hit, branch.c:108-145 */
int pgo_mark_hits(pga_ctx_t *c, const uint64_t *arc_x, const uint8_t *arc_w, int64_t n_arc, int64_t *n_marked, int32_t then_filter)
{
	int32_t j;
	int64_t k, n = 0;
	if (arc_x == 0) arc_x = c->br_x, arc_w = c->br_weak, n_arc = c->br_n; /* the arcs of the last branch_pairs/decide */
	for (j = 0; j < c->n_genome; ++j) {
		uint32_t v = (uint32_t)-1, w;
		int64_t vi = -1;
		for (k = c->off[j]; k < c->off[j + 1]; ++k) {
			int64_t a = c->yo[k];
			int32_t sid, e, cur;
			if (c->flags[a] & (PGA_F_FLT | PGA_F_SHADOW)) continue;
			sid = c->g2s[c->gid[a]];
			if (vi >= 0 && c->cid[a] != c->cid[vi]) v = (uint32_t)-1;
			if (vi >= 0 && c->cid[a] == c->cid[vi] && c->cm[a] == c->cm[vi]) { c->hz.h2_cm_tie++; hz_note(c, j, c->cid[a]); }
			w = (uint32_t)sid << 1 | (c->flags[a] & PGA_F_REV ? 1 : 0);
			if (v != (uint32_t)-1) {
				e = arc_weak(arc_x, arc_w, n_arc, (uint64_t)v << 32 | w);
				cur = weak_of(c, vi);
				if (e > cur) c->flags[vi] = (c->flags[vi] & ~PGA_F_WEAK_MASK) | (uint32_t)e << PGA_F_WEAK_SHIFT;
				e = arc_weak(arc_x, arc_w, n_arc, (uint64_t)(w^1) << 32 | (v^1));
				cur = weak_of(c, a);
				if (e > cur) c->flags[a] = (c->flags[a] & ~PGA_F_WEAK_MASK) | (uint32_t)e << PGA_F_WEAK_SHIFT;
			}
			v = w, vi = a;
		}
		for (k = c->off[j]; k < c->off[j + 1]; ++k)
			if (c->flags[k] & PGA_F_WEAK_MASK) ++n;
	}
	if (n_marked) *n_marked = n;
	return then_filter ? pgo_set_filter(c, PGA_FLT_WEAK2) : PGA_OK;
}

/* exact-order override (see pangene_hip.h): re-permute one contig segment of the X-ordered arrays, or
 * rewrite a slice of the Y order */
int pgo_override_order(pga_ctx_t *c, int32_t which, int32_t n_seg, const int32_t *seg_genome, const int32_t *seg_start,
                       const int64_t *seg_off, const int32_t *file_idx)
{
	int32_t s;
	for (s = 0; s < n_seg; ++s) {
		int32_t j = seg_genome[s];
		int64_t st = c->off[j], en = c->off[j + 1], p0 = st + seg_start[s], n = seg_off[s + 1] - seg_off[s], i, k;
		const int32_t *fl = file_idx + seg_off[s];
		int64_t *inv = MALLOC(int64_t, en - st); /* file index -> current X position */
		for (i = st; i < en; ++i) inv[c->fidx[i]] = i;
		if (which == 1) {
			for (k = 0; k < n; ++k) c->yo[p0 + k] = (int32_t)inv[fl[k]];
		} else {
			int32_t *remap = MALLOC(int32_t, en - st);
#define PERM_ARR(type, arr) do { \
				type *t_ = MALLOC(type, n); \
				for (k = 0; k < n; ++k) { t_[k] = c->arr[inv[fl[k]]]; } \
				for (k = 0; k < n; ++k) { c->arr[p0 + k] = t_[k]; } \
				free(t_); \
			} while (0)
			for (i = st; i < en; ++i) remap[i - st] = (int32_t)i;
			for (k = 0; k < n; ++k) remap[inv[fl[k]] - st] = (int32_t)(p0 + k);
			PERM_ARR(int32_t, pid); PERM_ARR(int32_t, gid); PERM_ARR(int32_t, cid); PERM_ARR(int32_t, rank);
			PERM_ARR(int32_t, score_ori); PERM_ARR(int32_t, score_adj); PERM_ARR(int32_t, score_dom); PERM_ARR(int32_t, n_exon_of);
			PERM_ARR(int32_t, off_exon); PERM_ARR(int64_t, cs); PERM_ARR(int64_t, ce); PERM_ARR(int64_t, cm); PERM_ARR(int32_t, cds);
			PERM_ARR(int32_t, pid_dom); PERM_ARR(int32_t, pid_dom0); PERM_ARR(uint32_t, flags);
			PERM_ARR(int32_t, fidx); /* last: inv[] was built from it */
#undef PERM_ARR
			for (i = st; i < en; ++i) c->yo[i] = remap[c->yo[i] - st];
			free(remap);
			if (p0 == st) c->head[j] = st; /* the genome's first contig now follows the exact order: array index 0 is its first hit */
		}
		free(inv);
	}
	return PGA_OK;
}

int pgo_set_head(pga_ctx_t *c, const int32_t *head_file)
{
	int32_t j;
	for (j = 0; j < c->n_genome; ++j) {
		int64_t i;
		c->head[j] = c->off[j];
		if (head_file[j] < 0) continue;
		for (i = c->off[j]; i < c->off[j + 1]; ++i)
			if (c->fidx[i] == head_file[j]) { c->head[j] = i; break; }
	}
	return PGA_OK;
}

int pgo_fetch(pga_ctx_t *c, void *dst, const void *src, size_t nbytes)
{
	(void)c;
	memcpy(dst, src, nbytes);
	return PGA_OK;
}

int pgo_put(pga_ctx_t *c, void *dst, const void *src, size_t nbytes) { (void)c; memcpy(dst, src, nbytes); return PGA_OK; }
int pgo_copy(pga_ctx_t *c, void *dst, const void *src, size_t nbytes) { (void)c; memmove(dst, src, nbytes); return PGA_OK; }
int pgo_scratch(pga_ctx_t *c, size_t nbytes, void **ptr)
{
	if (nbytes > c->m_scratch) { free(c->scratch); c->scratch = malloc(nbytes); c->m_scratch = nbytes; }
	*ptr = c->scratch;
	return c->scratch ? PGA_OK : PGA_ERR_NOMEM;
}

int pgo_download(pga_ctx_t *c, const pga_hit_state_t *o)
{
	int32_t j;
	int64_t i, k;
	if (o->flt_x_bits) {
		memset(o->flt_x_bits, 0, (size_t)((c->n_hit + 63) / 64) * sizeof(uint64_t));
		for (i = 0; i < c->n_hit; ++i)
			if (c->flags[i] & PGA_F_FLT) o->flt_x_bits[i >> 6] |= 1ULL << (i & 63);
	}
	for (j = 0; j < c->n_genome; ++j) {
		int64_t st = c->off[j];
		for (i = st; i < c->off[j + 1]; ++i) {
			int64_t f = st + c->fidx[i];
			if (o->flags) o->flags[f] = c->flags[i];
			if (o->rank) o->rank[f] = c->rank[i];
			if (o->score_dom) o->score_dom[f] = c->score_dom[i];
			if (o->pid_dom) o->pid_dom[f] = c->pid_dom[i];
			if (o->pid_dom0) o->pid_dom0[f] = c->pid_dom0[i];
			if (o->pos_x) o->pos_x[f] = (int32_t)(i - st);
		}
		if (o->pos_y)
			for (k = st; k < c->off[j + 1]; ++k)
				o->pos_y[st + c->fidx[c->yo[k]]] = (int32_t)(k - st);
	}
	return PGA_OK;
}

/* pangene.js gfa2matrix (pangene.js:1168-1247) over what the W-lines contain (format.c:183-225: the hits with flt == 0) */
int pgo_ctg_counts(pga_ctx_t *c, int32_t *cnt)
{
	int32_t j; int64_t i;
	memset(cnt, 0, (size_t)c->ctg_base[c->n_genome] * sizeof(int32_t));
	for (j = 0; j < c->n_genome; ++j)
		for (i = c->off[j]; i < c->off[j + 1]; ++i)
			if (!(c->flags[i] & PGA_F_FLT)) ++cnt[c->ctg_base[j] + c->cid[i]];
	return PGA_OK;
}

int pgo_gene_matrix(pga_ctx_t *c, const int32_t *asm_of_ctg, int32_t n_asm, int32_t n_seg, int32_t *mat)
{
	int32_t j; int64_t i;
	if (n_seg != c->n_seg) return PGA_ERR_ARG;
	memset(mat, 0, (size_t)n_seg * (size_t)n_asm * sizeof(int32_t));
	for (j = 0; j < c->n_genome; ++j)
		for (i = c->off[j]; i < c->off[j + 1]; ++i) {
			int32_t sid = c->g2s[c->gid[i]], col = asm_of_ctg[c->ctg_base[j] + c->cid[i]];
			if (!(c->flags[i] & PGA_F_FLT) && sid >= 0 && col >= 0) ++mat[(int64_t)sid * n_asm + col];
		}
	return PGA_OK;
}

int pgo_hazards(pga_ctx_t *c, pga_hazard_t *out) { *out = c->hz; return PGA_OK; }
int pgo_hazard_segs(pga_ctx_t *c, int32_t *segs, int32_t cap, int64_t *n_total)
{
	int64_t n = c->hz_n < PGA_HAZARD_CAP ? c->hz_n : PGA_HAZARD_CAP;
	if (n > cap) n = cap;
	if (n > 0) memcpy(segs, c->hz_seg, (size_t)n * sizeof(int32_t));
	*n_total = c->hz_n;
	return PGA_OK;
}

const pga_backend_t *pgo_backend(void)
{
	static const pga_backend_t b = {
		"oracle", pgo_create, pgo_destroy, pgo_begin, pgo_ingest, pgo_post_partials, pgo_post_apply, pgo_shadow, pgo_set_filter,
		pgo_vtx_partials, pgo_flag_vtx, pgo_arc_round, pgo_arc_merge, pgo_arc_set_current, pgo_rep_pos, pgo_n_local, pgo_branch_pairs, pgo_branch_decide, pgo_mark_hits, pgo_override_order, pgo_set_head, pgo_fetch, pgo_put, pgo_copy, pgo_scratch, pgo_download,
		pgo_hazards, pgo_is_device, pgo_strerror, 0, 0, pgo_sync, pgo_fetch_later, pgo_hazard_segs, pgo_host_alloc, pgo_host_free, pgo_arc_round_local, pgo_ctg_counts, pgo_gene_matrix, pgo_arc_table, pgo_arc_round_finish, pgo_branch_decide_filter
	};
	return &b;
}
