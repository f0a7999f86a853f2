/*
 * pangene_hip.h -- thin C ABI between the host side of the graph-construction path and the code that
 * owns the per-hit data ("backend").  Plain pointers and sizes only; no torch / C++ types.
 *
 * The reference (lh3/pangene v1.1-r231) has no FFI layer: its per-hit work is reached through
 * pg_read_paf()'s tail (read.c:243-260), pg_post_process() (graph.c:7-32) and pg_graph_gen()
 * (graph.c:280-322).  Each entry point below replaces the per-hit loops named in its comment; the
 * S/A-sized logic between them (vertex greedy, branch marking, pruning, formatting) stays on the host
 * (pangene_amd.h).
 *
 * Two implementations of this ABI exist:
 *   pga_*  (libpangene_amd.so)  hand-written HIP kernels for gfx950 -- the product; fails loudly
 *                               without a GPU, has no CPU fallback.
 *   pgo_*  (oracle/liboracle.so) plain-C restatement -- TEST INFRASTRUCTURE ONLY (checker).
 * Both export the same functions (prefix differs) and a vtable (pga_backend()/pgo_backend()).
 *
 * Memory spaces: pointers handed IN are host memory unless stated.  Pointers handed OUT through a
 * `**` argument live in the backend's space (HBM for pga_*, host for pgo_*) and stay valid until the
 * next call of the same function or pga_destroy(); they are what the exchange (all-reduce /
 * all-gather over RCCL) operates on.  Use fetch() to copy them to the host.
 *
 * Canonical order: X = hits sorted by (genome, contig, cs, file index);
 * Y = hits sorted by (genome, contig, cm, X position).  Both are computed once; keys never change.
 */
#ifndef PANGENE_HIP_H
#define PANGENE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* per-hit flag word: same bit positions as the bitfield at pangene.h:70 (rev:1 flt:1 ... weak_br:2) */
#define PGA_F_REV        0x001u
#define PGA_F_FLT        0x002u
#define PGA_F_ISO_SUB    0x004u  /* flt_iso_sub_self */
#define PGA_F_ISO_OV     0x008u  /* flt_iso_ov */
#define PGA_F_CHAIN      0x010u  /* flt_chain */
#define PGA_F_PSEUDO     0x020u
#define PGA_F_VTX        0x040u
#define PGA_F_SHADOW     0x080u
#define PGA_F_REP        0x100u
#define PGA_F_WEAK_SHIFT 9
#define PGA_F_WEAK_MASK  0x600u

/* which field PG_SET_FILTER (pgpriv.h:109-116) tests */
enum { PGA_FLT_PSEUDO = 0, PGA_FLT_VTX0 = 1, PGA_FLT_WEAK2 = 2, PGA_FLT_SHADOW = 3 };

/* status codes: 0 ok; negative = error (never aborts the process) */
enum {
	PGA_OK = 0,
	PGA_ERR_NO_DEVICE = -1,   /* no usable MI355X / HIP runtime error */
	PGA_ERR_RANGE = -2,       /* a coordinate or id does not fit the device layout */
	PGA_ERR_ARG = -3,
	PGA_ERR_NOMEM = -4,
	PGA_ERR_INVARIANT = -5    /* an invariant the reference asserts (e.g. vertex.c:38) was violated */
};

/* One genome of a shard, packed by the host right after its PAF has been parsed (pg_read_paf / pg_read_paf_batch), in FILE
 * order: what read.c:128-236 leaves in pg_genome_t::hit / ::exon (pangene.h:44-46,61-72,79-87), structure-of-arrays.
 * `data` is ONE buffer (pinned host memory when it came from host_alloc(), so the upload is a plain DMA):
 *   10 planes of n_hit int32:  pid, cid, rank, score_ori, score_adj, n_exon, off_exon (into THIS genome's exon list), cs, ce, cm
 *   n_hit bytes rev, padded to a multiple of 4
 *   n_exon pairs of int32 (os, oe), relative to cs, ascending (pangene.h:44-46)
 * The maxima let the backend size its sort keys without looking at the data on the host. */
#define PGA_BLOCK_PLANES 10
typedef struct {
	int32_t n_hit, n_exon, n_ctg;
	int32_t max_cs, max_cm, max_score_adj;   /* over the hits of the genome; 0 when it has none */
	int32_t any_neg_score_adj, any_multi_exon;
	const int32_t *data;
	size_t n_words;                          /* 10 * n_hit + (n_hit + 3) / 4 + 2 * n_exon */
	/* 64-bit contig coordinates (pangene.h:71: int64_t cs, cm, ce) through a 32-bit device layout: VIRTUAL CONTIGS.  A contig whose
	 * coordinates do not fit 31 bits is cut by the packer at hit-free gaps into pieces of < 2^30 bp plus one cluster of overlapping hits; the
	 * block then counts the pieces as contigs (n_ctg, the contig plane) and stores cs / ce / cm relative to the piece's base.  No hit
	 * spans a cut, so every overlap test, sort and filter works on the pieces unchanged; the steps that look ACROSS hits of one contig
	 * -- the walk of pg_gen_arc (graph.c:113-121: same-contig test, dist = cm - vpos, which the reference itself truncates to int32_t,
	 * graph.c:73) and pg_gen_rep_pos / pg_n_local (branch.c:6-46: same contig, 64-bit |cm1 - cm2|) -- put the pieces together again
	 * with these two tables.  vfirst[v] = the first piece of the contig piece v belongs to (the contig's identity; v itself for a contig
	 * that was not cut); vbase[v] = what was subtracted from the coordinates of piece v.  Both NULL: no contig of the genome was cut.
	 * The pieces of a contig are consecutive and in coordinate order (vfirst[v] <= v, vfirst[v] == vfirst[v - 1] or v, vbase not decreasing
	 * inside a contig: PGA_ERR_ARG otherwise), and every hit of piece v + 1 starts AFTER the last base of piece v (the packer's rule:
	 * graph_driver.cpp, virtual_contigs) -- the (contig, cs) and (contig, cm) orders of the pieces are then those of the contig. */
	const int32_t *vfirst;                   /* [n_ctg] or NULL */
	const int64_t *vbase;                    /* [n_ctg] or NULL */
} pga_genome_block_t;

/* One shard = a set of genomes with all their hits (pg_hit_t pangene.h:61-72, pg_exon_t 44-46, pg_genome_t 79-87).
 * Every hit is checked on the device against what its block declares (contig id < n_ctg, 0 <= cs <= ce, cs <= max_cs, cm <= max_cm,
 * score_adj <= max_score_adj or any_neg_score_adj, exon range inside the block's exon list, n_exon > 1 only with any_multi_exon):
 * pga_create answers PGA_ERR_RANGE for a block that breaks its own declaration instead of indexing out of bounds later.
 * Device layout limits (PGA_ERR_RANGE otherwise): coordinates inside a block < 2^31 (contigs beyond that arrive as virtual contigs,
 * see pga_genome_block_t; what remains out of reach is a single cluster of overlapping hits spanning 2^30 bp), < 2^30 hits and < 2^31
 * exons per shard, < 2^20 genes, < 2^24 genomes. */
#define PGA_ABI_VERSION 5u  /* bumped whenever a struct of this header or the order of pga_backend_t changes; pga_create refuses another */
typedef struct {
	uint32_t abi_version;        /* = PGA_ABI_VERSION of the header the caller was compiled against (PGA_ERR_ARG otherwise) */
	int32_t n_genome;            /* genomes in this shard (may include genomes with 0 hits) */
	int32_t n_genome_global;     /* G of the whole run (all shards) */
	const int32_t *genome_global;/* [n_genome] global genome index of each local genome */
	int32_t n_prot, n_gene;      /* global table sizes (ids are assigned on the host before upload) */
	int64_t n_hit, n_exon;       /* sums over the blocks */
	const pga_genome_block_t *block; /* [n_genome] */
	const int32_t *prot_gid;     /* [n_prot] */
	const uint8_t *gene_pref;    /* [n_gene] pg_gene_t::preferred */
} pga_shard_t;

typedef struct {
	double  min_ov_ratio;        /* pg_opt_t::min_ov_ratio (overlap.c:136) */
	int32_t check_strand;        /* PG_F_CHECK_STRAND */
	int32_t drop_sgl_exon;       /* PG_F_DROP_SGL_EXON (hit.c:180) -- evaluated on the host, kept for reference */
	int32_t reserved[4];
} pga_params_t;

/* per-hit state, FILE order, host memory (any pointer may be NULL = not wanted) */
typedef struct {
	uint32_t *flags;             /* PGA_F_* */
	int32_t *rank, *score_dom, *pid_dom, *pid_dom0;
	int32_t *pos_x;              /* position of the hit inside its genome in X (cs) order */
	int32_t *pos_y;              /* position of the hit inside its genome in Y (cm) order */
	uint64_t *flt_x_bits;        /* [(n_hit+63)/64] bit (hit_off[g] + pos_x) = flt of that hit: all the W-line writer needs */
} pga_hit_state_t;

/* one partially reduced arc: sums over the LOCAL genomes of the per-genome collapsed values
 * (graph.c:128-145 then the integer part of 153-169); final roundings are done on the host after
 * the cross-shard merge */
typedef struct {
	uint64_t x;                  /* v<<32|w, v = sid<<1|rev */
	int32_t  n_genome, tot_cnt;
	uint64_t sum_dist;           /* sum over genomes of dist_genome * n_genome_local_count */
	int64_t  sum_s1, sum_s2;     /* sum over genomes of per-genome max s1 / s2 */
} pga_arc_part_t;

typedef struct pga_ctx pga_ctx_t;

/* hazard counters: situations in which the reference's unstable sort could make its
 * output depend on tie order.  0 everywhere = canonical order provably gives the reference's result. */
typedef struct {
	int64_t h1_head_tie;         /* index-0 quirk outcome depends on who sits at index 0 */
	int64_t h2_cm_tie;           /* two walkable hits share (contig, cm) */
	int64_t h2_cs_tie;           /* walkable hits sharing (contig, cs) get pg_gen_rep_pos's counter r in tie order (branch.c:14,22-24): counts the
	                              * pg_n_local evaluations (branch.c:31-46) whose |r1 - r2| <= local_count test is not the same for every order of
	                              * the tie group(s) (interval form), and tie groups holding two walkable hits of one gene */
	int64_t h3_dom_tie;          /* dominator arg-max tie / subopt-isoform tie at equal (contig, cs) */
} pga_hazard_t;

#define PGA_HAZARD_CAP 4096

#define PGA_DECLARE(pfx) \
	/* copy a shard (file order) into backend memory; nothing is computed yet */ \
	int  pfx##_create(pga_ctx_t **ctx, const pga_shard_t *sh, const pga_params_t *par); \
	void pfx##_destroy(pga_ctx_t *ctx); \
	/* (re)start a run on the resident shard: per-hit constants (gene id, CDS length pg_cds_len \
	 * overlap.c:45-51, 64-bit score), X and Y orders (pg_hit_sort hit.c:29-64), all state reset to what \
	 * read.c:133-134 leaves after parsing.  May be called again to repeat the run without re-uploading. */ \
	int  pfx##_begin(pga_ctx_t *ctx); \
	/* stage A, read.c:243-260: pg_flag_pseudo, PG_SET_FILTER(pseudo), sort, pg_shadow(cal_dom_sc=1), \
	 * pid_dom0/reset, pg_flt_ov_isoform, pg_flt_chain_shadow, pg_flt_subopt_isoform. \
	 * stats: [n_genome*4] n_pseudo, n_flt_ov_iso, n_flt_chain, n_flt_subopt (host; may be NULL) */ \
	int  pfx##_ingest(pga_ctx_t *ctx, int32_t *stats); \
	/* stage B partials: max_ori[P] = max score_ori over all hits (hit.c:230-238); sums[6P] = \
	 * {sum score_adj, count} of rank-0 non-flt hits (hit.c:196-204) and c0,c1,s0,s1 (hit.c:158-169), \
	 * stored as six planes of P int64 */ \
	int  pfx##_post_partials(pga_ctx_t *ctx, int32_t **max_ori, int64_t **sums); \
	/* cap score_dom (hit.c:239-246, using the reduced max_ori still in backend memory), hit.rep \
	 * (hit.c:219-224), joint pseudo flag for single-exon hits of proteins with prot_pj[pid] (hit.c:170-184) */ \
	int  pfx##_post_apply(pga_ctx_t *ctx, const uint8_t *prot_rep, const uint8_t *prot_pj, int64_t *n_pseudo); \
	/* pg_shadow (overlap.c:101-178) over every genome. stats: [n_genome*2] {#non-flt, #shadowed} or NULL */ \
	int  pfx##_shadow(pga_ctx_t *ctx, int32_t cal_dom_sc, int32_t *stats); \
	/* PG_SET_FILTER (pgpriv.h:109-116) */ \
	int  pfx##_set_filter(pga_ctx_t *ctx, int32_t which); \
	/* pg_gen_vtx per-genome part (vertex.c:28-51): cnt[2Q] = n_dom[Q] then n_sub[Q]; records of \
	 * 1 + ceil(n_genome_global / 64) words: sub_gene<<20 | dom_gene, then the set (bit = global genome index) of this \
	 * shard's genomes in which sub_gene is sub-ordinate to dom_gene and dom_gene is dominant (the only cells the greedy \
	 * vertex.c:60-80 can observe).  A (sub, dom) key may occur in more than one record; their sets are disjoint. */ \
	int  pfx##_vtx_partials(pga_ctx_t *ctx, int32_t **cnt, uint64_t **records, int64_t *n_records); \
	/* pg_graph_flag_vtx (graph.c:61-69); g2s: [n_gene] host.  then_filter != 0: followed at once by PG_SET_FILTER(vtx == 0), \
	 * as everywhere in pg_graph_gen (graph.c:287-288,295,312), in the same pass over the hits */ \
	int  pfx##_flag_vtx(pga_ctx_t *ctx, const int32_t *g2s, int32_t n_seg, int32_t then_filter); \
	/* per-genome part of pg_gen_arc (graph.c:97-146) + local reduce-by-key of its global part. \
	 * seg_cnt[2S] = n_genome[S] then tot_cnt[S] (graph.c:125-126); arcs sorted by x */ \
	int  pfx##_arc_round(pga_ctx_t *ctx, int32_t use_ori, int32_t **seg_cnt, pga_arc_part_t **arcs, int64_t *n_arcs); \
	/* cross-shard reduce-by-key of the locally reduced arc tables after their all-gather: `gathered` (backend memory) \
	 * holds W slots of `slot` entries, count[r] (host) of them valid in slot r, each slot sorted by x.  Sums the \
	 * integer fields of equal x (graph.c:153-169 is a sum, so the order of the shards is irrelevant). */ \
	int  pfx##_arc_merge(pga_ctx_t *ctx, const pga_arc_part_t *gathered, const int64_t *count, int32_t W, int64_t slot, \
	                     pga_arc_part_t **out, int64_t *n_out); \
	/* Declare which arc table in backend memory (the output of arc_round, or of arc_merge when sharded) is the graph's \
	 * arc table of this round.  The backend derives what the following steps read -- per-arc s1 (the double rounding \
	 * of graph.c:171), target gene, the arc range of every oriented vertex -- and keeps it resident, so the arcs only \
	 * travel to the host once, after the last round: branch_pairs(arc_x = NULL), mark_hits(arc_x = NULL) use it. \
	 * deg (host, [2*n_seg]) receives the out-degree of every oriented vertex (pg_flt_high_occ, graph.c:243-250). */ \
	int  pfx##_arc_set_current(pga_ctx_t *ctx, const pga_arc_part_t *arcs, int64_t n_arc, int32_t n_seg, int32_t *deg); \
	/* pg_gen_arc (graph.c:87-177) of a run that is NOT sharded, in one call: the round's table becomes the graph's table at once \
	 * (as after arc_set_current), the host receives seg_cnt[2 * n_seg] (n_genome[S] then tot_cnt[S]), the out-degree of every \
	 * oriented vertex deg[2 * n_seg] after a single wait.  The table itself stays where the backend likes it (arc_table). \
	 * seg_cnt == NULL: nothing waits; the steps that read the table on the backend (rep_pos, branch_pairs, branch_decide) may be \
	 * queued behind it, and arc_round_finish -- to be called before anything that changes the hits (mark_hits, filters) -- \
	 * delivers the two arrays.  It returns 1 when the round has to be repeated (a plain arc_round_local call; whatever was \
	 * queued behind the first one has to be repeated after it as well). */ \
	int  pfx##_arc_round_local(pga_ctx_t *ctx, int32_t use_ori, int32_t n_seg, int32_t *seg_cnt, int32_t *deg); \
	int  pfx##_arc_round_finish(pga_ctx_t *ctx, int32_t n_seg, int32_t *seg_cnt, int32_t *deg); /* seg_cnt = deg = NULL: the status only */ \
	/* the graph's current arc table (of arc_round_local or arc_set_current) as ONE array sorted by x, in backend memory */ \
	int  pfx##_arc_table(pga_ctx_t *ctx, const pga_arc_part_t **arcs, int64_t *n_arc); \
	/* pg_gen_rep_pos (branch.c:6-29) kept in backend memory */ \
	int  pfx##_rep_pos(pga_ctx_t *ctx); \
	/* pg_n_local (branch.c:31-46) for n gene pairs (pairs[2i], pairs[2i+1]) over the local genomes.  The reference only tests the \
	 * result against zero (branch.c:76,86): cnt[i] > 0 iff the pair is local in some local genome; a backend may stop counting \
	 * at the first such genome (the HIP one does), so the exact value is unspecified beyond that */ \
	int  pfx##_n_local(pga_ctx_t *ctx, const int32_t *pairs, int64_t n, int32_t local_dist, int32_t local_count, \
	                   int32_t frag_mode, int32_t **cnt); \
	/* pg_mark_branch_flt_arc (branch.c:48-106), split in two so that the counts can be all-reduced in between. \
	 * branch_pairs: arcs sorted by x with their s1, seg_gid[S] = gene of each segment.  For every oriented vertex \
	 * with >= 2 arcs it lists the gene pairs the reference hands to pg_n_local -- the (best-scoring target, weaker \
	 * target) pairs of branch.c:70-75, then every i<j pair of 83-88 -- and counts each over the local genomes \
	 * (needs rep_pos): cnt[n_pairs] in backend memory.  arc_x = NULL: the table of arc_set_current.  branch_decide (after the all-reduce of cnt): branch.c:76-77 \
	 * and 82-90 -> weak_br per arc (arc_weak: host, n_arc bytes of the table of arc_set_current / branch_pairs(arc_x), or NULL), \
	 * n_dist_loci (host, 2S) and, when asked for (non-NULL), the numbers of arcs marked 1 and 2.  The arcs and their weak_br stay \
	 * resident: mark_hits(NULL, NULL, n_arc) uses them.  n_pairs may be NULL when nobody outside the backend needs the count \
	 * (a run that is not sharded): the backend then does not wait for it either. */ \
	int  pfx##_branch_pairs(pga_ctx_t *ctx, const uint64_t *arc_x, const int32_t *arc_s1, int64_t n_arc, const int32_t *seg_gid, int32_t n_seg, \
	                        double branch_diff, int32_t local_dist, int32_t local_count, int32_t frag_mode, int32_t **cnt, int64_t *n_pairs); \
	int  pfx##_branch_decide(pga_ctx_t *ctx, double branch_diff, double branch_diff_dist, double branch_diff_cut, uint8_t *arc_weak, \
	                         int32_t *n_dist_loci, int64_t *n_flt1, int64_t *n_flt2); \
	/* branch_decide for a run that is not sharded, behind an arc round whose host results have not been collected yet \
	 * (arc_round_local(seg_cnt = NULL)): n_dist_loci stays on the backend, and with do_filter != 0 the three tests of \
	 * pg_flt_high_occ (graph.c:226-258) are made there too -- del[s] = 1 (host, n_seg bytes) when segment s has \
	 * tot_cnt > max_tot_cnt, an oriented vertex with more than max_degree out-arcs, or more than max_dist_loci distant loci \
	 * on a side -- so that the round's counters, degrees and n_dist_loci (12 n_seg words) need not travel; the caller \
	 * then asks arc_round_finish(NULL, NULL) whether the round stands.  Waits.  Returns 2 when the preconditions do not \
	 * hold (the round took the sort path): the caller uses branch_decide + arc_round_finish + its own tests instead. */ \
	int  pfx##_branch_decide_filter(pga_ctx_t *ctx, double branch_diff, double branch_diff_dist, double branch_diff_cut, int32_t do_filter, \
	                                int32_t max_tot_cnt, int32_t max_degree, int32_t max_dist_loci, uint8_t *del); \
	/* pg_mark_branch_flt_hit (branch.c:108-145); arcs sorted by x with their weak_br (0 allowed).  then_filter != 0: followed at \
	 * once by PG_SET_FILTER(weak_br == 2) (graph.c:309), in the same pass over the hits */ \
	int  pfx##_mark_hits(pga_ctx_t *ctx, const uint64_t *arc_x, const uint8_t *arc_weak, int64_t n_arc, int64_t *n_marked, int32_t then_filter); \
	/* Replace the order inside contig segments by the exact order the reference's unstable radix sort \
	 * (ksort.h:52-87) leaves there (computed by the host).  which 0 = cs order (the \
	 * physical array order: index 0 of a genome, first-wins ties), 1 = cm order.  Segment s covers the \
	 * positions [seg_start[s], seg_start[s] + seg_off[s+1]-seg_off[s]) of local genome seg_genome[s]; \
	 * file_idx lists the hits to put there by their FILE index inside the genome. */ \
	int  pfx##_override_order(pga_ctx_t *ctx, int32_t which, int32_t n_seg, const int32_t *seg_genome, const int32_t *seg_start, \
	                          const int64_t *seg_off, const int32_t *file_idx); \
	/* Index-0 channel only (cheap form of override_order): head_file[g] = FILE index of the hit the \
	 * reference has at array index 0 of local genome g (the one pg_shadow never resets, overlap.c:108); \
	 * -1 = the first hit of the canonical order */ \
	int  pfx##_set_head(pga_ctx_t *ctx, const int32_t *head_file); \
	/* copy backend memory to the host / host to backend / backend to backend */ \
	int  pfx##_fetch(pga_ctx_t *ctx, void *dst_host, const void *src_backend, size_t nbytes); \
	int  pfx##_put(pga_ctx_t *ctx, void *dst_backend, const void *src_host, size_t nbytes); \
	int  pfx##_copy(pga_ctx_t *ctx, void *dst_backend, const void *src_backend, size_t nbytes); \
	/* one reusable scratch buffer in backend memory (staging area for padded all-gathers); a second \
	 * call invalidates the first pointer */ \
	int  pfx##_scratch(pga_ctx_t *ctx, size_t nbytes, void **ptr); \
	/* wait until everything issued so far has finished (only needed before handing backend memory to code that does not \
	 * run on the backend's stream; fetch and the calls that return host values synchronise by themselves) */ \
	int  pfx##_sync(pga_ctx_t *ctx); \
	/* fetch without waiting: the bytes are copied to host memory owned by the backend, *host_view points at them and is \
	 * valid from the next synchronising call (fetch, sync, arc_set_current, ...) until the next fetch_later */ \
	int  pfx##_fetch_later(pga_ctx_t *ctx, const void *src_backend, size_t nbytes, const void **host_view); \
	/* pangene.js gfa2matrix (pangene.js:1168-1247) as a reduction over what the W-lines would contain.  ctg_counts: the number \
	 * of hits that survive (flt == 0) on every contig of the shard (genome-major list of contigs, host array of that length): \
	 * a contig prints a W-line iff it has one (format.c:209).  gene_matrix: asm_of_ctg[contig] = column of the contig's \
	 * sample#haplotype (-1: none); mat[n_seg * n_asm] (host) receives, per segment and column, the number of surviving hits of \
	 * the segment's gene on the column's contigs, i.e. the occurrences of the segment in that assembly's walks */ \
	int  pfx##_ctg_counts(pga_ctx_t *ctx, int32_t *cnt); \
	int  pfx##_gene_matrix(pga_ctx_t *ctx, const int32_t *asm_of_ctg, int32_t n_asm, int32_t n_seg, int32_t *mat); \
	/* per-hit state in file order */ \
	int  pfx##_download(pga_ctx_t *ctx, const pga_hit_state_t *out); \
	int  pfx##_hazards(pga_ctx_t *ctx, pga_hazard_t *out); \
	/* where the h2_cm_tie / h2_cs_tie / h3_dom_tie events of the run happened: up to cap contig-segment ids (position of the contig in \
	 * the shard's genome-major list of contigs) in segs, one per event, duplicates possible; *n_total = number of events \
	 * (the backend keeps at most PGA_HAZARD_CAP of them: n_total larger than what was returned = list incomplete) */ \
	int  pfx##_hazard_segs(pga_ctx_t *ctx, int32_t *segs, int32_t cap, int64_t *n_total); \
	/* 1 if `**` pointers are device memory (exchange must use the device collective) */ \
	int  pfx##_is_device(void); \
	/* host memory the backend can upload from without staging (hipHostMalloc for pga_*, malloc for pgo_*); usable before \
	 * any context exists: the PAF reader packs every genome into such a buffer as soon as it has been parsed */ \
	int  pfx##_host_alloc(size_t nbytes, void **ptr); \
	void pfx##_host_free(void *ptr); \
	const char *pfx##_strerror(int code);

PGA_DECLARE(pga)

/* the same ABI as a table, so the host driver is written once */
struct pga_branch_par_s; struct pga_loop_xchg_s;
typedef struct {
	const char *name;
	int  (*create)(pga_ctx_t **, const pga_shard_t *, const pga_params_t *);
	void (*destroy)(pga_ctx_t *);
	int  (*begin)(pga_ctx_t *);
	int  (*ingest)(pga_ctx_t *, int32_t *);
	int  (*post_partials)(pga_ctx_t *, int32_t **, int64_t **);
	int  (*post_apply)(pga_ctx_t *, const uint8_t *, const uint8_t *, int64_t *);
	int  (*shadow)(pga_ctx_t *, int32_t, int32_t *);
	int  (*set_filter)(pga_ctx_t *, int32_t);
	int  (*vtx_partials)(pga_ctx_t *, int32_t **, uint64_t **, int64_t *);
	int  (*flag_vtx)(pga_ctx_t *, const int32_t *, int32_t, int32_t);
	int  (*arc_round)(pga_ctx_t *, int32_t, int32_t **, pga_arc_part_t **, int64_t *);
	int  (*arc_merge)(pga_ctx_t *, const pga_arc_part_t *, const int64_t *, int32_t, int64_t, pga_arc_part_t **, int64_t *);
	int  (*arc_set_current)(pga_ctx_t *, const pga_arc_part_t *, int64_t, int32_t, int32_t *);
	int  (*rep_pos)(pga_ctx_t *);
	int  (*n_local)(pga_ctx_t *, const int32_t *, int64_t, int32_t, int32_t, int32_t, int32_t **);
	int  (*branch_pairs)(pga_ctx_t *, const uint64_t *, const int32_t *, int64_t, const int32_t *, int32_t, double, int32_t, int32_t, int32_t, int32_t **, int64_t *);
	int  (*branch_decide)(pga_ctx_t *, double, double, double, uint8_t *, int32_t *, int64_t *, int64_t *);
	int  (*mark_hits)(pga_ctx_t *, const uint64_t *, const uint8_t *, int64_t, int64_t *, int32_t);
	int  (*override_order)(pga_ctx_t *, int32_t, int32_t, const int32_t *, const int32_t *, const int64_t *, const int32_t *);
	int  (*set_head)(pga_ctx_t *, const int32_t *);
	int  (*fetch)(pga_ctx_t *, void *, const void *, size_t);
	int  (*put)(pga_ctx_t *, void *, const void *, size_t);
	int  (*copy)(pga_ctx_t *, void *, const void *, size_t);
	int  (*scratch)(pga_ctx_t *, size_t, void **);
	int  (*download)(pga_ctx_t *, const pga_hit_state_t *);
	int  (*hazards)(pga_ctx_t *, pga_hazard_t *);
	int  (*is_device)(void);
	const char *(*strerror)(int);
	int  (*timing_reset)(pga_ctx_t *);  /* may be NULL */
	int  (*timing_get)(pga_ctx_t *, int32_t, double *, int64_t *, int64_t *);
	int  (*sync)(pga_ctx_t *);
	int  (*fetch_later)(pga_ctx_t *, const void *, size_t, const void **);
	int  (*hazard_segs)(pga_ctx_t *, int32_t *, int32_t, int64_t *);
	int  (*host_alloc)(size_t, void **);
	void (*host_free)(void *);
	int  (*arc_round_local)(pga_ctx_t *, int32_t, int32_t, int32_t *, int32_t *);
	int  (*ctg_counts)(pga_ctx_t *, int32_t *);
	int  (*gene_matrix)(pga_ctx_t *, const int32_t *, int32_t, int32_t, int32_t *);
	int  (*arc_table)(pga_ctx_t *, const pga_arc_part_t **, int64_t *);
	int  (*arc_round_finish)(pga_ctx_t *, int32_t, int32_t *, int32_t *);
	int  (*branch_decide_filter)(pga_ctx_t *, double, double, double, int32_t, int32_t, int32_t, int32_t, uint8_t *); /* may be NULL */
	int  (*branch_loop)(pga_ctx_t *, int32_t, const struct pga_branch_par_s *, const int32_t *, const int32_t *, const int32_t *, uint8_t *, const struct pga_loop_xchg_s *, int32_t *, int32_t *); /* may be NULL */
	void (*host_trim)(size_t); /* may be NULL */
	int  (*set_device)(int32_t); /* may be NULL */
	int  (*device_count)(void);  /* may be NULL */
	int  (*arc_round_x)(pga_ctx_t *, int32_t, int32_t, const struct pga_loop_xchg_s *, int32_t *, int32_t *, int64_t *); /* may be NULL */
	int  (*copy_gbps)(size_t, int32_t, double *); /* may be NULL */
	int  (*warm)(void); /* may be NULL */
	int  (*reserve)(int64_t, int64_t, int32_t, int32_t, int32_t, int64_t); /* may be NULL */
	int  (*stage)(const void *, size_t); /* may be NULL */
	void (*stage_drop)(const void *);    /* may be NULL */
} pga_backend_t;

const pga_backend_t *pga_backend(void);

/* The branch-filter rounds of pg_graph_gen (graph.c:300-314) without the host in between.  Behind a deferred arc round of a run
 * that is not sharded (arc_round_local(seg_cnt = NULL)), rounds r = 0 .. n_round-1 are queued back to back:
 *   pg_mark_branch_flt_arc (rep_pos, branch_pairs, branch_decide), pg_mark_branch_flt_hit + PG_SET_FILTER(weak_br == 2),
 *   for r > 0 pg_flt_high_occ's three tests with max_tot_cnt[r] / max_degree[r] / max_dist_loci[r] (graph.c:226-258) and
 *   PG_SET_FILTER(vtx == 0), and -- except after the last round -- the next pg_gen_arc.
 * A deleted segment is not renumbered away: its gene loses its vertex (g2s = -1), its two vertices keep their numbers and
 * simply have no arcs from then on (renumbering is monotone, so the order of every arc list, pair list and group number --
 * all the rounds look at -- is the same with the holes as without).  seg_alive[n_seg] (host) tells the caller which
 * segments are left; it renumbers them, hands the new g2s over (flag_vtx) and runs the arc round the loop left out.
 * ONE wait, at the end.  Returns 0; 1 = something the queued rounds could not handle happened on the way (a hub gene
 * overflowed its LDS table, the pair list its capacity): the shard's state is then undefined and the caller repeats the
 * run with host-driven rounds; 2 = not applicable here (nothing was queued); 3 (sharded form only) = an exchange buffer
 * was too small: state undefined as with 1, but the capacities have been raised and a later run can queue its rounds again.
 * 4 (unsharded form) = a pair list was longer than the room the loop had reserved, and nothing else went wrong: state undefined as with 1,
 * the room has been made (up to 2^27 pairs) and the caller repeats the run WITH queued rounds.
 *
 * Sharded form (x != NULL): behind pga_arc_set_current (the merged table of a host-driven round), every rank queues the same
 * rounds; where the host-driven route exchanges (graph_driver.cpp gen_arc / mark_branch_flt_arc) the backend calls x's two
 * collectives between its kernels -- per round ONE all-gather (every rank's slot: arc count, segment counters, arc table; the
 * tables are merged by key on the device, counts never leave it) and ONE all-reduce (the n_local counts of the round's pair
 * list, at a capacity all ranks share), plus one small all-reduce at the end that makes the return value collective.  A
 * callback returns when its collective is ORDERED with the context's stream: enqueued on it (RCCL on pga_active_stream), or
 * completed after the callee waited for the stream itself.  Every rank must call with the same n_round, capacities follow
 * from values all ranks share (arc_cap_hint: the largest local arc table of the last host-driven round, over all ranks). */
typedef struct pga_loop_xchg_s {
	void *user;
	int32_t rank, world;
	int64_t arc_cap_hint;
	int (*allreduce_i32_sum)(void *user, void *buf, int64_t count);                  /* in place, device memory */
	int (*allgather)(void *user, const void *in, void *out, int64_t bytes_per_rank); /* out: world slots, rank order */
} pga_loop_xchg_t;
/* Page-locked host memory the backend keeps between contexts (mailboxes, staging areas): give back what exceeds keep_bytes. */
void pga_host_trim(size_t keep_bytes);
/* the HIP device of this process (before the first context); the number of visible devices */
int pga_set_device(int32_t device);
int pga_device_count(void);
/* Measurement support (SURVEY.md 8d asks for a copy kernel beside the spec figure): the bandwidth, GB/s of read + write, that a
 * 16-byte-per-lane copy of `bytes` reaches on the current device; best of `reps`. */
int pga_copy_gbps(size_t bytes, int32_t reps, double *gbps);
/* Bring the HIP runtime up and load this library's code object (a first launch does both, ~0.1-0.3 s): a command line calls it on a
 * helper thread while its PAF files are parsed, so that the first pga_create finds the device ready. */
int pga_warm(void);

typedef struct pga_branch_par_s {
	double branch_diff, branch_diff_dist, branch_diff_cut; int32_t local_dist, local_count, frag_mode, use_ori;
	/* pre_on != 0 (unsharded form only): the deferred arc round behind which the loop is called is graph 1's (graph.c:291), and the loop
	 * starts with graph 2's pg_flt_high_occ (graph.c:294-295: its tests with these three limits, n_dist_loci still 0) + the next
	 * pg_gen_arc (graph.c:296) before round 0 -- deletions as holes, like those of the rounds */
	int32_t pre_on, pre_max_tot_cnt, pre_max_degree, pre_max_dist_loci;
	/* final_on != 0 (unsharded form only): n_round = ALL the rounds, and the arc round behind the last one (graph.c:313 of round
	 * n-1, the graph that is written) is queued as well.  seg_cnt[2 n_seg] (graph.c:125-126 of that round) and n_dist_loci[2 n_seg]
	 * (branch.c:90 of the last branch step) come back with seg_alive, all in the numbering the loop was entered with; the table
	 * (arc_table) is in that numbering too: the caller renumbers segments and arcs (monotone, so every order stands). */
	int32_t final_on;
} pga_branch_par_t;
/* pg_gen_arc (graph.c:87-177) of a sharded run with ONE wait: arc_round + the exchange + arc_merge + arc_set_current, every table
 * size left on the device; the ranks' tables travel in slots of a capacity all ranks share (the largest local table the shard has
 * seen, with a margin; x->arc_cap_hint before the backend has seen one).  seg_cnt[2 n_seg], deg[2 n_seg] (host) and *n_arc are the
 * global results; the merged table is the current one (arc_table).  Returns 1 on EVERY rank when the round is void on some rank (a
 * hub gene beyond its LDS table, a table beyond its slot): the caller repeats the round through arc_round (which then takes the sort
 * path, without a second sweep) and the host-driven exchange; 2 = not applicable (no capacity known, n_seg = 0): nothing happened. */
int pga_arc_round_x(pga_ctx_t *ctx, int32_t use_ori, int32_t n_seg, const pga_loop_xchg_t *x, int32_t *seg_cnt, int32_t *deg, int64_t *n_arc);
int pga_branch_loop(pga_ctx_t *ctx, int32_t n_round, const pga_branch_par_t *par, const int32_t *max_tot_cnt, const int32_t *max_degree,
                    const int32_t *max_dist_loci, uint8_t *seg_alive, const pga_loop_xchg_t *x, int32_t *seg_cnt /* final_on: [2 n_seg], else NULL */, int32_t *n_dist_loci /* likewise */);

/* Optional: run every kernel on this hipStream_t instead of the library's own stream (lets a host
 * framework order its collectives with the kernels without extra synchronisation). */
int pga_set_stream(pga_ctx_t *ctx, void *hip_stream);
/* The hipStream_t the live context runs on (collectives of a sharded run are enqueued on it). */
void *pga_active_stream(void);

/* Kernel timing hooks for bench.py: HIP events bracket every launch of the named kernel class on the
 * library's stream, from the first pga_timing_reset on.  which: 0 = "k1" (stage A sweep, the hit-filter+overlap kernel),
 * 1 = the pg_flt_ov_isoform sweep, 2 = the other pg_shadow sweeps, 3 = the whole of stage A (pga_begin + pga_ingest), 4 = host waits
 * (count only); with PANGENE_TIME_ROUNDS=1 in the environment at pga_timing_reset also 5 = every pg_gen_arc round (graph.c:87-177: sweep, walk,
 * temp arcs, collapse -- two more events per round, so for a pass that is not itself timed) and 6 = its walk scan alone.
 * which | (k + 1) << 8 selects the k-th timed launch of the class alone.  7 is not a timing: which build of K1 the sweeps of stage A and
 * pg_post_process run on this upload -- n_launch = 1 k_sweep (the tile's exon lists staged in LDS), 0 k_sweep_lean (read where they are),
 * -1 not decided yet; total_ms = the exons a tile would stage, as sampled (units = tiles sampled; 0 when PANGENE_SWEEP_LISTS fixed the choice). */
/* Optional, before pga_create: the device memory a shard of about this size will ask for (hits, exons, proteins, genes, genomes, words of
 * packed blocks), allocated now and kept for the pga_create that follows -- hipMalloc of tens of gigabytes takes seconds, and a reader
 * that knows its files' sizes can have it done while it parses (pg_read_paf_batch does).  Too small an estimate costs nothing but the
 * attempt; blocks nobody claims go back with pga_host_trim(0).  Safe to call from any thread. */
int pga_reserve(int64_t n_hit, int64_t n_exon, int32_t n_prot, int32_t n_gene, int32_t n_genome, int64_t raw_words);
/* Blocks on their way before there is a context (round 6; SURVEY 8(d): "device upload included").  The reader packs the genomes into slabs of page-locked host
 * memory (host_alloc) while later files are still parsed; a slab that is full and completely written is handed over here: its first `bytes` bytes are copied to a
 * device buffer of the library's on a stream of its own, and nothing waits.  pga_create() takes the blocks it finds inside such a slab from that copy (device to
 * device, behind the copy's event) instead of across the host link again.  The caller promises not to write to [host, host + bytes) until pga_stage_drop(host),
 * which it calls before the slab is reused or freed (it waits for a copy still in flight).  Safe to call from any thread; 0 or an error code -- a slab that was
 * not staged is simply uploaded by pga_create() as before. */
int pga_stage_h2d(const void *host, size_t bytes);
void pga_stage_drop(const void *host);
int pga_timing_reset(pga_ctx_t *ctx);
int pga_timing_get(pga_ctx_t *ctx, int32_t which, double *total_ms, int64_t *n_launch, int64_t *units);

#ifdef __cplusplus
}
#endif
#endif
