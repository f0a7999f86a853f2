/*
 * pangene_amd.h -- public C surface of the MI355X-native pangene graph-construction path.
 *
 * Drop-in boundary.  The reference has no plugin/FFI layer; its boundary *is* pangene.h:126-141, called
 * by main.c:117-142 in a fixed order.  This header re-declares that surface with identical struct
 * layouts (sizes checked by tests/test_abi.py: opt 128, hit 88, exon 8, prot 32, gene 16, ctg 16,
 * genome 56, data 72, seg 32, arc 32, graph 56) and identical function names and argument meaning, so a
 * program written against pangene.h links against libpangene_amd.so unchanged.
 *
 *   entry point            replaces (reference file:line)
 *   pg_opt_init            option.c:6-26
 *   pg_data_init/destroy   read.c:10-32
 *   pg_read_paf            read.c:107-262   (parse only; the per-genome filters of read.c:243-260 are
 *                                            deferred to pg_post_process and run on the GPU -- exact,
 *                                            SURVEY.md 9.5; nothing can observe the difference because
 *                                            main.c calls pg_post_process next)
 *   pg_post_process        graph.c:7-32     (+ the deferred read.c:243-260) -> HIP kernels.  The hits are taken as pg_read_paf
 *                                            left them (each genome is packed for the device while the next file is parsed);
 *                                            a genome edited in between is packed again: the pack carries a signature over
 *                                            every field of every hit and exon it was made from, checked by pg_post_process
 *   pg_graph_init/gen/destroy graph.c:34-47, 280-322 -> HIP kernels + host round driver
 *   pg_write_bed/graph/walk format.c:113-225
 *   pg_read_list_dict, pg_dict_destroy  read.c:305-318, dict.c:38-49 (main.c:73-75,140-142 need them)
 *   pg_realtime/cputime/peakrss sys.c:117-140   (main.c:117,149 need them)
 *
 * Additions (not in the reference) are at the end: the exchange hook used to shard genomes across
 * GPUs/processes, id-only scanning of PAFs owned by another shard, and error reporting.
 */
#ifndef PANGENE_AMD_H
#define PANGENE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PG_VERSION "1.1-r231-mi355x"

/* pg_opt_t::flag bits (values as in pangene.h:8-17) */
#define PG_F_WRITE_BED_RAW   0x1
#define PG_F_WRITE_BED_WALK  0x2
#define PG_F_WRITE_BED_FLAG  0x4
#define PG_F_WRITE_NO_WALK   0x8
#define PG_F_WRITE_VTX_SEL   0x10
#define PG_F_FRAG_MODE       0x20
#define PG_F_NO_JOINT_PSEUDO 0x40
#define PG_F_ORI_FOR_BRANCH  0x80
#define PG_F_CHECK_STRAND    0x100
#define PG_F_DROP_SGL_EXON   0x200

typedef struct { uint64_t x, y; } pg128_t;

/* options; layout of pangene.h:23-42 */
typedef struct {
	uint32_t flag;
	int32_t  gene_delim;          /* -d */
	double   min_prot_ratio;      /* -l */
	double   min_prot_iden;       /* -e */
	double   score_adj_coef;      /* -m */
	double   min_ov_ratio;        /* -f */
	double   min_vertex_ratio;    /* -p */
	double   branch_diff;         /* -b */
	double   branch_diff_dist;    /* -y */
	double   branch_diff_cut;     /* -B */
	int32_t  max_avg_occ;         /* -c */
	int32_t  max_degree;          /* -g */
	int32_t  max_dist_loci;       /* -r */
	int32_t  n_branch_flt;        /* -T */
	int32_t  min_arc_cnt;         /* -a */
	int32_t  local_dist;          /* -D */
	int32_t  local_count;         /* -C */
	void    *excl, *incl, *preferred; /* opaque name sets from pg_read_list_dict (-X -I -P) */
} pg_opt_t;

typedef struct { int32_t os, oe; } pg_exon_t;       /* exon [os,oe) relative to pg_hit_t::cs */

typedef struct {
	const char *name;
	int32_t len, gid;
	int32_t rep;
	int32_t n, avg_score_adj, max_score_ori;
} pg_prot_t;

typedef struct {
	const char *name;
	uint32_t len:30, preferred:1, included:1;
	int32_t rep_pid;
} pg_gene_t;

typedef struct {                                     /* 88 bytes; cs@64 cm@72 ce@80 */
	int32_t pid;
	int32_t qs, qe;
	int32_t cid;
	int32_t mlen, blen, lof;
	int32_t rank;
	int32_t score_ori, score_adj, score_dom;
	int32_t n_exon, off_exon;
	int32_t pid_dom, pid_dom0;
	uint32_t rev:1, flt:1, flt_iso_sub_self:1, flt_iso_ov:1, flt_chain:1, pseudo:1, vtx:1, shadow:1, rep:1, weak_br:2;
	int64_t cs, cm, ce;
} pg_hit_t;

typedef struct { const char *name; int64_t len; } pg_ctg_t;

typedef struct {
	int32_t n_ctg, m_ctg;   pg_ctg_t *ctg;
	int32_t n_hit, m_hit;   pg_hit_t *hit;
	int32_t n_exon, m_exon; pg_exon_t *exon;
	char *label;
} pg_genome_t;

typedef struct {
	void *d_ctg, *d_gene, *d_prot;
	int32_t n_genome, m_genome; pg_genome_t *genome;
	int32_t n_gene, m_gene;     pg_gene_t *gene;
	int32_t n_prot, m_prot;     pg_prot_t *prot;
} pg_data_t;

typedef struct {
	int32_t gid, n_dom, n_sub;
	int32_t n_genome;
	int32_t tot_cnt;
	uint32_t del:1, dummy:31;
	int32_t n_dist_loci[2];
} pg_seg_t;

typedef struct {
	uint64_t x;                  /* v<<32|w with v = segment<<1|strand */
	int32_t n_genome;
	int32_t tot_cnt;
	int32_t avg_dist;
	int32_t s1, s2;
	uint32_t del:1, weak_br:2, dummy:29;
} pg_arc_t;

typedef struct {
	pg_data_t *d;
	int32_t *g2s;
	int32_t n_seg, m_seg; pg_seg_t *seg;
	int32_t n_arc, m_arc; pg_arc_t *arc;
	uint64_t *idx;
} pg_graph_t;

extern int pg_verbose;

void       pg_opt_init(pg_opt_t *opt);
pg_data_t *pg_data_init(void);
void       pg_data_destroy(pg_data_t *d);
int32_t    pg_read_paf(const pg_opt_t *opt, pg_data_t *d, const char *fn);   /* -1 if fn cannot be opened */
void       pg_post_process(const pg_opt_t *opt, pg_data_t *d);
pg_graph_t *pg_graph_init(pg_data_t *d);
void       pg_graph_gen(const pg_opt_t *opt, pg_graph_t *q);
void       pg_graph_destroy(pg_graph_t *g);
void       pg_write_bed(const pg_data_t *d, int32_t is_walk);
void       pg_write_graph(const pg_graph_t *g);
void       pg_write_walk(pg_graph_t *g);
void      *pg_read_list_dict(const char *o);
void       pg_dict_destroy(void *h);
/* private in the reference too (pgpriv.h, sys.c:117-140) but called by main.c:117,149 */
double     pg_realtime(void);   /* seconds since the first call */
double     pg_cputime(void);    /* user + system CPU seconds */
long       pg_peakrss(void);    /* bytes */

/* ---------------------------------------------------------------------------------------------
 * Additions
 * ------------------------------------------------------------------------------------------- */

/* pangene.js `gfa2matrix` (pangene.js:1168-1247), the gene x assembly presence / copy-number matrix:
 * pg_write_matrix prints it for the graph in memory (after pg_graph_gen; the per-hit reduction runs on the GPU),
 * pg_gfa2matrix_file for any GFA file, plain or gzipped, exactly as the script does (clstr_fn: optional CD-HIT cluster
 * file, option -d there; print_cd: its -p).  Output: "Gene<TAB>asm..." then one row per segment. */
void pg_write_matrix(pg_graph_t *g, int32_t copy_number);
int  pg_gfa2matrix_file(const char *gfa_fn, int32_t copy_number, const char *clstr_fn, int32_t print_cd);

/* Last error of the path (0 = none).  The reference aborts on invariant violations; this library
 * records a status instead, prints one line to stderr, and leaves the graph empty. */
int         pg_last_error(void);
const char *pg_last_error_str(void);

/* Redirect what the writers would print to stdout into a file descriptor-less sink:
 * pg_set_output(path) makes pg_write_* append to `path` (NULL restores stdout). */
int pg_set_output(const char *path);

/* Register the gene/protein names (and lengths) of a PAF whose hits belong to ANOTHER shard, so that
 * first-seen id numbering (read.c:151-168) is identical on every rank.  Adds an empty genome. */
int32_t pg_scan_paf_ids(const pg_opt_t *opt, pg_data_t *d, const char *fn);

/* Parse n PAFs on host threads (n_threads <= 0: as many as the process may really use -- the affinity mask capped by the control
 * group's CPU quota -- and at most 64) and append them in the given order; gene / protein / contig ids and every attribute of every
 * gene and protein come out exactly as after n sequential pg_read_paf calls (tests/test_reader.py).  The hit / exon arrays of plain
 * (not gzipped) files lie in one huge-page mapping owned by `d`: pg_data_destroy releases it, nobody else may free() them.  ids_only (may be NULL): per file, non-zero = register
 * the names only, as pg_scan_paf_ids does.  Returns minus the number of files that could not be opened. */
int32_t pg_read_paf_batch(const pg_opt_t *opt, pg_data_t *d, int32_t n, const char *const *fns, const uint8_t *ids_only, int32_t n_threads);

/* Exchange hook: genomes shard across processes (one per GPU); the only communication is a handful
 * of small integer reductions / gathers per round.  NULL (default) = single process. */
enum { PG_X_I32 = 0, PG_X_I64 = 1 };
enum { PG_X_SUM = 0, PG_X_MAX = 1 };
typedef struct {
	int32_t rank, world;
	void *user;
	/* in-place all-reduce of `count` elements; is_device: buf is HBM (use RCCL) else host memory */
	int (*allreduce)(void *user, void *buf, int64_t count, int32_t dtype, int32_t op, int32_t is_device);
	/* all-gather `nbytes` from every rank into out[world*nbytes] */
	int (*allgather)(void *user, const void *in, void *out, int64_t nbytes, int32_t is_device);
	/* non-zero: the callbacks enqueue on the backend's own stream (pga_active_stream), so the library need not wait for
	 * its kernels before calling them; zero: it synchronises first */
	int32_t stream_ordered;
} pg_exchange_t;
void pg_set_exchange(const pg_exchange_t *x);

/* Built-in exchange for the HIP backend: RCCL collectives enqueued on the kernels' own stream (no host round trip).
 * Rank 0 obtains the 128-byte bootstrap id, the launcher hands it to every rank, every rank calls pg_rccl_init with
 * its HIP device current; this installs the exchange (pg_set_exchange).  0 on success; pg_rccl_error() says why not. */
int pg_rccl_unique_id(void *out128);
int pg_rccl_init(int32_t rank, int32_t world, const void *id128);
int pg_rccl_finalize(void);
const char *pg_rccl_error(void);

/* The exchange between the processes one command forks, for backends whose vectors live in host memory (the HIP backend uses
 * RCCL): the launcher maps a shared region before the fork (pg_shm_create), every rank installs it after (pg_shm_init). */
void *pg_shm_create(int32_t world, int64_t slot_bytes);
int pg_shm_init(void *region, int32_t rank);
/* 1 when the linked backend keeps its vectors in device memory (HIP), 0 otherwise; the HIP device a process is to use */
int pg_backend_is_device(void);
int pg_set_device(int32_t device);
int pg_device_count(void);

/* Wall-clock seconds spent inside the last pg_post_process + pg_graph_gen (stages A+B+C), and the
 * number of hits they processed (local shard). */
double  pg_last_path_seconds(void);
int64_t pg_last_path_hits(void);
/* how many times the last pg_graph_gen ran stages A-C: 1, plus one for every repeat after a tie-order hazard on a contig that did
 * not follow the reference's exact order yet (mode auto; DESIGN.md "bit-identity") */
int     pg_last_attempts(void);
double  pg_last_upload_seconds(void);   /* allocation + H2D copy + order-replay set-up of the last pg_post_process (0 for a resident rerun) */
double  pg_last_pack_seconds(void);     /* wall seconds the reader spent packing the genomes of the last upload into their blocks */
/* hits and exons of the local shard of d (what the last pg_post_process uploaded) */
int     pg_shard_counts(const pg_data_t *d, int64_t *n_hit, int64_t *n_exon);

/* pg_post_process / pg_graph_gen leave the per-hit FLAG fields of the host records (flt, shadow, rank, ...) on the
 * device until a writer needs them: pg_write_graph/pg_write_walk only fetch one flt bit per hit, pg_write_bed
 * fetches everything.  A caller that reads d->genome[j].hit[i] itself calls this first. */
int pg_sync_host(pg_data_t *d);

/* Tie-order policy (see DESIGN.md "bit-identity"): 0 canonical stable order everywhere; 1 (default,
 * env PANGENE_EXACT=auto) replay the reference's unstable sort for the first contig of genomes whose
 * leading tie group has >= 2 hits; 2 (PANGENE_EXACT=all) replay it for every contig. */
void pg_set_exact_mode(int mode);

/* Benchmark support: make the next pg_post_process(opt, d) restart stages A+B+C on the shard that is
 * already resident in HBM (no re-pack, no PCIe upload).  The host hit arrays are left as they are. */
int pg_rerun_resident(pg_data_t *d);

/* HIP-event timing of kernel classes of the runs since the last pg_kernel_timing_reset (which also switches the
 * timing on): which 0 = stage-A sweep pg_shadow(cal_dom_sc=1) ("K1", the hit-filter+overlap kernel), 1 = pg_flt_ov_isoform
 * sweep, 2 = (not timed any more: the stage-C sweeps), 3 = all of stage A (sorts, per-hit constants, pg_flag_pseudo, sweeps, filters),
 * 4 = no kernel class: n_launch = the number of times the host waited for the backend's stream since the reset; 5 / 6 (with PANGENE_TIME_ROUNDS=1 at
 * the reset) = every pg_gen_arc round / its walk alone, which | (k + 1) << 8 = the k-th timed launch of the class alone; 7 = no timing: which
 * build of K1 the upload got (n_launch 1 = exon lists staged in LDS, 0 = read in place, -1 = not decided), total_ms = the sampled exons a tile. */
int pg_kernel_timing(pg_data_t *d, int32_t which, double *total_ms, int64_t *n_launch, int64_t *units);
double pg_last_reserve_seconds(void); /* what the last pg_read_paf_batch spent reserving device memory before it parsed (large data sets: pga_reserve) */
int pg_kernel_timing_reset(pg_data_t *d);
/* collectives the driver has issued through the exchange callbacks so far, in this process (bench.py: per pass of a sharded run) */
int64_t pg_collective_count(void);

/* Page-locked host memory: the library keeps the pinned slabs of an upload for the next one (up to 4 GiB while a data set is
 * alive; 256 MiB after the last pg_data_destroy).  A long-lived host calls this to give back what exceeds keep_bytes (0 = all). */
void pg_trim_host_cache(size_t keep_bytes);

/* Bring the device runtime up and load the library's kernels ahead of the first pg_post_process (a command line calls it on a
 * helper thread while the PAF files are parsed); a no-op on backends without a device */
int pg_device_warm(void);

/* HBM bandwidth a plain copy kernel reaches on the current device, GB/s of read + write (measurement support: the calibration
 * SURVEY.md 8d asks for beside the spec peak); negative without a device */
double pg_device_copy_gbps(size_t bytes, int32_t reps);

/* wall seconds the host driver spent per phase of the last run (names via pg_phase_name) */
int pg_phase_times(double *out, int n);
const char *pg_phase_name(int i);

#ifdef __cplusplus
}
#endif
#endif
