// pangene command line: `pangene [options] <in.paf> [...] > graph.gfa` with the reference's option
// letters, defaults and usage text (main.c:12-152, option.c:9-25), on top of libpangene_amd.
#include <getopt.h>
#include <sys/resource.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "pangene_amd.h"

static int usage(FILE *fp, const pg_opt_t *opt)
{
	std::fprintf(fp, "Usage: pangene [options] <in.paf> [...]\n");
	std::fprintf(fp, "Options:\n");
	std::fprintf(fp, "  Input preprocessing:\n");
	std::fprintf(fp, "    -d CHAR       gene-protein delimiter [%c]\n", opt->gene_delim);
	std::fprintf(fp, "    -X STR/@FILE  exclude genes in STR list or in @FILE []\n");
	std::fprintf(fp, "    -I STR/@FILE  include genes in the output graph []\n");
	std::fprintf(fp, "    -P STR/@FILE  prioritize genes in the output graph []\n");
	std::fprintf(fp, "    -e FLOAT      drop an alignment if its identity <FLOAT [%g]\n", opt->min_prot_iden);
	std::fprintf(fp, "    -l FLOAT      drop an alignment if <FLOAT fraction of the protein aligned [%g]\n", opt->min_prot_ratio);
	std::fprintf(fp, "    -m FLOAT      score adjustment coefficient [%g]\n", opt->score_adj_coef);
	std::fprintf(fp, "  Graph construction:\n");
	std::fprintf(fp, "    -f FLOAT      min overlap fraction [%g]\n", opt->min_ov_ratio);
	std::fprintf(fp, "    -J            don't filter pseudogenes across samples\n");
	std::fprintf(fp, "    -E            ignore genes that are single-exon in all genomes\n");
	std::fprintf(fp, "    -p FLOAT      gene considered if dominant in FLOAT fraction of genes [%g]\n", opt->min_vertex_ratio);
	std::fprintf(fp, "    -c INT        drop a gene if average occurrence is >INT [%d]\n", opt->max_avg_occ);
	std::fprintf(fp, "    -g INT        drop a gene if its in- or out-degree >INT [%d]\n", opt->max_degree);
	std::fprintf(fp, "    -r INT        drop a gene if it connects >INT distant loci [%d]\n", opt->max_dist_loci);
	std::fprintf(fp, "    -b FLOAT      demote a branching arc if weaker than the best by FLOAT [%g]\n", opt->branch_diff);
	std::fprintf(fp, "    -B FLOAT      cut a branching arc if weaker by FLOAT [%g]\n", opt->branch_diff_cut);
	std::fprintf(fp, "    -y FLOAT      cut a distant branching arc if weaker by FLOAT [%g]\n", opt->branch_diff_dist);
	std::fprintf(fp, "    -T INT        apply branch cutting for INT times [%d]\n", opt->n_branch_flt);
	std::fprintf(fp, "    -F            don't consider genes on different contigs as distant\n");
	std::fprintf(fp, "    -a INT        prune an arc if it is supported by <INT genomes [%d]\n", opt->min_arc_cnt);
	std::fprintf(fp, "  Output:\n");
	std::fprintf(fp, "    -w            Suppress walk lines (W-lines)\n");
	std::fprintf(fp, "    --bed[=STR]   output 12-column BED where STR is walk, raw or flag [walk]\n");
	std::fprintf(fp, "    --matrix[=STR] output the gene x assembly matrix of pangene.js gfa2matrix, STR presence or count [presence]\n");
	std::fprintf(fp, "    --version     print version number\n");
	std::fprintf(fp, "  Also: pangene gfa2matrix [-c] [-d FILE] [-p] <in.gfa>   (pangene.js gfa2matrix on a GFA file)\n");
	return fp == stdout ? 0 : 1;
}

static int64_t parse_num(const char *s) // "2m", "500k", ... (main.c:45-55)
{
	char *p;
	double x = std::strtod(s, &p);
	if (*p == 'G' || *p == 'g') x *= 1e9;
	else if (*p == 'M' || *p == 'm') x *= 1e6;
	else if (*p == 'K' || *p == 'k') x *= 1e3;
	return (int64_t)(x + .499);
}

static int main_gfa2matrix(int argc, char *argv[]) // pangene.js:1168-1183
{
	int c, copy_number = 0, print_cd = 0;
	const char *clstr = nullptr;
	while ((c = getopt(argc, argv, "cd:p")) >= 0) {
		if (c == 'c') copy_number = 1;
		else if (c == 'd') clstr = optarg;
		else if (c == 'p') print_cd = 1;
	}
	if (argc - optind < 1) {
		std::puts("Usage: pangene gfa2matrix [options] <in.gfa>\nOptions:\n  -c        output counts\n  -d FILE   CD-HIT cluster file to merge paralogs []");
		return 0;
	}
	return pg_gfa2matrix_file(argv[optind], copy_number, clstr, print_cd) == 0 ? 0 : 1;
}

int main(int argc, char *argv[])
{
	if (argc >= 2 && std::strcmp(argv[1], "gfa2matrix") == 0) return main_gfa2matrix(argc - 1, argv + 1);
	int matrix = 0; // 1 presence, 2 counts
	static const struct option lopts[] = {
		{ "bed", optional_argument, nullptr, 301 }, { "ori-sc", no_argument, nullptr, 302 }, { "matrix", optional_argument, nullptr, 303 },
		{ "version", no_argument, nullptr, 401 }, { nullptr, 0, nullptr, 0 } };
	pg_opt_t opt;
	pg_opt_init(&opt);
	int c;
	while ((c = getopt_long(argc, argv, "d:e:l:f:g:p:b:B:y:Fr:c:a:wv:GD:C:T:X:I:P:m:JOSE", lopts, nullptr)) >= 0) {
		switch (c) {
		case 'd': opt.gene_delim = *optarg; break;
		case 'X': opt.excl = pg_read_list_dict(optarg); break;
		case 'I': opt.incl = pg_read_list_dict(optarg); break;
		case 'P': opt.preferred = pg_read_list_dict(optarg); break;
		case 'e': opt.min_prot_iden = std::atof(optarg); break;
		case 'l': opt.min_prot_ratio = std::atof(optarg); break;
		case 'm': opt.score_adj_coef = std::atof(optarg); break;
		case 'f': opt.min_ov_ratio = std::atof(optarg); break;
		case 'p': opt.min_vertex_ratio = std::atof(optarg); break;
		case 'c': opt.max_avg_occ = std::atoi(optarg); break;
		case 'g': opt.max_degree = std::atoi(optarg); break;
		case 'r': opt.max_dist_loci = std::atoi(optarg); break;
		case 'J': opt.flag |= PG_F_NO_JOINT_PSEUDO; break;
		case 'E': opt.flag |= PG_F_DROP_SGL_EXON; break;
		case 'b': opt.branch_diff = std::atof(optarg); break;
		case 'B': opt.branch_diff_cut = std::atof(optarg); break;
		case 'y': opt.branch_diff_dist = std::atof(optarg); break;
		case 'T': opt.n_branch_flt = (int32_t)std::atof(optarg); break;
		case 'a': opt.min_arc_cnt = std::atoi(optarg); break;
		case 'F': opt.flag |= PG_F_FRAG_MODE; break;
		case 'D': opt.local_dist = (int32_t)parse_num(optarg); break;
		case 'C': opt.local_count = std::atoi(optarg); break;
		case 'S': opt.flag |= PG_F_CHECK_STRAND; break;
		case 'w': opt.flag |= PG_F_WRITE_NO_WALK; break;
		case 'G': opt.flag |= PG_F_WRITE_VTX_SEL; break;
		case 'v': pg_verbose = std::atoi(optarg); break;
		case 'O': break; // accepted and ignored, as in the reference
		case 301:
			if (optarg == nullptr || std::strcmp(optarg, "walk") == 0) opt.flag |= PG_F_WRITE_BED_WALK;
			else if (std::strcmp(optarg, "raw") == 0) opt.flag |= PG_F_WRITE_BED_RAW;
			else if (std::strcmp(optarg, "flag") == 0) opt.flag |= PG_F_WRITE_BED_FLAG;
			else { std::fprintf(stderr, "ERROR: unrecognized --bed argument. Should be 'raw' or 'walk'.\n"); return 1; }
			break;
		case 302: opt.flag |= PG_F_ORI_FOR_BRANCH; break;
		case 303: matrix = (optarg && std::strcmp(optarg, "count") == 0) ? 2 : 1; break;
		case 401: std::puts(PG_VERSION); return 0;
		default: break;
		}
	}
	if (argc - optind < 1) return usage(stderr, &opt);
	pg_data_t *d = pg_data_init();
	pg_read_paf_batch(&opt, d, argc - optind, argv + optind, nullptr, 0); // parallel parse, ids as in sequential pg_read_paf calls
	pg_post_process(&opt, d);
	int rc = 0;
	if (pg_last_error()) rc = 2;
	else if (opt.flag & PG_F_WRITE_BED_RAW) pg_write_bed(d, 0);
	else {
		pg_graph_t *g = pg_graph_init(d);
		pg_graph_gen(&opt, g);
		if (pg_last_error()) rc = 2;
		else if (matrix) pg_write_matrix(g, matrix == 2);
		else if (opt.flag & PG_F_WRITE_BED_WALK) pg_write_bed(d, 1);
		else if (opt.flag & PG_F_WRITE_BED_FLAG) pg_write_bed(d, 0);
		else {
			pg_write_graph(g);
			if (!(opt.flag & PG_F_WRITE_NO_WALK)) pg_write_walk(g);
		}
		pg_graph_destroy(g);
	}
	pg_data_destroy(d);
	if (opt.excl) pg_dict_destroy(opt.excl);
	if (opt.incl) pg_dict_destroy(opt.incl);
	if (opt.preferred) pg_dict_destroy(opt.preferred);
	if (pg_verbose >= 3) {
		struct rusage r;
		getrusage(RUSAGE_SELF, &r);
		std::fprintf(stderr, "[M::%s] Version: %s\n[M::%s] CMD:", __func__, PG_VERSION, __func__);
		for (int i = 0; i < argc; ++i) std::fprintf(stderr, " %s", argv[i]);
		std::fprintf(stderr, "\n[M::%s] stages A+B+C: %.3f sec for %ld hits; peak RSS: %.3f GB\n", __func__, pg_last_path_seconds(),
		             (long)pg_last_path_hits(), r.ru_maxrss / 1024.0 / 1024.0);
	}
	return rc;
}
