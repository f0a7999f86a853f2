// pangene command line: `pangene [options] <in.paf> [...] > graph.gfa` with the reference's option
// letters, defaults and usage text (main.c:12-152, option.c:9-25), on top of libpangene_amd.
#include <dlfcn.h>
#include <getopt.h>
#include <signal.h>
#include <sys/prctl.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <string>
#include <thread>
#include <vector>
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
	std::fprintf(fp, "    --gpus=INT    shard the genomes over INT GPUs of this node: one process per device, RCCL over xGMI [1]\n");
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

// ---------------------------------------------------------------------------------------------------------------
// `pangene --gpus N`: main.c:117-142 for N devices of one node.  The command forks N - 1 workers BEFORE anything touches the GPU;
// rank r takes device r and the r-th contiguous block of the PAF files (so that the ranks' W / BED lines, concatenated in rank
// order, are in command-line order) and registers the names of the other files (ids as in a sequential read).  The exchange is
// RCCL on the kernels' stream (the 128-byte id travels through a pipe); on a backend without a device (the oracle host of the
// tests) it is a shared-memory region mapped before the fork.  Rank 0 prints the graph; every rank writes the lines of its own
// genomes to a temporary file that rank 0 copies to stdout in rank order.
// ---------------------------------------------------------------------------------------------------------------
struct Output { int matrix = 0; };

static int run_path(pg_opt_t &opt, int n_files, char **files, const uint8_t *ids_only, const Output &o, bool graph_lines, bool own_lines, int device = -1)
{
	// PANGENE_TIMING=1: where the wall time of the command goes (stderr: the line bench.py's cli leg reads), beside the library's own lines
	const bool timing = std::getenv("PANGENE_TIMING") != nullptr;
	const double t0 = pg_realtime();
	// (Tried: HIP initialisation + code-object load on a helper thread while the files are parsed -- pg_device_warm().  The runtime's
	// start-up maps and registers memory for ~0.3 s and every one of those calls stalls the page faults of the parser threads of the
	// same address space: parsing 100 files took 0.30 s instead of 0.05 s and the command got slower, 0.55 s against 0.46 s.  So the
	// device comes up when pg_post_process first needs it; `device_warm_s` below is that start-up, measured on its own.)
	pg_data_t *d = pg_data_init();
	pg_read_paf_batch(&opt, d, n_files, files, ids_only, 0); // parallel parse, ids as in sequential pg_read_paf calls
	const double t1 = pg_realtime();
	if (device >= 0) pg_set_device(device);
	if (timing) pg_device_warm(); // (only to itemise it: pg_post_process would pay it otherwise)
	const double t2 = pg_realtime();
	const double t_warm = t2 - t1;
	pg_post_process(&opt, d);
	const double t3 = pg_realtime();
	double t4 = t3;
	int rc = 0;
	if (pg_last_error()) rc = 2;
	else if (opt.flag & PG_F_WRITE_BED_RAW) { if (own_lines) pg_write_bed(d, 0); }
	else {
		pg_graph_t *g = pg_graph_init(d);
		pg_graph_gen(&opt, g);
		t4 = pg_realtime();
		if (pg_last_error()) rc = 2;
		else if (o.matrix) pg_write_matrix(g, o.matrix == 2);
		else if (opt.flag & PG_F_WRITE_BED_WALK) { if (own_lines) pg_write_bed(d, 1); }
		else if (opt.flag & PG_F_WRITE_BED_FLAG) { if (own_lines) pg_write_bed(d, 0); }
		else {
			if (graph_lines) { pg_write_graph(g); std::fflush(stdout); }
			if (own_lines && !(opt.flag & PG_F_WRITE_NO_WALK)) pg_write_walk(g);
		}
		pg_graph_destroy(g);
	}
	std::fflush(stdout);
	const double t5 = pg_realtime();
	pg_data_destroy(d);
	if (timing) std::fprintf(stderr, "[cli_timing] {\"parse_s\": %.4f, \"device_warm_s\": %.4f, \"post_process_s\": %.4f, \"graph_gen_s\": %.4f, \"write_s\": %.4f, \"teardown_s\": %.4f}\n",
	                         t1 - t0, t_warm, t3 - t2, t4 - t3, t5 - t4, pg_realtime() - t5);
	return rc;
}

static bool read_all(int fd, void *buf, size_t n) { char *p = (char *)buf; while (n) { ssize_t k = read(fd, p, n); if (k <= 0) return false; p += k, n -= (size_t)k; } return true; }
static bool write_all(int fd, const void *buf, size_t n) { const char *p = (const char *)buf; while (n) { ssize_t k = write(fd, p, n); if (k <= 0) return false; p += k, n -= (size_t)k; } return true; }

// what a signal handler / the watchdog needs to take the whole command down: the workers' pids and the temporary files
static std::vector<pid_t> g_kids;
static std::vector<std::string> g_tmp;
static std::atomic<int> g_kid_failed{0}; // a worker ended with an error while rank 0 was still at work

static void take_down(bool unlink_tmp)
{
	for (pid_t p : g_kids) if (p > 0) kill(p, SIGKILL);
	if (unlink_tmp) for (const std::string &t : g_tmp) unlink(t.c_str());
}
static void on_signal(int sig) { take_down(true); _exit(128 + sig); }

// Files of rank r = a contiguous block of the command line (the ranks' W / BED lines, concatenated in rank order, are then in
// command-line order), cut so that the blocks carry about the same number of HITS.  The hits of a file are not
// known before it is parsed; its size is (one PAF line is one alignment; a .gz counts five times its size, as in the reader).
static std::vector<int> partition_files(int W, int n_files, char **files)
{
	std::vector<double> w((size_t)n_files, 1.0);
	double tot = 0;
	for (int i = 0; i < n_files; ++i) {
		struct stat sb;
		const size_t len = std::strlen(files[i]);
		if (stat(files[i], &sb) == 0 && sb.st_size > 0) w[(size_t)i] = (double)sb.st_size * (len > 3 && std::strcmp(files[i] + len - 3, ".gz") == 0 ? 5.0 : 1.0);
		tot += w[(size_t)i];
	}
	std::vector<int> cut((size_t)W + 1, n_files);
	cut[0] = 0;
	double acc = 0;
	int r = 1;
	for (int i = 0; i < n_files && r < W; ++i) { // rank r starts at the first file at which the weight before it reaches r / W of the total
		while (r < W && acc >= tot * r / W) cut[(size_t)r++] = i;
		acc += w[(size_t)i];
	}
	return cut;
}

static int run_sharded(pg_opt_t &opt, int W, int n_files, char **files, const Output &o)
{
	if (o.matrix) { std::fprintf(stderr, "ERROR: --matrix needs every genome in one process; run it without --gpus\n"); return 1; }
	const bool dev = pg_backend_is_device() != 0;
	typedef int (*uid_fn)(void *); typedef int (*init_fn)(int32_t, int32_t, const void *); typedef int (*fin_fn)(void);
	uid_fn rccl_uid = nullptr; init_fn rccl_init = nullptr; fin_fn rccl_fin = nullptr;
	void *region = nullptr;
	if (dev) { // bound at run time: only the HIP build of the library has them
		rccl_uid = (uid_fn)dlsym(RTLD_DEFAULT, "pg_rccl_unique_id"), rccl_init = (init_fn)dlsym(RTLD_DEFAULT, "pg_rccl_init"), rccl_fin = (fin_fn)dlsym(RTLD_DEFAULT, "pg_rccl_finalize");
		if (!rccl_uid || !rccl_init) { std::fprintf(stderr, "ERROR: this build of the library has no RCCL exchange\n"); return 1; }
	} else if ((region = pg_shm_create(W, (int64_t)16 << 20)) == nullptr) { std::fprintf(stderr, "ERROR: cannot map the exchange region\n"); return 1; }
	const std::vector<int> cut = partition_files(W, n_files, files);
	std::vector<std::string> &tmp = g_tmp;
	tmp.assign((size_t)W, std::string());
	std::vector<int> id_pipe((size_t)W * 2, -1), st_pipe((size_t)W * 2, -1);
	std::vector<pid_t> &kid = g_kids;
	kid.assign((size_t)W, 0);
	const char *td = std::getenv("TMPDIR");
	for (int r = 0; r < W; ++r) {
		std::string t = std::string(td && *td ? td : "/tmp") + "/pangene_rank" + std::to_string(r) + "_XXXXXX";
		const int fd = mkstemp(&t[0]);
		if (fd < 0) { std::fprintf(stderr, "ERROR: cannot create a temporary file for rank %d\n", r); take_down(true); return 1; }
		close(fd);
		tmp[(size_t)r] = t;
		if (r && (pipe(&id_pipe[(size_t)r * 2]) != 0 || pipe(&st_pipe[(size_t)r * 2]) != 0)) { std::fprintf(stderr, "ERROR: pipe()\n"); take_down(true); return 1; }
	}
	std::fflush(stdout); std::fflush(stderr);
	const pid_t parent = getpid();
	int rank = 0;
	for (int r = 1; r < W; ++r) {
		const pid_t p = fork();
		if (p < 0) { std::fprintf(stderr, "ERROR: fork()\n"); take_down(true); return 1; }
		if (p == 0) {
			rank = r;
			prctl(PR_SET_PDEATHSIG, SIGKILL); // a worker never outlives rank 0 (it would wait in a collective for ever)
			if (getppid() != parent) _exit(3);
			for (int k = 1; k < W; ++k) { // the other workers' pipes are none of this one's business (an inherited write end would keep a dead rank 0's pipe open)
				if (k == r) continue;
				for (int e = 0; e < 2; ++e) { if (id_pipe[(size_t)k * 2 + e] >= 0) close(id_pipe[(size_t)k * 2 + e]); if (st_pipe[(size_t)k * 2 + e] >= 0) close(st_pipe[(size_t)k * 2 + e]); }
			}
			std::fill(kid.begin(), kid.end(), 0);
			break;
		}
		kid[(size_t)r] = p;
	}
	if (rank == 0) { signal(SIGINT, on_signal); signal(SIGTERM, on_signal); signal(SIGHUP, on_signal); }
	std::vector<uint8_t> ids_only((size_t)n_files, 1);
	for (int i = cut[(size_t)rank]; i < cut[(size_t)rank + 1]; ++i) ids_only[(size_t)i] = 0;
	int rc = 0;
	if (dev && pg_set_device(rank) != 0) { std::fprintf(stderr, "[E::pangene] rank %d: no HIP device %d\n", rank, rank); rc = 3; }
	// bootstrap: rank 0 hands the id out and hears from every worker before anybody enters the communicator
	unsigned char id[128] = { 0 };
	if (rank == 0) {
		if (dev && rc == 0 && rccl_uid(id) != 0) rc = 3;
		unsigned char ok = rc == 0 ? 1 : 0;
		for (int r = 1; r < W; ++r) {
			close(id_pipe[(size_t)r * 2]), close(st_pipe[(size_t)r * 2 + 1]);
			if (!write_all(id_pipe[(size_t)r * 2 + 1], &ok, 1) || !write_all(id_pipe[(size_t)r * 2 + 1], id, sizeof(id))) ok = 0;
		}
		for (int r = 1; r < W; ++r) { unsigned char s = 0; if (!read_all(st_pipe[(size_t)r * 2], &s, 1) || !s) ok = 0; }
		for (int r = 1; r < W; ++r) write_all(id_pipe[(size_t)r * 2 + 1], &ok, 1); // go / no go
		if (!ok) { for (int r = 1; r < W; ++r) { int st; waitpid(kid[(size_t)r], &st, 0); kid[(size_t)r] = 0; } take_down(true); return 3; } // (reaped: take_down must not signal a pid the system may have given to somebody else)
	} else {
		close(id_pipe[(size_t)rank * 2 + 1]), close(st_pipe[(size_t)rank * 2]);
		unsigned char ok = 0, go = 0, mine = rc == 0 ? 1 : 0;
		if (!read_all(id_pipe[(size_t)rank * 2], &ok, 1) || !read_all(id_pipe[(size_t)rank * 2], id, sizeof(id))) ok = 0;
		if (!ok) mine = 0;
		write_all(st_pipe[(size_t)rank * 2 + 1], &mine, 1);
		if (!read_all(id_pipe[(size_t)rank * 2], &go, 1) || !go) _exit(3);
		opt.flag &= ~PG_F_WRITE_VTX_SEL; // (-G lines come from rank 0 only)
		if (pg_verbose > 1) pg_verbose = 1; // one log, rank 0's (the ROUTES of a sharded run follow a level all ranks agree on: graph_driver.cpp route_v)
	}
	// From here on a rank that fails alone would leave the others waiting in a collective for ever.  Rank 0 watches its workers: the
	// first one that ends with an error (or by a signal) takes the command down -- workers and temporary files -- with status 2.
	std::atomic<bool> watch_on{rank == 0};
	std::vector<int> kid_status((size_t)W, -1); // exit status of the workers the watchdog has reaped
	std::thread watchdog;
	if (rank == 0 && W > 1) watchdog = std::thread([&]() {
		int left = W - 1;
		while (watch_on.load() && left > 0) {
			bool any = false;
			for (int r = 1; r < W; ++r) {
				if (kid[(size_t)r] <= 0 || kid_status[(size_t)r] >= 0) continue;
				int st = 0;
				const pid_t p = waitpid(kid[(size_t)r], &st, WNOHANG);
				if (p != kid[(size_t)r]) continue;
				any = true, --left;
				kid_status[(size_t)r] = (WIFEXITED(st) && WEXITSTATUS(st) == 0) ? 0 : 1;
				if (kid_status[(size_t)r] == 0) kid[(size_t)r] = 0; // (reaped: the pid is nobody's any more)
				else {
					std::fprintf(stderr, "[E::pangene] rank %d failed; stopping the other ranks\n", r);
					g_kid_failed.store(1);
					kid[(size_t)r] = 0;
					take_down(true);
					_exit(2);
				}
			}
			if (!any) usleep(20000);
		}
	});
	if ((dev ? rccl_init(rank, W, id) : pg_shm_init(region, rank)) != 0) { std::fprintf(stderr, "[E::pangene] rank %d: cannot join the exchange\n", rank); rc = 3; }
	if (const char *fr = std::getenv("PANGENE_FAULT_RANK")) if (std::atoi(fr) == rank) { std::fprintf(stderr, "[E::pangene] rank %d: injected fault (PANGENE_FAULT_RANK)\n", rank); rc = 7; } // (tests: a rank that fails alone)
	if (rc == 0) {
		if (rank) { if (pg_set_output(tmp[(size_t)rank].c_str()) != 0) rc = 3; }
		if (rc == 0) rc = run_path(opt, n_files, files, ids_only.data(), o, rank == 0, true, dev ? rank : -1);
		if (rank) pg_set_output(nullptr);
	}
	if (rank) { // (no RCCL teardown on a failed rank: the others may be inside a collective -- rank 0's watchdog ends them)
		if (rc == 0 && dev && rccl_fin) rccl_fin();
		std::fflush(stderr);
		_exit(rc);
	}
	if (rc != 0) { // rank 0 failed: the workers may be waiting for it
		watch_on.store(false);
		if (watchdog.joinable()) watchdog.join();
		take_down(true);
		return rc;
	}
	if (dev && rccl_fin) rccl_fin();
	std::fflush(stdout);
	watch_on.store(false);
	if (watchdog.joinable()) watchdog.join();
	for (int r = 1; r < W; ++r) {
		if (kid_status[(size_t)r] == 0) continue; // (reaped by the watchdog, ended well)
		int st = 0;
		if (kid[(size_t)r] <= 0 || waitpid(kid[(size_t)r], &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) { std::fprintf(stderr, "[E::pangene] rank %d failed\n", r); rc = rc ? rc : 2; }
		kid[(size_t)r] = 0;
	}
	for (int r = 1; r < W && rc == 0; ++r) { // the other ranks' lines, in rank (= command-line) order
		FILE *fp = std::fopen(tmp[(size_t)r].c_str(), "rb");
		if (!fp) { rc = 2; break; }
		char buf[1 << 16];
		size_t k;
		while ((k = std::fread(buf, 1, sizeof(buf), fp)) > 0) std::fwrite(buf, 1, k, stdout);
		std::fclose(fp);
	}
	take_down(true);
	return rc;
}

int main(int argc, char *argv[])
{
	if (argc >= 2 && std::strcmp(argv[1], "gfa2matrix") == 0) return main_gfa2matrix(argc - 1, argv + 1);
	int matrix = 0, n_gpus = 1; // matrix: 1 presence, 2 counts
	static const struct option lopts[] = {
		{ "bed", optional_argument, nullptr, 301 }, { "ori-sc", no_argument, nullptr, 302 }, { "matrix", optional_argument, nullptr, 303 },
		{ "gpus", required_argument, nullptr, 304 }, { "procs", required_argument, nullptr, 304 },
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
		case 304: n_gpus = std::atoi(optarg); break;
		case 401: std::puts(PG_VERSION); return 0;
		default: break;
		}
	}
	if (argc - optind < 1) return usage(stderr, &opt);
	Output o;
	o.matrix = matrix;
	int rc;
	if (n_gpus > 1) {
		rc = run_sharded(opt, n_gpus, argc - optind, argv + optind, o);
	} else rc = run_path(opt, argc - optind, argv + optind, nullptr, o, true, true);
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
