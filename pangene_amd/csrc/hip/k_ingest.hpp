// k_ingest.hpp -- stage A kernels except the sweep: per-hit constants and sort keys, pg_flag_pseudo (hit.c:66-105), and the filters that consume the sweep's flags (pg_flt_ov_isoform's apply step, pg_flt_chain_shadow, pg_flt_subopt_isoform, hit.c:107-146).
// Included by pga_backend.hip (one translation unit); uses the context types, BLOCK / WAVE and dev_prims.hpp from there.
#pragma once

// ------------------------------------------------------------------------------------------------
// create: derive per-hit constants in file order, sort into X order, gather, pm, Y order
// ------------------------------------------------------------------------------------------------
// The shard arrives as one block per genome (pga_genome_block_t: 10 planes of n_hit int32, rev bytes, exon pairs), copied as
// it is into `raw`.  One pass spreads the blocks into flat file-order arrays (plane f of the shard at up + f * N; rev bytes at
// plane 14) and makes the exon offsets shard-wide.
// The blocks are also CHECKED here (once per upload): a hit whose contig id, coordinates or exon range lie outside what its genome
// block declares (n_ctg, n_exon, max_cs, max_cm, max_score_adj, the two any_* marks) would index out of bounds or lose key bits
// later on -- dcnt[8] counts them and pga_create answers PGA_ERR_RANGE.
__global__ __launch_bounds__(BLOCK) void k_unblock(const int32_t *raw, const int64_t *woff, const int32_t *goff, const int32_t *eoff, int n_genome, int n, int32_t *up,
                                                     const int32_t *ctg_base, int32_t max_cs, int32_t max_cm, int32_t max_sadj, int neg_sadj, int multi, int n_prot, int64_t *dcnt)
{
	const int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	const int g = genome_of(goff, n_genome, i), li = i - goff[g], ng = goff[g + 1] - goff[g];
	const int32_t *b = raw + woff[g];
	int32_t w[PGA_BLOCK_PLANES];
#pragma unroll
	for (int f = 0; f < PGA_BLOCK_PLANES; ++f) {
		int32_t v = b[(int64_t)f * ng + li];
		w[f] = v;
		if (f == 6) v += eoff[g]; // off_exon
		up[(int64_t)f * n + i] = v;
	}
	{ // planes: 0 pid, 1 contig, 2 rank, 3 score_ori, 4 score_adj, 5 n_exon, 6 off_exon, 7 cs, 8 ce, 9 cm
		const int n_ctg = ctg_base[g + 1] - ctg_base[g], n_ex = eoff[g + 1] - eoff[g];
		const bool bad = w[1] < 0 || w[1] >= n_ctg || w[7] < 0 || w[8] < w[7] || w[9] < 0 || w[7] > max_cs || w[9] > max_cm || w[5] < 0 || w[6] < 0 || w[6] + w[5] > n_ex ||
		                 (w[4] < 0 ? !neg_sadj : w[4] > max_sadj) || (w[5] != 1 && !multi) || w[0] < 0 || w[0] >= n_prot;
		if (bad) atomicAdd((unsigned long long *)&dcnt[8], 1ull);
	}
	((uint8_t *)(up + 14 * (int64_t)n))[i] = ((const uint8_t *)(b + (int64_t)PGA_BLOCK_PLANES * ng))[li];
}

__global__ __launch_bounds__(BLOCK) void k_unblock_exons(const int32_t *raw, const int64_t *woff, const int32_t *goff, const int32_t *eoff, int n_genome, int n_exon, int2 *exon)
{
	const int e = blockIdx.x * BLOCK + threadIdx.x;
	if (e >= n_exon) return;
	const int g = genome_of(eoff, n_genome, e), ng = goff[g + 1] - goff[g];
	const int32_t *x = raw + woff[g] + (int64_t)PGA_BLOCK_PLANES * ng + (ng + 3) / 4 + 2 * (int64_t)(e - eoff[g]);
	exon[e] = make_int2(x[0], x[1]);
}

struct FileHits { const int32_t *pid, *cid, *rank, *sori, *sadj, *nex, *offx, *cs, *ce, *cm; const uint8_t *rev; };

__global__ __launch_bounds__(BLOCK) void k_prepare(FileHits f, int n, const int32_t *goff, int n_genome, const int32_t *ctg_base,
                                                     const int2 *exon, const int32_t *prot_gid, const uint8_t *gene_pref,
                                                     int32_t *gnm_f, int32_t *seg_f, int32_t *gid_f, int32_t *cds_f, uint64_t *key, uint32_t *val,
                                                     int rk_shift, const int32_t *hrank, int32_t *rk_f, int32_t *fl_f, int64_t *irregular = nullptr)
{
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	int g = genome_of(goff, n_genome, i);
	// skip empty genomes that share the same offset: genome_of returns the LAST g with goff[g] <= i, which is the owner
	int sg = ctg_base[g] + f.cid[i];
	int gid = prot_gid[f.pid[i]];
	int len = 0, ne = f.nex[i], ox = f.offx[i];
	bool odd = false; int prev = INT32_MIN;
	for (int e = 0; e < ne; ++e) { int2 x = exon[ox + e]; len += x.y - x.x; odd = odd || x.y < x.x || x.x < prev; prev = x.y; } // pg_cds_len, overlap.c:45-51
	// an exon list that is not sorted and disjoint (a U / V intron shorter than 3 bp, read.c:59-62): the sweeps then merge step by step (cds_inter_ref)
	if (odd && irregular) atomicAdd((unsigned long long *)irregular, 1ull);
	gnm_f[i] = g, seg_f[i] = sg, gid_f[i] = gid, cds_f[i] = len;
	fl_f[i] = (int32_t)((f.rev[i] ? PGA_F_REV : 0u) | (ne != 1 ? F_MULTI : 0u)); // the static bits of the flag word
	if (rk_shift >= 0) { // the same order in 32 bits (see pga_ctx::rk_shift): 0 exactly when the 64-bit key is 0
		rk_f[i] = (int32_t)((uint32_t)f.sadj[i] << rk_shift | (uint32_t)gene_pref[gid] << (rk_shift - 1) | (uint32_t)hrank[f.pid[i]]);
		return;
	}
	key[i] = (uint64_t)(int64_t)f.sadj[i] << 33 | (uint64_t)gene_pref[gid] << 32 | hash_u32((uint32_t)f.pid[i]); // the score key of overlap.c:137
	val[i] = (uint32_t)i;
}

__global__ __launch_bounds__(BLOCK) void k_hkey(int P, uint64_t *key, uint32_t *val)
{
	int p = blockIdx.x * BLOCK + threadIdx.x;
	if (p < P) key[p] = hash_u32((uint32_t)p), val[p] = (uint32_t)p;
}

__global__ __launch_bounds__(BLOCK) void k_hrank(const uint64_t *ks, const uint32_t *vs, int P, int32_t *hrank)
{
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i < P) hrank[vs[i]] = ks[i] == 0 ? 0 : i + 1; // distinct proteins have distinct hashes; rank + 1 keeps 0 for a hash of 0
}

// The sweep only ever COMPARES score keys, so every hit gets the dense rank of its key over the shard (one sort per
// run): 32-bit compares instead of 64-bit ones, and rank and partner slot fit one 64-bit word for a single LDS
// atomicMax ("best winner, first in array order").  Key 0 keeps rank 0: such a hit never becomes a dominator.
__global__ __launch_bounds__(BLOCK) void k_rank_scatter(const uint64_t *ks, const uint32_t *vs, const int32_t *incl, int n, int32_t *rk_f)
{
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i < n) rk_f[vs[i]] = ks[i] == 0 ? 0 : incl[i]; // incl >= 1; when key 0 exists it owns rank value 1, which then stays unused
}

__global__ __launch_bounds__(BLOCK) void k_xkey(const int32_t *seg_f, const int32_t *cs_f, int n, int cs_bits, uint64_t *key, uint32_t *val)
{
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i < n) key[i] = (uint64_t)seg_f[i] << cs_bits | (uint32_t)cs_f[i], val[i] = (uint32_t)i;
}

// the 64-bit score keys of overlap.c:137 alone (shards whose key does not fit 32 bits rank them by a sort every pass): everything else
// k_prepare computes -- gene, CDS length (a walk over the exon list), static flag bits -- stands from the upload on
__global__ __launch_bounds__(BLOCK) void k_score_key(const int32_t *pid_f, const int32_t *sadj_f, const int32_t *gid_f, const uint8_t *gene_pref, int n, uint64_t *key, uint32_t *val)
{
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i < n) key[i] = (uint64_t)(int64_t)sadj_f[i] << 33 | (uint64_t)gene_pref[gid_f[i]] << 32 | hash_u32((uint32_t)pid_f[i]), val[i] = (uint32_t)i;
}

// CONTIG BINS (k_segsort.hpp: GenomeSort::bins): the file-order planes of the upload, grouped by contig once per upload.  perm = the file-order
// indices sorted (stably) by contig segment; plane 17 of the result = the hit's file index inside its genome.  Plane 14 holds bytes (rev).
__global__ __launch_bounds__(BLOCK) void k_cgroup(const int32_t *up, const uint32_t *perm, int64_t n, const int32_t *goff, int32_t *out)
{
	const int64_t c = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (c >= n) return;
	const int64_t s = perm[c];
#pragma unroll
	for (int f = 0; f < 17; ++f) if (f != 14) out[(int64_t)f * n + c] = up[(int64_t)f * n + s];
	((uint8_t *)(out + 14 * n))[c] = ((const uint8_t *)(up + 14 * n))[s];
	out[17 * n + c] = (int32_t)(s - goff[up[10 * n + s]]);
}

struct HitArrays {
	int32_t *fidx, *gnm, *seg, *pid, *gid, *cs, *ce, *cm, *cds, *nex, *offx, *sori, *sadj, *rank, *sdom, *pdom, *pdom0;
	int32_t *rk; uint32_t *flags;
};

__global__ __launch_bounds__(BLOCK) void k_gather(FileHits f, const int32_t *gnm_f, const int32_t *seg_f, const int32_t *gid_f, const int32_t *cds_f,
                                                    const int32_t *rk_f, const uint32_t *perm, int n, const int32_t *goff, HitArrays o)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	int s = (int)perm[h];
	int g = gnm_f[s];
	o.fidx[h] = s - goff[g], o.gnm[h] = g, o.seg[h] = seg_f[s], o.pid[h] = f.pid[s], o.gid[h] = gid_f[s];
	o.cs[h] = f.cs[s], o.ce[h] = f.ce[s], o.cm[h] = f.cm[s], o.cds[h] = cds_f[s], o.nex[h] = f.nex[s], o.offx[h] = f.offx[s];
	o.sori[h] = f.sori[s], o.sadj[h] = f.sadj[s], o.rank[h] = f.rank[s], o.rk[h] = rk_f[s];
	o.sdom[h] = 0, o.pdom[h] = -1, o.pdom0[h] = 0; // read.c:133-134
	o.flags[h] = (f.rev[s] ? PGA_F_REV : 0u) | (h == goff[g] ? F_HEAD : 0u) | (f.nex[s] != 1 ? F_MULTI : 0u);
}

__global__ __launch_bounds__(BLOCK) void k_ykey(const int32_t *seg, const int32_t *cm, int n, int cm_bits, uint64_t *key, uint32_t *val)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	key[h] = (uint64_t)seg[h] << cm_bits | (uint32_t)cm[h];
	val[h] = (uint32_t)h;
}

// ------------------------------------------------------------------------------------------------
// pg_flag_pseudo (hit.c:66-105) with a (genome, protein) table instead of a sort by pid<<32|rank
// ------------------------------------------------------------------------------------------------
// Round 6: only a protein with SEVERAL hits in the genome can be marked (one hit: max_n == min_n, and hit.c:84 wants max_n > 1 with min_n == 1 or
// 2 min_n <= max_n), and two hits of one protein in one file never share a rank (read.c counts the protein's lines): such a protein has a hit of
// rank >= 1.  So the hits of rank >= 1 name the (genome, protein) cells that matter -- a bit each in a (genome x protein) bitmap that stays in the L2
// (3.4 MB for 500 x 55 k) -- and initialise those cells of the three tables themselves; every other hit tests its bit and leaves.  Rounds 1-5 filled the
// three tables (3 x 110 MB on the human-shaped shard) and sent two atomics per hit into them: 0.9 ms of that shard's 1.95 ms of stage A.
__global__ __launch_bounds__(BLOCK) void k_pseudo0(const int32_t *gnm, const int32_t *pid, const int32_t *rank, int n, int P, uint32_t *bits, int32_t *tmax, int32_t *tmin, int32_t *tr1)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n || rank[h] < 1) return;
	const int64_t t = (int64_t)gnm[h] * P + pid[h];
	atomicOr(&bits[t >> 5], 1u << (t & 31));
	tmax[t] = 0, tmin[t] = INT32_MAX, tr1[t] = INT32_MAX; // (the same three values from every writer of the cell; the atomics on them come with the next launch)
}
__device__ __forceinline__ bool ps_cell(const uint32_t *bits, int64_t t) { return (bits[t >> 5] >> (t & 31)) & 1u; }

__global__ __launch_bounds__(BLOCK) void k_pseudo1(const int32_t *gnm, const int32_t *pid, const int32_t *nex, int n, int P, const uint32_t *bits, int32_t *tmax, int32_t *tmin)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	int64_t t = (int64_t)gnm[h] * P + pid[h];
	if (!ps_cell(bits, t)) return;
	atomicMax(&tmax[t], nex[h]);
	atomicMin(&tmin[t], nex[h]);
}

__global__ __launch_bounds__(BLOCK) void k_pseudo2(const int32_t *gnm, const int32_t *pid, const int32_t *nex, const int32_t *rank, uint32_t *flags,
                                                     int n, int P, const uint32_t *bits, const int32_t *tmax, const int32_t *tmin, int32_t *tr1, int32_t *stats)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	int64_t t = (int64_t)gnm[h] * P + pid[h];
	if (!ps_cell(bits, t)) return;
	int mx = tmax[t], mn = tmin[t], ne = nex[h];
	if (!(mx > 1 && (mn == 1 || mn * 2 <= mx))) return; // hit.c:84
	if (ne == 1 || ne * 2 <= mx) {
		flags[h] |= PGA_F_PSEUDO | PGA_F_FLT; // hit.c:89 + PG_SET_FILTER(pseudo), read.c:246
		if (stats) atomicAdd(&stats[gnm[h] * 4 + 0], 1);
	} else atomicMin(&tr1[t], rank[h]);
}

// (C: the sweep's record C carries a copy of the rank -- the few ranks that change are patched here instead of by a pass over every record)
__global__ __launch_bounds__(BLOCK) void k_pseudo3(const int32_t *gnm, const int32_t *pid, int32_t *rank, int n, int P,
                                                     const uint32_t *bits, const int32_t *tmax, const int32_t *tmin, const int32_t *tr1, int4 *C)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	int64_t t = (int64_t)gnm[h] * P + pid[h];
	if (!ps_cell(bits, t)) return;
	int mx = tmax[t], mn = tmin[t], r1 = tr1[t];
	if (!(mx > 1 && (mn == 1 || mn * 2 <= mx)) || r1 == INT32_MAX || r1 == 0) return;
	int r = rank[h];
	if (r < r1) rank[h] = r + 1, ((int32_t *)&C[h])[0] = r + 1; // hit.c:95-97
	else if (r == r1) rank[h] = 0, ((int32_t *)&C[h])[0] = 0;
}

// ------------------------------------------------------------------------------------------------
// after the sweeps of stage A: counters for the log, isoform / chain / sub-optimal filters
// ------------------------------------------------------------------------------------------------
// log-only counters (graph.c:23-27).  Hits are genome-major, so a workgroup mostly sees one genome: count
// that genome in LDS and add once; stragglers of the next genome go to global memory directly.
__global__ __launch_bounds__(BLOCK) void k_count_shadow(const uint32_t *flags, const int32_t *gnm, int n, int32_t *stats)
{
	__shared__ int s_cnt[2];
	__shared__ int s_g;
	const int h = blockIdx.x * BLOCK + threadIdx.x;
	if (threadIdx.x == 0) s_cnt[0] = s_cnt[1] = 0, s_g = gnm[blockIdx.x * BLOCK];
	__syncthreads();
	if (h < n) {
		const uint32_t f = flags[h];
		if (!(f & PGA_F_FLT)) {
			const int g = gnm[h];
			if (g == s_g) { atomicAdd(&s_cnt[0], 1); if (f & PGA_F_SHADOW) atomicAdd(&s_cnt[1], 1); }
			else { atomicAdd(&stats[g * 2], 1); if (f & PGA_F_SHADOW) atomicAdd(&stats[g * 2 + 1], 1); }
		}
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		if (s_cnt[0]) atomicAdd(&stats[s_g * 2], s_cnt[0]);
		if (s_cnt[1]) atomicAdd(&stats[s_g * 2 + 1], s_cnt[1]);
	}
}

// ------------------------------------------------------------------------------------------------
// read.c:249-256 with ONE workgroup per genome and the per-genome tables in LDS: the reset after pg_shadow (read.c:249-253), the
// consequence of pg_flt_ov_isoform (overlap.c:89-91), pg_flt_chain_shadow (hit.c:130-146) and pg_flt_subopt_isoform (hit.c:107-128).
// A genome's hits are one contiguous block of the cs order; what the four kernels below keep in (genome x protein) bytes and
// (genome x gene) 64-bit words of HBM -- cleared and re-read every pass -- is P bytes + Q words of LDS here, and the hits are read
// once per step by the same thread (36 B/hit of traffic instead of ~140).  Used when P + 8 Q fits the LDS (k_iso_apply .. k_subopt2
// otherwise, and under PANGENE_FILTERS=global).
// ------------------------------------------------------------------------------------------------
constexpr int GF_T = 1024;
struct GenomeFilters { uint32_t *flags; const int32_t *pid, *gid, *rank, *sadj; int32_t *pdom, *pdom0; const int32_t *goff; const int4 *A; int P, Q; int32_t *stats; int64_t *dcnt; int32_t *hz_list;
                       int pos_bits; /* K32: bits of a position inside a genome */ };
// K32: no score_adj of the shard is negative and score_adj and a position inside a genome fit 32 bits together -- the per-gene `best`
// entries are 4 bytes then, and the tables of a 20 000-gene, 55 000-protein human annotation fit the LDS (8 Q + P bytes did not: those
// shards took four kernels over (genome x gene / protein) tables in HBM instead)
static inline size_t gf_lds_bytes(int P, int Q, bool k32 = false) { return (k32 ? 4 : 8) * (((size_t)Q + 1) & ~(size_t)1) + (((size_t)P + 7) & ~(size_t)7) + 64; }

template <bool K32>
__global__ __launch_bounds__(GF_T) void k_genome_filters(GenomeFilters a)
{
	typedef typename std::conditional<K32, uint32_t, unsigned long long>::type best_t;
	extern __shared__ unsigned long long gf_lds[];
	best_t *best = (best_t *)gf_lds;                                      // [Q] hit.c:111 `best`: score_adj, then the first position wins
	uint8_t *noiso = (uint8_t *)(best + (((size_t)a.Q + 1) & ~(size_t)1)); // [P] the protein has a hit here without flt_iso_ov (= !flag[] of hit.c:134-138)
	const uint32_t pmask = K32 ? (1u << a.pos_bits) - 1u : 0xffffffffu;
	int32_t *cnt = (int32_t *)(noiso + (((size_t)a.P + 7) & ~(size_t)7)); // [4] the genome's counts for the log line (read.c:257)
	const int g = blockIdx.x, tid = threadIdx.x, h0 = a.goff[g], h1 = a.goff[g + 1];
	for (int k = tid; k < a.Q; k += GF_T) best[k] = 0;
	for (int k = tid; k < (a.P + 7) / 8 * 2; k += GF_T) ((uint32_t *)noiso)[k] = 0;
	if (tid < 4) cnt[tid] = 0;
	__syncthreads();
	int n_iso = 0, n_chain = 0, n_sub = 0;
	constexpr int U = 4; // hits a thread has in flight per step: the loads of a step are issued together (the phases are latency-bound otherwise)
	{
		for (int hb = h0 + tid; hb < h1; hb += U * GF_T) { // overlap.c:89-91 (the marks come from k_sweep<3>, which has done read.c:249-253 itself) + the first loop of hit.c:136-138
			int32_t pi[U]; uint32_t fl[U];
#pragma unroll
			for (int u = 0; u < U; ++u) { const int h = hb + u * GF_T; if (h < h1) fl[u] = a.flags[h], pi[u] = a.pid[h]; }
#pragma unroll
			for (int u = 0; u < U; ++u) {
				const int h = hb + u * GF_T;
				if (h >= h1) break;
				if (fl[u] & PGA_F_ISO_OV) a.flags[h] = fl[u] | PGA_F_FLT, ++n_iso;
				else noiso[pi[u]] = 1;
			}
		}
	}
	__syncthreads();
	for (int hb = h0 + tid; hb < h1; hb += U * GF_T) { // hit.c:139-144, then the first loop of hit.c:112-118 (it skips what the chain filter just removed)
		int32_t p0[U], rk[U], sa[U], gi[U]; uint32_t fl[U];
#pragma unroll
		for (int u = 0; u < U; ++u) { const int h = hb + u * GF_T; if (h < h1) p0[u] = a.pdom0[h], fl[u] = a.flags[h], rk[u] = a.rank[h], sa[u] = a.sadj[h], gi[u] = a.gid[h]; }
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int h = hb + u * GF_T;
			if (h >= h1) break;
			uint32_t f = fl[u];
			if (p0[u] >= 0 && !noiso[p0[u]]) f |= PGA_F_FLT | PGA_F_CHAIN, a.flags[h] = f, ++n_chain;
			if ((f & PGA_F_FLT) || rk[u] > 0) continue;
			const uint32_t pos = (uint32_t)(h - h0);
			if (K32) { if (sa[u] > 0) atomicMax((uint32_t *)&best[gi[u]], (uint32_t)sa[u] << a.pos_bits | (pmask - pos)); }
			else if (sa[u] > 0) atomicMax((unsigned long long *)&best[gi[u]], (unsigned long long)(uint32_t)sa[u] << 32 | (0xffffffffu - pos));
			else if (sa[u] < 0) atomicMax((unsigned long long *)&best[gi[u]], 1ull << 63 | pos);
		}
	}
	__syncthreads();
	for (int hb = h0 + tid; hb < h1; hb += U * GF_T) { // the second loop of hit.c:119-125 (+ the tie hazard of k_subopt2)
		int32_t pi[U], gi[U], bp[U]; uint32_t fl[U]; unsigned long long kk[U];
#pragma unroll
		for (int u = 0; u < U; ++u) { const int h = hb + u * GF_T; if (h < h1) fl[u] = a.flags[h], pi[u] = a.pid[h], gi[u] = a.gid[h]; }
#pragma unroll
		for (int u = 0; u < U; ++u) { // the winner's protein: one more (dependent) load, again all of the step's together
			const int h = hb + u * GF_T;
			kk[u] = 0, bp[u] = 0; // hit.c:111: calloc'ed best => pid 0 when the gene has no candidate
			if (h < h1 && !(fl[u] & PGA_F_FLT)) {
				kk[u] = best[gi[u]];
				if (kk[u]) bp[u] = a.pid[h0 + (int)(K32 ? pmask - ((uint32_t)kk[u] & pmask) : (kk[u] >> 63) ? (uint32_t)kk[u] : 0xffffffffu - (uint32_t)kk[u])];
			}
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int h = hb + u * GF_T;
			if (h >= h1) break;
			if (fl[u] & PGA_F_FLT) continue;
			const unsigned long long k = kk[u];
			if (k && pi[u] != bp[u] && a.rank[h] == 0) { // a losing candidate: could it have been first?
				const int s = a.sadj[h];
				if (K32 ? (s > 0 && (uint32_t)s == (uint32_t)k >> a.pos_bits) : (k >> 63) ? s < 0 : (s > 0 && (uint32_t)s == (uint32_t)(k >> 32))) {
					const int w = h0 + (int)(K32 ? pmask - ((uint32_t)k & pmask) : (k >> 63) ? (uint32_t)k : 0xffffffffu - (uint32_t)k);
					const int4 ah = a.A[h], aw = a.A[w];
					if (ah.x == aw.x && ah.y == aw.y) { atomicAdd((unsigned long long *)&a.dcnt[7], 1ull); hz_note(&a.dcnt[14], a.hz_list, ah.y); }
				}
			}
			if (pi[u] != bp[u]) a.flags[h] = fl[u] | PGA_F_FLT | PGA_F_ISO_SUB, ++n_sub;
		}
	}
	n_iso = wave_sum(n_iso), n_chain = wave_sum(n_chain), n_sub = wave_sum(n_sub);
	if ((tid & 63) == 0) { atomicAdd(&cnt[1], n_iso); atomicAdd(&cnt[2], n_chain); atomicAdd(&cnt[3], n_sub); }
	__syncthreads();
	if (tid >= 1 && tid < 4) a.stats[g * 4 + tid] = cnt[tid];
}

// read.c:249-253 (pid_dom0 = pid_dom, pid_dom = -1, shadow = 0) + tail of pg_flt_ov_isoform (overlap.c:89-91) + first loop of
// pg_flt_chain_shadow (hit.c:136-138).  noiso: one byte per (genome, protein), set when the protein has a hit in the genome that
// does not carry flt_iso_ov (the complement of hit.c:134-138's flag[], for the proteins that occur at all -- pid_dom0 always does).
__global__ __launch_bounds__(BLOCK) void k_iso_apply(uint32_t *flags, const int32_t *gnm, const int32_t *pid, int32_t *pdom, int32_t *pdom0, int n, int P, uint8_t *noiso, int32_t *stats)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return; // (k_sweep<3> has done read.c:249-253 itself)
	const uint32_t f = flags[h];
	uint32_t nf = f & ~PGA_F_SHADOW;
	if (f & PGA_F_ISO_OV) {
		nf |= PGA_F_FLT;
		if (stats) atomicAdd(&stats[gnm[h] * 4 + 1], 1);
	} else {
		noiso[(int64_t)gnm[h] * P + pid[h]] = 1; // (plain byte stores of the same value: no atomics needed)
	}
	if (nf != f) flags[h] = nf;
}

// second loop of pg_flt_chain_shadow (hit.c:139-143)
__global__ __launch_bounds__(BLOCK) void k_chain(uint32_t *flags, const int32_t *gnm, const int32_t *pdom0, int n, int P, const uint8_t *noiso, int32_t *stats)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	int p0 = pdom0[h];
	if (p0 < 0) return;
	if (!noiso[(int64_t)gnm[h] * P + p0]) {
		flags[h] |= PGA_F_FLT | PGA_F_CHAIN;
		if (stats) atomicAdd(&stats[gnm[h] * 4 + 2], 1);
	}
}

// pg_flt_subopt_isoform (hit.c:107-128).  best[gene] of one genome = first maximum of score_adj in
// array order; the (int32 > uint64) comparison of hit.c:116 lets a negative score_adj always win, the
// last one in array order staying.
__global__ __launch_bounds__(BLOCK) void k_subopt1(const uint32_t *flags, const int32_t *gnm, const int32_t *gid, const int32_t *rank, const int32_t *sadj,
                                                     const int32_t *goff, int n, int Q, unsigned long long *tbest)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	if ((flags[h] & PGA_F_FLT) || rank[h] > 0) return;
	int s = sadj[h], g = gnm[h];
	uint32_t pos = (uint32_t)(h - goff[g]);
	unsigned long long k;
	if (s > 0) k = (unsigned long long)(uint32_t)s << 32 | (0xffffffffu - pos);
	else if (s < 0) k = 1ull << 63 | pos;
	else return;
	atomicMax(&tbest[(int64_t)g * Q + gid[h]], k);
}

// Tie order: the winner is the FIRST candidate with the maximal score_adj in array order (the last one among
// negative scores); a candidate of another protein with the winner's score and the winner's (contig, cs) could sit before it
// in the reference's unstable order: hazard.
__global__ __launch_bounds__(BLOCK) void k_subopt2(uint32_t *flags, const int32_t *gnm, const int32_t *gid, const int32_t *pid, const int32_t *goff, int n, int Q,
                                                     const unsigned long long *tbest, int32_t *stats, const int32_t *rank, const int32_t *sadj, const int4 *A,
                                                     int64_t *dcnt, int32_t *hz_list)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	uint32_t f = flags[h];
	if (f & PGA_F_FLT) return;
	int g = gnm[h];
	unsigned long long k = tbest[(int64_t)g * Q + gid[h]];
	int best_pid = 0; // hit.c:111: calloc'ed best => pid 0 when the gene has no candidate
	if (k) {
		uint32_t pos = (k >> 63) ? (uint32_t)k : 0xffffffffu - (uint32_t)k;
		const int w = goff[g] + (int)pos;
		best_pid = pid[w];
		if (pid[h] != best_pid && rank[h] == 0) { // a losing candidate: could it have been first?
			const int s = sadj[h];
			if ((k >> 63) ? s < 0 : (s > 0 && (uint32_t)s == (uint32_t)(k >> 32))) {
				const int4 ah = A[h], aw = A[w];
				if (ah.x == aw.x && ah.y == aw.y) { atomicAdd((unsigned long long *)&dcnt[7], 1ull); hz_note(&dcnt[14], hz_list, ah.y); }
			}
		}
	}
	if (pid[h] != best_pid) {
		flags[h] = f | PGA_F_FLT | PGA_F_ISO_SUB;
		if (stats) atomicAdd(&stats[g * 4 + 3], 1);
	}
}
