// pga_backend.hip -- the MI355X (gfx950) implementation of the thin device ABI in
// include/pangene_hip.h.  All per-hit work of the pangene graph-construction path runs here as
// hand-written HIP kernels over a structure-of-arrays shard that stays resident in HBM for the whole
// run (upload once, 19 interval-dominance sweeps, 17 arc rounds, 15 branch rounds, one download).
//
// Data layout (DESIGN.md "HBM layout"): hits are physically stored in X order = (genome, contig, cs,
// file index), one 32-bit array per field, so a wave reads 256 B contiguous per field and the sweep's
// neighbours are adjacent in memory.  `seg` is the dense (genome, contig) id, `pm` the per-contig
// running maximum of ce (bounds the look-back of the sweep), `yperm` the cm order as a permutation of
// X positions.  Keys never change, so the two sorts the reference repeats 67 times per genome
// (hit.c:29-64) are done exactly once.
//
// Everything is integer work except three IEEE-double expressions (overlap.c:134,170) -- compile with
// -ffp-contract=off.  No MFMA: this path is HBM/latency bound.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "pangene_hip.h"
#include "dev_prims.hpp"

using namespace pgd;

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
	fprintf(stderr, "[E::pga] %s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return PGA_ERR_NO_DEVICE; } } while (0)

#define F_HEAD 0x80000000u   // static: first hit of its genome in X order (index-0 quirk, overlap.c:108)
#define F_MULTI 0x40000000u  // static: the hit has more than one exon (lets the sweep skip the exon records)
#define F_PUBLIC 0x7ffu

static inline unsigned nblk(int64_t n, int per = BLOCK) { return (unsigned)((n + per - 1) / per); }

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct DevPool { // persistent, grow-only device temporaries keyed by slot
	std::vector<void *> p; std::vector<size_t> cap;
	void *get(int slot, size_t bytes)
	{
		if ((int)p.size() <= slot) p.resize(slot + 1, nullptr), cap.resize(slot + 1, 0);
		if (bytes == 0) bytes = 16;
		if (cap[slot] < bytes) {
			if (p[slot]) (void)hipFree(p[slot]);
			size_t want = bytes + bytes / 4 + 256;
			if (hipMalloc(&p[slot], want) != hipSuccess) { p[slot] = nullptr; cap[slot] = 0; return nullptr; }
			cap[slot] = want;
		}
		return p[slot];
	}
	void release() { for (void *q : p) if (q) (void)hipFree(q); p.clear(); cap.clear(); }
};

enum { // pool slots
	S_KEY_A, S_KEY_B, S_VAL_A, S_VAL_B, S_TABLE, S_TILE, S_I32_A, S_I32_B, S_I32_C, S_TAB_A, S_TAB_B, S_TAB_C, S_TAB_D,
	S_TDIST, S_TS1, S_TS2, S_TGEN, S_SDIST, S_SS1, S_SS2, S_SGEN, S_HEAD, S_SLOT, S_ARCS, S_SEGCNT, S_BITS, S_TRIPLES,
	S_WALK_VAL, S_WALK_PREV, S_PERM, S_OVPOS, S_OVFILE, S_RUNSTART, S_CDN, S_MG_KEY, S_MG_VAL, S_MG_SRC, S_MG_OUT, S_MG_HEAD, S_MG_SLOT, S_MG_RUN, S_BR_S1, S_BR_GID, S_BR_VS, S_BR_VE, S_BR_PC, S_BR_POFF, S_BR_GRP, S_BR_NDL, S_BR_SEGGID, S_PAIRS, S_NLCNT, S_ARCX, S_ARCW, S_WEAKNEW, S_RP_SEG, S_RP_R, S_RP_CM, S_RP_POS, S_DL, S_SCRATCH, S_UPLOAD, S_STATS, S_G2S, S_MISC, S_SLOW, S_HZLIST,
	S_COUNT
};

struct TimedLaunch { hipEvent_t a, b; int which; int64_t units; };

struct pga_ctx {
	hipStream_t st = nullptr; bool own_stream = false;
	int32_t n_genome = 0, n_genome_global = 0, P = 0, Q = 0, n_seg_ctg = 0;
	int32_t N = 0, E = 0;
	int n_cu = 256;
	uint32_t sweep_seq = 0; // parity selects the slow-list counter (dcnt[12] / dcnt[13])
	pga_params_t par;
	std::vector<int32_t> h_goff, h_ggl;
	// static per hit (X order)
	int32_t *fidx = 0, *gnm = 0, *seg = 0, *pid = 0, *gid = 0, *cs = 0, *ce = 0, *cm = 0, *cds = 0, *nex = 0, *offx = 0, *sori = 0, *sadj = 0, *pm = 0;
	int32_t *rk = 0;        // dense rank of the score key (score_adj, preferred, hash(pid)) of overlap.c:137 over the shard; 0 = key 0
	int sc_bits = 64;       // significant bits of that key
	bool any_multi = true;  // some hit has more than one exon
	bool rp_compact = false; // 8-byte (gene, genome) position records (see k_rep_fill)
	int4 *recA = 0, *recB = 0, *recC = 0; // packed sweep records (derived from the arrays above, see k_pack_rec)
	// dynamic per hit
	int32_t *rank = 0, *sdom = 0, *pdom = 0, *pdom0 = 0; uint32_t *flags = 0;
	int32_t *yperm = 0, *goff = 0, *ggl = 0, *ctg_base = 0, *inv = 0, *headpos = 0;
	int cs_bits = 1, cm_bits = 1, seg_bits = 1;
	int2 *exon = 0; int32_t *prot_gid = 0; uint8_t *gene_pref = 0;
	// exchange vectors
	int32_t *max_ori = 0; int64_t *sums = 0; int32_t *vtx_cnt = 0; int32_t *g2s = 0; int32_t n_seg = 0;
	int64_t *dcnt = 0;      // device counters: [0] triples [1] arcs-temp [2] misc [3] invariant flag, [4..7] hazards
	int64_t *h_cnt = 0;     // pinned mirror
	int64_t *h_box = 0;     // the same memory as the device sees it
	void *h_stage = nullptr; size_t h_stage_cap = 0; // pinned landing area of fetch_later
	int32_t *h_g2s = nullptr; size_t h_g2s_cap = 0; hipEvent_t g2s_done = nullptr; // pinned staging of flag_vtx's gene -> segment map
	DevPool pool;
	bool walk_valid = false; // S_WALK_VAL / S_WALK_PREV match the current flags and cm order
	int4 *yrecA = 0, *yrecB = 0; bool yrec_valid = false; // Y-order static records (k_pack_yrec), rebuilt after anything that changes their sources
	int64_t br_n = 0, br_np = 0; int32_t br_S = 0; // arcs / pairs / segments of the last branch_pairs
	std::vector<TimedLaunch> timed;
	std::vector<void *> owned;
};

template <class T> static int dalloc(pga_ctx *c, T **p, size_t n)
{
	void *q = nullptr;
	if (hipMalloc(&q, (n ? n : 1) * sizeof(T)) != hipSuccess) return PGA_ERR_NOMEM;
	*p = (T *)q;
	c->owned.push_back(q);
	return 0;
}

extern "C" int pga_is_device(void) { return 1; }

extern "C" const char *pga_strerror(int code)
{
	switch (code) {
	case PGA_OK: return "ok";
	case PGA_ERR_NO_DEVICE: return "no usable HIP device / HIP runtime error (this library has no CPU fallback)";
	case PGA_ERR_RANGE: return "value out of range for the device layout";
	case PGA_ERR_ARG: return "bad argument";
	case PGA_ERR_NOMEM: return "out of device memory";
	case PGA_ERR_INVARIANT: return "reference invariant violated";
	}
	return "unknown";
}

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t hash_u32(uint32_t key) // pg_hash_uint32, pgpriv.h:88-97
{
	key += ~(key << 15);
	key ^=  (key >> 10);
	key +=  (key << 3);
	key ^=  (key >> 6);
	key += ~(key << 11);
	key ^=  (key >> 16);
	return key;
}

__global__ void k_fill_i32(int32_t *p, int64_t n, int32_t v)
{
	int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i < n) p[i] = v;
}

// mailbox[k] = a[0] + b[0]: totals of a scan land in the device mailbox so that one 128-byte copy brings every
// size the host needs (one round trip instead of one per value)
// dcnt[10] = a[0] + b[0] (element count after a compaction scan), then all 16 device counters go straight into the pinned
// host mirror: the host reads them after the stream sync without a separate copy command
__global__ void k_mail_sum(const int32_t *a, const int32_t *b, int64_t *dcnt, int64_t *host_box)
{
	if (threadIdx.x == 0) dcnt[10] = (int64_t)a[0] + b[0];
	__syncthreads();
	if (threadIdx.x < 16) host_box[threadIdx.x] = dcnt[threadIdx.x];
}

__global__ void k_mail_flush(const int64_t *dcnt, int64_t *host_box)
{
	if (threadIdx.x < 16) host_box[threadIdx.x] = dcnt[threadIdx.x];
}

struct ZeroList { void *p[4]; unsigned long long dwords[4]; };
// several small clears in one launch (each hipMemsetAsync is a launch of its own; a round needs a dozen of them)
__global__ __launch_bounds__(BLOCK) void k_zero_multi(ZeroList z)
{
	unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		if (i < z.dwords[k]) { ((uint32_t *)z.p[k])[i] = 0; return; }
		i -= z.dwords[k];
	}
}

__device__ __forceinline__ int genome_of(const int32_t *goff, int n_genome, int i) // last g with goff[g] <= i
{
	int lo = 0, hi = n_genome;
	while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (goff[mid] <= i) lo = mid; else hi = mid; }
	return lo;
}

// ------------------------------------------------------------------------------------------------
// create: derive per-hit constants in file order, sort into X order, gather, pm, Y order
// ------------------------------------------------------------------------------------------------
struct FileHits { const int32_t *pid, *cid, *rank, *sori, *sadj, *nex, *offx, *cs, *ce, *cm; const uint8_t *rev; };

__global__ __launch_bounds__(BLOCK) void k_prepare(FileHits f, int n, const int32_t *goff, int n_genome, const int32_t *ctg_base,
                                                     const int2 *exon, const int32_t *prot_gid, const uint8_t *gene_pref,
                                                     int32_t *gnm_f, int32_t *seg_f, int32_t *gid_f, int32_t *cds_f, uint64_t *key, uint32_t *val)
{
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	int g = genome_of(goff, n_genome, i);
	// skip empty genomes that share the same offset: genome_of returns the LAST g with goff[g] <= i, which is the owner
	int sg = ctg_base[g] + f.cid[i];
	int gid = prot_gid[f.pid[i]];
	int len = 0, ne = f.nex[i], ox = f.offx[i];
	for (int e = 0; e < ne; ++e) { int2 x = exon[ox + e]; len += x.y - x.x; } // pg_cds_len, overlap.c:45-51
	gnm_f[i] = g, seg_f[i] = sg, gid_f[i] = gid, cds_f[i] = len;
	key[i] = (uint64_t)(int64_t)f.sadj[i] << 33 | (uint64_t)gene_pref[gid] << 32 | hash_u32((uint32_t)f.pid[i]); // the score key of overlap.c:137
	val[i] = (uint32_t)i;
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
__global__ __launch_bounds__(BLOCK) void k_pseudo1(const int32_t *gnm, const int32_t *pid, const int32_t *nex, int n, int P, int32_t *tmax, int32_t *tmin)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	int64_t t = (int64_t)gnm[h] * P + pid[h];
	atomicMax(&tmax[t], nex[h]);
	atomicMin(&tmin[t], nex[h]);
}

__global__ __launch_bounds__(BLOCK) void k_pseudo2(const int32_t *gnm, const int32_t *pid, const int32_t *nex, const int32_t *rank, uint32_t *flags,
                                                     int n, int P, const int32_t *tmax, const int32_t *tmin, int32_t *tr1, int32_t *stats)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	int64_t t = (int64_t)gnm[h] * P + pid[h];
	int mx = tmax[t], mn = tmin[t], ne = nex[h];
	if (!(mx > 1 && (mn == 1 || mn * 2 <= mx))) return; // hit.c:84
	if (ne == 1 || ne * 2 <= mx) {
		flags[h] |= PGA_F_PSEUDO | PGA_F_FLT; // hit.c:89 + PG_SET_FILTER(pseudo), read.c:246
		atomicAdd(&stats[gnm[h] * 4 + 0], 1);
	} else atomicMin(&tr1[t], rank[h]);
}

__global__ __launch_bounds__(BLOCK) void k_pseudo3(const int32_t *gnm, const int32_t *pid, int32_t *rank, int n, int P,
                                                     const int32_t *tmax, const int32_t *tmin, const int32_t *tr1)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	int64_t t = (int64_t)gnm[h] * P + pid[h];
	int mx = tmax[t], mn = tmin[t], r1 = tr1[t];
	if (!(mx > 1 && (mn == 1 || mn * 2 <= mx)) || r1 == INT32_MAX || r1 == 0) return;
	int r = rank[h];
	if (r < r1) rank[h] = r + 1; // hit.c:95-97
	else if (r == r1) rank[h] = 0;
}

// ------------------------------------------------------------------------------------------------
// the interval-dominance sweep: pg_shadow (overlap.c:101-178) and pg_flt_ov_isoform (58-93)
// ------------------------------------------------------------------------------------------------
// Packed per-hit records for the sweep: a partner costs 16-byte loads instead of a dozen 4-byte ones.
//   A = {cs, seg, ce, pm}   B = {rk, gid, cds, pid}   C = {rank, n_exon, off_exon, score_ori}
// (A.xy read as one 64-bit word is seg << 32 | cs: the sort key of the X order, non-decreasing along the array)
// C is only needed for multi-exon hits, for two hits with the same score key and for score_dom.
__global__ __launch_bounds__(BLOCK) void k_pack_rec(const int32_t *seg, const int32_t *cs, const int32_t *ce, const int32_t *pm, const int32_t *rk,
                                                      const int32_t *gid, const int32_t *cds, const int32_t *rank, const int32_t *nex, const int32_t *offx,
                                                      const int32_t *pid, const int32_t *sori, int n, int4 *A, int4 *B, int4 *C)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	A[h] = make_int4(cs[h], seg[h], ce[h], pm[h]);
	B[h] = make_int4(rk[h], gid[h], cds[h], pid[h]);
	C[h] = make_int4(rank[h], nex[h], offx[h], sori[h]);
}

// where a tie-order hazard (h2_cm_tie / h3_dom_tie) happened: contig-segment ids, at most PGA_HAZARD_CAP of them (counter: dcnt[14])
__device__ __forceinline__ void hz_note(int64_t *cnt14, int32_t *list, int seg)
{
	const unsigned long long at = atomicAdd((unsigned long long *)cnt14, 1ull);
	if (at < (unsigned long long)PGA_HAZARD_CAP) list[at] = seg;
}

struct SweepView {
	const int4 *A, *B, *C; const int32_t *sori; const int2 *exon;
	uint32_t *flags; int32_t *pdom, *sdom;
	int n; double min_ov; int check_strand;
	int stage_c; // some hit of the shard has several exons: stage the C records with the others
	int64_t *hz;
	int64_t *slow_cnt; int32_t *slow_list; // work list for k_sweep_slow
	long long *prof; // PGA_SW_PROFILE builds only
	int32_t *hz_list; // hz[10] (= dcnt[14]) counts its entries
};

// CDS intersection of hit a (exons ea[na], start ca) and hit b: pg_hit_overlap, overlap.c:6-42
__device__ __forceinline__ int cds_inter(const int2 *__restrict__ ex, int oa, int na, int ca, int ea_end, int ob, int nb, int cb, int eb_end)
{
	if (!(ca < eb_end && ea_end > cb)) return 0;
	if (na == 1 && nb == 1) { // single-exon x single-exon: plain interval intersection
		int s = ca > cb ? ca : cb, e = ea_end < eb_end ? ea_end : eb_end;
		return e > s ? e - s : 0;
	}
	int ia = 0, ib = 0, inter = 0;
	int2 xa = ex[oa], xb = ex[ob];
	while (true) {
		int s0 = ca + xa.x, e0 = ca + xa.y, s1 = cb + xb.x, e1 = cb + xb.y;
		bool adv_a;
		if (s0 < s1) {
			if (e0 < e1) { int o = e0 - s1; inter += o > 0 ? o : 0; adv_a = true; }
			else { inter += e1 - s1; adv_a = false; }
		} else {
			if (e1 < e0) { int o = e1 - s0; inter += o > 0 ? o : 0; adv_a = false; }
			else { inter += e0 - s0; adv_a = true; }
		}
		if (adv_a) { if (++ia >= na) break; xa = ex[oa + ia]; }
		else { if (++ib >= nb) break; xb = ex[ob + ib]; }
	}
	return inter;
}

struct SwHit { // the hit a thread works for
	int sg, cs, ce, gid, cds, rank, nex, offx, weak; uint32_t fl; uint32_t sc;
};
struct SwBest { bool lose; uint32_t best; int j, ov, pid, cds; };

// Thread-per-hit form of one pair, used by k_sweep_slow: partner p (records a/b/c, flags fp, array index pi) of hit t;
// EARLIER: p precedes t in the array.  overlap.c:126-154 (pg_shadow) / 76-87 (pg_flt_ov_isoform).
__device__ __forceinline__ int4 sw_scse(int4 r) { return make_int4(r.y, r.x, r.z, r.w); } // record A -> (seg, cs, ce, pm)

template <int MODE, bool EARLIER>
__device__ __forceinline__ void sw_pair(const SweepView &v, const SwHit &t, SwBest &r, const int4 a, const uint32_t fp, const int4 b, const int4 c, int pi, bool ok)
{
	ok = ok && !(fp & PGA_F_FLT);
	if (v.check_strand) ok = ok && !((fp ^ t.fl) & PGA_F_REV);
	const bool same_gene = b.y == t.gid;
	if (MODE == 2) ok = ok && same_gene;
	const int x = !ok ? 0 : EARLIER ? cds_inter(v.exon, c.z, c.y, a.y, a.z, t.offx, t.nex, t.cs, t.ce)
	                                : cds_inter(v.exon, t.offx, t.nex, t.cs, t.ce, c.z, c.y, a.y, a.z);
	ok = ok && x > 0; // overlap.c:132
	const uint32_t sp = (uint32_t)b.x;
	// "i" of the reference is the later hit of the pair: i loses if (si < sj || (si == sj && rank_i > rank_j))
	const uint32_t s_i = EARLIER ? t.sc : sp, s_j = EARLIER ? sp : t.sc;
	const int rk_i = EARLIER ? t.rank : c.x, rk_j = EARLIER ? c.x : t.rank;
	bool i_loses = s_i < s_j || (s_i == s_j && rk_i > rk_j);
	if (MODE != 2) {
		const int m = t.cds < b.z ? t.cds : b.z;
		// cov_short < min_ov_ratio (overlap.c:134-136).  For the default 0.5 the test is exactly 2x < m: x/m is within
		// 2^-32 of 0.5 only when it equals it, far above double rounding; other ratios take the IEEE division.
		bool too_short;
		if (v.min_ov == 0.5) too_short = 2u * (uint32_t)x < (uint32_t)m;
		else too_short = (double)x / (m > 0 ? m : 1) < v.min_ov;
		ok = ok && (same_gene || !too_short);
		const int wk_p = (int)((fp & PGA_F_WEAK_MASK) >> PGA_F_WEAK_SHIFT);
		const int wk_i = EARLIER ? t.weak : wk_p, wk_j = EARLIER ? wk_p : t.weak;
		i_loses = (!same_gene && wk_i != wk_j) ? wk_i > wk_j : i_loses; // overlap.c:139-147
	}
	const bool t_loses = ok && (EARLIER ? i_loses : !i_loses);
	r.lose = r.lose || t_loses;
	if (MODE == 2) return;
	// dominator = best-scoring winner, first in array order on ties (overlap.c:150,153).  Earlier partners are visited in
	// DEscending index order, so an equal score replaces; later partners in ascending order, so it does not.
	const bool upd = t_loses && (EARLIER ? (sp > 0 && sp >= r.best) : (sp > r.best));
	if (t_loses && sp == r.best && sp > 0) { atomicAdd((unsigned long long *)&v.hz[3], 1ull); hz_note(&v.hz[10], v.hz_list, t.sg); } // hazard H3, rare
	r.best = upd ? sp : r.best, r.j = upd ? pi : r.j, r.ov = upd ? x : r.ov, r.pid = upd ? b.w : r.pid, r.cds = upd ? b.z : r.cds;
}

__device__ __forceinline__ void wave_sync() // LDS hand-over between lanes of ONE wave (the LDS queue of a wave is in order)
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// exclusive prefix sum over the wave of a small count (c < 256), one ballot per bit: no LDS traffic, no cross-lane moves
__device__ __forceinline__ int wave_scan_small(int c, int *total)
{
	int off = 0, tot = 0;
#pragma unroll
	for (int b = 0; b < 8; ++b) {
		const unsigned long long mk = __ballot((c >> b) & 1);
		off += (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u)) << b;
		tot += __popcll(mk) << b;
	}
	*total = tot;
	return off;
}

// The interval-dominance sweep as an LDS pair list.
//
// A workgroup stages SW_TILE consecutive hits (cs order) plus SW_HALO neighbours on each side (36 B/hit, 52 when the C
// records are needed; coalesced 16-byte loads) and after ONE barrier its waves work independently: a wave owns 64 hits
// and looks at a window of SW_HALO more slots on each side.  Because hits are cs-sorted inside a contig, the later
// partners of a hit are a contiguous run; the runs are counted, prefix-summed over the wave and expanded into a list
// of (earlier, later) slot pairs with at least one member among the wave's hits.  The list is evaluated one pair per
// lane (full lanes, every pair once -- a thread-per-hit walk evaluates each pair twice and runs as long as the busiest
// lane).  The outcome reaches the loser as ONE 64-bit LDS atomicMax of (winner's score rank, "lost" bit, inverted
// winner slot): the maximum is the best-scoring winner and, among equals, the first in array order (overlap.c:150).
// Pairs across a wave or tile border are evaluated by both sides, each updating only its own hit: no global atomics,
// no inter-wave synchronisation.  Hits whose partners reach beyond the window, and waves whose list overflows, go
// to a work list for k_sweep_slow.
// MODE 0: pg_shadow(cal_dom_sc=0); 1: pg_shadow(cal_dom_sc=1); 2: pg_flt_ov_isoform
constexpr int SW_HALO = 32, SW_TILE = 256, SW_LDS = SW_TILE + 2 * SW_HALO, SW_WCAP = 512, SW_NW = SW_TILE / 64;

#ifdef PGA_SW_PROFILE // tuning build: s_memtime stamps of lane 0 of every wave at the phase boundaries
#define SW_STAMP(k) do { if (v.prof && (threadIdx.x & 63) == 0) v.prof[((long long)blockIdx.x * SW_NW + (threadIdx.x >> 6)) * 8 + (k)] = clock64(); } while (0)
#else
#define SW_STAMP(k) do { } while (0)
#endif

// epilogue of a hit, overlap.c:157-175.  The hit at index 0 of a genome is never reset (loop starts at 1, overlap.c:108).
template <int MODE>
__device__ __forceinline__ void sw_finish(const SweepView &v, int h, uint32_t fl, bool lose, bool has_dom, int pid_w, int ov, int cds_h, int cds_w, int sori_h, int sori_w)
{
	if (MODE == 2) {
		if (lose) v.flags[h] = fl | PGA_F_ISO_OV;
		return;
	}
	uint32_t nf = (fl & F_HEAD) ? fl : (fl & ~PGA_F_SHADOW);
	if (lose) nf |= PGA_F_SHADOW;
	if (nf != fl) v.flags[h] = nf;
	v.pdom[h] = has_dom ? pid_w : -1;
	if (MODE == 1) {
		int sd = -1;
		if (has_dom) sd = (int32_t)(sori_h * (1.0 - (double)ov / cds_h) + sori_w * ((double)ov / cds_w) + .499); // overlap.c:170
		v.sdom[h] = sd;
	}
}

template <int MODE, bool STAGE_C>
__global__ __launch_bounds__(SW_TILE) void k_sweep(SweepView v)
{
	static_assert(2 * SW_HALO == 64 && SW_NW == 4 && SW_LDS <= 1024 && 64 + 2 * SW_HALO <= 128, "the slots past SW_TILE are staged one array per wave; window-relative slot ids are packed in 7 bits, winner slots in 10");
	constexpr bool STAGE_ORI = MODE == 1 && !STAGE_C; // score_dom needs score_ori: out of the C records when they are staged, else staged alone
	__shared__ int4 sA[SW_LDS + 4], sB[SW_LDS], sC[STAGE_C ? SW_LDS : 1]; // sA: four sentinel slots close the array
	__shared__ uint32_t sF[SW_LDS];
	__shared__ int32_t sOri[STAGE_ORI ? SW_LDS : 1];
	__shared__ uint16_t sPairAll[SW_NW][SW_WCAP]; // (earlier slot - window start) << 7 | (later slot - first own slot): both < 96
	__shared__ unsigned long long sKeyAll[SW_NW][64];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	uint16_t *sPair = sPairAll[wave];
	unsigned long long *sKey = sKeyAll[wave];
	const int tile = blockIdx.x, base = tile * SW_TILE - SW_HALO;
	SW_STAMP(0);
	{
		const int g = base + tid;
		int4 a = make_int4(0, -2, 0, 0), b = make_int4(0, 0, 0, 0), c = b; // slots outside the array: contig -2, filtered
		uint32_t f = PGA_F_FLT;
		int32_t so = 0;
		if (g >= 0 && g < v.n) {
			a = v.A[g], b = v.B[g], f = v.flags[g];
			if (STAGE_C) c = v.C[g];
			if (STAGE_ORI) so = v.sori[g];
		}
		// the 2 * SW_HALO slots past SW_TILE: one array per wave, so that no wave has more to stage than the others
		const int l2 = SW_TILE + lane, g2 = base + l2;
		const bool in2 = lane < 2 * SW_HALO && g2 >= 0 && g2 < v.n;
		if (wave == 0) sA[l2] = in2 ? v.A[g2] : make_int4(0, -2, 0, 0);
		else if (wave == 1) sB[l2] = in2 ? v.B[g2] : make_int4(0, 0, 0, 0);
		else if (wave == 2) sF[l2] = in2 ? v.flags[g2] : PGA_F_FLT;
		else if (STAGE_C) sC[l2] = in2 ? v.C[g2] : make_int4(0, 0, 0, 0);
		else if (STAGE_ORI) sOri[l2] = in2 ? v.sori[g2] : 0;
		sA[tid] = a, sB[tid] = b, sF[tid] = f;
		if (STAGE_C) sC[tid] = c;
		if (STAGE_ORI) sOri[tid] = so;
	}
	if (tid < 4) sA[SW_LDS + tid] = make_int4(0, -2, 0, 0);
	sKey[lane] = 0;
	SW_STAMP(1);
	__syncthreads();
	SW_STAMP(2);
	// ---- from here on every wave is on its own ----
	const int lo = SW_HALO + wave * 64, wend = lo + 64 + SW_HALO; // own slots [lo, lo+64), window [lo-SW_HALO, wend)
	// Later partners of a slot l: the run (l, e) with e = the first slot whose sort key (contig, cs) is not below
	// (contig_l, ce_l); the keys are non-decreasing, so four candidates are tested per round trip to LDS and the tests are
	// independent.  Lane t looks after its own slot and, the first SW_HALO lanes, after a slot of the left context, whose
	// run matters from the wave's first hit on.
	const int l1 = lo + lane, l0 = lo - SW_HALO + (lane & (SW_HALO - 1));
	int m1 = l1 + 1, m0 = lo, c1, c0;
	{
		const int4 a1 = sA[l1], a0 = sA[l0];
		const unsigned long long t1 = (unsigned long long)(uint32_t)a1.y << 32 | (uint32_t)a1.z, t0 = (unsigned long long)(uint32_t)a0.y << 32 | (uint32_t)a0.z;
		bool go1 = !(sF[l1] & PGA_F_FLT), go0 = lane < SW_HALO && !(sF[l0] & PGA_F_FLT);
		const int f1 = m1;
		while (go0 || go1) {
			unsigned long long q1[4], q0[4];
#pragma unroll
			for (int u = 0; u < 4; ++u) q1[u] = *(const unsigned long long *)&sA[m1 + u], q0[u] = *(const unsigned long long *)&sA[m0 + u];
			int n1 = 0, n0 = 0;
#pragma unroll
			for (int u = 0; u < 4; ++u) n1 += q1[u] < t1 ? 1 : 0, n0 += q0[u] < t0 ? 1 : 0;
			n1 = go1 ? n1 : 0, n0 = go0 ? n0 : 0;
			m1 += n1, m0 += n0;
			go1 = n1 == 4 && m1 < wend, go0 = n0 == 4 && m0 < wend;
		}
		c1 = (m1 < wend ? m1 : wend) - f1, c0 = (m0 < wend ? m0 : wend) - lo;
	}
	SW_STAMP(3);
	// The pair list, k-th partners of all slots together: their places follow from one ballot, no prefix sum needed.
	int tot = 0;
#pragma nounroll
	for (int k = 0;; ++k) {
		const unsigned long long mk = __ballot(c0 > k);
		if (mk == 0) break;
		const int at = tot + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
		if (c0 > k && at < SW_WCAP) sPair[at] = (uint16_t)((lane & (SW_HALO - 1)) << 7 | k);
		tot += __popcll(mk);
	}
#pragma nounroll
	for (int k = 0;; ++k) {
		const unsigned long long mk = __ballot(c1 > k);
		if (mk == 0) break;
		const int at = tot + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
		if (c1 > k && at < SW_WCAP) sPair[at] = (uint16_t)((SW_HALO + lane) << 7 | (lane + 1 + k));
		tot += __popcll(mk);
	}
	const bool listed = tot <= SW_WCAP; // wave-uniform
	wave_sync();
	SW_STAMP(4);
	if (listed) {
		// one pair per lane: slot l precedes slot m in the array (l is "j", m is "i" of overlap.c:126-154 / 76-87)
		for (int p = lane; p < tot; p += 64) {
			const uint32_t w = sPair[p];
			const int l = lo - SW_HALO + (int)(w >> 7), m = lo + (int)(w & 127u);
			const uint32_t fj = sF[l], fi = sF[m];
			const int csj = sA[l].x, cej = sA[l].z, csi = sA[m].x, cei = sA[m].z;
			const int4 bj = sB[l], bi = sB[m]; // {rk, gid, cds, pid}
			bool ok = !((fj | fi) & PGA_F_FLT);
			if (v.check_strand) ok = ok && !((fj ^ fi) & PGA_F_REV);
			const bool same_gene = bj.y == bi.y;
			if (MODE == 2) ok = ok && same_gene;
			int x;
			{
				const int s0 = csj > csi ? csj : csi, e0 = cej < cei ? cej : cei;
				x = e0 > s0 ? e0 - s0 : 0; // single-exon x single-exon: the CDS intersection is the interval intersection
			}
			bool i_loses = (uint32_t)bi.x < (uint32_t)bj.x;
			// the C records only when a pair of the wave needs them: multi-exon hits, or two hits with the same score key
			// (the same protein with the same score) whose order the rank decides
			const bool multi = ok && ((fj | fi) & F_MULTI), tie = ok && bi.x == bj.x;
			if (__ballot(multi || tie)) {
				if (multi || tie) {
					const int4 cj = STAGE_C ? sC[l] : v.C[base + l], ci = STAGE_C ? sC[m] : v.C[base + m]; // {rank, n_exon, off_exon, score_ori}
					if (multi) x = cds_inter(v.exon, cj.z, cj.y, csj, cej, ci.z, ci.y, csi, cei);
					if (tie) i_loses = ci.x > cj.x; // rank_i > rank_j
				}
			}
			ok = ok && x > 0; // overlap.c:132
			if (MODE != 2) {
				const int mn = bi.z < bj.z ? bi.z : bj.z;
				// cov_short < min_ov_ratio (overlap.c:134-136).  For the default 0.5 the test is exactly 2x < min(cds): x/m is
				// within 2^-32 of 0.5 only when it equals it, far above double rounding; other ratios take the IEEE division.
				bool too_short;
				if (v.min_ov == 0.5) too_short = 2u * (uint32_t)x < (uint32_t)mn;
				else too_short = (double)x / (mn > 0 ? mn : 1) < v.min_ov;
				ok = ok && (same_gene || !too_short);
				const uint32_t wk_i = fi & PGA_F_WEAK_MASK, wk_j = fj & PGA_F_WEAK_MASK;
				i_loses = (!same_gene && wk_i != wk_j) ? wk_i > wk_j : i_loses; // overlap.c:139-147
			}
			const int L = i_loses ? m : l, W = i_loses ? l : m, Lt = L - lo;
			if (ok && (unsigned)Lt < 64u) {
				const uint32_t rw = (uint32_t)(i_loses ? bj.x : bi.x);
				const unsigned long long key = (unsigned long long)rw << 32 | 0x80000000u | (uint32_t)(1023 - W);
				const unsigned long long old = atomicMax(&sKey[Lt], key);
				if (MODE != 2 && rw != 0 && (uint32_t)(old >> 32) == rw) { atomicAdd((unsigned long long *)&v.hz[3], 1ull); hz_note(&v.hz[10], v.hz_list, sA[L].y); } // hazard H3: two winners with one key
			}
		}
		wave_sync();
	}
	SW_STAMP(5);
	SW_STAMP(6);
	{
		const int h = tile * SW_TILE + wave * 64 + lane, lh = lo + lane;
		const uint32_t fl = sF[lh];
		if (h < v.n && !(fl & PGA_F_FLT)) { // filtered hits keep stale shadow/pid_dom (overlap.c:112)
			const int4 a = sA[lh]; // {cs, seg, ce, pm}
			// partners outside the window?  (pm = running max of ce is non-decreasing inside a contig)
			const int4 w0 = sA[lo - SW_HALO], w1 = sA[wend - 1];
			const bool open = (w0.y == a.y && w0.w > a.x) || (w1.y == a.y && w1.x < a.z);
			if (!listed || open) {
				const unsigned long long at = atomicAdd((unsigned long long *)v.slow_cnt, 1ull);
				v.slow_list[at] = h;
			} else {
				const unsigned long long key = sKey[lane];
				const bool lose = key != 0, has_dom = MODE != 2 && (key >> 32) != 0;
				int pid_w = -1, ov = 0, cds_w = 1, so_w = 0, so_h = 0, cds_h = 1;
				if (has_dom) {
					const int W = 1023 - (int)(key & 1023u);
					const int4 bw = sB[W];
					pid_w = bw.w, cds_w = bw.z;
					if (MODE == 1) {
						const int4 aw = sA[W], cw = STAGE_C ? sC[W] : make_int4(0, 1, 0, sOri[W]), c2 = STAGE_C ? sC[lh] : make_int4(0, 1, 0, sOri[lh]);
						so_w = cw.w, so_h = c2.w, cds_h = sB[lh].z;
						const int s0 = aw.x > a.x ? aw.x : a.x, e0 = aw.z < a.z ? aw.z : a.z;
						ov = e0 > s0 ? e0 - s0 : 0;
						if (STAGE_C && ((fl | sF[W]) & F_MULTI)) { // the earlier hit goes first, as in the pair evaluation
							const bool wf = W < lh;
							ov = cds_inter(v.exon, wf ? cw.z : c2.z, wf ? cw.y : c2.y, wf ? aw.x : a.x, wf ? aw.z : a.z,
							               wf ? c2.z : cw.z, wf ? c2.y : cw.y, wf ? a.x : aw.x, wf ? a.z : aw.z);
						}
					}
				}
				sw_finish<MODE>(v, h, fl, lose, has_dom, pid_w, ov, cds_h, cds_w, so_h, so_w);
			}
		}
	}
	SW_STAMP(7);
}

// The rare hits k_sweep could not finish inside its LDS window: one thread per listed hit walks all its partners in
// global memory, in both directions (the plain thread-per-hit formulation of the sweep).
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_sweep_slow(SweepView v, long long *next_cnt)
{
	const long long n_slow = *v.slow_cnt;
	if (blockIdx.x == 0 && threadIdx.x == 0) *next_cnt = 0; // the counter the NEXT sweep will use (ping-pong; nobody reads it now)
	for (long long q = blockIdx.x * (long long)BLOCK + threadIdx.x; q < n_slow; q += (long long)gridDim.x * BLOCK) {
		const int h = v.slow_list[q];
		const uint32_t fl = v.flags[h];
		SwHit t;
		const int4 ch = v.C[h];
		{
			const int4 a = sw_scse(v.A[h]), b = v.B[h];
			t.sg = a.x, t.cs = a.y, t.ce = a.z, t.gid = b.y, t.cds = b.z, t.rank = ch.x, t.nex = ch.y, t.offx = ch.z;
			t.weak = (int)((fl & PGA_F_WEAK_MASK) >> PGA_F_WEAK_SHIFT), t.fl = fl;
			t.sc = (uint32_t)b.x;
		}
		SwBest r = { false, 0, -1, 0, -1, 0 };
		// partners before h: every j with ce_j > cs_h.  pm (running max of ce) is non-decreasing inside a contig, so the
		// walk stops at the first j whose pm is <= cs_h.
		for (int j = h - 1; j >= 0; --j) {
			const int4 a = sw_scse(v.A[j]);
			if (a.x != t.sg || a.w <= t.cs) break;
			sw_pair<MODE, true>(v, t, r, a, v.flags[j], v.B[j], v.C[j], j, a.z > t.cs);
		}
		// partners after h: every i with cs_i < ce_h
		for (int i = h + 1; i < v.n; ++i) {
			const int4 a = sw_scse(v.A[i]);
			if (a.x != t.sg || a.y >= t.ce) break;
			sw_pair<MODE, false>(v, t, r, a, v.flags[i], v.B[i], v.C[i], i, true);
		}
		sw_finish<MODE>(v, h, fl, r.lose, r.best > 0, r.pid, r.ov, t.cds, r.cds, ch.w, MODE == 1 && r.best > 0 ? v.C[r.j].w : 0);
	}
}

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

// read.c:249-253
__global__ __launch_bounds__(BLOCK) void k_ingest_reset(uint32_t *flags, int32_t *pdom, int32_t *pdom0, int n)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	pdom0[h] = pdom[h];
	pdom[h] = -1;
	flags[h] &= ~PGA_F_SHADOW;
}

// tail of pg_flt_ov_isoform (overlap.c:89-91) + first loop of pg_flt_chain_shadow (hit.c:136-138)
__global__ __launch_bounds__(BLOCK) void k_iso_apply(uint32_t *flags, const int32_t *gnm, const int32_t *pid, int n, int P, int32_t *tiso, int32_t *stats)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	uint32_t f = flags[h];
	if (f & PGA_F_ISO_OV) {
		flags[h] = f | PGA_F_FLT;
		atomicAdd(&stats[gnm[h] * 4 + 1], 1);
	} else tiso[(int64_t)gnm[h] * P + pid[h]] = 0;
}

// second loop of pg_flt_chain_shadow (hit.c:139-143)
__global__ __launch_bounds__(BLOCK) void k_chain(uint32_t *flags, const int32_t *gnm, const int32_t *pdom0, int n, int P, const int32_t *tiso, int32_t *stats)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	int p0 = pdom0[h];
	if (p0 >= 0 && tiso[(int64_t)gnm[h] * P + p0]) {
		flags[h] |= PGA_F_FLT | PGA_F_CHAIN;
		atomicAdd(&stats[gnm[h] * 4 + 2], 1);
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

__global__ __launch_bounds__(BLOCK) void k_subopt2(uint32_t *flags, const int32_t *gnm, const int32_t *gid, const int32_t *pid, const int32_t *goff, int n, int Q,
                                                     const unsigned long long *tbest, int32_t *stats)
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
		best_pid = pid[goff[g] + (int)pos];
	}
	if (pid[h] != best_pid) {
		flags[h] = f | PGA_F_FLT | PGA_F_ISO_SUB;
		atomicAdd(&stats[g * 4 + 3], 1);
	}
}

// ------------------------------------------------------------------------------------------------
// stage B (hit.c:153-247)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_post_part(const uint32_t *flags, const int32_t *pid, const int32_t *rank, const int32_t *sori, const int32_t *sadj,
                                                       const int32_t *nex, int n, int P, int32_t *max_ori, unsigned long long *sums)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	int p = pid[h];
	atomicMax(&max_ori[p], sori[h]);
	if (rank[h] == 0 && !(flags[h] & PGA_F_FLT)) {
		int w = nex[h] == 1 ? 0 : 1;
		atomicAdd(&sums[p], (unsigned long long)(long long)sadj[h]);
		atomicAdd(&sums[(int64_t)P + p], 1ull);
		atomicAdd(&sums[(int64_t)(2 + w) * P + p], 1ull);
		atomicAdd(&sums[(int64_t)(4 + w) * P + p], (unsigned long long)(long long)sori[h]);
	}
}

__global__ __launch_bounds__(BLOCK) void k_post_apply(uint32_t *flags, const int32_t *pid, const int32_t *nex, int32_t *sdom, int n,
                                                        const int32_t *max_ori, const uint8_t *rep, const uint8_t *pj, int64_t *cnt)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	int p = pid[h];
	int mo = max_ori[p];
	if (sdom[h] > mo) sdom[h] = mo; // hit.c:243-244
	uint32_t f = flags[h], nf = rep[p] ? (f | PGA_F_REP) : (f & ~PGA_F_REP);
	if (!(f & (PGA_F_FLT | PGA_F_PSEUDO)) && nex[h] == 1 && pj[p]) { // hit.c:175-182
		nf |= PGA_F_PSEUDO;
		atomicAdd((unsigned long long *)cnt, 1ull);
	}
	if (nf != f) flags[h] = nf;
}

__global__ __launch_bounds__(BLOCK) void k_set_filter(uint32_t *flags, int n, int which) // pgpriv.h:109-116
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	uint32_t f = flags[h];
	bool hit = which == PGA_FLT_PSEUDO ? (f & PGA_F_PSEUDO) != 0
	         : which == PGA_FLT_VTX0 ? (f & PGA_F_VTX) == 0
	         : which == PGA_FLT_WEAK2 ? ((f & PGA_F_WEAK_MASK) >> PGA_F_WEAK_SHIFT) == 2
	         : (f & PGA_F_SHADOW) != 0;
	if (hit && !(f & PGA_F_FLT)) flags[h] = f | PGA_F_FLT;
}

// ------------------------------------------------------------------------------------------------
// pg_gen_vtx, per-genome part (vertex.c:28-51)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_vtx1(const uint32_t *flags, const int32_t *gnm, const int32_t *gid, const int32_t *rank, const int32_t *pdom,
                                                  int n, int Q, int32_t *cnt, uint32_t *dombits, int64_t words_per_genome, int64_t *dcnt)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	uint32_t f = flags[h];
	if ((f & PGA_F_FLT) || rank[h] != 0) return;
	int g = gid[h];
	if (f & PGA_F_SHADOW) {
		if (pdom[h] < 0) atomicAdd((unsigned long long *)&dcnt[3], 1ull); // vertex.c:38
		atomicAdd(&cnt[Q + g], 1);
	} else {
		atomicAdd(&cnt[g], 1);
		uint32_t old = atomicOr(&dombits[(int64_t)gnm[h] * words_per_genome + (g >> 5)], 1u << (g & 31));
		if (old & (1u << (g & 31))) atomicAdd((unsigned long long *)&dcnt[3], 1ull); // two rank-0 hits of one gene: cannot happen after hit.c:107-128
	}
}

// Fold of the (genome, sub gene, dom gene) relation into one genome bitset per (sub, dom) pair, the form the host greedy
// consumes (vertex.c:60-80 marks cell (genome, dom) for every genome of the pair).  A sub gene has very few distinct dom
// genes, so each gene owns VTX_K slots: a slot is claimed for a dom gene with atomicCAS, the genome bit is an atomicOr.
// A gene with more than VTX_K dom genes spills single-genome records into an overflow area.
constexpr int VTX_K = 8;

__global__ __launch_bounds__(BLOCK) void k_vtx_fold(const uint32_t *flags, const int32_t *gnm, const int32_t *gid, const int32_t *rank, const int32_t *pdom,
                                                      const int32_t *prot_gid, const int32_t *ggl, int n, const uint32_t *dombits, int64_t words_per_genome,
                                                      int32_t *dom_tab, unsigned long long *bits, int nw, unsigned long long *ovf, long long ovf_cap, int64_t *dcnt)
{
	const int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	const uint32_t f = flags[h];
	if ((f & PGA_F_FLT) || rank[h] != 0 || !(f & PGA_F_SHADOW) || pdom[h] < 0) return;
	const int j = gnm[h], D = prot_gid[pdom[h]], g = gid[h];
	if (!(dombits[(int64_t)j * words_per_genome + (D >> 5)] >> (D & 31) & 1u)) return; // dom is not dominant in this genome: the greedy never looks
	const int jg = ggl[j];
	int k = 0;
	for (; k < VTX_K; ++k) {
		int32_t *p = &dom_tab[(int64_t)g * VTX_K + k];
		int cur = *(volatile int32_t *)p;
		if (cur < 0) cur = atomicCAS(p, -1, D), cur = cur < 0 ? D : cur;
		if (cur == D) break;
	}
	if (k < VTX_K) {
		atomicOr(&bits[((int64_t)g * VTX_K + k) * nw + (jg >> 6)], 1ull << (jg & 63));
	} else {
		const long long at = (long long)atomicAdd((unsigned long long *)&dcnt[0], 1ull);
		if (at < ovf_cap) {
			unsigned long long *r = ovf + at * (1 + nw);
			r[0] = (unsigned long long)g << 20 | (unsigned long long)D;
			for (int w = 0; w < nw; ++w) r[1 + w] = w == (jg >> 6) ? 1ull << (jg & 63) : 0ull;
		}
	}
}

struct InDomSet { const int32_t *tab; __device__ __forceinline__ I32 operator()(int64_t i) const { return I32{tab[i] >= 0 ? 1 : 0}; } };

// slot -> record: key (sub << 20 | dom), then the genome words; also mails the record count (dcnt[10]) and the counters to the host
__global__ __launch_bounds__(BLOCK) void k_vtx_compact(const int32_t *dom_tab, const int32_t *slot, int64_t n_slot, const unsigned long long *bits, int nw,
                                                         unsigned long long *out, int64_t *dcnt, int64_t *host_box)
{
	const int64_t s = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (s >= n_slot) return;
	const int D = dom_tab[s];
	if (D >= 0) {
		unsigned long long *r = out + (int64_t)slot[s] * (1 + nw);
		r[0] = (unsigned long long)(s / VTX_K) << 20 | (unsigned long long)D;
		for (int w = 0; w < nw; ++w) r[1 + w] = bits[s * nw + w];
	}
	if (s == n_slot - 1) {
		dcnt[10] = slot[s] + (D >= 0 ? 1 : 0);
		__threadfence();
		for (int t = 0; t < 16; ++t) host_box[t] = dcnt[t];
	}
}

__global__ __launch_bounds__(BLOCK) void k_flag_vtx(uint32_t *flags, const int32_t *gid, int n, const int32_t *g2s) // graph.c:61-69
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	uint32_t f = flags[h], nf = g2s[gid[h]] >= 0 ? (f | PGA_F_VTX) : (f & ~PGA_F_VTX);
	if (nf != f) flags[h] = nf;
}

// ------------------------------------------------------------------------------------------------
// pg_gen_arc, per-genome part (graph.c:97-146)
// ------------------------------------------------------------------------------------------------
constexpr int SEGCNT_COPIES = 64;
__global__ __launch_bounds__(BLOCK) void k_segcnt_sum(int32_t *seg_cnt, int n2s)
{
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n2s) return;
	int t = 0;
	for (int k = 0; k < SEGCNT_COPIES; ++k) t += seg_cnt[(int64_t)k * n2s + i];
	seg_cnt[i] = t;
}

// walkable = !flt && !shadow; val[y] = y if the y-th hit in cm order is walkable else -1
__global__ __launch_bounds__(BLOCK) void k_walk_mark(const uint32_t *flags, const int32_t *yperm, int n, int32_t *val)
{
	int y = blockIdx.x * BLOCK + threadIdx.x;
	if (y >= n) return;
	val[y] = (flags[yperm[y]] & (PGA_F_FLT | PGA_F_SHADOW)) ? -1 : y;
}

struct InWalk { const int32_t *val; __device__ __forceinline__ I32 operator()(int64_t i) const { return I32{val[i]}; } };
struct OutPrev { int32_t *prev; __device__ __forceinline__ void operator()(int64_t i, I32, I32 ex) const { prev[i] = ex.v; } };

// Static per-hit fields in Y (cm) order, packed once per run: the arc kernels walk the hits in that order and would otherwise
// gather every field through yperm.   YA = {seg, gid, genome, cm}   YB = {score_ori, score_dom, gene of pid_dom0's protein
// (-1: none), X position << 1 | rev}
__global__ __launch_bounds__(BLOCK) void k_pack_yrec(const int32_t *yperm, const int32_t *seg, const int32_t *gid, const int32_t *gnm, const int32_t *cm,
                                                       const int32_t *sori, const int32_t *sdom, const int32_t *pdom0, const int32_t *prot_gid, const uint32_t *flags,
                                                       int n, int4 *YA, int4 *YB)
{
	int y = blockIdx.x * BLOCK + threadIdx.x;
	if (y >= n) return;
	const int a = yperm[y], p0 = pdom0[a];
	YA[y] = make_int4(seg[a], gid[a], gnm[a], cm[a]);
	YB[y] = make_int4(sori[a], sdom[a], p0 < 0 ? -1 : prot_gid[p0], a << 1 | (flags[a] & PGA_F_REV ? 1 : 0));
}

// has_arc[y] = 1 if walkable y has a walkable predecessor on the same contig; also per-segment counts
// (graph.c:113,125-126) and hazard H2a (equal cm of two consecutive walkable hits)
__global__ __launch_bounds__(BLOCK) void k_arc_flag(const int32_t *val, const int32_t *prev, const int4 *YA, const int32_t *g2s, int n, int S, int32_t *has, int32_t *seg_cnt,
                                                      uint32_t *seen, int64_t words_per_genome, int64_t *dcnt, int32_t *hz_list)
{
	int y = blockIdx.x * BLOCK + threadIdx.x;
	if (y >= n) return;
	int out = 0;
	if (val[y] >= 0) {
		const int4 ra = YA[y];
		const int sid = g2s[ra.y];
		if (sid < 0) atomicAdd((unsigned long long *)&dcnt[3], 1ull); // graph.c:111
		else {
			int32_t *copy = seg_cnt + (int64_t)(blockIdx.x & (SEGCNT_COPIES - 1)) * 2 * S; // 64 copies: 64x less contention per address
			atomicAdd(&copy[S + sid], 1);
			uint32_t old = atomicOr(&seen[(int64_t)ra.z * words_per_genome + (sid >> 5)], 1u << (sid & 31));
			if (!(old >> (sid & 31) & 1u)) atomicAdd(&copy[sid], 1);
		}
		const int p = prev[y];
		if (p >= 0) {
			const int4 rb = YA[p];
			if (rb.x == ra.x) {
				out = 1;
				if (rb.w == ra.w) { atomicAdd((unsigned long long *)&dcnt[5], 1ull); hz_note(&dcnt[14], hz_list, ra.x); }
			}
		}
	}
	has[y] = out;
}

__device__ __forceinline__ int arc_score(const int4 yb, int ori, const int32_t *g2s)
{ // pg_get_score, graph.c:82-85: score_ori unless the dominator's gene is not a vertex and score_dom is at least as large
	return (ori || yb.x > yb.y || yb.z < 0 || g2s[yb.z] >= 0) ? yb.x : yb.y;
}

struct ArcEmit {
	const int32_t *has, *slot, *prev; const int4 *YA, *YB; const int32_t *g2s;
	uint64_t *key; uint32_t *idx; int4 *pay; // payload {dist, s1, s2, genome}
	int n, ori, vbits;
};

__global__ __launch_bounds__(BLOCK) void k_arc_emit(ArcEmit e)
{
	int y = blockIdx.x * BLOCK + threadIdx.x;
	if (y >= e.n || !e.has[y]) return;
	const int p = e.prev[y];
	const int4 aA = e.YA[y], bA = e.YA[p], aB = e.YB[y], bB = e.YB[p];
	uint32_t w = (uint32_t)e.g2s[aA.y] << 1 | (uint32_t)(aB.w & 1);
	uint32_t v = (uint32_t)e.g2s[bA.y] << 1 | (uint32_t)(bB.w & 1);
	int sa = arc_score(aB, e.ori, e.g2s);
	int sb = arc_score(bB, e.ori, e.g2s);
	int d = aA.w - bA.w, g = aA.z;
	int64_t o = (int64_t)e.slot[y] * 2;
	e.key[o] = (uint64_t)v << e.vbits | w;           e.idx[o] = (uint32_t)o;         // v -> w      (graph.c:117)
	e.pay[o] = make_int4(d, sb, sa, g);
	e.key[o + 1] = (uint64_t)(w ^ 1) << e.vbits | (v ^ 1); e.idx[o + 1] = (uint32_t)(o + 1); // w^1 -> v^1 (graph.c:119)
	e.pay[o + 1] = make_int4(d, sa, sb, g);
}

__global__ __launch_bounds__(BLOCK) void k_arc_gather(const uint32_t *idx, int64_t m, const int4 *pay, int4 *opay)
{
	int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i < m) opay[i] = pay[idx[i]]; // one random 16-byte read per temp arc
}

__global__ __launch_bounds__(BLOCK) void k_arc_head(const uint64_t *key, int64_t m, int32_t *head)
{
	int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i >= m) return;
	head[i] = (i == 0 || key[i] != key[i - 1]) ? 1 : 0;
}

// Two-level collapse of the sorted temp arcs (graph.c:128-175).  Equal keys are adjacent and, inside one key,
// grouped by genome (stable sort of a genome-major emission).
// level 1: the first element of every (key, genome) run collapses its run -- almost always a single element --
//          into (n, rounded mean dist * n, max s1, max s2) stored at its own position; other positions hold zeros;
// level 2: one wave per distinct key sums those records over the key's run with coalesced strided reads.
__global__ __launch_bounds__(BLOCK) void k_arc_l1(const uint64_t *key, int64_t m, const int4 *pay, const int32_t *head, const int32_t *slot, int32_t *run_start,
                                                    int32_t *o_n, uint64_t *o_dn, int32_t *o_s1, int32_t *o_s2)
{
	int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i >= m) return;
	const uint64_t k = key[i];
	const int4 p = pay[i];
	const int g = p.w;
	if (head[i]) run_start[slot[i]] = (int32_t)i;
	if (i > 0 && key[i - 1] == k && pay[i - 1].w == g) { o_n[i] = 0, o_dn[i] = 0, o_s1[i] = 0, o_s2[i] = 0; return; }
	int n = 1, m1 = p.y, m2 = p.z;
	uint64_t sd = (uint64_t)(int64_t)p.x;
	for (int64_t j = i + 1; j < m && key[j] == k; ++j) { // almost always empty: one adjacency per (arc, genome)
		const int4 q = pay[j];
		if (q.w != g) break;
		sd += (uint64_t)(int64_t)q.x;
		m1 = m1 > q.y ? m1 : q.y;
		m2 = m2 > q.z ? m2 : q.z;
		++n;
	}
	const int dg = (int32_t)((double)sd / n + .499); // graph.c:141
	o_n[i] = n, o_dn[i] = (uint64_t)(int64_t)dg * (uint64_t)n, o_s1[i] = m1, o_s2[i] = m2;
}

__global__ __launch_bounds__(BLOCK) void k_arc_l2(const uint64_t *key, int64_t m, int64_t n_run, const int32_t *run_start, const int32_t *c_n, const uint64_t *c_dn,
                                                    const int32_t *c_s1, const int32_t *c_s2, int vbits, pga_arc_part_t *out)
{
	const int64_t w = (int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	if (w >= n_run) return;
	const int64_t st = run_start[w], en = w + 1 < n_run ? run_start[w + 1] : m;
	int ng = 0, tot = 0;
	uint64_t sd = 0;
	int64_t a1 = 0, a2 = 0;
	for (int64_t j = st + lane; j < en; j += WAVE) {
		const int n = c_n[j];
		ng += n > 0, tot += n, sd += c_dn[j], a1 += c_s1[j], a2 += c_s2[j];
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		ng += __shfl_xor(ng, o, WAVE), tot += __shfl_xor(tot, o, WAVE);
		sd += (uint64_t)__shfl_xor((long long)sd, o, WAVE), a1 += __shfl_xor((long long)a1, o, WAVE), a2 += __shfl_xor((long long)a2, o, WAVE);
	}
	if (lane == 0) {
		const uint64_t k = key[st];
		pga_arc_part_t r;
		r.x = (k >> vbits) << 32 | (k & ((1ull << vbits) - 1));
		r.n_genome = ng, r.tot_cnt = tot, r.sum_dist = sd, r.sum_s1 = a1, r.sum_s2 = a2;
		out[w] = r;
	}
}


// ------------------------------------------------------------------------------------------------
// cross-shard merge of arc tables (after the all-gather): gather valid entries, sort by x, wave-per-run sums
// ------------------------------------------------------------------------------------------------
// Every rank's table arrives sorted by x with unique keys, so the merged order needs no sort: the place of an entry is
// the number of entries before it in all the tables (binary searches; equal keys keep rank order).
struct MergeLists { int32_t W; int64_t slot_sz; const int64_t *off; }; // off[r] = entries of ranks < r, off[W] = total

__device__ __forceinline__ int64_t mg_bound(const pga_arc_part_t *a, int64_t n, uint64_t x, bool upper)
{
	int64_t lo = 0, hi = n;
	while (lo < hi) {
		const int64_t mid = (lo + hi) >> 1;
		const uint64_t y = a[mid].x;
		if (upper ? y <= x : y < x) lo = mid + 1; else hi = mid;
	}
	return lo;
}

__global__ __launch_bounds__(BLOCK) void k_mg_rank(const pga_arc_part_t *g, MergeLists L, uint64_t *key, uint32_t *val)
{
	const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i >= L.off[L.W]) return;
	int r = 0;
	while (L.off[r + 1] <= i) ++r; // the table entry i belongs to (W is small)
	const int64_t k = i - L.off[r], src = r * L.slot_sz + k;
	const uint64_t x = g[src].x;
	int64_t pos = k;
	for (int q = 0; q < L.W; ++q)
		if (q != r) pos += mg_bound(g + q * L.slot_sz, L.off[q + 1] - L.off[q], x, q < r);
	key[pos] = x, val[pos] = (uint32_t)src;
}

struct InMgHead { const uint64_t *key; __device__ __forceinline__ I32 operator()(int64_t i) const { return I32{(i == 0 || key[i] != key[i - 1]) ? 1 : 0}; } };

__global__ __launch_bounds__(BLOCK) void k_mg_count(const uint64_t *key, const int32_t *slot, int64_t m, int64_t *box) // number of distinct keys
{
	if (blockIdx.x == 0 && threadIdx.x == 0) *box = slot[m - 1] + ((m == 1 || key[m - 1] != key[m - 2]) ? 1 : 0); // slot = exclusive count of run heads
}

__global__ __launch_bounds__(BLOCK) void k_mg_runstart(const uint64_t *key, const int32_t *slot, int64_t m, int32_t *run_start)
{
	int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i < m && (i == 0 || key[i] != key[i - 1])) run_start[slot[i]] = (int32_t)i;
}

__global__ __launch_bounds__(BLOCK) void k_mg_sum(const pga_arc_part_t *g, const uint32_t *val, int64_t m, int64_t n_run, const int32_t *run_start, pga_arc_part_t *out)
{
	const int64_t w = (int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	if (w >= n_run) return;
	const int64_t st = run_start[w], en = w + 1 < n_run ? run_start[w + 1] : m;
	int ng = 0, tot = 0;
	uint64_t sd = 0, x = 0;
	int64_t a1 = 0, a2 = 0;
	for (int64_t j = st + lane; j < en; j += WAVE) {
		const pga_arc_part_t p = g[val[j]];
		x = p.x, ng += p.n_genome, tot += p.tot_cnt, sd += p.sum_dist, a1 += p.sum_s1, a2 += p.sum_s2;
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		ng += __shfl_xor(ng, o, WAVE), tot += __shfl_xor(tot, o, WAVE);
		sd += (uint64_t)__shfl_xor((long long)sd, o, WAVE), a1 += __shfl_xor((long long)a1, o, WAVE), a2 += __shfl_xor((long long)a2, o, WAVE);
	}
	if (lane == 0) { // lane 0 always owns element st
		pga_arc_part_t r;
		r.x = x, r.n_genome = ng, r.tot_cnt = tot, r.sum_dist = sd, r.sum_s1 = a1, r.sum_s2 = a2;
		out[w] = r;
	}
}

// ------------------------------------------------------------------------------------------------
// branch.c on device: pg_gen_rep_pos (6-29), pg_n_local (31-46), pg_mark_branch_flt_hit (108-145)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_walk_x(const uint32_t *flags, int n, int32_t *wk)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	wk[h] = (flags[h] & (PGA_F_FLT | PGA_F_SHADOW)) ? 0 : 1;
}

// the last walkable hit of a gene in array order wins (branch.c:22-23 overwrite)
__global__ __launch_bounds__(BLOCK) void k_rep_last(const int32_t *wk, const int32_t *gnm, const int32_t *gid, int n, int GL, int32_t *rp_pos)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n || !wk[h]) return;
	atomicMax(&rp_pos[(int64_t)gid[h] * GL + gnm[h]], h + 1);
}

// Position record of (gene, genome): {contig, rank among the walkable hits of the genome, cm}.  COMPACT (every genome has
// < 4096 contigs and < 2^20 hits, decided once in create): 8 bytes {cm, local contig << 20 | rank}, half the L2 traffic of
// pg_n_local, which reads two records per (pair, genome); otherwise 16 bytes {global contig, rank, cm, 0}.  Absent: -1.
template <bool COMPACT>
__global__ __launch_bounds__(BLOCK) void k_rep_fill(const int32_t *rp_pos, int64_t n_ent, int GL, const int32_t *seg, const int32_t *cm, const int32_t *rx,
                                                      const int32_t *goff, const int32_t *ctg_base, void *rp_out)
{
	int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (e >= n_ent) return;
	const int p = rp_pos[e];
	if (COMPACT) {
		int2 *rp = (int2 *)rp_out;
		if (p == 0) { rp[e] = make_int2(0, -1); return; }
		const int h = p - 1, j = (int)(e % GL);
		rp[e] = make_int2(cm[h], (seg[h] - ctg_base[j]) << 20 | (rx[h] - rx[goff[j]]));
	} else {
		int4 *rp = (int4 *)rp_out;
		if (p == 0) { rp[e] = make_int4(-1, 0, 0, 0); return; }
		const int h = p - 1, j = (int)(e % GL);
		rp[e] = make_int4(seg[h], rx[h] - rx[goff[j]], cm[h], 0);
	}
}

// one wave per gene pair, lanes over the local genomes (branch.c:31-46); the count is a popcount of ballots: no
// cross-lane reduction
template <bool COMPACT>
__global__ __launch_bounds__(BLOCK) void k_n_local(const int32_t *pairs, int64_t n_pair, int GL, const void *rp_in,
                                                     int local_dist, int local_count, int frag_mode, int32_t *cnt)
{
	const int64_t k = (int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	if (k >= n_pair) return;
	const int64_t g1 = (int64_t)pairs[2 * k] * GL, g2 = (int64_t)pairs[2 * k + 1] * GL;
	int c = 0;
	for (int j0 = 0; j0 < GL; j0 += WAVE) {
		const int j = j0 + lane;
		bool hit = false;
		if (j < GL) {
			if (COMPACT) {
				const int2 a = ((const int2 *)rp_in)[g1 + j], b = ((const int2 *)rp_in)[g2 + j];
				const int64_t d = (int64_t)a.x - (int64_t)b.x;
				const int cc = (a.y & 0xfffff) - (b.y & 0xfffff);
				hit = (a.y | b.y) >= 0 && (frag_mode || ((a.y ^ b.y) >> 20) == 0) &&
				      ((d >= -(int64_t)local_dist && d <= local_dist) || (cc >= -local_count && cc <= local_count));
			} else {
				const int4 a = ((const int4 *)rp_in)[g1 + j], b = ((const int4 *)rp_in)[g2 + j];
				const int64_t d = (int64_t)a.z - (int64_t)b.z;
				const int cc = a.y - b.y;
				hit = a.x >= 0 && b.x >= 0 && (frag_mode || a.x == b.x) &&
				      ((d >= -(int64_t)local_dist && d <= local_dist) || (cc >= -local_count && cc <= local_count));
			}
		}
		c += __popcll(__ballot(hit));
	}
	if (lane == 0) cnt[k] = c;
}

// ------------------------------------------------------------------------------------------------
// pg_mark_branch_flt_arc (branch.c:48-106) on the arc table: one thread per oriented vertex
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_br_prep(const uint64_t *ax, int64_t n_arc, const int32_t *seg_gid, int32_t *agid, int32_t *vs, int32_t *ve)
{
	int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n_arc) return;
	const uint64_t x = ax[i];
	const uint32_t v = (uint32_t)(x >> 32);
	agid[i] = seg_gid[(uint32_t)x >> 1];
	if (i == 0 || (uint32_t)(ax[i - 1] >> 32) != v) vs[v] = (int32_t)i;
	if (i == n_arc - 1 || (uint32_t)(ax[i + 1] >> 32) != v) ve[v] = (int32_t)i + 1;
}

// the round's arc table -> what branch marking reads (see pga_arc_set_current)
__global__ __launch_bounds__(BLOCK) void k_seg_gid(const int32_t *g2s, int Q, int n_seg, int32_t *seg_gid)
{
	int g = blockIdx.x * BLOCK + threadIdx.x;
	if (g < Q) { int s = g2s[g]; if (s >= 0 && s < n_seg) seg_gid[s] = g; }
}

__global__ __launch_bounds__(BLOCK) void k_cur_prep(const pga_arc_part_t *arcs, int64_t n_arc, const int32_t *seg_gid, uint64_t *ax, int32_t *s1, int32_t *agid,
                                                      int32_t *vs, int32_t *ve)
{
	int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n_arc) return;
	const pga_arc_part_t a = arcs[i];
	const uint32_t v = (uint32_t)(a.x >> 32);
	ax[i] = a.x;
	s1[i] = (int32_t)((double)a.sum_s1 / a.n_genome + .499); // graph.c:171
	agid[i] = seg_gid[(uint32_t)a.x >> 1];
	if (i == 0 || (uint32_t)(arcs[i - 1].x >> 32) != v) vs[v] = (int32_t)i;
	if (i == n_arc - 1 || (uint32_t)(arcs[i + 1].x >> 32) != v) ve[v] = (int32_t)i + 1;
}

__global__ __launch_bounds__(BLOCK) void k_deg(const int32_t *vs, const int32_t *ve, int n_vtx, int32_t *deg)
{
	int v = blockIdx.x * BLOCK + threadIdx.x;
	if (v < n_vtx) deg[v] = ve[v] - vs[v];
}

// number of pg_n_local calls of vertex v: n_max * n_weak (branch.c:70-75) + n(n-1)/2 (branch.c:83-88)
__global__ __launch_bounds__(BLOCK) void k_br_count(int n_vtx, const int32_t *vs, const int32_t *ve, const int32_t *s1, double bd, int32_t *pc)
{
	const int v = blockIdx.x * BLOCK + threadIdx.x;
	if (v >= n_vtx) return;
	const int a0 = vs[v], n = ve[v] - a0;
	if (n < 2) { pc[v] = 0; return; }
	int max_s1 = 0, n_max = 0, n_weak = 0;
	for (int i = 0; i < n; ++i) max_s1 = max_s1 > s1[a0 + i] ? max_s1 : s1[a0 + i];
	for (int i = 0; i < n; ++i) {
		n_max += s1[a0 + i] == max_s1;
		n_weak += (1.0 - (double)s1[a0 + i] / max_s1) > bd; // branch.c:71-72
	}
	pc[v] = n_max * n_weak + n * (n - 1) / 2;
}

// sequential form (one lane), used for vertices with more than 64 arcs.  MODE 1: write pairs; 2: decide.
template <int MODE>
__device__ void br_vertex_seq(int a0, int n, const int32_t *s1, const int32_t *agid, double bd, int64_t k, int32_t *pairs, const int32_t *cnt,
                              double bdist, double bcut, uint8_t *weak, int32_t *grp, int32_t *ndl_out, int64_t *dcnt)
{
	int max_s1 = 0;
	for (int i = 0; i < n; ++i) max_s1 = max_s1 > s1[a0 + i] ? max_s1 : s1[a0 + i];
	for (int i = 0; i < n; ++i) {
		const double r = 1.0 - (double)s1[a0 + i] / max_s1;
		if (!(r > bd)) continue;
		int n_local = 0;
		for (int j = 0; j < n; ++j) {
			if (s1[a0 + j] != max_s1) continue;
			if (MODE == 1) pairs[2 * k] = agid[a0 + j], pairs[2 * k + 1] = agid[a0 + i];
			if (MODE == 2) n_local += cnt[k];
			++k;
		}
		if (MODE == 2) {
			weak[a0 + i] = ((n_local == 0 && r > bdist) || r > bcut) ? 2 : 1;
		}
	}
	int n_group = 0;
	for (int i = 0; i < n; ++i) {
		if (MODE == 2 && grp[a0 + i] == 0) grp[a0 + i] = ++n_group;
		for (int j = i + 1; j < n; ++j) {
			if (MODE == 1) pairs[2 * k] = agid[a0 + i], pairs[2 * k + 1] = agid[a0 + j];
			if (MODE == 2 && cnt[k] > 0 && grp[a0 + j] == 0) grp[a0 + j] = grp[a0 + i];
			++k;
		}
	}
	if (MODE == 2) *ndl_out = n_group;
}

// One WAVE per oriented vertex; lane j holds arc j (score, target gene, group mark) in registers and arcs are
// broadcast with shuffles, so the O(n^2) pair loops of branch.c:70-90 touch memory only for the pair list
// (coalesced stores, MODE 1) or the all-reduced counts (coalesced loads, MODE 2).
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_br_wave(int n_vtx, const int32_t *vs, const int32_t *ve, const int32_t *s1g, const int32_t *agidg, double bd,
                                                     const int32_t *poff, int32_t *pairs, const int32_t *cnt, double bdist, double bcut,
                                                     uint8_t *weak, int32_t *grpg, int32_t *ndl, int64_t *dcnt)
{
	const int v = blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (v >= n_vtx) return;
	const int a0 = vs[v], n = ve[v] - a0;
	if (n < 2) return;
	const int64_t k0 = poff[v];
	if (n > WAVE) {
		if (lane == 0) { int32_t g = 0; br_vertex_seq<MODE>(a0, n, s1g, agidg, bd, k0, pairs, cnt, bdist, bcut, weak, grpg, &g, dcnt); if (MODE == 2) ndl[v] = g; }
		return;
	}
	const bool in = lane < n;
	const int my_s1 = in ? s1g[a0 + lane] : 0, my_gid = in ? agidg[a0 + lane] : 0;
	int max_s1 = my_s1;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(max_s1, o, WAVE); max_s1 = max_s1 > t ? max_s1 : t; }
	const double r = in ? 1.0 - (double)my_s1 / max_s1 : 0.0; // branch.c:71
	const bool is_weak = in && r > bd, is_max = in && my_s1 == max_s1;
	const unsigned long long m_weak = __ballot(is_weak), m_max = __ballot(is_max);
	const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
	const int n_max = __popcll(m_max), mrank = __popcll(m_max & lt);
	// part 1 (branch.c:70-77): for every weak arc i (ascending), one pair per best-scoring arc j (ascending)
	int wb = 0;
	for (unsigned long long m = m_weak; m; m &= m - 1, ++wb) {
		const int i = __ffsll((long long)m) - 1;
		const int gid_i = __shfl(my_gid, i, WAVE);
		const int64_t k = k0 + (int64_t)wb * n_max + mrank;
		if (MODE == 1) { if (is_max) pairs[2 * k] = my_gid, pairs[2 * k + 1] = gid_i; }
		else {
			int c = is_max ? cnt[k] : 0;
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, WAVE);
			if (lane == i) {
				weak[a0 + i] = ((c == 0 && r > bdist) || r > bcut) ? 2 : 1;
			}
		}
	}
	// part 2 (branch.c:82-90): all i<j pairs, row i starts after i*n - i(i+1)/2 earlier pairs
	const int64_t k2 = k0 + (int64_t)n_max * __popcll(m_weak);
	int grp = 0, n_group = 0;
	for (int i = 0; i < n; ++i) {
		const int64_t k = k2 + (int64_t)i * n - (int64_t)i * (i + 1) / 2 + (lane - i - 1);
		if (MODE == 1) {
			const int gid_i = __shfl(my_gid, i, WAVE);
			if (lane > i && in) pairs[2 * k] = gid_i, pairs[2 * k + 1] = my_gid;
		} else {
			int gi = __shfl(grp, i, WAVE);
			if (gi == 0) { gi = ++n_group; if (lane == i) grp = gi; } // uniform: every lane sees the same gi
			if (lane > i && in && grp == 0 && cnt[k] > 0) grp = gi;
		}
	}
	if (MODE == 2 && lane == 0) ndl[v] = n_group;
}

__device__ __forceinline__ int arc_weak(const uint64_t *ax, const uint8_t *aw, int64_t n, uint64_t x) // pg_get_arc, pgpriv.h:99-107
{
	int64_t lo = 0, hi = n;
	while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (ax[mid] < x) lo = mid + 1; else hi = mid; }
	return (lo < n && ax[lo] == x) ? aw[lo] : 0;
}

// pg_get_arc as in the reference (pgpriv.h:99-107): scan the few arcs leaving v; vs/ve = arc range of each vertex
__device__ __forceinline__ int arc_weak_v(const uint64_t *ax, const uint8_t *aw, const int32_t *vs, const int32_t *ve, uint32_t v, uint32_t w)
{
	for (int i = vs[v], e = ve[v]; i < e; ++i)
		if ((uint32_t)ax[i] == w) return aw[i];
	return 0;
}

__global__ __launch_bounds__(BLOCK) void k_mark_hits(const int32_t *val, const int32_t *prev, const int4 *YA, const int4 *YB, const int32_t *g2s, int n,
                                                       const uint64_t *ax, const uint8_t *aw, int64_t n_arc, const int32_t *vs, const int32_t *ve, int32_t *weak_new)
{
	int y = blockIdx.x * BLOCK + threadIdx.x;
	if (y >= n || val[y] < 0) return;
	int p = prev[y];
	if (p < 0) return;
	const int4 aA = YA[y], bA = YA[p];
	if (aA.x != bA.x) return; // branch.c:124
	const int aw_ = YB[y].w, bw_ = YB[p].w; // X position << 1 | rev
	uint32_t w = (uint32_t)g2s[aA.y] << 1 | (uint32_t)(aw_ & 1);
	uint32_t v = (uint32_t)g2s[bA.y] << 1 | (uint32_t)(bw_ & 1);
	int e1 = vs ? arc_weak_v(ax, aw, vs, ve, v, w) : arc_weak(ax, aw, n_arc, (uint64_t)v << 32 | w);                       // branch.c:128-130: marks the earlier hit
	if (e1) atomicMax(&weak_new[bw_ >> 1], e1);
	int e2 = vs ? arc_weak_v(ax, aw, vs, ve, w ^ 1, v ^ 1) : arc_weak(ax, aw, n_arc, (uint64_t)(w ^ 1) << 32 | (v ^ 1)); // branch.c:131-133: marks this hit
	if (e2) atomicMax(&weak_new[aw_ >> 1], e2);
}

__global__ __launch_bounds__(BLOCK) void k_weak_merge(uint32_t *flags, const int32_t *weak_new, int n, int64_t *cnt)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	const bool in = h < n;
	if (!in) h = n - 1;
	uint32_t f = flags[h];
	int cur = in ? (int)((f & PGA_F_WEAK_MASK) >> PGA_F_WEAK_SHIFT) : 0, nw = in ? weak_new[h] : 0;
	if (nw > cur) { cur = nw; flags[h] = (f & ~PGA_F_WEAK_MASK) | (uint32_t)nw << PGA_F_WEAK_SHIFT; }
	if (cnt) { // log-only counter (branch.c:137-139): one atomic per wave, and only when somebody asks
		const unsigned long long m = __ballot(cur != 0);
		if (m && (threadIdx.x & 63) == (unsigned)__ffsll((long long)m) - 1) atomicAdd((unsigned long long *)cnt, (unsigned long long)__popcll(m));
	}
}

// hazard H2b: two consecutive walkable hits (cs order) share (contig, cs)
__global__ __launch_bounds__(BLOCK) void k_hz_cs(const int32_t *wk, const int32_t *seg, const int32_t *cs, int n, int64_t *dcnt)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n || h == 0 || !wk[h]) return;
	for (int j = h - 1; j >= 0 && seg[j] == seg[h] && cs[j] == cs[h]; --j)
		if (wk[j]) { atomicAdd((unsigned long long *)&dcnt[6], 1ull); break; }
}

// ------------------------------------------------------------------------------------------------
// download: per-hit state back to file order
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_to_file(const int32_t *fidx, const int32_t *gnm, const int32_t *goff, const uint32_t *flags, const int32_t *rank,
                                                     const int32_t *sdom, const int32_t *pdom, const int32_t *pdom0, const int32_t *yperm, int n,
                                                     uint32_t *oflags, int32_t *orank, int32_t *osdom, int32_t *opdom, int32_t *opdom0, int32_t *opx, int32_t *opy)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	int g = gnm[h], f = goff[g] + fidx[h];
	oflags[f] = flags[h] & F_PUBLIC, orank[f] = rank[h], osdom[f] = sdom[h], opdom[f] = pdom[h], opdom0[f] = pdom0[h];
	opx[f] = h - goff[g];
	int x = yperm[h]; // h doubles as a Y position here
	opy[goff[gnm[x]] + fidx[x]] = h - goff[gnm[x]];
}


// ------------------------------------------------------------------------------------------------
// exact-order overrides (pangene_hip.h): re-permute contig segments of the physical (X) order, or
// rewrite slices of the Y permutation.  Rare (a few calls per run), not tuned.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_ov_inv(const int32_t *fidx, const int32_t *gnm, const int32_t *goff, int n, int32_t *inv, int32_t *remap)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n) return;
	inv[goff[gnm[h]] + fidx[h]] = h;
	remap[h] = h;
}

__global__ __launch_bounds__(BLOCK) void k_ov_sety(const int32_t *ov_pos, const int32_t *ov_file, int64_t t, const int32_t *inv, int32_t *yperm)
{
	int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i < t) yperm[ov_pos[i]] = inv[ov_file[i]];
}

__global__ __launch_bounds__(BLOCK) void k_inv_only(const int32_t *fidx, const int32_t *gnm, const int32_t *goff, int n, int32_t *inv)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h < n) inv[goff[gnm[h]] + fidx[h]] = h;
}

// move the "index 0" mark of each genome to the hit the reference has there (overlap.c:108)
__global__ __launch_bounds__(BLOCK) void k_set_head(const int32_t *head_file, const int32_t *goff, const int32_t *inv, int n_genome, int32_t *headpos, uint32_t *flags)
{
	int g = blockIdx.x * BLOCK + threadIdx.x;
	if (g >= n_genome || goff[g] == goff[g + 1]) return;
	int np = head_file[g] < 0 ? goff[g] : inv[goff[g] + head_file[g]], op = headpos[g];
	if (np == op) return;
	flags[op] &= ~F_HEAD;
	flags[np] |= F_HEAD;
	headpos[g] = np;
}

struct PermArrays { int32_t *a[17]; }; // a[15] = flags, a[16] = rk

__global__ __launch_bounds__(BLOCK) void k_ov_gather(PermArrays p, const int32_t *ov_pos, const int32_t *ov_file, int64_t t, const int32_t *inv,
                                                       int32_t *tmp, int32_t *remap)
{
	int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i >= t) return;
	int src = inv[ov_file[i]];
	remap[src] = ov_pos[i];
#pragma unroll
	for (int k = 0; k < 17; ++k) tmp[(int64_t)k * t + i] = p.a[k][src];
}

__global__ __launch_bounds__(BLOCK) void k_ov_scatter(PermArrays p, const int32_t *ov_pos, int64_t t, const int32_t *tmp,
                                                        const int32_t *gnm, const int32_t *goff)
{
	int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i >= t) return;
	int pos = ov_pos[i];
#pragma unroll
	for (int k = 0; k < 15; ++k) p.a[k][pos] = tmp[(int64_t)k * t + i];
	uint32_t f = (uint32_t)tmp[(int64_t)15 * t + i] & ~F_HEAD; // a[15] = flags; the head mark is positional
	if (pos == goff[gnm[pos]]) f |= F_HEAD;
	p.a[15][pos] = (int32_t)f;
	p.a[16][pos] = tmp[(int64_t)16 * t + i];
}

__global__ __launch_bounds__(BLOCK) void k_ov_remap_y(int32_t *yperm, int n, const int32_t *remap)
{
	int y = blockIdx.x * BLOCK + threadIdx.x;
	if (y < n) yperm[y] = remap[yperm[y]];
}

__global__ __launch_bounds__(BLOCK) void k_flt_bits(const uint32_t *flags, int n, unsigned long long *bits)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	const unsigned long long m = __ballot(h < n && (flags[h < n ? h : n - 1] & PGA_F_FLT));
	if ((threadIdx.x & 63) == 0 && h < n) bits[h >> 6] = m;
}

// ================================================================================================
// host side of the ABI
// ================================================================================================
static int sync_st(pga_ctx *c) { HIPCHK(hipStreamSynchronize(c->st)); return 0; }

static int bits_for(uint32_t maxv) { int b = 1; while (b < 32 && (maxv >> b)) ++b; return b; }

static int make_sweep_view(pga_ctx *c, SweepView *v)
{
	v->A = c->recA, v->B = c->recB, v->C = c->recC, v->sori = c->sori, v->exon = c->exon, v->flags = c->flags, v->pdom = c->pdom, v->sdom = c->sdom;
	v->n = c->N, v->min_ov = c->par.min_ov_ratio, v->check_strand = c->par.check_strand, v->hz = c->dcnt + 4, v->stage_c = c->any_multi;
	v->slow_cnt = nullptr, v->slow_list = (int32_t *)c->pool.get(S_SLOW, sizeof(int32_t) * (size_t)c->N);
	v->hz_list = (int32_t *)c->pool.get(S_HZLIST, sizeof(int32_t) * PGA_HAZARD_CAP);
	if (!v->slow_list || !v->hz_list) return PGA_ERR_NOMEM;
	return 0;
}

static void pack_records(pga_ctx *c)
{
	if (c->N) hipLaunchKernelGGL(k_pack_rec, dim3(nblk(c->N)), dim3(BLOCK), 0, c->st, c->seg, c->cs, c->ce, c->pm, c->rk, c->gid, c->cds, c->rank, c->nex, c->offx,
	                             c->pid, c->sori, c->N, c->recA, c->recB, c->recC);
}

template <int MODE> static int launch_sweep(pga_ctx *c, int timed_which)
{
	SweepView v;
	if (c->N == 0) return 0;
	{ const int rc = make_sweep_view(c, &v); if (rc) return rc; }
	TimedLaunch t; t.which = timed_which; t.units = c->N;
	static const int reps = [] { const char *e = getenv("PGA_SW_REPS"); return e && atoi(e) > 0 ? atoi(e) : 1; }(); // tuning aid: the sweep is idempotent
	const bool timed = timed_which >= 0;
	if (timed) {
		HIPCHK(hipEventCreate(&t.a)); HIPCHK(hipEventCreate(&t.b));
		if (reps != 1) HIPCHK(hipEventRecord(t.a, c->st));
	}
	c->walk_valid = false;
	const int nt = (int)nblk(c->N, SW_TILE);
	v.prof = nullptr;
#ifdef PGA_SW_PROFILE
	HIPCHK(hipMalloc((void **)&v.prof, sizeof(long long) * 8 * SW_NW * (size_t)nt));
#endif
	for (int rep = 0; rep < reps; ++rep) {
		v.slow_cnt = c->dcnt + 12 + (c->sweep_seq & 1);
		// a timed launch carries its own start/stop events: they take the dispatch's begin and end time stamps, i.e. the
		// duration of k_sweep itself, the figure rocprofv3 --kernel-trace reports for it
		hipEvent_t ea = timed && reps == 1 ? t.a : nullptr, eb = timed && reps == 1 ? t.b : nullptr;
		if (c->any_multi) hipExtLaunchKernelGGL((k_sweep<MODE, true>), dim3(nt), dim3(SW_TILE), 0, c->st, ea, eb, 0, v);
		else hipExtLaunchKernelGGL((k_sweep<MODE, false>), dim3(nt), dim3(SW_TILE), 0, c->st, ea, eb, 0, v);
		hipLaunchKernelGGL((k_sweep_slow<MODE>), dim3(64), dim3(BLOCK), 0, c->st, v, (long long *)(c->dcnt + 12 + ((c->sweep_seq + 1) & 1)));
		++c->sweep_seq;
	}
#ifdef PGA_SW_PROFILE
	{
		std::vector<long long> hp((size_t)8 * SW_NW * nt);
		HIPCHK(hipStreamSynchronize(c->st));
		HIPCHK(hipMemcpy(hp.data(), v.prof, hp.size() * sizeof(long long), hipMemcpyDeviceToHost));
		(void)hipFree(v.prof);
		double d[8] = { 0 };
		for (size_t w = 0; w < (size_t)SW_NW * nt; ++w)
			for (int k = 1; k < 8; ++k) d[k] += (double)(hp[w * 8 + k] - hp[w * 8 + k - 1]);
		fprintf(stderr, "[sweep<%d> profile, cycles/wave] load+put %.0f | barrier %.0f | count+scan %.0f | list %.0f | eval %.0f | - %.0f | finish %.0f\n", MODE,
		        d[1] / (SW_NW * nt), d[2] / (SW_NW * nt), d[3] / (SW_NW * nt), d[4] / (SW_NW * nt), d[5] / (SW_NW * nt), d[6] / (SW_NW * nt), d[7] / (SW_NW * nt));
	}
#endif
	if (timed_which >= 0) {
		if (reps != 1) HIPCHK(hipEventRecord(t.b, c->st));
		c->timed.push_back(t);
	}
	return 0;
}

static int radix_sort_pool(pga_ctx *c, uint64_t *keys, uint32_t *vals, int64_t n, int n_bits, uint64_t **kres, uint32_t **vres)
{
	RadixBufs b;
	if (n > 2 * (int64_t)c->N) return PGA_ERR_ARG; // work buffers are sized once, in create, for 2N items
	b.k_alt = (uint64_t *)c->pool.get(S_KEY_B, 0);
	b.v_alt = (uint32_t *)c->pool.get(S_VAL_B, 0);
	b.table = (uint32_t *)c->pool.get(S_TABLE, 0);
	b.tile_buf = (int32_t *)c->pool.get(S_TILE, 0);
	if (!b.k_alt || !b.v_alt || !b.table || !b.tile_buf) return PGA_ERR_NOMEM;
	device_radix_sort(keys, vals, n, n_bits, b, kres, vres, c->st);
	return 0;
}

extern "C" void pga_destroy(pga_ctx_t *c)
{
	if (c == nullptr) return;
	if (c->st) (void)hipStreamSynchronize(c->st);
	for (auto &t : c->timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
	for (void *q : c->owned) (void)hipFree(q);
	c->pool.release();
	if (c->h_cnt) (void)hipHostFree(c->h_cnt);
	if (c->h_stage) (void)hipHostFree(c->h_stage);
	if (c->h_g2s) (void)hipHostFree(c->h_g2s);
	if (c->g2s_done) (void)hipEventDestroy(c->g2s_done);
	if (c->own_stream && c->st) (void)hipStreamDestroy(c->st);
	delete c;
}

static hipStream_t g_active_stream = nullptr; // stream of the live context: collectives of a sharded run are enqueued here

extern "C" void *pga_active_stream(void) { return (void *)g_active_stream; }

extern "C" int pga_set_stream(pga_ctx_t *c, void *hip_stream)
{
	if (c == nullptr) return PGA_ERR_ARG;
	if (c->st) HIPCHK(hipStreamSynchronize(c->st));
	if (c->own_stream && c->st) (void)hipStreamDestroy(c->st);
	c->st = (hipStream_t)hip_stream, c->own_stream = false;
	g_active_stream = c->st;
	return 0;
}

// clears up to four buffers (byte counts are rounded up to whole dwords; every pool buffer has that slack) in one launch
static void zero_multi(pga_ctx *c, void *p0, size_t b0, void *p1 = nullptr, size_t b1 = 0, void *p2 = nullptr, size_t b2 = 0, void *p3 = nullptr, size_t b3 = 0)
{
	ZeroList z = { { p0, p1, p2, p3 }, { (b0 + 3) / 4, (b1 + 3) / 4, (b2 + 3) / 4, (b3 + 3) / 4 } };
	const unsigned long long tot = z.dwords[0] + z.dwords[1] + z.dwords[2] + z.dwords[3];
	if (tot) hipLaunchKernelGGL(k_zero_multi, dim3((unsigned)((tot + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, c->st, z);
}

template <class T> static int upload(pga_ctx *c, T *dst, const T *src, size_t n)
{
	if (n == 0) return 0;
	HIPCHK(hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyHostToDevice, c->st));
	return 0;
}

#define TRY(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

static int create_impl(pga_ctx *c, const pga_shard_t *sh)
{
	const int N = c->N, E = c->E, GL = c->n_genome;
	HIPCHK(hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking));
	g_active_stream = c->st;
	{
		int dev = 0, ncu = 0;
		if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && ncu > 0) c->n_cu = ncu;
	}
	c->own_stream = true;
	HIPCHK(hipHostMalloc((void **)&c->h_cnt, 16 * sizeof(int64_t), hipHostMallocDefault));
	HIPCHK(hipHostGetDevicePointer((void **)&c->h_box, c->h_cnt, 0));
	TRY(dalloc(c, &c->dcnt, 16));
	// persistent arrays
	TRY(dalloc(c, &c->fidx, N)); TRY(dalloc(c, &c->gnm, N)); TRY(dalloc(c, &c->seg, N)); TRY(dalloc(c, &c->pid, N)); TRY(dalloc(c, &c->gid, N));
	TRY(dalloc(c, &c->cs, N)); TRY(dalloc(c, &c->ce, N)); TRY(dalloc(c, &c->cm, N)); TRY(dalloc(c, &c->cds, N)); TRY(dalloc(c, &c->nex, N));
	TRY(dalloc(c, &c->offx, N)); TRY(dalloc(c, &c->sori, N)); TRY(dalloc(c, &c->sadj, N)); TRY(dalloc(c, &c->pm, N)); TRY(dalloc(c, &c->rk, N)); TRY(dalloc(c, &c->recA, N)); TRY(dalloc(c, &c->recB, N)); TRY(dalloc(c, &c->recC, N)); TRY(dalloc(c, &c->yrecA, N)); TRY(dalloc(c, &c->yrecB, N));
	TRY(dalloc(c, &c->rank, N)); TRY(dalloc(c, &c->sdom, N)); TRY(dalloc(c, &c->pdom, N)); TRY(dalloc(c, &c->pdom0, N)); TRY(dalloc(c, &c->flags, N));
	TRY(dalloc(c, &c->yperm, N)); TRY(dalloc(c, &c->goff, GL + 1)); TRY(dalloc(c, &c->ggl, GL)); TRY(dalloc(c, &c->ctg_base, GL + 1)); TRY(dalloc(c, &c->inv, N)); TRY(dalloc(c, &c->headpos, GL + 1)); TRY(dalloc(c, &c->exon, E));
	TRY(dalloc(c, &c->prot_gid, c->P)); TRY(dalloc(c, &c->gene_pref, c->Q));
	TRY(dalloc(c, &c->max_ori, c->P)); TRY(dalloc(c, &c->sums, 6 * (size_t)c->P)); TRY(dalloc(c, &c->vtx_cnt, 2 * (size_t)c->Q)); TRY(dalloc(c, &c->g2s, c->Q));

	// host-side small tables
	std::vector<int32_t> ctg_base((size_t)GL + 1, 0);
	c->h_goff.resize((size_t)GL + 1);
	for (int g = 0; g <= GL; ++g) c->h_goff[(size_t)g] = (int32_t)sh->hit_off[g];
	for (int g = 0; g < GL; ++g) ctg_base[(size_t)g + 1] = ctg_base[(size_t)g] + sh->n_ctg[g];
	c->rp_compact = true;
	for (int g = 0; g < GL; ++g) if (sh->n_ctg[g] >= 4096 || sh->hit_off[g + 1] - sh->hit_off[g] >= (1 << 20)) c->rp_compact = false;
	c->n_seg_ctg = ctg_base[(size_t)GL];
	c->h_ggl.assign(sh->genome_global, sh->genome_global + GL);
	uint32_t max_cs = 0, max_cm = 0, max_sadj = 0;
	bool neg_sadj = false, multi = false;
	for (int i = 0; i < N; ++i) {
		if (sh->cs[i] < 0 || sh->ce[i] < sh->cs[i] || sh->cm[i] < 0 || sh->cid[i] < 0) return PGA_ERR_RANGE;
		max_cs = std::max(max_cs, (uint32_t)sh->cs[i]), max_cm = std::max(max_cm, (uint32_t)sh->cm[i]);
		if (sh->score_adj[i] < 0) neg_sadj = true; else max_sadj = std::max(max_sadj, (uint32_t)sh->score_adj[i]);
		multi = multi || sh->n_exon_of[i] != 1;
	}
	c->cs_bits = bits_for(max_cs), c->cm_bits = bits_for(max_cm), c->seg_bits = bits_for((uint32_t)std::max(1, c->n_seg_ctg));
	c->sc_bits = neg_sadj ? 64 : std::min(64, 33 + bits_for(max_sadj)); // score key = score_adj << 33 | preferred << 32 | hash(pid)
	c->any_multi = multi;
	std::vector<int2> hex((size_t)E);
	for (int e = 0; e < E; ++e) hex[(size_t)e] = make_int2(sh->exon_os[e], sh->exon_oe[e]);

	// the shard in file order stays resident (S_UPLOAD) so that begin() can restart a run without PCIe traffic
	int32_t *up = (int32_t *)c->pool.get(S_UPLOAD, sizeof(int32_t) * (size_t)N * 16 + 64);
	if (!up) return PGA_ERR_NOMEM;
	TRY(upload(c, up, sh->pid, N)); TRY(upload(c, up + (size_t)N, sh->cid, N)); TRY(upload(c, up + 2 * (size_t)N, sh->rank, N));
	TRY(upload(c, up + 3 * (size_t)N, sh->score_ori, N)); TRY(upload(c, up + 4 * (size_t)N, sh->score_adj, N));
	TRY(upload(c, up + 5 * (size_t)N, sh->n_exon_of, N)); TRY(upload(c, up + 6 * (size_t)N, sh->off_exon, N));
	TRY(upload(c, up + 7 * (size_t)N, sh->cs, N)); TRY(upload(c, up + 8 * (size_t)N, sh->ce, N)); TRY(upload(c, up + 9 * (size_t)N, sh->cm, N));
	TRY(upload(c, (uint8_t *)(up + 14 * (size_t)N), sh->rev, N));
	TRY(upload(c, c->goff, c->h_goff.data(), (size_t)GL + 1)); TRY(upload(c, c->ggl, c->h_ggl.data(), GL));
	TRY(upload(c, c->ctg_base, ctg_base.data(), (size_t)GL + 1));
	TRY(upload(c, c->exon, hex.data(), E)); TRY(upload(c, c->prot_gid, sh->prot_gid, c->P)); TRY(upload(c, c->gene_pref, sh->gene_pref, c->Q));
	{ // work buffers shared by every sort / scan of the run: sized for the largest input (2N temp arcs)
		const int64_t W = 2 * (int64_t)N + 2;
		if (!c->pool.get(S_KEY_A, sizeof(uint64_t) * (size_t)W) || !c->pool.get(S_VAL_A, sizeof(uint32_t) * (size_t)W) ||
		    !c->pool.get(S_KEY_B, sizeof(uint64_t) * (size_t)W) || !c->pool.get(S_VAL_B, sizeof(uint32_t) * (size_t)W) ||
		    !c->pool.get(S_TABLE, sizeof(uint32_t) * (size_t)rs_table_len(W)) ||
		    !c->pool.get(S_TILE, sizeof(int64_t) * (size_t)(scan_tiles(std::max<int64_t>(rs_table_len(W), W)) + 8))) return PGA_ERR_NOMEM;
	}
	return sync_st(c);
}

// per-hit constants in file order, X order (sort + gather), running max of ce, Y order; resets all state
extern "C" int pga_begin(pga_ctx_t *c)
{
	c->yrec_valid = false;
	const int N = c->N, GL = c->n_genome;
	c->walk_valid = false;
	HIPCHK(hipMemsetAsync(c->dcnt, 0, 16 * sizeof(int64_t), c->st));
	if (c->Q) hipLaunchKernelGGL(k_fill_i32, dim3(nblk(c->Q)), dim3(BLOCK), 0, c->st, c->g2s, (int64_t)c->Q, -1);
	c->n_seg = 0;
	if (N == 0) return sync_st(c);
	int32_t *up = (int32_t *)c->pool.get(S_UPLOAD, 0);
	int32_t *f_pid = up, *f_cid = up + (size_t)N, *f_rank = up + 2 * (size_t)N, *f_sori = up + 3 * (size_t)N, *f_sadj = up + 4 * (size_t)N, *f_nex = up + 5 * (size_t)N,
		*f_offx = up + 6 * (size_t)N, *f_cs = up + 7 * (size_t)N, *f_ce = up + 8 * (size_t)N, *f_cm = up + 9 * (size_t)N,
		*f_gnm = up + 10 * (size_t)N, *f_seg = up + 11 * (size_t)N, *f_gid = up + 12 * (size_t)N, *f_cds = up + 13 * (size_t)N;
	uint8_t *f_rev = (uint8_t *)(up + 14 * (size_t)N);
	int32_t *rk_f = (int32_t *)c->pool.get(S_TAB_A, sizeof(uint64_t) * (size_t)N);
	int32_t *head = (int32_t *)c->pool.get(S_HEAD, sizeof(int32_t) * ((size_t)N + 1)), *incl = (int32_t *)c->pool.get(S_SLOT, sizeof(int32_t) * ((size_t)N + 1));
	uint64_t *key = (uint64_t *)c->pool.get(S_KEY_A, 0);
	uint32_t *val = (uint32_t *)c->pool.get(S_VAL_A, 0);
	if (!up || !rk_f || !head || !incl || !key || !val) return PGA_ERR_NOMEM;
	FileHits f = { f_pid, f_cid, f_rank, f_sori, f_sadj, f_nex, f_offx, f_cs, f_ce, f_cm, f_rev };
	hipLaunchKernelGGL(k_prepare, dim3(nblk(N)), dim3(BLOCK), 0, c->st, f, N, c->goff, GL, c->ctg_base, c->exon, c->prot_gid, c->gene_pref,
	                   f_gnm, f_seg, f_gid, f_cds, key, val);
	uint64_t *ks; uint32_t *vs;
	{ // dense rank of the score keys (see k_rank_scatter)
		TRY(radix_sort_pool(c, key, val, N, c->sc_bits, &ks, &vs));
		hipLaunchKernelGGL(k_arc_head, dim3(nblk(N)), dim3(BLOCK), 0, c->st, ks, (int64_t)N, head);
		I32 *tile = (I32 *)c->pool.get(S_TILE, 0);
		device_scan<I32>(InI32{head}, OutInclI32{incl}, N, tile, OpSum{}, I32{0}, c->st);
		hipLaunchKernelGGL(k_rank_scatter, dim3(nblk(N)), dim3(BLOCK), 0, c->st, ks, vs, incl, N, rk_f);
	}
	// X order: pg_hit_sort(g, 0), hit.c:29-64, for every genome at once; stable => ties keep file order
	key = (uint64_t *)c->pool.get(S_KEY_A, 0), val = (uint32_t *)c->pool.get(S_VAL_A, 0);
	hipLaunchKernelGGL(k_xkey, dim3(nblk(N)), dim3(BLOCK), 0, c->st, f_seg, f_cs, N, c->cs_bits, key, val);
	TRY(radix_sort_pool(c, key, val, N, c->cs_bits + c->seg_bits, &ks, &vs));
	HitArrays o = { c->fidx, c->gnm, c->seg, c->pid, c->gid, c->cs, c->ce, c->cm, c->cds, c->nex, c->offx, c->sori, c->sadj, c->rank, c->sdom, c->pdom, c->pdom0, c->rk, c->flags };
	hipLaunchKernelGGL(k_gather, dim3(nblk(N)), dim3(BLOCK), 0, c->st, f, f_gnm, f_seg, f_gid, f_cds, rk_f, vs, N, c->goff, o);
	hipLaunchKernelGGL(k_inv_only, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->fidx, c->gnm, c->goff, N, c->inv);
	HIPCHK(hipMemcpyAsync(c->headpos, c->goff, sizeof(int32_t) * ((size_t)GL + 1), hipMemcpyDeviceToDevice, c->st));
	// running max of ce per contig
	SegMax *tile = (SegMax *)c->pool.get(S_TILE, 0);
	device_scan<SegMax>(InSegMax{c->seg, c->ce}, OutSegMax{c->pm}, N, tile, OpSegMax{}, SegMax{SEG_EMPTY, 0}, c->st);
	pack_records(c);
	// Y order: pg_hit_sort(g, 1); ties keep X order
	key = (uint64_t *)c->pool.get(S_KEY_A, 0), val = (uint32_t *)c->pool.get(S_VAL_A, 0);
	hipLaunchKernelGGL(k_ykey, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->seg, c->cm, N, c->cm_bits, key, val);
	TRY(radix_sort_pool(c, key, val, N, c->cm_bits + c->seg_bits, &ks, &vs));
	HIPCHK(hipMemcpyAsync(c->yperm, vs, sizeof(int32_t) * (size_t)N, hipMemcpyDeviceToDevice, c->st));
	return 0;
}

extern "C" int pga_create(pga_ctx_t **out, const pga_shard_t *sh, const pga_params_t *par)
{
	if (out == nullptr || sh == nullptr || par == nullptr) return PGA_ERR_ARG;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
		fprintf(stderr, "[E::pga_create] no HIP device is visible; libpangene_amd has no CPU fallback\n");
		return PGA_ERR_NO_DEVICE;
	}
	if (sh->n_hit >= INT32_MAX || sh->n_exon >= INT32_MAX || sh->n_gene >= (1 << 20) || sh->n_genome_global >= (1 << 24)) return PGA_ERR_RANGE;
	pga_ctx *c = new pga_ctx();
	c->n_genome = sh->n_genome, c->n_genome_global = sh->n_genome_global, c->P = sh->n_prot, c->Q = sh->n_gene;
	c->N = (int32_t)sh->n_hit, c->E = (int32_t)sh->n_exon, c->par = *par;
	int rc = create_impl(c, sh);
	if (rc) { pga_destroy(c); return rc; }
	*out = c;
	return 0;
}

// stage A (read.c:243-260) for all genomes of the shard
extern "C" int pga_ingest(pga_ctx_t *c, int32_t *stats)
{
	c->yrec_valid = false;
	const int N = c->N, GL = c->n_genome, P = c->P, Q = c->Q;
	int32_t *d_stats = (int32_t *)c->pool.get(S_STATS, sizeof(int32_t) * 4 * (size_t)GL + 16);
	if (!d_stats) return PGA_ERR_NOMEM;
	HIPCHK(hipMemsetAsync(d_stats, 0, sizeof(int32_t) * 4 * (size_t)GL + 16, c->st));
	if (N) {
		const int64_t TP = (int64_t)GL * P, TQ = (int64_t)GL * Q;
		int32_t *tmax = (int32_t *)c->pool.get(S_TAB_A, sizeof(int32_t) * (size_t)TP);
		int32_t *tmin = (int32_t *)c->pool.get(S_TAB_B, sizeof(int32_t) * (size_t)TP);
		int32_t *tr1 = (int32_t *)c->pool.get(S_TAB_C, sizeof(int32_t) * (size_t)TP);
		unsigned long long *tbest = (unsigned long long *)c->pool.get(S_TAB_D, sizeof(uint64_t) * (size_t)TQ);
		if (!tmax || !tmin || !tr1 || !tbest) return PGA_ERR_NOMEM;
		HIPCHK(hipMemsetAsync(tmax, 0, sizeof(int32_t) * (size_t)TP, c->st));
		hipLaunchKernelGGL(k_fill_i32, dim3(nblk(TP)), dim3(BLOCK), 0, c->st, tmin, TP, INT32_MAX);
		hipLaunchKernelGGL(k_fill_i32, dim3(nblk(TP)), dim3(BLOCK), 0, c->st, tr1, TP, INT32_MAX);
		hipLaunchKernelGGL(k_pseudo1, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->gnm, c->pid, c->nex, N, P, tmax, tmin);
		hipLaunchKernelGGL(k_pseudo2, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->gnm, c->pid, c->nex, c->rank, c->flags, N, P, tmax, tmin, tr1, d_stats);
		hipLaunchKernelGGL(k_pseudo3, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->gnm, c->pid, c->rank, N, P, tmax, tmin, tr1);
		pack_records(c); // rank changed
		TRY(launch_sweep<1>(c, 0)); // pg_shadow(cal_dom_sc=1), read.c:248 -- "K1", the hit-filter+overlap kernel
		hipLaunchKernelGGL(k_ingest_reset, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->flags, c->pdom, c->pdom0, N);
		TRY(launch_sweep<2>(c, 1)); // pg_flt_ov_isoform, read.c:254
		int32_t *tiso = tmax; // reuse: 1 = every hit of (genome, protein) carries flt_iso_ov
		hipLaunchKernelGGL(k_fill_i32, dim3(nblk(TP)), dim3(BLOCK), 0, c->st, tiso, TP, 1);
		hipLaunchKernelGGL(k_iso_apply, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->flags, c->gnm, c->pid, N, P, tiso, d_stats);
		hipLaunchKernelGGL(k_chain, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->flags, c->gnm, c->pdom0, N, P, tiso, d_stats);
		HIPCHK(hipMemsetAsync(tbest, 0, sizeof(uint64_t) * (size_t)TQ, c->st));
		hipLaunchKernelGGL(k_subopt1, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->flags, c->gnm, c->gid, c->rank, c->sadj, c->goff, N, Q, tbest);
		hipLaunchKernelGGL(k_subopt2, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->flags, c->gnm, c->gid, c->pid, c->goff, N, Q, tbest, d_stats);
	}
	if (stats) {
		HIPCHK(hipMemcpyAsync(stats, d_stats, sizeof(int32_t) * 4 * (size_t)GL, hipMemcpyDeviceToHost, c->st));
		return sync_st(c);
	}
	return 0;
}

extern "C" int pga_post_partials(pga_ctx_t *c, int32_t **max_ori, int64_t **sums)
{
	zero_multi(c, c->max_ori, sizeof(int32_t) * (size_t)std::max(1, c->P), c->sums, sizeof(int64_t) * 6 * (size_t)std::max(1, c->P));
	if (c->N) hipLaunchKernelGGL(k_post_part, dim3(nblk(c->N)), dim3(BLOCK), 0, c->st, c->flags, c->pid, c->rank, c->sori, c->sadj, c->nex, c->N, c->P,
	                             c->max_ori, (unsigned long long *)c->sums);
	*max_ori = c->max_ori, *sums = c->sums;
	return 0; // no wait: a consumer that is not on this stream calls pga_sync first
}

extern "C" int pga_post_apply(pga_ctx_t *c, const uint8_t *prot_rep, const uint8_t *prot_pj, int64_t *n_pseudo)
{
	c->yrec_valid = false;
	uint8_t *d = (uint8_t *)c->pool.get(S_MISC, 2 * (size_t)c->P + 16);
	if (!d) return PGA_ERR_NOMEM;
	c->walk_valid = false;
	TRY(upload(c, d, prot_rep, (size_t)c->P)); TRY(upload(c, d + c->P, prot_pj, (size_t)c->P));
	HIPCHK(hipMemsetAsync(c->dcnt + 2, 0, sizeof(int64_t), c->st));
	if (c->N) hipLaunchKernelGGL(k_post_apply, dim3(nblk(c->N)), dim3(BLOCK), 0, c->st, c->flags, c->pid, c->nex, c->sdom, c->N, c->max_ori, d, d + c->P, c->dcnt + 2);
	HIPCHK(hipMemcpyAsync(c->h_cnt, c->dcnt, 16 * sizeof(int64_t), hipMemcpyDeviceToHost, c->st));
	TRY(sync_st(c));
	if (n_pseudo) *n_pseudo = c->h_cnt[2];
	return 0;
}

extern "C" int pga_shadow(pga_ctx_t *c, int32_t cal_dom_sc, int32_t *stats)
{
	c->yrec_valid = false;
	if (cal_dom_sc) TRY(launch_sweep<1>(c, -1)); else TRY(launch_sweep<0>(c, 2));
	if (stats) {
		int32_t *d_stats = (int32_t *)c->pool.get(S_STATS, sizeof(int32_t) * 4 * (size_t)c->n_genome + 16);
		if (!d_stats) return PGA_ERR_NOMEM;
		HIPCHK(hipMemsetAsync(d_stats, 0, sizeof(int32_t) * 2 * (size_t)c->n_genome + 16, c->st));
		if (c->N) hipLaunchKernelGGL(k_count_shadow, dim3(nblk(c->N)), dim3(BLOCK), 0, c->st, c->flags, c->gnm, c->N, d_stats);
		HIPCHK(hipMemcpyAsync(stats, d_stats, sizeof(int32_t) * 2 * (size_t)c->n_genome, hipMemcpyDeviceToHost, c->st));
		return sync_st(c);
	}
	return 0;
}

extern "C" int pga_set_filter(pga_ctx_t *c, int32_t which)
{
	if (which < 0 || which > 3) return PGA_ERR_ARG;
	c->walk_valid = false;
	if (c->N) hipLaunchKernelGGL(k_set_filter, dim3(nblk(c->N)), dim3(BLOCK), 0, c->st, c->flags, c->N, which);
	return 0;
}

static int check_invariant(pga_ctx *c, bool flushed = false) // flushed: a k_mail_sum just sent the counters to the host mirror
{
	if (!flushed) hipLaunchKernelGGL(k_mail_flush, dim3(1), dim3(64), 0, c->st, c->dcnt, c->h_box);
	TRY(sync_st(c));
	return c->h_cnt[3] ? PGA_ERR_INVARIANT : 0;
}

extern "C" int pga_vtx_partials(pga_ctx_t *c, int32_t **cnt, uint64_t **records, int64_t *n_records)
{
	const int N = c->N, Q = c->Q, GL = c->n_genome;
	const int64_t wpg = (Q + 31) / 32, n_slot = (int64_t)std::max(1, Q) * VTX_K;
	const int nw = (c->n_genome_global + 63) / 64;
	const long long ovf_cap = 65536;
	uint32_t *bits = (uint32_t *)c->pool.get(S_BITS, sizeof(uint32_t) * (size_t)(wpg * GL) + 16);
	// [dom_tab: n_slot i32][slot: n_slot i32][pair bits: n_slot * nw u64][records: (n_slot + ovf_cap) * (1 + nw) u64]
	const size_t b_tab = sizeof(int32_t) * (size_t)n_slot, b_bits = sizeof(uint64_t) * (size_t)n_slot * (size_t)nw, b_rec = sizeof(uint64_t) * (size_t)(n_slot + ovf_cap) * (size_t)(1 + nw);
	char *blk = (char *)c->pool.get(S_TRIPLES, 2 * b_tab + b_bits + b_rec + 64);
	if (!bits || !blk) return PGA_ERR_NOMEM;
	int32_t *dom_tab = (int32_t *)blk, *slot = (int32_t *)(blk + b_tab);
	unsigned long long *pbits = (unsigned long long *)(blk + 2 * b_tab), *rec = (unsigned long long *)(blk + 2 * b_tab + b_bits);
	zero_multi(c, bits, sizeof(uint32_t) * (size_t)(wpg * GL) + 16, c->vtx_cnt, sizeof(int32_t) * 2 * (size_t)std::max(1, Q), c->dcnt, sizeof(int64_t), pbits, b_bits);
	HIPCHK(hipMemsetAsync(dom_tab, 0xff, b_tab, c->st)); // every slot empty (-1)
	*n_records = 0, *records = (uint64_t *)rec, *cnt = c->vtx_cnt;
	if (N == 0) return sync_st(c);
	hipLaunchKernelGGL(k_vtx1, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->flags, c->gnm, c->gid, c->rank, c->pdom, N, Q, c->vtx_cnt, bits, wpg, c->dcnt);
	hipLaunchKernelGGL(k_vtx_fold, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->flags, c->gnm, c->gid, c->rank, c->pdom, c->prot_gid, c->ggl, N, bits, wpg,
	                   dom_tab, pbits, nw, rec + n_slot * (1 + nw), ovf_cap, c->dcnt);
	I32 *tile = (I32 *)c->pool.get(S_TILE, 0);
	if (!tile) return PGA_ERR_NOMEM;
	device_scan<I32>(InDomSet{dom_tab}, OutExclI32{slot}, n_slot, tile, OpSum{}, I32{0}, c->st);
	hipLaunchKernelGGL(k_vtx_compact, dim3(nblk(n_slot)), dim3(BLOCK), 0, c->st, dom_tab, slot, n_slot, pbits, nw, rec, c->dcnt, c->h_box);
	TRY(sync_st(c));
	if (c->h_cnt[3]) return PGA_ERR_INVARIANT;
	const int64_t n_rec = c->h_cnt[10], n_ovf = c->h_cnt[0];
	if (n_ovf > ovf_cap) return PGA_ERR_RANGE; // > 65536 (genome, gene) cells beyond VTX_K dominators per gene
	if (n_ovf) { // the spilled single-genome records follow the folded ones
		HIPCHK(hipMemcpyAsync(rec + n_rec * (1 + nw), rec + n_slot * (1 + nw), sizeof(uint64_t) * (size_t)n_ovf * (size_t)(1 + nw), hipMemcpyDeviceToDevice, c->st));
		TRY(sync_st(c));
	}
	*n_records = n_rec + n_ovf;
	return 0;
}

extern "C" int pga_flag_vtx(pga_ctx_t *c, const int32_t *g2s, int32_t n_seg)
{
	// g2s is caller memory: it is copied into a pinned staging area so that the call need not wait for the upload
	const size_t nb = sizeof(int32_t) * (size_t)c->Q;
	if (c->h_g2s_cap < nb) {
		if (c->h_g2s) { HIPCHK(hipStreamSynchronize(c->st)); (void)hipHostFree(c->h_g2s); c->h_g2s = nullptr; }
		HIPCHK(hipHostMalloc((void **)&c->h_g2s, nb + 64, hipHostMallocDefault));
		c->h_g2s_cap = nb;
	}
	if (!c->g2s_done) HIPCHK(hipEventCreateWithFlags(&c->g2s_done, hipEventDisableTiming));
	else HIPCHK(hipEventSynchronize(c->g2s_done)); // the previous upload out of the staging area (long finished in practice)
	if (nb) memcpy(c->h_g2s, g2s, nb);
	TRY(upload(c, c->g2s, (const int32_t *)c->h_g2s, (size_t)c->Q));
	HIPCHK(hipEventRecord(c->g2s_done, c->st));
	c->n_seg = n_seg;
	if (c->N) hipLaunchKernelGGL(k_flag_vtx, dim3(nblk(c->N)), dim3(BLOCK), 0, c->st, c->flags, c->gid, c->N, c->g2s);
	return 0;
}

static void ensure_yrec(pga_ctx *c)
{
	if (c->yrec_valid || c->N == 0) return;
	hipLaunchKernelGGL(k_pack_yrec, dim3(nblk(c->N)), dim3(BLOCK), 0, c->st, c->yperm, c->seg, c->gid, c->gnm, c->cm, c->sori, c->sdom, c->pdom0, c->prot_gid, c->flags,
	                   c->N, c->yrecA, c->yrecB);
	c->yrec_valid = true;
}

// walkable marks in cm order + predecessor; shared by arc_round and mark_hits
static int walk_prev(pga_ctx *c, int32_t **val_out, int32_t **prev_out)
{
	const int N = c->N;
	int32_t *val = (int32_t *)c->pool.get(S_WALK_VAL, sizeof(int32_t) * (size_t)N);
	int32_t *prev = (int32_t *)c->pool.get(S_WALK_PREV, sizeof(int32_t) * (size_t)N);
	I32 *tile = (I32 *)c->pool.get(S_TILE, 0);
	if (!val || !prev || !tile) return PGA_ERR_NOMEM;
	*val_out = val, *prev_out = prev;
	if (c->walk_valid) return 0; // pg_mark_branch_flt_hit walks exactly what the pg_gen_arc before it walked: nothing changed in between
	hipLaunchKernelGGL(k_walk_mark, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->flags, c->yperm, N, val);
	device_scan<I32>(InWalk{val}, OutPrev{prev}, N, tile, OpMax{}, I32{-1}, c->st); // exclusive running max = previous walkable
	c->walk_valid = true;
	return 0;
}

extern "C" int pga_arc_round(pga_ctx_t *c, int32_t use_ori, int32_t **seg_cnt_out, pga_arc_part_t **arcs_out, int64_t *n_arcs_out)
{
	const int N = c->N, S = c->n_seg, GL = c->n_genome;
	int32_t *seg_cnt = (int32_t *)c->pool.get(S_SEGCNT, sizeof(int32_t) * 2 * (size_t)std::max(1, S) * SEGCNT_COPIES);
	if (!seg_cnt) return PGA_ERR_NOMEM;
	*seg_cnt_out = seg_cnt, *arcs_out = nullptr, *n_arcs_out = 0;
	if (N == 0) {
		HIPCHK(hipMemsetAsync(seg_cnt, 0, sizeof(int32_t) * 2 * (size_t)std::max(1, S) * SEGCNT_COPIES, c->st));
		return sync_st(c);
	}
	TRY(launch_sweep<0>(c, 2)); // graph.c:102
	int32_t *val, *prev;
	TRY(walk_prev(c, &val, &prev));
	const int64_t wpg = (S + 31) / 32;
	uint32_t *seen = (uint32_t *)c->pool.get(S_BITS, sizeof(uint32_t) * (size_t)(wpg * GL) + 16);
	int32_t *has = (int32_t *)c->pool.get(S_I32_C, sizeof(int32_t) * (size_t)N);
	int32_t *slot = (int32_t *)c->pool.get(S_SLOT, sizeof(int32_t) * (size_t)(2 * (int64_t)N + 2));
	if (!seen || !has || !slot) return PGA_ERR_NOMEM;
	zero_multi(c, seen, sizeof(uint32_t) * (size_t)(wpg * GL) + 16, seg_cnt, sizeof(int32_t) * 2 * (size_t)std::max(1, S) * SEGCNT_COPIES);
	ensure_yrec(c);
	hipLaunchKernelGGL(k_arc_flag, dim3(nblk(N)), dim3(BLOCK), 0, c->st, val, prev, c->yrecA, c->g2s, N, S, has, seg_cnt, seen, wpg, c->dcnt, (int32_t *)c->pool.get(S_HZLIST, sizeof(int32_t) * PGA_HAZARD_CAP));
	if (S) hipLaunchKernelGGL(k_segcnt_sum, dim3(nblk(2 * S)), dim3(BLOCK), 0, c->st, seg_cnt, 2 * S);
	I32 *tile = (I32 *)c->pool.get(S_TILE, 0);
	device_scan<I32>(InI32{has}, OutExclI32{slot}, N, tile, OpSum{}, I32{0}, c->st);
	// number of adjacencies = slot[N-1] + has[N-1]
	hipLaunchKernelGGL(k_mail_sum, dim3(1), dim3(64), 0, c->st, slot + (N - 1), has + (N - 1), c->dcnt, c->h_box);
	TRY(check_invariant(c, true));
	const int64_t M = 2 * c->h_cnt[10];
	if (M == 0) return sync_st(c);
	const int vbits = bits_for((uint32_t)(2 * std::max(1, S)));
	uint64_t *key = (uint64_t *)c->pool.get(S_KEY_A, sizeof(uint64_t) * (size_t)M);
	uint32_t *idx = (uint32_t *)c->pool.get(S_VAL_A, sizeof(uint32_t) * (size_t)M);
	int4 *tpay = (int4 *)c->pool.get(S_TDIST, sizeof(int4) * (size_t)M), *spay = (int4 *)c->pool.get(S_SDIST, sizeof(int4) * (size_t)M);
	int32_t *head = (int32_t *)c->pool.get(S_HEAD, sizeof(int32_t) * (size_t)M);
	if (!key || !idx || !tpay || !spay || !head) return PGA_ERR_NOMEM;
	ArcEmit e = { has, slot, prev, c->yrecA, c->yrecB, c->g2s, key, idx, tpay, N, use_ori, vbits };
	hipLaunchKernelGGL(k_arc_emit, dim3(nblk(N)), dim3(BLOCK), 0, c->st, e);
	uint64_t *ks; uint32_t *vs;
	TRY(radix_sort_pool(c, key, idx, M, 2 * vbits, &ks, &vs)); // graph.c:127 and :151 in one stable sort
	hipLaunchKernelGGL(k_arc_gather, dim3(nblk(M)), dim3(BLOCK), 0, c->st, vs, M, tpay, spay);
	hipLaunchKernelGGL(k_arc_head, dim3(nblk(M)), dim3(BLOCK), 0, c->st, ks, M, head);
	tile = (I32 *)c->pool.get(S_TILE, 0);
	device_scan<I32>(InI32{head}, OutExclI32{slot}, M, tile, OpSum{}, I32{0}, c->st);
	hipLaunchKernelGGL(k_mail_sum, dim3(1), dim3(64), 0, c->st, slot + (M - 1), head + (M - 1), c->dcnt, c->h_box);
	TRY(sync_st(c));
	const int64_t A = c->h_cnt[10];
	pga_arc_part_t *arcs = (pga_arc_part_t *)c->pool.get(S_ARCS, sizeof(pga_arc_part_t) * (size_t)A);
	if (!arcs) return PGA_ERR_NOMEM;
	{
		int32_t *run_start = (int32_t *)c->pool.get(S_RUNSTART, sizeof(int32_t) * (size_t)A + 16);
		int32_t *c_n = (int32_t *)tpay, *c_s1 = c_n + (size_t)M, *c_s2 = c_n + 2 * (size_t)M; // the unsorted payload is free again: reuse it
		uint64_t *c_dn = (uint64_t *)c->pool.get(S_CDN, sizeof(uint64_t) * (size_t)M);
		if (!run_start || !c_dn) return PGA_ERR_NOMEM;
		hipLaunchKernelGGL(k_arc_l1, dim3(nblk(M)), dim3(BLOCK), 0, c->st, ks, M, spay, head, slot, run_start, c_n, c_dn, c_s1, c_s2);
		hipLaunchKernelGGL(k_arc_l2, dim3(nblk(A, BLOCK / WAVE)), dim3(BLOCK), 0, c->st, ks, M, A, run_start, c_n, c_dn, c_s1, c_s2, vbits, arcs);
	}
	*arcs_out = arcs, *n_arcs_out = A;
	return sync_st(c);
}


extern "C" int pga_arc_merge(pga_ctx_t *c, const pga_arc_part_t *gathered, const int64_t *count, int32_t W, int64_t slot_sz,
                             pga_arc_part_t **out, int64_t *n_out)
{
	int64_t tot = 0;
	for (int r = 0; r < W; ++r) tot += count[r];
	*out = nullptr, *n_out = 0;
	if (tot == 0) return 0;
	std::vector<int64_t> off((size_t)W + 1, 0);
	for (int r = 0; r < W; ++r) off[(size_t)r + 1] = off[(size_t)r] + count[r];
	int64_t *d_off = (int64_t *)c->pool.get(S_MG_SRC, sizeof(int64_t) * ((size_t)W + 1));
	uint64_t *key = (uint64_t *)c->pool.get(S_MG_KEY, sizeof(uint64_t) * (size_t)tot + 64);
	uint32_t *val = (uint32_t *)c->pool.get(S_MG_VAL, sizeof(uint32_t) * (size_t)tot + 64);
	int32_t *slot = (int32_t *)c->pool.get(S_MG_SLOT, sizeof(int32_t) * (size_t)tot);
	int32_t *tile = (int32_t *)c->pool.get(S_TILE, 0);
	if (!d_off || !key || !val || !slot || !tile) return PGA_ERR_NOMEM;
	if (tot > 2 * (int64_t)c->N + 2) { // the scan buffer is sized for 2N items
		tile = (int32_t *)c->pool.get(S_TILE, sizeof(int64_t) * (size_t)(scan_tiles(std::max<int64_t>(rs_table_len(tot), tot)) + 8));
		if (!tile) return PGA_ERR_NOMEM;
	}
	TRY(upload(c, d_off, off.data(), (size_t)W + 1));
	MergeLists L = { W, slot_sz, d_off };
	hipLaunchKernelGGL(k_mg_rank, dim3(nblk(tot)), dim3(BLOCK), 0, c->st, gathered, L, key, val);
	device_scan<I32>(InMgHead{key}, OutExclI32{slot}, tot, (I32 *)tile, OpSum{}, I32{0}, c->st);
	int64_t *box = nullptr;
	HIPCHK(hipHostGetDevicePointer((void **)&box, c->h_cnt, 0)); // the count goes straight into the pinned mirror
	hipLaunchKernelGGL(k_mg_count, dim3(1), dim3(64), 0, c->st, key, slot, tot, box + 10);
	TRY(sync_st(c));
	const uint64_t *ks = key; const uint32_t *vs = val;
	const int64_t A = c->h_cnt[10];
	int32_t *run_start = (int32_t *)c->pool.get(S_MG_RUN, sizeof(int32_t) * (size_t)A + 16);
	pga_arc_part_t *res = (pga_arc_part_t *)c->pool.get(S_MG_OUT, sizeof(pga_arc_part_t) * (size_t)A + 16);
	if (!run_start || !res) return PGA_ERR_NOMEM;
	hipLaunchKernelGGL(k_mg_runstart, dim3(nblk(tot)), dim3(BLOCK), 0, c->st, ks, slot, tot, run_start);
	hipLaunchKernelGGL(k_mg_sum, dim3(nblk(A, BLOCK / WAVE)), dim3(BLOCK), 0, c->st, gathered, vs, tot, A, run_start, res);
	*out = res, *n_out = A;
	return 0;
}


extern "C" int pga_arc_set_current(pga_ctx_t *c, const pga_arc_part_t *arcs, int64_t n_arc, int32_t n_seg, int32_t *deg)
{
	const int n_vtx = 2 * n_seg;
	c->br_n = n_arc, c->br_S = n_seg, c->br_np = 0;
	if (n_vtx) memset(deg, 0, sizeof(int32_t) * (size_t)n_vtx);
	uint64_t *ax = (uint64_t *)c->pool.get(S_ARCX, sizeof(uint64_t) * (size_t)n_arc + 16);
	uint8_t *aw = (uint8_t *)c->pool.get(S_ARCW, (size_t)n_arc + 16);
	int32_t *s1 = (int32_t *)c->pool.get(S_BR_S1, sizeof(int32_t) * (size_t)n_arc + 16), *agid = (int32_t *)c->pool.get(S_BR_GID, sizeof(int32_t) * (size_t)n_arc + 16);
	int32_t *vs = (int32_t *)c->pool.get(S_BR_VS, sizeof(int32_t) * (size_t)n_vtx + 16), *ve = (int32_t *)c->pool.get(S_BR_VE, sizeof(int32_t) * (size_t)n_vtx + 16);
	int32_t *sg = (int32_t *)c->pool.get(S_BR_SEGGID, sizeof(int32_t) * (size_t)n_seg + 16), *dg = (int32_t *)c->pool.get(S_BR_PC, sizeof(int32_t) * (size_t)n_vtx + 16);
	if (!ax || !aw || !s1 || !agid || !vs || !ve || !sg || !dg) return PGA_ERR_NOMEM;
	if (n_vtx == 0) return 0;
	zero_multi(c, vs, sizeof(int32_t) * (size_t)n_vtx, ve, sizeof(int32_t) * (size_t)n_vtx, aw, (size_t)n_arc);
	if (n_arc) {
		hipLaunchKernelGGL(k_seg_gid, dim3(nblk(c->Q)), dim3(BLOCK), 0, c->st, c->g2s, c->Q, n_seg, sg);
		hipLaunchKernelGGL(k_cur_prep, dim3(nblk(n_arc)), dim3(BLOCK), 0, c->st, arcs, n_arc, sg, ax, s1, agid, vs, ve);
	}
	hipLaunchKernelGGL(k_deg, dim3(nblk(n_vtx)), dim3(BLOCK), 0, c->st, vs, ve, n_vtx, dg);
	HIPCHK(hipMemcpyAsync(deg, dg, sizeof(int32_t) * (size_t)n_vtx, hipMemcpyDeviceToHost, c->st));
	return sync_st(c);
}

extern "C" int pga_rep_pos(pga_ctx_t *c)
{
	const int N = c->N, GL = c->n_genome, Q = c->Q;
	const int64_t n_ent = (int64_t)Q * GL;
	int32_t *rp_pos = (int32_t *)c->pool.get(S_RP_POS, sizeof(int32_t) * (size_t)n_ent);
	int4 *rp = (int4 *)c->pool.get(S_RP_SEG, sizeof(int4) * (size_t)n_ent);
	if (!rp_pos || !rp) return PGA_ERR_NOMEM;
	HIPCHK(hipMemsetAsync(rp_pos, 0, sizeof(int32_t) * (size_t)n_ent, c->st));
	if (N) {
		int32_t *wk = (int32_t *)c->pool.get(S_I32_A, sizeof(int32_t) * (size_t)N);
		int32_t *rx = (int32_t *)c->pool.get(S_I32_B, sizeof(int32_t) * (size_t)N);
		I32 *tile = (I32 *)c->pool.get(S_TILE, 0);
		if (!wk || !rx || !tile) return PGA_ERR_NOMEM;
		hipLaunchKernelGGL(k_walk_x, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->flags, N, wk);
		device_scan<I32>(InI32{wk}, OutExclI32{rx}, N, tile, OpSum{}, I32{0}, c->st);
		hipLaunchKernelGGL(k_rep_last, dim3(nblk(N)), dim3(BLOCK), 0, c->st, wk, c->gnm, c->gid, N, GL, rp_pos);
		hipLaunchKernelGGL(k_hz_cs, dim3(nblk(N)), dim3(BLOCK), 0, c->st, wk, c->seg, c->cs, N, c->dcnt);
		if (n_ent && c->rp_compact) hipLaunchKernelGGL((k_rep_fill<true>), dim3(nblk(n_ent)), dim3(BLOCK), 0, c->st, rp_pos, n_ent, GL, c->seg, c->cm, rx, c->goff, c->ctg_base, (void *)rp);
		else if (n_ent) hipLaunchKernelGGL((k_rep_fill<false>), dim3(nblk(n_ent)), dim3(BLOCK), 0, c->st, rp_pos, n_ent, GL, c->seg, c->cm, rx, c->goff, c->ctg_base, (void *)rp);
	} else if (n_ent) {
		hipLaunchKernelGGL(k_fill_i32, dim3(nblk(4 * n_ent)), dim3(BLOCK), 0, c->st, (int32_t *)rp, 4 * n_ent, -1); // "absent" in either record form
	}
	return 0;
}

static int n_local_dev(pga_ctx *c, const int32_t *d_pairs, int64_t n, int32_t local_dist, int32_t local_count, int32_t frag_mode, int32_t **cnt)
{
	int32_t *d_cnt = (int32_t *)c->pool.get(S_NLCNT, sizeof(int32_t) * (size_t)n + 16);
	int4 *rp = (int4 *)c->pool.get(S_RP_SEG, 0);
	if (!d_cnt || !rp) return PGA_ERR_NOMEM;
	*cnt = d_cnt;
	if (n && c->rp_compact) hipLaunchKernelGGL((k_n_local<true>), dim3(nblk(n, BLOCK / WAVE)), dim3(BLOCK), 0, c->st, d_pairs, n, c->n_genome, (const void *)rp, local_dist, local_count, frag_mode, d_cnt);
	else if (n) hipLaunchKernelGGL((k_n_local<false>), dim3(nblk(n, BLOCK / WAVE)), dim3(BLOCK), 0, c->st, d_pairs, n, c->n_genome, (const void *)rp, local_dist, local_count, frag_mode, d_cnt);
	return 0;
}

extern "C" int pga_n_local(pga_ctx_t *c, const int32_t *pairs, int64_t n, int32_t local_dist, int32_t local_count, int32_t frag_mode, int32_t **cnt)
{
	int32_t *d_pairs = (int32_t *)c->pool.get(S_PAIRS, sizeof(int32_t) * 2 * (size_t)n + 16);
	if (!d_pairs) return PGA_ERR_NOMEM;
	if (n) TRY(upload(c, d_pairs, pairs, 2 * (size_t)n));
	TRY(n_local_dev(c, d_pairs, n, local_dist, local_count, frag_mode, cnt));
	return sync_st(c); // pairs is caller memory; the exchange may run on another stream
}

extern "C" int pga_branch_pairs(pga_ctx_t *c, const uint64_t *arc_x, const int32_t *arc_s1, int64_t n_arc, const int32_t *seg_gid, int32_t n_seg,
                                double branch_diff, int32_t local_dist, int32_t local_count, int32_t frag_mode, int32_t **cnt, int64_t *n_pairs)
{
	if (arc_x == nullptr) n_arc = c->br_n, n_seg = c->br_S; // the table of pga_arc_set_current
	const int n_vtx = 2 * n_seg;
	uint64_t *ax = (uint64_t *)c->pool.get(S_ARCX, sizeof(uint64_t) * (size_t)n_arc + 16);
	uint8_t *aw = (uint8_t *)c->pool.get(S_ARCW, (size_t)n_arc + 16);
	int32_t *s1 = (int32_t *)c->pool.get(S_BR_S1, sizeof(int32_t) * (size_t)n_arc + 16), *agid = (int32_t *)c->pool.get(S_BR_GID, sizeof(int32_t) * (size_t)n_arc + 16);
	int32_t *vs = (int32_t *)c->pool.get(S_BR_VS, sizeof(int32_t) * (size_t)n_vtx + 16), *ve = (int32_t *)c->pool.get(S_BR_VE, sizeof(int32_t) * (size_t)n_vtx + 16);
	int32_t *pc = (int32_t *)c->pool.get(S_BR_PC, sizeof(int32_t) * (size_t)n_vtx + 16), *poff = (int32_t *)c->pool.get(S_BR_POFF, sizeof(int32_t) * (size_t)n_vtx + 16);
	int32_t *sg = (int32_t *)c->pool.get(S_BR_SEGGID, sizeof(int32_t) * (size_t)n_seg + 16);
	if (!ax || !aw || !s1 || !agid || !vs || !ve || !pc || !poff || !sg) return PGA_ERR_NOMEM;
	c->br_n = n_arc, c->br_S = n_seg, c->br_np = 0;
	*n_pairs = 0, *cnt = (int32_t *)c->pool.get(S_NLCNT, 16);
	if (n_arc == 0 || n_vtx == 0) return sync_st(c);
	if (arc_x) {
		TRY(upload(c, ax, arc_x, (size_t)n_arc)); TRY(upload(c, s1, arc_s1, (size_t)n_arc)); TRY(upload(c, sg, seg_gid, (size_t)n_seg));
		HIPCHK(hipMemsetAsync(vs, 0, sizeof(int32_t) * (size_t)n_vtx, c->st)); HIPCHK(hipMemsetAsync(ve, 0, sizeof(int32_t) * (size_t)n_vtx, c->st));
		hipLaunchKernelGGL(k_br_prep, dim3(nblk(n_arc)), dim3(BLOCK), 0, c->st, ax, n_arc, sg, agid, vs, ve);
	}
	HIPCHK(hipMemsetAsync(aw, 0, (size_t)n_arc, c->st));
	hipLaunchKernelGGL(k_br_count, dim3(nblk(n_vtx)), dim3(BLOCK), 0, c->st, n_vtx, vs, ve, s1, branch_diff, pc);
	I32 *tile = (I32 *)c->pool.get(S_TILE, 0);
	device_scan<I32>(InI32{pc}, OutExclI32{poff}, n_vtx, tile, OpSum{}, I32{0}, c->st);
	hipLaunchKernelGGL(k_mail_sum, dim3(1), dim3(64), 0, c->st, poff + (n_vtx - 1), pc + (n_vtx - 1), c->dcnt, c->h_box);
	TRY(sync_st(c));
	const int64_t np = c->h_cnt[10];
	c->br_np = np, *n_pairs = np;
	int32_t *pairs = (int32_t *)c->pool.get(S_PAIRS, sizeof(int32_t) * 2 * (size_t)np + 16);
	if (!pairs) return PGA_ERR_NOMEM;
	if (np) hipLaunchKernelGGL((k_br_wave<1>), dim3(nblk(n_vtx, BLOCK / WAVE)), dim3(BLOCK), 0, c->st, n_vtx, vs, ve, s1, agid, branch_diff, poff, pairs,
	                           (const int32_t *)nullptr, 0.0, 0.0, (uint8_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, c->dcnt);
	TRY(n_local_dev(c, pairs, np, local_dist, local_count, frag_mode, cnt));
	return 0; // no wait: a consumer that is not on this stream calls pga_sync first
}

extern "C" int pga_branch_decide(pga_ctx_t *c, double branch_diff, double branch_diff_dist, double branch_diff_cut, uint8_t *arc_weak,
                                 int32_t *n_dist_loci, int64_t *n_flt1, int64_t *n_flt2)
{
	const int n_vtx = 2 * c->br_S;
	const int64_t n_arc = c->br_n;
	if (n_flt1) *n_flt1 = 0;
	if (n_flt2) *n_flt2 = 0;
	if (n_vtx) memset(n_dist_loci, 0, sizeof(int32_t) * (size_t)n_vtx);
	if (n_arc == 0 || n_vtx == 0) return 0;
	uint8_t *aw = (uint8_t *)c->pool.get(S_ARCW, 0);
	int32_t *s1 = (int32_t *)c->pool.get(S_BR_S1, 0), *agid = (int32_t *)c->pool.get(S_BR_GID, 0), *vs = (int32_t *)c->pool.get(S_BR_VS, 0), *ve = (int32_t *)c->pool.get(S_BR_VE, 0);
	int32_t *pc = (int32_t *)c->pool.get(S_BR_PC, 0), *poff = (int32_t *)c->pool.get(S_BR_POFF, 0), *cnt = (int32_t *)c->pool.get(S_NLCNT, 0);
	int32_t *grp = (int32_t *)c->pool.get(S_BR_GRP, sizeof(int32_t) * (size_t)n_arc + 16), *ndl = (int32_t *)c->pool.get(S_BR_NDL, sizeof(int32_t) * (size_t)n_vtx + 16);
	if (!grp || !ndl) return PGA_ERR_NOMEM;
	zero_multi(c, grp, sizeof(int32_t) * (size_t)n_arc, ndl, sizeof(int32_t) * (size_t)n_vtx);
	hipLaunchKernelGGL((k_br_wave<2>), dim3(nblk(n_vtx, BLOCK / WAVE)), dim3(BLOCK), 0, c->st, n_vtx, vs, ve, s1, agid, branch_diff, poff, (int32_t *)nullptr, cnt,
	                   branch_diff_dist, branch_diff_cut, aw, grp, ndl, c->dcnt);
	if (arc_weak) HIPCHK(hipMemcpyAsync(arc_weak, aw, (size_t)n_arc, hipMemcpyDeviceToHost, c->st));
	HIPCHK(hipMemcpyAsync(n_dist_loci, ndl, sizeof(int32_t) * (size_t)n_vtx, hipMemcpyDeviceToHost, c->st));
	TRY(sync_st(c));
	int64_t f1 = 0, f2 = 0;
	if (arc_weak) for (int64_t i = 0; i < n_arc; ++i) f1 += arc_weak[i] == 1, f2 += arc_weak[i] == 2;
	if (n_flt1) *n_flt1 = f1;
	if (n_flt2) *n_flt2 = f2;
	return 0;
}

extern "C" int pga_mark_hits(pga_ctx_t *c, const uint64_t *arc_x, const uint8_t *arc_weak, int64_t n_arc, int64_t *n_marked)
{
	const int N = c->N;
	if (n_marked) *n_marked = 0;
	if (N == 0) return 0;
	if (arc_x == nullptr) n_arc = c->br_n; // the arcs (and weak_br) left resident by branch_pairs / branch_decide
	uint64_t *ax = (uint64_t *)c->pool.get(S_ARCX, arc_x ? sizeof(uint64_t) * (size_t)n_arc + 16 : 0);
	uint8_t *aw = (uint8_t *)c->pool.get(S_ARCW, arc_x ? (size_t)n_arc + 16 : 0);
	int32_t *wn = (int32_t *)c->pool.get(S_WEAKNEW, sizeof(int32_t) * (size_t)N);
	if (!ax || !aw || !wn) return PGA_ERR_NOMEM;
	if (arc_x) { TRY(upload(c, ax, arc_x, (size_t)n_arc)); TRY(upload(c, aw, arc_weak, (size_t)n_arc)); }
	zero_multi(c, wn, sizeof(int32_t) * (size_t)N, c->dcnt + 2, sizeof(int64_t));
	int32_t *val, *prev;
	TRY(walk_prev(c, &val, &prev));
	const int32_t *vs = arc_x ? nullptr : (const int32_t *)c->pool.get(S_BR_VS, 0), *ve = arc_x ? nullptr : (const int32_t *)c->pool.get(S_BR_VE, 0);
	ensure_yrec(c);
	hipLaunchKernelGGL(k_mark_hits, dim3(nblk(N)), dim3(BLOCK), 0, c->st, val, prev, c->yrecA, c->yrecB, c->g2s, N, ax, aw, n_arc, vs, ve, wn);
	hipLaunchKernelGGL(k_weak_merge, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->flags, wn, N, n_marked ? c->dcnt + 2 : (int64_t *)nullptr);
	if (n_marked) {
		HIPCHK(hipMemcpyAsync(c->h_cnt, c->dcnt, 16 * sizeof(int64_t), hipMemcpyDeviceToHost, c->st));
		TRY(sync_st(c));
		*n_marked = c->h_cnt[2];
	}
	return 0;
}


extern "C" int pga_override_order(pga_ctx_t *c, int32_t which, int32_t n_seg, const int32_t *seg_genome, const int32_t *seg_start,
                                  const int64_t *seg_off, const int32_t *file_idx)
{
	const int N = c->N;
	c->walk_valid = false, c->yrec_valid = false;
	if (n_seg <= 0 || N == 0) return 0;
	const int64_t T = seg_off[n_seg];
	if (T == 0) return 0;
	std::vector<int32_t> pos((size_t)T), fil((size_t)T);
	for (int32_t s = 0; s < n_seg; ++s) {
		const int32_t g = seg_genome[s], base = c->h_goff[(size_t)g];
		for (int64_t k = seg_off[s]; k < seg_off[s + 1]; ++k)
			pos[(size_t)k] = base + seg_start[s] + (int32_t)(k - seg_off[s]), fil[(size_t)k] = base + file_idx[k];
	}
	int32_t *d_pos = (int32_t *)c->pool.get(S_OVPOS, sizeof(int32_t) * (size_t)T), *d_fil = (int32_t *)c->pool.get(S_OVFILE, sizeof(int32_t) * (size_t)T);
	int32_t *inv = (int32_t *)c->pool.get(S_I32_A, sizeof(int32_t) * (size_t)N), *remap = (int32_t *)c->pool.get(S_I32_B, sizeof(int32_t) * (size_t)N);
	if (!d_pos || !d_fil || !inv || !remap) return PGA_ERR_NOMEM;
	TRY(upload(c, d_pos, pos.data(), (size_t)T)); TRY(upload(c, d_fil, fil.data(), (size_t)T));
	hipLaunchKernelGGL(k_ov_inv, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->fidx, c->gnm, c->goff, N, inv, remap);
	if (which == 1) {
		hipLaunchKernelGGL(k_ov_sety, dim3(nblk(T)), dim3(BLOCK), 0, c->st, d_pos, d_fil, T, inv, c->yperm);
		return sync_st(c);
	}
	int32_t *tmp = (int32_t *)c->pool.get(S_PERM, sizeof(int32_t) * 18 * (size_t)T + 64);
	if (!tmp) return PGA_ERR_NOMEM;
	PermArrays p = { { c->fidx, c->pid, c->gid, c->cs, c->ce, c->cm, c->cds, c->nex, c->offx, c->sori, c->sadj, c->rank, c->sdom, c->pdom, c->pdom0, (int32_t *)c->flags, c->rk } };
	hipLaunchKernelGGL(k_ov_gather, dim3(nblk(T)), dim3(BLOCK), 0, c->st, p, d_pos, d_fil, T, inv, tmp, remap);
	hipLaunchKernelGGL(k_ov_scatter, dim3(nblk(T)), dim3(BLOCK), 0, c->st, p, d_pos, T, tmp, c->gnm, c->goff);
	hipLaunchKernelGGL(k_ov_remap_y, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->yperm, N, remap);
	SegMax *tile = (SegMax *)c->pool.get(S_TILE, 0);
	device_scan<SegMax>(InSegMax{c->seg, c->ce}, OutSegMax{c->pm}, N, tile, OpSegMax{}, SegMax{SEG_EMPTY, 0}, c->st);
	hipLaunchKernelGGL(k_inv_only, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->fidx, c->gnm, c->goff, N, c->inv);
	pack_records(c);
	return sync_st(c);
}

extern "C" int pga_set_head(pga_ctx_t *c, const int32_t *head_file)
{
	const int GL = c->n_genome;
	if (GL == 0 || c->N == 0) return 0;
	int32_t *d = (int32_t *)c->pool.get(S_OVFILE, sizeof(int32_t) * (size_t)GL);
	if (!d) return PGA_ERR_NOMEM;
	TRY(upload(c, d, head_file, (size_t)GL));
	hipLaunchKernelGGL(k_set_head, dim3(nblk(GL)), dim3(BLOCK), 0, c->st, d, c->goff, c->inv, GL, c->headpos, c->flags);
	return sync_st(c); // head_file is caller memory
}

extern "C" int pga_hazard_segs(pga_ctx_t *c, int32_t *segs, int32_t cap, int64_t *n_total)
{
	hipLaunchKernelGGL(k_mail_flush, dim3(1), dim3(64), 0, c->st, c->dcnt, c->h_box);
	TRY(sync_st(c));
	*n_total = c->h_cnt[14];
	int64_t n = std::min<int64_t>(std::min<int64_t>(*n_total, PGA_HAZARD_CAP), cap);
	const int32_t *list = (const int32_t *)c->pool.get(S_HZLIST, sizeof(int32_t) * PGA_HAZARD_CAP);
	if (n > 0 && list) { HIPCHK(hipMemcpyAsync(segs, list, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->st)); TRY(sync_st(c)); }
	return 0;
}

extern "C" int pga_sync(pga_ctx_t *c) { return sync_st(c); }

extern "C" int pga_fetch_later(pga_ctx_t *c, const void *src_backend, size_t nbytes, const void **host_view)
{
	if (c->h_stage_cap < nbytes) {
		if (c->h_stage) { HIPCHK(hipStreamSynchronize(c->st)); (void)hipHostFree(c->h_stage); c->h_stage = nullptr; }
		HIPCHK(hipHostMalloc(&c->h_stage, nbytes + nbytes / 2 + 256, hipHostMallocDefault));
		c->h_stage_cap = nbytes + nbytes / 2 + 256;
	}
	*host_view = c->h_stage;
	if (nbytes) HIPCHK(hipMemcpyAsync(c->h_stage, src_backend, nbytes, hipMemcpyDeviceToHost, c->st));
	return 0;
}

extern "C" int pga_fetch(pga_ctx_t *c, void *dst_host, const void *src_backend, size_t nbytes)
{
	if (nbytes == 0) return 0;
	HIPCHK(hipMemcpyAsync(dst_host, src_backend, nbytes, hipMemcpyDeviceToHost, c->st));
	return sync_st(c);
}

extern "C" int pga_put(pga_ctx_t *c, void *dst_backend, const void *src_host, size_t nbytes)
{
	if (nbytes == 0) return 0;
	HIPCHK(hipMemcpyAsync(dst_backend, src_host, nbytes, hipMemcpyHostToDevice, c->st));
	return sync_st(c);
}

extern "C" int pga_copy(pga_ctx_t *c, void *dst_backend, const void *src_backend, size_t nbytes)
{
	if (nbytes == 0) return 0;
	HIPCHK(hipMemcpyAsync(dst_backend, src_backend, nbytes, hipMemcpyDeviceToDevice, c->st));
	return sync_st(c);
}

extern "C" int pga_scratch(pga_ctx_t *c, size_t nbytes, void **ptr)
{
	*ptr = c->pool.get(S_SCRATCH, nbytes);
	return *ptr ? 0 : PGA_ERR_NOMEM;
}

extern "C" int pga_download(pga_ctx_t *c, const pga_hit_state_t *o)
{
	const int N = c->N;
	if (N == 0) return 0;
	if (o->flt_x_bits) {
		unsigned long long *bits = (unsigned long long *)c->pool.get(S_MISC, sizeof(uint64_t) * (size_t)((N + 63) / 64) + 16);
		if (!bits) return PGA_ERR_NOMEM;
		hipLaunchKernelGGL(k_flt_bits, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->flags, N, bits);
		HIPCHK(hipMemcpyAsync(o->flt_x_bits, bits, sizeof(uint64_t) * (size_t)((N + 63) / 64), hipMemcpyDeviceToHost, c->st));
		if (!o->flags && !o->rank && !o->score_dom && !o->pid_dom && !o->pid_dom0 && !o->pos_x && !o->pos_y) return sync_st(c);
	}
	int32_t *dl = (int32_t *)c->pool.get(S_DL, sizeof(int32_t) * 7 * (size_t)N);
	if (!dl) return PGA_ERR_NOMEM;
	hipLaunchKernelGGL(k_to_file, dim3(nblk(N)), dim3(BLOCK), 0, c->st, c->fidx, c->gnm, c->goff, c->flags, c->rank, c->sdom, c->pdom, c->pdom0, c->yperm, N,
	                   (uint32_t *)dl, dl + (size_t)N, dl + 2 * (size_t)N, dl + 3 * (size_t)N, dl + 4 * (size_t)N, dl + 5 * (size_t)N, dl + 6 * (size_t)N);
	void *dst[7] = { o->flags, o->rank, o->score_dom, o->pid_dom, o->pid_dom0, o->pos_x, o->pos_y };
	for (int k = 0; k < 7; ++k)
		if (dst[k]) HIPCHK(hipMemcpyAsync(dst[k], dl + (size_t)k * N, sizeof(int32_t) * (size_t)N, hipMemcpyDeviceToHost, c->st));
	return sync_st(c);
}

extern "C" int pga_hazards(pga_ctx_t *c, pga_hazard_t *out)
{
	HIPCHK(hipMemcpyAsync(c->h_cnt, c->dcnt, 16 * sizeof(int64_t), hipMemcpyDeviceToHost, c->st));
	TRY(sync_st(c));
	out->h1_head_tie = c->h_cnt[4], out->h2_cm_tie = c->h_cnt[5], out->h2_cs_tie = c->h_cnt[6], out->h3_dom_tie = c->h_cnt[7];
	return 0;
}

extern "C" int pga_timing_reset(pga_ctx_t *c)
{
	TRY(sync_st(c));
	for (auto &t : c->timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
	c->timed.clear();
	return 0;
}

extern "C" int pga_timing_get(pga_ctx_t *c, int32_t which, double *total_ms, int64_t *n_launch, int64_t *units)
{
	TRY(sync_st(c));
	double ms = 0; int64_t n = 0, u = 0;
	for (auto &t : c->timed) {
		if (t.which != which) continue;
		float f = 0;
		HIPCHK(hipEventElapsedTime(&f, t.a, t.b));
		ms += f, ++n, u += t.units;
	}
	if (total_ms) *total_ms = ms;
	if (n_launch) *n_launch = n;
	if (units) *units = u;
	return 0;
}

extern "C" const pga_backend_t *pga_backend(void)
{
	static const pga_backend_t b = {
		"hip-gfx950", pga_create, pga_destroy, pga_begin, pga_ingest, pga_post_partials, pga_post_apply, pga_shadow, pga_set_filter,
		pga_vtx_partials, pga_flag_vtx, pga_arc_round, pga_arc_merge, pga_arc_set_current, pga_rep_pos, pga_n_local, pga_branch_pairs, pga_branch_decide, pga_mark_hits, pga_override_order, pga_set_head, pga_fetch, pga_put, pga_copy, pga_scratch,
		pga_download, pga_hazards, pga_is_device, pga_strerror, pga_timing_reset, pga_timing_get, pga_sync, pga_fetch_later, pga_hazard_segs
	};
	return &b;
}

// ------------------------------------------------------------------------------------------------
// self-test hooks for the device primitives (tests/test_prims_gpu.py): sort / scan arbitrary host data
// ------------------------------------------------------------------------------------------------
extern "C" int pga_selftest_sort(uint64_t *keys, uint32_t *vals, int64_t n, int32_t n_bits)
{
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return PGA_ERR_NO_DEVICE;
	uint64_t *ka, *kb; uint32_t *va, *vb, *table; int32_t *tile;
	HIPCHK(hipMalloc((void **)&ka, sizeof(uint64_t) * (size_t)(n + 1))); HIPCHK(hipMalloc((void **)&kb, sizeof(uint64_t) * (size_t)(n + 1)));
	HIPCHK(hipMalloc((void **)&va, sizeof(uint32_t) * (size_t)(n + 1))); HIPCHK(hipMalloc((void **)&vb, sizeof(uint32_t) * (size_t)(n + 1)));
	HIPCHK(hipMalloc((void **)&table, sizeof(uint32_t) * (size_t)(rs_table_len(n) + 1)));
	HIPCHK(hipMalloc((void **)&tile, sizeof(int64_t) * (size_t)(scan_tiles(std::max<int64_t>(rs_table_len(n), n)) + 8)));
	HIPCHK(hipMemcpy(ka, keys, sizeof(uint64_t) * (size_t)n, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(va, vals, sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice));
	RadixBufs b = { kb, vb, table, tile };
	uint64_t *kr; uint32_t *vr;
	device_radix_sort(ka, va, n, n_bits, b, &kr, &vr, 0);
	HIPCHK(hipDeviceSynchronize());
	HIPCHK(hipMemcpy(keys, kr, sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(vals, vr, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost));
	(void)hipFree(ka); (void)hipFree(kb); (void)hipFree(va); (void)hipFree(vb); (void)hipFree(table); (void)hipFree(tile);
	return 0;
}

// cross-shard arc merge on host data: `gathered` holds W slots of slot_sz entries (count[r] valid, sorted by x, unique keys)
extern "C" int pga_selftest_merge(const pga_arc_part_t *gathered, const int64_t *count, int32_t W, int64_t slot_sz, pga_arc_part_t *out, int64_t *n_out)
{
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return PGA_ERR_NO_DEVICE;
	pga_ctx c; // a bare context: stream, counters, pool
	HIPCHK(hipStreamCreateWithFlags(&c.st, hipStreamNonBlocking));
	TRY(dalloc(&c, &c.dcnt, 16));
	HIPCHK(hipHostMalloc((void **)&c.h_cnt, 16 * sizeof(int64_t), hipHostMallocDefault));
	pga_arc_part_t *dg = nullptr, *res = nullptr;
	HIPCHK(hipMalloc((void **)&dg, sizeof(pga_arc_part_t) * (size_t)(W * slot_sz + 1)));
	HIPCHK(hipMemcpy(dg, gathered, sizeof(pga_arc_part_t) * (size_t)(W * slot_sz), hipMemcpyHostToDevice));
	int rc = pga_arc_merge(&c, dg, count, W, slot_sz, &res, n_out);
	if (rc == 0 && *n_out) rc = hipMemcpyAsync(out, res, sizeof(pga_arc_part_t) * (size_t)*n_out, hipMemcpyDeviceToHost, c.st) == hipSuccess ? 0 : PGA_ERR_NO_DEVICE;
	(void)hipStreamSynchronize(c.st);
	(void)hipFree(dg); (void)hipFree(c.dcnt); (void)hipHostFree(c.h_cnt);
	c.pool.release();
	(void)hipStreamDestroy(c.st);
	return rc;
}

// mode 0: exclusive sum; 1: exclusive max (identity -1); 2: segmented inclusive max with seg[]
extern "C" int pga_selftest_scan(const int32_t *in, const int32_t *seg, int32_t *out, int64_t n, int32_t mode)
{
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return PGA_ERR_NO_DEVICE;
	int32_t *di, *ds, *dout; int64_t *tile;
	HIPCHK(hipMalloc((void **)&di, sizeof(int32_t) * (size_t)(n + 1))); HIPCHK(hipMalloc((void **)&ds, sizeof(int32_t) * (size_t)(n + 1)));
	HIPCHK(hipMalloc((void **)&dout, sizeof(int32_t) * (size_t)(n + 1))); HIPCHK(hipMalloc((void **)&tile, sizeof(int64_t) * (size_t)(scan_tiles(n) + 8)));
	HIPCHK(hipMemcpy(di, in, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice));
	if (seg) HIPCHK(hipMemcpy(ds, seg, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice));
	if (mode == 0) device_scan<I32>(InI32{di}, OutExclI32{dout}, n, (I32 *)tile, OpSum{}, I32{0}, 0);
	else if (mode == 1) device_scan<I32>(InI32{di}, OutExclI32{dout}, n, (I32 *)tile, OpMax{}, I32{-1}, 0);
	else device_scan<SegMax>(InSegMax{ds, di}, OutSegMax{dout}, n, (SegMax *)tile, OpSegMax{}, SegMax{SEG_EMPTY, 0}, 0);
	HIPCHK(hipDeviceSynchronize());
	HIPCHK(hipMemcpy(out, dout, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost));
	(void)hipFree(di); (void)hipFree(ds); (void)hipFree(dout); (void)hipFree(tile);
	return 0;
}
