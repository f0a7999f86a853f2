// k_genes.hpp -- pg_gen_arc (graph.c:87-177) and pg_mark_branch_flt_hit (branch.c:108-145) on a GENE-major index, without a global sort.
// Included by pga_backend.hip (one translation unit); uses the context types, BLOCK / WAVE and dev_prims.hpp from there.
//
// The reference sorts all 2 x (#adjacencies) temp arcs by (v, w) every round (graph.c:127,151).  The source vertex of a temp
// arc is an oriented GENE, and which hits belong to a gene never changes: with the hits indexed gene-major once per run
// (Z order = (gene, genome, X position); zoff[] = CSR offsets), every temp arc leaving the two vertices of gene g comes from
// a hit of g -- the arc v -> w from the hit that plays v (its successor adjacency), the mirrored arc w^1 -> v^1 from the hit
// that plays w (its predecessor adjacency).  So:
//   (A) the scan over the cm order that finds each walkable hit's walkable predecessor also leaves, per hit, two 16-byte
//       half-arc records (successor / predecessor adjacency: target vertex, distance, the two scores);
//   (B) one workgroup per gene reads its hits' half-arcs (sequential in Z order), collapses the (arc, genome) duplicates
//       (graph.c:128-145; genomes are contiguous inside a gene), sums over the genomes in an LDS hash table keyed by
//       (orientation, target) (graph.c:153-169: integer sums, order-free), sorts the few entries and writes the gene's arcs;
//       it also counts the gene's walkable hits and genomes (graph.c:125-126) -- no global atomics except one per gene;
//   (C) a scan over the segments places every gene's arcs: the table comes out sorted by x = v << 32 | w because
//       segments are numbered in gene order (vertex.c:85-94).
// pg_mark_branch_flt_hit then needs no walk at all: every hit looks its own two half-arcs up in the arc lists of its own
// gene's vertices and raises its own weak_br.
#pragma once

constexpr uint32_t HA_NONE = 0x1fffffu;  // "no adjacency" target (gene ids stay below 2^20 - 1)
constexpr int HA_TAG_SHIFT = 21;         // half-arc word 0 = round tag << 21 | target vertex (gene << 1 | rev)
constexpr uint32_t HA_TAG_MAX = 0x7ffu;

// ------------------------------------------------------------------------------------------------
// the gene-major index (static per run: rebuilt only when an order override moves hits)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_zkey(const int32_t *gid, int n, uint64_t *key, uint32_t *val)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h < n) key[h] = (uint64_t)(uint32_t)gid[h], val[h] = (uint32_t)h;
}

// zrec[z] = {X position, local genome << 1 | rev}; zpos[x] = z
__global__ __launch_bounds__(BLOCK) void k_zrec(const uint32_t *perm, const int32_t *gnm, const uint32_t *flags, int n, int2 *zrec, int32_t *zpos)
{
	int z = blockIdx.x * BLOCK + threadIdx.x;
	if (z >= n) return;
	const int x = (int)perm[z];
	zrec[z] = make_int2(x, gnm[x] << 1 | ((flags[x] & PGA_F_REV) ? 1 : 0));
	zpos[x] = z;
}

__global__ __launch_bounds__(BLOCK) void k_zoff(const uint64_t *ks, int n, int Q, int32_t *zoff) // zoff[g] = first z with gene >= g
{
	int g = blockIdx.x * BLOCK + threadIdx.x;
	if (g > Q) return;
	int lo = 0, hi = n;
	while (lo < hi) { const int mid = (lo + hi) >> 1; if (ks[mid] < (uint64_t)g) lo = mid + 1; else hi = mid; }
	zoff[g] = lo;
}

__global__ __launch_bounds__(BLOCK) void k_zpos_y(const int32_t *yperm, const int32_t *zpos, int n, int32_t *zposy)
{
	int y = blockIdx.x * BLOCK + threadIdx.x;
	if (y < n) zposy[y] = zpos[yperm[y]];
}

// ------------------------------------------------------------------------------------------------
// (A) output step of the walk scan (cm order): previous walkable hit -> half-arc records
// ------------------------------------------------------------------------------------------------
struct OutHalfArcs {
	const int4 *YA, *YB; const int32_t *zposy, *g2s; int4 *hf, *hb; uint32_t tag; int ori; int64_t *dcnt; int32_t *hz_list;
	__device__ __forceinline__ void operator()(int64_t i, I32 incl, I32 ex) const
	{
		if (incl.v == ex.v) return; // not walkable (graph.c:108)
		const int p = ex.v;
		const int4 aA = YA[i]; // {seg, gid, genome, cm}
		int4 rec = make_int4((int)(tag << HA_TAG_SHIFT | HA_NONE), 0, 0, 0);
		if (p >= 0) {
			const int4 bA = YA[p];
			if (bA.x == aA.x) { // same contig: adjacency p -> i (graph.c:113-121)
				const int4 aB = YB[i], bB = YB[p];
				const uint32_t w = (uint32_t)aA.y << 1 | (uint32_t)(aB.w & 1), v = (uint32_t)bA.y << 1 | (uint32_t)(bB.w & 1);
				const int sa = arc_score(aB, ori, g2s), sb = arc_score(bB, ori, g2s), d = aA.w - bA.w;
				if (aA.w == bA.w) { atomicAdd((unsigned long long *)&dcnt[5], 1ull); hz_note(&dcnt[14], hz_list, aA.x); } // hazard H2a: equal cm
				hf[zposy[p]] = make_int4((int)(tag << HA_TAG_SHIFT | w), d, sb, sa);       // v -> w,       s1 = score(v), s2 = score(w) (graph.c:117)
				rec = make_int4((int)(tag << HA_TAG_SHIFT | (v ^ 1u)), d, sa, sb);          // w^1 -> v^1,   s1 = score(w), s2 = score(v) (graph.c:119)
			}
		}
		hb[zposy[i]] = rec;
	}
};

// ------------------------------------------------------------------------------------------------
// (B) one workgroup per gene
// ------------------------------------------------------------------------------------------------
constexpr int GA_CAP = 512; // distinct (orientation, target) pairs of one gene the LDS table holds; more = the round takes the sort path

struct GeneArcs {
	const int2 *zrec; const int32_t *zoff; const uint32_t *flags; const int4 *hf, *hb; const int32_t *g2s;
	int Q, S; uint32_t tag; int cap_log2; // table size actually used (<= GA_CAP; tests shrink it to reach the overflow path)
	int32_t *seg_cnt, *seg_gid;       // [2S] n_genome then tot_cnt (graph.c:125-126); [S] gene of each segment
	pga_arc_part_t *stage; int4 *gmeta; // arcs of a gene at stage[gmeta.x ...): gmeta = {base, #arcs leaving (sid, +), #arcs leaving (sid, -), 0}
	int64_t *dcnt;                      // [3] invariant, [8] staged arcs, [9] genes that overflowed the table
};

__device__ __forceinline__ bool ha_valid(const int4 h, uint32_t tag) { return ((uint32_t)h.x >> HA_TAG_SHIFT) == tag && ((uint32_t)h.x & HA_NONE) != HA_NONE; }

__global__ __launch_bounds__(BLOCK) void k_gene_arcs(GeneArcs a)
{
	__shared__ uint32_t t_key[GA_CAP];
	__shared__ int32_t t_ng[GA_CAP], t_tot[GA_CAP];
	__shared__ unsigned long long t_sd[GA_CAP], t_s1[GA_CAP], t_s2[GA_CAP];
	__shared__ uint16_t t_dense[GA_CAP];
	__shared__ int s_tot, s_ngen, s_over, s_m, s_m0, s_base;
	const int g = blockIdx.x, tid = threadIdx.x;
	const int sid = a.g2s[g], z0 = a.zoff[g], z1 = a.zoff[g + 1];
	if (sid < 0) { // not a vertex: none of its hits may be walkable (graph.c:111)
		for (int z = z0 + tid; z < z1; z += BLOCK)
			if (!(a.flags[a.zrec[z].x] & (PGA_F_FLT | PGA_F_SHADOW))) atomicAdd((unsigned long long *)&a.dcnt[3], 1ull);
		return;
	}
	for (int k = tid; k < GA_CAP; k += BLOCK) t_key[k] = 0xffffffffu, t_ng[k] = 0, t_tot[k] = 0, t_sd[k] = 0, t_s1[k] = 0, t_s2[k] = 0;
	if (tid == 0) s_tot = 0, s_ngen = 0, s_over = 0, s_m = 0, s_m0 = 0;
	__syncthreads();
	for (int z = z0 + tid; z < z1; z += BLOCK) {
		const int2 zr = a.zrec[z];
		if (a.flags[zr.x] & (PGA_F_FLT | PGA_F_SHADOW)) continue;
		const int genome = zr.y >> 1, rev = zr.y & 1;
		// the hits of this gene in this genome: [gs, ge) around z (one or two as a rule)
		int gs = z, ge = z + 1;
		while (gs > z0 && (a.zrec[gs - 1].y >> 1) == genome) --gs;
		while (ge < z1 && (a.zrec[ge].y >> 1) == genome) ++ge;
		bool first = true; // first walkable hit of the group: it counts the genome (graph.c:125)
		for (int q = gs; q < z; ++q) first = first && (a.flags[a.zrec[q].x] & (PGA_F_FLT | PGA_F_SHADOW)) != 0;
		atomicAdd(&s_tot, 1);
		if (first) atomicAdd(&s_ngen, 1);
#pragma unroll
		for (int dir = 0; dir < 2; ++dir) {
			const int4 h = dir ? a.hb[z] : a.hf[z];
			if (!ha_valid(h, a.tag)) continue;
			const uint32_t key = (uint32_t)(dir ? !rev : rev) << HA_TAG_SHIFT | ((uint32_t)h.x & HA_NONE);
			// level 1 (graph.c:128-145): the first item of the group with this key collapses the group's items with the same key
			bool leader = true;
			int n = 1, m1 = h.z, m2 = h.w;
			unsigned long long sd = (unsigned long long)(long long)h.y;
			for (int q = gs; q < ge && leader; ++q) {
				const int rq = a.zrec[q].y & 1;
#pragma unroll
				for (int d2 = 0; d2 < 2; ++d2) {
					if (q == z && d2 == dir) continue;
					const int4 o = d2 ? a.hb[q] : a.hf[q];
					if (!ha_valid(o, a.tag) || ((uint32_t)(d2 ? !rq : rq) << HA_TAG_SHIFT | ((uint32_t)o.x & HA_NONE)) != key) continue;
					if (q < z || (q == z && d2 < dir)) { leader = false; break; }
					++n, sd += (unsigned long long)(long long)o.y, m1 = m1 > o.z ? m1 : o.z, m2 = m2 > o.w ? m2 : o.w;
				}
			}
			if (!leader) continue;
			m1 = m1 > 0 ? m1 : 0, m2 = m2 > 0 ? m2 : 0; // the reference's running maxima start at 0 (graph.c:133)
			const int dg = (int32_t)((double)(long long)sd / n + .499); // graph.c:141
			// level 2 (graph.c:153-169): sums over the genomes, LDS table keyed by (orientation, target)
			const int cap = 1 << a.cap_log2;
			uint32_t slot = (key * 2654435761u) >> (32 - a.cap_log2);
			int probes = 0;
			for (; probes < cap; ++probes, slot = (slot + 1) & (cap - 1)) {
				const uint32_t old = atomicCAS(&t_key[slot], 0xffffffffu, key);
				if (old == 0xffffffffu || old == key) break;
			}
			if (probes == cap) { s_over = 1; continue; }
			atomicAdd(&t_ng[slot], 1); atomicAdd(&t_tot[slot], n);
			atomicAdd(&t_sd[slot], (unsigned long long)(long long)dg * (unsigned long long)n);
			atomicAdd(&t_s1[slot], (unsigned long long)(long long)m1); atomicAdd(&t_s2[slot], (unsigned long long)(long long)m2);
		}
	}
	__syncthreads();
	if (s_over) { // too many distinct neighbours for the table (a hub gene): this round is redone on the sort path
		if (tid == 0) atomicAdd((unsigned long long *)&a.dcnt[9], 1ull), a.gmeta[sid] = make_int4(0, 0, 0, 0);
		return;
	}
	for (int k = tid; k < GA_CAP; k += BLOCK)
		if (t_key[k] != 0xffffffffu) { const int at = atomicAdd(&s_m, 1); t_dense[at] = (uint16_t)k; if (!(t_key[k] >> HA_TAG_SHIFT)) atomicAdd(&s_m0, 1); }
	__syncthreads();
	const int m = s_m;
	if (tid == 0) {
		s_base = m ? (int)atomicAdd((unsigned long long *)&a.dcnt[8], (unsigned long long)m) : 0;
		a.seg_cnt[sid] = s_ngen, a.seg_cnt[a.S + sid] = s_tot, a.seg_gid[sid] = g;
	}
	__syncthreads();
	if (tid == 0) a.gmeta[sid] = make_int4(s_base, s_m0, m - s_m0, 0);
	for (int e = tid; e < m; e += BLOCK) { // rank among the gene's entries = place in the table (keys are distinct)
		const int k = t_dense[e];
		const uint32_t key = t_key[k];
		int r = 0;
		for (int j = 0; j < m; ++j) r += t_key[t_dense[j]] < key;
		pga_arc_part_t o;
		const uint32_t t = key & HA_NONE;
		o.x = (uint64_t)((uint32_t)sid << 1 | (key >> HA_TAG_SHIFT)) << 32 | (uint32_t)((uint32_t)a.g2s[t >> 1] << 1 | (t & 1u));
		o.n_genome = t_ng[k], o.tot_cnt = t_tot[k], o.sum_dist = t_sd[k], o.sum_s1 = (int64_t)t_s1[k], o.sum_s2 = (int64_t)t_s2[k];
		a.stage[s_base + r] = o;
	}
}

// ------------------------------------------------------------------------------------------------
// (C) place every gene's arcs (segment order = gene order) and derive what the branch steps read
// ------------------------------------------------------------------------------------------------
struct InGmeta { const int4 *gm; __device__ __forceinline__ I32 operator()(int64_t i) const { const int4 m = gm[i]; return I32{m.y + m.z}; } };

struct ArcFinal {
	const int4 *gmeta; const int32_t *off; int S; const pga_arc_part_t *stage; const int32_t *seg_gid;
	pga_arc_part_t *arcs; uint64_t *ax; int32_t *s1, *agid, *vs, *ve, *deg; uint8_t *aw, *vwk; int64_t *dcnt, *host_box;
};

__global__ __launch_bounds__(BLOCK) void k_arc_final(ArcFinal f)
{
	const int sid = blockIdx.x * BLOCK + threadIdx.x;
	if (sid >= f.S) return;
	const int4 m = f.gmeta[sid];
	const int o = f.off[sid], n = m.y + m.z;
	for (int i = 0; i < n; ++i) {
		const pga_arc_part_t a = f.stage[m.x + i];
		f.arcs[o + i] = a;
		f.ax[o + i] = a.x;
		f.s1[o + i] = (int32_t)((double)a.sum_s1 / a.n_genome + .499); // graph.c:171
		f.agid[o + i] = f.seg_gid[(uint32_t)a.x >> 1];
		f.aw[o + i] = 0;
	}
	f.vs[2 * sid] = o, f.ve[2 * sid] = o + m.y, f.vs[2 * sid + 1] = o + m.y, f.ve[2 * sid + 1] = o + n;
	f.deg[2 * sid] = m.y, f.deg[2 * sid + 1] = m.z;
	f.vwk[2 * sid] = 0, f.vwk[2 * sid + 1] = 0;
	if (sid == f.S - 1) { // table size and the device counters for the host (read after its next wait)
		f.dcnt[10] = o + n;
		__threadfence();
		for (int t = 0; t < 16; ++t) f.host_box[t] = f.dcnt[t];
	}
}

// ------------------------------------------------------------------------------------------------
// pg_mark_branch_flt_hit (branch.c:108-145): a hit is marked by the weak arcs among its own two half-arcs
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_mark_hits_z(const int2 *zrec, const int4 *hf, const int4 *hb, uint32_t tag, int n, const int32_t *g2s, const int32_t *gid,
                                                        const uint64_t *ax, const uint8_t *aw, const int32_t *vs, const int32_t *ve, const uint8_t *vwk,
                                                        uint32_t *flags, int64_t *cnt)
{
	int z = blockIdx.x * BLOCK + threadIdx.x;
	int cur = 0;
	if (z < n) {
		const int2 zr = zrec[z];
		const uint32_t f = flags[zr.x];
		cur = (int)((f & PGA_F_WEAK_MASK) >> PGA_F_WEAK_SHIFT);
		if (!(f & (PGA_F_FLT | PGA_F_SHADOW))) {
			const int sid = g2s[gid[zr.x]], rev = zr.y & 1;
			int nw = 0;
#pragma unroll
			for (int dir = 0; dir < 2; ++dir) {
				const int4 h = dir ? hb[z] : hf[z];
				if (!ha_valid(h, tag) || sid < 0) continue;
				const uint32_t u = (uint32_t)sid << 1 | (uint32_t)(dir ? !rev : rev), t = (uint32_t)h.x & HA_NONE;
				if (!vwk[u]) continue; // the vertex has no weak out-arc (the common case)
				const uint32_t w = (uint32_t)g2s[t >> 1] << 1 | (t & 1u);
				const int e = arc_weak_v(ax, aw, vs, ve, u, w); // dir 0: arc v -> w marks the earlier hit (branch.c:128-130); dir 1: w^1 -> v^1 marks the later one (131-133)
				nw = nw > e ? nw : e;
			}
			if (nw > cur) { cur = nw; flags[zr.x] = (f & ~PGA_F_WEAK_MASK) | (uint32_t)nw << PGA_F_WEAK_SHIFT; }
		}
	}
	if (cnt) { // log-only counter (branch.c:137-139)
		const unsigned long long mk = __ballot(z < n && cur != 0);
		if (mk && (threadIdx.x & 63) == (unsigned)__ffsll((long long)mk) - 1) atomicAdd((unsigned long long *)cnt, (unsigned long long)__popcll(mk));
	}
}
