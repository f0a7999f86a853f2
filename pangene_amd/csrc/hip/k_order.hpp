// k_order.hpp -- per-hit state download and the exact-order overrides.
// Included by pga_backend.hip (one translation unit); uses the context types, BLOCK / WAVE and dev_prims.hpp from there.
#pragma once

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
// rewrite slices of the Y permutation.  Every step works on the t hits of the overridden contigs only (a contig occupies the same
// index range in both orders, so what refers to its hits -- yperm, the gene-major index, inv, pm, the tie marks -- is found through
// the override's own position list): an isoform-rich shard overrides a million hits of twenty-two 67 times a pass, and eight
// shard-wide kernels per override were 110 of the 163 ms of its pass.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_ov_sety(const int32_t *ov_pos, const int32_t *ov_file, int64_t t, const int32_t *inv, int32_t *yperm, const int32_t *lpos /* or NULL */, int32_t *ylist)
{
	int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i >= t) return;
	const int x = inv[ov_file[i]];
	yperm[ov_pos[i]] = x;
	if (lpos && lpos[i] >= 0) ylist[lpos[i]] = x; // the members' list in cm order follows
}

// With live lists (pga_ctx::live_on) an override also has to be told in the lists' coordinates.  A contig keeps its index range in both orders
// and the same members, so its stretch of the members' list starts at lx[first position of the contig] in either order, and the k-th listed hit
// that is a member takes the place (members listed before it in its contig) behind that: a scan over the override's list.  lpos[k] = that
// place, or -1 for a hit that is not a member.  ov_first[k] = the list index of the first entry of k's contig.
struct InOvMember { const uint32_t *flags; const int32_t *inv, *ov_file; __device__ __forceinline__ I32 operator()(int64_t k) const { return I32{(flags[inv[ov_file[k]]] & F_MEMBER) ? 1 : 0}; } };
__global__ __launch_bounds__(BLOCK) void k_ovl_pos(const int32_t *ex, const uint32_t *flags, const int32_t *inv, const int32_t *ov_pos, const int32_t *ov_file, const int32_t *ov_first, int64_t t, const int32_t *lx, int32_t *lpos)
{
	const int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (k >= t) return;
	const int f = ov_first[k];
	lpos[k] = (flags[inv[ov_file[k]]] & F_MEMBER) ? lx[ov_pos[f]] + (ex[k] - ex[f]) : -1;
}
// after a cs override: the entries of the members' cm-order list that lie in the overridden contigs (every place of those stretches is some
// lpos[k]) hold X positions inside those contigs, which were renumbered
__global__ __launch_bounds__(BLOCK) void k_ovl_remap_ylist(int32_t *ylist, const int32_t *lpos, int64_t t, const int32_t *remap)
{
	const int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (k < t && lpos[k] >= 0) ylist[lpos[k]] = remap[ylist[lpos[k]]];
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

// what moves with a hit: the 4-byte planes (OV_FLAGS = the flag word, whose head mark is positional) and the three packed records
constexpr int OV_PLANES = 12, OV_FLAGS = 10;
struct PermArrays { int32_t *a[OV_PLANES]; int4 *r[3]; };

__global__ __launch_bounds__(BLOCK) void k_ov_gather(PermArrays p, const int32_t *ov_pos, const int32_t *ov_file, int64_t t, const int32_t *inv,
                                                       int32_t *tmp, int32_t *remap, const int32_t *zpos /* or NULL */)
{
	int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i >= t) return;
	int src = inv[ov_file[i]];
	remap[src] = ov_pos[i]; // (defined for the hits of the overridden contigs: nobody asks for another)
	if (zpos) tmp[(int64_t)(OV_PLANES + 12) * t + i] = zpos[src]; // the hit's place in the gene-major index moves with it
#pragma unroll
	for (int k = 0; k < OV_PLANES; ++k) tmp[(int64_t)k * t + i] = p.a[k][src];
	int4 *tr = (int4 *)(tmp + (int64_t)OV_PLANES * t);
#pragma unroll
	for (int k = 0; k < 3; ++k) tr[(int64_t)k * t + i] = p.r[k][src];
}

__global__ __launch_bounds__(BLOCK) void k_ov_scatter(PermArrays p, const int32_t *ov_pos, const int32_t *ov_file, int64_t t, const int32_t *tmp,
                                                        const int32_t *gnm, const int32_t *goff, int32_t *inv, int32_t *zx, int32_t *zpos /* or NULL */)
{
	int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i >= t) return;
	int pos = ov_pos[i];
	inv[ov_file[i]] = pos; // (k_ov_gather, the launch before, was the last reader of the old entries)
	if (zpos) { const int z = tmp[(int64_t)(OV_PLANES + 12) * t + i]; zpos[pos] = z; if (z >= 0) zx[z] = pos; } // (z < 0: not in the index -- live lists)
#pragma unroll
	for (int k = 0; k < OV_PLANES; ++k) {
		int32_t v = tmp[(int64_t)k * t + i];
		if (k == OV_FLAGS) { // the head mark is positional
			uint32_t f = (uint32_t)v & ~F_HEAD;
			if (pos == goff[gnm[pos]]) f |= F_HEAD;
			v = (int32_t)f;
		}
		p.a[k][pos] = v;
	}
	const int4 *tr = (const int4 *)(tmp + (int64_t)OV_PLANES * t);
#pragma unroll
	for (int k = 0; k < 3; ++k) p.r[k][pos] = tr[(int64_t)k * t + i];
}

// the Y positions of an overridden contig are its X positions (same index range): their entries point into the contig
__global__ __launch_bounds__(BLOCK) void k_ov_remap_y(int32_t *yperm, const int32_t *ov_pos, int64_t t, const int32_t *remap)
{
	int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i < t) { const int y = ov_pos[i]; yperm[y] = remap[yperm[y]]; }
}

// the static tie marks of the cs order (k_cstie) for the listed positions only
__global__ __launch_bounds__(BLOCK) void k_cstie_list(const int4 *A, const int32_t *ov_pos, int64_t t, int n, uint32_t *flags)
{
	int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (i >= t) return;
	const int h = ov_pos[i];
	const int4 a = A[h];
	bool tie = false;
	if (h > 0) { const int4 p = A[h - 1]; tie = p.y == a.y && p.x == a.x; }
	if (!tie && h + 1 < n) { const int4 q = A[h + 1]; tie = q.y == a.y && q.x == a.x; }
	const uint32_t f = flags[h], nf = tie ? f | F_CSTIE : f & ~F_CSTIE;
	if (nf != f) flags[h] = nf;
}

// pm (record A, word 3) over the listed positions: a contig's positions are consecutive in the list, and a running maximum never
// crosses from one contig into the next (segmented by the contig id)
struct InSegMaxList { const int4 *A; const int32_t *pos; __device__ __forceinline__ SegMax operator()(int64_t i) const { const int4 a = A[pos[i]]; return SegMax{a.y, a.z}; } };
struct OutSegMaxList { int4 *A; const int32_t *pos; __device__ __forceinline__ void operator()(int64_t i, SegMax in, SegMax) const { ((int32_t *)&A[pos[i]])[3] = in.v; } };

__global__ __launch_bounds__(BLOCK) void k_flt_bits(const uint32_t *flags, int n, unsigned long long *bits)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	const unsigned long long m = __ballot(h < n && (flags[h < n ? h : n - 1] & PGA_F_FLT));
	if ((threadIdx.x & 63) == 0 && h < n) bits[h >> 6] = m;
}

// ------------------------------------------------------------------------------------------------
// gfa2matrix (pangene.js:1168-1247): the gene x assembly occurrence matrix as a reduction over the walks
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_ctg_counts(const uint32_t *flags, const int32_t *seg, int n, int32_t *cnt)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h < n && !(flags[h] & PGA_F_FLT)) atomicAdd(&cnt[seg[h]], 1);
}

__global__ __launch_bounds__(BLOCK) void k_gene_matrix(const uint32_t *flags, const int32_t *seg, const int32_t *gid, const int32_t *g2s, int n,
                                                         const int32_t *asm_of_ctg, int n_asm, int32_t *mat)
{
	int h = blockIdx.x * BLOCK + threadIdx.x;
	if (h >= n || (flags[h] & PGA_F_FLT)) return;
	const int sid = g2s[gid[h]], col = asm_of_ctg[seg[h]];
	if (sid >= 0 && col >= 0) atomicAdd(&mat[(int64_t)sid * n_asm + col], 1); // a walk step whose gene is not a segment is ignored (pangene.js:178)
}
