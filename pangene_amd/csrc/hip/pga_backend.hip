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
#include <time.h>
#include <vector>
#include <mutex>
#include <atomic>
#include <condition_variable>
#include <algorithm>
#include "pangene_hip.h"
#include "dev_prims.hpp"

using namespace pgd;

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
	fprintf(stderr, "[E::pga] %s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return PGA_ERR_NO_DEVICE; } } while (0)

// a switch that takes a comma-separated list of words (PANGENE_FILTERS=k32,global  PANGENE_LOOP=nopre,nofinal,noskip)
static bool env_has(const char *name, const char *word)
{
	const char *e = getenv(name);
	const size_t n = strlen(word);
	for (; e && *e; ) { const char *c = strchr(e, ','); const size_t len = c ? (size_t)(c - e) : strlen(e); if (len == n && strncmp(e, word, n) == 0) return true; e = c ? c + 1 : nullptr; }
	return false;
}
// PANGENE_XLOOP_CAP=pairs[,arcs] (tests: exchange buffers of the sharded queued rounds that are too small at first; 0 or absent = as learned)
static long long xloop_cap(int which)
{
	const char *e = getenv("PANGENE_XLOOP_CAP");
	if (!e) return 0;
	if (which == 0) return atoll(e);
	const char *c = strchr(e, ',');
	return c ? atoll(c + 1) : 0;
}
// debugging aid: PANGENE_POISON=1 fills every fresh device / pinned allocation with a pattern, so that a read of memory nobody
// wrote shows up the same way in every run (recycled memory otherwise holds whatever the previous context left there)
static bool poison_on() { static const bool f = getenv("PANGENE_POISON") != nullptr; return f; }

#define F_HEAD 0x80000000u   // static: first hit of its genome in X order (index-0 quirk, overlap.c:108)
#define F_MULTI 0x40000000u  // static: the hit has more than one exon (lets the sweep skip the exon records)
#define F_CSTIE 0x20000000u  // static: an X-order neighbour shares (contig, cs) -- member of a tie group of the cs sort (hazard H2b; set by k_pack_rec)
#define F_MEMBER 0x10000000u // the hit is in the live lists (pga_ctx::live_on): it was not filtered when they were built; travels with the hit through order overrides
#define F_PUBLIC 0x7ffu

#include "pga_host_context.hpp"

extern "C" int pga_is_device(void) { return 1; }

// pinned host memory: what the reader packs the genomes into, so that the upload is plain DMA
extern "C" int pga_host_alloc(size_t nbytes, void **ptr)
{
	*ptr = nullptr;
	return hipHostMalloc(ptr, nbytes ? nbytes : 1, hipHostMallocDefault) == hipSuccess ? 0 : PGA_ERR_NOMEM;
}
extern "C" void pga_host_free(void *ptr) { if (ptr) (void)hipHostFree(ptr); }

extern "C" int pga_set_device(int32_t device) { if (hipSetDevice(device) != hipSuccess) return PGA_ERR_NO_DEVICE; g_last_dev.store(device); return 0; }
extern "C" int pga_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }

extern "C" void pga_host_trim(size_t keep_bytes)
{
	if (keep_bytes == 0) { // "give everything back": the cached device blocks too
		std::lock_guard<std::mutex> lk(g_dev_mu);
		for (DevBlock &b : g_dev_cache) (void)hipFree(b.p);
		g_dev_cache.clear();
	}
	std::lock_guard<std::mutex> lk(g_pin_mu);
	size_t kept = 0, n_keep = 0;
	for (; n_keep < g_pin_cache.size() && kept + g_pin_cache[n_keep].cap <= keep_bytes; ++n_keep) kept += g_pin_cache[n_keep].cap;
	for (size_t i = n_keep; i < g_pin_cache.size(); ++i) (void)hipHostFree(g_pin_cache[i].p);
	g_pin_cache.resize(n_keep);
}

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

// The kernels live in one header per part of the path (k_*.hpp), the host side of the C ABI in one per stage (pga_host_*.hpp, split in
// round 5: this file had grown to 2 500 lines); everything is ONE translation unit, included in this order.
#include "k_common.hpp"
#include "k_ingest.hpp"
#include "k_sweep.hpp"
#include "k_segsort.hpp"
#include "k_segsort2.hpp"
#include "k_stage_b.hpp"
#include "k_vertex.hpp"
#include "k_arcs.hpp"
#include "k_branch.hpp"
#include "k_genes.hpp"
#include "k_order.hpp"
#include "pga_host_common.hpp"
#include "pga_host_stage_a.hpp"
#include "pga_host_stage_b.hpp"
#include "pga_host_arcs.hpp"
#include "pga_host_branch.hpp"
#include "pga_host_order.hpp"
#include "pga_host_io.hpp"

extern "C" const pga_backend_t *pga_backend(void)
{
	static const pga_backend_t b = {
		"hip-gfx950", pga_create, pga_destroy, pga_begin, pga_ingest, pga_post_partials, pga_post_apply, pga_shadow, pga_set_filter,
		pga_vtx_partials, pga_flag_vtx, pga_arc_round, pga_arc_merge, pga_arc_set_current, pga_rep_pos, pga_n_local, pga_branch_pairs, pga_branch_decide, pga_mark_hits, pga_override_order, pga_set_head, pga_fetch, pga_put, pga_copy, pga_scratch,
		pga_download, pga_hazards, pga_is_device, pga_strerror, pga_timing_reset, pga_timing_get, pga_sync, pga_fetch_later, pga_hazard_segs, pga_host_alloc, pga_host_free, pga_arc_round_local, pga_ctg_counts, pga_gene_matrix, pga_arc_table, pga_arc_round_finish, pga_branch_decide_filter, pga_branch_loop, pga_host_trim, pga_set_device, pga_device_count, pga_arc_round_x, pga_copy_gbps, pga_warm, pga_reserve, pga_stage_h2d, pga_stage_drop
	};
	return &b;
}
#include "pga_host_selftest.hpp"
