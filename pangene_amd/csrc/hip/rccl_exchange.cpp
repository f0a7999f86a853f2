// Native exchange for sharded runs: RCCL collectives enqueued on the backend's own HIP stream.
//
// One process per GPU; each holds a shard of genomes.  The host driver asks (pg_exchange_t, include/pangene_amd.h) for a
// handful of small integer all-reduces / all-gathers per round on buffers that live in HBM.  Here they
// are ncclAllReduce / ncclAllGather on the stream the kernels run on: the collective is ordered after the kernels
// that produced its input and before the ones that consume its output, with no host synchronisation and no trip
// through Python -- over xGMI a small all-reduce is then a ~10-20 us stream operation instead of a ~100 us
// host round trip.
//
// RCCL is bound at run time (dlopen by soname): in a PyTorch process this resolves to the librccl.so.1 torch has
// already loaded, so the process has exactly one RCCL; a plain C host gets /opt/rocm/lib/librccl.so.1.  The library
// itself therefore has no link-time dependency on RCCL and single-GPU users never load it.
//
// Bootstrap: rank 0 calls pg_rccl_unique_id(), the launcher broadcasts the 128 bytes by whatever means it has
// (bench.py: torch.distributed), every rank calls pg_rccl_init(rank, world, id).
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include "pangene_amd.h"
#include "pangene_hip.h"

namespace {

struct Api {
	void *h = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
} g_api;

ncclComm_t g_comm = nullptr;
int g_rank = 0, g_world = 1;
char g_err[256] = "";

bool load_api()
{
	if (g_api.h) return true;
	const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
	void *h = nullptr;
	for (const char *n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
	if (!h) { std::snprintf(g_err, sizeof(g_err), "cannot load RCCL: %s", dlerror()); return false; }
#define BIND(field, sym) do { *(void **)(&g_api.field) = dlsym(h, sym); if (!g_api.field) { std::snprintf(g_err, sizeof(g_err), "RCCL lacks %s", sym); return false; } } while (0)
	BIND(GetUniqueId, "ncclGetUniqueId"); BIND(CommInitRank, "ncclCommInitRank"); BIND(CommDestroy, "ncclCommDestroy");
	BIND(AllReduce, "ncclAllReduce"); BIND(AllGather, "ncclAllGather"); BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
	g_api.h = h;
	return true;
}

int fail(ncclResult_t r, const char *what)
{
	std::snprintf(g_err, sizeof(g_err), "%s: %s", what, g_api.GetErrorString ? g_api.GetErrorString(r) : "?");
	std::fprintf(stderr, "[E::pg_rccl] %s\n", g_err);
	return -1;
}

// pg_exchange_t callbacks.  Device buffers go through RCCL on the backend's stream; host buffers do not occur with the
// HIP backend (its vectors live in HBM), so they are refused rather than silently left un-reduced.
int x_allreduce(void *, void *buf, int64_t count, int32_t dtype, int32_t op, int32_t is_device)
{
	if (!is_device || !g_comm) { std::snprintf(g_err, sizeof(g_err), "allreduce on a host buffer / without a communicator"); return -1; }
	hipStream_t st = (hipStream_t)pga_active_stream();
	const ncclResult_t r = g_api.AllReduce(buf, buf, (size_t)count, dtype == PG_X_I64 ? ncclInt64 : ncclInt32, op == PG_X_MAX ? ncclMax : ncclSum, g_comm, st);
	return r == ncclSuccess ? 0 : fail(r, "ncclAllReduce");
}

int x_allgather(void *, const void *in, void *out, int64_t nbytes, int32_t is_device)
{
	if (!is_device || !g_comm) { std::snprintf(g_err, sizeof(g_err), "allgather on a host buffer / without a communicator"); return -1; }
	hipStream_t st = (hipStream_t)pga_active_stream();
	const ncclResult_t r = g_api.AllGather(in, out, (size_t)nbytes, ncclInt8, g_comm, st);
	return r == ncclSuccess ? 0 : fail(r, "ncclAllGather");
}

} // namespace

extern "C" {

int pg_rccl_unique_id(void *out128)
{
	static_assert(sizeof(ncclUniqueId) == 128, "the bootstrap id is passed around as 128 opaque bytes");
	if (!load_api()) return -1;
	ncclUniqueId id;
	const ncclResult_t r = g_api.GetUniqueId(&id);
	if (r != ncclSuccess) return fail(r, "ncclGetUniqueId");
	std::memcpy(out128, &id, sizeof(id));
	return 0;
}

int pg_rccl_init(int32_t rank, int32_t world, const void *id128)
{
	if (!load_api()) return -1;
	if (g_comm) { g_api.CommDestroy(g_comm); g_comm = nullptr; }
	ncclUniqueId id;
	std::memcpy(&id, id128, sizeof(id));
	const ncclResult_t r = g_api.CommInitRank(&g_comm, world, id, rank); // on the calling thread's current HIP device
	if (r != ncclSuccess) { g_comm = nullptr; return fail(r, "ncclCommInitRank"); }
	g_rank = rank, g_world = world;
	pg_exchange_t x;
	x.rank = rank, x.world = world, x.user = nullptr, x.allreduce = x_allreduce, x.allgather = x_allgather, x.stream_ordered = 1;
	pg_set_exchange(&x);
	return 0;
}

int pg_rccl_finalize(void)
{
	pg_set_exchange(nullptr);
	if (g_comm && g_api.CommDestroy) g_api.CommDestroy(g_comm);
	g_comm = nullptr;
	return 0;
}

const char *pg_rccl_error(void) { return g_err; }

} // extern "C"
