// shm_exchange.cpp -- the exchange of a sharded run between the processes ONE `pangene --gpus N` command forks, for
// backends whose vectors live in host memory (the oracle host of the tests; the HIP backend talks RCCL instead:
// hip/rccl_exchange.cpp).  The launcher maps one anonymous shared region before it forks; every rank then owns a slot in it.
// An all-reduce / all-gather is: copy my part into my slot, barrier, read every slot in rank order, barrier -- in pieces when
// a vector is longer than a slot.  Integer sums and maxima only: the result does not depend on the order.
#include <pthread.h>
#include <sys/mman.h>
#include <cstring>
#include <cstdio>
#include <algorithm>
#include "pangene_amd.h"

namespace {

struct ShmHeader { pthread_barrier_t bar; int32_t world; int64_t slot_bytes; };
ShmHeader *g_hdr = nullptr;
int g_rank = 0;

inline char *slot(int r) { return (char *)(g_hdr + 1) + (size_t)r * (size_t)g_hdr->slot_bytes; }

int x_allreduce(void *, void *buf, int64_t count, int32_t dtype, int32_t op, int32_t is_device)
{
	if (is_device || !g_hdr) { std::fprintf(stderr, "[E::pg_shm] all-reduce on a device buffer: this exchange serves host-memory backends only\n"); return -1; }
	const int64_t esz = dtype == PG_X_I64 ? 8 : 4, per = g_hdr->slot_bytes / esz;
	for (int64_t off = 0; off < count; off += per) {
		const int64_t n = std::min(per, count - off);
		std::memcpy(slot(g_rank), (char *)buf + off * esz, (size_t)(n * esz));
		pthread_barrier_wait(&g_hdr->bar);
		for (int64_t i = 0; i < n; ++i) {
			if (esz == 8) {
				int64_t a = 0;
				for (int r = 0; r < g_hdr->world; ++r) { const int64_t v = ((const int64_t *)slot(r))[i]; a = r == 0 ? v : op == PG_X_MAX ? std::max(a, v) : a + v; }
				((int64_t *)buf)[off + i] = a;
			} else {
				int32_t a = 0;
				for (int r = 0; r < g_hdr->world; ++r) { const int32_t v = ((const int32_t *)slot(r))[i]; a = r == 0 ? v : op == PG_X_MAX ? std::max(a, v) : a + v; }
				((int32_t *)buf)[off + i] = a;
			}
		}
		pthread_barrier_wait(&g_hdr->bar);
	}
	return 0;
}

int x_allgather(void *, const void *in, void *out, int64_t nbytes, int32_t is_device)
{
	if (is_device || !g_hdr) { std::fprintf(stderr, "[E::pg_shm] all-gather on a device buffer: this exchange serves host-memory backends only\n"); return -1; }
	for (int64_t off = 0; off < nbytes; off += g_hdr->slot_bytes) {
		const int64_t n = std::min<int64_t>(g_hdr->slot_bytes, nbytes - off);
		std::memcpy(slot(g_rank), (const char *)in + off, (size_t)n);
		pthread_barrier_wait(&g_hdr->bar);
		for (int r = 0; r < g_hdr->world; ++r) std::memcpy((char *)out + (size_t)r * (size_t)nbytes + off, slot(r), (size_t)n);
		pthread_barrier_wait(&g_hdr->bar);
	}
	return 0;
}

} // namespace

extern "C" {

// called by the launcher BEFORE it forks: the region every rank will see
void *pg_shm_create(int32_t world, int64_t slot_bytes)
{
	const size_t tot = sizeof(ShmHeader) + (size_t)world * (size_t)slot_bytes;
	void *p = mmap(nullptr, tot, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
	if (p == MAP_FAILED) return nullptr;
	ShmHeader *h = (ShmHeader *)p;
	pthread_barrierattr_t at;
	pthread_barrierattr_init(&at);
	pthread_barrierattr_setpshared(&at, PTHREAD_PROCESS_SHARED);
	pthread_barrier_init(&h->bar, &at, (unsigned)world);
	pthread_barrierattr_destroy(&at);
	h->world = world, h->slot_bytes = slot_bytes;
	return p;
}

// called by every rank after the fork: installs the exchange (pg_set_exchange)
int pg_shm_init(void *region, int32_t rank)
{
	if (region == nullptr) return -1;
	g_hdr = (ShmHeader *)region, g_rank = rank;
	pg_exchange_t x;
	x.rank = rank, x.world = g_hdr->world, x.user = nullptr, x.allreduce = x_allreduce, x.allgather = x_allgather, x.stream_ordered = 0;
	pg_set_exchange(&x);
	return 0;
}

} // extern "C"
