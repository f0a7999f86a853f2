// How fast can T threads of ONE process get fresh memory on this host?  (The batch PAF reader's wall time does not move with the
// thread count on the 256-thread GPU box: is first-touch page faulting the serial resource?)
//   g++ -O2 -std=c++17 -pthread pagefault_bench.cpp -o pagefault_bench && ./pagefault_bench <MB per thread> <threads...>
// Modes: 4K = malloc'd / mmap'd memory touched page by page; THP = 2 MiB-aligned mmap + MADV_HUGEPAGE; POP = + MADV_POPULATE_WRITE;
// REUSE = the same buffer touched a second time (no faults: the memory-bandwidth ceiling of the loop itself).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
#include <thread>
#include <vector>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void touch(char *p, size_t n) { for (size_t i = 0; i < n; i += 4096) p[i] = 1; }
int main(int argc, char **argv)
{
	const size_t mb = argc > 1 ? std::strtoul(argv[1], nullptr, 10) : 32, bytes = mb << 20;
	for (int a = 2; a < argc; ++a) {
		const int T = std::atoi(argv[a]);
		for (int mode = 0; mode < 4; ++mode) {
			std::atomic<int> ready{0}; std::atomic<bool> go{false};
			std::vector<double> dt((size_t)T, 0.0);
			std::vector<std::thread> th;
			for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
				ready.fetch_add(1); while (!go.load()) { }
				const double t0 = now();
				char *p = (char *)mmap(nullptr, bytes + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
				char *q = (char *)(((uintptr_t)p + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1));
				if (mode >= 1 && mode <= 2) madvise(q, bytes, MADV_HUGEPAGE);
				if (mode == 2) madvise(q, bytes, MADV_POPULATE_WRITE);
				touch(q, bytes);
				if (mode == 3) { const double t1 = now(); touch(q, bytes); dt[(size_t)t] = now() - t1; }
				else dt[(size_t)t] = now() - t0;
				munmap(p, bytes + (2u << 20));
			});
			while (ready.load() < T) { }
			const double w0 = now(); go.store(true);
			for (auto &x : th) x.join();
			const double wall = now() - w0;
			double mx = 0; for (double x : dt) mx = x > mx ? x : mx;
			static const char *nm[4] = { "4K", "THP", "THP+POP", "REUSE" };
			std::printf("threads %3d  %-8s %6zu MB/thread: slowest thread %.3f s, wall %.3f s => %.2f GB/s of fresh memory (%.2f M 4K-pages/s)\n", T, nm[mode], mb, mx, wall,
			            (double)bytes * T / mx * 1e-9, (double)bytes * T / mx / 4096 * 1e-6);
		}
	}
	return 0;
}
