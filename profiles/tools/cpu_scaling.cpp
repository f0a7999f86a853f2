// How many cores does this host really give one process?  T threads each do the same fixed amount of register-only arithmetic;
// if the per-thread time grows with T, the threads share fewer cores (a CPU quota, SMT siblings, a frequency cap) than T.
//   g++ -O2 -std=c++17 -pthread cpu_scaling.cpp -o cpu_scaling && ./cpu_scaling 1 16 32 64 128 256
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
int main(int argc, char **argv)
{
	double alone = 0;
	for (int a = 1; a < argc; ++a) {
		const int T = std::atoi(argv[a]);
		std::vector<std::thread> th; std::vector<double> dt((size_t)T);
		const auto t0 = std::chrono::steady_clock::now();
		for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
			const auto s = std::chrono::steady_clock::now();
			uint64_t y = 88172645463325252ull + (uint64_t)t;
			for (long i = 0; i < 150000000L; ++i) { y ^= y << 13; y ^= y >> 7; y ^= y << 17; }
			volatile uint64_t sink = y; (void)sink;
			dt[(size_t)t] = std::chrono::duration<double>(std::chrono::steady_clock::now() - s).count();
		});
		for (auto &x : th) x.join();
		const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		double mx = 0, sum = 0;
		for (double d : dt) { mx = d > mx ? d : mx; sum += d; }
		if (a == 1) alone = sum / T;
		std::printf("threads %3d: wall %.3f s, per thread mean %.3f s, slowest %.3f s => throughput of %.1f threads running alone\n", T, wall, sum / T, mx, alone * T / wall);
	}
	return 0;
}
