// Test support: sorts the same (key, index) arrays with the reference's radix sort (ksort_ref_stub.c) and with
// pangene_amd/csrc/host/ksort_exact.hpp and requires the same permutation -- the order-exact restatement claim of SURVEY 9.1.
#include <cstdint>
#include <cstdio>
#include <vector>
#include "ksort_exact.hpp"
struct t128_t { uint64_t x, y; };
extern "C" void ref_radix_sort_128(t128_t *a, size_t n);
static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
int main()
{
	long n_case = 0, n_bad = 0;
	const size_t sizes[] = { 1, 2, 63, 64, 65, 100, 257, 1000, 4097, 20000, 100003 };
	for (size_t n : sizes)
		for (int bits = 1; bits <= 40; bits += 3)        // few distinct keys ... many
			for (int rep = 0; rep < (n < 5000 ? 8 : 2); ++rep) {
				std::vector<t128_t> a(n), b;
				const uint64_t mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1);
				const int shift = (int)(rnd() % 24);          // keys that only differ in higher bytes too
				for (size_t i = 0; i < n; ++i) a[i].x = (rnd() & mask) << shift, a[i].y = i;
				if (rep & 1) for (size_t i = 0; i < n; ++i) a[i].x = a[i].x / 300 * 300; // the 300-bp grid of the fuzz sets: long tie groups
				b = a;
				ref_radix_sort_128(a.data(), n);
				pgx::ksort_exact(b.data(), n, [](const t128_t &e) { return e.x; });
				++n_case;
				for (size_t i = 0; i < n; ++i)
					if (a[i].x != b[i].x || a[i].y != b[i].y) { ++n_bad; std::printf("MISMATCH n=%zu bits=%d rep=%d at %zu\n", n, bits, rep, i); break; }
			}
	std::printf("ksort check: %ld cases, %ld mismatches\n", n_case, n_bad);
	return n_bad != 0;
}
