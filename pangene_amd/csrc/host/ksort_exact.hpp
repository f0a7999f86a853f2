// ksort_exact.hpp -- order-exact restatement of the reference's 64-bit-key sort (ksort.h:34-87):
// an in-place MSD ("American flag") radix sort on 8-bit digits starting at the top byte, which hands
// any bucket of <= 64 elements to a stable insertion sort.  The sort is UNSTABLE for bigger inputs and
// the order it leaves among equal keys reaches the GFA through two host-side sorts (proteins by
// sum-score, hit.c:209; genes by preferred|n_dom|avg_score, vertex.c:59), so those two call sites must
// reproduce it exactly.  Same element moves as the reference, own formulation.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <utility>

namespace pgx {

template <class T, class KeyFn>
static void ks_insertion(T *a, size_t n, KeyFn key)
{
	for (size_t i = 1; i < n; ++i) {
		if (key(a[i]) < key(a[i - 1])) {
			T t = a[i];
			size_t j = i;
			while (j > 0 && key(t) < key(a[j - 1])) { a[j] = a[j - 1]; --j; }
			a[j] = t;
		}
	}
}

// A bucket that holds most of its parent (the proteins without a representative hit all carry key 0: 82 % of 106 653 on an isoform-rich set) is looked at
// first: a pass in which every key has the same digit moves nothing, so the recursion may start at the highest byte that still varies -- or stop when
// none does (the reference walks every level over such a bucket and moves nothing either).  Returns the shift to go on with, or -1.
template <class T, class KeyFn>
static int ks_skip_levels(const T *a, size_t n, int shift, KeyFn key)
{
	const uint64_t below = shift >= 56 ? ~0ull : ((1ull << (shift + 8)) - 1);
	uint64_t diff = 0;
	const uint64_t k0 = key(a[0]);
	for (size_t i = 1; i < n; ++i) diff |= key(a[i]) ^ k0;
	diff &= below;
	if (diff == 0) return -1;
	int top = 63;
	while (!(diff >> top & 1)) --top;
	return top / 8 * 8;
}

template <class T, class KeyFn>
static void ks_flag_pass(T *a, size_t n, int shift, KeyFn key)
{
	// The reference walks all 256 buckets; empty ones are skipped there without effect, so only the occupied
	// digits (kept in a 256-bit set) are visited here -- same element moves, far fewer iterations for the small
	// buckets that dominate the recursion.
	uint32_t cnt[256];
	size_t head[256], tail[256];
	uint64_t occ[4] = {0, 0, 0, 0};
	std::memset(cnt, 0, sizeof(cnt));
	for (size_t i = 0; i < n; ++i) {
		unsigned d = (unsigned)((key(a[i]) >> shift) & 0xff);
		++cnt[d];
		occ[d >> 6] |= 1ULL << (d & 63);
	}
	unsigned digits[256];
	int nd = 0;
	for (int w = 0; w < 4; ++w)
		for (uint64_t m = occ[w]; m; m &= m - 1) digits[nd++] = (unsigned)(w * 64 + __builtin_ctzll(m));
	size_t acc = 0;
	for (int t = 0; t < nd; ++t) { unsigned b = digits[t]; head[b] = acc; acc += cnt[b]; tail[b] = acc; }
	// cycle-leader permutation: the element carried out of bucket k is dropped at the write cursor of its own
	// bucket, whatever sits there is carried on, until something that belongs to k comes back
	for (int t = 0; t < nd;) {
		const unsigned k = digits[t];
		if (head[k] == tail[k]) { ++t; continue; }
		unsigned l = (unsigned)((key(a[head[k]]) >> shift) & 0xff);
		if (l == k) { ++head[k]; continue; }
		T carry = a[head[k]];
		do {
			std::swap(carry, a[head[l]]);
			++head[l];
			l = (unsigned)((key(carry) >> shift) & 0xff);
		} while (l != k);
		a[head[k]++] = carry;
	}
	if (shift == 0) return;
	int next = shift > 8 ? shift - 8 : 0;
	size_t st = 0;
	for (int t = 0; t < nd; ++t) {
		size_t m = cnt[digits[t]];
		if (m > 64) { const int nx = m * 2 > n ? ks_skip_levels(a + st, m, next, key) : next; if (nx >= 0) ks_flag_pass(a + st, m, nx, key); }
		else if (m > 1) ks_insertion(a + st, m, key);
		st += m;
	}
}

// sorts [a, a+n) ascending by the uint64_t key(a[i]) with exactly the reference's tie order
template <class T, class KeyFn>
static void ksort_exact(T *a, size_t n, KeyFn key)
{
	if (n <= 64) { ks_insertion(a, n, key); return; }
	// A pass in which every key has the same digit moves nothing (each element is already "home") and then
	// recurses on the whole array with the next digit: start directly at the highest byte that varies.
	uint64_t diff = 0, k0 = key(a[0]);
	for (size_t i = 1; i < n; ++i) diff |= key(a[i]) ^ k0;
	if (diff == 0) return;
	int top = 63;
	while (!(diff >> top & 1)) --top;
	ks_flag_pass(a, n, top / 8 * 8, key);
}

} // namespace pgx
