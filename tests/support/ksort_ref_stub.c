/* Test support (build container only): instantiates the reference's own radix sort from the header where it lies
 * (-I/root/reference; nothing is copied) so that tests/test_oracle_vs_ref.py can compare ksort_exact.hpp with it directly. */
#include <stdint.h>
#include <stdlib.h>
#include "ksort.h"
typedef struct { uint64_t x, y; } t128_t;
#define t128_key(a) ((a).x)
KRADIX_SORT_INIT(t128, t128_t, t128_key, 8)
void ref_radix_sort_128(t128_t *a, size_t n) { radix_sort_t128(a, a + n); }
