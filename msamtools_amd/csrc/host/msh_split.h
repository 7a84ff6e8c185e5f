/* The byte range [lo, hi) of n bytes that thread tid of nth takes when the ranges are to begin at multiples of `al`
 * (the parallel pwrite of msh_write_framed: whole megabytes per thread).  The ranges of tid = 0..nth-1 are disjoint,
 * in order and cover [0, n) exactly: the share is ceil(n / nth) rounded UP to `al`, so nth shares never fall short of
 * n (a floor division here once left the last n % nth bytes unwritten when n / nth was a multiple of `al`); threads
 * behind the end get an empty range.  tests/c/split_test.c. */
#ifndef MSH_SPLIT_H
#define MSH_SPLIT_H
#include <stddef.h>
static inline void msh_split_aligned(size_t n, int nth, int tid, size_t al, size_t *lo, size_t *hi) {
	const size_t t = nth > 0 ? (size_t)nth : 1;
	const size_t per = ((n + t - 1) / t + al - 1) / al * al;
	size_t a = per * (size_t)tid, b;
	if (a > n) a = n;
	b = n - a > per ? a + per : n;
	*lo = a;
	*hi = b;
}
#endif
