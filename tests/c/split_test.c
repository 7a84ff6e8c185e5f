/* msh_split_aligned (msamtools_amd/csrc/host/msh_split.h): the ranges of the parallel pwrite cover [0, n) exactly,
 * in order, without overlap, every range but the first beginning at a multiple of the alignment -- over the sizes
 * n = k * nth * al + r that a floor division got wrong. */
#include <stdio.h>
#include <stdlib.h>
#include "../../msamtools_amd/csrc/host/msh_split.h"

static long bad = 0, cases = 0;
static void check(size_t n, int nth, size_t al) {
	size_t at = 0;
	cases++;
	for (int t = 0; t < nth; t++) {
		size_t lo, hi;
		msh_split_aligned(n, nth, t, al, &lo, &hi);
		if (lo != at || hi < lo || hi > n || (lo < n && lo % al)) { bad++; return; }
		at = hi;
	}
	if (at != n) bad++;
}
int main(void) {
	const size_t al = (size_t)1 << 20;
	for (int nth = 1; nth <= 17; nth++)
		for (size_t k = 0; k <= 6; k++)
			for (long r = -3; r <= 40; r++) {
				const size_t base = k * (size_t)nth * al;
				if (r < 0 && base < (size_t)-r) continue;
				check(base + (size_t)r, nth, al);
				check(base + (size_t)r * 65311u, nth, al);
			}
	check(4 * 2 * al + 3, 4, al);          /* the advisor's example: n / nth an exact multiple, n % nth != 0 */
	for (int i = 0; i < 200000; i++) check((size_t)rand() * 977u % ((size_t)3 << 30), 1 + rand() % 64, (size_t)1 << (rand() % 24));
	printf("cases=%ld bad=%ld\n", cases, bad);
	return bad != 0;
}
