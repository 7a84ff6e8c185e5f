/* huff_lengths_test.c -- df_huff_lengths (msx_deflate_model.h, the host restatement of the device encoder's tree builder)
 * under frequency vectors that push the tree past its length limit: every result must be a COMPLETE prefix code (Kraft
 * sum exactly 1) with no length above the limit and a code for every used symbol.  Round 4 shipped, for an hour, a repair
 * step that counted the leaves moved instead of the Kraft excess and wrote over-subscribed 7-bit code-length codes for
 * 42 of 80 810 blocks of one file -- bench.py's digest check caught it; this test pins the repair.
 * gcc -O2 -o huff_lengths_test huff_lengths_test.c && ./huff_lengths_test */
#include <stdio.h>
#include <stdlib.h>
#include "../../msamtools_amd/csrc/msx_deflate_model.h"

static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 11); }

static int check(const uint32_t *f, int n, int maxbits, const char *what, int trial) {
	uint8_t len[DF_NLL];
	int i, used = 0, coded = 0;
	unsigned long long kraft = 0;
	df_huff_lengths(f, n, maxbits, len);
	for (i = 0; i < n; i++) {
		if (f[i]) used++;
		if (len[i]) { coded++; kraft += 1ull << (maxbits - len[i]); }
		if (len[i] > maxbits) { printf("%s trial %d: length %d above %d\n", what, trial, len[i], maxbits); return 1; }
		if (f[i] && !len[i]) { printf("%s trial %d: used symbol %d without a code\n", what, trial, i); return 1; }
	}
	if (coded < 2 || coded < used) { printf("%s trial %d: %d codes for %d used symbols\n", what, trial, coded, used); return 1; }
	if (kraft != 1ull << maxbits) { printf("%s trial %d: Kraft sum %llu / %llu\n", what, trial, kraft, 1ull << maxbits); return 1; }
	return 0;
}

int main(void) {
	static const struct { int n, maxbits; const char *what; } shapes[3] = {{DF_NCL, 7, "code-length code"}, {DF_ND, 15, "distance code"}, {DF_NLL, 15, "literal/length code"}};
	int s, t, i, bad = 0;
	for (s = 0; s < 3; s++) {
		const int n = shapes[s].n;
		for (t = 0; t < 4000 && !bad; t++) {
			uint32_t f[DF_NLL] = {0};
			const int kind = t % 5, m = 1 + (int)(rnd() % (uint32_t)n);
			uint32_t a = 1, b = 1;
			for (i = 0; i < m; i++) {
				const int sym = (int)(rnd() % (uint32_t)n);
				uint32_t v;
				switch (kind) {
				case 0: v = 1 + rnd() % 1000; break;                          /* flat */
				case 1: v = a; { uint32_t c = a + b; a = b; b = c; if (b > 40000) { a = 1; b = 1; } } break;   /* Fibonacci: the deepest trees */
				case 2: v = 1u << (rnd() % 16); break;                         /* powers of two */
				case 3: v = (rnd() % 8 == 0) ? 60000 : 1; break;               /* one giant, many ones */
				default: v = 1 + (rnd() % 3); break;                           /* nearly equal */
				}
				f[sym] = v;
			}
			bad |= check(f, n, shapes[s].maxbits, shapes[s].what, t);
		}
	}
	if (!bad) printf("ok\n");
	return bad;
}
