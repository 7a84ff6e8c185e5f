/* msh_fmt_g8 (msamtools_amd/csrc/host/msh_fmt.h) writes what snprintf("%.8g") writes: random bit patterns, log-uniform magnitudes
 * with random mantissas, decimal-looking values that sit next to rounding ties, powers of ten and their neighbours, the ends of
 * the fixed / scientific ranges, exactly representable ties, zeros, subnormals, infinities, NaN. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../msamtools_amd/csrc/host/msh_fmt.h"

static uint64_t s = 88172645463325252ull;
static uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static long bad = 0, n = 0;
static void check(double v) {
	char a[64], b[64];
	const int la = msh_fmt_g8(v, a);
	const int lb = snprintf(b, sizeof b, "%.8g", v);
	n++;
	if (la != lb || strcmp(a, b)) { if (bad++ < 10) fprintf(stderr, "%a: got '%s' want '%s'\n", v, a, b); }
}
int main(int argc, char **argv) {
	const long N = argc > 1 ? atol(argv[1]) : 4000000;
	long i;
	int e, k;
	msh_fmt_init();
	for (i = 0; i < N; i++) { uint64_t u = rnd(); double v; memcpy(&v, &u, 8); check(v); }                 /* any bit pattern */
	for (i = 0; i < N; i++) {                                                                                /* the magnitudes a profile holds */
		const double m = 1.0 + (double)(rnd() >> 11) / 9007199254740992.0;
		const int ex = (int)(rnd() % 60) - 45;
		check(ldexp(m, ex)); check(-ldexp(m, ex));
	}
	for (i = 0; i < N; i++) {                                                                                /* k / 10^j: nine digits ending in 5, and friends */
		const uint64_t digits = 100000000ull + rnd() % 900000000ull;
		const int j = (int)(rnd() % 30);
		double v = (double)(digits / 10 * 10 + 5);
		int q;
		for (q = 0; q < j; q++) v /= 10.0;
		check(v);
		v = (double)digits; for (q = 0; q < j; q++) v /= 10.0; check(v);
		v = (double)digits; for (q = 0; q < (j % 12); q++) v *= 10.0; check(v);
	}
	for (e = -320; e <= 308; e++) {                                                                          /* powers of ten and their neighbours */
		const double p = pow(10.0, e);
		check(p); check(nextafter(p, 0)); check(nextafter(p, 1e308)); check(p * 9.9999999); check(p * 9.99999995); check(p * 1.00000005);
		for (k = 1; k < 10; k++) check(p * k);
	}
	check(0.0); check(-0.0); check(1.0 / 0.0); check(-1.0 / 0.0); check(0.0 / 0.0); check(4.9e-324); check(2.2250738585072014e-308);
	check(12345678.5); check(12345679.5); check(0.5); check(0.25); check(99999999.5); check(99999998.5); check(9999999.95); check(1e8); check(99999999.0);
	check(0.0001); check(0.00009999999949); check(0.000099999999951); check(123456789.0); check(1.7976931348623157e308);
	printf("values=%ld bad=%ld\n", n, bad);
	return bad != 0;
}
