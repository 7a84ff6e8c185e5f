/* msh_fmt.h -- "%.8g" without printf.
 *
 * The profile's text is a line "%s\t%.8g\n" per feature (mMatrix.c:359-376); with a million features the million
 * vsnprintf calls were a third of the report's time.  msh_fmt_g8 writes what snprintf(out, n, "%.8g", v) writes -- the same
 * bytes, checked against it over tens of millions of values in tests/c/fmt_g8_test.c -- by scaling to eight integer digits
 * in 80-bit arithmetic; a value whose ninth digit onward lies within 1e-6 of a rounding tie (the scaling's error is below
 * 1e-8 there), and anything that is not a finite normal number, is handed to snprintf itself. */
#ifndef MSH_FMT_H
#define MSH_FMT_H
#include <math.h>
#include <stdio.h>
#include <string.h>

#define MSH_P10_MIN (-340)
#define MSH_P10_MAX 320
static long double msh_p10_tab[MSH_P10_MAX - MSH_P10_MIN + 1];
static int msh_p10_ready;
static inline void msh_fmt_init(void) {           /* (idempotent; call once before threads format) */
	int i;
	if (msh_p10_ready) return;
	for (i = MSH_P10_MIN; i <= MSH_P10_MAX; i++) msh_p10_tab[i - MSH_P10_MIN] = powl(10.0L, (long double)i);
	msh_p10_ready = 1;
}
static inline long double msh_p10(int e) { return msh_p10_tab[e - MSH_P10_MIN]; }

/* out: at least 32 bytes; returns the number of characters written (no terminator counted, one is written) */
static inline int msh_fmt_g8(double v, char *out) {
	char dig[9];
	int e2, e10, nd, n = 0, i;
	long double d, fl, fr;
	unsigned long long D;
	double a = v;
	if (!msh_p10_ready || !(v == v) || v - v != 0.0) return snprintf(out, 32, "%.8g", v);      /* NaN, infinities */
	if (v == 0.0) { if (signbit(v)) { out[0] = '-'; out[1] = '0'; out[2] = 0; return 2; } out[0] = '0'; out[1] = 0; return 1; }
	if (a < 0) a = -a;
	if (a < 2.3e-308) return snprintf(out, 32, "%.8g", v);                                      /* subnormals */
	(void)frexp(a, &e2);                                  /* a in [2^(e2-1), 2^e2) */
	e10 = (int)floor((double)(e2 - 1) * 0.30102999566398120);
	if (e10 < MSH_P10_MIN + 10 || e10 > MSH_P10_MAX - 10) return snprintf(out, 32, "%.8g", v);
	if ((long double)a >= msh_p10(e10 + 1)) e10++;
	else if ((long double)a < msh_p10(e10)) e10--;
	d = 7 - e10 >= 0 ? (long double)a * msh_p10(7 - e10) : (long double)a / msh_p10(e10 - 7);   /* in [1e7, 1e8) up to the tables' last bit */
	fl = floorl(d);
	fr = d - fl;
	if (fr > 0.5L - 1e-6L && fr < 0.5L + 1e-6L) return snprintf(out, 32, "%.8g", v);            /* next to a tie: let printf decide */
	D = (unsigned long long)fl + (fr > 0.5L ? 1ull : 0ull);
	if (D < 10000000ull || D > 100000000ull) return snprintf(out, 32, "%.8g", v);                /* (the exponent estimate was off: not expected) */
	if (D == 100000000ull) { D = 10000000ull; e10++; }
	for (i = 7; i >= 0; i--) { dig[i] = (char)('0' + D % 10ull); D /= 10ull; }
	nd = 8;
	while (nd > 1 && dig[nd - 1] == '0') nd--;
	if (v < 0) out[n++] = '-';
	if (e10 < -4 || e10 >= 8) {                           /* d[.ddd]e[+-]XX */
		int x = e10 < 0 ? -e10 : e10;
		out[n++] = dig[0];
		if (nd > 1) { out[n++] = '.'; memcpy(out + n, dig + 1, (size_t)(nd - 1)); n += nd - 1; }
		out[n++] = 'e';
		out[n++] = e10 < 0 ? '-' : '+';
		if (x >= 100) { out[n++] = (char)('0' + x / 100); x %= 100; out[n++] = (char)('0' + x / 10); out[n++] = (char)('0' + x % 10); }
		else { out[n++] = (char)('0' + x / 10); out[n++] = (char)('0' + x % 10); }
	} else if (e10 >= 0) {                                /* ddd[.ddd]: e10 + 1 digits in front of the point */
		const int ip = e10 + 1;
		for (i = 0; i < ip; i++) out[n++] = i < nd ? dig[i] : '0';
		if (nd > ip) { out[n++] = '.'; memcpy(out + n, dig + ip, (size_t)(nd - ip)); n += nd - ip; }
	} else {                                              /* 0.000ddd */
		out[n++] = '0'; out[n++] = '.';
		for (i = 0; i < -e10 - 1; i++) out[n++] = '0';
		memcpy(out + n, dig, (size_t)nd); n += nd;
	}
	out[n] = 0;
	return n;
}
#endif
