/* Host harness for the flat MD walk of k_aln_stats_flat (msx_md.h, "Flat walk"):
 * one simulated wave of 64 lanes takes up to 128 consecutive MD strings stored back
 * to back, 16 bytes per lane and 1 KiB per pass, resolves the carries between lanes
 * with two "ballots" and one 64-bit addition, keeps per-word counts + prefix sums,
 * and reads every string's edit count as a difference of two prefix values -- the
 * steps of the kernel, in the same order, with plain loops standing in for lanes.
 * The result must equal the byte-at-a-time rule (md_byte, mBamVector.c:112-118) for
 * arbitrary bytes, empty strings, strings longer than a pass and every alignment of
 * the buffer.  Built and run by tests/test_md_flat_cpu.py. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../msamtools_amd/csrc/msx_md.h"

#define LANES 64
#define PASS_BYTES 1024u
#define PASS_WORDS 256u

static uint32_t F_at(const uint32_t *c80, const uint32_t *pre, uint32_t pb) {
	const uint32_t w = pb >> 2, k = pb & 3u;
	return pre[w] + (uint32_t)__builtin_popcount(c80[w] & ((1u << (8u * k)) - 1u));
}

/* edit[i] for strings i in [0, nrec) with byte offsets off[0..nrec] into md (address alignment d = addr & 15) */
static void wave_flat(const unsigned char *arena, uint32_t d, const uint32_t *off, int nrec, uint32_t *edit) {
	const unsigned char *md = arena + d;                 /* arena is 16-byte aligned: (md address) & 15 == d */
	const uint32_t m_begin = off[0], m_end = off[nrec];
	const uint32_t B = (m_begin + d) & ~15u;             /* position 0 of pass 0, in (offset + d) space */
	const uint32_t span = m_end + d - B;
	const uint32_t npass = span ? (span + PASS_BYTES - 1) / PASS_BYTES : 0;
	uint32_t cin_pass = 0, p;
	int i;
	for (i = 0; i < nrec; i++) edit[i] = 0;
	for (p = 0; p < npass; p++) {
		uint32_t start[PASS_WORDS + 1], c80[PASS_WORDS + 1], pre[PASS_WORDS + 1];
		uint32_t nd[LANES][4], M[LANES][4], S[LANES][4], u[LANES][4];
		unsigned long long G = 0, P = 0, a, b, s, cinv;
		uint32_t run = 0;
		int l, q;
		memset(start, 0, sizeof start);
		for (i = 0; i < nrec; i++) {                      /* ds_or: every string marks its first byte */
			const uint32_t pos = off[i] + d - B;
			if (pos >= p * PASS_BYTES && pos < (p + 1) * PASS_BYTES) {
				const uint32_t pb = pos - p * PASS_BYTES;
				start[pb >> 2] |= 1u << (8u * (pb & 3u));
			}
		}
		for (l = 0; l < LANES; l++) {
			uint32_t x[4] = {0, 0, 0, 0};
			const uint32_t cpos = p * PASS_BYTES + 16u * (uint32_t)l;      /* chunk start in pass space */
			if (cpos < span) memcpy(x, arena + B + cpos, 16);              /* aligned 16-byte block holding >= 1 valid byte */
			md_chunk_prepare(x, &start[4 * l], nd[l], M[l], S[l]);
			{
				uint32_t t[4];
				const uint32_t g = md_chunk_chain(M[l], S[l], 0u, t);
				const int prop = (t[0] & t[1] & t[2] & t[3]) == 0xffffffffu;
				if (g) G |= 1ull << l;
				if (prop) P |= 1ull << l;
			}
		}
		a = G | P; b = G;
		s = a + b + cin_pass;
		{
			const int co = (s < a) || (cin_pass && s == a);                /* carry out of the 64-bit add */
			cinv = s ^ P;                                                   /* carry INTO every lane */
			cin_pass = co ? 1u : 0u;
		}
		for (l = 0; l < LANES; l++) {
			md_chunk_chain(M[l], S[l], (uint32_t)((cinv >> l) & 1ull), u[l]);
			for (q = 0; q < 4; q++) {
				const uint32_t c = u[l][q] & nd[l][q];
				c80[4 * l + q] = c;
				pre[4 * l + q] = run;
				run += (uint32_t)__builtin_popcount(c);
			}
		}
		c80[PASS_WORDS] = 0;
		pre[PASS_WORDS] = run;
		for (i = 0; i < nrec; i++) {
			const uint32_t lo = p * PASS_BYTES, hi = lo + PASS_BYTES;
			uint32_t ps = off[i] + d - B, pe = off[i + 1] + d - B;
			ps = ps < lo ? lo : (ps > hi ? hi : ps);
			pe = pe < lo ? lo : (pe > hi ? hi : pe);
			edit[i] += F_at(c80, pre, pe - lo) - F_at(c80, pre, ps - lo);
		}
	}
}

int main(int argc, char **argv) {
	const char alpha[] = "0123456789^ACGTNacgt=*\x7f\x80\xff\x01 /:;@[]`{";
	static unsigned char arena_raw[1 << 16];
	unsigned char *arena = (unsigned char *)(((uintptr_t)arena_raw + 15) & ~(uintptr_t)15);
	long iters = argc > 1 ? atol(argv[1]) : 20000, bad = 0, it, checked = 0;
	srand(4242);
	for (it = 0; it < iters; it++) {
		uint32_t off[129], edit[128];
		const uint32_t d = (uint32_t)(rand() % 16);
		const int nrec = 1 + rand() % 128;
		const int mode = rand() % 8;          /* 0: long strings (several passes); 1: letter-heavy; else MD-like */
		uint32_t pos = (uint32_t)(rand() % 40);
		int i;
		uint32_t j;
		memset(arena, (it & 1) ? 'A' : '^', 40000);
		for (i = 0; i < nrec; i++) {
			uint32_t len = (mode == 0) ? (uint32_t)(rand() % 700) : (rand() % 6 == 0 ? 0u : (uint32_t)(rand() % 28));
			if (mode == 0 && rand() % 20 == 0) len = 1500u + (uint32_t)(rand() % 1200);
			if (pos + len + d > 39000u) len = 0;
			off[i] = pos;
			for (j = 0; j < len; j++) {
				unsigned char ch;
				if (mode == 1) ch = (rand() % 5) ? 'A' : (unsigned char)alpha[rand() % (sizeof alpha - 1)];
				else ch = (rand() % 8 == 0) ? (unsigned char)rand() : (unsigned char)alpha[rand() % (sizeof alpha - 1)];
				arena[d + pos + j] = ch;
			}
			pos += len;
		}
		off[nrec] = pos;
		wave_flat(arena, d, off, nrec, edit);
		for (i = 0; i < nrec; i++) {
			MdState b = {0, 0, 0, 0};
			for (j = off[i]; j < off[i + 1]; j++) md_byte(b, arena[d + j], 1);
			checked++;
			if ((int32_t)edit[i] != b.edit) {
				if (bad < 5) fprintf(stderr, "mismatch it=%ld rec=%d len=%u got=%u want=%d\n", it, i, off[i + 1] - off[i], edit[i], b.edit);
				bad++;
			}
		}
	}
	printf("checked=%ld bad=%ld\n", checked, bad);
	return bad != 0;
}
