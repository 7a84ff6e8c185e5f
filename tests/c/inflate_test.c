/* msh_fast_inflate (msh_inflate.c) against zlib: random payloads of six kinds deflated at every level and
 * strategy (stored, fixed, dynamic, Huffman-only, RLE) must come back byte for byte with nothing written past
 * the end; corrupted and truncated streams and wrong expected lengths must be survived (run under
 * -fsanitize=address,undefined by tests/test_inflate_cpu.py).  With an argument: speed against zlib. */
#include "msh.h"
#include <zlib.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
static size_t deflate_raw(const unsigned char *in, size_t n, unsigned char *out, size_t cap, int level, int strategy) {
	z_stream zs; memset(&zs, 0, sizeof zs);
	deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy);
	zs.next_in = (Bytef *)in; zs.avail_in = (uInt)n; zs.next_out = out; zs.avail_out = (uInt)cap;
	if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { printf("deflate failed\n"); exit(2); }
	size_t r = zs.total_out; deflateEnd(&zs); return r;
}
int main(int argc, char **argv) {
	size_t N = 65280, bad = 0, t, cases = 0, rejected = 0;
	unsigned char *src = malloc(N), *comp = malloc(2 * N + 1024), *out = malloc(N + 64);
	srand(777);
	for (t = 0; t < 3000; t++) {
		size_t n = 1 + (size_t)rand() % N, i, clen;
		int kind = rand() % 6, level = 1 + rand() % 9, strat = (rand() % 5 == 0) ? Z_FIXED : (rand() % 7 == 0 ? Z_HUFFMAN_ONLY : (rand() % 9 == 0 ? Z_RLE : Z_DEFAULT_STRATEGY));
		if (rand() % 20 == 0) level = 0;
		for (i = 0; i < n; i++) {
			switch (kind) {
			case 0: src[i] = (unsigned char)rand(); break;                       /* incompressible */
			case 1: src[i] = (unsigned char)("ACGT"[rand() & 3]); break;
			case 2: src[i] = (unsigned char)(i % 97 < 90 ? 'a' + (i * 7 % 13) : rand()); break;
			case 3: src[i] = (unsigned char)(i > 100 && rand() % 10 ? src[i - 1 - rand() % 100] : rand()); break;   /* many matches */
			case 4: src[i] = (unsigned char)(rand() % 50 ? 0 : rand()); break;   /* runs */
			default: src[i] = (unsigned char)((i / 71) * 31 + (i % 71 < 60 ? i % 71 : rand() % 4)); break;   /* record-like */
			}
		}
		clen = deflate_raw(src, n, comp, 2 * N + 1024, level, strat);
		memset(out, 0xEE, n + 64);
		{
			int ok = msh_fast_inflate(comp, clen, out, n);
			cases++;
			if (!ok) { rejected++; if (rejected < 5) printf("REJECTED valid stream: kind %d level %d strat %d n %zu\n", kind, level, strat, n); }
			else if (memcmp(out, src, n) != 0) { bad++; printf("WRONG OUTPUT kind %d level %d n %zu\n", kind, level, n); }
			for (i = n; i < n + 64; i++) if (out[i] != 0xEE) { bad++; printf("WROTE PAST END\n"); break; }
		}
		/* corruptions: must not crash, must not write past the end */
		for (i = 0; i < 6; i++) {
			size_t pos = (size_t)rand() % clen; unsigned char sv = comp[pos];
			size_t cut = (rand() % 3 == 0) ? (size_t)rand() % clen : clen;
			comp[pos] ^= (unsigned char)(1 << (rand() & 7));
			memset(out, 0xEE, n + 64);
			(void)msh_fast_inflate(comp, cut, out, n);
			{ size_t q; for (q = n; q < n + 64; q++) if (out[q] != 0xEE) { bad++; printf("WROTE PAST END (corrupt)\n"); break; } }
			comp[pos] = sv;
		}
		/* wrong expected length */
		(void)msh_fast_inflate(comp, clen, out, n > 1 ? n - 1 : n);
		if (msh_fast_inflate(comp, clen, out, n + 1) == 1) { bad++; printf("accepted a longer length\n"); }
	}
	printf("cases=%zu rejected=%zu bad=%zu\n", cases, rejected, bad);
	if (argc > 1) {   /* speed on a record-like payload */
		size_t n = N, i, clen; struct timespec t0, t1; int r;
		for (i = 0; i < n; i++) src[i] = (unsigned char)((i / 71) * 31 + (i % 71 < 60 ? i % 71 : rand() % 4));
		clen = deflate_raw(src, n, comp, 2 * N + 1024, 6, Z_DEFAULT_STRATEGY);
		clock_gettime(CLOCK_MONOTONIC, &t0);
		for (r = 0; r < 20000; r++) msh_fast_inflate(comp, clen, out, n);
		clock_gettime(CLOCK_MONOTONIC, &t1);
		printf("fast inflate: %.0f MB/s (ratio %.1f)\n", 20000.0 * n / ((t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9) / 1e6, (double)n / clen);
		{
			z_stream zs; memset(&zs, 0, sizeof zs); inflateInit2(&zs, -15);
			clock_gettime(CLOCK_MONOTONIC, &t0);
			for (r = 0; r < 20000; r++) { inflateReset(&zs); zs.next_in = comp; zs.avail_in = (uInt)clen; zs.next_out = out; zs.avail_out = (uInt)n; inflate(&zs, Z_FINISH); }
			clock_gettime(CLOCK_MONOTONIC, &t1);
			printf("zlib inflate: %.0f MB/s\n", 20000.0 * n / ((t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9) / 1e6);
		}
	}
	return bad != 0 || rejected != 0;
}
