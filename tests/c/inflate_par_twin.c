/* The lane-parallel inflater's host restatement (msx_inflate_par_model.h) against zlib.
 *   inflate_par_twin                  random payloads of six kinds at every level and strategy: every valid stream must come
 *                                     back byte for byte; corrupted and truncated ones must be survived (ASan/UBSan build)
 *   inflate_par_twin file.bam [n]     the first n BGZF blocks of a file: equal zlib's bytes, and what the lanes did
 */
#include "../../msamtools_amd/csrc/msx_inflate_par_model.h"
#include <zlib.h>
#include <stdio.h>
#include <stdlib.h>

static size_t deflate_raw(const unsigned char *in, size_t n, unsigned char *out, size_t cap, int level, int strategy, int mem) {
	z_stream zs; memset(&zs, 0, sizeof zs);
	deflateInit2(&zs, level, Z_DEFLATED, -15, mem, strategy);
	zs.next_in = (Bytef *)in; zs.avail_in = (uInt)n; zs.next_out = out; zs.avail_out = (uInt)cap;
	if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { printf("deflate failed\n"); exit(2); }
	size_t r = zs.total_out; deflateEnd(&zs); return r;
}
static void stats(const ip_state *S, size_t blocks, size_t out_bytes) {
	printf("  %zu blocks, %zu bytes out: %llu deflate blocks, %llu segments, %llu lane walks (%.2f per lane and segment), "
	       "%llu symbols decoded (%.2f per output symbol-equivalent), rounds: %.2f per segment, max %llu, %llu restarts with longer lanes (%llu reach 4096 bits: handed back by the kernel)\n",
	       blocks, out_bytes, (unsigned long long)S->deflate_blocks, (unsigned long long)S->segments,
	       (unsigned long long)S->lane_decodes, (double)S->lane_decodes / ((double)S->segments * IP_LANES + 1e-9),
	       (unsigned long long)S->tokens, 0.0, (double)S->rounds / ((double)S->segments + 1e-9), (unsigned long long)S->max_rounds, (unsigned long long)S->restarts, (unsigned long long)S->handbacks);
	printf("  resolve: %.2f levels per window of %d pieces\n", (double)S->resolve_rounds / ((double)S->resolve_windows + 1e-9), IP_LANES);
	printf("  rounds per segment:");
	for (int k = 0; k < 16; k++) printf(" %llu", (unsigned long long)S->round_hist[k]);
	printf("\n");
}
int main(int argc, char **argv) {
	static ip_state S;
	static ip_match ml[65536 / 3 + 65536 / 16 + 64];
	static unsigned char out[65536 + 64], ref[65536 + 64];
	if (argc > 1) {
		FILE *f = fopen(argv[1], "rb");
		if (!f) { perror(argv[1]); return 2; }
		size_t want = argc > 2 ? (size_t)atol(argv[2]) : 1000, nb = 0, bad = 0, total = 0;
		unsigned char hdr[18], *pay = malloc(70000);
		while (nb < want && fread(hdr, 1, 18, f) == 18) {
			const unsigned xlen = hdr[10] | hdr[11] << 8, bsize = (hdr[16] | hdr[17] << 8) + 1u;
			const unsigned plen = bsize - 12 - xlen - 8;
			if (xlen != 6) { printf("unexpected extra field\n"); return 2; }
			if (fread(pay, 1, plen + 8, f) != plen + 8) break;
			const unsigned isize = pay[plen + 4] | pay[plen + 5] << 8 | pay[plen + 6] << 16 | (unsigned)pay[plen + 7] << 24;
			z_stream zs; memset(&zs, 0, sizeof zs);
			inflateInit2(&zs, -15);
			zs.next_in = pay; zs.avail_in = plen; zs.next_out = ref; zs.avail_out = sizeof ref;
			if (inflate(&zs, Z_FINISH) != Z_STREAM_END || zs.total_out != isize) { printf("zlib refuses block %zu\n", nb); return 2; }
			inflateEnd(&zs);
			memset(out, 0xEE, sizeof out);
			if (!ip_inflate(&S, pay, plen, out, isize, ml) || memcmp(out, ref, isize)) { bad++; printf("block %zu: WRONG\n", nb); }
			total += isize;
			nb++;
		}
		stats(&S, nb, total);
		printf("blocks=%zu bad=%zu\n", nb, bad);
		return bad != 0;
	}
	size_t N = 65536, bad = 0, cases = 0, rejected = 0;
	unsigned char *src = malloc(N), *comp = malloc(2 * N + 1024);
	srand(4242);
	const size_t n_cases = getenv("IP_TWIN_CASES") ? (size_t)atol(getenv("IP_TWIN_CASES")) : 1500;
	for (size_t t = 0; t < n_cases; t++) {
		size_t n = (t % 7 == 0) ? N - (size_t)(rand() % 300) : 1 + (size_t)rand() % N, i, clen;
		int kind = rand() % 6, level = 1 + rand() % 9, mem = rand() % 4 == 0 ? 1 + rand() % 8 : 8;
		int strat = (rand() % 5 == 0) ? Z_FIXED : (rand() % 7 == 0 ? Z_HUFFMAN_ONLY : (rand() % 9 == 0 ? Z_RLE : Z_DEFAULT_STRATEGY));
		if (rand() % 20 == 0) level = 0;
		for (i = 0; i < n; i++) {
			switch (kind) {
			case 0: src[i] = (unsigned char)rand(); break;
			case 1: src[i] = (unsigned char)("ACGT"[rand() & 3]); break;
			case 2: src[i] = (unsigned char)(i % 97 < 90 ? 'a' + (i * 7 % 13) : rand()); break;
			case 3: src[i] = (unsigned char)(i > 100 && rand() % 10 ? src[i - 1 - rand() % 100] : rand()); break;
			case 4: src[i] = (unsigned char)(rand() % 50 ? 0 : rand()); break;
			default: src[i] = (unsigned char)((i / 71) * 31 + (i % 71 < 60 ? i % 71 : rand() % 4)); break;
			}
		}
		clen = deflate_raw(src, n, comp, 2 * N + 1024, level, strat, mem);
		if (clen > 65535 + 1024) continue;
		memset(out, 0xEE, n + 64);
		{
			int ok = ip_inflate(&S, comp, (uint32_t)clen, out, (uint32_t)n, ml);
			cases++;
			if (!ok) { rejected++; if (rejected < 5) printf("REJECTED valid stream: kind %d level %d strat %d n %zu\n", kind, level, strat, n); }
			else if (memcmp(out, src, n) != 0) { bad++; printf("WRONG OUTPUT kind %d level %d n %zu\n", kind, level, n); }
			for (i = n; i < n + 64; i++) if (out[i] != 0xEE) { bad++; printf("WROTE PAST END\n"); break; }
		}
		for (i = 0; i < 4; i++) {
			size_t pos = (size_t)rand() % clen; unsigned char sv = comp[pos];
			size_t cut = (rand() % 3 == 0) ? (size_t)rand() % clen : clen;
			comp[pos] ^= (unsigned char)(1 << (rand() & 7));
			memset(out, 0xEE, n + 64);
			(void)ip_inflate(&S, comp, (uint32_t)cut, out, (uint32_t)n, ml);
			{ size_t q; for (q = n; q < n + 64; q++) if (out[q] != 0xEE) { bad++; printf("WROTE PAST END (corrupt)\n"); break; } }
			comp[pos] = sv;
		}
		(void)ip_inflate(&S, comp, (uint32_t)clen, out, (uint32_t)(n > 1 ? n - 1 : n), ml);
		if (n < N && ip_inflate(&S, comp, (uint32_t)clen, out, (uint32_t)n + 1, ml) == 1) { bad++; printf("accepted a longer length\n"); }
	}
	stats(&S, cases, 0);
	printf("cases=%zu rejected=%zu bad=%zu\n", cases, rejected, bad);
	return bad != 0 || rejected != 0;
}
