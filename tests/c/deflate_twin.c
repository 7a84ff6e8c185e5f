/* deflate_twin.c -- host twin of the device DEFLATE encoder (msamtools_amd/csrc/msx_deflate.hip, k_bgzf_deflate).
 *
 * Test infrastructure: restates, one position at a time, what the kernel's 64 lanes do side by side, so that (a) the
 * compression ratio of a design can be measured without a GPU, (b) the CPU test-suite can check that the algorithm's
 * blocks decode (zlib) to their input, and (c) the GPU tests can ask for the kernel's bytes to equal these, bit for bit.
 * Nothing in the product links or runs this file.
 *
 * The algorithm (per BGZF block of <= 0xff00 input bytes; one wave on the device):
 *   pass 1  LZ77, in steps of DF_STEP positions.  Every position hashes its next 4 bytes (and, second table, its next 8),
 *           reads the most recent earlier position with that hash -- as the tables stood BEFORE the step: the lanes of a
 *           step do not see each other -- and measures the match there (<= 258, distance <= DF_WINDOW); positions then
 *           enter the tables (the highest position of a step wins a slot: atomic max).  A position also measures the
 *           match at distance 1 (runs), which no table is needed for.  The step's positions are then resolved
 *           in order: inside an earlier match: skipped; match >= 4 (or 3 at a short distance) and the next position's
 *           match is not longer by more than 1 (one-step lazy evaluation): a match token; otherwise a literal.
 *   Huffman lengths from the token histogram: symbols ranked by (frequency, symbol), Moffat-Katajainen in place on the
 *           ranked frequencies, zlib's overflow rule for lengths above the limit (15; 7 for the code-length code), at
 *           least two used symbols per tree.
 *   pass 2  the dynamic-block header (code lengths run-length coded with symbols 16/17/18) and the tokens as bits.
 *           If a stored block is shorter, the block is stored instead.
 *
 * usage: deflate_twin [-L level] [-w window] [-b bits] [-B bits] [-8 0|1] [-l 0|1] [-r 0|1] [-s step] [-f] <in> [<out.bgzf>]
 *        -L: the geometry the device takes for that level (default 6: msx_deflate_model.h df_opts_for_level)
 *        prints input bytes, output bytes, blocks; with <out> writes the BGZF blocks (no EOF block)
 * gcc -O2 -o deflate_twin deflate_twin.c -lz */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include "../../msamtools_amd/csrc/msx_deflate_model.h"

int main(int argc, char **argv) {
	df_opts O = df_default_opts();
	const char *in_path = NULL, *out_path = NULL;
	int k;
	for (k = 1; k < argc; k++) {
		if (!strcmp(argv[k], "-w") && k + 1 < argc) O.window = (uint32_t)atoi(argv[++k]);
		else if (!strcmp(argv[k], "-8") && k + 1 < argc) O.use_h8 = atoi(argv[++k]);
		else if (!strcmp(argv[k], "-l") && k + 1 < argc) O.lazy = atoi(argv[++k]);
		else if (!strcmp(argv[k], "-r") && k + 1 < argc) O.use_rep = atoi(argv[++k]);
		else if (!strcmp(argv[k], "-s") && k + 1 < argc) O.step = (uint32_t)atoi(argv[++k]);
		else if (!strcmp(argv[k], "-b") && k + 1 < argc) O.hash_bits = atoi(argv[++k]);
		else if (!strcmp(argv[k], "-B") && k + 1 < argc) O.hash_bits8 = atoi(argv[++k]);
		else if (!strcmp(argv[k], "-L") && k + 1 < argc) O = df_opts_for_level(atoi(argv[++k]));     /* (first: later options adjust it) */
		else if (!strcmp(argv[k], "-f")) O.fixed_only = 1;
		else if (!in_path) in_path = argv[k];
		else out_path = argv[k];
	}
	if (!in_path) { fprintf(stderr, "usage: deflate_twin [options] <in> [<out>]\n"); return 2; }
	FILE *f = fopen(in_path, "rb");
	if (!f) { perror(in_path); return 1; }
	fseek(f, 0, SEEK_END);
	long n = ftell(f);
	fseek(f, 0, SEEK_SET);
	uint8_t *d = (uint8_t *)malloc((size_t)n + 16);
	if (fread(d, 1, (size_t)n, f) != (size_t)n) { perror("read"); return 1; }
	memset(d + n, 0, 16);
	fclose(f);
	FILE *fo = out_path ? fopen(out_path, "wb") : NULL;
	uint8_t *blk = (uint8_t *)malloc(DF_SLOT);
	long p, total = 0, nb = 0, n_stored = 0;
	for (p = 0; p < n; p += DF_PAYLOAD) {
		uint32_t len = (uint32_t)(n - p < DF_PAYLOAD ? n - p : DF_PAYLOAD);
		uint32_t kind = 0;
		uint32_t sz = df_block(d + p, len, blk, &O, &kind);
		/* self-check: the block inflates to its input */
		{
			z_stream zs;
			static uint8_t back[DF_PAYLOAD + 16];
			memset(&zs, 0, sizeof zs);
			inflateInit2(&zs, -15);
			zs.next_in = blk + 18; zs.avail_in = sz - 26; zs.next_out = back; zs.avail_out = sizeof back;
			int rc = inflate(&zs, Z_FINISH);
			if (rc != Z_STREAM_END || zs.total_out != len || memcmp(back, d + p, len)) {
				fprintf(stderr, "block %ld: does not inflate to its input (rc %d, %lu bytes of %u)\n", nb, rc, zs.total_out, len);
				return 1;
			}
			inflateEnd(&zs);
			uint32_t crc = (uint32_t)crc32(0, d + p, len);
			if (memcmp(blk + sz - 8, &crc, 4) || memcmp(blk + sz - 4, &len, 4) || (uint32_t)(blk[16] | blk[17] << 8) != sz - 1) {
				fprintf(stderr, "block %ld: bad trailer / BSIZE\n", nb);
				return 1;
			}
		}
		if (fo) fwrite(blk, 1, sz, fo);
		total += sz;
		nb++;
		n_stored += kind == 0;
	}
	if (fo) fclose(fo);
	printf("%ld %ld %ld %ld\n", n, total, nb, n_stored);
	return 0;
}
