/* msh_crc32 (carry-less multiplication folding, msh_io.c) against zlib's crc32 on random buffers:
 * every length class (below 64, non-multiples of 16, whole BGZF payloads), every alignment. */
#include "msh.h"
#include <stdio.h>
#include <stdlib.h>
#include <zlib.h>

int main(void) {
	size_t N = 1 << 20, i, bad = 0;
	unsigned char *b = (unsigned char *)malloc(N + 64);
	srand(12345);
	for (i = 0; i < N + 64; i++) b[i] = (unsigned char)rand();
	for (i = 0; i < 4000; i++) {
		size_t off = (size_t)rand() % 64, len = i < 300 ? i : (size_t)rand() % (i < 2000 ? 700 : N);
		uint32_t a = msh_crc32(b + off, len), z = (uint32_t)crc32(crc32(0L, NULL, 0), b + off, (uInt)len);
		if (a != z) {
			if (bad < 5) printf("MISMATCH off %zu len %zu: %08x vs %08x\n", off, len, a, z);
			bad++;
		}
	}
	printf("bad=%zu\n", bad);
	free(b);
	return bad != 0;
}
