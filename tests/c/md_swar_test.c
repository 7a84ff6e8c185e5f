/* Host harness: the dword-at-a-time MD walks of the stats kernel (md_word, and
 * md_word_aligned on realigned words as the kernel's fast path forms them) must
 * agree with the byte-at-a-time rule (md_byte) on arbitrary byte strings at
 * every alignment.  Built and run by tests/test_md_swar_cpu.py. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../msamtools_amd/csrc/msx_md.h"

int main(int argc, char **argv) {
	const char alpha[] = "0123456789^ACGTNacgt=*\x7f\x80\xff\x01 /:;@[]`{";
	unsigned char buf[64 + 8];
	long iters = argc > 1 ? atol(argv[1]) : 2000000, bad = 0, it;
	srand(12345);
	for (it = 0; it < iters; it++) {
		int off = rand() % 4, n = rand() % 28, i;
		MdState a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
		uint32_t bs = (uint32_t)off, be = (uint32_t)(off + n), w;
		memset(buf, (it & 1) ? 'A' : '5', sizeof buf);      /* neighbouring bytes must not leak in */
		for (i = 0; i < n; i++)
			buf[off + i] = (rand() % 8 == 0) ? (unsigned char)rand() : (unsigned char)alpha[rand() % (sizeof alpha - 1)];
		for (w = bs >> 2; (w << 2) < be; ++w) {
			uint32_t x, p = w << 2;
			memcpy(&x, buf + 4 * w, 4);
			md_word(a, x, bs > p ? bs - p : 0u, be - p < 4u ? be - p : 4u);
		}
		for (i = 0; i < n; i++) md_byte(b, buf[off + i], 1);
		if (a.edit != b.edit) bad++;
		{
			/* the fast path: words realigned to the start of the string, full words then one masked tail */
			MdBits c = {0, 0, 0, 0};
			const uint32_t sh = bs & 3u, w0 = bs >> 2, nfull = (uint32_t)n >> 2, rem = (uint32_t)n & 3u;
			uint32_t lo, hi, q;
			memcpy(&lo, buf + 4 * w0, 4);
			for (q = 0; q < nfull; q++) {
				memcpy(&hi, buf + 4 * (w0 + q + 1), 4);
				md_word_aligned(c, MSX_ALIGNBYTE(hi, lo, sh), 0x80808080u);
				lo = hi;
			}
			if (rem) {
				memcpy(&hi, buf + 4 * (w0 + nfull + 1), 4);
				md_word_aligned(c, MSX_ALIGNBYTE(hi, lo, sh), 0x80808080u >> (8u * (4u - rem)));
			}
			if ((int32_t)c.edit != b.edit) bad++;
		}
	}
	printf("checked=%ld bad=%ld\n", iters, bad);
	return bad != 0;
}
