/* msh_inflate.c -- a raw-DEFLATE (RFC 1951) decoder for BGZF blocks.
 *
 * Why: with the device side at tens of G alignments/s the command line is bound by what the host does per
 * byte, and on compressed BAM that is zlib's inflate (about 400 MB/s of output per core here).  A BGZF
 * block is a complete, small (<= 64 KB in, <= 64 KB out) deflate stream whose length and CRC-32 are known
 * before it is decoded, which allows a decoder that zlib cannot be: no streaming state, the whole input and
 * the whole output in memory, a 64-bit bit buffer refilled eight bytes at a time, two-level tables indexed by
 * 11 (literal/length) and 8 (distance) bits, matches copied eight bytes at a time.
 *
 * Safety: the decoder never reads outside [in, in + in_len) nor writes outside [out, out + out_len),
 * whatever the input; it returns 0 ("I do not vouch for this") on anything it does not expect -- a code it
 * has no entry for, a distance before the start of the output, output that does not end exactly at
 * out_len, bits consumed past the end of the input -- and the caller then hands the block to zlib, which
 * produces the reference diagnostics.  The caller checks the CRC-32 of what was produced in either case.
 *
 * Written from RFC 1951; the refill and table layout follow the common practice of modern decoders.
 */
#include "msh.h"

#include <string.h>

#define LL_BITS 11
#define D_BITS 8
#define LL_MAX_SUB 16                  /* 2^(15 - LL_BITS) */
#define D_MAX_SUB 128                  /* 2^(15 - D_BITS) */
#define LL_TAB ((1 << LL_BITS) + 288 * LL_MAX_SUB)
#define D_TAB ((1 << D_BITS) + 32 * D_MAX_SUB)

/* table entry: bits 0-7 bits to consume, 8-11 extra bits (or sub-table index bits), 12-15 kind, 16-31 payload */
#define K_LIT (1u << 12)
#define K_BASE (2u << 12)              /* a length or distance base, extra bits follow */
#define K_EOB (3u << 12)
#define K_SUB (4u << 12)
#define K_MASK (15u << 12)

static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static uint32_t ll_entry(int sym) {
	if (sym < 256) return K_LIT | ((uint32_t)sym << 16);
	if (sym == 256) return K_EOB;
	if (sym > 285) return 0;                               /* 286, 287: never valid */
	return K_BASE | ((uint32_t)len_extra[sym - 257] << 8) | ((uint32_t)len_base[sym - 257] << 16);
}

static uint32_t d_entry(int sym) {
	if (sym > 29) return 0;
	return K_BASE | ((uint32_t)dist_extra[sym] << 8) | ((uint32_t)dist_base[sym] << 16);
}

static uint32_t bit_reverse(uint32_t c, int n) {
	uint32_t r = 0;
	int i;
	for (i = 0; i < n; i++) { r = (r << 1) | (c & 1u); c >>= 1; }
	return r;
}

/* Canonical Huffman code of `lens` into a two-level table indexed by the bit-reversed code.  Returns 0 if the
 * lengths over-subscribe the code space.  Unused codes keep entry 0, which the decoder refuses. */
static int build_table(uint32_t *tab, int tbits, int max_sub, const uint8_t *lens, int nsym, uint32_t (*entry)(int), int tab_cap) {
	uint16_t count[16], next[16];
	uint16_t sub_off[1 << LL_BITS];
	uint8_t sub_bits[1 << LL_BITS];
	uint32_t rev_of[288];
	int i, len, used = 1 << tbits, any_long = 0;
	long left = 1;
	memset(count, 0, sizeof count);
	for (i = 0; i < nsym; i++) count[lens[i]]++;
	count[0] = 0;
	for (len = 1; len <= 15; len++) {
		left = left * 2 - count[len];
		if (left < 0) return 0;
	}
	next[1] = 0;
	for (len = 1; len < 15; len++) next[len + 1] = (uint16_t)((next[len] + count[len]) << 1);
	memset(tab, 0, (size_t)(1 << tbits) * sizeof(uint32_t));
	for (i = 0; i < nsym; i++) {
		len = lens[i];
		if (!len) continue;
		rev_of[i] = bit_reverse(next[len]++, len);
		if (len <= tbits) {
			const uint32_t e = entry(i) | (uint32_t)len;
			uint32_t k;
			for (k = rev_of[i]; k < (1u << tbits); k += 1u << len) tab[k] = e;
		} else {
			any_long = 1;
		}
	}
	if (!any_long) return 1;
	memset(sub_bits, 0, (size_t)1 << tbits);
	for (i = 0; i < nsym; i++) {
		len = lens[i];
		if (len > tbits) {
			const uint32_t pre = rev_of[i] & ((1u << tbits) - 1u);
			if (len - tbits > sub_bits[pre]) sub_bits[pre] = (uint8_t)(len - tbits);
		}
	}
	for (i = 0; i < nsym; i++) {
		len = lens[i];
		if (len > tbits) {
			const uint32_t pre = rev_of[i] & ((1u << tbits) - 1u);
			const int sb = sub_bits[pre];
			uint32_t k, e;
			if ((tab[pre] & K_MASK) != K_SUB) {
				if (tab[pre] != 0 || sb > 15 - tbits || (1 << sb) > max_sub || used + (1 << sb) > tab_cap) return 0;
				sub_off[pre] = (uint16_t)used;
				memset(tab + used, 0, ((size_t)1 << sb) * sizeof(uint32_t));
				tab[pre] = K_SUB | ((uint32_t)sb << 8) | (uint32_t)tbits | ((uint32_t)used << 16);
				used += 1 << sb;
			}
			e = entry(i) | (uint32_t)(len - tbits);
			for (k = rev_of[i] >> tbits; k < (1u << sb); k += 1u << (len - tbits)) tab[sub_off[pre] + k] = e;
		}
	}
	return 1;
}

static inline uint64_t load64(const uint8_t *p) {
	uint64_t v;
	memcpy(&v, p, 8);                                    /* (little-endian host: x86-64, the only one this is built for) */
	return v;
}

typedef struct {
	const uint8_t *in, *in_end;
	uint64_t buf;
	int cnt;                                             /* valid bits in buf; negative: more consumed than the input holds */
} bitr;

#define REFILL(b) do { \
	if ((b).cnt < 0) return 0; \
	if ((b).in + 8 <= (b).in_end) { \
		(b).buf |= load64((b).in) << (b).cnt; \
		(b).in += (63 - (b).cnt) >> 3; \
		(b).cnt |= 56; \
	} else { \
		while ((b).cnt <= 56 && (b).in < (b).in_end) { (b).buf |= (uint64_t)*(b).in++ << (b).cnt; (b).cnt += 8; } \
	} } while (0)
#define BITS(b, n) ((uint32_t)((b).buf & (((uint64_t)1 << (n)) - 1u)))
#define DROP(b, n) do { (b).buf >>= (n); (b).cnt -= (int)(n); } while (0)

int msh_fast_inflate(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len) {
	static __thread uint32_t ll_tab[LL_TAB], d_tab[D_TAB];
	static __thread uint32_t fix_ll[LL_TAB], fix_d[D_TAB];
	static __thread int fixed_ready = 0;
	uint8_t *const out0 = out, *const out_end = out + out_len;
	bitr b;
	int last;
	b.in = in; b.in_end = in + in_len; b.buf = 0; b.cnt = 0;
	do {
		const uint32_t *ll, *dt;
		uint32_t type;
		REFILL(b);
		if (b.cnt < 3) return 0;
		last = (int)BITS(b, 1);
		type = (b.buf >> 1) & 3u;
		DROP(b, 3);
		if (type == 0) {
			/* stored: to the next byte boundary, LEN, NLEN, bytes */
			uint32_t len, nlen;
			DROP(b, b.cnt & 7);
			if (b.cnt < 0) return 0;
			b.in -= b.cnt >> 3;                          /* give the whole bytes still in the buffer back */
			b.buf = 0; b.cnt = 0;
			if (b.in_end - b.in < 4) return 0;
			len = (uint32_t)b.in[0] | ((uint32_t)b.in[1] << 8);
			nlen = (uint32_t)b.in[2] | ((uint32_t)b.in[3] << 8);
			b.in += 4;
			if ((len ^ 0xffffu) != nlen || (size_t)(b.in_end - b.in) < len || (size_t)(out_end - out) < len) return 0;
			memcpy(out, b.in, len);
			out += len;
			b.in += len;
			continue;
		}
		if (type == 3) return 0;
		if (type == 1) {
			if (!fixed_ready) {
				uint8_t lens[288];
				int i;
				for (i = 0; i < 144; i++) lens[i] = 8;
				for (; i < 256; i++) lens[i] = 9;
				for (; i < 280; i++) lens[i] = 7;
				for (; i < 288; i++) lens[i] = 8;
				if (!build_table(fix_ll, LL_BITS, LL_MAX_SUB, lens, 288, ll_entry, LL_TAB)) return 0;
				for (i = 0; i < 32; i++) lens[i] = 5;
				if (!build_table(fix_d, D_BITS, D_MAX_SUB, lens, 32, d_entry, D_TAB)) return 0;
				fixed_ready = 1;
			}
			ll = fix_ll; dt = fix_d;
		} else {
			static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
			uint8_t lens[288 + 32 + 140], pre_lens[19];
			uint32_t pre_tab[128];
			int hlit, hdist, hclen, i, n;
			REFILL(b);
			if (b.cnt < 14) return 0;
			hlit = (int)BITS(b, 5) + 257; DROP(b, 5);
			hdist = (int)BITS(b, 5) + 1; DROP(b, 5);
			hclen = (int)BITS(b, 4) + 4; DROP(b, 4);
			if (hlit > 286 || hdist > 30) return 0;
			memset(pre_lens, 0, sizeof pre_lens);
			for (i = 0; i < hclen; i++) {
				REFILL(b);
				pre_lens[order[i]] = (uint8_t)BITS(b, 3);
				DROP(b, 3);
			}
			{   /* the code-length code: at most 7 bits, one direct table; entry = length | symbol << 8, 0 = unused */
				uint16_t count[8], next[8];
				long left = 1;
				int len;
				memset(count, 0, sizeof count);
				for (i = 0; i < 19; i++) count[pre_lens[i]]++;
				count[0] = 0;
				for (len = 1; len <= 7; len++) { left = left * 2 - count[len]; if (left < 0) return 0; }
				next[1] = 0;
				for (len = 1; len < 7; len++) next[len + 1] = (uint16_t)((next[len] + count[len]) << 1);
				memset(pre_tab, 0, sizeof pre_tab);
				for (i = 0; i < 19; i++) {
					uint32_t k;
					len = pre_lens[i];
					if (!len) continue;
					for (k = bit_reverse(next[len]++, len); k < 128; k += 1u << len) pre_tab[k] = (uint32_t)len | ((uint32_t)i << 8) | 0x10000u;
				}
			}
			n = 0;
			while (n < hlit + hdist) {
				uint32_t e, sym;
				REFILL(b);
				e = pre_tab[BITS(b, 7)];
				if (!e) return 0;
				DROP(b, e & 0xff);
				sym = (e >> 8) & 0xff;
				if (sym < 16) {
					lens[n++] = (uint8_t)sym;
				} else {
					int rep;
					uint8_t v = 0;
					if (sym == 16) {
						if (n == 0) return 0;
						v = lens[n - 1];
						rep = 3 + (int)BITS(b, 2); DROP(b, 2);
					} else if (sym == 17) {
						rep = 3 + (int)BITS(b, 3); DROP(b, 3);
					} else {
						rep = 11 + (int)BITS(b, 7); DROP(b, 7);
					}
					if (n + rep > hlit + hdist) return 0;
					memset(lens + n, v, (size_t)rep);
					n += rep;
				}
				if (b.cnt < 0) return 0;
			}
			if (lens[256] == 0) return 0;                    /* a block must be able to end */
			if (!build_table(ll_tab, LL_BITS, LL_MAX_SUB, lens, hlit, ll_entry, LL_TAB)) return 0;
			if (!build_table(d_tab, D_BITS, D_MAX_SUB, lens + hlit, hdist, d_entry, D_TAB)) return 0;
			ll = ll_tab; dt = d_tab;
		}
		/* ---- the symbols of one block ----
		 * The fast loop runs while the longest match plus the overcopy of its last eight-byte step fits into the
		 * output and every refill can load eight bytes: no bounds are checked inside it (a distance before the
		 * start of the output is), and the bit count cannot run out.  The careful loop below finishes the block. */
		while ((size_t)(out_end - out) >= 258 + 3 + 8 && (size_t)(b.in_end - b.in) >= 8) {
			uint32_t e, len, dist;
			const uint8_t *src;
			uint8_t *stop;
			b.buf |= load64(b.in) << b.cnt;
			b.in += (63 - b.cnt) >> 3;
			b.cnt |= 56;
			e = ll[BITS(b, LL_BITS)];
			if ((e & K_MASK) == K_LIT) {
				DROP(b, e & 0xff);
				*out++ = (uint8_t)(e >> 16);
				e = ll[BITS(b, LL_BITS)];
				if ((e & K_MASK) != K_LIT) goto fast_nonlit;
				DROP(b, e & 0xff);
				*out++ = (uint8_t)(e >> 16);
				e = ll[BITS(b, LL_BITS)];
				if ((e & K_MASK) != K_LIT) goto fast_nonlit;
				DROP(b, e & 0xff);
				*out++ = (uint8_t)(e >> 16);
				continue;
			}
fast_nonlit:
			/* (after up to two literals of at most 15 bits each, 26 bits remain for the length code and its extra
			 * bits; the distance needs up to 28 more -- so refill once more unless plenty is left) */
			if ((e & K_MASK) == K_SUB) {
				DROP(b, LL_BITS);
				e = ll[(e >> 16) + BITS(b, (e >> 8) & 15u)];
			}
			DROP(b, e & 0xff);
			if ((e & K_MASK) == K_LIT) { *out++ = (uint8_t)(e >> 16); continue; }
			if ((e & K_MASK) == K_EOB) goto block_done;
			if ((e & K_MASK) != K_BASE) return 0;
			len = (e >> 16) + BITS(b, (e >> 8) & 15u);
			DROP(b, (e >> 8) & 15u);
			if (b.cnt < 28) {
				if ((size_t)(b.in_end - b.in) < 8) {         /* (cannot happen while the loop condition holds for whole words; be safe) */
					while (b.cnt <= 56 && b.in < b.in_end) { b.buf |= (uint64_t)*b.in++ << b.cnt; b.cnt += 8; }
				} else {
					b.buf |= load64(b.in) << b.cnt;
					b.in += (63 - b.cnt) >> 3;
					b.cnt |= 56;
				}
			}
			e = dt[BITS(b, D_BITS)];
			if ((e & K_MASK) == K_SUB) {
				DROP(b, D_BITS);
				e = dt[(e >> 16) + BITS(b, (e >> 8) & 15u)];
			}
			DROP(b, e & 0xff);
			if ((e & K_MASK) != K_BASE) return 0;
			dist = (e >> 16) + BITS(b, (e >> 8) & 15u);
			DROP(b, (e >> 8) & 15u);
			if (b.cnt < 0 || dist > (size_t)(out - out0)) return 0;
			src = out - dist;
			stop = out + len;
			if (dist >= 8) {
				memcpy(out, src, 8);
				memcpy(out + 8, src + 8, 8);
				if (len > 16) {
					out += 16; src += 16;
					do { memcpy(out, src, 8); out += 8; src += 8; } while (out < stop);
				}
			} else if (dist == 1) {
				memset(out, *src, len);
			} else {
				do { *out++ = *src++; } while (out < stop);
			}
			out = stop;
		}
		for (;;) {
			uint32_t e, len, dist;
			REFILL(b);
			e = ll[BITS(b, LL_BITS)];
			if ((e & K_MASK) == K_SUB) {
				DROP(b, LL_BITS);
				e = ll[(e >> 16) + BITS(b, (e >> 8) & 15u)];
			}
			DROP(b, e & 0xff);
			if ((e & K_MASK) == K_LIT) {
				if (out == out_end) return 0;
				*out++ = (uint8_t)(e >> 16);
				/* up to two more literals out of the same refill (56 bits hold three codes of 15) */
				e = ll[BITS(b, LL_BITS)];
				if ((e & K_MASK) != K_LIT || out == out_end) continue;
				DROP(b, e & 0xff);
				*out++ = (uint8_t)(e >> 16);
				e = ll[BITS(b, LL_BITS)];
				if ((e & K_MASK) != K_LIT || out == out_end) continue;
				DROP(b, e & 0xff);
				*out++ = (uint8_t)(e >> 16);
				continue;
			}
			if ((e & K_MASK) == K_EOB) break;
			if ((e & K_MASK) != K_BASE) return 0;
			len = (e >> 16) + BITS(b, (e >> 8) & 15u);
			DROP(b, (e >> 8) & 15u);
			e = dt[BITS(b, D_BITS)];
			if ((e & K_MASK) == K_SUB) {
				DROP(b, D_BITS);
				e = dt[(e >> 16) + BITS(b, (e >> 8) & 15u)];
			}
			DROP(b, e & 0xff);
			if ((e & K_MASK) != K_BASE) return 0;
			dist = (e >> 16) + BITS(b, (e >> 8) & 15u);
			DROP(b, (e >> 8) & 15u);
			if (b.cnt < 0) return 0;
			if (dist > (size_t)(out - out0) || len > (size_t)(out_end - out)) return 0;
			{
				const uint8_t *src = out - dist;
				uint8_t *const stop = out + len;
				if (dist >= 8 && (size_t)(out_end - stop) >= 8) {
					do { memcpy(out, src, 8); out += 8; src += 8; } while (out < stop);   /* (up to 7 bytes past the match, inside the output) */
					out = stop;
				} else if (dist == 1 && (size_t)(out_end - stop) >= 8) {
					memset(out, *src, len);
					out = stop;
				} else {
					do { *out++ = *src++; } while (out < stop);
				}
			}
		}
block_done:
		if (b.cnt < 0) return 0;
	} while (!last);
	return out == out_end && b.cnt >= 0;
}
