/* msx_deflate_model.h -- the device DEFLATE encoder's algorithm, one position at a time, in plain C.
 *
 * Host code only (tests/c/deflate_twin.c): the serial restatement of what k_bgzf_deflate's 64 lanes do side by side
 * (msx_deflate.hip), kept next to the kernel so that the two are read together; the constants below (code tables,
 * hash multipliers, token layout) are the ones the kernel uses.  Nothing in libmsamtools_amd.so calls these functions.
 * What the encoder replaces: bgzf_write -> deflate under sam_write1 (msam_helper.c:270-272, "wb": msam_filter.c:464-470). */
#ifndef MSX_DEFLATE_MODEL_H
#define MSX_DEFLATE_MODEL_H
#include <stdint.h>
#include <string.h>

#define DF_PAYLOAD 0xff00u            /* input bytes per BGZF block */
#define DF_SLOT (DF_PAYLOAD + 1024u)  /* bytes reserved per block while it is being built */
#define DF_MAXMATCH 258u
#define DF_NLL 286                    /* literal/length symbols in use (0..285) */
#define DF_ND 30                      /* distance symbols */
#define DF_NCL 19                     /* code-length symbols */
#define DF_MUL4 2654435761u
#define DF_MUL8 0x9E3779B185EBCA87ull
/* a token: literal = the byte; match = DF_TOK_MATCH | (len - 3) << 15 | (dist - 1) */
#define DF_TOK_MATCH 0x80000000u

typedef struct {
	uint32_t window;     /* largest distance looked at (<= 32768) */
	uint32_t step;       /* positions resolved side by side (64 on the device) */
	int hash_bits;       /* log2 of the 4-byte table's size */
	int use_h8;          /* second table keyed by the next 8 bytes */
	int use_rep;         /* the match at distance 1 (runs) */
	int lazy;            /* one-step lazy evaluation inside a step */
	int fixed_only;      /* fixed Huffman codes only (first milestone; no header, no tree) */
	int hash_bits8;      /* log2 of the 8-byte table's size */
} df_opts;

/* What the level dials on the device is the geometry of pass 1 -- window, table sizes: the LDS a wave takes, hence the waves a
 * compute unit keeps resident (msx_bgzf_deflate_launch): levels 7-9 <window 8192, tables 11 / 11 bits> (round 4's encoder),
 * 4-6 <2560, 10 / 11> (what -b asks for: htslib's default level is 6), 1-3 <2560, 10 / 10>. */
static inline df_opts df_opts_for_level(int level) {
	df_opts o = {8192u, 64u, 11, 1, 1, 1, 0, 11};
	if (level <= 6) { o.window = 2560u; o.hash_bits = 10; o.hash_bits8 = level >= 4 ? 11 : 10; }
	return o;
}
static inline df_opts df_default_opts(void) { return df_opts_for_level(6); }

/* ---- RFC 1951 tables ---------------------------------------------------------------------------------------------- */
static const uint16_t df_len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t df_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t df_dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t df_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t df_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

/* length 3..258 -> symbol index 0..28 (arithmetic, as the kernel does it: no table lookup per token) */
static inline uint32_t df_len_sym(uint32_t len) {
	uint32_t l = len - 3u;
	if (l < 8u) return l;
	if (len == 258u) return 28u;
	{ uint32_t b = 31u - (uint32_t)__builtin_clz(l); return ((b - 1u) << 2) + ((l >> (b - 2u)) & 3u); }
}
/* distance 1..32768 -> symbol 0..29 */
static inline uint32_t df_dist_sym(uint32_t dist) {
	uint32_t d = dist - 1u;
	if (d < 4u) return d;
	{ uint32_t b = 31u - (uint32_t)__builtin_clz(d); return (b << 1) + ((d >> (b - 1u)) & 1u); }
}

/* ---- Huffman code lengths ------------------------------------------------------------------------------------------
 * freq[n] -> len[n], lengths <= maxbits; at least two symbols get a code (zlib's rule: a tree is never a lone code).
 * Symbols are ranked by (frequency, symbol); Moffat & Katajainen's in-place algorithm gives the depths of the ranked
 * symbols; lengths above the limit are repaired on the counts per length as zlib's gen_bitlen does, and the counts
 * are handed back out in rank order (rarest symbols get the longest codes). */
static inline void df_huff_lengths(const uint32_t *freq_in, int n, int maxbits, uint8_t *len) {
	uint32_t freq[DF_NLL], A[DF_NLL];
	int order[DF_NLL], m = 0, i, j;
	for (i = 0; i < n; i++) { freq[i] = freq_in[i]; len[i] = 0; }
	for (i = 0; i < n; i++) m += freq[i] != 0;
	for (i = 0; m < 2; i++)              /* force two codes: the lowest unused symbols, frequency 1 */
		if (freq[i] == 0) { freq[i] = 1; m++; }
	/* rank: position of symbol i among the used ones by (freq, symbol) -- on the device every lane counts for its symbols */
	for (i = 0; i < n; i++) {
		if (!freq[i]) continue;
		int r = 0;
		for (j = 0; j < n; j++)
			if (freq[j] && (freq[j] < freq[i] || (freq[j] == freq[i] && j < i))) r++;
		order[r] = i;
		A[r] = freq[i];
	}
	if (m == 2) { len[order[0]] = 1; len[order[1]] = 1; return; }
	{
		int root = 0, leaf = 2, next;
		A[0] += A[1];
		for (next = 1; next < m - 1; next++) {
			if (leaf >= m || A[root] < A[leaf]) { A[next] = A[root]; A[root++] = (uint32_t)next; } else A[next] = A[leaf++];
			if (leaf >= m || (root < next && A[root] < A[leaf])) { A[next] += A[root]; A[root++] = (uint32_t)next; } else A[next] += A[leaf++];
		}
		A[m - 2] = 0;
		for (next = m - 3; next >= 0; next--) A[next] = A[A[next]] + 1;
		{
			int avbl = 1, used = 0, dpth = 0;
			root = m - 2; next = m - 1;
			while (avbl > 0) {
				while (root >= 0 && (int)A[root] == dpth) { used++; root--; }
				while (avbl > used) { A[next--] = (uint32_t)dpth; avbl--; }
				avbl = 2 * used; dpth++; used = 0;
			}
		}
	}
	/* A[r] = depth of the r-th rarest symbol (non-increasing in r) */
	{
		int cnt[40], bits;
		long excess;
		memset(cnt, 0, sizeof cnt);
		for (i = 0; i < m; i++) {
			int d = (int)A[i];
			if (d > maxbits) d = maxbits;
			cnt[d]++;
		}
		/* Leaves deeper than the limit were moved up to it: the code is over-subscribed by `excess` units of
		 * 2^-maxbits (Kraft sum).  zlib's step (trees.c gen_bitlen) -- a leaf of the deepest level below the limit that has
		 * one goes down a level and takes a leaf of the limit's level along as its sibling -- takes exactly one unit away.
		 * (zlib counts the steps from the nodes it clamps on its way down the tree; counted from the leaves' true depths
		 * the number must come from the Kraft sum: leaves two and more levels too deep weigh more than half a unit each --
		 * the first version counted them as a half and wrote over-subscribed 7-bit code-length codes.) */
		excess = -(1L << maxbits);
		for (bits = 1; bits <= maxbits; bits++) excess += (long)cnt[bits] << (maxbits - bits);
		while (excess > 0) {
			bits = maxbits - 1;
			while (cnt[bits] == 0) bits--;
			cnt[bits]--;
			cnt[bits + 1] += 2;
			cnt[maxbits]--;
			excess--;
		}
		i = 0;
		for (bits = maxbits; bits >= 1; bits--)
			for (j = 0; j < cnt[bits]; j++) len[order[i++]] = (uint8_t)bits;
	}
}

/* canonical codes (RFC 1951 3.2.2), bit-reversed so that they can be written LSB first */
static inline void df_huff_codes(const uint8_t *len, int n, uint16_t *code) {
	uint32_t cnt[16], next[16], c = 0;
	int i, b;
	memset(cnt, 0, sizeof cnt);
	for (i = 0; i < n; i++) cnt[len[i]]++;
	cnt[0] = 0;
	for (b = 1; b <= 15; b++) { c = (c + cnt[b - 1]) << 1; next[b] = c; }
	for (i = 0; i < n; i++) {
		uint32_t l = len[i], v, r = 0, k;
		if (!l) { code[i] = 0; continue; }
		v = next[l]++;
		for (k = 0; k < l; k++) r |= ((v >> k) & 1u) << (l - 1 - k);
		code[i] = (uint16_t)r;
	}
}

/* ---- bit writer ----------------------------------------------------------------------------------------------------- */
typedef struct { uint8_t *p; uint64_t acc; uint32_t nacc; size_t n; } df_bw;
static inline void df_put(df_bw *w, uint64_t v, uint32_t nb) {       /* nb <= 48 */
	w->acc |= v << w->nacc;
	w->nacc += nb;
	while (w->nacc >= 8) { w->p[w->n++] = (uint8_t)w->acc; w->acc >>= 8; w->nacc -= 8; }
}
static inline void df_flush(df_bw *w) { if (w->nacc) { w->p[w->n++] = (uint8_t)w->acc; w->acc = 0; w->nacc = 0; } }

/* ---- pass 1: tokens ------------------------------------------------------------------------------------------------- */
static inline uint32_t df_ld32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t df_ld64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t df_match_len(const uint8_t *in, uint32_t c, uint32_t p, uint32_t maxl) {
	uint32_t l = 0;
	while (l < maxl && in[c + l] == in[p + l]) l++;
	return l;
}

/* tokens of in[0..n) -> tok[], returns their number; lf/df: symbol frequencies (end-of-block included) */
static inline uint32_t df_tokens(const uint8_t *in, uint32_t n, const df_opts *O, uint32_t *tok, uint32_t *lf, uint32_t *dfq) {
	uint32_t *h4 = (uint32_t *)calloc((size_t)1 << O->hash_bits, 4), *h8 = (uint32_t *)calloc((size_t)1 << O->hash_bits8, 4);
	uint32_t ml[1024], md[1024], nt = 0, next_free = 0, p0, i;
	memset(lf, 0, 4 * DF_NLL);
	memset(dfq, 0, 4 * DF_ND);
	for (p0 = 0; p0 < n; p0 += O->step) {
		const uint32_t cnt = n - p0 < O->step ? n - p0 : O->step;
		for (i = 0; i < cnt; i++) {                  /* every lane: the candidates as the tables stood before the step */
			const uint32_t p = p0 + i, maxl = n - p < DF_MAXMATCH ? n - p : DF_MAXMATCH;
			uint32_t bl = 0, bd = 0;
			/* candidates in a fixed order -- the near distances, the 4-byte table, the 8-byte table -- a later one must be
			 * strictly longer (so that the kernel can drop it after one look at the bytes around the best length so far) */
			if (O->use_rep && p + 4 <= n) {          /* the nearest of the distances 1..8 whose next 4 bytes repeat (runs, short periods) */
				uint32_t d;
				for (d = 1; d <= 8u && d <= p; d++)
					if (df_ld32(in + p - d) == df_ld32(in + p)) {
						bl = df_match_len(in, p - d, p, maxl);
						bd = d;
						break;
					}
			}
			if (p + 4 <= n) {
				const uint32_t c = h4[(df_ld32(in + p) * DF_MUL4) >> (32 - O->hash_bits)];
				if (c && p - (c - 1) <= O->window) {
					const uint32_t l = df_match_len(in, c - 1, p, maxl);
					if (l >= 4 && l > bl) { bl = l; bd = p - (c - 1); }
				}
			}
			if (O->use_h8 && p + 8 <= n) {
				const uint32_t c = h8[(uint32_t)((df_ld64(in + p) * DF_MUL8) >> (64 - O->hash_bits8))];
				if (c && p - (c - 1) <= O->window) {
					const uint32_t l = df_match_len(in, c - 1, p, maxl);
					if (l >= 4 && l > bl) { bl = l; bd = p - (c - 1); }
				}
			}
			ml[i] = bl; md[i] = bd;
		}
		for (i = 0; i < cnt; i++) {                  /* insert: the highest position of the step wins a slot */
			const uint32_t p = p0 + i;
			uint32_t d, near = 0;
			/* a position that repeats at a near distance stays out of the tables: inside runs and short periods every
			 * position would enter with the same keys, and what the tables then remember is the nearest repeat -- not the
			 * start of the run in an earlier record, from where the long match goes on behind the run's end */
			if (O->use_rep && p + 4 <= n)
				for (d = 1; d <= 8u && d <= p; d++)
					if (df_ld32(in + p - d) == df_ld32(in + p)) { near = 1; break; }
			if (near) continue;
			if (p + 4 <= n) h4[(df_ld32(in + p) * DF_MUL4) >> (32 - O->hash_bits)] = p + 1;
			if (O->use_h8 && p + 8 <= n) h8[(uint32_t)((df_ld64(in + p) * DF_MUL8) >> (64 - O->hash_bits8))] = p + 1;
		}
		for (i = 0; i < cnt; i++) {                  /* resolve in order */
			const uint32_t p = p0 + i;
			if (p < next_free) continue;
			if (ml[i] >= 3 && !(O->lazy && i + 1 < cnt && ml[i + 1] > ml[i])) {
				tok[nt++] = DF_TOK_MATCH | (ml[i] - 3u) << 15 | (md[i] - 1u);
				lf[257 + df_len_sym(ml[i])]++;
				dfq[df_dist_sym(md[i])]++;
				next_free = p + ml[i];
			} else {
				tok[nt++] = in[p];
				lf[in[p]]++;
				next_free = p + 1;
			}
		}
	}
	lf[256]++;
	free(h4);
	free(h8);
	return nt;
}

/* ---- the dynamic header: code lengths of both trees, run-length coded ------------------------------------------------
 * seq[]: symbols 0..18 with their extra-bit values in the high half (sym | extra << 8); returns the count */
static inline int df_rle_lengths(const uint8_t *lens, int n, uint16_t *seq, uint32_t *clf) {
	int i = 0, ns = 0;
	while (i < n) {
		const uint8_t v = lens[i];
		int run = 1;
		while (i + run < n && lens[i + run] == v) run++;
		if (v == 0) {
			int r = run;
			while (r >= 11) { const int t = r > 138 ? 138 : r; seq[ns++] = (uint16_t)(18 | (t - 11) << 8); clf[18]++; r -= t; }
			if (r >= 3) { seq[ns++] = (uint16_t)(17 | (r - 3) << 8); clf[17]++; r = 0; }
			while (r-- > 0) { seq[ns++] = 0; clf[0]++; }
		} else {
			int r = run - 1;
			seq[ns++] = v; clf[v]++;
			while (r >= 3) { const int t = r > 6 ? 6 : r; seq[ns++] = (uint16_t)(16 | (t - 3) << 8); clf[16]++; r -= t; }
			while (r-- > 0) { seq[ns++] = v; clf[v]++; }
		}
		i += run;
	}
	return ns;
}

/* ---- one BGZF block -------------------------------------------------------------------------------------------------
 * in[0..n) (n <= DF_PAYLOAD; in must be readable 8 bytes past n) -> out (DF_SLOT bytes); returns the block's size;
 * *kind: 0 stored, 1 fixed codes, 2 dynamic codes */
static inline uint32_t df_crc32(const uint8_t *p, uint32_t n) {
	static uint32_t tab[256];
	uint32_t c = 0xffffffffu, i, k;
	if (!tab[1]) for (i = 0; i < 256; i++) { uint32_t v = i; for (k = 0; k < 8; k++) v = (v >> 1) ^ ((v & 1u) ? 0xedb88320u : 0u); tab[i] = v; }
	for (i = 0; i < n; i++) c = tab[(c ^ p[i]) & 0xffu] ^ (c >> 8);
	return ~c;
}

static inline uint32_t df_block(const uint8_t *in, uint32_t n, uint8_t *out, const df_opts *O, uint32_t *kind) {
	static const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
	uint32_t *tok = (uint32_t *)malloc(4 * (size_t)(n + 1));
	uint32_t lf[DF_NLL], dq[DF_ND], clf[DF_NCL], nt, t;
	uint8_t ll[DF_NLL + DF_ND], cl[DF_NCL];
	uint16_t lc[DF_NLL], dc[DF_ND], cc[DF_NCL], seq[DF_NLL + DF_ND];
	uint64_t bits_dyn = 0, bits_fix = 0, extra = 0;
	int hlit, hdist, hclen, ns, i;
	df_bw w = {out + 18, 0, 0, 0};
	nt = df_tokens(in, n, O, tok, lf, dq);
	/* the dynamic trees and what they cost */
	df_huff_lengths(lf, DF_NLL, 15, ll);
	df_huff_lengths(dq, DF_ND, 15, ll + DF_NLL);
	for (hlit = DF_NLL; hlit > 257 && ll[hlit - 1] == 0; hlit--) {}
	for (hdist = DF_ND; hdist > 1 && ll[DF_NLL + hdist - 1] == 0; hdist--) {}
	{
		uint8_t both[DF_NLL + DF_ND];
		memcpy(both, ll, (size_t)hlit);
		memcpy(both + hlit, ll + DF_NLL, (size_t)hdist);
		memset(clf, 0, sizeof clf);
		ns = df_rle_lengths(both, hlit + hdist, seq, clf);
	}
	df_huff_lengths(clf, DF_NCL, 7, cl);
	for (hclen = DF_NCL; hclen > 4 && cl[df_cl_order[hclen - 1]] == 0; hclen--) {}
	for (i = 0; i < 29; i++) extra += (uint64_t)lf[257 + i] * df_len_extra[i];
	for (i = 0; i < DF_ND; i++) extra += (uint64_t)dq[i] * df_dist_extra[i];
	bits_dyn = 3 + 5 + 5 + 4 + 3 * (uint64_t)hclen + extra;
	for (i = 0; i < DF_NCL; i++) bits_dyn += (uint64_t)clf[i] * cl[i];
	bits_dyn += 2 * (uint64_t)clf[16] + 3 * (uint64_t)clf[17] + 7 * (uint64_t)clf[18];
	for (i = 0; i < DF_NLL; i++) bits_dyn += (uint64_t)lf[i] * ll[i];
	for (i = 0; i < DF_ND; i++) bits_dyn += (uint64_t)dq[i] * ll[DF_NLL + i];
	{
		/* the three codes must be complete (Kraft sum exactly 1); should they ever not be, the block is written with the
		 * fixed codes or stored -- as the kernel does */
		uint64_t kl = 0, kd = 0, kc = 0;
		for (i = 0; i < DF_NLL; i++) if (ll[i]) kl += 1ull << (15 - ll[i]);
		for (i = 0; i < DF_ND; i++) if (ll[DF_NLL + i]) kd += 1ull << (15 - ll[DF_NLL + i]);
		for (i = 0; i < DF_NCL; i++) if (cl[i]) kc += 1ull << (7 - cl[i]);
		if (kl != 1ull << 15 || kd != 1ull << 15 || kc != 1ull << 7) bits_dyn = 0xffffffffull;
	}
	bits_fix = 3 + extra;
	for (i = 0; i < DF_NLL; i++) bits_fix += (uint64_t)lf[i] * (i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8);
	for (i = 0; i < DF_ND; i++) bits_fix += (uint64_t)dq[i] * 5;
	{
		const uint64_t bits_stored = 8 * (5 + (uint64_t)n);
		int use = O->fixed_only ? 1 : (bits_fix <= bits_dyn ? 1 : 2);
		const uint64_t best = use == 1 ? bits_fix : bits_dyn;
		if (bits_stored <= best || n == 0) use = 0;
		*kind = (uint32_t)use;
		if (use == 0) {
			out[18] = 1; out[19] = (uint8_t)n; out[20] = (uint8_t)(n >> 8); out[21] = (uint8_t)~n; out[22] = (uint8_t)(~n >> 8);
			memcpy(out + 23, in, n);
			w.n = 5 + n;
		} else {
			if (use == 1) {
				for (i = 0; i < DF_NLL; i++) ll[i] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8);
				ll[286 - 1] = 8;     /* (286 and 287 take part in the code's construction) */
				{
					uint8_t fl[288];
					uint16_t fc[288];
					for (i = 0; i < 288; i++) fl[i] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8);
					df_huff_codes(fl, 288, fc);
					memcpy(lc, fc, sizeof lc);
				}
				for (i = 0; i < DF_ND; i++) ll[DF_NLL + i] = 5;
				{
					uint8_t fl[32];
					uint16_t fc[32];
					memset(fl, 5, 32);
					df_huff_codes(fl, 32, fc);
					memcpy(dc, fc, sizeof dc);
				}
				df_put(&w, 1 | 1 << 1, 3);
			} else {
				df_huff_codes(ll, DF_NLL, lc);
				df_huff_codes(ll + DF_NLL, DF_ND, dc);
				df_huff_codes(cl, DF_NCL, cc);
				df_put(&w, 1 | 2 << 1, 3);
				df_put(&w, (uint64_t)(hlit - 257), 5);
				df_put(&w, (uint64_t)(hdist - 1), 5);
				df_put(&w, (uint64_t)(hclen - 4), 4);
				for (i = 0; i < hclen; i++) df_put(&w, cl[df_cl_order[i]], 3);
				for (i = 0; i < ns; i++) {
					const int s = seq[i] & 0xff, e = seq[i] >> 8;
					df_put(&w, cc[s], cl[s]);
					if (s == 16) df_put(&w, (uint64_t)e, 2);
					else if (s == 17) df_put(&w, (uint64_t)e, 3);
					else if (s == 18) df_put(&w, (uint64_t)e, 7);
				}
			}
			for (t = 0; t < nt; t++) {
				const uint32_t k = tok[t];
				if (k & DF_TOK_MATCH) {
					const uint32_t len = ((k >> 15) & 0xffu) + 3u, dist = (k & 0x7fffu) + 1u;
					const uint32_t ls = df_len_sym(len), ds = df_dist_sym(dist);
					df_put(&w, lc[257 + ls], ll[257 + ls]);
					df_put(&w, len - df_len_base[ls], df_len_extra[ls]);
					df_put(&w, dc[ds], ll[DF_NLL + ds]);
					df_put(&w, dist - df_dist_base[ds], df_dist_extra[ds]);
				} else {
					df_put(&w, lc[k], ll[k]);
				}
			}
			df_put(&w, lc[256], ll[256]);
			df_flush(&w);
			if (w.n != (best + 7) / 8) { /* the cost model and the writer must agree */ *kind = 99; }
		}
	}
	free(tok);
	{
		const uint32_t total = 18 + (uint32_t)w.n + 8, crc = df_crc32(in, n);
		memcpy(out, head, 16);
		out[16] = (uint8_t)((total - 1) & 0xff); out[17] = (uint8_t)((total - 1) >> 8);
		memcpy(out + 18 + w.n, &crc, 4);
		memcpy(out + 18 + w.n + 4, &n, 4);
		return total;
	}
}
#endif
