/* msx_inflate_par_model.h -- the lane-parallel BGZF inflater (msx_inflate.hip: k_bgzf_inflate_wave) restated on the host,
 * one lane after the other.  Test infrastructure (tests/c/inflate_par_twin.c): it pins the ALGORITHM -- which lane decodes
 * what from where, how the lanes' chains are joined, when a deflate block ends, how matches become pieces and in which order
 * pieces may be copied, what is refused -- on real streams without a GPU; the kernel's bytes are checked against zlib's on the
 * device (tests/test_gpu_inflate.py, scripts/bench_inflate.py: every block).
 *
 * The reader loop's BGZF layer (htslib's bgzf_read under mSamRead, msam_helper.c:246-268) inflates one block at a time, one
 * symbol after the other: a DEFLATE code's position depends on every code before it.  What breaks the chain is that Huffman
 * decoding SELF-SYNCHRONISES: a decoder started at a wrong bit position decodes garbage for a few symbols and then, with high
 * probability, falls onto a true code boundary and stays on the true chain from there on.  So the symbols of a deflate block
 * are decoded by the 64 lanes of a wave at once:
 *
 *   segment    the next IP_LANES x IP_SUB_BITS bits of the stream behind the block's header; lane L owns the tokens (a
 *              literal, or length + distance with their extra bits, or the end-of-block code) that BEGIN in
 *              [seg + L * SUB, seg + (L + 1) * SUB)
 *   pass A     every lane decodes from the first bit of its range (lane 0: from the true position) to the first token
 *              boundary at or behind its range's end, counting only: where it ended, how many bytes and pieces it saw
 *   pass B     rounds: lane L takes the end of lane L - 1 as its start; if that differs from the start it used, it decodes
 *              again.  Lane 0 is true from the start, lane L after L rounds at the latest; almost every lane has fallen onto
 *              the true chain inside its own range in pass A already, so its END was right although its start was not, and
 *              the later rounds find next to nothing to do.  A lane that meets the end-of-block code, an invalid code or the
 *              payload's end stops there and the lanes behind it sit out.
 *              Not converged after IP_MAX_ROUNDS rounds (long tokens: few per lane): four times the bits per lane, again
 *              (the kernel hands the block to the serial kernel when that reaches 4096 bits; the model goes on, counting it).
 *   pass C     exclusive sums of the lanes' byte and piece counts give every lane its place in the output; the lanes decode
 *              a last time, literals go straight to their bytes, matches to a list -- in PIECES of at most 16 bytes: a piece
 *              of a match that does not overlap itself copies `len` bytes from `from`; one that does (distance < length)
 *              copies byte (phase + i) mod period from the period in front of the match, never what the match itself writes
 *   resolve    a window of IP_LANES pieces at a time, one per lane: a piece whose source reaches into the outputs of earlier
 *              pieces of the window waits until exactly those are done, the others copy at once
 *
 * then the next segment from where the last lane ended, or the next deflate block's header behind the end-of-block code.
 * Whatever is wrong on the TRUE chain -- a code without a symbol, a distance before the block's start, more bytes than ISIZE --
 * refuses the block (the serial kernel, then the host's decoder and zlib produce the diagnostics, as before).
 */
#ifndef MSX_INFLATE_PAR_MODEL_H
#define MSX_INFLATE_PAR_MODEL_H
#include <stdint.h>
#include <string.h>

#ifndef IP_LANES
#define IP_LANES 64
#endif
#ifndef IP_SUB_BITS
#define IP_SUB_BITS 384u               /* bits per lane to begin with */
#endif
#define IP_SEG_BITS (IP_LANES * IP_SUB_BITS)
#ifndef IP_MAX_ROUNDS
#define IP_MAX_ROUNDS 8u
#endif

enum { IP_OK = 0, IP_EOB = 1, IP_BAD = 2, IP_PAST = 3, IP_DEAD = 4 };

typedef struct { uint32_t end, nbytes, nmatch; uint8_t st; } ip_lane;

typedef struct {
	uint16_t count[16], sym[320];
} ip_code;

typedef struct {
	const uint8_t *in;
	uint64_t end_bit;              /* bits of payload */
	uint8_t *out;
	uint32_t out_len, pos;         /* bytes produced */
	ip_code ll, d;
	/* statistics */
	uint64_t lane_decodes, tokens, segments, rounds, max_rounds, deflate_blocks, round_hist[16], restarts, handbacks, resolve_rounds, resolve_windows;
} ip_state;

static inline uint32_t ip_bits(const ip_state *S, uint64_t at, int n) {      /* n <= 16 bits at bit `at`; zeros behind the end */
	uint32_t v = 0;
	const uint64_t nb = (S->end_bit + 7) >> 3;
	for (int k = 0; k < 3; k++) {
		const uint64_t b = (at >> 3) + (uint64_t)k;
		if (b < nb) v |= (uint32_t)S->in[b] << (8 * k);
	}
	return (v >> (at & 7)) & ((1u << n) - 1u);
}

/* canonical code from lengths; returns 0 if over-subscribed, or incomplete other than a lone distance code */
static int ip_build(ip_code *c, const uint8_t *lens, int n, int is_dist) {
	int offs[16], left = 1, ncodes = 0;
	memset(c->count, 0, sizeof c->count);
	for (int i = 0; i < n; i++) c->count[lens[i]]++;
	c->count[0] = 0;
	for (int l = 1; l <= 15; l++) { left = left * 2 - c->count[l]; if (left < 0) return 0; ncodes += c->count[l]; }
	if (left > 0 && !(is_dist && ncodes <= 1)) return 0;
	offs[1] = 0;
	for (int l = 1; l < 15; l++) offs[l + 1] = offs[l] + c->count[l];
	for (int i = 0; i < n; i++) if (lens[i]) c->sym[offs[lens[i]]++] = (uint16_t)i;
	return 1;
}
/* one symbol at *at; -1 if the bits are no code */
static inline int ip_sym(const ip_state *S, const ip_code *c, uint64_t *at) {
	int code = 0, first = 0, index = 0;
	for (int l = 1; l <= 15; l++) {
		code |= (int)ip_bits(S, *at + (uint64_t)(l - 1), 1);
		const int cnt = c->count[l];
		if (code - cnt < first) { *at += (uint64_t)l; return c->sym[index + (code - first)]; }
		index += cnt; first += cnt; first <<= 1; code <<= 1;
	}
	return -1;
}

static const uint16_t ip_lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t ip_lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t ip_dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t ip_dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

#define IP_PIECE 16u
typedef struct { uint32_t pos, from; uint16_t len, period, phase; } ip_match;      /* a piece: len <= IP_PIECE; period 0: no overlap */

/* A lane's walk: tokens from `start` while they begin in front of `limit`.  emit: pass C (base = the lane's first output
 * byte, ml = its first slot of the match list); otherwise counting only.  Returns 0 only in pass C, for what refuses the block. */
static int ip_walk(ip_state *S, uint64_t start, uint64_t limit, ip_lane *r, int emit, uint32_t base, ip_match *ml) {
	uint64_t at = start;
	uint32_t nb = 0, nm = 0;
	r->st = IP_OK;
	while (at < limit) {
		if (at >= S->end_bit) { r->st = IP_PAST; break; }
		const int s = ip_sym(S, &S->ll, &at);
		S->tokens++;
		if (s < 0 || s > 285) { r->st = IP_BAD; break; }
		if (s < 256) {
			if (emit) { if (base + nb >= S->out_len) return 0; S->out[base + nb] = (uint8_t)s; }
			nb++;
		} else if (s == 256) {
			r->st = IP_EOB;
			break;
		} else {
			const uint32_t len = ip_lbase[s - 257] + ip_bits(S, at, ip_lext[s - 257]);
			at += ip_lext[s - 257];
			const int ds = ip_sym(S, &S->d, &at);
			if (ds < 0 || ds > 29) { r->st = IP_BAD; break; }
			const uint32_t dist = ip_dbase[ds] + ip_bits(S, at, ip_dext[ds]);
			at += ip_dext[ds];
			if (emit) {
				const uint32_t p = base + nb;
				uint32_t k = 0;
				if (dist > p || len > S->out_len - p) return 0;
				for (uint32_t o = 0; o < len; o += IP_PIECE, k++) {
					ip_match *q = &ml[nm + k];
					q->pos = p + o; q->len = (uint16_t)(len - o < IP_PIECE ? len - o : IP_PIECE);
					if (dist >= len) { q->from = p + o - dist; q->period = 0; q->phase = 0; }
					else { q->from = p - dist; q->period = (uint16_t)dist; q->phase = (uint16_t)(o % dist); }
				}
			}
			nm += (len + IP_PIECE - 1u) / IP_PIECE;
			nb += len;
		}
		if (at > S->end_bit) { r->st = IP_PAST; break; }
	}
	r->end = (uint32_t)at; r->nbytes = nb; r->nmatch = nm;
	return 1;
}

/* the symbols of one deflate block, from bit *at (behind its header) to behind its end-of-block code */
static int ip_block_symbols(ip_state *S, uint64_t *at, ip_match *ml) {
	uint64_t sub = IP_SUB_BITS;            /* bits per lane; stays raised for the rest of the deflate block */
	for (;;) {
		const uint64_t seg = *at;
		ip_lane r[IP_LANES];
		uint64_t used[IP_LANES], rounds;
		int nl;
		for (;;) {
			/* a segment is what the staging area holds (IP_SEG_BITS), however many lanes share it */
			nl = (int)(IP_SEG_BITS / sub);
			S->segments++;
			/* pass A */
			for (int L = 0; L < nl; L++) {
				const uint64_t st = seg + (uint64_t)L * sub;
				used[L] = st;
				if (st >= S->end_bit && L > 0) { r[L].st = IP_DEAD; r[L].end = 0; r[L].nbytes = r[L].nmatch = 0; continue; }
				ip_walk(S, st, seg + (uint64_t)(L + 1) * sub, &r[L], 0, 0, 0);
				S->lane_decodes++;
			}
			/* pass B: Jacobi rounds -- every lane looks at what its left neighbour held BEFORE the round.  Lanes behind the
			 * first one that stopped (end of block, no code, end of payload) are of no interest and sit the round out.
			 * Streams whose tokens are long (a BAM header's text: matches of 258 bytes, 25 bits each) leave a lane too few
			 * tokens to fall onto the true chain and the rounds crawl lane by lane: after IP_MAX_ROUNDS the segment is
			 * started again with four times the bits per lane (a quarter of the lanes), down to one lane. */
			int converged = 0;
			for (rounds = 0; rounds < IP_MAX_ROUNDS || nl == 1; rounds++) {
				ip_lane prev[IP_LANES];
				int changed = 0, stop = nl;
				memcpy(prev, r, sizeof r);
				for (int L = 0; L < nl; L++) if (prev[L].st != IP_OK) { stop = L; break; }
				for (int L = 1; L < nl && L <= stop; L++) {
					const uint64_t ns = prev[L - 1].end;
					if (ns == used[L]) continue;
					used[L] = ns;
					changed = 1;
					ip_walk(S, ns, seg + (uint64_t)(L + 1) * sub, &r[L], 0, 0, 0);
					S->lane_decodes++;
				}
				if (!changed) { converged = 1; break; }
			}
			if (converged) break;
			sub *= 4u;
			S->restarts++;
			if (sub >= 4096u && sub < 4u * 4096u) S->handbacks++;
		}
		S->rounds += rounds;
		if (rounds > S->max_rounds) S->max_rounds = rounds;
		S->round_hist[rounds < 15 ? rounds : 15]++;
		/* the lanes of the true chain: up to the first that did not reach its range's end */
		int last = nl - 1;
		for (int L = 0; L < nl; L++) if (r[L].st != IP_OK) { last = L; break; }
		if (r[last].st == IP_BAD || r[last].st == IP_PAST || r[last].st == IP_DEAD) return 0;
		uint32_t base = S->pos, mb = 0;
		uint64_t total = 0;
		for (int L = 0; L <= last; L++) total += r[L].nbytes;
		if (total > S->out_len - S->pos) return 0;
		/* pass C */
		for (int L = 0; L <= last; L++) {
			ip_lane again;
			if (!ip_walk(S, used[L], seg + (uint64_t)(L + 1) * sub, &again, 1, base, ml + mb)) return 0;
			S->lane_decodes++;
			base += r[L].nbytes; mb += r[L].nmatch;
		}
		/* resolve: a window of IP_LANES matches at a time, one per thread.  A match whose source reaches into the outputs of
		 * earlier matches of the window -- [a0, b], found by two searches over the window's positions -- waits until those
		 * are done; the others copy at once (here: the ready ones of a round in REVERSE order, which must not matter).  A
		 * self-overlapping match reads byte i mod distance, so no match reads its own output. */
		for (uint32_t w0 = 0; w0 < mb; w0 += IP_LANES) {
			const uint32_t nw = mb - w0 < IP_LANES ? mb - w0 : IP_LANES;
			const ip_match *W = ml + w0;
			uint32_t a0[IP_LANES], bb[IP_LANES];
			uint8_t dep[IP_LANES], done[IP_LANES];
			for (uint32_t t = 0; t < nw; t++) {
				const uint32_t from = W[t].from, send = from + (W[t].period ? W[t].period : W[t].len);
				dep[t] = 0; done[t] = 0; a0[t] = bb[t] = 0;
				if (t > 0 && send > W[0].pos) {
					uint32_t lo = 0, hi = t;
					while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (W[mid].pos < send) lo = mid + 1; else hi = mid; }
					bb[t] = lo - 1;
					lo = 0; hi = t;
					while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (W[mid].pos <= from) lo = mid + 1; else hi = mid; }
					a0[t] = lo ? lo - 1 : 0;
					if (W[a0[t]].pos + W[a0[t]].len <= from) a0[t]++;
					dep[t] = a0[t] <= bb[t];
				}
			}
			for (uint32_t left = nw; left;) {
				uint8_t snap[IP_LANES];
				memcpy(snap, done, nw);
				S->resolve_rounds++;
				for (uint32_t t = nw; t-- > 0;) {
					if (snap[t]) continue;
					int ready = 1;
					if (dep[t]) for (uint32_t j = a0[t]; j <= bb[t]; j++) if (!snap[j]) ready = 0;
					if (!ready) continue;
					for (uint32_t i = 0; i < W[t].len; i++)
						S->out[W[t].pos + i] = S->out[W[t].from + (W[t].period ? (W[t].phase + i) % W[t].period : i)];
					done[t] = 1;
					left--;
				}
			}
			S->resolve_windows++;
		}
		S->pos = base;
		*at = r[last].end;
		if (r[last].st == IP_EOB) return 1;
	}
}

/* one BGZF block's payload; 1 = inflated to exactly out_len bytes, 0 = refused */
static int ip_inflate(ip_state *S, const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_len, ip_match *ml) {
	S->in = in; S->end_bit = (uint64_t)in_len * 8u; S->out = out; S->out_len = out_len; S->pos = 0;
	uint64_t at = 0;
	if (out_len == 0) return 1;
	for (;;) {
		if (at + 3 > S->end_bit) return 0;
		const uint32_t last = ip_bits(S, at, 1), type = ip_bits(S, at + 1, 2);
		at += 3;
		if (type == 3) return 0;
		if (type == 0) {
			at = (at + 7) & ~7ull;
			if (at + 32 > S->end_bit) return 0;
			const uint32_t len = ip_bits(S, at, 16), nlen = ip_bits(S, at + 16, 16);
			at += 32;
			if ((len ^ 0xffffu) != nlen || at + 8ull * len > S->end_bit || len > out_len - S->pos) return 0;
			memcpy(out + S->pos, in + (at >> 3), len);
			S->pos += len;
			at += 8ull * len;
		} else {
			uint8_t lens[320];
			int hlit = 288, hdist = 32;
			if (type == 1) {
				for (int s = 0; s < 320; s++) lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5);
			} else {
				static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
				uint8_t pl[19];
				ip_code pc;
				hlit = (int)ip_bits(S, at, 5) + 257; hdist = (int)ip_bits(S, at + 5, 5) + 1;
				const int hclen = (int)ip_bits(S, at + 10, 4) + 4;
				at += 14;
				if (hlit > 286 || hdist > 30) return 0;
				memset(pl, 0, sizeof pl);
				for (int i = 0; i < hclen; i++) { pl[order[i]] = (uint8_t)ip_bits(S, at, 3); at += 3; }
				if (at > S->end_bit) return 0;
				{   /* the code-length code must be complete */
					int left = 1;
					memset(pc.count, 0, sizeof pc.count);
					for (int i = 0; i < 19; i++) pc.count[pl[i]]++;
					pc.count[0] = 0;
					for (int l = 1; l <= 7; l++) left = left * 2 - pc.count[l];
					if (left != 0) return 0;
					if (!ip_build(&pc, pl, 19, 0)) return 0;
				}
				int n = 0, prev = 0;
				while (n < hlit + hdist) {
					const int s = ip_sym(S, &pc, &at);
					if (s < 0) return 0;
					if (s < 16) { lens[n++] = (uint8_t)s; prev = s; }
					else {
						int rep, v = 0;
						if (s == 16) { if (n == 0) return 0; v = prev; rep = 3 + (int)ip_bits(S, at, 2); at += 2; }
						else if (s == 17) { rep = 3 + (int)ip_bits(S, at, 3); at += 3; prev = 0; }
						else { rep = 11 + (int)ip_bits(S, at, 7); at += 7; prev = 0; }
						if (n + rep > hlit + hdist) return 0;
						while (rep--) lens[n++] = (uint8_t)v;
					}
					if (at > S->end_bit) return 0;
				}
				if (lens[256] == 0) return 0;
			}
			if (!ip_build(&S->ll, lens, hlit, 0) || !ip_build(&S->d, lens + hlit, hdist, 1)) return 0;
			S->deflate_blocks++;
			if (!ip_block_symbols(S, &at, ml)) return 0;
		}
		if (last) break;
	}
	return S->pos == out_len;
}
#endif
