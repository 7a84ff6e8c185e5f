// msx_inflate.hip -- BGZF blocks inflated on the device.
//
// mSamRead (msam_helper.c:246-268) reads through htslib's BGZF layer: every 64 KB block of a BAM file is a raw DEFLATE
// stream (RFC 1951) with its inflated length and CRC-32 in the trailer.  With the record walk on the device
// (msx_unpack.hip) the command line was bound by inflate on the host cores; here the compressed blocks are uploaded as
// they are (a sixth of the bytes) and inflated where their records are used.
//
// One wave per block.  DEFLATE is serial inside a block -- every code's position depends on the code before it -- so
// the decode loop runs on wave-uniform values (bit buffer, table entries through readfirstlane: the scalar unit does
// the arithmetic), and the lanes do together what can be done together: building the Huffman tables of a dynamic
// block (code assignment by ballots, table fill one symbol per lane), copying a match (one byte per lane), staging
// the compressed bytes into LDS half a kilobyte ahead, writing the output back.  The launch is persistent: as many
// waves as are to be resident draw block numbers from a ticket.  What a block costs is latency and scalar issue, so
// everything the loop touches lives in LDS, 11.3 KB per wave (fourteen waves per compute unit):
//   ll / dt       primary decode tables indexed by the next 10 / 8 bits (bit-reversed codes); longer codes are rare
//                 symbols and take the canonical route (limit per length, sorted symbols)
//   in            the compressed stream, two halves of 512 bytes; the half behind the read position is refilled from
//                 registers that were loaded a half earlier
//   ring          the last 4 KB of output.  Matches within reach (distance <= 3.5 KB) are LDS-to-LDS copies; farther
//                 ones read what has been written back to global memory (behind a workgroup-scope fence: the wave
//                 reads its own earlier stores).  Every kilobyte the ring's new bytes go out as aligned 16-byte vectors.
// A fast loop decodes the symbols whose codes the primary tables hold, between write-backs and input refills; anything
// else leaves it in front of the symbol for a general step.
// A block the decoder does not vouch for -- a code without a table entry, a distance before the start, lengths that
// do not add up, a CRC mismatch -- is marked in status[] and left to the caller (the command line inflates such a
// batch on the host, whose decoder and zlib produce the reference diagnostics).  No loop is unbounded: every
// iteration consumes input bits or produces output bytes, and running past the block's last bit ends the block.
// k_bgzf_crc checks the CRC-32 of every block's output: passes of 4 KB, 64 bytes per lane, table-driven, the lanes'
// states joined by multiplication with x^(8 * length) mod P.
// Integer / byte work; no MFMA.
#include "msx_internal.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

// LDS per block: 4 KB + 1 KB + 1 KB + 4 KB + 1.3 KB = 11.3 KB, fourteen blocks per compute unit
#define IF_LL_ROOT 10
#define IF_D_ROOT 8
#ifndef IF_RING
#define IF_RING 4096u
#endif
#define IF_RMASK (IF_RING - 1u)
#define IF_FLUSH (IF_RING / 4u)
#define IF_NEAR (IF_RING - 512u)
#ifndef IF_IN_HALF
#define IF_IN_HALF 128u              // words per half of the input ring: 64 lanes x 2
#endif
#define IF_IN_DW (2u * IF_IN_HALF)
#define IF_LANE_DW (IF_IN_HALF / 64u)

// table entry: bits 0-3 code length, 4-7 kind, 8-11 extra bits, 16-31 payload (literal / base)
#define IF_LIT 1u
#define IF_BASE 2u
#define IF_EOB 4u
#define IF_LONG 8u

// status of a block
#define IF_OK 0u
#define IF_BAD_TYPE 1u
#define IF_BAD_CODE 2u
#define IF_BAD_DIST 3u
#define IF_OUT_OVER 4u
#define IF_IN_OVER 5u
#define IF_LEN_MISMATCH 6u
#define IF_BAD_STORED 7u
#define IF_BAD_CRC 8u
#define IF_BAD_LENS 9u

struct IfShared {
	uint32_t ll[1 << IF_LL_ROOT];
	uint32_t dt[1 << IF_D_ROOT];
	uint32_t in[IF_IN_DW];
	__attribute__((aligned(16))) uint8_t ring[IF_RING];
	uint32_t lim[2][16], first[2][16]; // per code length, left-aligned to 15 bits: end and start of its codes
	uint16_t off[2][16];               // first place of a length's symbols in sorted[]
	union {
		uint16_t sorted[320];          // [0, 288) literal/length symbols by code, [288, 320) distance symbols
		uint32_t pre[128];             // the code-length code (7 bits): done with before sorted[] is written
	};
	uint8_t lens[352];
	uint8_t pl[32];
};
#define IF_SORTED(S, which) ((S).sorted + ((which) ? 288 : 0))

#ifndef IF_FAR_LOAD
#define IF_FAR_LOAD(p) (*(const uint8_t *)(p))
#endif
#define IFU(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))

// the order in which a dynamic block sends the lengths of its code-length code (RFC 1951, 3.2.7), five bits each
constexpr uint64_t if_pack_order(int from, int to) {
	constexpr int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
	uint64_t v = 0;
	for (int i = from; i < to; i++) v |= (uint64_t)order[i] << (5 * (i - from));
	return v;
}
#define IF_ORDER_LO if_pack_order(0, 12)
#define IF_ORDER_HI if_pack_order(12, 19)

__device__ __forceinline__ uint32_t if_ll_entry(uint32_t sym) {
	if (sym < 256u) return (IF_LIT << 4) | (sym << 16);
	if (sym == 256u) return IF_EOB << 4;
	if (sym > 285u) return 0u;
	const uint32_t c = sym - 257u;
	if (c < 8u) return (IF_BASE << 4) | ((3u + c) << 16);
	if (c == 28u) return (IF_BASE << 4) | (258u << 16);
	const uint32_t ex = (c >> 2) - 1u;
	return (IF_BASE << 4) | (ex << 8) | ((3u + ((4u + (c & 3u)) << ex)) << 16);
}
__device__ __forceinline__ uint32_t if_d_entry(uint32_t sym) {
	if (sym > 29u) return 0u;
	if (sym < 4u) return (IF_BASE << 4) | ((1u + sym) << 16);
	const uint32_t ex = (sym >> 1) - 1u;
	return (IF_BASE << 4) | (ex << 8) | ((1u + ((2u + (sym & 1u)) << ex)) << 16);
}

// the state of one block's decoder: wave-uniform
struct IfState {
	uint64_t buf;
	int32_t cnt;
	uint32_t ip;          // next dword of the stream (index from the aligned base)
	uint32_t cross;       // dword index at which the next half of `in` is refilled
	uint32_t pos, flushed;
	uint32_t status;
};

struct __attribute__((aligned(4 * IF_LANE_DW))) if_vec { uint32_t v[IF_LANE_DW]; };   // a lane's share of a chunk of the input ring
struct IfIn {
	const uint32_t *g;    // aligned base of the stream
	uint32_t n_bytes;     // bytes that may be read from g
	if_vec ahead;         // this lane's words of the chunk after the two staged ones
};

// word d of the stream where the buffer ends inside or in front of it
__device__ __noinline__ uint32_t if_edge_word(const uint8_t *p, uint32_t n_bytes, uint32_t d) {
	uint32_t v = 0u;
	for (uint32_t k = 0; k < 4u; k++)
		if (4u * d + k < n_bytes) v |= (uint32_t)p[4u * d + k] << (8u * k);
	return v;
}
__device__ __forceinline__ if_vec if_load_chunk(const IfIn &I, uint32_t chunk, uint32_t lane) {
	const uint32_t d = chunk * IF_IN_HALF + lane * IF_LANE_DW;
	if_vec v;
	if (4u * d + 4u * IF_LANE_DW <= I.n_bytes) {
		v = *reinterpret_cast<const if_vec *>(I.g + d);
	} else {
		const uint8_t *p = reinterpret_cast<const uint8_t *>(I.g);
#pragma unroll
		for (uint32_t k = 0; k < IF_LANE_DW; k++) v.v[k] = if_edge_word(p, I.n_bytes, d + k);
	}
	return v;
}
__device__ __forceinline__ void if_store_chunk(IfShared &S, uint32_t chunk, uint32_t lane, const if_vec &v) {
	*reinterpret_cast<if_vec *>(&S.in[(chunk & 1u) * IF_IN_HALF + lane * IF_LANE_DW]) = v;
}

// one word of the stream taken: when the read position enters the next chunk, the chunk behind it is done with -- the
// chunk two ahead takes its half (out of the registers it has been travelling in), the one three ahead starts travelling
__device__ __forceinline__ void if_advance(IfShared &S, IfState &T, IfIn &I, uint32_t lane) {
	T.ip++;
	if (T.ip == T.cross) {
		const uint32_t k = T.ip / IF_IN_HALF;
		if_store_chunk(S, k + 1u, lane, I.ahead);
		I.ahead = if_load_chunk(I, k + 2u, lane);
		T.cross += IF_IN_HALF;
	}
}

// (re)position the reader at byte `byte_off` of the stream.  w: the word at T.ip, read ahead of its use (an LDS read
// waited for only where the next LDS result is waited for anyway)
__device__ __forceinline__ void if_seek(IfShared &S, IfState &T, IfIn &I, uint32_t &w, uint32_t byte_off, uint32_t lane) {
	const uint32_t dw = byte_off >> 2;
	const uint32_t c0 = dw / IF_IN_HALF;
	if_store_chunk(S, c0, lane, if_load_chunk(I, c0, lane));
	if_store_chunk(S, c0 + 1u, lane, if_load_chunk(I, c0 + 1u, lane));
	I.ahead = if_load_chunk(I, c0 + 2u, lane);
	T.cross = (c0 + 1u) * IF_IN_HALF;
	T.ip = dw;
	// two words, then the bytes in front of byte_off go
	T.buf = (uint64_t)IFU(S.in[T.ip & (IF_IN_DW - 1u)]);
	if_advance(S, T, I, lane);
	T.buf |= (uint64_t)IFU(S.in[T.ip & (IF_IN_DW - 1u)]) << 32;
	if_advance(S, T, I, lane);
	w = S.in[T.ip & (IF_IN_DW - 1u)];
	const uint32_t skip = (byte_off & 3u) * 8u;
	T.buf >>= skip;
	T.cnt = 64 - (int32_t)skip;
}

__device__ __forceinline__ void if_refill(IfShared &S, IfState &T, IfIn &I, uint32_t &w, uint32_t lane) {
	if (T.cnt <= 32) {
		T.buf |= (uint64_t)IFU(w) << T.cnt;
		T.cnt += 32;
		if_advance(S, T, I, lane);
		w = S.in[T.ip & (IF_IN_DW - 1u)];
	}
}
#define IF_PEEK(T, n) ((uint32_t)((T).buf & ((1ull << (n)) - 1ull)))
#define IF_DROP(T, n) do { (T).buf >>= (n); (T).cnt -= (int32_t)(n); } while (0)
// bits of the stream consumed so far (from the aligned base)
#define IF_BITPOS(T) ((uint64_t)(T).ip * 32ull - (uint64_t)(T).cnt)

// write the ring's bytes [T.flushed, upto) of the output back.  Ring index and global address agree modulo 16.
__device__ __forceinline__ void if_flush(IfShared &S, IfState &T, uint8_t *og, uint32_t upto, uint32_t lane) {
	const uintptr_t base = (uintptr_t)og & ~(uintptr_t)15;          // ring index 0 modulo IF_RING
	const uintptr_t lo = (uintptr_t)og + T.flushed, hi = (uintptr_t)og + upto;
	const uintptr_t a = lo & ~(uintptr_t)15;
	for (uintptr_t va = a + 16u * lane; va < hi; va += 16u * 64u) {
		const uint32_t ri = (uint32_t)(va - base) & IF_RMASK;
		if (va >= lo && va + 16u <= hi) {
			*reinterpret_cast<uint4 *>(va) = *reinterpret_cast<const uint4 *>(&S.ring[ri]);
		} else {
			for (uint32_t k = 0; k < 16u; k++)
				if (va + k >= lo && va + k < hi) *reinterpret_cast<uint8_t *>(va + k) = S.ring[ri + k];
		}
	}
	T.flushed = upto;
}

// Canonical Huffman code of S.lens[base, base + nsym) into a primary table of `root` bits and the per-length arrays
// of the canonical route.  which: 0 literal/length, 1 distance.  Returns 0 if the lengths are refused.
__device__ __forceinline__ uint32_t if_build(IfShared &S, uint32_t which, uint32_t base, uint32_t nsym, uint32_t lane) {
	uint32_t *tab = which ? S.dt : S.ll;
	const uint32_t root = which ? IF_D_ROOT : IF_LL_ROOT;
	const uint32_t tsize = 1u << root;
	for (uint32_t k = lane; k < tsize; k += 64u) tab[k] = 0u;
	uint32_t L[5], code[5];
#pragma unroll
	for (int j = 0; j < 5; j++) {
		const uint32_t s = lane + 64u * j;
		L[j] = s < nsym ? (uint32_t)S.lens[base + s] : 0u;
		code[j] = 0u;
	}
	const unsigned long long lt = (1ull << lane) - 1ull;
	uint32_t nxt = 0u, offs = 0u, n_codes = 0u;
	int32_t left = 1;
	for (uint32_t len = 1; len <= 15u; len++) {
		uint32_t cnt = 0u;
#pragma unroll
		for (int j = 0; j < 5; j++) {
			const unsigned long long b = __ballot(L[j] == len);
			if (L[j] == len) code[j] = nxt + cnt + (uint32_t)__popcll(b & lt);
			cnt += (uint32_t)__popcll(b);
		}
		left = left * 2 - (int32_t)cnt;
		if (left < 0) return 0u;
		if (lane == 0) {
			S.first[which][len] = nxt << (15u - len);
			S.lim[which][len] = (nxt + cnt) << (15u - len);
			S.off[which][len] = (uint16_t)offs;
		}
		// sorted[]: the symbols of this length in symbol order
#pragma unroll
		for (int j = 0; j < 5; j++)
			if (L[j] == len) IF_SORTED(S, which)[offs + (code[j] - nxt)] = (uint16_t)(lane + 64u * j);
		offs += cnt;
		n_codes += cnt;
		nxt = (nxt + cnt) << 1;
	}
	if (left > 0 && !(which == 1u && n_codes <= 1u)) return 0u;   // incomplete: only a distance code of one symbol (or none) is
#pragma unroll
	for (int j = 0; j < 5; j++) {
		const uint32_t s = lane + 64u * j, l = L[j];
		if (l == 0u) continue;
		const uint32_t rev = __brev(code[j]) >> (32u - l);
		if (l <= root) {
			const uint32_t e = (which ? if_d_entry(s) : if_ll_entry(s)) | l;
			for (uint32_t k = rev; k < tsize; k += 1u << l) tab[k] = e;
		} else {
			tab[rev & (tsize - 1u)] = IF_LONG << 4;
		}
	}
	return 1u;
}

// a code longer than the primary table's index: the canonical route.  Returns the entry (with its length), 0 if none.
__device__ __forceinline__ uint32_t if_long(IfShared &S, uint32_t which, uint32_t bits15) {
	const uint32_t c15 = __brev(bits15) >> 17;
	const uint32_t root = which ? IF_D_ROOT : IF_LL_ROOT;
	for (uint32_t len = root + 1u; len <= 15u; len++) {
		const uint32_t lim = IFU(S.lim[which][len]);
		if (c15 < lim) {
			const uint32_t first = IFU(S.first[which][len]);
			if (c15 < first) return 0u;
			const uint32_t sym = IFU(IF_SORTED(S, which)[IFU(S.off[which][len]) + ((c15 - first) >> (15u - len))]);
			const uint32_t e = which ? if_d_entry(sym) : if_ll_entry(sym);
			return e ? (e | len) : 0u;
		}
	}
	return 0u;
}

template <int DBG, bool VEC>
__global__ __launch_bounds__(64) void k_bgzf_inflate(const uint8_t *__restrict__ comp, size_t comp_len,
                                                     const msx_bgzf_block *__restrict__ blk, uint32_t n_blocks,
                                                     uint8_t *__restrict__ out, uint32_t *__restrict__ status,
                                                     uint32_t *__restrict__ ticket, uint32_t *__restrict__ stats) {
	__shared__ IfShared S;
	const uint32_t lane = threadIdx.x;
	// the launch holds as many waves as are to run at a time; each takes block after block
	for (;;) {
	uint32_t n_lit = 0u, n_match = 0u, n_far = 0u, n_dyn = 0u;      // (MSX_INFLATE_STATS: what the blocks are made of)
	uint32_t bi = 0u;
	if (lane == 0) bi = atomicAdd(ticket, 1u);
	bi = IFU(bi);
	if (bi >= n_blocks) return;
	const msx_bgzf_block B = blk[bi];
	uint8_t *og = out + B.out_off;
	const uint32_t out_len = B.out_len;
	if (out_len == 0u) {
		if (lane == 0) status[bi] = IF_OK;
		continue;
	}
	IfIn I;
	{
		const uintptr_t p = (uintptr_t)(comp + B.in_off);
		I.g = reinterpret_cast<const uint32_t *>(p & ~(uintptr_t)3);
		const size_t from = (size_t)((const uint8_t *)I.g - comp);
		const size_t avail = comp_len > from ? comp_len - from : 0u;
		I.n_bytes = avail > 0xfffffff0u ? 0xfffffff0u : (uint32_t)avail;
	}
	const uint32_t skew = (uint32_t)((uintptr_t)(comp + B.in_off) & 3u);
	const uint64_t end_bit = ((uint64_t)skew + B.in_len) * 8ull;
	const uint32_t rshift = (uint32_t)((uintptr_t)og & 15u);
	IfState T;
	T.pos = 0u; T.flushed = 0u; T.status = IF_OK;
	uint32_t w;
	if_seek(S, T, I, w, skew, lane);
#define IF_RI(q) (((q) + rshift) & IF_RMASK)
#define IF_FAIL(code) do { T.status = (code); goto done; } while (0)
	for (;;) {
		if_refill(S, T, I, w, lane);
		const uint32_t last = IF_PEEK(T, 1);
		const uint32_t type = (uint32_t)(T.buf >> 1) & 3u;
		IF_DROP(T, 3);
		if (IF_BITPOS(T) > end_bit) IF_FAIL(IF_IN_OVER);
		if (type == 3u) IF_FAIL(IF_BAD_TYPE);
		if (type == 0u) {
			// stored: to the next byte boundary, LEN, NLEN, bytes (through the ring: a later block may refer to them)
			IF_DROP(T, (uint32_t)T.cnt & 7u);
			if_refill(S, T, I, w, lane);
			const uint32_t len = IF_PEEK(T, 16);
			IF_DROP(T, 16);
			const uint32_t nlen = IF_PEEK(T, 16);
			IF_DROP(T, 16);
			const uint64_t bp = IF_BITPOS(T) >> 3;
			if ((len ^ 0xffffu) != nlen) IF_FAIL(IF_BAD_STORED);
			if ((bp + len) * 8ull > end_bit) IF_FAIL(IF_IN_OVER);
			if (len > out_len - T.pos) IF_FAIL(IF_OUT_OVER);
			const uint8_t *src = reinterpret_cast<const uint8_t *>(I.g) + bp;
			for (uint32_t done_b = 0; done_b < len;) {
				const uint32_t piece = len - done_b < IF_FLUSH ? len - done_b : IF_FLUSH;
				for (uint32_t i = lane; i < piece; i += 64u) S.ring[IF_RI(T.pos + i)] = src[done_b + i];
				T.pos += piece;
				done_b += piece;
				if (T.pos - T.flushed >= IF_FLUSH) {
					const uint32_t upto = (uint32_t)((((uintptr_t)og + T.pos) & ~(uintptr_t)15) - (uintptr_t)og);
					if_flush(S, T, og, upto, lane);
				}
			}
			if_seek(S, T, I, w, (uint32_t)(bp + len), lane);
			if (last) break;
			continue;
		}
		uint32_t hlit, hdist;
		if (type == 1u) {
			hlit = 288u; hdist = 32u;
			for (uint32_t s = lane; s < 320u; s += 64u)
				S.lens[s] = (uint8_t)(s < 144u ? 8u : s < 256u ? 9u : s < 280u ? 7u : s < 288u ? 8u : 5u);
		} else {
			hlit = IF_PEEK(T, 5) + 257u; IF_DROP(T, 5);
			hdist = IF_PEEK(T, 5) + 1u; IF_DROP(T, 5);
			const uint32_t hclen = IF_PEEK(T, 4) + 4u; IF_DROP(T, 4);
			if (hlit > 286u || hdist > 30u) IF_FAIL(IF_BAD_LENS);
			if (lane < 19u) S.pl[lane] = 0;
			for (uint32_t i = 0; i < hclen; i++) {
				if_refill(S, T, I, w, lane);
				const uint32_t o = (uint32_t)((i < 12u ? IF_ORDER_LO >> (5u * i) : IF_ORDER_HI >> (5u * (i - 12u))) & 31ull);
				if (lane == 0) S.pl[o] = (uint8_t)IF_PEEK(T, 3);
				IF_DROP(T, 3);
			}
			if (IF_BITPOS(T) > end_bit) IF_FAIL(IF_IN_OVER);
			// the code-length code into a direct table of 7 bits
			{
				const uint32_t Lp = lane < 19u ? (uint32_t)S.pl[lane] : 0u;
				uint32_t code = 0u, nxt = 0u;
				int32_t left = 1;
				const unsigned long long lt = (1ull << lane) - 1ull;
				for (uint32_t len = 1; len <= 7u; len++) {
					const unsigned long long b = __ballot(Lp == len);
					const uint32_t cnt = (uint32_t)__popcll(b);
					if (Lp == len) code = nxt + (uint32_t)__popcll(b & lt);
					left = left * 2 - (int32_t)cnt;
					nxt = (nxt + cnt) << 1;
					if (left < 0) break;
				}
				if (left != 0) IF_FAIL(IF_BAD_LENS);
				S.pre[lane] = 0u; S.pre[lane + 64u] = 0u;
				if (Lp) {
					const uint32_t rev = __brev(code) >> (32u - Lp);
					for (uint32_t k = rev; k < 128u; k += 1u << Lp) S.pre[k] = Lp | (lane << 8) | 0x10000u;
				}
			}
			const uint32_t total = hlit + hdist;
			uint32_t n = 0u, prev = 0u;
			while (n < total) {
				if_refill(S, T, I, w, lane);
				const uint32_t e = IFU(S.pre[IF_PEEK(T, 7)]);
				if (!e) IF_FAIL(IF_BAD_CODE);
				IF_DROP(T, e & 0xffu);
				const uint32_t sym = (e >> 8) & 0xffu;
				if (sym < 16u) {
					if (lane == 0) S.lens[n] = (uint8_t)sym;
					prev = sym;
					n++;
				} else {
					uint32_t rep, v = 0u;
					if (sym == 16u) {
						if (n == 0u) IF_FAIL(IF_BAD_LENS);
						v = prev;
						rep = 3u + IF_PEEK(T, 2); IF_DROP(T, 2);
					} else if (sym == 17u) {
						rep = 3u + IF_PEEK(T, 3); IF_DROP(T, 3);
						prev = 0u;
					} else {
						rep = 11u + IF_PEEK(T, 7); IF_DROP(T, 7);
						prev = 0u;
					}
					if (n + rep > total) IF_FAIL(IF_BAD_LENS);
					for (uint32_t i = lane; i < rep; i += 64u) S.lens[n + i] = (uint8_t)v;
					n += rep;
				}
				if (IF_BITPOS(T) > end_bit) IF_FAIL(IF_IN_OVER);
			}
			if (S.lens[256] == 0) IF_FAIL(IF_BAD_LENS);       // a block must be able to end
		}
		if (DBG == 4) n_dyn++;
		if (!if_build(S, 0u, 0u, hlit, lane)) IF_FAIL(IF_BAD_LENS);
		if (!if_build(S, 1u, hlit, hdist, lane)) IF_FAIL(IF_BAD_LENS);
		// ---- the symbols of the block ----
		// The table entry of the NEXT symbol is asked for as soon as its bits are known -- before a literal is stored, before
		// a match is copied -- and waited for at the top of the loop.
		if_refill(S, T, I, w, lane);
		uint32_t ev = S.ll[IF_PEEK(T, IF_LL_ROOT)];
		for (;;) {
			// ---- the fast loop: symbols whose codes the primary tables hold, while neither a write-back nor a refill of
			// the input ring is due (a symbol takes two words at most).  Anything else -- a long literal/length code, the
			// end of the block, the output's end, a chunk boundary -- leaves it in front of the symbol for the general step
			// below; only errors leave it inside a symbol.
			if (VEC && T.cross > T.ip + 2u) {
				// ---- the same loop on the VECTOR unit (round 4).  The bit buffer, its count and the read position are wave-
				// uniform values, so the compiler keeps them in scalar registers and the scalar unit does the arithmetic: 55
				// scalar instructions per symbol, and a compute unit has ONE scalar unit for all its waves -- it saturates at
				// 12 of the 14 waves the LDS holds (profiles/round3/inflate_summary.json).  Here the same values live in vector
				// registers (laundered through v_mov so that no analysis moves them back): shifts, masks and the refill are
				// VALU instructions -- four SIMDs per compute unit issue them side by side -- the refill is a select instead of a
				// branch, and only what a branch or a copy loop needs crosses to the scalar side (one v_readfirstlane of the
				// table entry per symbol, two more for a match's length and distance).  The loop runs for as many symbols as
				// cannot reach the input ring's next refill (a symbol takes two words at most), so the read position needs no
				// test; a symbol the primary tables do not hold, or an error, puts the state back in front of the symbol and
				// leaves it to the scalar loops below.
#define IF_TO_V(dst, src) asm volatile("v_mov_b32 %0, %1" : "=v"(dst) : "s"(src))
#define IF_VREFILL() do { const bool take_ = vcnt <= 32; vb |= take_ ? (uint64_t)w << vcnt : 0ull; vcnt += take_ ? 32 : 0;     \
				                  vip += take_ ? 1u : 0u; w = S.in[vip & (IF_IN_DW - 1u)]; } while (0)
				uint32_t vlo, vhi, vip, vc_;
				IF_TO_V(vlo, (uint32_t)T.buf); IF_TO_V(vhi, (uint32_t)(T.buf >> 32)); IF_TO_V(vc_, (uint32_t)T.cnt); IF_TO_V(vip, T.ip);
				uint64_t vb = (uint64_t)vhi << 32 | vlo;
				int32_t vcnt = (int32_t)vc_;
				const uint32_t lim = T.flushed + IF_FLUSH < out_len ? T.flushed + IF_FLUSH : out_len;
				for (uint32_t budget = (T.cross - T.ip - 2u) / 2u; budget != 0u && T.pos < lim; budget--) {
					const uint32_t e = ev, es = IFU(e);
					if (es & (IF_LIT << 4)) {
						const uint32_t n = e & 15u;
						vb >>= n; vcnt -= (int32_t)n;
						IF_VREFILL();
						ev = S.ll[(uint32_t)vb & ((1u << IF_LL_ROOT) - 1u)];
						S.ring[IF_RI(T.pos)] = (uint8_t)(es >> 16);          // (every lane the same byte to the same place)
						T.pos++;
						if (DBG == 4) n_lit++;
						continue;
					}
					if (((es >> 4) & 15u) != IF_BASE) break;
					const uint64_t vb0 = vb;
					const int32_t vcnt0 = vcnt;
					const uint32_t vip0 = vip, w0 = w;
					uint32_t n = e & 15u;
					vb >>= n; vcnt -= (int32_t)n;
					const uint32_t xl = (e >> 8) & 15u;
					const uint32_t vlen = (e >> 16) + ((uint32_t)vb & ((1u << xl) - 1u));
					vb >>= xl; vcnt -= (int32_t)xl;
					IF_VREFILL();
					const uint32_t d = S.dt[(uint32_t)vb & ((1u << IF_D_ROOT) - 1u)];
					if (((IFU(d) >> 4) & 15u) != IF_BASE) { vb = vb0; vcnt = vcnt0; vip = vip0; w = w0; break; }
					n = d & 15u;
					vb >>= n; vcnt -= (int32_t)n;
					const uint32_t xd = (d >> 8) & 15u;
					const uint32_t vdist = (d >> 16) + ((uint32_t)vb & ((1u << xd) - 1u));
					vb >>= xd; vcnt -= (int32_t)xd;
					const uint32_t len = IFU(vlen), dist = IFU(vdist);
					if (dist > T.pos || len > out_len - T.pos) { vb = vb0; vcnt = vcnt0; vip = vip0; w = w0; break; }
					IF_VREFILL();
					ev = S.ll[(uint32_t)vb & ((1u << IF_LL_ROOT) - 1u)];
					const uint32_t from = T.pos - dist;
					if (dist <= IF_NEAR) {
						if (dist >= 64u || dist >= len) {
							for (uint32_t b = 0; b < len; b += 64u) {
								const uint32_t i = b + lane;
								if (i < len) S.ring[IF_RI(T.pos + i)] = S.ring[IF_RI(from + i)];
							}
						} else {
							const float rf = 1.0f / (float)dist;
							for (uint32_t b = 0; b < len; b += 64u) {
								const uint32_t i = b + lane;
								const uint32_t q = (uint32_t)(((float)i + 0.5f) * rf);
								if (i < len) S.ring[IF_RI(T.pos + i)] = S.ring[IF_RI(from + (i - q * dist))];
							}
						}
					} else {
						if (DBG == 4) n_far++;
						__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
						for (uint32_t b = 0; b < len; b += 64u) {
							const uint32_t i = b + lane;
							if (i < len) S.ring[IF_RI(T.pos + i)] = IF_FAR_LOAD(og + from + i);
						}
					}
					T.pos += len;
					if (DBG == 4) n_match++;
				}
				T.buf = (uint64_t)IFU((uint32_t)(vb >> 32)) << 32 | (uint64_t)IFU((uint32_t)vb);
				T.cnt = (int32_t)IFU((uint32_t)vcnt);
				T.ip = IFU(vip);
#undef IF_VREFILL
#undef IF_TO_V
			}
			{
				uint32_t err = 0u;
				for (;;) {
					const uint32_t lim = T.flushed + IF_FLUSH < out_len ? T.flushed + IF_FLUSH : out_len;
					if (T.pos >= lim || T.ip + 2u >= T.cross) break;
					const uint32_t e = IFU(ev);
					if (e & (IF_LIT << 4)) {
						IF_DROP(T, e & 15u);
						if (T.cnt <= 32) {
							T.buf |= (uint64_t)IFU(w) << T.cnt;
							T.cnt += 32;
							T.ip++;
							w = S.in[T.ip & (IF_IN_DW - 1u)];
						}
						ev = S.ll[IF_PEEK(T, IF_LL_ROOT)];
						if (lane == 0) S.ring[IF_RI(T.pos)] = (uint8_t)(e >> 16);
						T.pos++;
						if (DBG == 4) n_lit++;
						continue;
					}
					if (((e >> 4) & 15u) != IF_BASE) break;
					IF_DROP(T, e & 15u);
					const uint32_t xl = (e >> 8) & 15u;
					const uint32_t len = (e >> 16) + IF_PEEK(T, xl);
					IF_DROP(T, xl);
					if (T.cnt <= 32) {
						T.buf |= (uint64_t)IFU(w) << T.cnt;
						T.cnt += 32;
						T.ip++;
						w = S.in[T.ip & (IF_IN_DW - 1u)];
					}
					uint32_t d = IFU(S.dt[IF_PEEK(T, IF_D_ROOT)]);
					if (((d >> 4) & 15u) == IF_LONG) d = if_long(S, 1u, IF_PEEK(T, 15));
					if (((d >> 4) & 15u) != IF_BASE) { err = IF_BAD_CODE; break; }
					IF_DROP(T, d & 15u);
					const uint32_t xd = (d >> 8) & 15u;
					const uint32_t dist = (d >> 16) + IF_PEEK(T, xd);
					IF_DROP(T, xd);
					if (dist > T.pos) { err = IF_BAD_DIST; break; }
					if (len > out_len - T.pos) { err = IF_OUT_OVER; break; }
					if (T.cnt <= 32) {
						T.buf |= (uint64_t)IFU(w) << T.cnt;
						T.cnt += 32;
						T.ip++;
						w = S.in[T.ip & (IF_IN_DW - 1u)];
					}
					ev = S.ll[IF_PEEK(T, IF_LL_ROOT)];
					const uint32_t from = T.pos - dist;
					if (dist <= IF_NEAR) {
						if (dist >= 64u || dist >= len) {
							for (uint32_t b = 0; b < len; b += 64u) {
								const uint32_t i = b + lane;
								if (i < len) S.ring[IF_RI(T.pos + i)] = S.ring[IF_RI(from + i)];
							}
						} else {
							const float rf = 1.0f / (float)dist;
							for (uint32_t b = 0; b < len; b += 64u) {
								const uint32_t i = b + lane;
								const uint32_t q = (uint32_t)(((float)i + 0.5f) * rf);
								if (i < len) S.ring[IF_RI(T.pos + i)] = S.ring[IF_RI(from + (i - q * dist))];
							}
						}
					} else {
						if (DBG == 4) n_far++;
						__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
						for (uint32_t b = 0; b < len; b += 64u) {
							const uint32_t i = b + lane;
							if (i < len) S.ring[IF_RI(T.pos + i)] = IF_FAR_LOAD(og + from + i);
						}
					}
					T.pos += len;
					if (DBG == 4) n_match++;
				}
				if (err) IF_FAIL(err);
			}
			// ---- the general step: one symbol, whatever it takes ----
			uint32_t e = IFU(ev);
			if (((e >> 4) & 15u) == IF_LONG) e = if_long(S, 0u, IF_PEEK(T, 15));
			const uint32_t kind = (e >> 4) & 15u;
			if (kind == 0u) IF_FAIL(IF_BAD_CODE);
			IF_DROP(T, e & 15u);
			if (kind == IF_LIT) {
				if (T.pos >= out_len) IF_FAIL(IF_OUT_OVER);
				if_refill(S, T, I, w, lane);
				ev = S.ll[IF_PEEK(T, IF_LL_ROOT)];
				if (lane == 0) S.ring[IF_RI(T.pos)] = (uint8_t)(e >> 16);
				T.pos++;
				if (DBG == 4) n_lit++;
			} else if (kind == IF_EOB) {
				break;
			} else {
				const uint32_t xl = (e >> 8) & 15u;
				const uint32_t len = (e >> 16) + IF_PEEK(T, xl);
				IF_DROP(T, xl);
				if_refill(S, T, I, w, lane);
				uint32_t d = IFU(S.dt[IF_PEEK(T, IF_D_ROOT)]);
				if (((d >> 4) & 15u) == IF_LONG) d = if_long(S, 1u, IF_PEEK(T, 15));
				if (((d >> 4) & 15u) != IF_BASE) IF_FAIL(IF_BAD_CODE);
				IF_DROP(T, d & 15u);
				const uint32_t xd = (d >> 8) & 15u;
				const uint32_t dist = (d >> 16) + IF_PEEK(T, xd);
				IF_DROP(T, xd);
				if (dist > T.pos) IF_FAIL(IF_BAD_DIST);
				if (len > out_len - T.pos) IF_FAIL(IF_OUT_OVER);
				if_refill(S, T, I, w, lane);
				ev = S.ll[IF_PEEK(T, IF_LL_ROOT)];
				const uint32_t from = T.pos - dist;
				if (dist <= IF_NEAR) {
					if (dist >= 64u || dist >= len) {
						// a round's sources lie in front of the round: earlier rounds (LDS keeps a wave's order) or earlier symbols
						for (uint32_t b = 0; b < len; b += 64u) {
							const uint32_t i = b + lane;
							if (i < len) S.ring[IF_RI(T.pos + i)] = S.ring[IF_RI(from + i)];
						}
					} else {
						// the match overlaps itself: byte i repeats byte i mod dist
						const float rf = 1.0f / (float)dist;
						for (uint32_t b = 0; b < len; b += 64u) {
							const uint32_t i = b + lane;
							const uint32_t q = (uint32_t)(((float)i + 0.5f) * rf);
							if (i < len) S.ring[IF_RI(T.pos + i)] = S.ring[IF_RI(from + (i - q * dist))];
						}
					}
				} else {
					// out of the ring's reach: those bytes were written back at least IF_NEAR - IF_FLUSH - 258 bytes ago (the
					// wave reads its own stores: workgroup scope)
					if (DBG == 4) n_far++;
					__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
					for (uint32_t b = 0; b < len; b += 64u) {
						const uint32_t i = b + lane;
						if (i < len) S.ring[IF_RI(T.pos + i)] = IF_FAR_LOAD(og + from + i);
					}
				}
				T.pos += len;
				if (DBG == 4) n_match++;
			}
			if (IF_BITPOS(T) > end_bit) IF_FAIL(IF_IN_OVER);
			if (T.pos - T.flushed >= IF_FLUSH) {
				const uint32_t upto = (uint32_t)((((uintptr_t)og + T.pos) & ~(uintptr_t)15) - (uintptr_t)og);
				if_flush(S, T, og, upto, lane);
			}
		}
		if (last) break;
	}
	if (T.pos != out_len) T.status = IF_LEN_MISMATCH;
done:
	if (T.status == IF_OK) if_flush(S, T, og, T.pos, lane);
	if (lane == 0) status[bi] = T.status;
	if (DBG == 4 && stats && lane == 0) {
		atomicAdd(&stats[0], n_lit); atomicAdd(&stats[1], n_match); atomicAdd(&stats[2], n_far); atomicAdd(&stats[3], n_dyn);
	}
	}
#undef IF_RI
#undef IF_FAIL
}

// ---------------------------------------------------------------------------
// CRC-32 (the gzip polynomial, reflected) of every block's output
// ---------------------------------------------------------------------------
#include "msx_crc.h"

// One wave per block (msx_crc.h: crc_wave), compared with the trailer's value.
__global__ __launch_bounds__(64) void k_bgzf_crc(const msx_bgzf_block *__restrict__ blk, uint32_t n_blocks,
                                                 const uint8_t *__restrict__ out, uint32_t *__restrict__ status,
                                                 uint32_t *__restrict__ n_bad) {
	__shared__ uint32_t tab[256];
	const uint32_t lane = threadIdx.x, bi = blockIdx.x;
	if (bi >= n_blocks) return;
	const uint32_t st = status[bi];
	if (st != IF_OK) {
		if (lane == 0) atomicAdd(n_bad, 1u);
		return;
	}
	crc_table_fill(tab, lane);
	__syncthreads();
	const msx_bgzf_block B = blk[bi];
	const uint32_t n = B.out_len;
	if (n == 0u) {
		if (lane == 0 && B.crc32 != 0u) { status[bi] = IF_BAD_CRC; atomicAdd(n_bad, 1u); }
		return;
	}
	const uint32_t crc = crc_wave(out + B.out_off, n, tab, lane);
	if (lane == 0 && crc != B.crc32) { status[bi] = IF_BAD_CRC; atomicAdd(n_bad, 1u); }
}

__global__ void k_bgzf_refuse(uint32_t n_blocks, uint32_t every, uint32_t *__restrict__ status, uint32_t *__restrict__ n_bad) {
	for (uint32_t i = threadIdx.x * every; i < n_blocks; i += 64u * every)
		if (status[i] == IF_OK) { status[i] = IF_BAD_CODE; atomicAdd(n_bad, 1u); }
}

// ---------------------------------------------------------------------------
// ABI
// ---------------------------------------------------------------------------
#define IF_PER_CU 14                      // waves per compute unit: what its LDS holds
static uint32_t *if_stats = nullptr;      // MSX_INFLATE_STATS (msx_bgzf_inflate only): device counters

int msx_bgzf_inflate_launch(msx_ctx *ctx, hipStream_t stream, int waves_per_cu, const uint8_t *d_comp, size_t comp_len,
                            const msx_bgzf_block *d_blocks, int64_t n_blocks, uint8_t *d_out, uint32_t *d_status, uint32_t *d_n_bad) {
	if (n_blocks <= 0) return MSX_OK;
	static int per_cu_env = -1;
	if (per_cu_env < 0) per_cu_env = getenv("MSX_INFLATE_WAVES") ? atoi(getenv("MSX_INFLATE_WAVES")) : 0;
	int per_cu = per_cu_env > 0 ? per_cu_env : waves_per_cu > 0 ? waves_per_cu : IF_PER_CU;
	if (per_cu > IF_PER_CU) per_cu = IF_PER_CU;
	int64_t grid = (int64_t)per_cu * ctx->num_cu;
	if (grid > n_blocks) grid = n_blocks;
	MSX_HIP(ctx, hipMemsetAsync(d_n_bad + 1, 0, 4, stream));      // the ticket
	// MSX_INFLATE_VEC=1: the decode loop's bit buffer on the vector unit (round 4's experiment: measured SLOWER, 45.1 against
	// 50.1 GB/s on lean records, 101 against 108 with SEQ/QUAL -- DESIGN.md section 3; kept as an A/B switch, off by default)
	static int vec = -1;
	if (vec < 0) vec = getenv("MSX_INFLATE_VEC") ? atoi(getenv("MSX_INFLATE_VEC")) != 0 : 0;
#define IF_LAUNCH(D, V) hipLaunchKernelGGL((k_bgzf_inflate<D, V>), dim3((unsigned)grid), dim3(64), 0, stream, d_comp, comp_len, d_blocks, \
	                   (uint32_t)n_blocks, d_out, d_status, d_n_bad + 1, if_stats)
	if (if_stats) { if (vec) IF_LAUNCH(4, true); else IF_LAUNCH(4, false); }      // (4: the same kernel counting its symbols, MSX_INFLATE_STATS)
	else { if (vec) IF_LAUNCH(0, true); else IF_LAUNCH(0, false); }
	hipLaunchKernelGGL(k_bgzf_crc, dim3((unsigned)n_blocks), dim3(64), 0, stream, d_blocks, (uint32_t)n_blocks,
	                   (const uint8_t *)d_out, d_status, d_n_bad);
	// MSX_INFLATE_REFUSE=<n> (tests): every n-th block is reported as refused, whatever the decoder made of it
	static int refuse = -1;
	if (refuse < 0) refuse = getenv("MSX_INFLATE_REFUSE") ? atoi(getenv("MSX_INFLATE_REFUSE")) : 0;
	if (refuse > 0) hipLaunchKernelGGL(k_bgzf_refuse, dim3(1), dim3(64), 0, stream, (uint32_t)n_blocks, (uint32_t)refuse, d_status, d_n_bad);
	MSX_HIP(ctx, hipGetLastError());
	return MSX_OK;
}

extern "C" int msx_bgzf_inflate(msx_ctx *ctx, const void *d_comp, size_t comp_len, const msx_bgzf_block *d_blocks,
                                int64_t n_blocks, void *d_out, uint32_t *d_status, int64_t *n_refused) {
	if (!ctx || (n_blocks > 0 && (!d_comp || !d_blocks || !d_out || !d_status))) return MSX_ERR_ARG;
	if (n_blocks < 0 || n_blocks > 0x7fffffff) return msx_fail(ctx, MSX_ERR_ARG, "msx_bgzf_inflate: block count");
	msx_join(ctx);
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	int rc;
	if ((rc = msx_reserve(ctx, &ctx->scan_l3, 64))) return rc;
	uint32_t *d_bad = (uint32_t *)ctx->scan_l3.p;
	MSX_HIP(ctx, hipMemsetAsync(d_bad, 0, 4, ctx->stream));
	const bool want_stats = getenv("MSX_INFLATE_STATS") != nullptr;
	if (want_stats) { if_stats = d_bad + 4; MSX_HIP(ctx, hipMemsetAsync(if_stats, 0, 16, ctx->stream)); }
	rc = msx_bgzf_inflate_launch(ctx, ctx->stream, 0, (const uint8_t *)d_comp, comp_len, d_blocks, n_blocks, (uint8_t *)d_out, d_status, d_bad);
	if_stats = nullptr;
	if (rc) return rc;
	if (want_stats) {
		uint32_t h[4];
		MSX_HIP(ctx, hipMemcpyAsync(h, d_bad + 4, 16, hipMemcpyDeviceToHost, ctx->stream));
		MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		fprintf(stderr, "# inflate: %lld blocks: %u literals, %u matches (%u beyond the ring), %u coded deflate blocks\n",
		        (long long)n_blocks, h[0], h[1], h[2], h[3]);
	}
	uint32_t bad = 0;
	MSX_HIP(ctx, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
	MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (n_refused) *n_refused = bad;
	return MSX_OK;
}

// msx_runtime_warmup: this translation unit's code object loaded onto the device ahead of its first launch (the runtime loads a
// module when one of its kernels is first asked for: 2-10 ms each, otherwise paid by the first batches of a command)
void msx_touch_inflate(void) {
	hipFuncAttributes attr;
	(void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_bgzf_crc));
}
