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
// a literal/length entry in 16 bits (the lane-parallel kernel's sorted[]: what a long code decodes to, ready to use):
// payload (9 bits) | extra bits << 9 | kind << 12
__device__ __forceinline__ uint32_t if_pack16(uint32_t e) { return (e >> 16) | (((e >> 8) & 7u) << 9) | (((e >> 4) & 15u) << 12); }
__device__ __forceinline__ uint32_t if_unpack16(uint32_t p) { return ((p & 0x1ffu) << 16) | (((p >> 9) & 7u) << 8) | ((p >> 12) << 4); }

template <bool PACK = false, class SH>
__device__ __forceinline__ uint32_t if_build(SH &S, uint32_t which, uint32_t base, uint32_t nsym, uint32_t lane) {
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
			if (L[j] == len) IF_SORTED(S, which)[offs + (code[j] - nxt)] = (uint16_t)((PACK && which == 0u) ? if_pack16(if_ll_entry(lane + 64u * j)) : lane + 64u * j);
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
template <class SH>
__device__ __forceinline__ uint32_t if_long(SH &S, uint32_t which, uint32_t bits15) {
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

template <int DBG>
__global__ __launch_bounds__(64) void k_bgzf_inflate(const uint8_t *__restrict__ comp, size_t comp_len,
                                                     const msx_bgzf_block *__restrict__ blk, uint32_t n_blocks,
                                                     uint8_t *__restrict__ out, uint32_t *__restrict__ status,
                                                     uint32_t *__restrict__ ticket, uint32_t *__restrict__ stats,
                                                     const uint32_t *__restrict__ list, const uint32_t *__restrict__ list_n) {
	__shared__ IfShared S;
	const uint32_t lane = threadIdx.x;
	// list: only the blocks k_bgzf_inflate_wave handed back (list_n of them, written by that kernel)
	if (list) n_blocks = IFU(*list_n);
	// the launch holds as many waves as are to run at a time; each takes block after block
	for (;;) {
	uint32_t n_lit = 0u, n_match = 0u, n_far = 0u, n_dyn = 0u;      // (MSX_INFLATE_STATS: what the blocks are made of)
	uint32_t bi = 0u;
	if (lane == 0) bi = atomicAdd(ticket, 1u);
	bi = IFU(bi);
	if (bi >= n_blocks) return;
	if (list) bi = IFU(list[bi]);
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
// The lane-parallel inflater (round 6)
// ---------------------------------------------------------------------------
// One wave per block, one symbol after the other, is a latency chain on the compute unit's ONE scalar unit (55 scalar
// instructions per symbol, 14 waves sharing the unit: 50 GB/s).  But Huffman decoding self-synchronises: a decoder started at a
// wrong bit decodes garbage for a few symbols and then falls onto a true code boundary, from where it IS the true chain.  So a
// deflate block's symbols are decoded by the 64 lanes of its wave at once (the algorithm, its passes and what it refuses are
// restated one lane after the other in msx_inflate_par_model.h, which the CPU tests run against zlib):
//   segment   3 KB of the stream staged in LDS (skewed by one word per 32 so that lanes a whole number of words apart do not
//             share a bank); lane L owns the tokens that begin in bits [seg + L * sub, seg + (L + 1) * sub), sub = 384 (measured:
//             256 / 320 / 384 / 448 / 512 / 640 bits give 100 / 119 / 122 / 116 / 117 / 116 GB/s on lean records and 183 / 187 / 199 /
//             182 / 180 / 163 with SEQ/QUAL -- fewer rounds of pass B against fewer waves per compute unit and idler last segments);
//   pass A    every lane walks its range from its first bit (lane 0: from the true position), counting bytes and pieces;
//   pass B    rounds: a lane whose left neighbour ended elsewhere than it started walks again from there (ends by shuffle,
//             votes by ballot); lanes behind the first one that stopped (end of block, no code, payload's end) sit out.  No
//             change: converged.  Eight rounds without convergence (long tokens, few per lane: a BAM header's text): four
//             times the bits per lane, again (384, 1536); beyond 4096 bits per lane the block is handed back;
//   pass C    exclusive sums of the counts place every lane in the output; the lanes walk a last time, literals go straight
//             to their bytes of the output IN GLOBAL MEMORY, matches -- in pieces of at most 16 bytes: position, length,
//             source, 8 bytes -- to a list in global memory;
//   resolve   64 pieces at a time, one per lane: a piece whose source reaches into the outputs of earlier pieces of the
//             window waits until exactly those are done (two searches over the window's positions by shuffle, done bits in a
//             register pair), everything else copies at once -- one 16-byte load, 16 / 8 / 4 / 2 / 1-byte stores; a
//             self-overlapping match reads byte i mod distance: independent of its own stores -- and a level's stores are
//             waited for before the next level's loads (the wave reads its own stores through its compute unit's L1, as the
//             serial kernel's far matches do);
//   then the next segment, or the next deflate block's header (wave-uniform values, as the serial kernel parses it).
// A first form gave a block four waves and kept its WHOLE output in LDS (80 KB: two blocks per compute unit, two waves per
// SIMD).  It was correct and no faster than the serial kernel (47.8 against 50.2 GB/s): per 8192 lean blocks it issued 2.14 G
// vector, 1.72 G scalar and 0.20 G LDS wave-instructions -- 4.1 + 3.3 + ~2 ms of its units' time if nothing overlapped -- and
// with two waves per SIMD hardly anything did (profiles/round6/inflate_lanes.md).  This form takes 9.8 KB of LDS and 127
// registers per block: sixteen blocks per compute unit, one block's table reads beside another's arithmetic, no barrier
// anywhere.  122 GB/s on lean records, 198 GB/s with SEQ/QUAL (the serial kernel: 50 / 108), every block equal to zlib's.
// Whatever is wrong on the true chain hands the block to the serial kernel (IF_RETRY: a list of block numbers, a second
// launch over it), whose verdict stands as before.
#ifndef IP_SUB0
#define IP_SUB0 384u
#endif
#define IP_MAX_ROUNDS 8u
#define IP_SKEW(d) ((d) + ((d) >> 5))
#define IP_PIECE 16u                     // a match goes to the list in pieces of at most 16 bytes
#define IP_MATCH_CAP 8192u               // pieces of one segment the list holds (24 576 bits could hold 12 288 two-bit matches and a block's 64 KB
                                         // 4096 further pieces: a segment with more than the list holds goes to the serial kernel)
#define IP_HANDBACK_SUB 4096u            // a segment that needs this many bits per lane to converge goes to the serial kernel.  (Not 1024: four
                                         // in a thousand record blocks restart once, and a block handed back costs the launch a serial kernel's
                                         // tail -- 32 hand-backs of 8192 blocks: 100 -> 69 GB/s)
#define IF_RETRY 10u

enum { IP_OK = 0u, IP_EOB = 1u, IP_BAD = 2u, IP_PAST = 3u, IP_DEAD = 4u };

// codes longer than the primary tables' index, without a loop: the per-length limits (left-aligned to 15 bits, ascending)
// and K[len] = off[len] - (first[len] >> (15 - len)), wave-uniform, read once per deflate block.  A code c (15 bits,
// left-aligned) has the first length whose limit exceeds it, and its symbol is sorted[K[len] + (c >> (15 - len))].
struct IpLong { uint32_t lim0[5]; uint32_t lim1[7]; };

struct IpLane { uint32_t end, nb, nm, st, trips; };

// a distance symbol's entry, without a branch (if_d_entry)
__device__ __forceinline__ uint32_t ip_d_entry(uint32_t s) {
	const uint32_t h = s >> 1, ex = (h > 1u ? h : 1u) - 1u;
	const uint32_t b = s < 2u ? 1u + s : 1u + ((2u + (s & 1u)) << ex);
	return s > 29u ? 0u : ((IF_BASE << 4) | (ex << 8) | (b << 16));
}

// A lane's walk (msx_inflate_par_model.h: ip_walk): tokens from bit `start` while they begin in front of `limit`.  Bit
// positions count from the block's aligned base; seg[0] is word win_dw0 of it.  EMIT: pass C.
// What an iteration costs is its chain of dependent LDS reads and the instructions around them (the first version --
// refill, literal/length table, the long route's loop, refill, distance table, every step its own branch -- took 2300
// clocks per token; the second, one table read per iteration but a branch per case, compiled to 450 instructions and 44
// branches per iteration and was no faster).  So: the stream's next word travels in a register, asked for an iteration
// before it is needed; an iteration reads ONE table entry -- a literal/length code (mode 0) or a distance code (mode 1) --
// and consumes it with the same arithmetic either way (code length, extra bits, base + extra); a code longer than the
// primary table's index finds its place in sorted[] from register-held limits (mode | 2) and reads its entry there in the
// next iteration; both of those rare paths sit behind wave-uniform branches.
template <bool EMIT, class SH, class LIT>
__device__ __forceinline__ IpLane ip_walk(SH &S, const IpLong &Q, uint32_t start, uint32_t limit, uint32_t win_dw0, uint32_t end_bit,
                                          uint32_t base, LIT lit, uint2 *__restrict__ ml, uint32_t &bad_dist) {
	uint32_t ip = (start >> 5) - win_dw0;
	uint64_t buf = (uint64_t)S.seg[IP_SKEW(ip)] | (uint64_t)S.seg[IP_SKEW(ip + 1u)] << 32;
	ip += 2u;
	uint32_t w = S.seg[IP_SKEW(ip)];
	buf >>= (start & 31u);
	int32_t cnt = 64 - (int32_t)(start & 31u);
	IpLane r;
	r.st = IP_OK; r.nb = 0u; r.nm = 0u; r.trips = 0u;
	uint32_t mode = 0u, mlen = 0u, sidx = 0u, slen = 0u;
	const uint32_t *tab = S.ll;            // (dt follows ll in the shared struct: one array of 1024 + 256 entries)
	const uint32_t bit0 = win_dw0 * 32u;
	// (written flat: every `if` around a few instructions costs a save / restore of the execution mask and a branch -- the
	//  nested form of this loop compiled to 146 scalar instructions per trip beside 135 vector ones; the cases are selects,
	//  the only branches are the two ways out, the two rare long-code paths and, in pass C, the stores)
	for (;;) {
		r.trips++;
		{
			const uint32_t at = bit0 + ip * 32u - (uint32_t)cnt;
			const bool tok = mode == 0u;
			const bool s_lim = tok && at >= limit, s_end = tok && at >= end_bit;
			if (s_lim || s_end) { if (!s_lim) r.st = IP_PAST; break; }
		}
		{
			const bool take = cnt <= 32;
			const uint64_t in = (uint64_t)w << ((uint32_t)cnt & 63u);
			buf |= take ? in : 0ull;
			cnt += take ? 32 : 0;
			ip += take ? 1u : 0u;
			w = S.seg[IP_SKEW(ip)];
		}
		const uint32_t bits = (uint32_t)buf;
		const bool pend = mode >= 2u;
		const uint32_t m = mode & 1u;
		uint32_t e = tab[m ? (1u << IF_LL_ROOT) + (bits & ((1u << IF_D_ROOT) - 1u)) : (bits & ((1u << IF_LL_ROOT) - 1u))];
		if (__ballot(pend)) {
			// (a long code's entry: what sorted[] holds at the place computed a trip ago)
			const uint32_t sv = IF_SORTED(S, m)[sidx];
			const uint32_t el = m ? ip_d_entry(sv) : if_unpack16(sv);
			const uint32_t eo = ((el >> 4) & 15u) ? (el | slen) : 0u;
			e = pend ? eo : e;
		}
		const bool is_long = !pend && ((e >> 4) & 15u) == IF_LONG;
		bool bad = false;
		if (__ballot(is_long)) {
			// the code's length from the register-held limits, its place in sorted[] from the per-length constants in LDS
			const uint32_t c15 = __brev(bits & 0x7fffu) >> 17;
			const uint32_t n0 = (uint32_t)(c15 >= Q.lim0[0]) + (uint32_t)(c15 >= Q.lim0[1]) + (uint32_t)(c15 >= Q.lim0[2]) + (uint32_t)(c15 >= Q.lim0[3]);
			const uint32_t n1 = (uint32_t)(c15 >= Q.lim1[0]) + (uint32_t)(c15 >= Q.lim1[1]) + (uint32_t)(c15 >= Q.lim1[2]) + (uint32_t)(c15 >= Q.lim1[3]) +
			                    (uint32_t)(c15 >= Q.lim1[4]) + (uint32_t)(c15 >= Q.lim1[5]);
			const uint32_t n = m ? n1 : n0;
			const int32_t k = S.qk[m][n];
			const uint32_t sl = (m ? IF_D_ROOT + 1u : IF_LL_ROOT + 1u) + n;
			bad = is_long && c15 >= (m ? Q.lim1[6] : Q.lim0[4]);
			slen = is_long ? sl : slen;
			sidx = is_long ? (uint32_t)(k + (int32_t)(c15 >> (15u - sl))) : sidx;
		}
		// the entry consumed (a long code consumes nothing yet): code, extra bits, value
		const uint32_t kind = (e >> 4) & 15u;
		const uint32_t n1_ = is_long ? 0u : (e & 15u), x = is_long ? 0u : ((e >> 8) & 15u);
		buf >>= n1_;
		const uint32_t val = (e >> 16) + ((uint32_t)buf & ((1u << x) - 1u));
		buf >>= x;
		cnt -= (int32_t)(n1_ + x);
		const bool ll = !is_long && m == 0u, dd = !is_long && m == 1u;
		const bool is_lit = ll && kind == IF_LIT, is_len = ll && kind == IF_BASE, is_eob = ll && kind == IF_EOB;
		const bool is_dist = dd && kind == IF_BASE;
		bad = bad || (ll && !(is_lit || is_len || is_eob)) || (dd && !is_dist);
		if (EMIT) {
			if (is_lit) lit[base + r.nb] = (uint8_t)val;
			if (is_dist) {
				// the match as pieces of at most 16 bytes: {position | length - 1 << 16 | phase << 20, source | period << 16}.  A piece
				// of a match that does not overlap itself reads `length` bytes from `source`; one that does (distance < length) reads
				// byte (phase + i) mod period from the period in front of the match -- never what the match itself writes.
				const uint32_t p = base + r.nb, dist = val;
				if (dist > p) bad_dist = 1u;
				uint32_t k = 0u;
				if (dist >= mlen) {
					for (uint32_t o = 0u; o < mlen; o += IP_PIECE, k++) {
						const uint32_t pl = mlen - o < IP_PIECE ? mlen - o : IP_PIECE;
						ml[r.nm + k] = make_uint2((p + o) | ((pl - 1u) << 16), (p + o - dist) & 0xffffu);
					}
				} else {
					for (uint32_t o = 0u; o < mlen; o += IP_PIECE, k++) {
						const uint32_t pl = mlen - o < IP_PIECE ? mlen - o : IP_PIECE;
						ml[r.nm + k] = make_uint2((p + o) | ((pl - 1u) << 16) | ((o % dist) << 20), ((p - dist) & 0xffffu) | (dist << 16));
					}
				}
			}
		}
		r.nb += is_lit ? 1u : is_dist ? mlen : 0u;
		r.nm += is_dist ? (mlen + IP_PIECE - 1u) / IP_PIECE : 0u;
		mlen = is_len ? val : mlen;
		mode = is_long ? (m | 2u) : is_len ? 1u : 0u;
		const bool past = (is_lit || is_dist) && (bit0 + ip * 32u - (uint32_t)cnt) > end_bit;
		if (is_eob || bad || past) { r.st = is_eob ? IP_EOB : bad ? IP_BAD : IP_PAST; break; }
	}
	r.end = bit0 + ip * 32u - (uint32_t)cnt;
	return r;
}

// the next seg_dw + 8 words of the stream from word w0 into seg[] (what lies behind the readable bytes: zeros)
template <class SH>
__device__ __forceinline__ void ip_stage(SH &S, const IfIn &I, uint32_t w0, uint32_t tid, uint32_t seg_dw, uint32_t threads) {
	const uint8_t *p = reinterpret_cast<const uint8_t *>(I.g);
	for (uint32_t k = tid; k < seg_dw + 8u; k += threads) {
		const uint32_t d = w0 + k;
		uint32_t v;
		if (4u * (uint64_t)d + 4u <= I.n_bytes) v = I.g[d];
		else v = if_edge_word(p, I.n_bytes, d);
		S.seg[IP_SKEW(k)] = v;
	}
}

__device__ __forceinline__ uint32_t ip_wave_incl(uint32_t v, uint32_t lane) {
#pragma unroll
	for (uint32_t d = 1u; d < 64u; d <<= 1) {
		const uint32_t o = (uint32_t)__shfl_up((int)v, d);
		if (lane >= d) v += o;
	}
	return v;
}

// One deflate block's header, parsed by ONE wave on wave-uniform values off the staged stream (seg[0] = word win_dw0 of the
// block's aligned base, `at` the header's first bit): stored / fixed / dynamic, the code lengths of a dynamic block into S.lens.
// Returns 0 or the IF_* code that refuses the block.
struct IpHeader { uint32_t type, last, hlit, hdist, stored_len, at; };
template <class SH>
__device__ __forceinline__ uint32_t ip_header(SH &S, uint32_t at, uint32_t win_dw0, uint32_t end_bit, uint32_t out_left, uint32_t lane, IpHeader &H) {
	// the stream's words in a register, 64 at a time (lane i: word hbase + i), fetched by v_readlane: a dynamic
	// header's ~150 code-length symbols were two LDS round trips each (stream word, table entry: 115 K clocks per
	// block); its table is held the same way (pre0 / pre1 below)
	uint32_t hip_ = 0u, hbase = 0u;
	uint32_t hw = S.seg[IP_SKEW(lane)];
	uint64_t hb = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hw, 0) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hw, 1) << 32;
	hip_ += 2u;
	hb >>= (at & 31u);
	int32_t hc = 64 - (int32_t)(at & 31u);
#define IH_AT() ((win_dw0 + hip_) * 32u - (uint32_t)hc)
#define IH_REFILL() do { if (hc <= 32) {                                                                     \
		if (hip_ - hbase >= 64u) { hbase += 64u; hw = S.seg[IP_SKEW(hbase + lane)]; }                    \
		hb |= (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hw, (int)(hip_ - hbase)) << hc; hc += 32; hip_++; } } while (0)
#define IH_PEEK(n) ((uint32_t)(hb & ((1ull << (n)) - 1ull)))
#define IH_DROP(n) do { hb >>= (n); hc -= (int32_t)(n); } while (0)
	uint32_t herr = 0u, hlit = 288u, hdist = 32u, stored_len = 0u;
	const uint32_t lastb = IH_PEEK(1);
	const uint32_t type = (uint32_t)(hb >> 1) & 3u;
	IH_DROP(3);
	do {
		if (IH_AT() > end_bit) { herr = IF_IN_OVER; break; }
		if (type == 3u) { herr = IF_BAD_TYPE; break; }
		if (type == 0u) {
			IH_DROP((uint32_t)hc & 7u);
			IH_REFILL();
			stored_len = IH_PEEK(16);
			IH_DROP(16);
			const uint32_t nlen = IH_PEEK(16);
			IH_DROP(16);
			if ((stored_len ^ 0xffffu) != nlen) { herr = IF_BAD_STORED; break; }
			if ((uint64_t)IH_AT() + 8ull * stored_len > end_bit) { herr = IF_IN_OVER; break; }
			if (stored_len > out_left) { herr = IF_OUT_OVER; break; }
			break;
		}
		if (type == 1u) {
			for (uint32_t s = lane; s < 320u; s += 64u)
				S.lens[s] = (uint8_t)(s < 144u ? 8u : s < 256u ? 9u : s < 280u ? 7u : s < 288u ? 8u : 5u);
			break;
		}
		IH_REFILL();
		hlit = IH_PEEK(5) + 257u; IH_DROP(5);
		hdist = IH_PEEK(5) + 1u; IH_DROP(5);
		const uint32_t hclen = IH_PEEK(4) + 4u; IH_DROP(4);
		if (hlit > 286u || hdist > 30u) { herr = IF_BAD_LENS; break; }
		if (lane < 19u) S.pl[lane] = 0;
		for (uint32_t i = 0; i < hclen; i++) {
			IH_REFILL();
			const uint32_t o = (uint32_t)((i < 12u ? IF_ORDER_LO >> (5u * i) : IF_ORDER_HI >> (5u * (i - 12u))) & 31ull);
			if (lane == 0) S.pl[o] = (uint8_t)IH_PEEK(3);
			IH_DROP(3);
		}
		if (IH_AT() > end_bit) { herr = IF_IN_OVER; break; }
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
			if (left != 0) { herr = IF_BAD_LENS; break; }
			S.pre[lane] = 0u; S.pre[lane + 64u] = 0u;
			if (Lp) {
				const uint32_t rev = __brev(code) >> (32u - Lp);
				for (uint32_t k = rev; k < 128u; k += 1u << Lp) S.pre[k] = Lp | (lane << 8) | 0x10000u;
			}
		}
		const uint32_t total = hlit + hdist;
		const uint32_t pre0 = S.pre[lane], pre1 = S.pre[lane + 64u];
		uint32_t n = 0u, prev = 0u;
		while (n < total) {
			IH_REFILL();
			const uint32_t pk = IH_PEEK(7);
			const uint32_t e = pk < 64u ? (uint32_t)__builtin_amdgcn_readlane((int)pre0, (int)pk)
			                            : (uint32_t)__builtin_amdgcn_readlane((int)pre1, (int)(pk - 64u));
			if (!e) { herr = IF_BAD_CODE; break; }
			IH_DROP(e & 0xffu);
			const uint32_t sym = (e >> 8) & 0xffu;
			if (sym < 16u) {
				if (lane == 0) S.lens[n] = (uint8_t)sym;
				prev = sym;
				n++;
			} else {
				uint32_t rep, v = 0u;
				if (sym == 16u) {
					if (n == 0u) { herr = IF_BAD_LENS; break; }
					v = prev;
					rep = 3u + IH_PEEK(2); IH_DROP(2);
				} else if (sym == 17u) {
					rep = 3u + IH_PEEK(3); IH_DROP(3);
					prev = 0u;
				} else {
					rep = 11u + IH_PEEK(7); IH_DROP(7);
					prev = 0u;
				}
				if (n + rep > total) { herr = IF_BAD_LENS; break; }
				for (uint32_t i = lane; i < rep; i += 64u) S.lens[n + i] = (uint8_t)v;
				n += rep;
			}
			if (IH_AT() > end_bit) { herr = IF_IN_OVER; break; }
		}
		if (!herr && S.lens[256] == 0) herr = IF_BAD_LENS;
	} while (0);
	H.type = type; H.last = lastb; H.hlit = hlit; H.hdist = hdist; H.stored_len = stored_len; H.at = IH_AT();
#undef IH_AT
#undef IH_REFILL
#undef IH_PEEK
#undef IH_DROP
	return herr;
}

// (PH: MSX_INFLATE_STATS=3 -- thread 0's clock at the phase boundaries, summed over the blocks into stats[8..])
#define IP_PH(k) do { if (PH) { const long long now_ = clock64(); if (tid == 0) ph[k] += (unsigned long long)(now_ - t_ph); t_ph = now_; } } while (0)
// the kernel: one wave per block (16 per compute unit: IW_PER_CU)
#ifndef IW_SEG_DW
#define IW_SEG_DW 768u
#endif
#define IW_SEG_BITS (IW_SEG_DW * 32u)
struct IwShared {
	uint32_t ll[1 << IF_LL_ROOT];
	uint32_t dt[1 << IF_D_ROOT];
	uint32_t seg[IP_SKEW(IW_SEG_DW + 8u) + 1u];
	uint32_t lim[2][16], first[2][16];
	uint16_t off[2][16];
	union {
		uint16_t sorted[320];
		uint32_t pre[128];
	};
	uint8_t lens[352];
	uint8_t pl[32];
	int32_t qk[2][8];                      // long codes: per length, off[len] - (first[len] >> (15 - len))
};

struct __attribute__((packed, aligned(1))) iw_u16 { unsigned long long lo, hi; };

template <bool PH>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_bgzf_inflate_wave(const uint8_t *__restrict__ comp, size_t comp_len,
                                                          const msx_bgzf_block *__restrict__ blk, uint32_t n_blocks,
                                                          uint8_t *__restrict__ out, uint32_t *__restrict__ status,
                                                          uint32_t *__restrict__ ticket, uint2 *__restrict__ match_scratch,
                                                          uint32_t *__restrict__ retry_list, uint32_t *__restrict__ retry_n,
                                                          unsigned long long *__restrict__ stats) {
	__shared__ IwShared S;
	const uint32_t lane = threadIdx.x, tid = lane;
	uint2 *const ml = match_scratch + (size_t)blockIdx.x * IP_MATCH_CAP;
	unsigned long long ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
	long long t_ph = PH ? clock64() : 0;
	for (;;) {
		uint32_t bi = 0u;
		if (lane == 0) bi = atomicAdd(ticket, 1u);
		bi = IFU(bi);
		if (bi >= n_blocks) {
			if (PH && tid == 0) for (int k = 0; k < 16; k++) atomicAdd(&stats[k], ph[k]);
			return;
		}
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
		const uint32_t end_bit = (skew + B.in_len) * 8u;
		uint32_t at = skew * 8u, pos = 0u, fail = 0u;
		bool last_block = false;
		if (out_len > 65536u || B.in_len > 0x100000u) fail = IF_OUT_OVER;
		IP_PH(0);
		while (!fail && !last_block) {
			uint32_t win_dw0 = at >> 5;
			ip_stage(S, I, win_dw0, lane, IW_SEG_DW, 64u);
			IP_PH(1);
			IpHeader H;
			fail = ip_header(S, at, win_dw0, end_bit, out_len - pos, lane, H);
			IP_PH(2);
			if (fail) break;
			last_block = H.last != 0u;
			at = H.at;
			if (H.type == 0u) {
				const uint8_t *src = reinterpret_cast<const uint8_t *>(I.g) + (at >> 3);
				for (uint32_t i = lane; i < H.stored_len; i += 64u) og[pos + i] = src[i];
				pos += H.stored_len;
				at += 8u * H.stored_len;
				continue;
			}
			if (!if_build<true>(S, 0u, 0u, H.hlit, lane) || !if_build<true>(S, 1u, H.hlit, H.hdist, lane)) { fail = IF_BAD_LENS; break; }
			IpLong Q;
#pragma unroll
			for (uint32_t k = 0; k < 5u; k++) Q.lim0[k] = IFU(S.lim[0][IF_LL_ROOT + 1u + k]);
#pragma unroll
			for (uint32_t k = 0; k < 7u; k++) Q.lim1[k] = IFU(S.lim[1][IF_D_ROOT + 1u + k]);
			if (tid < 16u) {
				const uint32_t wq = tid >> 3, kq = tid & 7u, len = (wq ? IF_D_ROOT : IF_LL_ROOT) + 1u + kq;
				S.qk[wq][kq] = len <= 15u ? (int32_t)S.off[wq][len] - (int32_t)(S.first[wq][len] >> (15u - len)) : 0;
			}
			IP_PH(3);
			uint32_t sub = IP_SUB0;
			for (;;) {
				const uint32_t seg0 = at;
				uint32_t nl, used = 0u, limit = 0u;
				IpLane r;
				uint32_t dummy = 0u;
				for (;;) {
					const uint32_t win_end = win_dw0 * 32u + IW_SEG_BITS;
					nl = (win_end - seg0 + sub - 1u) / sub;
					if (nl > 64u) nl = 64u;
					used = seg0 + lane * sub;
					limit = used + sub < win_end ? used + sub : win_end;
					r.st = IP_DEAD; r.end = 0u; r.nb = 0u; r.nm = 0u; r.trips = 0u;
					if (lane < nl && (lane == 0u || used < end_bit)) r = ip_walk<false>(S, Q, used, limit, win_dw0, end_bit, 0u, (uint8_t *)nullptr, nullptr, dummy);
					IP_PH(4);
					bool converged = false;
					for (uint32_t rounds = 0u; rounds < IP_MAX_ROUNDS || nl == 1u; rounds++) {
						const uint32_t mine = r.end | (r.st << 24);
						const unsigned long long bal = __ballot(lane < nl && r.st != IP_OK);
						const uint32_t stop = bal ? (uint32_t)__ffsll((unsigned long long)bal) - 1u : 0xffffu;
						const uint32_t left = (uint32_t)__shfl_up((int)mine, 1);
						bool ch = false;
						if (lane >= 1u && lane < nl && lane <= stop) {
							const uint32_t ns = left & 0xffffffu;
							if (ns != used) { used = ns; ch = true; }
						}
						if (!__ballot(ch)) { converged = true; break; }
						if (ch) r = ip_walk<false>(S, Q, used, limit, win_dw0, end_bit, 0u, (uint8_t *)nullptr, nullptr, dummy);
						if (PH && tid == 0) ph[10]++;
					}
					IP_PH(5);
					if (converged) break;
					sub *= 4u;
					if (sub >= IP_HANDBACK_SUB) { fail = IF_RETRY; break; }
				}
				if (fail) break;
				const unsigned long long bal = __ballot(lane < nl && r.st != IP_OK);
				const uint32_t stop = bal ? (uint32_t)__ffsll((unsigned long long)bal) - 1u : 0xffffu;
				const uint32_t lastl = stop < nl ? stop : nl - 1u;
				const uint32_t le = (uint32_t)__shfl((int)(r.end | (r.st << 24)), (int)lastl);
				const uint32_t st_last = le >> 24;
				if (st_last != IP_OK && st_last != IP_EOB) { fail = st_last == IP_PAST ? IF_IN_OVER : IF_BAD_CODE; break; }
				const uint32_t my_nb = lane <= lastl ? r.nb : 0u, my_nm = lane <= lastl ? r.nm : 0u;
				const uint32_t inb = ip_wave_incl(my_nb, lane), inm = ip_wave_incl(my_nm, lane);
				const uint32_t total = (uint32_t)__shfl((int)inb, 63), mtot = (uint32_t)__shfl((int)inm, 63);
				if (total > out_len - pos) { fail = IF_OUT_OVER; break; }
				if (mtot > IP_MATCH_CAP) { fail = IF_RETRY; break; }
				IP_PH(6);
				uint32_t bad_dist = 0u;
				if (lane <= lastl && r.st != IP_DEAD)
					(void)ip_walk<true>(S, Q, used, limit, win_dw0, end_bit, pos + inb - my_nb, og, ml + (inm - my_nm), bad_dist);
				if (__ballot(bad_dist != 0u)) { fail = IF_BAD_DIST; break; }
				// what pass C stored -- literals, the list -- is read below by other lanes of this wave
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
				IP_PH(7);
				uint2 m_next = make_uint2(0u, 0u);
				if (lane < mtot) m_next = ml[lane];
				for (uint32_t w0 = 0u; w0 < mtot; w0 += 64u) {
					const bool have = w0 + lane < mtot;
					const uint2 m = m_next;
					// (the next window's entries are on their way while this one's are searched and copied)
					m_next = make_uint2(0u, 0u);
					if (w0 + 64u + lane < mtot) m_next = ml[w0 + 64u + lane];
					const uint32_t p = m.x & 0xffffu, l = ((m.x >> 16) & 15u) + 1u, ph0 = m.x >> 20, from = m.y & 0xffffu, per = m.y >> 16;
					const uint32_t send = from + (per ? per : l);
					const uint32_t wp = have ? p : 0x7fffffffu, we = p + l;
					const uint32_t wp0 = (uint32_t)__shfl((int)wp, 0);
					uint32_t a0 = 0u, b = 0u;
					bool dep = false;
					// the earlier pieces of the window this one reads from, [a0, b]: two searches over the lanes' positions
					{
						const bool look = have && lane > 0u && send > wp0;
						uint32_t lo = 0u, hi = look ? lane : 0u, lo2 = 0u, hi2 = hi;
#pragma unroll
						for (int it = 0; it < 6; it++) {                 // (64 lanes: six halvings; every lane takes part in the shuffles)
							const uint32_t mid = (lo + hi) >> 1, mid2 = (lo2 + hi2) >> 1;
							const uint32_t v = (uint32_t)__shfl((int)wp, (int)(mid & 63u)), v2 = (uint32_t)__shfl((int)wp, (int)(mid2 & 63u));
							if (lo < hi) { if (v < send) lo = mid + 1u; else hi = mid; }
							if (lo2 < hi2) { if (v2 <= from) lo2 = mid2 + 1u; else hi2 = mid2; }
						}
						// (a seventh step for the range [0, 63]: lo < hi can still hold after six halvings of 63)
						{
							const uint32_t mid = (lo + hi) >> 1, mid2 = (lo2 + hi2) >> 1;
							const uint32_t v = (uint32_t)__shfl((int)wp, (int)(mid & 63u)), v2 = (uint32_t)__shfl((int)wp, (int)(mid2 & 63u));
							if (lo < hi) { if (v < send) lo = mid + 1u; else hi = mid; }
							if (lo2 < hi2) { if (v2 <= from) lo2 = mid2 + 1u; else hi2 = mid2; }
						}
						const uint32_t a_ = lo2 ? lo2 - 1u : 0u;
						const uint32_t ea = (uint32_t)__shfl((int)we, (int)(a_ & 63u));
						if (look) {
							b = lo - 1u;
							a0 = a_ + (ea <= from ? 1u : 0u);
							dep = a0 <= b;
						}
					}
					const unsigned long long need = dep ? ((b == 63u ? ~0ull : ((1ull << (b + 1u)) - 1ull)) & ~((1ull << a0) - 1ull)) : 0ull;
					bool done = !have;
					for (uint32_t level = 0u;; level++) {
						const unsigned long long dm = __ballot(done);
						if (dm == ~0ull) break;
						if (level > 64u) { fail = IF_RETRY; break; }              // (cannot happen: the lowest open piece is always ready)
						const bool ready = !done && (dm & need) == need;
						// the stores of the level before (of the window before) are in place before this level's loads: waited for
						// here and not behind the stores -- the searches above ran while they were on their way
						__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
						__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
						if (ready) {
							const uint8_t *src = og + from;
							uint8_t *dst = og + p;
							if (per == 0u) {
								// 16 bytes in one load wherever the piece ends (what lies behind it is not used; the output buffers
								// carry 64 bytes of slack), stored as 16 / 8 / 4 / 2 / 1 bytes: one cache-line visit per lane and
								// instruction instead of sixteen -- with a byte at a time a level cost 2 K cycles of the address path
								iw_u16 q;
								__builtin_memcpy(&q, src, 16);
								if (l == 16u) {
									__builtin_memcpy(dst, &q, 16);
								} else {
									unsigned long long lo = q.lo, hi = q.hi;
									uint32_t o = 0u;
									if (l & 8u) { __builtin_memcpy(dst, &lo, 8); lo = hi; o = 8u; }
									if (l & 4u) { const uint32_t w = (uint32_t)lo; __builtin_memcpy(dst + o, &w, 4); lo >>= 32; o += 4u; }
									if (l & 2u) { const uint16_t w = (uint16_t)lo; __builtin_memcpy(dst + o, &w, 2); lo >>= 16; o += 2u; }
									if (l & 1u) dst[o] = (uint8_t)lo;
								}
							} else {
								uint8_t v[IP_PIECE];
								const float rf = 1.0f / (float)per;
#pragma unroll
								for (uint32_t j = 0; j < IP_PIECE; j++) {
									const uint32_t x = ph0 + j;
									const uint32_t q = (uint32_t)(((float)x + 0.5f) * rf);
									v[j] = j < l ? src[x - q * per] : (uint8_t)0;
								}
#pragma unroll
								for (uint32_t j = 0; j < IP_PIECE; j++) if (j < l) dst[j] = v[j];
							}
							done = true;
						}
						if (PH && tid == 0) ph[11]++;
					}
					if (fail) break;
				}
				IP_PH(8);
				if (fail) break;
				pos += total;
				if ((le & 0xffffffu) <= at) { fail = IF_BAD_CODE; break; }
				at = le & 0xffffffu;
				if (st_last == IP_EOB) break;
				win_dw0 = at >> 5;
				ip_stage(S, I, win_dw0, lane, IW_SEG_DW, 64u);
			}
		}
		if (!fail && pos != out_len) fail = IF_LEN_MISMATCH;
		if (lane == 0) {
			if (!fail) status[bi] = IF_OK;
			else { status[bi] = IF_RETRY; retry_list[atomicAdd(retry_n, 1u)] = bi; }
		}
		IP_PH(9);
	}
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

#ifdef MSX_DEBUG_SWITCHES
__global__ void k_bgzf_refuse(uint32_t n_blocks, uint32_t every, uint32_t *__restrict__ status, uint32_t *__restrict__ n_bad) {
	for (uint32_t i = threadIdx.x * every; i < n_blocks; i += 64u * every)
		if (status[i] == IF_OK) { status[i] = IF_BAD_CODE; atomicAdd(n_bad, 1u); }
}
#endif

// ---------------------------------------------------------------------------
// ABI
// ---------------------------------------------------------------------------
#define IF_PER_CU 14                      // serial kernel: waves per compute unit, what its LDS holds
#define IW_PER_CU 16                      // lane-parallel kernel: waves per compute unit (8.7 KB of LDS and 127 VGPRs each: four per SIMD)
static uint32_t *if_stats = nullptr;      // MSX_INFLATE_STATS (msx_bgzf_inflate only): device counters
static unsigned long long *ip_phase_stats = nullptr;      // MSX_INFLATE_STATS=3: the lane-parallel kernel's clocks per phase

// d_n_bad: [0] the refused blocks (zeroed by the caller), [1] the lane-parallel launch's ticket, [2] the number of blocks it
// handed back, [3] the serial launch's ticket
int msx_bgzf_inflate_launch(msx_ctx *ctx, hipStream_t stream, int waves_per_cu, const uint8_t *d_comp, size_t comp_len,
                            const msx_bgzf_block *d_blocks, int64_t n_blocks, uint8_t *d_out, uint32_t *d_status, uint32_t *d_n_bad) {
	if (n_blocks <= 0) return MSX_OK;
	// The lane-parallel kernel for every block, the serial kernel for what it hands back.  MSX_INFLATE_SERIAL=1: the serial
	// kernel alone (rounds 3-5: one wave per block, one symbol after the other) -- the A/B of scripts/bench_inflate.py, and
	// what the hand-backs' path is tested with.  Read per launch: the tests switch it.
	const char *so_ = getenv("MSX_INFLATE_SERIAL");
	const int serial_only = so_ && atoi(so_) != 0;
	const char *we_ = getenv("MSX_INFLATE_WAVES");
	const int per_cu_env = we_ ? atoi(we_) : 0;
	int per_cu = per_cu_env > 0 ? per_cu_env : waves_per_cu > 0 ? waves_per_cu : IF_PER_CU;
	if (per_cu > IF_PER_CU) per_cu = IF_PER_CU;
	int64_t grid = (int64_t)per_cu * ctx->num_cu;
	if (grid > n_blocks) grid = n_blocks;
	MSX_HIP(ctx, hipMemsetAsync(d_n_bad + 1, 0, 12, stream));      // tickets and the hand-back count
#define IF_LAUNCH(D, L, LN) hipLaunchKernelGGL((k_bgzf_inflate<D>), dim3((unsigned)grid), dim3(64), 0, stream, d_comp, comp_len, d_blocks, \
	                   (uint32_t)n_blocks, d_out, d_status, d_n_bad + 3, if_stats, (const uint32_t *)(L), (const uint32_t *)(LN))
	if (serial_only || if_stats) {
		if (if_stats) IF_LAUNCH(4, nullptr, nullptr);                // (4: the same kernel counting its symbols, MSX_INFLATE_STATS=1)
		else IF_LAUNCH(0, nullptr, nullptr);
	} else {
		// the scratch set of this stream: the pieces' lists of the resident waves, the list of blocks handed back
		msx_ctx::inf_set *is = nullptr;
		for (auto &c : ctx->inf) if (c.used && c.stream == stream) is = &c;
		if (!is) for (auto &c : ctx->inf) if (!c.used) { is = &c; c.used = true; c.stream = stream; break; }
		if (!is) return msx_fail(ctx, MSX_ERR_ARG, "msx_bgzf_inflate_launch: more than four streams inflate on one context");
		const int per_cu_w = (per_cu_env > 0 && per_cu_env <= IW_PER_CU) ? per_cu_env : IW_PER_CU;
		int64_t wgrid = (int64_t)per_cu_w * ctx->num_cu;
		if (wgrid > n_blocks) wgrid = n_blocks;
		const size_t want_m = (size_t)IW_PER_CU * ctx->num_cu * IP_MATCH_CAP * sizeof(uint2);
		const size_t want_r = ((size_t)n_blocks + 64) * 4;
		if (is->matches.cap < want_m || is->retry.cap < want_r) {
			MSX_HIP(ctx, hipStreamSynchronize(stream));             // (an earlier launch of this stream may still read them)
			int rc;
			if ((rc = msx_reserve(ctx, &is->matches, want_m))) return rc;
			if ((rc = msx_reserve(ctx, &is->retry, want_r < (1u << 18) ? (1u << 18) : want_r))) return rc;
		}
		if (ip_phase_stats)
			hipLaunchKernelGGL(k_bgzf_inflate_wave<true>, dim3((unsigned)wgrid), dim3(64), 0, stream, d_comp, comp_len, d_blocks, (uint32_t)n_blocks,
			                   d_out, d_status, d_n_bad + 1, (uint2 *)is->matches.p, (uint32_t *)is->retry.p, d_n_bad + 2, ip_phase_stats);
		else
			hipLaunchKernelGGL(k_bgzf_inflate_wave<false>, dim3((unsigned)wgrid), dim3(64), 0, stream, d_comp, comp_len, d_blocks, (uint32_t)n_blocks,
			                   d_out, d_status, d_n_bad + 1, (uint2 *)is->matches.p, (uint32_t *)is->retry.p, d_n_bad + 2, (unsigned long long *)nullptr);
		// what it handed back: the serial kernel over that list (a launch that finds the list empty returns at once)
		if (grid > 4 * (int64_t)ctx->num_cu) grid = 4 * (int64_t)ctx->num_cu;
		IF_LAUNCH(0, is->retry.p, d_n_bad + 2);
	}
#undef IF_LAUNCH
	hipLaunchKernelGGL(k_bgzf_crc, dim3((unsigned)n_blocks), dim3(64), 0, stream, d_blocks, (uint32_t)n_blocks,
	                   (const uint8_t *)d_out, d_status, d_n_bad);
#ifdef MSX_DEBUG_SWITCHES
	// MSX_INFLATE_REFUSE=<n> (libmsamtools_amd_dbg.so only; tests): every n-th block is reported as refused, whatever the
	// decoder made of it
	static const int refuse = getenv("MSX_INFLATE_REFUSE") ? atoi(getenv("MSX_INFLATE_REFUSE")) : 0;
	if (refuse > 0) hipLaunchKernelGGL(k_bgzf_refuse, dim3(1), dim3(64), 0, stream, (uint32_t)n_blocks, (uint32_t)refuse, d_status, d_n_bad);
#endif
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
	const int stats_mode = getenv("MSX_INFLATE_STATS") ? atoi(getenv("MSX_INFLATE_STATS")) : 0;   // 1: symbol counts (serial kernel); 2: blocks handed back
	const bool want_stats = stats_mode == 1;
	if (want_stats) { if_stats = d_bad + 8; MSX_HIP(ctx, hipMemsetAsync(if_stats, 0, 16, ctx->stream)); }
	if (stats_mode == 3) {
		if ((rc = msx_reserve(ctx, &ctx->scan_l2, 256))) return rc;
		ip_phase_stats = (unsigned long long *)ctx->scan_l2.p;
		MSX_HIP(ctx, hipMemsetAsync(ip_phase_stats, 0, 128, ctx->stream));
	}
	rc = msx_bgzf_inflate_launch(ctx, ctx->stream, 0, (const uint8_t *)d_comp, comp_len, d_blocks, n_blocks, (uint8_t *)d_out, d_status, d_bad);
	if_stats = nullptr;
	if (rc) { ip_phase_stats = nullptr; return rc; }
	if (ip_phase_stats) {
		unsigned long long h[16];
		MSX_HIP(ctx, hipMemcpyAsync(h, ip_phase_stats, 128, hipMemcpyDeviceToHost, ctx->stream));
		MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		ip_phase_stats = nullptr;
		static const char *names[10] = {"ticket", "stage", "header", "tables", "pass A", "pass B", "sums", "pass C", "resolve", "write-back"};
		fprintf(stderr, "# inflate phases, clocks per block (thread 0):");
		for (int k = 0; k < 10; k++) fprintf(stderr, " %s %.0f", names[k], (double)h[k] / (double)n_blocks);
		fprintf(stderr, "; pass-B rounds %.2f, resolve windows %.2f per block; pass A of wave 0: %.1f loop trips per walk (the slowest lane), %.1f the average lane, %.2f walks per block\n",
		        (double)h[10] / (double)n_blocks, (double)h[11] / (double)n_blocks, (double)h[12] / (double)(h[14] ? h[14] : 1),
		        (double)h[13] / 64.0 / (double)(h[14] ? h[14] : 1), (double)h[14] / (double)n_blocks);
	}
	if (want_stats) {
		uint32_t h[4];
		MSX_HIP(ctx, hipMemcpyAsync(h, d_bad + 8, 16, hipMemcpyDeviceToHost, ctx->stream));
		MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		fprintf(stderr, "# inflate: %lld blocks: %u literals, %u matches (%u beyond the ring), %u coded deflate blocks\n",
		        (long long)n_blocks, h[0], h[1], h[2], h[3]);
	}
	uint32_t bad[4] = {0, 0, 0, 0};
	MSX_HIP(ctx, hipMemcpyAsync(bad, d_bad, 16, hipMemcpyDeviceToHost, ctx->stream));
	MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (stats_mode == 2) fprintf(stderr, "# inflate: %lld blocks, %u handed back to the serial kernel, %u refused\n", (long long)n_blocks, bad[2], bad[0]);
	if (n_refused) *n_refused = bad[0];
	return MSX_OK;
}

// msx_runtime_warmup: this translation unit's code object loaded onto the device ahead of its first launch (the runtime loads a
// module when one of its kernels is first asked for: 2-10 ms each, otherwise paid by the first batches of a command)
void msx_touch_inflate(void) {
	hipFuncAttributes attr;
	(void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_bgzf_crc));
	(void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_bgzf_inflate_wave<false>));
}
