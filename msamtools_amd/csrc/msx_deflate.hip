// msx_deflate.hip -- the BGZF layer under mSamWrite on the device.
//
// msamtools filter writes its records through htslib (sam_write1 -> bgzf_write: msam_helper.c:270-272; mode "wbu" or
// "wb": msam_filter.c:464-470): the BAM byte stream is cut into payloads of <= 0xff00 bytes, each becomes one gzip
// member with a BC extra field (BSIZE), a raw DEFLATE stream -- one stored block for -u, level-6 deflate for -b -- and a
// trailer of CRC-32 and ISIZE.  Round 3 brought filter's output records down as one byte string and left the framing to
// the host: every byte copied into a block slot and run through the CRC there (msh_io.c: msh_write_stream).  Here the
// device hands down finished blocks:
//   k_bgzf_store     one wave per block: payload copied behind an 18-byte header and a 5-byte stored-block header,
//                    CRC-32 by the wave (msx_crc.h), trailer.  Blocks of one call are full (0xff00) except the last, so
//                    block k starts at k * (0xff00 + 31): the framed stream is contiguous by construction.
// Which bytes a BGZF writer puts in its blocks is not pinned by the reference ("compare record
// lines / decompressed streams only"; tests/functions.sh compares records); the rule here is that every block is a
// valid gzip member that zlib, htslib and this repository's own inflater decode to the same record stream.
// Integer / byte work, HBM-bound (stored) -- no MFMA.
#include "msx_internal.h"

#include <mutex>
#include "msx_crc.h"

#define BZ_PAYLOAD 0xff00u          // bytes of BAM stream per block (htslib's BGZF_BLOCK_SIZE)
#define BZ_STORED_FRAME 31u         // 18 (gzip header with BC field) + 5 (stored block header) + 8 (CRC-32, ISIZE)

typedef uint32_t __attribute__((aligned(1))) dfl_u32_u;

__device__ __forceinline__ void bz_header(uint8_t *o, uint32_t total) {
	// 1f 8b 08 04 | mtime 0 | xfl 0 | os ff | xlen 6 | 'B' 'C' 2 0 | BSIZE = total - 1
	const uint8_t h[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
	for (int k = 0; k < 16; k++) o[k] = h[k];
	o[16] = (uint8_t)((total - 1u) & 0xffu);
	o[17] = (uint8_t)((total - 1u) >> 8);
}
__device__ __forceinline__ void bz_trailer(uint8_t *o, uint32_t crc, uint32_t n) {
	o[0] = (uint8_t)crc; o[1] = (uint8_t)(crc >> 8); o[2] = (uint8_t)(crc >> 16); o[3] = (uint8_t)(crc >> 24);
	o[4] = (uint8_t)n; o[5] = (uint8_t)(n >> 8); o[6] = (uint8_t)(n >> 16); o[7] = (uint8_t)(n >> 24);
}

// -u: one wave per block.  *d_total (device) or n_total bytes of input; blocks beyond the input do nothing.
__global__ __launch_bounds__(64) void k_bgzf_store(const uint8_t *__restrict__ in, const uint32_t *__restrict__ d_total, uint32_t n_total,
                                                   uint8_t *__restrict__ out) {
	__shared__ uint32_t tab[256];
	const uint32_t lane = threadIdx.x, bi = blockIdx.x;
	const uint32_t total = d_total ? *d_total : n_total;
	const uint64_t lo = (uint64_t)bi * BZ_PAYLOAD;
	if (lo >= total) return;
	const uint32_t n = total - lo < BZ_PAYLOAD ? (uint32_t)(total - lo) : BZ_PAYLOAD;
	crc_table_fill(tab, lane);
	__syncthreads();
	const uint8_t *src = in + lo;                         // 256-byte aligned: 0xff00 = 255 * 256
	uint8_t *blk = out + (uint64_t)bi * (BZ_PAYLOAD + BZ_STORED_FRAME);
	uint8_t *dst = blk + 23;
	// 16 bytes per lane and step in, four unaligned dwords out (the destination sits 23 + 31 * bi bytes off)
	const uint32_t n16 = n & ~15u;
	for (uint32_t i = lane * 16u; i < n16; i += 64u * 16u) {
		const uint4 v = *reinterpret_cast<const uint4 *>(src + i);
		dfl_u32_u *d = reinterpret_cast<dfl_u32_u *>(dst + i);
		d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
	}
	if (lane < (n & 15u)) dst[n16 + lane] = src[n16 + lane];
	const uint32_t crc = crc_wave(src, n, tab, lane);
	if (lane == 0) {
		bz_header(blk, n + BZ_STORED_FRAME);
		blk[18] = 1;                                      // BFINAL = 1, BTYPE = 00 (stored)
		blk[19] = (uint8_t)n; blk[20] = (uint8_t)(n >> 8); blk[21] = (uint8_t)~n; blk[22] = (uint8_t)(~n >> 8);
		bz_trailer(dst + n, crc, n);
	}
}

// Frames (level 0) the byte string in[0 .. total) -- total = *d_total if given (a device word), else n_cap -- into out;
// the launch is sized by n_cap.  Framed size: total + 31 * ceil(total / 0xff00), block k at k * (0xff00 + 31).
int msx_bgzf_store_launch(msx_ctx *ctx, hipStream_t stream, const uint8_t *d_in, const uint32_t *d_total, size_t n_cap, uint8_t *d_out) {
	if (n_cap == 0) return MSX_OK;
	const size_t nblk = (n_cap + BZ_PAYLOAD - 1) / BZ_PAYLOAD;
	hipLaunchKernelGGL(k_bgzf_store, dim3((unsigned)nblk), dim3(64), 0, stream, d_in, d_total, (uint32_t)n_cap, d_out);
	MSX_HIP(ctx, hipGetLastError());
	return MSX_OK;
}

// ---------------------------------------------------------------------------
// -b: DEFLATE on the device
// ---------------------------------------------------------------------------
// One wave per BGZF block (blocks are drawn from a ticket: the launch holds as many waves as the chip keeps resident).
// DEFLATE is two things.  *LZ77*: which earlier bytes a position repeats.  zlib walks a hash chain per position, one
// position after another; here 64 consecutive positions look at once -- every lane hashes the 4 and the 8 bytes at its
// position, reads the most recent earlier position with that hash as the tables stood before the step (the lanes of a
// step do not see each other; the distances 1..8, which that rule hides, are compared directly), measures the match
// in a 16 KB ring of the input kept in LDS, and enters its own position (atomic max: the highest position of a step
// wins a slot).  The step's positions are then resolved in order -- inside an earlier match: nothing; a match whose
// right-hand neighbour is not longer (one-step lazy evaluation): a match token; otherwise a literal -- by a scalar loop
// that runs once per MATCH taken, literals between matches being ranges of a 64-bit mask.  *Huffman coding*: the
// token histogram (LDS atomics) is ranked, the two-queue construction runs on the ranked weights (the one serial
// part: 2 picks per used symbol), depths by pointer jumping, lengths limited to 15 bits as zlib's gen_bitlen does,
// canonical codes by ballots; the tokens are then coded 64 at a time (bit offsets by a wave scan, bits OR-ed into a
// 2 KB staging area in LDS, whole kilobytes written out).  A block that would be shorter with the fixed codes or
// stored is written that way.  The same algorithm, one position at a time, is msx_deflate_model.h (host, tests): the
// kernel's blocks equal its blocks bit for bit (tests/test_gpu_deflate.py).
// LDS: ring + tables in pass 1, reused by the coder: 13 KB per wave at the geometry -b takes -> 12 waves per compute unit
// (msx_bgzf_deflate_launch lists the geometries).
#define DF_SLOT (BZ_PAYLOAD + 1024u)  // bytes reserved per block while it is being built
#define DF_OUT0 32u                   // a slot's DEFLATE stream starts here (16-byte aligned); the block itself at DF_OUT0 - 18
#define DF_AHEAD 336u                 // bytes a step reads beyond its first position (63 + 258 + 8, rounded up)
#define DF_TOKCAP (BZ_PAYLOAD + 64u)  // tokens per block, at most (+ end of block)
#define DF_MAXMATCH 258u
#define DF_MUL4 2654435761u
#define DF_MUL8 0x9E3779B185EBCA87ull
#define DF_TOK_MATCH 0x80000000u
#define DF_LL 288                     // literal/length lengths live at ll[0..288), distance lengths at ll[288..320)

// the coder's working set (pass 2; it takes the place of the ring and the tables of pass 1)
struct df_hf {
	uint32_t stage[512];          // coded bits on their way out: two halves of 1 KB
	uint32_t crc_tab[256];
	uint32_t key[320];            // ranking keys
	uint32_t W[320];              // weights of the ranked symbols
	uint32_t inode[320];          // weights of the internal nodes, then their depths
	uint32_t cnt[336];            // internal nodes per depth / leaves per length
	uint32_t lcode[288], dcode[32], ccode[20];   // code | length << 16
	uint16_t par[320];            // parent of an internal node, then jump pointers
	uint16_t order[320];          // rank -> symbol
	uint16_t seq[336];            // both trees' lengths, run-length coded: symbol | extra << 8
	uint8_t ll[320];
	uint8_t cl[32];
};
// GEOMETRY of pass 1 (template parameters of the kernel): RING_DW dwords of input ring (a power of two; it must hold the
// window, a step, what a step reads ahead and the kilobyte the ring is filled by: 4 * RING_DW >= WINDOW + 64 + DF_AHEAD + 1024),
// HB4 / HB8 bits of hash for the two tables, WINDOW = the largest distance.  The kernel is a latency chain per wave (every LDS
// round trip exposed, the vector ALUs a third busy at one wave per SIMD), so what a launch achieves follows the waves a
// compute unit keeps resident, and that is the LDS a wave takes: msx_bgzf_deflate_launch lists the geometries.
template <int RING_DW, int HB4, int HB8>
struct df_lds {
	union {
		struct { uint32_t ring[RING_DW + 8]; uint16_t h4[1u << HB4]; uint16_t h8[1u << HB8]; } lz;   // (ring: its first 8 dwords once more behind the end;
		                                                                                               //  a table entry is a position + 1 <= 0xff00: 16 bits)
		df_hf hf;
	};
	uint32_t lf[288], dq[32], clf[32];
};

__device__ __forceinline__ uint32_t df_wave_incl_scan(uint32_t x) {   // row_shr 1/2/4/8 inside rows of 16, then row_bcast 15 and 31
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
	return x;
}
__device__ __forceinline__ uint32_t df_wave_sum(uint32_t x) { return (uint32_t)__builtin_amdgcn_readlane((int)df_wave_incl_scan(x), 63); }
__device__ __forceinline__ uint32_t df_wave_max(uint32_t x) {
	for (int s = 32; s >= 1; s >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)x, s); x = o > x ? o : x; }
	return x;
}
// (the ring's first 8 dwords are repeated behind its end, so that up to 5 consecutive dwords can be read from any index
//  without wrapping)
// 16 bytes from byte position pos
template <uint32_t RMASK>
__device__ __forceinline__ void df_ring128(const uint32_t *ring, uint32_t pos, uint32_t &a0, uint32_t &a1, uint32_t &a2, uint32_t &a3) {
	const uint32_t i = (pos >> 2) & RMASK, sh = pos & 3u;
	const uint32_t r0 = ring[i], r1 = ring[i + 1u], r2 = ring[i + 2u], r3 = ring[i + 3u], r4 = ring[i + 4u];
	a0 = __builtin_amdgcn_alignbyte(r1, r0, sh); a1 = __builtin_amdgcn_alignbyte(r2, r1, sh);
	a2 = __builtin_amdgcn_alignbyte(r3, r2, sh); a3 = __builtin_amdgcn_alignbyte(r4, r3, sh);
}
__device__ __forceinline__ uint32_t df_len_sym(uint32_t len) {
	const uint32_t l = len - 3u;
	if (l < 8u) return l;
	if (len == 258u) return 28u;
	const uint32_t b = 31u - (uint32_t)__builtin_clz(l);
	return ((b - 1u) << 2) + ((l >> (b - 2u)) & 3u);
}
__device__ __forceinline__ uint32_t df_dist_sym(uint32_t dist) {
	const uint32_t d = dist - 1u;
	if (d < 4u) return d;
	const uint32_t b = 31u - (uint32_t)__builtin_clz(d);
	return (b << 1) + ((d >> (b - 1u)) & 1u);
}

// Code lengths (<= maxbits) for freq[0..n), n <= 320, into len[] (LDS bytes, zero for unused symbols); at least two
// symbols get a code.  Every lane calls it; the wave works together.
__device__ void df_huff_lengths(df_hf &H, const uint32_t *freq, uint32_t n, uint32_t maxbits, uint8_t *len, uint32_t lane) {
	// used symbols; force two (the lowest unused ones, weight 1)
	uint32_t own_f[5], own_k[5], r[5];
	uint32_t used = 0;
	for (uint32_t k = 0; k < 5u; k++) {
		const uint32_t i = lane + 64u * k;
		own_f[k] = i < n ? freq[i] : 0u;
		used += own_f[k] != 0u;
	}
	uint32_t m = df_wave_sum(used);
	uint32_t force0 = 0xffffu, force1 = 0xffffu;
	if (m < 2u) {                                        // (wave-uniform; rare: an empty distance tree, a one-symbol block)
		for (uint32_t i = 0; m < 2u; i++)
			if (freq[i] == 0u) { if (force0 == 0xffffu) force0 = i; else force1 = i; m++; }
	}
	for (uint32_t k = 0; k < 5u; k++) {
		const uint32_t i = lane + 64u * k;
		if (i < n && (i == force0 || i == force1)) own_f[k] = 1u;
		own_k[k] = own_f[k] ? (own_f[k] << 9 | i) : 0xffffffffu;
		if (i < 320u) H.key[i] = own_k[k];
		if (i < n) len[i] = 0;
		r[k] = 0;
	}
	__syncthreads();
	// rank by (weight, symbol): every lane counts, for its symbols, the used keys below its own
	for (uint32_t j = 0; j < n; j++) {
		const uint32_t kj = H.key[j];
		for (uint32_t k = 0; k < 5u; k++) r[k] += kj < own_k[k];
	}
	for (uint32_t k = 0; k < 5u; k++)
		if (own_f[k]) { H.W[r[k]] = own_f[k]; H.order[r[k]] = (uint16_t)(lane + 64u * k); }
	for (uint32_t i = lane; i < 336u; i += 64u) H.cnt[i] = 0;
	__syncthreads();
	// the two-queue construction on the ranked weights (ties: the leaf first, as Moffat & Katajainen's in-place form
	// does): internal node `next` = the two lightest of (leaves from `leaf` on, internal nodes from `root` on).
	// Wave-uniform and serial: one dependent LDS read per pick.
	{
		uint32_t leaf = 0, root = 0;
		uint32_t wl = H.W[0], wi = 0;                 // the heads of the two queues (wi valid while root < next)
		for (uint32_t next = 0; next + 1u < m; next++) {
			uint32_t sum = 0;
			for (int pick = 0; pick < 2; pick++) {
				if (leaf < m && (root >= next || wl <= wi)) {
					sum += wl;
					leaf++;
					if (leaf < m) wl = H.W[leaf];
				} else {
					sum += wi;
					if (lane == 0) H.par[root] = (uint16_t)next;
					root++;
					if (root < next) wi = H.inode[root];
				}
			}
			if (lane == 0) H.inode[next] = sum;
			if (root == next) wi = sum;                      // the queue of internal nodes was empty: this one heads it
		}
		if (lane == 0) H.par[m - 2u] = (uint16_t)(m - 2u);   // the root
	}
	__syncthreads();
	// depths of the internal nodes by pointer jumping (a node's parent has a higher number; <= 9 doublings for 319 nodes)
	{
		const uint32_t ni = m - 1u;
		uint32_t d[5], q[5];
		for (uint32_t k = 0; k < 5u; k++) {
			const uint32_t i = lane + 64u * k;
			d[k] = (i < ni && i != ni - 1u) ? 1u : 0u;
			q[k] = i < ni ? H.par[i] : 0u;
		}
		for (uint32_t k = 0; k < 5u; k++) { const uint32_t i = lane + 64u * k; if (i < ni) H.inode[i] = d[k]; }
		__syncthreads();
		for (int round = 0; round < 9; round++) {
			uint32_t dq_[5], pq[5];
			for (uint32_t k = 0; k < 5u; k++) {
				const uint32_t i = lane + 64u * k;
				dq_[k] = i < ni ? H.inode[q[k]] : 0u;
				pq[k] = i < ni ? H.par[q[k]] : 0u;
			}
			__syncthreads();
			for (uint32_t k = 0; k < 5u; k++) {
				const uint32_t i = lane + 64u * k;
				if (i < ni) { d[k] += dq_[k]; q[k] = pq[k]; H.inode[i] = d[k]; H.par[i] = (uint16_t)q[k]; }
			}
			__syncthreads();
		}
		// leaves at depth d + 1 = 2 * (internal nodes at depth d) - (internal nodes at depth d + 1)
		for (uint32_t k = 0; k < 5u; k++) { const uint32_t i = lane + 64u * k; if (i < ni) atomicAdd(&H.cnt[d[k]], 1u); }
		__syncthreads();
		uint32_t lv[5];
		for (uint32_t k = 0; k < 5u; k++) {
			const uint32_t dd = lane + 64u * k;              // depth dd -> leaves at dd + 1
			lv[k] = dd < 320u ? 2u * H.cnt[dd] - H.cnt[dd + 1u] : 0u;
		}
		__syncthreads();
		uint32_t over = 0;
		for (uint32_t k = 0; k < 5u; k++) {
			const uint32_t dd = lane + 64u * k;
			if (dd < 320u) H.cnt[dd + 1u] = lv[k];
			if (dd + 1u > maxbits) over += lv[k];
		}
		if (lane == 0) H.cnt[0] = 0;
		__syncthreads();
		const uint32_t overflow = df_wave_sum(over);
		if (overflow) {
			// Leaves deeper than the limit go up to it: the code is then over-subscribed by `excess` units of 2^-maxbits
			// (Kraft sum).  zlib's step (trees.c gen_bitlen: a leaf of the deepest level below the limit that has one goes
			// down a level and takes a leaf of the limit's level along as its sibling) takes exactly one unit away -- the
			// number of steps comes from the Kraft sum, not from the number of leaves moved (msx_deflate_model.h).
			if (lane == 0) {
				H.cnt[maxbits] += overflow;
				long excess = -(1L << maxbits);
				for (uint32_t b = 1; b <= maxbits; b++) excess += (long)H.cnt[b] << (maxbits - b);
				while (excess > 0) {
					uint32_t bits = maxbits - 1u;
					while (H.cnt[bits] == 0u) bits--;
					H.cnt[bits]--;
					H.cnt[bits + 1u] += 2u;
					H.cnt[maxbits]--;
					excess--;
				}
			}
			__syncthreads();
		}
	}
	// the counts handed out in rank order: the rarest symbols get the longest codes
	{
		uint32_t got[5] = {0, 0, 0, 0, 0};
		uint32_t cum = 0;
		for (uint32_t bits = maxbits; bits >= 1u; bits--) {
			const uint32_t c = H.cnt[bits];
			for (uint32_t k = 0; k < 5u; k++) {
				const uint32_t rr = lane + 64u * k;
				if (rr >= cum && rr < cum + c) got[k] = bits;
			}
			cum += c;
		}
		for (uint32_t k = 0; k < 5u; k++) {
			const uint32_t rr = lane + 64u * k;
			if (rr < m) len[H.order[rr]] = (uint8_t)got[k];
		}
	}
	__syncthreads();
}

// canonical codes (RFC 1951 3.2.2) of len[0..n), bit-reversed for an LSB-first writer: out[i] = code | length << 16
__device__ void df_huff_codes(df_hf &H, const uint8_t *len, uint32_t n, uint32_t *out, uint32_t lane) {
	uint32_t l[5], base[5];
	for (uint32_t k = 0; k < 5u; k++) { const uint32_t i = lane + 64u * k; l[k] = i < n ? len[i] : 0u; base[k] = 0; }
	// per length: how many symbols, and every symbol's number among those of its length (symbol order)
	uint32_t next = 0, prev_cnt = 0;
	for (uint32_t b = 1; b <= 15u; b++) {
		next = (next + prev_cnt) << 1;
		uint32_t run = 0;
		for (uint32_t k = 0; k < 5u; k++) {
			const uint64_t mk = __ballot(l[k] == b);
			if (l[k] == b) base[k] = next + run + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull));
			run += (uint32_t)__popcll(mk);
		}
		prev_cnt = run;
	}
	for (uint32_t k = 0; k < 5u; k++) {
		const uint32_t i = lane + 64u * k;
		if (i < n) out[i] = l[k] ? (__brev(base[k]) >> (32u - l[k])) | l[k] << 16 : 0u;
	}
	__syncthreads();
}

// 64 (value, bit count <= 48) pairs appended to the coded stream: offsets by a wave scan, bits OR-ed into the staging
// area, every kilobyte that is complete written out (outp + 16-byte aligned offsets) and cleared
__device__ __forceinline__ void df_emit(df_hf &H, uint64_t acc, uint32_t nb, uint32_t &bitpos, uint32_t &flushed, uint8_t *outp, uint32_t lane) {
	const uint32_t incl = df_wave_incl_scan(nb);
	const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
	if (nb) {
		const uint32_t B = bitpos + incl - nb, w = B >> 5, s = B & 31u;
		const uint64_t lo = acc << s;
		atomicOr(&H.stage[w & 511u], (uint32_t)lo);
		if ((uint32_t)(lo >> 32)) atomicOr(&H.stage[(w + 1u) & 511u], (uint32_t)(lo >> 32));
		if (s > 16u) { const uint32_t hi = (uint32_t)(acc >> (64u - s)); if (hi) atomicOr(&H.stage[(w + 2u) & 511u], hi); }
	}
	bitpos += total;
	__syncthreads();
	while (bitpos - flushed >= 8192u) {
		const uint32_t half = (flushed >> 13) & 1u;
		uint4 *sp = reinterpret_cast<uint4 *>(&H.stage[half * 256u + lane * 4u]);
		*reinterpret_cast<uint4 *>(outp + (flushed >> 3) + lane * 16u) = *sp;
		*sp = make_uint4(0, 0, 0, 0);
		flushed += 8192u;
		__syncthreads();
	}
}

template <int RING_DW, int HB4, int HB8, int WINDOW, int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE))) void k_bgzf_deflate(const uint8_t *__restrict__ in, const uint32_t *__restrict__ d_total, uint32_t n_total,
                                                     uint8_t *__restrict__ slots, uint32_t *__restrict__ bsize, uint32_t *__restrict__ tok_all,
                                                     uint32_t *__restrict__ ticket, uint32_t *__restrict__ kinds) {
	// MSX_DEFLATE_STATS: kinds[0..2] count the blocks written stored / with the fixed codes / with codes of their own; behind them
	// (as 64-bit words from kinds + 4 on) the clocks the waves spent per phase
	unsigned long long *prof = kinds ? reinterpret_cast<unsigned long long *>(kinds + 4) : nullptr;
	unsigned long long t_mark = 0, t_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define DF_T0() do { if (prof) t_mark = __builtin_readcyclecounter(); } while (0)
#define DF_T(i) do { if (prof) { const unsigned long long now_ = __builtin_readcyclecounter(); t_acc[i] += now_ - t_mark; t_mark = now_; } } while (0)
	static_assert(4 * RING_DW >= WINDOW + 64 + (int)DF_AHEAD + 1024, "the ring holds window, step, look-ahead and a kilobyte of fill");
	constexpr uint32_t DF_RMASK = (uint32_t)RING_DW - 1u;
	__shared__ df_lds<RING_DW, HB4, HB8> S;
	__shared__ uint32_t s_bi;
	const uint32_t lane = threadIdx.x;
	const uint32_t total = d_total ? *d_total : n_total;
	const uint32_t nblk = (uint32_t)(((uint64_t)total + BZ_PAYLOAD - 1u) / BZ_PAYLOAD);
	uint32_t *tok = tok_all + (size_t)blockIdx.x * DF_TOKCAP;
	for (;;) {
		__syncthreads();
		if (lane == 0) s_bi = atomicAdd(ticket, 1u);
		__syncthreads();
		const uint32_t bi = s_bi;
		if (bi >= nblk) break;
		const uint64_t lo = (uint64_t)bi * BZ_PAYLOAD;
		const uint32_t avail = total - (uint32_t)lo;               // readable bytes from src on
		const uint32_t n = avail < BZ_PAYLOAD ? avail : BZ_PAYLOAD;
		const uint8_t *src = in + lo;
		uint8_t *slot = slots + (size_t)bi * DF_SLOT;
		// ---- pass 1: tokens and their histogram ----
		DF_T0();
		for (uint32_t i = lane * 8u; i < (1u << HB4) + (1u << HB8); i += 512u)
			*reinterpret_cast<uint4 *>(&S.lz.h4[i]) = make_uint4(0, 0, 0, 0);     // (h8 follows h4; eight 16-bit entries per store)
		for (uint32_t i = lane; i < 288u; i += 64u) S.lf[i] = 0;
		if (lane < 32u) { S.dq[lane] = 0; S.clf[lane] = 0; }
		uint32_t filled = 0, next_free = 0, nt = 0;
		for (uint32_t p0 = 0; p0 < n; p0 += 64u) {
			while (filled < n && filled < p0 + 64u + DF_AHEAD) {       // the ring, a kilobyte at a time
				const uint32_t off = filled + lane * 16u;
				uint4 v = make_uint4(0, 0, 0, 0);
				if (off + 16u <= avail) {
					v = *reinterpret_cast<const uint4 *>(src + off);
				} else if (off < avail) {
					uint32_t w[4] = {0, 0, 0, 0};
					for (uint32_t k = 0; off + k < avail; k++) w[k >> 2] |= (uint32_t)src[off + k] << (8u * (k & 3u));
					v = make_uint4(w[0], w[1], w[2], w[3]);
				}
				const uint32_t ri = (off >> 2) & DF_RMASK;
				*reinterpret_cast<uint4 *>(&S.lz.ring[ri]) = v;
				if (ri < 8u) *reinterpret_cast<uint4 *>(&S.lz.ring[(uint32_t)RING_DW + ri]) = v;      // (lanes 0 and 1 of the chunk that wraps)
				filled += 1024u;
			}
			__syncthreads();
			DF_T(0);
			const uint32_t p = p0 + lane;
			const bool active = p < n;
			const bool has4 = p + 4u <= n, has8 = p + 8u <= n;
			// bytes p - 8 .. p + 7
			const uint32_t q = p - 8u, qi = (uint32_t)((int32_t)q >> 2), sh = q & 3u;
			const uint32_t qm = qi & DF_RMASK;
			const uint32_t r0 = S.lz.ring[qm], r1 = S.lz.ring[qm + 1u], r2 = S.lz.ring[qm + 2u], r3 = S.lz.ring[qm + 3u], r4 = S.lz.ring[qm + 4u];
			const uint32_t W0 = __builtin_amdgcn_alignbyte(r1, r0, sh), W1 = __builtin_amdgcn_alignbyte(r2, r1, sh),
			               W2 = __builtin_amdgcn_alignbyte(r3, r2, sh), W3 = __builtin_amdgcn_alignbyte(r4, r3, sh);
			const uint32_t h4i = (W2 * DF_MUL4) >> (32 - HB4);
			const uint32_t h8i = (uint32_t)((((uint64_t)W3 << 32 | W2) * DF_MUL8) >> (64 - HB8));
			const uint32_t c4 = has4 ? S.lz.h4[h4i] : 0u, c8 = has8 ? S.lz.h8[h8i] : 0u;
			// the nearest of the distances 1..8 whose 4 bytes repeat (runs, short periods: what the tables of earlier steps
			// cannot show a position).  Such a position stays out of the tables: inside a run every position would enter
			// with the same keys, and the tables would remember the nearest repeat instead of the run's start in an
			// earlier record, from where the long match goes on behind the run's end.
			uint32_t nd = 0;
			if (has4) {
				const uint64_t A = (uint64_t)W2 << 32 | W1, B = (uint64_t)W1 << 32 | W0;
#pragma unroll
				for (uint32_t d = 8; d >= 5u; d--) if ((uint32_t)(B >> (8u * (8u - d))) == W2 && d <= p) nd = d;
#pragma unroll
				for (uint32_t d = 4; d >= 1u; d--) if ((uint32_t)(A >> (8u * (4u - d))) == W2 && d <= p) nd = d;
			}
			// Entering the step's positions: the highest position of a step wins a slot (earlier steps' entries are lower: a plain
			// store replaces them).  16-bit entries have no atomic max in LDS: every lane stores, reads back, and a lane that finds a
			// LOWER position than its own in its slot stores again -- the lanes of a wave that meet in a slot (rare) settle in as
			// many rounds as there are of them, one round otherwise.  (32-bit entries and ds_max_u32 cost 12 KB of tables instead of
			// 6: nine waves per compute unit instead of twelve.)
			{
				bool w4 = has4 && !nd, w8 = has8 && !nd;
				const uint16_t mine = (uint16_t)(p + 1u);
				do {
					if (w4) S.lz.h4[h4i] = mine;
					if (w8) S.lz.h8[h8i] = mine;
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
					w4 = w4 && ((volatile uint16_t *)S.lz.h4)[h4i] < mine;
					w8 = w8 && ((volatile uint16_t *)S.lz.h8)[h8i] < mine;
				} while (__ballot(w4 || w8));
			}
			uint32_t bl = 0, bd = 0;
			const bool need = active && p >= next_free;
			DF_T(1);
			if (__ballot(need)) {
				const uint32_t maxl = (n - p) < DF_MAXMATCH ? (n - p) : DF_MAXMATCH;
				if (!need) nd = 0;
				const uint32_t a4 = c4 - 1u, a8 = c8 - 1u;
				const bool v4 = need && c4 != 0u && p - a4 <= (uint32_t)WINDOW;
				const bool v8 = need && c8 != 0u && p - a8 <= (uint32_t)WINDOW && !(v4 && c8 == c4);
				// The candidates in a fixed order -- near, 4-byte table, 8-byte table; a later one must be strictly longer.
				// FIRST their first 16 bytes, all three side by side against the same 16 bytes at the position: on BAM records most
				// matches end there, and most steps with them (one look instead of one trip per candidate and lane).  THEN one loop
				// for those that go on: every trip compares 16 more bytes of each lane's current candidate, a lane that has finished
				// one moves on to its next (the wave waits for the lane with the most bytes to compare in all, not for the longest
				// match of each kind in turn).  A later candidate is first looked at where it would have to match to be longer --
				// the 16 bytes that end behind the best length so far -- and dropped there mostly.
				const uint32_t NONE = 0xffffffffu;
				const uint32_t c0 = nd ? p - nd : NONE, c1 = v4 ? a4 : NONE, c2 = v8 ? a8 : NONE;
				uint32_t q0 = NONE, q1 = NONE, q2 = NONE;         // the candidates that match their first 16 bytes, in order
				{
					uint32_t b0, b1, b2, b3;
					df_ring128<DF_RMASK>(S.lz.ring, p, b0, b1, b2, b3);
					const uint32_t cap16 = maxl < 16u ? maxl : 16u;
#define DF_FIRST16(C) do { \
					uint32_t a0, a1, a2, a3; \
					df_ring128<DF_RMASK>(S.lz.ring, (C) != NONE ? (C) : p, a0, a1, a2, a3); \
					const uint32_t x0 = a0 ^ b0, x1 = a1 ^ b1, x2 = a2 ^ b2, x3 = a3 ^ b3; \
					/* the first differing byte: one count of trailing zeros on the first word that differs (| 1u << 31 keeps it defined) */ \
					const uint32_t xs = x0 ? x0 : x1 ? x1 : x2 ? x2 : x3, at = x0 ? 0u : x1 ? 4u : x2 ? 8u : 12u; \
					uint32_t l16 = (x0 | x1 | x2 | x3) ? at + ((uint32_t)__builtin_ctz(xs | 0x80000000u) >> 3) : 16u; \
					l16 = l16 > cap16 ? cap16 : l16; \
					const bool here = (C) != NONE; \
					const bool goes_on = here && l16 == 16u && maxl > 16u; \
					const bool take = here && !goes_on && l16 >= 4u && l16 > bl; \
					bd = take ? p - (C) : bd; \
					bl = take ? l16 : bl; \
					const uint32_t s_ = goes_on ? (C) : NONE; \
					q2 = (q0 != NONE && q1 != NONE) ? s_ : q2; \
					q1 = (q0 != NONE && q1 == NONE) ? s_ : q1; \
					q0 = q0 == NONE ? s_ : q0; \
				} while (0)
					DF_FIRST16(c0);
					DF_FIRST16(c1);
					DF_FIRST16(c2);
#undef DF_FIRST16
				}
				uint32_t cur = q0, l = 16u;
				bool quick = false, act = need && cur != NONE;
				while (__ballot(act)) {
					if (act) {                                   // (one masked region; inside it selects, no branches)
						const uint32_t off = quick ? bl - 15u : l;
						uint32_t a0, a1, a2, a3, b0, b1, b2, b3;
						df_ring128<DF_RMASK>(S.lz.ring, cur + off, a0, a1, a2, a3);
						df_ring128<DF_RMASK>(S.lz.ring, p + off, b0, b1, b2, b3);
						const uint32_t x0 = a0 ^ b0, x1 = a1 ^ b1, x2 = a2 ^ b2, x3 = a3 ^ b3;
						const bool diff = (x0 | x1 | x2 | x3) != 0u;
						const uint32_t xs = x0 ? x0 : x1 ? x1 : x2 ? x2 : x3, at = x0 ? 0u : x1 ? 4u : x2 ? 8u : 12u;
						const uint32_t fd = at + ((uint32_t)__builtin_ctz(xs | 0x80000000u) >> 3);
						const uint32_t l16 = l + 16u;
						const bool end = !diff && l16 >= maxl;            // (measuring) ran into the longest a match may be
						const bool fin = quick ? diff : (diff || end);    // quick: a difference drops the candidate; none: measure it from byte 16
						uint32_t len = quick ? 0u : (diff ? l + fd : maxl);
						len = len > maxl ? maxl : len;
						const bool take = fin && len >= 4u && len > bl;
						bd = take ? p - cur : bd;
						bl = take ? len : bl;
						l = (quick || fin) ? 16u : l16;
						cur = fin ? q1 : cur;
						q1 = fin ? q2 : q1;
						q2 = fin ? NONE : q2;
						quick = fin && bl >= 32u;
						act = fin ? (cur != NONE && bl < maxl) : true;
					}
				}
				if (!need) bl = 0;
			}
			DF_T(2);
			// resolve the step in order: once around the loop per match taken
			const uint32_t cnt = (n - p0) < 64u ? (n - p0) : 64u;
			const uint32_t ml_next = (uint32_t)__shfl_down((int)bl, 1);
			const bool M = bl >= 3u && !(lane + 1u < cnt && ml_next > bl);
			const uint64_t Mmask = __ballot(M);
			uint32_t cur = next_free > p0 ? next_free - p0 : 0u;
			uint64_t tokmask = 0;
			const uint64_t upto_cnt = (2ull << (cnt - 1u)) - 1ull;          // bits [0, cnt)
			while (cur < cnt) {
				const uint64_t from_cur = ~0ull << cur, mm = Mmask & from_cur;
				if (!mm) { tokmask |= upto_cnt & from_cur; cur = cnt; break; }       // literals to the step's end
				const uint32_t j = (uint32_t)__builtin_ctzll(mm);
				tokmask |= ((2ull << j) - 1ull) & from_cur;                          // literals [cur, j), the match at j
				cur = j + (uint32_t)__builtin_amdgcn_readlane((int)bl, (int)j);
			}
			next_free = p0 + cur;
			DF_T(9);
			if ((tokmask >> lane) & 1ull) {
				const uint32_t rank = (uint32_t)__popcll(tokmask & ((1ull << lane) - 1ull));
				uint32_t t;
				if (M) {
					t = DF_TOK_MATCH | (bl - 3u) << 15 | (bd - 1u);
					atomicAdd(&S.lf[257u + df_len_sym(bl)], 1u);
					atomicAdd(&S.dq[df_dist_sym(bd)], 1u);
				} else {
					t = W2 & 0xffu;
					atomicAdd(&S.lf[t], 1u);
				}
				tok[nt + rank] = t;
			}
			nt += (uint32_t)__popcll(tokmask);
			DF_T(3);
		}
		if (lane == 0) { tok[nt] = 256u; S.lf[256] += 1u; }          // end of block
		nt += 1u;
		__threadfence_block();
		__syncthreads();
		// ---- CRC-32 of the input (the ring's memory now belongs to the coder) ----
		crc_table_fill(S.hf.crc_tab, lane);
		for (uint32_t i = lane; i < 512u; i += 64u) S.hf.stage[i] = 0;
		__syncthreads();
		const uint32_t crc = n ? crc_wave(src, n, S.hf.crc_tab, lane) : 0u;
		DF_T(4);
		// ---- the two trees, the code-length code, what each way of writing the block costs ----
		df_huff_lengths(S.hf, S.lf, 286u, 15u, S.hf.ll, lane);
		df_huff_lengths(S.hf, S.dq, 30u, 15u, S.hf.ll + DF_LL, lane);
		DF_T(5);
		uint32_t hlit, hdist;
		{
			uint32_t hi_l = 0, hi_d = 0;
			for (uint32_t k = 0; k < 5u; k++) { const uint32_t i = lane + 64u * k; if (i < 286u && S.hf.ll[i]) hi_l = i + 1u; }
			if (lane < 30u && S.hf.ll[DF_LL + lane]) hi_d = lane + 1u;
			hlit = df_wave_max(hi_l); hdist = df_wave_max(hi_d);
			if (hlit < 257u) hlit = 257u;
			if (hdist < 1u) hdist = 1u;
		}
		// run-length code the hlit + hdist lengths (symbols 16 / 17 / 18): the lengths sit in registers, 64 per register,
		// and a run's end is found by ballots; the entries are written by lane 0
		uint32_t ns = 0;
		{
			const uint32_t ntot = hlit + hdist;
			uint32_t v[5];
			for (uint32_t k = 0; k < 5u; k++) {
				const uint32_t i = lane + 64u * k;
				v[k] = i < hlit ? S.hf.ll[i] : i < ntot ? S.hf.ll[DF_LL + i - hlit] : 0xffu;
			}
			uint64_t ne[5];                                          // per register: where the value differs from the lane before
			uint32_t i = 0;
			while (i < ntot) {
				const uint32_t k0 = i >> 6, l0 = i & 63u;
				uint32_t val = 0;
				for (uint32_t k = 0; k < 5u; k++) if (k == k0) val = (uint32_t)__builtin_amdgcn_readlane((int)v[k], (int)l0);
				for (uint32_t k = 0; k < 5u; k++) ne[k] = __ballot(v[k] != val);
				uint32_t end = ntot;
				for (uint32_t k = k0; k < 5u; k++) {
					uint64_t mk = ne[k];
					if (k == k0) mk &= (l0 == 63u) ? 0ull : ~((1ull << (l0 + 1u)) - 1ull);
					if (mk) { end = k * 64u + (uint32_t)__builtin_ctzll(mk); break; }
				}
				if (end > ntot) end = ntot;
				uint32_t run = end - i;
				if (val == 0u) {
					while (run >= 11u) { const uint32_t t = run > 138u ? 138u : run; if (lane == 0) { S.hf.seq[ns] = (uint16_t)(18u | (t - 11u) << 8); S.clf[18] += 1u; } ns++; run -= t; }
					if (run >= 3u) { if (lane == 0) { S.hf.seq[ns] = (uint16_t)(17u | (run - 3u) << 8); S.clf[17] += 1u; } ns++; run = 0; }
					while (run > 0u) { if (lane == 0) { S.hf.seq[ns] = 0; S.clf[0] += 1u; } ns++; run--; }
				} else {
					if (lane == 0) { S.hf.seq[ns] = (uint16_t)val; S.clf[val] += 1u; }
					ns++; run--;
					while (run >= 3u) { const uint32_t t = run > 6u ? 6u : run; if (lane == 0) { S.hf.seq[ns] = (uint16_t)(16u | (t - 3u) << 8); S.clf[16] += 1u; } ns++; run -= t; }
					while (run > 0u) { if (lane == 0) { S.hf.seq[ns] = (uint16_t)val; S.clf[val] += 1u; } ns++; run--; }
				}
				i = end;
			}
		}
		__syncthreads();
		DF_T(6);
		df_huff_lengths(S.hf, S.clf, 19u, 7u, S.hf.cl, lane);
		const uint8_t cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
		uint32_t hclen;
		{
			uint32_t hi = 0;
			if (lane < 19u && S.hf.cl[cl_order[lane]]) hi = lane + 1u;
			hclen = df_wave_max(hi);
			if (hclen < 4u) hclen = 4u;
		}
		uint32_t bits_dyn, bits_fix;
		{
			uint32_t dyn = 0, fix = 0, extra = 0;
			for (uint32_t k = 0; k < 5u; k++) {
				const uint32_t i = lane + 64u * k;
				if (i < 286u) {
					const uint32_t f = S.lf[i];
					dyn += f * S.hf.ll[i];
					fix += f * (i < 144u ? 8u : i < 256u ? 9u : i < 280u ? 7u : 8u);
					if (i >= 265u && i < 285u) extra += f * ((i - 261u) >> 2);
				}
			}
			if (lane < 30u) {
				const uint32_t f = S.dq[lane];
				dyn += f * S.hf.ll[DF_LL + lane];
				fix += f * 5u;
				if (lane >= 4u) extra += f * ((lane >> 1) - 1u);
			}
			if (lane < 19u) dyn += S.clf[lane] * S.hf.cl[lane] + (lane == 16u ? 2u * S.clf[16] : lane == 17u ? 3u * S.clf[17] : lane == 18u ? 7u * S.clf[18] : 0u);
			extra = df_wave_sum(extra);
			bits_dyn = df_wave_sum(dyn) + extra + 3u + 5u + 5u + 4u + 3u * hclen;
			bits_fix = df_wave_sum(fix) + extra + 3u;
			// A belt beside the braces: the three codes must be complete prefix codes (Kraft sum exactly 1) -- an inflater
			// refuses anything else.  Should the tree builder ever hand out lengths that are not, the block is written with
			// the fixed codes or stored instead (both always valid), never with a header no reader accepts.
			uint32_t kl = 0, kd = 0, kc = 0;
			for (uint32_t k = 0; k < 5u; k++) {
				const uint32_t i = lane + 64u * k;
				if (i < 286u && S.hf.ll[i]) kl += 1u << (15u - S.hf.ll[i]);
			}
			if (lane < 30u && S.hf.ll[DF_LL + lane]) kd = 1u << (15u - S.hf.ll[DF_LL + lane]);
			if (lane < 19u && S.hf.cl[lane]) kc = 1u << (7u - S.hf.cl[lane]);
			if (df_wave_sum(kl) != (1u << 15) || df_wave_sum(kd) != (1u << 15) || df_wave_sum(kc) != (1u << 7)) bits_dyn = 0xffffffffu;
		}
		DF_T(7);
		const uint32_t bits_stored = 8u * (5u + n);
		uint32_t kind = bits_fix <= bits_dyn ? 1u : 2u;
		const uint32_t best = kind == 1u ? bits_fix : bits_dyn;
		if (bits_stored <= best || n == 0u) kind = 0u;
		uint8_t *blk = slot + DF_OUT0 - 18u;
		uint32_t nbytes;
		if (kind == 0u) {
			uint8_t *dst = slot + DF_OUT0 + 5u;
			const uint32_t n16 = n & ~15u;
			for (uint32_t i = lane * 16u; i < n16; i += 64u * 16u) {
				const uint4 v = *reinterpret_cast<const uint4 *>(src + i);
				dfl_u32_u *d = reinterpret_cast<dfl_u32_u *>(dst + i);
				d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
			}
			if (lane < (n & 15u)) dst[n16 + lane] = src[n16 + lane];
			if (lane == 0) {
				uint8_t *h = slot + DF_OUT0;
				h[0] = 1; h[1] = (uint8_t)n; h[2] = (uint8_t)(n >> 8); h[3] = (uint8_t)~n; h[4] = (uint8_t)(~n >> 8);
			}
			nbytes = 5u + n;
		} else {
			// ---- pass 2: the header and the tokens as bits ----
			uint32_t bitpos = 0, flushed = 0;
			uint8_t *outp = slot + DF_OUT0;
			if (kind == 2u) {
				df_huff_codes(S.hf, S.hf.ll, 286u, S.hf.lcode, lane);
				df_huff_codes(S.hf, S.hf.ll + DF_LL, 30u, S.hf.dcode, lane);
				df_huff_codes(S.hf, S.hf.cl, 19u, S.hf.ccode, lane);
				// BFINAL = 1, BTYPE = 10, HLIT, HDIST, HCLEN, then the code-length code's lengths in their fixed order
				uint64_t acc = 0;
				uint32_t nb = 0;
				if (lane == 0) { acc = 1u | 2u << 1 | (hlit - 257u) << 3 | (hdist - 1u) << 8 | (uint64_t)(hclen - 4u) << 13; nb = 17u; }
				else if (lane <= hclen) { acc = S.hf.cl[cl_order[lane - 1u]]; nb = 3u; }
				df_emit(S.hf, acc, nb, bitpos, flushed, outp, lane);
				for (uint32_t g = 0; g < ns; g += 64u) {
					acc = 0; nb = 0;
					if (g + lane < ns) {
						const uint32_t e = S.hf.seq[g + lane], sy = e & 0xffu, c = S.hf.ccode[sy];
						acc = c & 0xffffu; nb = c >> 16;
						acc |= (uint64_t)(e >> 8) << nb;
						nb += sy == 16u ? 2u : sy == 17u ? 3u : sy == 18u ? 7u : 0u;
					}
					df_emit(S.hf, acc, nb, bitpos, flushed, outp, lane);
				}
			} else {
				// the fixed codes (RFC 1951 3.2.6), bit-reversed
				for (uint32_t k = 0; k < 5u; k++) {
					const uint32_t i = lane + 64u * k;
					if (i < 286u) {
						const uint32_t l = i < 144u ? 8u : i < 256u ? 9u : i < 280u ? 7u : 8u;
						const uint32_t c = i < 144u ? 0x30u + i : i < 256u ? 0x190u + (i - 144u) : i < 280u ? i - 256u : 0xc0u + (i - 280u);
						S.hf.lcode[i] = (__brev(c) >> (32u - l)) | l << 16;
					}
				}
				if (lane < 30u) S.hf.dcode[lane] = (__brev(lane) >> 27) | 5u << 16;
				__syncthreads();
				df_emit(S.hf, 1u | 1u << 1, lane == 0 ? 3u : 0u, bitpos, flushed, outp, lane);
			}
			for (uint32_t g = 0; g < nt; g += 64u) {
				uint64_t acc = 0;
				uint32_t nb = 0;
				if (g + lane < nt) {
					const uint32_t t = tok[g + lane];
					if (t & DF_TOK_MATCH) {
						const uint32_t len = ((t >> 15) & 0xffu) + 3u, l3 = len - 3u, d1 = t & 0x7fffu;
						const uint32_t ls = df_len_sym(len), ds = df_dist_sym(d1 + 1u);
						const uint32_t lc = S.hf.lcode[257u + ls], dc = S.hf.dcode[ds];
						acc = lc & 0xffffu; nb = lc >> 16;
						if (ls >= 8u && ls < 28u) { const uint32_t eb = (ls >> 2) - 1u; acc |= (uint64_t)(l3 & ((1u << eb) - 1u)) << nb; nb += eb; }
						acc |= (uint64_t)(dc & 0xffffu) << nb; nb += dc >> 16;
						if (ds >= 4u) { const uint32_t eb = (ds >> 1) - 1u; acc |= (uint64_t)(d1 & ((1u << eb) - 1u)) << nb; nb += eb; }
					} else {
						const uint32_t lc = S.hf.lcode[t];
						acc = lc & 0xffffu; nb = lc >> 16;
					}
				}
				df_emit(S.hf, acc, nb, bitpos, flushed, outp, lane);
			}
			// what is left in the staging area: whole dwords by the lanes, the last bytes by lane 0
			nbytes = (bitpos + 7u) >> 3;
			const uint32_t fb = flushed >> 3, rem = nbytes - fb, nw = rem >> 2;
			for (uint32_t i = lane; i < nw; i += 64u)
				*reinterpret_cast<uint32_t *>(outp + fb + 4u * i) = S.hf.stage[((flushed >> 5) + i) & 511u];
			if (lane == 0) {
				const uint32_t w = S.hf.stage[((flushed >> 5) + nw) & 511u];
				for (uint32_t k = 0; k < (rem & 3u); k++) outp[fb + 4u * nw + k] = (uint8_t)(w >> (8u * k));
			}
		}
		DF_T(8);
		if (lane == 0) {
			bz_header(blk, 18u + nbytes + 8u);
			bz_trailer(blk + 18u + nbytes, crc, n);
			bsize[bi] = 18u + nbytes + 8u;
			if (kinds) atomicAdd(&kinds[kind], 1u);
			if (prof) for (int i = 0; i < 10; i++) { atomicAdd(&prof[i], t_acc[i]); }
		}
		if (prof) for (int i = 0; i < 10; i++) t_acc[i] = 0;
	}
}

// the finished blocks moved back to back: one wave per block; lane 0 of the last block leaves the stream's length
__global__ __launch_bounds__(64) void k_bgzf_compact(const uint8_t *__restrict__ slots, const uint32_t *__restrict__ bsize,
                                                     const uint32_t *__restrict__ boff, const uint32_t *__restrict__ d_total, uint32_t n_total,
                                                     uint8_t *__restrict__ out, uint32_t *__restrict__ out_total) {
	const uint32_t lane = threadIdx.x, bi = blockIdx.x;
	const uint32_t total = d_total ? *d_total : n_total;
	const uint32_t nblk = (uint32_t)(((uint64_t)total + BZ_PAYLOAD - 1u) / BZ_PAYLOAD);
	if (bi == 0 && lane == 0 && out_total) *out_total = nblk ? boff[nblk - 1u] + bsize[nblk - 1u] : 0u;
	if (bi >= nblk) return;
	const uint8_t *s = slots + (size_t)bi * DF_SLOT + DF_OUT0 - 18u;
	uint8_t *d = out + boff[bi];
	const uint32_t n = bsize[bi], n4 = n & ~3u;
	for (uint32_t i = lane * 4u; i < n4; i += 256u)
		*reinterpret_cast<dfl_u32_u *>(d + i) = *reinterpret_cast<const dfl_u32_u *>(s + i);
	if (lane < (n & 3u)) d[n4 + lane] = s[n4 + lane];
}

// ---------------------------------------------------------------------------
// ABI
// ---------------------------------------------------------------------------
extern "C" int64_t msx_bgzf_bound(int64_t n_bytes, int level) {
	if (n_bytes <= 0) return 0;
	const int64_t nblk = (n_bytes + BZ_PAYLOAD - 1) / BZ_PAYLOAD;
	(void)level;                      // (a block that does not compress is stored: no level needs more than level 0)
	return n_bytes + nblk * (int64_t)BZ_STORED_FRAME;
}

// exclusive sums of the blocks' sizes (one workgroup: a batch is a thousand blocks; 65 536 of them take 256 rounds)
__global__ __launch_bounds__(MSX_BLOCK) void k_bgzf_offsets(const uint32_t *__restrict__ bsize, uint32_t *__restrict__ boff, uint32_t n) {
	__shared__ uint32_t s_w[MSX_BLOCK / 64];
	const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
	uint32_t running = 0;
	for (uint32_t base = 0; base < n; base += MSX_BLOCK) {
		const uint32_t i = base + threadIdx.x;
		const uint32_t v = i < n ? bsize[i] : 0u;
		const uint32_t inc = df_wave_incl_scan(v);
		if (lane == 63u) s_w[w] = inc;
		__syncthreads();
		uint32_t woff = 0, tot = 0;
		for (uint32_t q = 0; q < MSX_BLOCK / 64; q++) { if (q < w) woff += s_w[q]; tot += s_w[q]; }
		if (i < n) boff[i] = running + woff + inc - v;
		running += tot;
		__syncthreads();
	}
	if (threadIdx.x == 0) boff[n] = running;
}

// Deflates (level >= 1) the byte string in[0 .. total) -- total = *d_total (a device word) if given, else n_cap -- into
// BGZF blocks, back to back in d_out; *d_out_total (device) receives the stream's length.  Enqueued on `stream` (every
// kernel its own: nothing of the context's scan workspace is touched, so the stream may run beside the context's).
int msx_bgzf_deflate_launch(msx_ctx *ctx, hipStream_t stream, const uint8_t *d_in, const uint32_t *d_total, size_t n_cap, uint8_t *d_out,
                            uint32_t *d_out_total, int level) {
	if (n_cap == 0) return MSX_OK;
	const size_t nblk = (n_cap + BZ_PAYLOAD - 1) / BZ_PAYLOAD;
	int rc;
	// The geometry of pass 1, <ring dwords, bits of the 4-byte table, bits of the 8-byte table, window>, by LEVEL -- what the level
	// dials is the LDS a wave takes, hence the waves a compute unit keeps resident, hence the rate.  Round 6, 16-bit table entries,
	// measured on 1 GB of lean records / 1.16 GB with SEQ/QUAL (15 249 / 17 782 blocks: several rounds of waves -- a launch of fewer
	// blocks than waves measures one block's latency, 2.7 ms, not the rate; profiles/round6/deflate_geometries.log):
	//   levels 7-9: <4096, 11, 11, 8192>  26 KB, 6 waves per compute unit   34.5 / 34.5 GB/s   1.079 / 1.120 of zlib -6's size
	//   levels 4-6: <1024, 10, 11, 2560>  13 KB, 12 waves                   53.9 / 52.6        1.087 / 1.129   (-b: htslib's default 6)
	//   levels 1-3: <1024, 10, 10, 2560>  11 KB, 12 waves (registers)       53.9 / 52.3        1.093 / 1.134
	// (round 5, 32-bit entries: 4 / 9 / 11 waves, 22 / 40 / 44 GB/s.)
	// MSX_DEFLATE_GEOM=0..5 overrides the level (0, 4, 3 are the three above; 1 = <2048, 11, 11, 6144>, 2 = <2048, 10, 10, 6144>,
	// 5 = <1024, 11, 11, 2560>: 46 GB/s at 8 waves, held there by their registers: measured, not mapped).  The host twin follows:
	// df_opts_for_level (msx_deflate_model.h).
	// (several device threads of one process launch the encoder at once: the lazily filled tables are filled under a lock)
	static std::mutex geom_mu;
	static int per_cu[6] = {0, 0, 0, 0, 0, 0};
	int geom_env = -1;
#ifdef MSX_DEBUG_SWITCHES
	geom_env = getenv("MSX_DEFLATE_GEOM") ? atoi(getenv("MSX_DEFLATE_GEOM")) : -1;     // (libmsamtools_amd_dbg.so only)
	if (geom_env < -1 || geom_env > 5) geom_env = -1;
#endif
	std::lock_guard<std::mutex> geom_lock(geom_mu);
	const int geom = geom_env >= 0 ? geom_env : level >= 7 ? 0 : level >= 4 ? 4 : 3;
	if (!per_cu[geom]) {
		int nb = 0;
		hipError_t e = geom == 5 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_bgzf_deflate<1024, 11, 11, 2560, 2>, 64, 0)
		             : geom == 4 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_bgzf_deflate<1024, 10, 11, 2560, 3>, 64, 0)
		             : geom == 3 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_bgzf_deflate<1024, 10, 10, 2560, 3>, 64, 0)
		             : geom == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_bgzf_deflate<2048, 10, 10, 6144, 2>, 64, 0)
		             : geom == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_bgzf_deflate<2048, 11, 11, 6144, 2>, 64, 0)
		                         : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_bgzf_deflate<4096, 11, 11, 8192, 1>, 64, 0);
		per_cu[geom] = (e == hipSuccess && nb > 0) ? nb : 4;
		if (getenv("MSX_DEFLATE_WAVES")) per_cu[geom] = atoi(getenv("MSX_DEFLATE_WAVES"));
		if (getenv("MSX_DEFLATE_STATS")) fprintf(stderr, "# deflate: geometry %d, %d waves per compute unit\n", geom, per_cu[geom]);
	}
	size_t waves = (size_t)ctx->num_cu * (size_t)per_cu[geom];
	if (waves > nblk) waves = nblk;
	// One set of scratch per context (slots, sizes with the ticket, token scratch), launches on whatever stream the caller
	// names: the launch is ordered behind the previous one's last kernel when that ran on another stream, and the stream
	// that last used the scratch is drained before any of it is freed to grow (msx_reserve drains ctx->stream only).
	const size_t need_slots = nblk * (size_t)DF_SLOT + 64, need_size = (nblk + 16) * 8 + 256, need_tok = waves * (size_t)DF_TOKCAP * 4 + 64;
	if (ctx->df_used && ctx->df_last != stream) {
		const bool grows = need_slots > ctx->df_slots.cap || need_size > ctx->df_size.cap || need_tok > ctx->df_tok.cap;
		if (grows) MSX_HIP(ctx, hipStreamSynchronize(ctx->df_last));
		else MSX_HIP(ctx, hipStreamWaitEvent(stream, ctx->df_done, 0));
	} else if (ctx->df_used && stream != ctx->stream &&
	           (need_slots > ctx->df_slots.cap || need_size > ctx->df_size.cap || need_tok > ctx->df_tok.cap)) {
		MSX_HIP(ctx, hipStreamSynchronize(stream));
	}
	if ((rc = msx_reserve(ctx, &ctx->df_slots, need_slots))) return rc;
	if ((rc = msx_reserve(ctx, &ctx->df_size, need_size))) return rc;
	if ((rc = msx_reserve(ctx, &ctx->df_tok, need_tok))) return rc;
	if (!ctx->df_done) MSX_HIP(ctx, hipEventCreateWithFlags(&ctx->df_done, hipEventDisableTiming));
	uint32_t *bsize = (uint32_t *)ctx->df_size.p, *boff = bsize + nblk + 8, *misc = boff + nblk + 4;
	MSX_HIP(ctx, hipMemsetAsync(bsize, 0, (nblk + 16) * 8 + 192, stream));
	const int want_kinds = getenv("MSX_DEFLATE_STATS") != nullptr;
#define DF_LAUNCH(R, H4, H8, W, E) hipLaunchKernelGGL((k_bgzf_deflate<R, H4, H8, W, E>), dim3((unsigned)waves), dim3(64), 0, stream, d_in, d_total, \
	                   (uint32_t)n_cap, (uint8_t *)ctx->df_slots.p, bsize, (uint32_t *)ctx->df_tok.p, misc, want_kinds ? misc + 4 : nullptr)
	switch (geom) {
	case 5: DF_LAUNCH(1024, 11, 11, 2560, 2); break;
	case 4: DF_LAUNCH(1024, 10, 11, 2560, 3); break;
	case 3: DF_LAUNCH(1024, 10, 10, 2560, 3); break;
	case 2: DF_LAUNCH(2048, 10, 10, 6144, 2); break;
	case 1: DF_LAUNCH(2048, 11, 11, 6144, 2); break;
	default: DF_LAUNCH(4096, 11, 11, 8192, 1); break;
	}
	hipLaunchKernelGGL(k_bgzf_offsets, dim3(1), dim3(MSX_BLOCK), 0, stream, (const uint32_t *)bsize, boff, (uint32_t)nblk);
	hipLaunchKernelGGL(k_bgzf_compact, dim3((unsigned)nblk), dim3(64), 0, stream, (const uint8_t *)ctx->df_slots.p, bsize, boff, d_total,
	                   (uint32_t)n_cap, d_out, d_out_total);
	MSX_HIP(ctx, hipGetLastError());
	MSX_HIP(ctx, hipEventRecord(ctx->df_done, stream));
	ctx->df_last = stream;
	ctx->df_used = true;
	if (want_kinds) {
		uint32_t h[4 + 2 * 10 + 2];
		MSX_HIP(ctx, hipMemcpyAsync(h, misc + 4, sizeof h, hipMemcpyDeviceToHost, stream));
		MSX_HIP(ctx, hipStreamSynchronize(stream));
		const unsigned long long *t = reinterpret_cast<const unsigned long long *>(h + 4);
		const double nb = (double)(h[0] + h[1] + h[2] ? h[0] + h[1] + h[2] : 1);
		fprintf(stderr, "# deflate: %zu blocks at most: %u stored, %u with the fixed codes, %u with codes of their own; clocks per block: ring+keys %.0f, "
		        "tables %.0f, matches %.0f, resolve %.0f, tokens %.0f, crc %.0f, two trees %.0f, run lengths %.0f, third tree+costs %.0f, coding %.0f\n",
		        nblk, h[0], h[1], h[2], t[0] / nb, t[1] / nb, t[2] / nb, t[9] / nb, t[3] / nb, t[4] / nb, t[5] / nb, t[6] / nb, t[7] / nb, t[8] / nb);
	}
	return MSX_OK;
}

extern "C" int msx_bgzf_deflate(msx_ctx *ctx, const void *d_in, size_t n_bytes, int level, void *d_out, size_t out_cap,
                                int64_t *n_out, int64_t *n_blocks) {
	if (!ctx || !n_out || (n_bytes > 0 && (!d_in || !d_out))) return MSX_ERR_ARG;
	// block offsets and the stream's length are 32-bit words on the device: what must stay below 2^32 is the OUTPUT's bound
	// (every block stored: n + 31 per 0xff00 bytes), not the input
	if (n_bytes > 0xfff00000ull || (level > 0 && (uint64_t)msx_bgzf_bound((int64_t)n_bytes, level) >= (1ull << 32)))
		return msx_fail(ctx, MSX_ERR_ARG, "msx_bgzf_deflate: more than 4 GB in one call");
	if (level < 0 || level > 9) return msx_fail(ctx, MSX_ERR_ARG, "msx_bgzf_deflate: level %d", level);
	msx_join(ctx);
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	*n_out = 0;
	if (n_blocks) *n_blocks = 0;
	if (n_bytes == 0) return MSX_OK;
	const int64_t nblk = (int64_t)((n_bytes + BZ_PAYLOAD - 1) / BZ_PAYLOAD);
	const int64_t need = msx_bgzf_bound((int64_t)n_bytes, level);
	if ((size_t)need > out_cap) return msx_fail(ctx, MSX_ERR_ARG, "msx_bgzf_deflate: output buffer too small (%lld > %zu)", (long long)need, out_cap);
	int rc;
	if (level == 0) {
		if ((rc = msx_bgzf_store_launch(ctx, ctx->stream, (const uint8_t *)d_in, nullptr, n_bytes, (uint8_t *)d_out))) return rc;
		MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		*n_out = need;
	} else {
		if ((rc = msx_reserve(ctx, &ctx->scan_l3, 64))) return rc;
		uint32_t *d_tot = (uint32_t *)ctx->scan_l3.p + 8, h_tot = 0;
		if ((rc = msx_bgzf_deflate_launch(ctx, ctx->stream, (const uint8_t *)d_in, nullptr, n_bytes, (uint8_t *)d_out, d_tot, level))) return rc;
		MSX_HIP(ctx, hipMemcpyAsync(&h_tot, d_tot, 4, hipMemcpyDeviceToHost, ctx->stream));
		MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		*n_out = h_tot;
	}
	if (n_blocks) *n_blocks = nblk;
	return MSX_OK;
}

// msx_runtime_warmup: this translation unit's code object loaded onto the device ahead of its first launch (the runtime loads a
// module when one of its kernels is first asked for: 2-10 ms each, otherwise paid by the first batches of a command)
void msx_touch_deflate(void) {
	hipFuncAttributes attr;
	(void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_bgzf_store));
}
