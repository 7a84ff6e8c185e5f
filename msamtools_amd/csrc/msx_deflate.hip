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

extern "C" int64_t msx_bgzf_bound(int64_t n_bytes, int level) {
	if (n_bytes <= 0) return 0;
	const int64_t nblk = (n_bytes + BZ_PAYLOAD - 1) / BZ_PAYLOAD;
	(void)level;
	return n_bytes + nblk * (int64_t)BZ_STORED_FRAME;
}

extern "C" int msx_bgzf_deflate(msx_ctx *ctx, const void *d_in, size_t n_bytes, int level, void *d_out, size_t out_cap,
                                int64_t *n_out, int64_t *n_blocks) {
	if (!ctx || !n_out || (n_bytes > 0 && (!d_in || !d_out))) return MSX_ERR_ARG;
	if (n_bytes > 0xfff00000ull) return msx_fail(ctx, MSX_ERR_ARG, "msx_bgzf_deflate: more than 4 GB in one call");
	msx_join(ctx);
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	*n_out = 0;
	if (n_blocks) *n_blocks = 0;
	if (n_bytes == 0) return MSX_OK;
	const int64_t nblk = (int64_t)((n_bytes + BZ_PAYLOAD - 1) / BZ_PAYLOAD);
	if (level != 0) return msx_fail(ctx, MSX_ERR_ARG, "msx_bgzf_deflate: level %d", level);
	const int64_t need = (int64_t)n_bytes + nblk * (int64_t)BZ_STORED_FRAME;
	if ((size_t)need > out_cap) return msx_fail(ctx, MSX_ERR_ARG, "msx_bgzf_deflate: output buffer too small (%lld > %zu)", (long long)need, out_cap);
	int rc = msx_bgzf_store_launch(ctx, ctx->stream, (const uint8_t *)d_in, nullptr, n_bytes, (uint8_t *)d_out);
	if (rc) return rc;
	MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	*n_out = need;
	if (n_blocks) *n_blocks = nblk;
	return MSX_OK;
}
