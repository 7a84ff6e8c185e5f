// msx_prop.hip -- proportional sharing of multi-mapped inserts
// (mInsertCountToAbundanceMatrix, msam_profile.c:317-410) without per-entry
// floating-point atomics.
//
// The reference loops over multi-mappers j, computes S_j = sum_{e in j} a[e]
// and adds a[e]/S_j to increment[e].  Here one iteration is
//   share[f] = sum over the lists containing f of w/S    segmented sum by feature (k_share_reduce;
//                                                         lists of <= 4 features carry their set with
//                                                         every entry, so S is summed there from a[])
//   recip[j] = w/S for the few longer lists                one lane per such list (k_general_recip);
//                                                         their entries gather it in k_share_reduce
//   a[f] = U[f] + a[f] * (share[f] + the partial sums of segments cut by a chunk boundary),
//   clamp, DELTA^2 and the convergence flag               (k_prop_apply, k_prop_finish)
// T's feature-major order is produced once per finalize by a stable LSD radix
// sort of (feature, list) pairs, so entries of one feature are contiguous and
// in ascending list order; hot features spread over many waves instead of
// serialising on one address.  a[f]*sum(1/S) differs from sum(a[f]/S) only in
// rounding (<= a few ulp, inside the 1e-6 relative bound of BASELINE.json).
// Across ranks `share` is the vector to all-reduce (C1 in SURVEY.md section 2).
#include "msx_internal.h"
#include <vector>
#include "msx_count.h"

#include <cstdlib>

#define RS_EPT 16                      // elements per thread in a radix pass
#define RS_TILE (RS_EPT * MSX_BLOCK)    // elements per workgroup tile
#define SR_CHUNK 2048                  // entries reduced by one wave

// ---------------------------------------------------------------------------
// build: (feature, list) pairs, stable radix sort by feature
// ---------------------------------------------------------------------------
// One radix pass (8-bit digit) = k_rs_hist + scan + k_rs_scatter, both on tiles of
// RS_TILE consecutive elements per 256-thread workgroup.
//   k_rs_hist     per-tile digit counts, hist[digit * n_tiles + tile]; keys read as 16-B vectors
//   k_rs_scatter  every thread first issues all its loads (RS_EPT keys + values, independent),
//                 ranks them (ballot match inside a row of 64, running per-wave digit counts in
//                 LDS, then a prefix over the four waves), moves the tile into LDS in digit
//                 order and writes it out from there: consecutive threads write consecutive
//                 addresses inside a digit's run instead of 64 scattered 4-byte stores per row.
// Order inside a digit is (wave, row, lane) = input order, so the pass is stable.
// SKIP: keys with bit 31 set are not part of the input (the per-pool word of msx_count.h: only
// a uniquely mapped insert holds a feature id there); n_ptr == nullptr: the element count is n_host.
#define RS_NOKEY 0xffffffffu
#define RS_SKIPPED(k) (((k) & 0x80000000u) != 0u)
// one key into the wave's digit counters; when every lane of the wave holds the same digit
// (sorted or heavily skewed input) one lane adds the whole row instead of 64 conflicting atomics
__device__ __forceinline__ void hist_add(uint32_t *cnt, uint32_t d, bool valid) {
	const unsigned long long act = __ballot(valid);
	if (!act) return;
	const uint32_t d0 = __shfl(d, __ffsll((long long)act) - 1, 64);
	if (__ballot(valid && d == d0) == act) {
		if ((threadIdx.x & 63) == __ffsll((long long)act) - 1) cnt[d0] += (uint32_t)__popcll(act);
	} else if (valid) {
		atomicAdd(&cnt[d], 1u);
	}
}

template <bool SKIP>
__global__ __launch_bounds__(MSX_BLOCK) void k_rs_hist(const uint32_t *__restrict__ keys,
                                                       const unsigned long long *__restrict__ n_ptr, int64_t n_host,
                                                       int shift, uint32_t dmask, uint32_t *__restrict__ hist,
                                                       int64_t n_tiles, int64_t tile0 = 0) {
	__shared__ uint32_t s_cnt[MSX_BLOCK / 64][256];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int64_t tile = blockIdx.x + tile0;      // (tile0: the tiles in front have their counts already, msx_sort_keys32)
	const int64_t E = n_ptr ? (int64_t)*n_ptr : n_host;
	const int64_t base = tile * RS_TILE;
	// The launch is sized for the host's upper bound of the element count.  With a device-side
	// count the table is laid out for the tiles that exist (stride nt = ceil(E / RS_TILE)), so a
	// tile beyond the data has no column at all -- 256 strided 4-byte stores per idle tile were
	// the larger part of this kernel's time when the bound is loose -- and the scan stops there too.
	if (base >= E) return;
	if (n_ptr) n_tiles = (E + RS_TILE - 1) / RS_TILE;
	for (int q = 0; q < 4; q++) s_cnt[w][lane + 64 * q] = 0;
	if (SKIP && base + RS_TILE <= E) {
		// (all of the thread's keys asked for before any is counted: a load per iteration under its bounds check left
		//  sixteen round trips in a row)
		const uint4 *kv = reinterpret_cast<const uint4 *>(keys + base);
		uint4 v[RS_EPT / 4];
#pragma unroll
		for (int q = 0; q < RS_EPT / 4; q++) v[q] = kv[q * MSX_BLOCK + threadIdx.x];
#pragma unroll
		for (int q = 0; q < RS_EPT / 4; q++) {
			if (!RS_SKIPPED(v[q].x)) atomicAdd(&s_cnt[w][(v[q].x >> shift) & dmask], 1u);
			if (!RS_SKIPPED(v[q].y)) atomicAdd(&s_cnt[w][(v[q].y >> shift) & dmask], 1u);
			if (!RS_SKIPPED(v[q].z)) atomicAdd(&s_cnt[w][(v[q].z >> shift) & dmask], 1u);
			if (!RS_SKIPPED(v[q].w)) atomicAdd(&s_cnt[w][(v[q].w >> shift) & dmask], 1u);
		}
	} else if (SKIP) {
		for (int q = 0; q < RS_EPT; q++) {
			const int64_t k = base + q * MSX_BLOCK + threadIdx.x;
			if (k < E) {
				const uint32_t key = keys[k];
				if (!RS_SKIPPED(key)) atomicAdd(&s_cnt[w][(key >> shift) & dmask], 1u);
			}
		}
	} else if (base + RS_TILE <= E) {
		const uint4 *kv = reinterpret_cast<const uint4 *>(keys + base);
		uint4 v[RS_EPT / 4];
#pragma unroll
		for (int q = 0; q < RS_EPT / 4; q++) v[q] = kv[q * MSX_BLOCK + threadIdx.x];
#pragma unroll
		for (int q = 0; q < RS_EPT / 4; q++) {
			hist_add(s_cnt[w], (v[q].x >> shift) & dmask, true);
			hist_add(s_cnt[w], (v[q].y >> shift) & dmask, true);
			hist_add(s_cnt[w], (v[q].z >> shift) & dmask, true);
			hist_add(s_cnt[w], (v[q].w >> shift) & dmask, true);
		}
	} else {
		for (int q = 0; q < RS_EPT; q++) {
			const int64_t k = base + q * MSX_BLOCK + threadIdx.x;
			const bool ok = k < E;
			hist_add(s_cnt[w], ok ? ((keys[k] >> shift) & dmask) : 0u, ok);
		}
	}
	__syncthreads();
	const int d = threadIdx.x;
	hist[(int64_t)d * n_tiles + tile] = s_cnt[0][d] + s_cnt[1][d] + s_cnt[2][d] + s_cnt[3][d];
}

// the digit counts of 8-bit values that are digits themselves (msx_sort_k32v8's first pass): 16 of them per load
__global__ __launch_bounds__(MSX_BLOCK) void k_rs_hist8(const uint8_t *__restrict__ vals, int64_t n, uint32_t *__restrict__ hist, int64_t n_tiles,
                                                        int64_t tile0) {
	__shared__ uint32_t s_cnt[MSX_BLOCK / 64][256];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int64_t tile = blockIdx.x + tile0, base = tile * RS_TILE;
	for (int q = 0; q < 4; q++) s_cnt[w][lane + 64 * q] = 0;
	if (base + RS_TILE <= n) {
		const uint4 v = reinterpret_cast<const uint4 *>(vals + base)[threadIdx.x];
		const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
		for (int q = 0; q < 4; q++)
#pragma unroll
			for (int b = 0; b < 4; b++) hist_add(s_cnt[w], (x[q] >> (8 * b)) & 255u, true);
	} else {
		for (int q = 0; q < RS_EPT; q++) {
			const int64_t k = base + q * MSX_BLOCK + threadIdx.x;
			const bool ok = k < n;
			hist_add(s_cnt[w], ok ? (uint32_t)vals[k] : 0u, ok);
		}
	}
	__syncthreads();
	const int d = threadIdx.x;
	hist[(int64_t)d * n_tiles + tile] = s_cnt[0][d] + s_cnt[1][d] + s_cnt[2][d] + s_cnt[3][d];
}

// The two passes of msx_sort_k32v8 need not be stable (what is ordered are marks that are added up), so a key's place
// inside its digit's run of the tile is whatever an LDS counter hands out: one atomic with return per key instead of the
// eight ballots and their mask arithmetic per row of 64 -- k_rs_scatter issues ~120 vector instructions per row and is
// bound by that, not by memory.  When most of a row holds one digit (a hot reference: its bucket, its tile) those lanes
// are counted by a ballot and one add.
//   DV : the digit is the 8-bit value beside the key; keys only are written, every value's run (a BUCKET) beginning at a
//        whole tile (the slots in between: k_rs_bucket_pad);
//   SEG: the pass inside the buckets, by key bits [shift, shift + 8): a tile lies in one bucket (btot[]: the buckets'
//        lengths); its digit runs go to  bucket start + (the bucket's keys of smaller digits) + (the digit's keys in the
//        bucket's earlier tiles) -- all from the row-wise scans of the digit counts (hoff) at the bucket's first tile,
//        at this tile and at the next bucket's first tile.  Tiles behind the last bucket and those of bucket
//        `skip_bucket` have nothing to do.
//   SKIPK: keys only, keys with bit 31 set are not part of the input (msx_count_keys: the per-pool word of msx_count.h), hoff
//        holds one scan over the whole table (digit-major), no dtot / btot
template <bool DV, bool SEG, bool SKIPK = false>
__global__ __launch_bounds__(MSX_BLOCK) void k_rs_scatter_any(const uint32_t *__restrict__ keys_in, const uint8_t *__restrict__ vals_in,
                                                              uint32_t *__restrict__ keys_out, int64_t n, int shift,
                                                              const uint32_t *__restrict__ hoff, int64_t n_tiles,
                                                              const uint32_t *__restrict__ dtot, const uint32_t *__restrict__ btot,
                                                              int skip_bucket) {
	__shared__ uint32_t s_key[RS_TILE];
	__shared__ uint8_t s_dig[DV ? RS_TILE : 1];       // DV: the digit of the key at each place of the ordered tile
	__shared__ uint4 s_v8[DV ? RS_TILE / 16 : 1];
	__shared__ uint32_t s_cnt[256], s_dstart[256], s_gbase[256];
	__shared__ uint32_t s_wsum[MSX_BLOCK / 64], s_gsum[MSX_BLOCK / 64], s_bfirst, s_bnext, s_nvalid;
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int64_t tile = blockIdx.x, base = tile * RS_TILE;
	if (base >= n) return;
	const uint32_t n_here = (uint32_t)((n - base) < (int64_t)RS_TILE ? (n - base) : (int64_t)RS_TILE);
	if (SEG) {                                        // which bucket is this tile in?  (thread d: bucket d)
		const int d = threadIdx.x;
		const uint32_t bp = (btot[d] + RS_TILE - 1) / RS_TILE;
		uint32_t binc = bp;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t t = __shfl_up(binc, o, 64);
			if (lane >= o) binc += t;
		}
		if (lane == 63) s_wsum[w] = binc;
		if (d == 0) s_bfirst = 0xffffffffu;
		__syncthreads();
		for (int q = 0; q < w; q++) binc += s_wsum[q];
		if ((uint32_t)tile >= binc - bp && (uint32_t)tile < binc && d != skip_bucket) { s_bfirst = binc - bp; s_bnext = binc; }
		__syncthreads();
		if (s_bfirst == 0xffffffffu) return;
	}
	s_cnt[threadIdx.x] = 0;
	uint32_t key[RS_EPT];
	const uint32_t wbase = (uint32_t)w * (RS_EPT * 64) + (uint32_t)lane;
	const bool v8 = DV && n_here == RS_TILE;          // the tile's values as one 16-byte load per thread, handed to their rows through LDS
	if (v8) s_v8[threadIdx.x] = reinterpret_cast<const uint4 *>(vals_in + base)[threadIdx.x];
#pragma unroll
	for (int r = 0; r < RS_EPT; r++) {
		const uint32_t i = wbase + (uint32_t)r * 64u;
		key[r] = i < n_here ? keys_in[base + i] : 0u;
	}
	__syncthreads();
	const unsigned long long lt = (1ull << lane) - 1ull;
	uint32_t pos[RS_EPT], dig[RS_EPT];
#pragma unroll
	for (int r = 0; r < RS_EPT; r++) {
		const uint32_t i = wbase + (uint32_t)r * 64u;
		const bool valid = i < n_here && !(SKIPK && RS_SKIPPED(key[r]));
		uint32_t d;
		if (DV) d = v8 ? (uint32_t) reinterpret_cast<const uint8_t *>(s_v8)[i] : (valid ? (uint32_t)vals_in[base + i] : 0u);
		else d = (key[r] >> shift) & 255u;
		dig[r] = d;
		const unsigned long long act = __ballot(valid);
		uint32_t p = 0;
		if (act) {
			const int lead = __ffsll((long long)act) - 1;
			const uint32_t d0 = (uint32_t)__shfl((int)d, lead, 64);
			const unsigned long long same = __ballot(valid && d == d0);
			uint32_t b0 = 0;
			if (lane == lead) b0 = atomicAdd(&s_cnt[d0], (uint32_t)__popcll(same));
			b0 = (uint32_t)__shfl((int)b0, lead, 64);
			if (valid && d == d0) p = b0 + (uint32_t)__popcll(same & lt);
			else if (valid) p = atomicAdd(&s_cnt[d], 1u);
		}
		pos[r] = p;
	}
	__syncthreads();
	{
		const int d = threadIdx.x;
		const uint32_t tot = s_cnt[d];
		uint32_t inc = tot;                                // inclusive scan over the 256 digits: where each begins in the tile
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t t = __shfl_up(inc, o, 64);
			if (lane >= o) inc += t;
		}
		// ... and in the output: over the digits' padded totals (DV), or over their totals inside the bucket (SEG)
		uint32_t p_first = 0;
		if (SEG) p_first = hoff[(int64_t)d * n_tiles + s_bfirst];
		const uint32_t gt = SKIPK ? 0u
		                  : SEG ? ((s_bnext < (uint32_t)n_tiles ? hoff[(int64_t)d * n_tiles + s_bnext] : dtot[d]) - p_first)
		                        : ((dtot[d] + RS_TILE - 1) & ~(uint32_t)(RS_TILE - 1));
		uint32_t ginc = gt;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t t = __shfl_up(ginc, o, 64);
			if (lane >= o) ginc += t;
		}
		if (lane == 63) { s_wsum[w] = inc; s_gsum[w] = ginc; }
		__syncthreads();
		uint32_t woff = 0, goff = 0;
		for (int q = 0; q < w; q++) { woff += s_wsum[q]; goff += s_gsum[q]; }
		const uint32_t ds = woff + inc - tot;
		s_dstart[d] = ds;
		s_gbase[d] = hoff[(int64_t)d * n_tiles + tile] + goff + ginc - gt - ds + (SEG ? s_bfirst * RS_TILE - p_first : 0u);
		if (d == 255) s_nvalid = ds + tot;
	}
	__syncthreads();
#pragma unroll
	for (int r = 0; r < RS_EPT; r++) {
		if (wbase + (uint32_t)r * 64u < n_here && !(SKIPK && RS_SKIPPED(key[r]))) {
			const uint32_t p = s_dstart[dig[r]] + pos[r];
			s_key[p] = key[r];
			if (DV) s_dig[p] = (uint8_t)dig[r];
		}
	}
	__syncthreads();
#pragma unroll
	for (int q = 0; q < RS_EPT; q++) {
		const uint32_t p = (uint32_t)q * MSX_BLOCK + threadIdx.x;
		if (p < (SKIPK ? s_nvalid : n_here)) {
			const uint32_t k = s_key[p];
			keys_out[s_gbase[DV ? (uint32_t)s_dig[p] : ((k >> shift) & 255u)] + p] = k;
		}
	}
}

// msx_sort_k32v8 between its passes: the slots between a bucket's keys and the next bucket's first tile, and everything
// behind the last bucket up to n_ub, become empty keys (all ones); lay[d] = where bucket d begins, lay[256 + d] = its keys
__global__ __launch_bounds__(MSX_BLOCK) void k_rs_bucket_pad(const uint32_t *__restrict__ btot, uint32_t *__restrict__ keys, int64_t n_ub,
                                                             uint32_t *__restrict__ lay) {
	__shared__ uint32_t s_wsum[MSX_BLOCK / 64];
	__shared__ uint32_t s_lo, s_hi;
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, d = threadIdx.x;
	const uint32_t cnt = btot[d], padded = (cnt + RS_TILE - 1) & ~(uint32_t)(RS_TILE - 1);
	uint32_t inc = padded;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t t = __shfl_up(inc, o, 64);
		if (lane >= o) inc += t;
	}
	if (lane == 63) s_wsum[w] = inc;
	__syncthreads();
	for (int q = 0; q < w; q++) inc += s_wsum[q];
	if (blockIdx.x == 0) { lay[d] = inc - padded; lay[256 + d] = cnt; }
	if (d == (int)blockIdx.x) { s_lo = inc - padded + cnt; s_hi = inc; }
	if (blockIdx.x >= 256 && d == 255) {                     // the tail, a slice per workgroup
		const uint32_t parts = gridDim.x - 256, part = blockIdx.x - 256, len = (uint32_t)n_ub - inc;
		const uint32_t per = ((len + parts - 1) / parts + 3u) & ~3u;
		s_lo = inc + (part * per < len ? part * per : len);
		s_hi = inc + ((part + 1) * per < len ? (part + 1) * per : len);
	}
	__syncthreads();
	for (uint32_t q = s_lo + threadIdx.x; q < s_hi; q += MSX_BLOCK) keys[q] = 0xffffffffu;
}

// Between the two: the per-tile digit counts become positions.  hist[] is laid out digit-major, so the
// tiles of one digit are a contiguous row: k_rs_rowscan scans every row by itself (256 workgroups, none
// waiting for another) and leaves the rows' totals in dtot[256]; the scatter kernel adds the exclusive
// prefix over those 256 totals itself.  One launch where a scan over the whole table took three.
__global__ __launch_bounds__(MSX_BLOCK) void k_rs_rowscan(const uint32_t *__restrict__ hist,
                                                          const unsigned long long *__restrict__ n_ptr, int64_t n_tiles,
                                                          uint32_t *__restrict__ off, uint32_t *__restrict__ dtot) {
	__shared__ uint32_t s_w[MSX_BLOCK / 64];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	if (n_ptr) n_tiles = ((int64_t)*n_ptr + RS_TILE - 1) / RS_TILE;
	const uint32_t *row = hist + (int64_t)blockIdx.x * n_tiles;
	uint32_t *out = off + (int64_t)blockIdx.x * n_tiles;
	uint32_t running = 0;
	for (int64_t base = 0; base < n_tiles; base += MSX_BLOCK * 8) {
		const int64_t i0 = base + (int64_t)threadIdx.x * 8;
		uint32_t v[8], s = 0;
#pragma unroll
		for (int q = 0; q < 8; q++) { v[q] = i0 + q < n_tiles ? row[i0 + q] : 0u; s += v[q]; }
		uint32_t inc = s;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t t = __shfl_up(inc, o, 64);
			if (lane >= o) inc += t;
		}
		if (lane == 63) s_w[w] = inc;
		__syncthreads();
		uint32_t woff = 0, tot = 0;
		for (int q = 0; q < MSX_BLOCK / 64; q++) { if (q < w) woff += s_w[q]; tot += s_w[q]; }
		uint32_t run = running + woff + inc - s;
#pragma unroll
		for (int q = 0; q < 8; q++) {
			if (i0 + q < n_tiles) out[i0 + q] = run;
			run += v[q];
		}
		running += tot;
		__syncthreads();
	}
	if (threadIdx.x == 0) dtot[blockIdx.x] = running;
}

// V: type of the value travelling with each key (uint32_t or unsigned long long); PAIRS = false: keys only;
// ROWS: hoff holds row-wise scans (k_rs_rowscan) and dtot the rows' totals, instead of one scan over the table
template <typename V, bool PAIRS, bool SKIP, bool ROWS = false>
__global__ __launch_bounds__(MSX_BLOCK) void k_rs_scatter(const uint32_t *__restrict__ keys_in,
                                                          const V *__restrict__ vals_in,
                                                          uint32_t *__restrict__ keys_out,
                                                          V *__restrict__ vals_out,
                                                          const unsigned long long *__restrict__ n_ptr, int64_t n_host,
                                                          int shift, uint32_t dmask,
                                                          const uint32_t *__restrict__ hoff, int64_t n_tiles,
                                                          const uint32_t *__restrict__ dtot = nullptr) {
	__shared__ uint32_t s_key[RS_TILE];
	__shared__ V s_val[PAIRS ? RS_TILE : 1];
	__shared__ uint32_t s_nvalid;
	__shared__ uint32_t s_cnt[MSX_BLOCK / 64][256];   // per wave: running digit counts, then the wave's offset
	__shared__ unsigned long long s_mask[MSX_BLOCK / 64][256];   // per wave and digit: the lanes of the current row that hold it
	__shared__ uint32_t s_dstart[256];                // first position of the digit inside the sorted tile
	__shared__ uint32_t s_gbase[256];                 // global position of the digit's run minus s_dstart
	__shared__ uint32_t s_wsum[MSX_BLOCK / 64], s_gsum[MSX_BLOCK / 64];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int64_t tile = blockIdx.x;
	const int64_t E = n_ptr ? (int64_t)*n_ptr : n_host;
	const int64_t base = tile * RS_TILE;
	if (base >= E) return;
	if (n_ptr) n_tiles = (E + RS_TILE - 1) / RS_TILE;      // the table's stride, as k_rs_hist laid it out
	const uint32_t n_here = (uint32_t)((E - base) < (int64_t)RS_TILE ? (E - base) : (int64_t)RS_TILE);
	for (int q = 0; q < 4; q++) { s_cnt[w][lane + 64 * q] = 0; s_mask[w][lane + 64 * q] = 0ull; }
	// 1. all loads up front; wave w owns elements [w*RS_EPT*64, (w+1)*RS_EPT*64) of the tile
	uint32_t key[RS_EPT];
	V val[RS_EPT];
	const uint32_t wbase = (uint32_t)w * (RS_EPT * 64) + (uint32_t)lane;
#pragma unroll
	for (int r = 0; r < RS_EPT; r++) {
		const uint32_t i = wbase + (uint32_t)r * 64u;
		key[r] = SKIP ? RS_NOKEY : 0u; val[r] = 0;
		if (i < n_here) {
			key[r] = keys_in[base + i];
			if (PAIRS) val[r] = vals_in[base + i];
		}
	}
	// 2. rank inside the wave
	const unsigned long long lt = (1ull << lane) - 1ull;
	uint32_t pos[RS_EPT];
#pragma unroll
	for (int r = 0; r < RS_EPT; r++) {
		const bool valid = (wbase + (uint32_t)r * 64u < n_here) && (!SKIP || !RS_SKIPPED(key[r]));
		const uint32_t d = (key[r] >> shift) & dmask;
		// m: the lanes of this row that hold the same digit.  The lanes that share the first valid lane's digit know it
		// from one ballot; the others OR their lane bit into an LDS word per digit and read the word back (the wave's
		// LDS operations complete in order), the lowest lane of each clears it for the next row.  (Eight ballots and
		// the mask arithmetic per row were ~50 of the kernel's ~120 vector instructions per row, and the kernel is
		// bound by what it issues.)
		unsigned long long m = 0;
		{
			const unsigned long long act = __ballot(valid);
			const uint32_t d0 = (uint32_t)__shfl((int)d, act ? __ffsll((long long)act) - 1 : 0, 64);
			const unsigned long long same = __ballot(valid && d == d0);
			if (valid && d == d0) m = same;
			else if (valid) {
				__hip_atomic_fetch_or(&s_mask[w][d], 1ull << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				m = __hip_atomic_load(&s_mask[w][d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				if ((m & lt) == 0ull) __hip_atomic_store(&s_mask[w][d], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		}
		const uint32_t rank = (uint32_t)__popcll(m & lt);
		const uint32_t cnt = (uint32_t)__popcll(m);
		uint32_t p0 = 0;
		if (valid) p0 = s_cnt[w][d];                       // every lane reads before the leader writes
		if (valid && rank == 0) s_cnt[w][d] = p0 + cnt;
		pos[r] = p0 + rank;
	}
	__syncthreads();
	// 3. digit d = threadIdx.x: offsets of the four waves inside the digit, digit starts inside the tile
	{
		const int d = threadIdx.x;
		const uint32_t c0 = s_cnt[0][d], c1 = s_cnt[1][d], c2 = s_cnt[2][d], c3 = s_cnt[3][d];
		const uint32_t tot = c0 + c1 + c2 + c3;
		uint32_t inc = tot;                                // inclusive scan over the 256 digits
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t t = __shfl_up(inc, o, 64);
			if (lane >= o) inc += t;
		}
		// (ROWS: the same scan over the digits' global totals gives the position where each digit's run begins)
		const uint32_t gt = ROWS ? dtot[d] : 0u;
		uint32_t ginc = gt;
		if (ROWS) {
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) {
				const uint32_t t = __shfl_up(ginc, o, 64);
				if (lane >= o) ginc += t;
			}
		}
		if (lane == 63) { s_wsum[w] = inc; if (ROWS) s_gsum[w] = ginc; }
		s_cnt[0][d] = 0; s_cnt[1][d] = c0; s_cnt[2][d] = c0 + c1; s_cnt[3][d] = c0 + c1 + c2;
		__syncthreads();
		uint32_t woff = 0, goff = 0;
		for (int q = 0; q < w; q++) { woff += s_wsum[q]; if (ROWS) goff += s_gsum[q]; }
		const uint32_t ds = woff + inc - tot;
		s_dstart[d] = ds;
		s_gbase[d] = hoff[(int64_t)d * n_tiles + tile] + (ROWS ? goff + ginc - gt : 0u) - ds;
		if (d == 255) s_nvalid = ds + tot;
	}
	__syncthreads();
	// 4. the tile in digit order, in LDS
#pragma unroll
	for (int r = 0; r < RS_EPT; r++) {
		if ((wbase + (uint32_t)r * 64u < n_here) && (!SKIP || !RS_SKIPPED(key[r]))) {
			const uint32_t d = (key[r] >> shift) & dmask;
			const uint32_t p = s_dstart[d] + s_cnt[w][d] + pos[r];
			s_key[p] = key[r];
			if (PAIRS) s_val[p] = val[r];
		}
	}
	__syncthreads();
	// 5. out: position p of the sorted tile goes to its digit's global run
	const uint32_t n_out = SKIP ? s_nvalid : n_here;
#pragma unroll
	for (int q = 0; q < RS_EPT; q++) {
		const uint32_t p = (uint32_t)q * MSX_BLOCK + threadIdx.x;
		if (p < n_out) {
			const uint32_t k = s_key[p];
			const uint32_t dst = s_gbase[(k >> shift) & dmask] + p;
			keys_out[dst] = k;
			if (PAIRS) vals_out[dst] = s_val[p];
		}
	}
}

// ---------------------------------------------------------------------------
// Per-reference insert counts without one global atomic per insert: the keys
// (feature of every uniquely mapped insert, RS_NOKEY for the other pools) are
// partitioned by their high bits in one radix pass, then every partition --
// <= PC_RANGE consecutive features -- is counted in LDS and added to ui[] with
// one write per feature.  (Scattered atomics run memory-side at ~27 G/s on this
// chip whatever their locality; 20 M of them cost more than the two passes.)
// ---------------------------------------------------------------------------
#define PC_RANGE 8192          // features per partition at most (32 KB of LDS counters)
#define PC_SPLIT 64            // a partition is shared by at most this many workgroups
#define PC_CHUNK 32768u        // keys per workgroup before a partition is split

__global__ __launch_bounds__(MSX_BLOCK) void k_part_count(const uint32_t *__restrict__ keys,
                                                          const uint32_t *__restrict__ hoff, int64_t n_tiles,
                                                          int shift, int32_t nf, uint32_t add,
                                                          uint32_t *__restrict__ ui) {
	__shared__ uint32_t s_cnt[PC_RANGE];
	const uint32_t d = blockIdx.y, b = blockIdx.x;
	const uint32_t start = hoff[(int64_t)d * n_tiles], end = hoff[(int64_t)(d + 1) * n_tiles];
	const uint32_t size = end - start;
	uint32_t chunk = (size + PC_SPLIT - 1) / PC_SPLIT;
	if (chunk < PC_CHUNK) chunk = PC_CHUNK;
	const uint64_t lo64 = (uint64_t)start + (uint64_t)b * chunk;
	if (lo64 >= end) return;
	const uint32_t lo = (uint32_t)lo64, hi = (end - lo > chunk) ? lo + chunk : end;
	const uint32_t range = 1u << shift, mask = range - 1u;
	for (uint32_t i = threadIdx.x; i < range; i += MSX_BLOCK) s_cnt[i] = 0;
	__syncthreads();
	uint32_t k = lo + threadIdx.x;
	for (; k + 3u * MSX_BLOCK < hi; k += 4u * MSX_BLOCK) {
		const uint32_t k0 = keys[k], k1 = keys[k + MSX_BLOCK], k2 = keys[k + 2 * MSX_BLOCK], k3 = keys[k + 3 * MSX_BLOCK];
		atomicAdd(&s_cnt[k0 & mask], add);
		atomicAdd(&s_cnt[k1 & mask], add);
		atomicAdd(&s_cnt[k2 & mask], add);
		atomicAdd(&s_cnt[k3 & mask], add);
	}
	for (; k < hi; k += MSX_BLOCK) atomicAdd(&s_cnt[keys[k] & mask], add);
	__syncthreads();
	const bool alone = size <= chunk;                      // this workgroup owns the whole partition
	for (uint32_t i = threadIdx.x; i < range; i += MSX_BLOCK) {
		const uint32_t c = s_cnt[i];
		const uint32_t f = (d << shift) + i;
		if (c && f < (uint32_t)nf) {
			if (alone) ui[f] += c;
			else atomicAdd(&ui[f], c);
		}
	}
}

// ui[key] += add for every key without bit 31 among keys[0..n); such keys < nf <= 256 * PC_RANGE.
// key2 = scratch of n keys.
int msx_count_keys(msx_ctx *ctx, msx_profile *p, const uint32_t *keys, uint32_t *key2, int64_t n, uint32_t add) {
	int shift = 0;
	while (((int64_t)256 << shift) < (int64_t)p->n_features) shift++;
	if ((1 << shift) > PC_RANGE) return msx_fail(ctx, MSX_ERR_ARG, "msx_count_keys: too many features");
	const int64_t n_tiles = (n + RS_TILE - 1) / RS_TILE;
	int rc;
	if ((rc = msx_reserve(ctx, &p->ck_hist, (size_t)(256 * n_tiles + 16) * 4))) return rc;
	if ((rc = msx_reserve(ctx, &p->ck_off, (size_t)(256 * n_tiles + 16) * 4))) return rc;
	MSX_TIMED(ctx, MSX_K_INSERT_COUNT,
	          hipLaunchKernelGGL(k_rs_hist<true>, dim3((unsigned)n_tiles), dim3(MSX_BLOCK), 0, ctx->stream, keys,
	                             (const unsigned long long *)nullptr, n, shift, 255u, (uint32_t *)p->ck_hist.p,
	                             n_tiles));
	if ((rc = msx_scan_u32(ctx, (const uint32_t *)p->ck_hist.p, (uint32_t *)p->ck_off.p, 256 * n_tiles))) return rc;
	msx_time_begin(ctx, MSX_K_INSERT_COUNT);
	// (the partition need not be stable -- what follows counts -- so a key's place comes from an LDS counter: k_rs_scatter_any)
	hipLaunchKernelGGL((k_rs_scatter_any<false, false, true>), dim3((unsigned)n_tiles), dim3(MSX_BLOCK), 0, ctx->stream, keys,
	                   (const uint8_t *)nullptr, key2, n, shift, (const uint32_t *)p->ck_off.p, n_tiles, (const uint32_t *)nullptr,
	                   (const uint32_t *)nullptr, -1);
	hipLaunchKernelGGL(k_part_count, dim3(PC_SPLIT, 256), dim3(MSX_BLOCK), 0, ctx->stream, (const uint32_t *)key2,
	                   (const uint32_t *)p->ck_off.p, n_tiles, shift, p->n_features, add, p->ui);
	msx_time_end(ctx);
	MSX_HIP(ctx, hipGetLastError());
	return MSX_OK;
}

// --- derived multi-mapper store: renumbered for locality, duplicates merged ---------
// The sharing iteration gathers a[feature] per list and recip[list] per feature
// entry.  Inserts that multi-map inside one family of similar references share
// features, so (1) ordering the lists by their smallest feature id puts the lists
// a feature belongs to next to each other (dense cache lines instead of one line
// per 8-byte gather), and (2) many inserts hit exactly the same set of references:
// identical lists are merged into one list with a weight w (it contributes
// w/S instead of w times 1/S).  Sort key = smallest feature in the high bits (the
// locality order) and an order-independent hash of the set in the low bits, so
// equal sets end up adjacent; adjacency is then verified exactly, set against
// set.  The accumulated store (m_off/m_fid) is left untouched for later batches.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
	x ^= x >> 16; x *= 0x7feb352du;
	x ^= x >> 15; x *= 0x846ca68bu;
	x ^= x >> 16;
	return x;
}

// Signature of a list: sets of <= 3 features (the common case) are packed exactly --
// three 21-bit fields in ascending order, unused fields = SIG_PAD -- so that equal
// signatures mean equal sets and the set itself can be rebuilt from the signature;
// anything else gets bit 63, a 31-bit hash of the set and, in the low half, the number of the
// list (so that it can be found again after the sort), and is compared entry by entry.
#define SIG_PAD 0x1fffffu
#define SIG_HASHED (1ull << 63)

__device__ __forceinline__ uint32_t sig_len(unsigned long long sg) {
	return 1u + (((sg >> 21) & SIG_PAD) != SIG_PAD) + (((sg >> 42) & SIG_PAD) != SIG_PAD);
}

__global__ __launch_bounds__(MSX_BLOCK) void k_list_key(const unsigned long long *__restrict__ csr_tot,
                                                        const uint32_t *__restrict__ m_off,
                                                        const int32_t *__restrict__ m_fid, int hash_bits, int coarse, int fbits,
                                                        uint32_t *__restrict__ key,
                                                        unsigned long long *__restrict__ sig) {
	const int64_t n_lists = (int64_t)csr_tot[0];
	const int64_t stride = (int64_t)gridDim.x * MSX_BLOCK;
	// one list: its bounds and (up to three features) its features have been fetched by the caller
	auto one = [&](int64_t j, uint32_t s, uint32_t e, uint32_t p0, uint32_t p1, uint32_t p2) {
		uint32_t mn = 0xffffffffu, h = (e - s) * 0x9e3779b9u, h2 = (e - s) * 0x85ebca6bu;
		unsigned long long sg;
		if (e - s <= 3u) {
			// a 3-element sorting network
			const uint32_t f0 = e - s > 0u ? p0 : SIG_PAD, f1 = e - s > 1u ? p1 : SIG_PAD, f2 = e - s > 2u ? p2 : SIG_PAD;
			const bool fits = (e - s > 0u) && f0 < SIG_PAD && (e - s < 2u || f1 < SIG_PAD) && (e - s < 3u || f2 < SIG_PAD);
			if (e - s > 0u) { h += mix32(f0); h2 += mix32(f0 ^ 0x5bd1e995u); }
			if (e - s > 1u) { h += mix32(f1); h2 += mix32(f1 ^ 0x5bd1e995u); }
			if (e - s > 2u) { h += mix32(f2); h2 += mix32(f2 ^ 0x5bd1e995u); }
			uint32_t a = f0, b = f1, c = f2, t;
			if (a > b) { t = a; a = b; b = t; }
			if (b > c) { t = b; b = c; c = t; }
			if (a > b) { t = a; a = b; b = t; }
			mn = a;
			sg = fits ? ((unsigned long long)a | ((unsigned long long)b << 21) | ((unsigned long long)c << 42))
			          : (SIG_HASHED | ((unsigned long long)(h2 >> 1) << 32) | (unsigned long long)(uint32_t)j);
			if (e == s) mn = 0xffffffffu;
		} else {
			for (uint32_t k = s; k < e; ++k) {
				const uint32_t f = (uint32_t)m_fid[k];
				mn = f < mn ? f : mn;
				h += mix32(f);                                   // commutative: a hash of the set
				h2 += mix32(f ^ 0x5bd1e995u);
			}
			sg = SIG_HASHED | ((unsigned long long)(h2 >> 1) << 32) | (unsigned long long)(uint32_t)j;
		}
		const uint32_t hb = hash_bits > 0 ? (h & ((1u << hash_bits) - 1u)) : 0u;
		// hash_bits < 0: the whole key is a hash of the set (-hash_bits bits of it): no order by smallest feature, but
		// equal sets meet whatever else shares their smallest feature
		if (hash_bits < 0) {
			// (experiment) -hash_bits key bits in all: the top `coarse` bits of the smallest feature, a hash of the set below
			const int kb = -hash_bits, hbits = kb - coarse;
			const uint32_t hh = mix32(h ^ (h2 * 0x9e3779b9u)) >> (32 - hbits);
			const uint32_t top = coarse > 0 ? ((mn == 0xffffffffu ? 0u : mn) >> (fbits > coarse ? fbits - coarse : 0)) : 0u;
			key[j] = (top << hbits) | hh;
		} else {
			key[j] = hash_bits > 0 ? ((mn << hash_bits) | hb) : mn;
		}
		sig[j] = sg;                                          // travels through the sort as the value
	};
	// two lists per thread at a time: both lists' bounds, then both lists' features, then the arithmetic (a list after the
	// other left the two dependent round trips of each exposed)
	for (int64_t j = (int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x; j < n_lists; j += 2 * stride) {
		const int64_t j2 = j + stride;
		const bool two = j2 < n_lists;
		const uint32_t s0 = m_off[j], e0 = m_off[j + 1];
		uint32_t s1 = 0, e1 = 0;
		if (two) { s1 = m_off[j2]; e1 = m_off[j2 + 1]; }
		uint32_t a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0;
		if (e0 - s0 <= 3u) {
			if (e0 - s0 > 0u) a0 = (uint32_t)m_fid[s0];
			if (e0 - s0 > 1u) a1 = (uint32_t)m_fid[s0 + 1];
			if (e0 - s0 > 2u) a2 = (uint32_t)m_fid[s0 + 2];
		}
		if (two && e1 - s1 <= 3u) {
			if (e1 - s1 > 0u) b0 = (uint32_t)m_fid[s1];
			if (e1 - s1 > 1u) b1 = (uint32_t)m_fid[s1 + 1];
			if (e1 - s1 > 2u) b2 = (uint32_t)m_fid[s1 + 2];
		}
		one(j, s0, e0, a0, a1, a2);
		if (two) one(j2, s1, e1, b0, b1, b2);
	}
}

// hl[i] = MSX_PINFO_LIST | length when the list at sorted position i is not the same set as its predecessor
// (a head), MSX_PINFO_NONE for a merged duplicate: the per-pool word of msx_count.h, so that the same
// chunk-sum scan serves.
// ssig = the signatures in sorted order (the values of the list sort): no gather is needed
__global__ __launch_bounds__(MSX_BLOCK) void k_dup_mark(const unsigned long long *__restrict__ csr_tot, int64_t m,
                                                        const uint32_t *__restrict__ skey,
                                                        const unsigned long long *__restrict__ ssig,
                                                        const uint32_t *__restrict__ m_off,
                                                        const int32_t *__restrict__ m_fid,
                                                        uint32_t *__restrict__ hl) {
	const int64_t n_lists = (int64_t)csr_tot[0];
	const int64_t stride = (int64_t)gridDim.x * MSX_BLOCK;
	const int lane = threadIdx.x & 63;
	// (positions at and beyond n_lists are not written: the scans that follow stop at n_lists)
	for (int64_t i0 = (int64_t)blockIdx.x * MSX_BLOCK; i0 < n_lists && i0 < m; i0 += stride) {
		const int64_t i = i0 + threadIdx.x;
		const bool in = i < n_lists;
		unsigned long long sg = 0;
		if (in) sg = ssig[i];
		// the predecessor's signature: the lane below has it, except for lane 0
		unsigned long long prev = __shfl_up(sg, 1, 64);
		if (in && lane == 0 && i > 0) prev = ssig[i - 1];
		uint32_t hd = 0, l = 0;
		if (in) {
			hd = 1;
			if (!(sg & SIG_HASHED)) {
				l = sig_len(sg);
				if (i > 0 && sg == prev) hd = 0;                  // exact: equal signature = equal set
			} else {
				const uint32_t j = (uint32_t)sg;
				const uint32_t s = m_off[j], e = m_off[j + 1];
				l = e - s;
				if (i > 0 && (sg >> 32) == (prev >> 32) && skey[i] == skey[i - 1] && l <= 32u) {
					const uint32_t jp = (uint32_t)prev;
					const uint32_t sp = m_off[jp], ep = m_off[jp + 1];
					if (ep - sp == l) {
						// both lists hold distinct features, so equal sizes + inclusion = equal sets
						bool same = true;
						for (uint32_t k = s; k < e && same; ++k) {
							const int32_t f = m_fid[k];
							bool found = false;
							for (uint32_t q = sp; q < ep; ++q) found |= (m_fid[q] == f);
							same = found;
						}
						if (same) hd = 0;
					}
				}
			}
		}
		if (in) hl[i] = hd ? (MSX_PINFO_LIST | l) : MSX_PINFO_NONE;
	}
}

// unique list u (sorted position i, a head): offsets, entries, and the sorted position itself
// (weights are differences of consecutive head positions)
__device__ __forceinline__ unsigned long long pack_others(uint32_t o1, uint32_t o2, uint32_t o3) {
	return (unsigned long long)o1 | ((unsigned long long)o2 << 21) | ((unsigned long long)o3 << 42);
}

// One workgroup per chunk of MSX_PINFO_CHUNK sorted positions: the number u of a head and the offset o of its
// entries are the chunk's base (msx_scan_pinfo_chunks over hl[]) plus a scan inside the workgroup; they go
// through LDS so that the writing runs one list per lane, neighbours side by side.
__global__ __launch_bounds__(MSX_BLOCK) void k_uniq_gather(const unsigned long long *__restrict__ csr_tot,
                                                           const uint32_t *__restrict__ hl,
                                                           const unsigned long long *__restrict__ chunk_base, int64_t n_chunks,
                                                           const unsigned long long *__restrict__ ssig,
                                                           const uint32_t *__restrict__ m_off,
                                                           const int32_t *__restrict__ m_fid,
                                                           uint32_t *__restrict__ d_off, int32_t *__restrict__ d_fid,
                                                           uint32_t *__restrict__ e_key,
                                                           unsigned long long *__restrict__ e_val,
                                                           uint32_t *__restrict__ hpos, int64_t m,
                                                           unsigned long long *__restrict__ d_tot) {
	__shared__ unsigned long long s_w[MSX_BLOCK / 64];
	__shared__ unsigned long long s_pos[MSX_PINFO_CHUNK];
	const int64_t n_lists = (int64_t)csr_tot[0];
	const int64_t c0 = (int64_t)blockIdx.x * MSX_PINFO_CHUNK;
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		const unsigned long long tot = chunk_base[n_chunks];
		const uint32_t U = (uint32_t)(tot >> 32), E2 = (uint32_t)tot;
		d_tot[0] = U;
		d_tot[1] = E2;
		d_tot[2] = 0;                  // general lists, counted by k_entry_weight
		d_off[U] = E2;                 // CSR sentinel
		hpos[U] = (uint32_t)n_lists;   // so that weight(u) = hpos[u+1] - hpos[u]
	}
	if (c0 >= n_lists) return;
	{
		const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
		const int64_t i0 = c0 + (int64_t)threadIdx.x * 8;
		unsigned long long v[8], sum = 0;
#pragma unroll
		for (int k = 0; k < 8; k++) {
			const uint32_t x = i0 + k < n_lists ? hl[i0 + k] : MSX_PINFO_NONE;
			v[k] = ((x & MSX_PINFO_LIST) && x != MSX_PINFO_NONE) ? ((1ull << 32) | (x & ~MSX_PINFO_LIST)) : 0ull;
			sum += v[k];
		}
		unsigned long long inc = sum;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const unsigned long long t = __shfl_up(inc, o, 64);
			if (lane >= o) inc += t;
		}
		if (lane == 63) s_w[w] = inc;
		__syncthreads();
		unsigned long long run = chunk_base[blockIdx.x] + inc - sum;
		for (int q = 0; q < w; q++) run += s_w[q];
#pragma unroll
		for (int k = 0; k < 8; k++) {
			s_pos[threadIdx.x * 8 + k] = v[k] ? run : ~0ull;
			run += v[k];
		}
		__syncthreads();
	}
	for (int r = 0; r < 8; r++) {
		const int64_t i = c0 + r * MSX_BLOCK + threadIdx.x;
		const unsigned long long sc = s_pos[r * MSX_BLOCK + threadIdx.x];
		if (sc == ~0ull) continue;
		const uint32_t u = (uint32_t)(sc >> 32);
		uint32_t o = (uint32_t)sc;
		const unsigned long long sg = ssig[i];
		d_off[u] = o;
		hpos[u] = (uint32_t)i;
		// A list of up to four features travels with its feature-major entries (k_share_reduce): the
		// entry of feature x carries the OTHER features of the list, ascending, three 21-bit fields.
		if (!(sg & SIG_HASHED)) {
			// the set is in the signature (ascending feature order)
			const uint32_t a = (uint32_t)(sg & SIG_PAD), b = (uint32_t)((sg >> 21) & SIG_PAD),
			               c = (uint32_t)((sg >> 42) & SIG_PAD);
			d_fid[o] = (int32_t)a; e_key[o] = a; e_val[o] = pack_others(b, c, SIG_PAD); o++;
			if (b != SIG_PAD) { d_fid[o] = (int32_t)b; e_key[o] = b; e_val[o] = pack_others(a, c, SIG_PAD); o++; }
			if (c != SIG_PAD) { d_fid[o] = (int32_t)c; e_key[o] = c; e_val[o] = pack_others(a, b, SIG_PAD); }
		} else {
			const uint32_t j = (uint32_t)sg;
			const uint32_t s = m_off[j], e = m_off[j + 1];
			bool four = (e - s == 4u);
			uint32_t f[4] = {SIG_PAD, SIG_PAD, SIG_PAD, SIG_PAD};
			if (four) {
#pragma unroll
				for (int k = 0; k < 4; k++) f[k] = (uint32_t)m_fid[s + k];
				four = f[0] < SIG_PAD && f[1] < SIG_PAD && f[2] < SIG_PAD && f[3] < SIG_PAD;
			}
			if (four) {
				// ascending copy (5-comparator network); entry k keeps the list's own order in d_fid
				uint32_t q0 = f[0], q1 = f[1], q2 = f[2], q3 = f[3], t;
				if (q0 > q1) { t = q0; q0 = q1; q1 = t; }
				if (q2 > q3) { t = q2; q2 = q3; q3 = t; }
				if (q0 > q2) { t = q0; q0 = q2; q2 = t; }
				if (q1 > q3) { t = q1; q1 = q3; q3 = t; }
				if (q1 > q2) { t = q1; q1 = q2; q2 = t; }
#pragma unroll
				for (int k = 0; k < 4; k++) {
					const uint32_t x = f[k];
					// the sorted list without x
					const uint32_t o1 = (x == q0) ? q1 : q0;
					const uint32_t o2 = (x == q0 || x == q1) ? q2 : q1;
					const uint32_t o3 = (x == q3) ? q2 : q3;
					d_fid[o] = (int32_t)x; e_key[o] = x; e_val[o] = pack_others(o1, o2, o3); o++;
				}
			} else {
				for (uint32_t k = s; k < e; ++k) {
					d_fid[o] = m_fid[k]; e_key[o] = (uint32_t)m_fid[k]; e_val[o] = SIG_HASHED | u; o++;
				}
			}
		}
	}
}

// The weight of a merged list (how many inserts had exactly this set) rides in the key bits
// above the feature id of its entries.  A list whose weight does not fit there, or whose set did
// not fit a signature (more than four features, feature ids of 21 bits and more), is a *general*
// list: its entries name the list (SIG_HASHED | u), its number goes to gl_idx[], k_general_recip
// computes recip[u] = w/S for it and k_share_reduce gathers that.
#define EW_STAGE 1024                  // general list numbers a workgroup collects before appending them
__global__ __launch_bounds__(MSX_BLOCK) void k_entry_weight(unsigned long long *__restrict__ d_tot,
                                                            const uint32_t *__restrict__ d_off,
                                                            const uint32_t *__restrict__ hpos, int bits,
                                                            uint32_t *__restrict__ e_key,
                                                            unsigned long long *__restrict__ e_val,
                                                            uint32_t *__restrict__ gl_idx) {
	// (one global add per workgroup: thousands of lanes adding 1 to the same word serialise at ~12 ns each)
	__shared__ uint32_t s_gl[EW_STAGE];
	__shared__ uint32_t s_n, s_base;
	if (threadIdx.x == 0) s_n = 0;
	__syncthreads();
	const int64_t n_lists = (int64_t)d_tot[0];
	const int64_t stride = (int64_t)gridDim.x * MSX_BLOCK;
	for (int64_t u0 = (int64_t)blockIdx.x * MSX_BLOCK; u0 < n_lists; u0 += stride) {
		const int64_t u = u0 + threadIdx.x;
		if (u < n_lists) {
			const uint32_t s = d_off[u], e = d_off[u + 1];
			const uint32_t w = hpos[u + 1] - hpos[u];
			const bool hashed = (e_val[s] & SIG_HASHED) != 0;
			// (w + 1: the all-ones key is the sentinel of k_share_reduce)
			const bool fits = bits < 32 && (((unsigned long long)w + 1ull) >> (32 - bits)) == 0ull;
			if (!hashed && fits) {
				for (uint32_t o = s; o < e; ++o) e_key[o] |= w << bits;
			} else {
				for (uint32_t o = s; o < e; ++o) e_val[o] = SIG_HASHED | (unsigned long long)u;
				s_gl[atomicAdd(&s_n, 1u)] = (uint32_t)u;     // (<= 256 per round: flushed below before it can overflow)
			}
		}
		__syncthreads();
		if (s_n > EW_STAGE - MSX_BLOCK || u0 + stride >= n_lists) {      // workgroup-uniform
			const uint32_t cnt = s_n;
			if (threadIdx.x == 0 && cnt) s_base = (uint32_t)atomicAdd(&d_tot[2], (unsigned long long)cnt);
			__syncthreads();
			for (uint32_t q = threadIdx.x; q < cnt; q += MSX_BLOCK) gl_idx[s_base + q] = s_gl[q];
			__syncthreads();
			if (threadIdx.x == 0) s_n = 0;
			__syncthreads();
		}
	}
}

// ---------------------------------------------------------------------------
// per-iteration kernels
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(MSX_BLOCK) void k_prop_begin(int32_t nf, const uint32_t *__restrict__ ui,
                                                          const double *__restrict__ d, double *__restrict__ U,
                                                          double *__restrict__ a, double *__restrict__ share,
                                                          double *__restrict__ delta, int32_t *iter_state) {
	const int64_t stride = (int64_t)gridDim.x * MSX_BLOCK;
	for (int64_t i = (int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x; i < nf; i += stride) {
		double u = 1.0 * ui[i] / 2;           // msam_profile.c:284-289
		if (d) u += d[i];                     // :303-308
		U[i] = u;
		a[i] = u;                             // :326
		share[i] = 0.0;
	}
	if (blockIdx.x == 0 && threadIdx.x < 20) delta[threadIdx.x] = 0.0;
	if (blockIdx.x == 0 && threadIdx.x == 0) { iter_state[0] = 0; iter_state[1] = 0; }
}

// recip[j] = w_j/S_j for every general list j (S_j = sum of a over its features, w_j the number of
// inserts it stands for; 0 when S_j == 0: msam_profile.c:358).  These are the lists of five and more
// features (0.6 % of the lists on the IGC-scale workload), reached through gl_idx[]: one lane per list.
struct RecipArgs {
	const unsigned long long *d_tot;
	const uint32_t *m_off;
	const int32_t *m_fid;
	const uint32_t *hpos;
	const uint32_t *gl_idx;
	const double *a;
	double *recip;
};

__device__ __forceinline__ void general_recip_body(const RecipArgs &G, int64_t first, int64_t stride) {
	const int64_t n = (int64_t)G.d_tot[2];
	for (int64_t i = first; i < n; i += stride) {
		const uint32_t j = G.gl_idx[i];
		const uint32_t s = G.m_off[j], e = G.m_off[j + 1];
		const uint32_t w = G.hpos[j + 1] - G.hpos[j];
		double sum = 0;
		if (e - s <= 8u) {
			// feature ids, then abundances, as independent loads; summed in list order
			int32_t f[8];
			double x[8];
#pragma unroll
			for (int q = 0; q < 8; q++) f[q] = (s + (uint32_t)q < e) ? G.m_fid[s + q] : -1;
#pragma unroll
			for (int q = 0; q < 8; q++) x[q] = (f[q] >= 0) ? G.a[f[q]] : 0.0;
#pragma unroll
			for (int q = 0; q < 8; q++)
				if (s + (uint32_t)q < e) sum += x[q];
		} else {
			for (uint32_t k = s; k < e; ++k) sum += G.a[G.m_fid[k]];
		}
		G.recip[j] = sum > 0 ? (double)w / sum : 0.0;
	}
}

// (a launch of its own before the first iteration only: afterwards the lists ride along with k_prop_finish)
__global__ __launch_bounds__(MSX_BLOCK) void k_general_recip(RecipArgs G, const int32_t *__restrict__ iter_state) {
	if (iter_state[0]) return;
	general_recip_body(G, (int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x, (int64_t)gridDim.x * MSX_BLOCK);
}

// share[f] = sum of w/S over the lists containing f -- a segmented sum over the
// feature-sorted entries.  k_share_reduce: the entries are split evenly over the W waves the
// launch holds at once (one contiguous chunk each, a multiple of SR_STEP, sized on the device from
// the true entry count: no second, partly filled round of waves); a wave walks its chunk in steps of
// SR_STEP entries: a lane sums SR_EPL consecutive entries itself, one segmented scan over the lanes
// joins the open ends, the open segment is carried in registers from step to step.  A feature
// whose entries all lie inside the chunk is written with a plain store.  The (at most two)
// segments cut by the chunk boundary are not added with atomics -- a hot feature spans hundreds
// of chunks and would serialise on one address -- but left as partial sums, two slots per wave in
// entry order, whose keys (part_key, fixed for a build: k_part_index) ascend.  Single GPU:
// k_prop_apply adds the slots that fall into its workgroup's feature range, in slot order.
// With a collective between the halves of an iteration (msx_profile_prop_local): k_partial_reduce
// folds the slots into share[] run by run, in slot order, without atomics.
#define SR_SENT 0xffffffffu
#define SR_EPL 4                       // consecutive entries summed by one lane
#define SR_STEP (64 * SR_EPL)          // entries per wave step
#define SR_WAVES_PER_SIMD 3            // 126 registers would allow 4; 3 measured faster (52 us against 55): ONE round of waves

struct SegRow {
	double v;         // after seg_scan: inclusive segmented sum
	bool tail;
	bool started;     // the segment holding this lane began inside this wave's chunk
};

// one 64-entry row: head/tail flags come from the global neighbours (pk/nk)
__device__ __forceinline__ SegRow seg_row(uint32_t key, double v, bool valid, uint32_t pk, uint32_t nk, int lane,
                                          double carry, bool carry_started) {
	const bool head = valid && (pk != key);
	SegRow r;
	r.tail = valid && (nk != key);
	if (lane == 0 && !head) v += carry;
	uint32_t f = (head || lane == 0) ? 1u : 0u;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const double ov = __shfl_up(v, d, 64);
		const uint32_t of = __shfl_up(f, d, 64);
		if (lane >= d && !f) { v += ov; f = of; }
	}
	const unsigned long long hb = __ballot(head);
	const unsigned long long upto = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
	r.started = ((hb & upto) == 0ull) ? carry_started : true;
	r.v = v;
	return r;
}

// entries per wave: the E entries in W equal chunks, rounded up to whole steps
__device__ __forceinline__ int64_t sr_chunk(int64_t E, int64_t W) {
	const int64_t per = (E + W - 1) / W;
	return (per + SR_STEP - 1) / SR_STEP * SR_STEP;
}

// NG: operand gathers issued for the "other features" of an entry (3 = all of them: the kernel; 1 and 2 exist to PRICE a
// class-major entry order -- a region of entries with one / two other features would issue exactly that many -- and give
// wrong sums on the real store: MSX_SR_GATHERS)
template <int NG>
__global__ __launch_bounds__(MSX_BLOCK) void k_share_reduce(const unsigned long long *__restrict__ csr_tot,
                                                            const uint32_t *__restrict__ t_key,
                                                            const unsigned long long *__restrict__ t_val,
                                                            const double *__restrict__ a, int bits, int64_t W,
                                                            double *__restrict__ share,
                                                            double *__restrict__ part_val,
                                                            const int32_t *__restrict__ iter_state,
                                                            uint32_t a_bytes, uint32_t recip_at, int64_t wave_lo, int64_t wave_hi) {
	if (iter_state[0]) return;
	const int64_t E = (int64_t)csr_tot[1];
	const int lane = threadIdx.x & 63;
	// (consecutive workgroups -- which the dispatcher deals round-robin to the 8 XCDs -- take
	// consecutive chunks, so the XCDs walk the feature range together.  Keeping windows of it on one
	// XCD measured slower the longer the window: groups of 4 / 16 / 64 workgroups per XCD 58 / 60 / 67 us
	// against 58, a contiguous eighth each 70 against 65 -- lines one XCD has fetched are then no longer
	// served to the others out of the Infinity Cache while they are there)
	// (a launch takes the waves [wave_lo, wave_hi) of the W the entries are dealt to: all of them, or one slice of them --
	// msx_profile_prop_local_slice)
	const int64_t wave = wave_lo + (((int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x) >> 6);
	if (wave >= wave_hi) return;
	const int64_t per = sr_chunk(E, W);
	const int64_t c0 = wave * per;
	if (c0 >= E) {   // idle wave: neutral slots
		if (lane < 2) part_val[2 * wave + lane] = 0.0;
		return;
	}
	const int64_t c1 = (c0 + per < E) ? c0 + per : E;
	const uint32_t fmask = bits < 32 ? ((1u << bits) - 1u) : 0xffffffffu;
	// the open segment carried from step to step (its sum so far, whether it began inside this chunk)
	double carry = 0.0;
	bool carry_started = false, carry_open = false;
	uint32_t prev_last = 0;
	double slot_a = 0.0;          // partial of the chunk's first segment when it began in an earlier chunk
	const unsigned long long below = (1ull << lane) - 1ull;
	// Entries of one step: SR_EPL consecutive ones per lane, and the key following the step.  Every load below is
	// unconditional (addresses clamped, results masked afterwards): with loads under branches the compiler
	// cannot tell how many are in flight and waits for all of them -- vmcnt(0) -- before the first use, which
	// undoes the pipeline.
	auto load_step = [&](int64_t base, uint32_t *k, unsigned long long *lv, uint32_t &after) {
		const int64_t r0 = base + (int64_t)lane * SR_EPL;            // (the buffers are padded by a step)
		const uint4 *kp = reinterpret_cast<const uint4 *>(t_key + r0);
		const ulonglong2 *vp = reinterpret_cast<const ulonglong2 *>(t_val + r0);
		// (non-temporal loads for these two streams, to keep a[] in the L2s, measured slower: 74 vs 58 us)
#pragma unroll
		for (int q = 0; q < SR_EPL / 4; q++) {
			const uint4 ka = kp[q];
			k[4 * q] = ka.x; k[4 * q + 1] = ka.y; k[4 * q + 2] = ka.z; k[4 * q + 3] = ka.w;
		}
#pragma unroll
		for (int q = 0; q < SR_EPL / 2; q++) {
			const ulonglong2 v = vp[q];
			lv[2 * q] = v.x; lv[2 * q + 1] = v.y;
		}
		const int64_t nx = base + SR_STEP < E ? base + SR_STEP : E - 1;
		after = t_key[nx];
	};
	// entries past the chunk's end: a neutral last segment
	auto mask_step = [&](int64_t base, uint32_t *k, unsigned long long *lv, uint32_t &after) {
		const int64_t r0 = base + (int64_t)lane * SR_EPL;
#pragma unroll
		for (int i = 0; i < SR_EPL; i++)
			if (r0 + i >= c1) { k[i] = SR_SENT; lv[i] = 0ull; }
		after = (lane == 63 && base + SR_STEP < E) ? (after & fmask) : SR_SENT;
	};
	// The operands of each entry's term w/S.  A list of <= 4 features travels with its entries (the other
	// features in the value, weight above the feature id in the key): S is summed from a[] -- 8 MB that
	// the caches hold well, unlike one 8-byte recip[] per list out of tens of MB.  General lists: recip[u].
	// The gathers are buffer loads: a lane that has no such operand asks for an offset beyond the buffer, which
	// the hardware answers with zero without going to the vector cache at all -- no branch (with loads under
	// branches the compiler cannot count what is in flight), no exec masking, no lane's worth of cache work
	// for the two fifths of the operand slots that are empty.
	// (measured slower: fetching an index only when it differs from the previous entry's -- a lane's
	// entries mostly share their feature -- 67 us against 55; a[] of the 512 features from the step's
	// first one on staged in LDS, in-window gathers as ds_read_b64 -- 65 us against 53)
	// recip[] of the general lists lives behind a[] in the same allocation (recip_at bytes in): one descriptor, and the
	// gather of a general entry's w/S takes the place of the gather of its own feature -- a fifth instruction per
	// entry for 0.6 % of the entries cost 3 us per launch
	const auto rs_a = __builtin_amdgcn_make_buffer_rsrc((void *)a, 0, (int)a_bytes, 0x00020000);
	auto bload = [](decltype(rs_a) rs, uint32_t byte_off) {
		const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)byte_off, 0, 0);
		double d;
		__builtin_memcpy(&d, &v, 8);
		return d;
	};
	const uint32_t NONE = 0xffffffffu;                            // (beyond any buffer)
	auto gather_step = [&](const uint32_t *k, const unsigned long long *lv, double *af, double *a1, double *a2, double *a3) {
#pragma unroll
		for (int i = 0; i < SR_EPL; i++) {
			const bool live = k[i] != SR_SENT;
			const bool general = live && (lv[i] & SIG_HASHED) != 0;
			const bool exact = live && !general;
			const uint32_t o1 = (uint32_t)(lv[i] & SIG_PAD), o2 = (uint32_t)((lv[i] >> 21) & SIG_PAD),
			               o3 = (uint32_t)((lv[i] >> 42) & SIG_PAD);
			af[i] = bload(rs_a, exact ? (k[i] & fmask) * 8u : general ? recip_at + (uint32_t)lv[i] * 8u : NONE);
			a1[i] = bload(rs_a, (exact && o1 != SIG_PAD) ? o1 * 8u : NONE);
			a2[i] = NG >= 2 ? bload(rs_a, (exact && o2 != SIG_PAD) ? o2 * 8u : NONE) : 0.0;
			a3[i] = NG >= 3 ? bload(rs_a, (exact && o3 != SIG_PAD) ? o3 * 8u : NONE) : 0.0;
		}
	};
	// A three-stage pipeline over the steps of the chunk: while step i is summed, the gathers of step i+1 and
	// the entries of step i+2 are in flight -- the two round trips of a step (entries, then what they point
	// at) were what the kernel waited for, with the vector ALUs busy 30 % of the time.
	struct Ent { uint32_t k[SR_EPL]; unsigned long long lv[SR_EPL]; uint32_t after; };
	struct Ops { double af[SR_EPL], a1[SR_EPL], a2[SR_EPL], a3[SR_EPL]; };
	Ent e0, e1, e2;
	Ops g0, g1;
	const int64_t n_steps = (c1 - c0 + SR_STEP - 1) / SR_STEP;
	const int64_t last_base = c0 + (n_steps - 1) * SR_STEP;
	int64_t b1 = c0 + SR_STEP < last_base ? c0 + SR_STEP : last_base;     // the step e1 holds
	load_step(c0, e0.k, e0.lv, e0.after);
	load_step(b1, e1.k, e1.lv, e1.after);
	mask_step(c0, e0.k, e0.lv, e0.after);
	gather_step(e0.k, e0.lv, g0.af, g0.a1, g0.a2, g0.a3);
	uint32_t before = 0;
	if (lane == 0 && c0 > 0) before = t_key[c0 - 1] & fmask;
	for (int64_t step = 0; step < n_steps; step++) {
		const int64_t base = c0 + step * SR_STEP;
		// (beyond the chunk's last step the same step is fetched again and not used)
		const int64_t b2 = base + 2 * SR_STEP < last_base ? base + 2 * SR_STEP : last_base;
		load_step(b2, e2.k, e2.lv, e2.after);
		mask_step(b1, e1.k, e1.lv, e1.after);
		gather_step(e1.k, e1.lv, g1.af, g1.a1, g1.a2, g1.a3);
		uint32_t k[SR_EPL];
		const uint32_t after = e0.after;
		// ((o1 + o2) + o3) + own.  (The entries of one list add the same numbers in different orders: their S
		// can differ in the last bit, 1e-16 relative; ordering them costs a third of this kernel's
		// instructions and buys nothing at the 1e-6 the profile is held to.)
		double x[SR_EPL];
#pragma unroll
		for (int i = 0; i < SR_EPL; i++) {
			const bool live = e0.k[i] != SR_SENT;
			const bool general = (e0.lv[i] & SIG_HASHED) != 0;
			const double sum = ((g0.a1[i] + g0.a2[i]) + g0.a3[i]) + g0.af[i];     // absent ones came back as +0.0
			const double w = (double)(bits < 32 ? (e0.k[i] >> bits) : 0u);
			x[i] = !live ? 0.0 : general ? g0.af[i] : (sum > 0 ? w / sum : 0.0);      // (general: af is recip[u])
			k[i] = live ? (e0.k[i] & fmask) : e0.k[i];                            // from here on: the feature id
		}
		e0 = e1; e1 = e2; g0 = g1;
		b1 = b2;

		// ---- neighbours across lanes ----
		uint32_t pk = __shfl_up(k[SR_EPL - 1], 1, 64);
		if (lane == 0) pk = (base == c0) ? (c0 > 0 ? before : ~k[0]) : prev_last;
		uint32_t nk = __shfl_down(k[0], 1, 64);
		if (lane == 63) nk = after;

		// ---- the lane's run: head segment, complete interior segments, tail segment ----
		const bool B = (k[0] != pk);                         // a segment starts at the lane's first entry
		uint32_t cur = k[0];
		const uint32_t head_key = k[0];
		double run = x[0], head_sum = 0.0;
		bool in_head = true;
#pragma unroll
		for (int i = 1; i < SR_EPL; i++) {
			if (k[i] != cur) {
				if (in_head) { head_sum = run; in_head = false; }
				else if (cur != SR_SENT) share[cur] = run;   // began and ended inside this lane's run
				cur = k[i];
				run = x[i];
			} else {
				run += x[i];
			}
		}
		const bool pass = in_head;                           // one key over the whole run
		const bool G = B || !pass;                           // the tail segment starts in this lane
		const bool Eend = (nk != cur);                       // ... and ends with this lane's last entry

		// ---- segmented inclusive scan of the tail sums over the lanes ----
		double v = run;
		if (lane == 0 && !G) v += carry;
		uint32_t f = (G || lane == 0) ? 1u : 0u;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const double ov = __shfl_up(v, d, 64);
			const uint32_t of = __shfl_up(f, d, 64);
			if (lane >= d && !f) { v += ov; f = of; }
		}
		double cin = __shfl_up(v, 1, 64);                    // what flows into this lane's first segment
		if (lane == 0) cin = carry;
		if (B) cin = 0.0;
		const unsigned long long gb = __ballot(G);
		const bool started_in = B ? true : (((gb & below) != 0ull) ? true : carry_started);
		// head segment: ends inside this lane
		if (!pass) {
			const double tot = head_sum + cin;
			if (started_in) { if (head_key != SR_SENT) share[head_key] = tot; }
			else slot_a = tot;
		}
		// tail segment: v is its sum up to this lane's last entry
		const bool started_tail = G ? true : started_in;
		if (Eend) {
			if (started_tail) { if (cur != SR_SENT) share[cur] = v; }
			else slot_a = v;
		}
		// slot_a is set by at most one lane of the chunk: share it
		{
			const unsigned long long cut = __ballot((!pass && !started_in) || (Eend && !started_tail));
			if (cut) slot_a = __shfl(slot_a, __ffsll((long long)cut) - 1, 64);
		}
		// carry out of lane 63
		const double cv = __shfl(v, 63, 64);
		const int ce = __shfl((int)Eend, 63, 64);
		const int cs = __shfl((int)started_tail, 63, 64);
		prev_last = __shfl(k[SR_EPL - 1], 63, 64);
		carry_open = !ce;
		carry = ce ? 0.0 : cv;
		carry_started = ce ? false : (cs != 0);
	}
	// slot 2w: the part of the chunk's first segment when it began in an earlier chunk (key = the chunk's
	// first feature); slot 2w + 1: the segment left open at the chunk's end (key = its last feature)
	if (lane == 0) {
		part_val[2 * wave] = slot_a;
		part_val[2 * wave + 1] = carry_open ? carry : 0.0;
	}
}

// The keys of the partial slots depend on the sorted entries and on W only, so they are set once per
// build (k_part_index), and so is what k_prop_apply needs to fold the slots into the update
// (k_part_runs): the slots of one feature are neighbours (the keys ascend), a *run*; runs[] lists every
// run as (feature, first slot, number of slots), and a bitmap marks the features that own one.
#define PA_FPB 1024                    // features per streaming workgroup of k_prop_apply
__device__ __forceinline__ uint32_t part_key_at(int64_t j, int64_t E, int64_t per, const uint32_t *t_key, uint32_t fmask) {
	const int64_t c0 = (j >> 1) * per;
	if (c0 >= E) return SR_SENT;          // idle wave
	if (!(j & 1)) return t_key[c0] & fmask;
	const int64_t c1 = (c0 + per < E) ? c0 + per : E;
	return t_key[c1 - 1] & fmask;
}

__global__ __launch_bounds__(MSX_BLOCK) void k_part_index(const unsigned long long *__restrict__ csr_tot,
                                                          const uint32_t *__restrict__ t_key, int bits, int64_t W,
                                                          uint32_t *__restrict__ part_key, unsigned long long *d_tot) {
	const int64_t E = (int64_t)csr_tot[1];
	const int64_t per = sr_chunk(E, W);
	const uint32_t fmask = bits < 32 ? ((1u << bits) - 1u) : 0xffffffffu;
	const int64_t stride = (int64_t)gridDim.x * MSX_BLOCK;
	for (int64_t j = (int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x; j < 2 * W; j += stride)
		part_key[j] = part_key_at(j, E, per, t_key, fmask);
	if (blockIdx.x == 0 && threadIdx.x == 0) { d_tot[3] = 0; d_tot[4] = 0; }   // short / long runs, counted by k_part_runs
}

struct PartRun {
	uint32_t key, first, n, pad;
};

// runs of up to PA_SHORT slots go to runs[0 .. d_tot[3]), longer ones (a hot reference cut by hundreds of
// chunk boundaries) to the far end of the same array, runs[cap - 1 - i] for i < d_tot[4].
// ONE workgroup walks the slots in index order and places every run by ballot ranks: the lists come out in slot order, the
// same for every build of the same store.  (Until round 5 the runs were appended through atomic counters, a workgroup per
// 256 slots: which lane of k_prop_apply met which run then varied from build to build, and with it the order in which the
// diff^2 of those features entered DELTA^2 -- its last bits differed from run to run while every abundance repeated bit for
// bit; tests/test_gpu_determinism.py.  The slots number two per wave of k_share_reduce -- 6144 -- so one workgroup is enough.)
#define PA_SHORT 8                     // slots one lane adds by itself
#define PR_BLOCK 1024
#define PR_PER_MAX 16                  // consecutive slots per thread: 16 K slots = 8192 waves of k_share_reduce, more than the chip holds
__global__ __launch_bounds__(PR_BLOCK) void k_part_runs(const uint32_t *__restrict__ part_key, int64_t M,
                                                        PartRun *__restrict__ runs, uint32_t *__restrict__ owned,
                                                        unsigned long long *d_tot) {
	__shared__ int64_t s_min[PR_BLOCK / 64];
	__shared__ uint32_t s_cnt[2][PR_BLOCK / 64];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int per = (int)((M + PR_BLOCK - 1) / PR_BLOCK);          // <= PR_PER_MAX (checked by the host)
	const int64_t jb = (int64_t)threadIdx.x * per;
	// a run ends where the next one begins: the boundaries (slots whose key differs from the slot before; the idle waves'
	// sentinel keys at the far end are one more "run", not listed) -- no searching
	uint32_t key[PR_PER_MAX + 1];
#pragma unroll
	for (int i = 0; i <= PR_PER_MAX; i++) {
		const int64_t q = jb - 1 + i;
		key[i] = (i <= per && q >= 0 && q < M) ? part_key[q] : SR_SENT;
	}
	bool bnd[PR_PER_MAX];
	int64_t first = M;                                             // the first boundary among this thread's slots
#pragma unroll
	for (int i = PR_PER_MAX - 1; i >= 0; i--) {
		const int64_t j = jb + i;
		bnd[i] = i < per && j < M && (j == 0 || key[i] != key[i + 1]);
		if (bnd[i]) first = j;
	}
	int64_t x = first;                                             // suffix minimum over the wave (inclusive)
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const int64_t o = __shfl_down(x, d, 64);
		if (lane + d < 64 && o < x) x = o;
	}
	if (lane == 0) s_min[w] = x;
	__syncthreads();
	int64_t nextb = __shfl_down(x, 1, 64);                         // the first boundary behind this thread's slots
	if (lane == 63) nextb = M;
	for (int k = w + 1; k < PR_BLOCK / 64; k++) if (s_min[k] < nextb) nextb = s_min[k];
	uint32_t len[PR_PER_MAX];
	uint32_t ns = 0, nb = 0;
#pragma unroll
	for (int i = PR_PER_MAX - 1; i >= 0; i--) {
		len[i] = 0;
		if (bnd[i]) {
			const int64_t j = jb + i;
			if (key[i + 1] != SR_SENT) {
				len[i] = (uint32_t)(nextb - j);
				if (len[i] > PA_SHORT) nb++; else ns++;
				atomicOr(&owned[key[i + 1] >> 5], 1u << (key[i + 1] & 31u));      // (a bitmap: order-free)
			}
			nextb = j;
		}
	}
	// exclusive ranks of this thread's short and long runs among all of them, in slot order
	uint32_t xs = ns, xb = nb;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t os = __shfl_up(xs, d, 64), ob = __shfl_up(xb, d, 64);
		if (lane >= d) { xs += os; xb += ob; }
	}
	if (lane == 63) { s_cnt[0][w] = xs; s_cnt[1][w] = xb; }
	__syncthreads();
	uint32_t o0 = xs - ns, o1 = xb - nb;
	for (int k = 0; k < w; k++) { o0 += s_cnt[0][k]; o1 += s_cnt[1][k]; }
#pragma unroll
	for (int i = 0; i < PR_PER_MAX; i++)
		if (len[i]) {
			const PartRun r = {key[i + 1], (uint32_t)(jb + i), len[i], 0u};
			if (r.n > PA_SHORT) runs[M - 1 - (int64_t)(o1++)] = r; else runs[o0++] = r;
		}
	if (threadIdx.x == 0) {
		uint32_t t0 = 0, t1 = 0;
		for (int k = 0; k < PR_BLOCK / 64; k++) { t0 += s_cnt[0][k]; t1 += s_cnt[1][k]; }
		d_tot[3] = t0; d_tot[4] = t1;
	}
}

// With a collective between the halves of an iteration share[] must be complete before it leaves the device: the
// partial slots are folded into it run by run -- one lane per short run, one wave per long run, slots added in slot
// order -- with plain stores: every feature owns at most one run (its slots are neighbours), so no two lanes write
// the same word, the order of the additions is fixed and the result repeats bit for bit (the atomic adds this
// replaces did not).  The runs are the ones k_part_runs lists for k_prop_apply<true>.
__global__ __launch_bounds__(MSX_BLOCK) void k_partial_reduce(int64_t n_slots, const unsigned long long *__restrict__ d_tot,
                                                              const PartRun *__restrict__ runs,
                                                              const double *__restrict__ part_val,
                                                              double *__restrict__ share,
                                                              const int32_t *__restrict__ iter_state, uint32_t key_lo, uint32_t key_hi) {
	if (iter_state[0]) return;
	const int lane = threadIdx.x & 63;
	const int64_t r = (int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x;
	// (the runs of the features [key_lo, key_hi): all of them, or a slice's)
	if (r < (int64_t)d_tot[3] && runs[r].key >= key_lo && runs[r].key < key_hi) {
		const PartRun R = runs[r];
		double sum = 0;
		for (uint32_t q = 0; q < R.n; ++q) sum += part_val[R.first + q];
		// += : a segment that ENDS with its chunk's last entry is stored by k_share_reduce itself and still owns the
		// chunk's (zero) end slot.  Precondition: share[] of a feature nothing was stored to this iteration is 0 --
		// k_prop_begin and prop_update (which zeroes what it has consumed) see to that.
		share[R.key] += sum;
	}
	const int64_t n_long = (int64_t)d_tot[4];
	const int64_t n_waves = (int64_t)gridDim.x * (MSX_BLOCK / 64);
	for (int64_t w = r >> 6; w < n_long; w += n_waves) {
		const PartRun R = runs[n_slots - 1 - w];
		if (R.key < key_lo || R.key >= key_hi) continue;
		double part = 0;
		for (uint32_t q = (uint32_t)lane; q < R.n; q += 64u) part += part_val[R.first + q];
		double tot = 0;
		for (int l = 0; l < 64; l++) tot += __shfl(part, l, 64);     // every lane computes the same total
		if (lane == 0) share[R.key] += tot;
	}
}

// a = U + a * share, clamp, DELTA^2, convergence (msam_profile.c:368-389).  The first `nsb` workgroups
// stream the features, PA_FPB each.  FUSED (single GPU, no collective inside the iteration): share[]
// holds the segments k_share_reduce stored directly; a feature whose segment was cut by a chunk boundary
// owns a run of partial slots instead -- the streaming workgroups leave those features (bitmap `owned`)
// to the workgroups behind them, where one thread per run adds its slots in slot order (a fixed
// summation order: results repeat bit for bit) and updates the feature.  Not FUSED: share[] is complete
// (k_partial_reduce and the caller's all-reduce have run).  Every workgroup leaves its sum
// of diff^2 for k_prop_finish.
__device__ __forceinline__ double prop_update(int64_t i, double sh, const double *U, double *a, double *share) {
	const double old = a[i];
	double v = U[i] + old * sh;
	if (v < 1e-20) v = 0;
	a[i] = v;
	share[i] = 0.0;
	const double diff = v - old;
	return diff * diff;
}

template <bool FUSED>
__global__ __launch_bounds__(MSX_BLOCK) void k_prop_apply(int32_t nf, int nsb, int64_t n_slots, const double *__restrict__ U,
                                                          double *__restrict__ share, double *__restrict__ a,
                                                          const unsigned long long *__restrict__ d_tot,
                                                          const PartRun *__restrict__ runs,
                                                          const double *__restrict__ part_val,
                                                          const uint32_t *__restrict__ owned,
                                                          double *__restrict__ partial,
                                                          const int32_t *__restrict__ iter_state) {
	__shared__ double s_w[MSX_BLOCK / 64];
	if (iter_state[0]) return;
	double acc = 0;
	if ((int)blockIdx.x < nsb) {
		const int64_t i0 = (int64_t)blockIdx.x * PA_FPB;
#pragma unroll
		for (int q = 0; q < PA_FPB / MSX_BLOCK; q++) {
			const int64_t i = i0 + q * MSX_BLOCK + threadIdx.x;
			if (i < nf) {
				// (32 consecutive lanes read the same bitmap word)
				if (FUSED && ((owned[i >> 5] >> (i & 31)) & 1u)) continue;
				acc += prop_update(i, share[i], U, a, share);
			}
		}
	} else if (FUSED) {
		// the workgroups behind the streaming ones: one lane per short run, then one wave per long run --
		// lane-strided partial sums, added lane by lane: a fixed order too
		const int lane = threadIdx.x & 63;
		const int64_t r = (int64_t)((int)blockIdx.x - nsb) * MSX_BLOCK + threadIdx.x;
		if (r < (int64_t)d_tot[3]) {
			const PartRun R = runs[r];
			double sum = 0;
			for (uint32_t q = 0; q < R.n; ++q) sum += part_val[R.first + q];
			acc = prop_update(R.key, share[R.key] + sum, U, a, share);
		}
		const int64_t n_long = (int64_t)d_tot[4];
		const int64_t n_waves = (int64_t)((int)gridDim.x - nsb) * (MSX_BLOCK / 64);
		for (int64_t w = ((int64_t)((int)blockIdx.x - nsb) * MSX_BLOCK + threadIdx.x) >> 6; w < n_long; w += n_waves) {
			const PartRun R = runs[n_slots - 1 - w];
			double part = 0;
			for (uint32_t q = (uint32_t)lane; q < R.n; q += 64u) part += part_val[R.first + q];
			double tot = 0;
			for (int l = 0; l < 64; l++) tot += __shfl(part, l, 64);     // every lane computes the same total
			if (lane == 0) acc += prop_update(R.key, share[R.key] + tot, U, a, share);
		}
	}
	for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d, 64);
	if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0) partial[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// DELTA^2 = sum of the per-workgroup sums, in index order, over n_features; convergence (:380-383).
// (A launch of its own: letting the last workgroup of k_prop_apply do it needs a ticket that a thousand
// workgroups draw from one address -- 12 us of serialised atomics -- or a release fence per workgroup,
// which writes back the whole L2 each time -- 44 us; a dependent launch costs 2.)
// Workgroups 1.. of the same launch: recip[] of the general lists for the NEXT iteration -- it depends on a[]
// like this sum does, and a launch of its own cost 5 us per iteration.  (They read the convergence flag as the
// iterations before left it: once more than needed when this one converges, into an array nobody reads again.)
__global__ __launch_bounds__(MSX_BLOCK) void k_prop_finish(int nparts, const double *__restrict__ partial, int32_t nf,
                                                           double *__restrict__ delta, int32_t *iter_state, int k,
                                                           RecipArgs G) {
	__shared__ double s_w[MSX_BLOCK / 64];
	if (iter_state[0]) return;
	if (blockIdx.x > 0) {
		general_recip_body(G, (int64_t)(blockIdx.x - 1) * MSX_BLOCK + threadIdx.x, (int64_t)(gridDim.x - 1) * MSX_BLOCK);
		return;
	}
	double acc = 0;
	for (int i = threadIdx.x; i < nparts; i += MSX_BLOCK) acc += partial[i];
	for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d, 64);
	if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0) {
		const double dl = (s_w[0] + s_w[1] + s_w[2] + s_w[3]) / nf;   // :380
		delta[k] = dl;
		iter_state[1] = k;
		if (dl < 1e-10) iter_state[0] = 1;                         // :383
	}
}

// multi-mappers whose features all ended at zero (msam_profile.c:394-404)
__global__ __launch_bounds__(MSX_BLOCK) void k_prop_purged(const unsigned long long *__restrict__ csr_tot,
                                                           const uint32_t *__restrict__ m_off,
                                                           const int32_t *__restrict__ m_fid,
                                                           const uint32_t *__restrict__ hpos,
                                                           const double *__restrict__ a, uint32_t *out_count) {
	__shared__ uint32_t s_w[MSX_BLOCK / 64];
	const int64_t n_lists = (int64_t)csr_tot[0];
	uint32_t c = 0;
	const int64_t stride = (int64_t)gridDim.x * MSX_BLOCK;
	for (int64_t j = (int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x; j < n_lists; j += stride) {
		const uint32_t s = m_off[j], e = m_off[j + 1];
		double sum = 0;
		for (uint32_t k = s; k < e; ++k) sum += a[m_fid[k]];
		if (sum == 0) c += hpos[j + 1] - hpos[j];
	}
	for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
	if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t v = s_w[0] + s_w[1] + s_w[2] + s_w[3];
		if (v) atomicAdd(out_count, v);
	}
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
#ifdef MSX_DEBUG_SWITCHES
// MSX_SR_CLASSES=1: how many entries of the derived store have 0 / 1 / 2 / 3 other features in their value, how many
// belong to general lists (what a class-major order would have to work with)
__global__ __launch_bounds__(MSX_BLOCK) void k_sr_classes(const unsigned long long *__restrict__ csr_tot, const unsigned long long *__restrict__ t_val,
                                                          unsigned long long *__restrict__ out) {
	const int64_t E = (int64_t)csr_tot[1];
	unsigned long long c[5] = {0, 0, 0, 0, 0};
	for (int64_t i = (int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x; i < E; i += (int64_t)gridDim.x * MSX_BLOCK) {
		const unsigned long long v = t_val[i];
		if (v & SIG_HASHED) { c[4]++; continue; }
		const int n = ((v & SIG_PAD) != SIG_PAD) + (((v >> 21) & SIG_PAD) != SIG_PAD) + (((v >> 42) & SIG_PAD) != SIG_PAD);
		c[n]++;
	}
	for (int k = 0; k < 5; k++) if (c[k]) atomicAdd(&out[k], c[k]);
}
#endif

// waves of one k_share_reduce launch: what the chip holds at once
int64_t msx_share_waves(msx_ctx *ctx) {
#ifdef MSX_DEBUG_SWITCHES
	static const int wps = [] {
		const char *e = getenv("MSX_SR_WPS");          // experiments (libmsamtools_amd_dbg only): waves per SIMD the launch is sized for
		const int v = e ? atoi(e) : 0;
		return (v >= 1 && v <= 8) ? v : SR_WAVES_PER_SIMD;
	}();
#else
	const int wps = SR_WAVES_PER_SIMD;
#endif
	return (int64_t)ctx->num_cu * 4 * wps;
}
int64_t msx_apply_blocks(int32_t nf) { return nf > 0 ? ((int64_t)nf + PA_FPB - 1) / PA_FPB : 1; }

static int nf_grid(msx_ctx *ctx, int32_t nf) {
	int g = msx_grid(ctx, nf, MSX_BLOCK);
	return g > PROP_MAX_BLOCKS ? PROP_MAX_BLOCKS : g;
}

// stable LSD radix sort of (key, val) pairs on the low `bits` bits of the key; *n_ptr (device)
// items, at most n_ub.  Values (8 bytes) ping-pong between p->t_val64[0] and [1], keys between
// p->t_key[0] and [1].  Returns the buffer holding the result.
template <typename V>
static int radix_sort_pairs(msx_ctx *ctx, msx_profile *p, const uint32_t *kin, const V *vin, int vin_buf,
                            const unsigned long long *n_ptr, int64_t n_ub, int bits, int *out_buf) {
	const int passes = (bits + 7) / 8;
	const int64_t n_waves = (n_ub + RS_TILE - 1) / RS_TILE;   // tiles
	const unsigned nblk = (unsigned)n_waves;
	static_assert(sizeof(V) == 8, "the value buffers hold 8-byte values");
	msx_buf *vbuf = p->t_val64;
	int cur = vin_buf;
	for (int ps = 0; ps < passes; ps++) {
		const int dst = cur ^ 1;
		const int left = bits - 8 * ps;
		const uint32_t dmask = left >= 8 ? 255u : ((1u << left) - 1u);
		msx_time_begin(ctx, MSX_K_RS_HIST);
		msx_time_bytes(ctx, 0, 4, n_ub, n_ptr);           // keys in (+ 1 KB of counters per 4096-key tile)
		hipLaunchKernelGGL(k_rs_hist<false>, dim3(nblk), dim3(MSX_BLOCK), 0, ctx->stream, kin, n_ptr,
		                   (int64_t)0, ps * 8, dmask, (uint32_t *)p->rs_hist.p, n_waves);
		msx_time_end(ctx);
		uint32_t *const dtot = (uint32_t *)p->rs_off.p + 256 * n_waves + 16;
		msx_time_begin(ctx, MSX_K_SCAN);
		msx_time_bytes(ctx, 0, 8, 256 * n_waves, n_ptr, RS_TILE, 256);   // the table in and out
		hipLaunchKernelGGL(k_rs_rowscan, dim3(256), dim3(MSX_BLOCK), 0, ctx->stream, (const uint32_t *)p->rs_hist.p, n_ptr,
		                   n_waves, (uint32_t *)p->rs_off.p, dtot);
		msx_time_end(ctx);
		msx_time_begin(ctx, MSX_K_RS_SCATTER);
		msx_time_bytes(ctx, 0, 24, n_ub, n_ptr);          // key + 8-byte value in and out
		hipLaunchKernelGGL((k_rs_scatter<V, true, false, true>), dim3(nblk), dim3(MSX_BLOCK), 0, ctx->stream, kin,
		                   vin, (uint32_t *)p->t_key[dst].p, (V *)vbuf[dst].p, n_ptr, (int64_t)0, ps * 8,
		                   dmask, (const uint32_t *)p->rs_off.p, n_waves, (const uint32_t *)dtot);
		msx_time_end(ctx);
		kin = (const uint32_t *)p->t_key[dst].p;
		vin = (const V *)vbuf[dst].p;
		cur = dst;
	}
	*out_buf = cur;
	return MSX_OK;
}

// Stable LSD radix sort of 32-bit keys on bits [shift0, shift0 + bits), for callers outside the profile
// (msx_coverage.hip): n keys (host count), ping-pong between k0 and k1; *sel tells which one holds the
// result.  hist / off: workspace, grown by msx_sort_keys32_reserve (which also tells the number of tiles and the
// table's address: a caller that produces the keys itself, tile by tile of MSX_SORT_TILE keys, can leave the first
// pass's digit counts of its tiles in the table -- hist[digit * n_tiles + tile] -- and name them as `counted_tiles`).
int msx_sort_keys32_reserve(msx_ctx *ctx, int64_t n, msx_buf *hist, msx_buf *off, int64_t *n_tiles_out) {
	const int64_t n_tiles = (n + RS_TILE - 1) / RS_TILE;
	int rc;
	if ((rc = msx_reserve(ctx, hist, (size_t)(256 * n_tiles + 16) * 4))) return rc;
	if ((rc = msx_reserve(ctx, off, (size_t)(256 * n_tiles + 16 + 512) * 4))) return rc;
	if (n_tiles_out) *n_tiles_out = n_tiles;
	return MSX_OK;
}

int msx_sort_keys32(msx_ctx *ctx, uint32_t *k0, uint32_t *k1, int64_t n, int shift0, int bits, msx_buf *hist, msx_buf *off,
                    int *sel, int64_t counted_tiles) {
	const int passes = (bits + 7) / 8;
	int64_t n_tiles = 0;
	uint32_t *kk[2] = {k0, k1};
	int cur = 0, rc;
	if ((rc = msx_sort_keys32_reserve(ctx, n, hist, off, &n_tiles))) return rc;
	uint32_t *const dtot = (uint32_t *)off->p + 256 * n_tiles + 16;
	for (int ps = 0; ps < passes; ps++) {
		const int left = bits - 8 * ps;
		const uint32_t dmask = left >= 8 ? 255u : ((1u << left) - 1u);
		const int64_t t0 = ps == 0 ? counted_tiles : 0;
		if (n_tiles > t0)
			hipLaunchKernelGGL(k_rs_hist<false>, dim3((unsigned)(n_tiles - t0)), dim3(MSX_BLOCK), 0, ctx->stream, (const uint32_t *)kk[cur],
			                   (const unsigned long long *)nullptr, n, shift0 + ps * 8, dmask, (uint32_t *)hist->p, n_tiles, t0);
		hipLaunchKernelGGL(k_rs_rowscan, dim3(256), dim3(MSX_BLOCK), 0, ctx->stream, (const uint32_t *)hist->p,
		                   (const unsigned long long *)nullptr, n_tiles, (uint32_t *)off->p, dtot);
		hipLaunchKernelGGL((k_rs_scatter<uint32_t, false, false, true>), dim3((unsigned)n_tiles), dim3(MSX_BLOCK), 0, ctx->stream,
		                   (const uint32_t *)kk[cur], (const uint32_t *)nullptr, kk[cur ^ 1], (uint32_t *)nullptr,
		                   (const unsigned long long *)nullptr, n, shift0 + ps * 8, dmask, (const uint32_t *)off->p, n_tiles,
		                   (const uint32_t *)dtot);
		cur ^= 1;
	}
	*sel = cur;
	MSX_HIP(ctx, hipGetLastError());
	return MSX_OK;
}

// Keys with an 8-bit value beside them, into the order (value, key bits [shift, shift + 8)), the value dropped on the way:
// pass 1 by the value (its digit counts for the first `counted_tiles` tiles already in hist[digit * n_tiles + tile], as
// with msx_sort_keys32) writes the keys only, every value's run -- a bucket -- beginning at a whole tile; pass 2 orders
// each bucket by the key's digit.  k0 (n keys) -> k1 -> k0: both hold msx_sort_k32v8_bound(n) keys.  lay (device,
// 512 words): where bucket v begins in the result, and how many keys it has (empty slots, all ones, fill the rest of
// its last tile).  Bucket `skip_bucket` is left unordered (the caller's empty slots).
int64_t msx_sort_k32v8_bound(int64_t n) { return ((n + RS_TILE - 1) / RS_TILE + 256) * RS_TILE; }
// (the workspace, before the caller leaves its counts in it; *n_tiles_out: the stride of the first pass's table)
int msx_sort_k32v8_reserve(msx_ctx *ctx, int64_t n, msx_buf *hist, msx_buf *off, int64_t *n_tiles_out) {
	if (n_tiles_out) *n_tiles_out = (n + RS_TILE - 1) / RS_TILE;
	return msx_sort_keys32_reserve(ctx, msx_sort_k32v8_bound(n), hist, off, nullptr);
}
int msx_sort_k32v8(msx_ctx *ctx, uint32_t *k0, const uint8_t *v0, uint32_t *k1, int64_t n, int shift, msx_buf *hist, msx_buf *off,
                   int64_t counted_tiles, int skip_bucket, uint32_t *lay) {
	int rc;
	const int64_t n_ub = msx_sort_k32v8_bound(n), tiles_ub = n_ub / RS_TILE, n_tiles = (n + RS_TILE - 1) / RS_TILE;
	if ((rc = msx_sort_k32v8_reserve(ctx, n, hist, off, nullptr))) return rc;
	uint32_t *const dtot = (uint32_t *)off->p + 256 * tiles_ub + 16, *const btot = dtot + 256;
	// pass 1 (the table's stride: n_tiles; the caller's counts are laid out for it -- msx_sort_keys32_reserve(n) told it so)
	if (n_tiles > counted_tiles)
		hipLaunchKernelGGL(k_rs_hist8, dim3((unsigned)(n_tiles - counted_tiles)), dim3(MSX_BLOCK), 0, ctx->stream, v0, n, (uint32_t *)hist->p,
		                   n_tiles, counted_tiles);
	hipLaunchKernelGGL(k_rs_rowscan, dim3(256), dim3(MSX_BLOCK), 0, ctx->stream, (const uint32_t *)hist->p,
	                   (const unsigned long long *)nullptr, n_tiles, (uint32_t *)off->p, btot);
	hipLaunchKernelGGL((k_rs_scatter_any<true, false>), dim3((unsigned)n_tiles), dim3(MSX_BLOCK), 0, ctx->stream, (const uint32_t *)k0, v0, k1, n, 0,
	                   (const uint32_t *)off->p, n_tiles, (const uint32_t *)btot, (const uint32_t *)nullptr, -1);
	hipLaunchKernelGGL(k_rs_bucket_pad, dim3(256 + 64), dim3(MSX_BLOCK), 0, ctx->stream, (const uint32_t *)btot, k1, n_ub, lay);
	// pass 2
	hipLaunchKernelGGL(k_rs_hist<false>, dim3((unsigned)tiles_ub), dim3(MSX_BLOCK), 0, ctx->stream, (const uint32_t *)k1,
	                   (const unsigned long long *)nullptr, n_ub, shift, 255u, (uint32_t *)hist->p, tiles_ub, (int64_t)0);
	hipLaunchKernelGGL(k_rs_rowscan, dim3(256), dim3(MSX_BLOCK), 0, ctx->stream, (const uint32_t *)hist->p,
	                   (const unsigned long long *)nullptr, tiles_ub, (uint32_t *)off->p, dtot);
	hipLaunchKernelGGL((k_rs_scatter_any<false, true>), dim3((unsigned)tiles_ub), dim3(MSX_BLOCK), 0, ctx->stream, (const uint32_t *)k1,
	                   (const uint8_t *)nullptr, k0, n_ub, shift, (const uint32_t *)off->p, tiles_ub, (const uint32_t *)dtot,
	                   (const uint32_t *)btot, skip_bucket);
	MSX_HIP(ctx, hipGetLastError());
	return MSX_OK;
}

int msx_prop_build(msx_ctx *ctx, msx_profile *p) {
	if (p->transposed_valid) return MSX_OK;
	const int64_t eub = p->entries_ub > 0 ? p->entries_ub : 1;
	const int64_t lub = p->lists_ub > 0 ? p->lists_ub : 1;
	int rc;
	for (int i = 0; i < 2; i++) {
		if ((rc = msx_reserve(ctx, &p->t_key[i], (size_t)(eub + 64 + SR_STEP) * 4))) return rc;
	}
	// a[] and recip[] in one allocation (k_share_reduce reaches both through one buffer descriptor); a[]'s contents are
	// dead here: k_prop_begin writes them.  32-bit byte offsets: both together must stay below 4 GB.
	{
		const size_t nf_pad = ((size_t)(p->n_features > 0 ? p->n_features : 1) + 7) / 8 * 8;
		const size_t bytes = (nf_pad + (size_t)lub + 8) * 8;
		if (bytes >= 0xfffffff0ull) return msx_fail(ctx, MSX_ERR_ARG, "proportional sharing: %lld features and %lld lists exceed the 4 GB the sharing kernel addresses", (long long)p->n_features, (long long)lub);
		if ((rc = msx_reserve(ctx, &p->recip, bytes))) return rc;
		if (!p->a_in_recip && p->a) (void)hipFree(p->a);
		p->a = (double *)p->recip.p;
		p->a_in_recip = true;
		p->recip_ptr = p->a + nf_pad;
		p->recip_at = (uint32_t)(nf_pad * 8);
	}
	{
		const int64_t W = msx_share_waves(ctx);
		if ((rc = msx_reserve(ctx, &p->part_key, (size_t)(2 * W + 8) * 4))) return rc;
		if ((rc = msx_reserve(ctx, &p->part_val, (size_t)(2 * W + 8) * 8))) return rc;
		if ((rc = msx_reserve(ctx, &p->runs, (size_t)(2 * W + 8) * sizeof(PartRun)))) return rc;
		if ((rc = msx_reserve(ctx, &p->owned, (size_t)(p->n_features / 32 + 8) * 4))) return rc;
	}
	const int64_t n_waves = ((eub > lub ? eub : lub) + RS_TILE - 1) / RS_TILE;
	if ((rc = msx_reserve(ctx, &p->rs_hist, (size_t)(256 * n_waves + 16) * 4))) return rc;
	if ((rc = msx_reserve(ctx, &p->rs_off, (size_t)(256 * n_waves + 16 + 256) * 4))) return rc;   // (+ the rows' totals)
	const unsigned long long *tot = p->csr_tot;
	int bits = 0;
	while (bits < 32 && ((int64_t)1 << bits) < (int64_t)p->n_features) bits++;

	// (a) derived store: lists ordered by (smallest feature, set hash), identical sets merged
	int ebuf = 1;
	{
		int hash_bits = 32 - bits;
		if (hash_bits > 12) hash_bits = 12;
		if (hash_bits < 0) hash_bits = 0;
		int key_bits_total = bits + hash_bits;
		int coarse = 0;
#ifdef MSX_DEBUG_SWITCHES
		// (libmsamtools_amd_dbg only; both measured and dropped in round 3)
		if (const char *e = getenv("MSX_LIST_KEY_HASH")) {        // experiment: the sort key is a hash of the set alone, this many bits
			const int v = atoi(e);
			if (v >= 8 && v <= 32) { hash_bits = -v; key_bits_total = v; }
		}
		if (const char *e = getenv("MSX_LIST_KEY_COARSE")) { const int v = atoi(e); if (v >= 0 && v < key_bits_total - 4) coarse = v; }
#endif
		if ((rc = msx_reserve(ctx, &p->m_off_alt, p->m_off.cap + 64))) return rc;
		if ((rc = msx_reserve(ctx, &p->m_fid_alt, p->m_fid.cap + 64))) return rc;
		if ((rc = msx_reserve(ctx, &p->head, (size_t)(lub + 8) * 4))) return rc;
		if ((rc = msx_reserve(ctx, &p->hpos, (size_t)(lub + 8) * 4))) return rc;
		for (int i = 0; i < 2; i++)
			if ((rc = msx_reserve(ctx, &p->t_val64[i], (size_t)(eub + 64 + SR_STEP) * 8))) return rc;
		MSX_TIMED(ctx, MSX_K_LIST_ORDER,
		          hipLaunchKernelGGL(k_list_key, dim3(msx_grid(ctx, lub, MSX_BLOCK)), dim3(MSX_BLOCK), 0, ctx->stream, tot,
		                             (const uint32_t *)p->m_off.p, (const int32_t *)p->m_fid.p, hash_bits, coarse, bits,
		                             (uint32_t *)p->t_key[0].p, (unsigned long long *)p->t_val64[0].p));
		int sb = 0;
		if ((rc = radix_sort_pairs(ctx, p, (const uint32_t *)p->t_key[0].p,
		                           (const unsigned long long *)p->t_val64[0].p, 0, tot + 0, lub, key_bits_total, &sb)))
			return rc;
		const uint32_t *skey = (const uint32_t *)p->t_key[sb].p;
		const unsigned long long *ssig = (const unsigned long long *)p->t_val64[sb].p;
		MSX_TIMED(ctx, MSX_K_LIST_ORDER,
		          hipLaunchKernelGGL(k_dup_mark, dim3(msx_grid(ctx, lub, MSX_BLOCK)), dim3(MSX_BLOCK), 0, ctx->stream, tot,
		                             lub, skey, ssig, (const uint32_t *)p->m_off.p, (const int32_t *)p->m_fid.p,
		                             (uint32_t *)p->head.p));
		const unsigned long long *chunk_base = nullptr;
		const int64_t n_chunks = (lub + MSX_PINFO_CHUNK - 1) / MSX_PINFO_CHUNK;
		if ((rc = msx_scan_pinfo_chunks(ctx, (const uint32_t *)p->head.p, lub, &chunk_base, tot))) return rc;
		// the feature-major entries are written as the derived store is built, into the buffers the
		// list sort is not holding its result in: key = feature, value = the list's signature or number
		ebuf = sb ^ 1;
		if ((rc = msx_reserve(ctx, &p->gl_idx, (size_t)(lub + 8) * 4))) return rc;
		MSX_TIMED(ctx, MSX_K_LIST_ORDER,
		          hipLaunchKernelGGL(k_uniq_gather, dim3((unsigned)n_chunks), dim3(MSX_BLOCK), 0, ctx->stream,
		                             tot, (const uint32_t *)p->head.p, chunk_base, n_chunks, ssig,
		                             (const uint32_t *)p->m_off.p, (const int32_t *)p->m_fid.p,
		                             (uint32_t *)p->m_off_alt.p, (int32_t *)p->m_fid_alt.p,
		                             (uint32_t *)p->t_key[ebuf].p, (unsigned long long *)p->t_val64[ebuf].p,
		                             (uint32_t *)p->hpos.p, lub, p->d_tot));
		MSX_TIMED(ctx, MSX_K_LIST_ORDER,
		          hipLaunchKernelGGL(k_entry_weight, dim3(msx_grid(ctx, lub, MSX_BLOCK)), dim3(MSX_BLOCK), 0, ctx->stream,
		                             p->d_tot, (const uint32_t *)p->m_off_alt.p,
		                             (const uint32_t *)p->hpos.p, bits, (uint32_t *)p->t_key[ebuf].p,
		                             (unsigned long long *)p->t_val64[ebuf].p, (uint32_t *)p->gl_idx.p));
	}
	tot = p->d_tot;      // everything below works on the derived store
	p->key_bits = bits;

	// (b) feature-major view: (feature, list) pairs sorted by feature
	int cur = ebuf;     // (a single feature: the list-major order is already feature-major)
	if (bits > 0 && (rc = radix_sort_pairs(ctx, p, (const uint32_t *)p->t_key[ebuf].p,
	                                       (const unsigned long long *)p->t_val64[ebuf].p, ebuf, tot + 1, eub, bits,
	                                       &cur)))
		return rc;
	p->sorted_buf = cur;
	// (c) keys of the partial slots, their runs, and the features that own one
	{
		const int64_t W = msx_share_waves(ctx);
		MSX_HIP(ctx, hipMemsetAsync(p->owned.p, 0, (size_t)(p->n_features / 32 + 1) * 4, ctx->stream));
		if (2 * W > (int64_t)PR_BLOCK * PR_PER_MAX) {              // (before the timing bracket opens and with the side lanes joined)
			msx_join(ctx);
			return msx_fail(ctx, MSX_ERR_ARG, "k_share_reduce with %lld waves: k_part_runs lists at most %d partial slots", (long long)W, PR_BLOCK * PR_PER_MAX);
		}
		msx_time_begin(ctx, MSX_K_LIST_ORDER);
		hipLaunchKernelGGL(k_part_index, dim3(msx_grid(ctx, 2 * W, MSX_BLOCK)), dim3(MSX_BLOCK), 0, ctx->stream,
		                   (const unsigned long long *)p->d_tot, (const uint32_t *)p->t_key[cur].p, bits, W,
		                   (uint32_t *)p->part_key.p, p->d_tot);
		hipLaunchKernelGGL(k_part_runs, dim3(1), dim3(PR_BLOCK), 0, ctx->stream,
		                   (const uint32_t *)p->part_key.p, 2 * W, (PartRun *)p->runs.p, (uint32_t *)p->owned.p, p->d_tot);
		msx_time_end(ctx);
#ifdef MSX_DEBUG_SWITCHES
		if (getenv("MSX_SR_CLASSES")) {
			unsigned long long h[5] = {0, 0, 0, 0, 0}, *d = (unsigned long long *)p->part_val.p;      // (not in use yet)
			MSX_HIP(ctx, hipMemsetAsync(d, 0, 40, ctx->stream));
			hipLaunchKernelGGL(k_sr_classes, dim3(1024), dim3(MSX_BLOCK), 0, ctx->stream, (const unsigned long long *)p->d_tot,
			                   (const unsigned long long *)p->t_val64[cur].p, d);
			MSX_HIP(ctx, hipMemcpyAsync(h, d, 40, hipMemcpyDeviceToHost, ctx->stream));
			MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
			fprintf(stderr, "# derived store: entries with 0 / 1 / 2 / 3 other features: %llu / %llu / %llu / %llu, of general lists: %llu\n",
			        h[0], h[1], h[2], h[3], h[4]);
		}
#endif
	}
	p->transposed_valid = true;
	p->slice_valid = false;
	MSX_HIP(ctx, hipGetLastError());
	return MSX_OK;
}

// One iteration's sums.  complete = false (single GPU, msx_profile_finalize_enqueue): share[] gets the
// segments that lie inside one chunk, the partial slots stay in part_val -- k_prop_apply<true> puts the
// two together.  complete = true (a collective follows): the slots are added into share[] here.
static int recip_grid(msx_ctx *ctx, const msx_profile *p) {
	int gg = msx_grid(ctx, p->lists_ub > 0 ? p->lists_ub : 1, MSX_BLOCK);
	return gg > 128 ? 128 : gg;                              // the general lists are few: a small grid strides over them
}

static RecipArgs recip_args(const msx_profile *p) {
	return RecipArgs{(const unsigned long long *)p->d_tot, (const uint32_t *)p->m_off_alt.p, (const int32_t *)p->m_fid_alt.p,
	                 (const uint32_t *)p->hpos.p, (const uint32_t *)p->gl_idx.p, (const double *)p->a, p->recip_ptr};
}

// the waves [wave_lo, wave_hi) of k_share_reduce and, if `complete`, the runs of the features [key_lo, key_hi) folded into share[]
static int prop_iteration_range(msx_ctx *ctx, msx_profile *p, bool complete, int64_t wave_lo, int64_t wave_hi, uint32_t key_lo, uint32_t key_hi) {
	if ((size_t)p->n_features * 8 > 0xfffffff0u)                 // (the gathers of k_share_reduce address a[] with 32-bit byte offsets)
		return msx_fail(ctx, MSX_ERR_ARG, "proportional sharing: too many features (%d)", p->n_features);
	const int64_t W = msx_share_waves(ctx);
	if (!p->recip_valid) {
		MSX_TIMED(ctx, MSX_K_GENERAL_RECIP,
		          hipLaunchKernelGGL(k_general_recip, dim3(recip_grid(ctx, p)), dim3(MSX_BLOCK), 0, ctx->stream,
		                             recip_args(p), (const int32_t *)p->iter_state));
		p->recip_valid = true;
	}
#ifdef MSX_DEBUG_SWITCHES
	// MSX_SR_GATHERS=1|2 (libmsamtools_amd_dbg only): the kernel instantiated for fewer operand gathers -- WRONG SUMS, the right
	// instruction stream: round 4's pricing of a class-major order.  Not in the product library.
	static const int ng = [] { const char *e = getenv("MSX_SR_GATHERS"); const int v = e ? atoi(e) : 3; return (v >= 1 && v <= 3) ? v : 3; }();
#else
	constexpr int ng = 3;
#endif
#define SR_LAUNCH(NG_)                                                                                                          \
	hipLaunchKernelGGL(k_share_reduce<NG_>, dim3((unsigned)((wave_hi - wave_lo + 3) / 4)), dim3(MSX_BLOCK), 0, ctx->stream,      \
	                   (const unsigned long long *)p->d_tot, (const uint32_t *)p->t_key[p->sorted_buf].p,                        \
	                   (const unsigned long long *)p->t_val64[p->sorted_buf].p, (const double *)p->a, p->key_bits, W, p->share,  \
	                   (double *)p->part_val.p, (const int32_t *)p->iter_state,                                                  \
	                   (uint32_t)(p->recip.cap < 0xfffffff0u ? p->recip.cap : 0xfffffff0u), p->recip_at, wave_lo, wave_hi)
	if (wave_hi <= wave_lo) return MSX_OK;
	msx_time_begin(ctx, MSX_K_SHARE_REDUCE);
#ifdef MSX_DEBUG_SWITCHES
	if (ng == 3) SR_LAUNCH(3); else if (ng == 2) SR_LAUNCH(2); else SR_LAUNCH(1);
#else
	(void)ng;
	SR_LAUNCH(3);
#endif
	msx_time_end(ctx);
#undef SR_LAUNCH
	if (complete) {
		const int64_t M = 2 * W;
		MSX_TIMED(ctx, MSX_K_PARTIAL_REDUCE,
		          hipLaunchKernelGGL(k_partial_reduce, dim3((unsigned)((M + MSX_BLOCK - 1) / MSX_BLOCK)), dim3(MSX_BLOCK), 0,
		                             ctx->stream, M, (const unsigned long long *)p->d_tot, (const PartRun *)p->runs.p,
		                             (const double *)p->part_val.p, p->share, (const int32_t *)p->iter_state, key_lo, key_hi));
	}
	return MSX_OK;
}
int msx_prop_iteration(msx_ctx *ctx, msx_profile *p, bool complete) {
	return prop_iteration_range(ctx, p, complete, 0, msx_share_waves(ctx), 0u, 0xffffffffu);
}

// SLICES of an iteration's local half (MSX_DIST_SLICES, msx_profile_prop_local_slice).  The feature range is cut into n equal
// parts [feat[i], feat[i + 1]) -- the SAME cuts on every rank, whatever its shard holds: they are what the ranks all-reduce.
// The entries are sorted by feature and dealt to the waves of k_share_reduce in equal chunks, so the entries of the features
// below feat[i + 1] end in front of the first wave whose first entry is at or beyond it: wave[i + 1].  Slice i launches the waves
// [wave[i], wave[i + 1]) and folds the runs of the features [key[i], key[i + 1]) -- key[i] = the feature of wave[i]'s first entry,
// >= feat[i] -- and with that every feature below feat[i + 1] is final: its all-reduce can travel while slice i + 1 is computed.
// The cuts are found once per store (the waves' first keys in one strided copy, one wait).
static int prop_slice_cuts(msx_ctx *ctx, msx_profile *p, int n) {
	if (p->slice_n == n && p->slice_valid) return MSX_OK;
	const int64_t W = msx_share_waves(ctx);
	unsigned long long tot[2] = {0, 0};
	MSX_HIP(ctx, hipMemcpyAsync(tot, p->d_tot, 16, hipMemcpyDeviceToHost, ctx->stream));
	MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	const int64_t E = (int64_t)tot[1];
	const int64_t per0 = (E + W - 1) / W, per = (per0 + SR_STEP - 1) / SR_STEP * SR_STEP;      // (sr_chunk)
	const uint32_t fmask = p->key_bits < 32 ? ((1u << p->key_bits) - 1u) : 0xffffffffu;
	const int64_t live = per > 0 ? (E + per - 1) / per : 0;       // waves that hold entries
	std::vector<uint32_t> first((size_t)(live > 0 ? live : 1), 0u);
	if (live > 0) {
		MSX_HIP(ctx, hipMemcpy2DAsync(first.data(), 4, (const uint32_t *)p->t_key[p->sorted_buf].p, (size_t)per * 4, 4, (size_t)live,
		                              hipMemcpyDeviceToHost, ctx->stream));
		MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	}
	const uint32_t nf = (uint32_t)p->n_features;
	p->slice_wave[0] = 0; p->slice_key[0] = 0; p->slice_feat[0] = 0;
	for (int i = 1; i < n; i++) {
		const uint32_t F = (uint32_t)((uint64_t)nf * (uint64_t)i / (uint64_t)n);
		int64_t lo = 0, hi = live;                                  // the first wave whose first entry's feature is >= F
		while (lo < hi) {
			const int64_t mid = (lo + hi) >> 1;
			if ((first[(size_t)mid] & fmask) >= F) hi = mid; else lo = mid + 1;
		}
		p->slice_feat[i] = F;
		p->slice_wave[i] = lo < live ? lo : W;                      // (beyond the live waves: the idle ones go with the last slice that has any)
		p->slice_key[i] = lo < live ? (first[(size_t)lo] & fmask) : 0xffffffffu;
	}
	p->slice_wave[n] = W; p->slice_key[n] = 0xffffffffu; p->slice_feat[n] = nf;
	// (idle waves write their neutral slots: they must be launched by exactly one slice -- the last)
	for (int i = 1; i < n; i++) if (p->slice_wave[i] > p->slice_wave[i + 1]) p->slice_wave[i] = p->slice_wave[i + 1];
	p->slice_n = n;
	p->slice_valid = true;
	return MSX_OK;
}

int msx_prop_apply_launch(msx_ctx *ctx, msx_profile *p, int k, bool fused) {
	const int32_t nf = p->n_features;
	const int nsb = (int)msx_apply_blocks(nf);
	const int nrb = (int)((2 * msx_share_waves(ctx) + MSX_BLOCK - 1) / MSX_BLOCK);   // one thread per run at most
	msx_time_begin(ctx, MSX_K_PROP_APPLY);
	if (fused)
		hipLaunchKernelGGL(k_prop_apply<true>, dim3((unsigned)(nsb + nrb)), dim3(MSX_BLOCK), 0, ctx->stream, nf, nsb,
		                   2 * msx_share_waves(ctx), (const double *)p->U, p->share, p->a, (const unsigned long long *)p->d_tot,
		                   (const PartRun *)p->runs.p, (const double *)p->part_val.p, (const uint32_t *)p->owned.p,
		                   p->partial, (const int32_t *)p->iter_state);
	else
		hipLaunchKernelGGL(k_prop_apply<false>, dim3((unsigned)nsb), dim3(MSX_BLOCK), 0, ctx->stream, nf, nsb,
		                   2 * msx_share_waves(ctx), (const double *)p->U, p->share, p->a, (const unsigned long long *)p->d_tot,
		                   (const PartRun *)p->runs.p, (const double *)p->part_val.p, (const uint32_t *)p->owned.p,
		                   p->partial, (const int32_t *)p->iter_state);
	// (the sum of diff^2 in workgroup 0, recip[] of the general lists for the next iteration in the others)
	hipLaunchKernelGGL(k_prop_finish, dim3(1 + (k < 19 ? recip_grid(ctx, p) : 0)), dim3(MSX_BLOCK), 0, ctx->stream,
	                   fused ? nsb + nrb : nsb, (const double *)p->partial, nf, p->delta, p->iter_state, k, recip_args(p));
	p->recip_valid = true;
	msx_time_end(ctx);
	return MSX_OK;
}

int msx_prop_purged_launch(msx_ctx *ctx, msx_profile *p, uint32_t *out_dev) {
	const int64_t lub = p->lists_ub > 0 ? p->lists_ub : 1;
	MSX_TIMED(ctx, MSX_K_PROP_APPLY,
	          hipLaunchKernelGGL(k_prop_purged, dim3(msx_grid(ctx, lub, MSX_BLOCK)), dim3(MSX_BLOCK), 0, ctx->stream,
	                             (const unsigned long long *)p->d_tot, (const uint32_t *)p->m_off_alt.p,
	                             (const int32_t *)p->m_fid_alt.p, (const uint32_t *)p->hpos.p, (const double *)p->a,
	                             out_dev));
	return MSX_OK;
}

extern "C" int msx_profile_prop_begin(msx_ctx *ctx, msx_profile *p) {
	if (!ctx || !p) return MSX_ERR_ARG;
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	// The derived store needs the multi-mapper lists only: it is built while the side lanes
	// msx_filter_profile_enqueue left running (the unique-insert counts, filter's output order) finish;
	// the abundances wait for the counts.
	if (p->share_type == MSX_MULTI_SHARE_PROPORTIONAL) {
		int rc = msx_prop_build(ctx, p);
		if (rc) { msx_join(ctx); return rc; }
	}
	msx_join(ctx);
	{ int frc = msx_profile_fold_equal(ctx, p); if (frc) return frc; }        // (--multi equal: msx_count.h)
	const int32_t nf = p->n_features;
	msx_time_begin(ctx, MSX_K_PROP_APPLY);
	hipLaunchKernelGGL(k_prop_begin, dim3(nf_grid(ctx, nf)), dim3(MSX_BLOCK), 0, ctx->stream, nf,
	                   (const uint32_t *)p->ui, (const double *)p->d, p->U, p->a, p->share, p->delta, p->iter_state);
	msx_time_end(ctx);
	MSX_HIP(ctx, hipMemsetAsync(p->purged_local, 0, 4, ctx->stream));
	MSX_HIP(ctx, hipMemsetAsync(p->counters + 3, 0, 4, ctx->stream));   // purged is recomputed per finalize
	p->iter_k = 0;
	p->recip_valid = false;
	p->begun = true;
	MSX_HIP(ctx, hipGetLastError());
	return MSX_OK;
}

extern "C" int msx_profile_prop_local(msx_ctx *ctx, msx_profile *p, double **inc) {
	if (!ctx || !p) return MSX_ERR_ARG;
	if (!p->begun) return msx_fail(ctx, MSX_ERR_ARG, "msx_profile_prop_local before msx_profile_prop_begin");
	msx_join(ctx);
	if (p->share_type != MSX_MULTI_SHARE_PROPORTIONAL)
		return msx_fail(ctx, MSX_ERR_ARG, "proportional sharing was not selected for this profile");
	{
		const int rc = msx_prop_iteration(ctx, p, true);         // share[] complete: the caller all-reduces it
		if (rc) return rc;
	}
	if (inc) *inc = p->share;
	MSX_HIP(ctx, hipGetLastError());
	return MSX_OK;
}

extern "C" int msx_profile_prop_local_slice(msx_ctx *ctx, msx_profile *p, int slice, int n_slices, double **inc, int32_t *first, int32_t *count) {
	if (!ctx || !p) return MSX_ERR_ARG;
	if (!p->begun) return msx_fail(ctx, MSX_ERR_ARG, "msx_profile_prop_local_slice before msx_profile_prop_begin");
	if (n_slices < 1 || n_slices > MSX_MAX_SLICES || slice < 0 || slice >= n_slices) return msx_fail(ctx, MSX_ERR_ARG, "msx_profile_prop_local_slice: slice %d of %d", slice, n_slices);
	msx_join(ctx);
	if (p->share_type != MSX_MULTI_SHARE_PROPORTIONAL)
		return msx_fail(ctx, MSX_ERR_ARG, "proportional sharing was not selected for this profile");
	int rc = prop_slice_cuts(ctx, p, n_slices);
	if (rc) return rc;
	if ((rc = prop_iteration_range(ctx, p, true, p->slice_wave[slice], p->slice_wave[slice + 1], p->slice_key[slice], p->slice_key[slice + 1])))
		return rc;
	if (inc) *inc = p->share;
	if (first) *first = (int32_t)p->slice_feat[slice];
	if (count) *count = (int32_t)(p->slice_feat[slice + 1] - p->slice_feat[slice]);
	MSX_HIP(ctx, hipGetLastError());
	return MSX_OK;
}

extern "C" int msx_profile_prop_apply(msx_ctx *ctx, msx_profile *p, double *delta) {
	if (!ctx || !p) return MSX_ERR_ARG;
	if (!p->begun) return msx_fail(ctx, MSX_ERR_ARG, "msx_profile_prop_apply before msx_profile_prop_begin");
	msx_join(ctx);
	if (p->iter_k >= 19) return msx_fail(ctx, MSX_ERR_ARG, "proportional sharing runs at most 19 iterations");
	p->iter_k++;
	msx_prop_apply_launch(ctx, p, p->iter_k, false);
	MSX_HIP(ctx, hipGetLastError());
	double dl = 0;
	MSX_HIP(ctx, hipMemcpyAsync(&dl, p->delta + p->iter_k, 8, hipMemcpyDeviceToHost, ctx->stream));
	MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (delta) *delta = dl;
	return MSX_OK;
}

extern "C" int msx_profile_share_dev(msx_ctx *ctx, msx_profile *p, double **share) {
	if (!ctx || !p || !share) return MSX_ERR_ARG;
	msx_join(ctx);
	*share = p->share;
	return MSX_OK;
}

extern "C" int msx_profile_prop_apply_enqueue(msx_ctx *ctx, msx_profile *p) {
	if (!ctx || !p) return MSX_ERR_ARG;
	if (!p->begun) return msx_fail(ctx, MSX_ERR_ARG, "msx_profile_prop_apply_enqueue before msx_profile_prop_begin");
	msx_join(ctx);
	if (p->iter_k >= 19) return msx_fail(ctx, MSX_ERR_ARG, "proportional sharing runs at most 19 iterations");
	p->iter_k++;
	msx_prop_apply_launch(ctx, p, p->iter_k, false);
	MSX_HIP(ctx, hipGetLastError());
	return MSX_OK;
}

extern "C" int msx_profile_prop_purged_enqueue(msx_ctx *ctx, msx_profile *p, uint32_t **purged_dev) {
	if (!ctx || !p) return MSX_ERR_ARG;
	if (!p->begun || (p->share_type == MSX_MULTI_SHARE_PROPORTIONAL && !p->transposed_valid))
		return msx_fail(ctx, MSX_ERR_ARG, "msx_profile_prop_purged_enqueue before msx_profile_prop_begin");
	msx_join(ctx);
	MSX_HIP(ctx, hipMemsetAsync(p->purged_local, 0, 4, ctx->stream));
	msx_prop_purged_launch(ctx, p, p->purged_local);
	if (purged_dev) *purged_dev = p->purged_local;
	MSX_HIP(ctx, hipGetLastError());
	return MSX_OK;
}

extern "C" int msx_profile_prop_purged(msx_ctx *ctx, msx_profile *p, uint32_t *purged_local) {
	if (!ctx || !p) return MSX_ERR_ARG;
	if (!p->begun || (p->share_type == MSX_MULTI_SHARE_PROPORTIONAL && !p->transposed_valid))
		return msx_fail(ctx, MSX_ERR_ARG, "msx_profile_prop_purged before msx_profile_prop_begin");
	msx_join(ctx);
	MSX_HIP(ctx, hipMemsetAsync(p->purged_local, 0, 4, ctx->stream));
	msx_prop_purged_launch(ctx, p, p->purged_local);
	MSX_HIP(ctx, hipGetLastError());
	uint32_t v = 0;
	MSX_HIP(ctx, hipMemcpyAsync(&v, p->purged_local, 4, hipMemcpyDeviceToHost, ctx->stream));
	MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (purged_local) *purged_local = v;
	return MSX_OK;
}

extern "C" int msx_profile_finalize_enqueue(msx_ctx *ctx, msx_profile *p) {
	if (!ctx || !p) return MSX_ERR_ARG;
	int rc = msx_profile_prop_begin(ctx, p);
	if (rc) return rc;
	if (p->share_type == MSX_MULTI_SHARE_PROPORTIONAL) {
		for (int k = 1; k < 20; k++) {            // msam_profile.c:331; converged iterations exit at once
			if ((rc = msx_prop_iteration(ctx, p, false))) return rc;
			msx_prop_apply_launch(ctx, p, k, true);
		}
		msx_prop_purged_launch(ctx, p, p->counters + 3);
	}
	MSX_HIP(ctx, hipGetLastError());
	return MSX_OK;
}

extern "C" int msx_profile_finalize_dist_enqueue(msx_ctx *ctx, msx_profile *p) {
	if (!ctx || !p) return MSX_ERR_ARG;
	msx_join(ctx);
	int rc = msx_profile_allreduce_counts(ctx, p);           // ui, d, {inserts, uniq, multi} over all shards
	if (rc) return rc;
	if ((rc = msx_profile_prop_begin(ctx, p))) return rc;      // a = U = ui/2 (+d): identical on every rank
	if (p->share_type == MSX_MULTI_SHARE_PROPORTIONAL) {
		// After convergence the local kernels are no-ops and leave `share` at zero on every rank, but a collective that
		// has been enqueued runs: 8 MB of zeros per remaining iteration.  So the loop looks at the convergence flag every
		// `poll` iterations (MSX_DIST_POLL, default 8; 0: never -- enqueue all 19 and return without waiting): one small
		// copy and one wait for the stream, and every rank -- they hold the same all-reduced numbers, hence the same flag
		// (:383) -- stops enqueueing at the same k.  The price is a drained stream per look (~22 us: 4.22 against 4.13 ms
		// for the one-rank c3 step at poll = 4), the gain every all-reduce behind the look that sees the flag.  The c3
		// batch itself runs all 19 iterations (DELTA^2 has not reached 1e-10 by then), so there the looks are pure cost --
		// hence two of them by default, not four; a sample that converges at k = 15 (the e2e file's 2 M-record prefix
		// does) saves the all-reduces of k = 17..19.
		const char *pe = getenv("MSX_DIST_POLL");
		const int poll = (ctx->dist && pe) ? atoi(pe) : (ctx->dist ? 8 : 0);
		// MSX_DIST_SLICES=2..4 (default 1): the local half in slices of the feature range (prop_slice_cuts), slice i's all-reduce
		// on the communicator's side stream while slice i + 1 is computed on the context's -- the 8 MB all-reduce of an
		// iteration is the part of a multi-GPU step no kernel hides otherwise.  The same additions in the same order: every
		// abundance bit as with one slice.  Off by default until a node with more than one GPU has timed it.
		const char *se = getenv("MSX_DIST_SLICES");
		const int slices = (se && atoi(se) >= 2 && atoi(se) <= MSX_MAX_SLICES) ? atoi(se) : 1;
		if (slices > 1 && (rc = prop_slice_cuts(ctx, p, slices))) return rc;
		for (int k = 1; k < 20; k++) {            // msam_profile.c:331
			if (slices > 1) {
				for (int i = 0; i < slices; i++) {
					if ((rc = prop_iteration_range(ctx, p, true, p->slice_wave[i], p->slice_wave[i + 1], p->slice_key[i], p->slice_key[i + 1])))
						return rc;
					if ((rc = msx_dist_allreduce_share_side(ctx, p, (int32_t)p->slice_feat[i], (int32_t)(p->slice_feat[i + 1] - p->slice_feat[i])))) return rc;
				}
				if ((rc = msx_dist_side_join(ctx))) return rc;
			} else {
				if ((rc = msx_prop_iteration(ctx, p, true))) return rc;     // share = this rank's part of the increment, complete
				if ((rc = msx_dist_allreduce_share(ctx, p))) return rc;
			}
			msx_prop_apply_launch(ctx, p, k, false);   // same numbers, same decision (:383) everywhere
			if (poll > 0 && k % poll == 0 && k < 19) {
				int32_t done = 0;
				MSX_HIP(ctx, hipMemcpyAsync(&done, p->iter_state, 4, hipMemcpyDeviceToHost, ctx->stream));
				MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
				if (done) break;
			}
		}
		msx_prop_purged_launch(ctx, p, p->counters + 3);       // this rank's multi-mappers whose sum is 0
		if ((rc = msx_dist_allreduce_u32(ctx, p->counters + 3, 1))) return rc;
	}
	MSX_HIP(ctx, hipGetLastError());
	return MSX_OK;
}

extern "C" int msx_profile_fetch(msx_ctx *ctx, msx_profile *p, double *abundance_host, msx_profile_stats *stats) {
	if (!ctx || !p) return MSX_ERR_ARG;
	if (!p->begun) return msx_fail(ctx, MSX_ERR_ARG, "msx_profile_fetch before finalize/prop_begin");
	msx_join(ctx);
	uint32_t c[4];
	int32_t it[2];
	double dl[20];
	if (abundance_host && p->n_features > 0)
		MSX_HIP(ctx, hipMemcpyAsync(abundance_host, p->a, (size_t)p->n_features * 8, hipMemcpyDeviceToHost, ctx->stream));
	MSX_HIP(ctx, hipMemcpyAsync(c, p->counters, 16, hipMemcpyDeviceToHost, ctx->stream));
	MSX_HIP(ctx, hipMemcpyAsync(it, p->iter_state, 8, hipMemcpyDeviceToHost, ctx->stream));
	MSX_HIP(ctx, hipMemcpyAsync(dl, p->delta, 160, hipMemcpyDeviceToHost, ctx->stream));
	MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (stats) {
		stats->insert_count = c[0];
		stats->uniq_mapper_count = c[1];
		stats->multi_mapper_count = c[2];
		stats->purged_insert_count = c[3];
		stats->converged = it[0];
		stats->iterations = it[1];
		for (int i = 0; i < 20; i++) stats->delta[i] = dl[i];
	}
	return MSX_OK;
}

extern "C" int msx_profile_finalize(msx_ctx *ctx, msx_profile *p, double *abundance_host, msx_profile_stats *stats) {
	int rc = msx_profile_finalize_enqueue(ctx, p);
	if (rc) return rc;
	return msx_profile_fetch(ctx, p, abundance_host, stats);
}

// msx_runtime_warmup: this translation unit's code object loaded onto the device ahead of its first launch (the runtime loads a
// module when one of its kernels is first asked for: 2-10 ms each, otherwise paid by the first batches of a command)
void msx_touch_prop(void) {
	hipFuncAttributes attr;
	(void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_rs_rowscan));
}
