// msx_coverage.hip -- device side of `msamtools coverage` (BASELINE.json
// configs[3]): per-base depth of every reference (msam_coverage.c:33-87, 106-139).
#include "msx_internal.h"

#include <cstdlib>

// ---------------------------------------------------------------------------
// coverage pile-up (msam_coverage.c:33-87).  The reference adds 1 to every base
// of every M/=/X run.  Here a run [p, p+w) is recorded as +1 at p and -1 at p+w
// in the per-base vector (two integer atomics per run instead of w), and
// msx_coverage_finish() turns the differences into depths with one in-place
// inclusive prefix sum over the concatenated targets.  Runs are cut at their
// target's ends (cov_add_run), so the running sum is 0 at every target boundary and
// one scan over all targets is exact (int32 wrap-around identical to counting).
// ---------------------------------------------------------------------------
// +1 at the first base of a run, -1 behind its last.  The reference adds per base with no bounds check
// (a record reaching past its target's end writes past the target's array there); here the run is cut
// at the target's ends, so that a malformed record cannot shift the depths of the targets behind it --
// inside the target the depths are the reference's.
__device__ __forceinline__ void cov_add_run(int32_t *c, int64_t s, int64_t e, int64_t t_len) {
	if (s < 0) s = 0;
	if (e > t_len) e = t_len;
	if (e > s) { atomicAdd(&c[s], 1); atomicAdd(&c[e], -1); }
}

__global__ __launch_bounds__(MSX_BLOCK) void k_coverage_pileup(int64_t n, const int32_t *__restrict__ tid,
                                                               const int32_t *__restrict__ pos,
                                                               const uint32_t *__restrict__ cigar_off,
                                                               const uint32_t *__restrict__ cigar,
                                                               const int64_t *__restrict__ cov_off,
                                                               int32_t *__restrict__ diff,
                                                               uint8_t *__restrict__ covered) {
	const int64_t stride = (int64_t)gridDim.x * MSX_BLOCK;
	for (int64_t i = (int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x; i < n; i += stride) {
		const int32_t t = tid[i];
		if (t < 0) continue;                                   // :42
		if (covered) covered[t] = 1;                           // :45-49 (same value from every lane)
		const int64_t t_beg = cov_off[t], t_len = cov_off[t + 1] - t_beg;
		int32_t *c = diff + t_beg;
		int64_t p = pos[i];
		const uint32_t ks = cigar_off[i], ke = cigar_off[i + 1];
		int64_t run_start = -1;                                // adjacent M/=/X runs merge into one interval
		for (uint32_t k = ks; k < ke; ++k) {
			const uint32_t op = cigar[k] & 0xf, w = cigar[k] >> 4;
			if (op == MSX_OP_MATCH || op == MSX_OP_EQUAL || op == MSX_OP_DIFF) {   // :63-74
				if (run_start < 0) run_start = p;
				p += w;
			} else if (op == MSX_OP_DEL || op == MSX_OP_REF_SKIP) {                // :75-78
				if (run_start >= 0 && w > 0) {
					cov_add_run(c, run_start, p, t_len);
					run_start = -1;
				}
				p += w;
			}
			// I, S, H, P and unknown ops: no reference bases, the covered interval continues
		}
		if (run_start >= 0) cov_add_run(c, run_start, p, t_len);
	}
}

// ---------------------------------------------------------------------------
// Large batches: the binned pile-up.  Scattered global atomics run memory-side at
// ~27 G/s on this chip (one per run end: 100 M of them for 50 M reads), so for a batch
// that is large against the depth array the +1 / -1 marks are not added one by one:
//   k_cov_emit   the two ends of a record's first run become items (one word: cell << 1 | sign)
//                in the record's own two slots
//   radix sort   of the items by tile = the bits above the cell-inside-the-tile field (keys only,
//                2 passes for 8 K-cell tiles of a 250 M-cell array)
//   k_cov_tile   one workgroup per chunk of sorted items adds them into an LDS image of their tile
//                (ds_add) and the image onto the depth array with atomic adds of consecutive cells
// The depth array keeps holding differences, so both paths can feed one sample.
// ---------------------------------------------------------------------------
#define CV_TILE_SHIFT 13
#define CV_TILE (1u << CV_TILE_SHIFT)
// Every record owns two item slots (2i, 2i+1): the ends of its first run of M/=/X bases, which is all a read
// without D/N has.  No counter and no prefix -- a shared counter costs one same-address atomic per wave
// (~12 ns each, 9 ms for 50 M reads) -- at the price of sorting the empty slots of unmapped records along
// (key n_tiles, behind every tile).  Further runs of a record (after a D or N) go to the depth array directly.
__global__ __launch_bounds__(MSX_BLOCK) void k_cov_emit(int64_t n, const int32_t *__restrict__ tid,
                                                        const int32_t *__restrict__ pos,
                                                        const uint32_t *__restrict__ cigar_off,
                                                        const uint32_t *__restrict__ cigar,
                                                        const int64_t *__restrict__ cov_off, int32_t *__restrict__ diff,
                                                        uint8_t *__restrict__ covered, uint2 *__restrict__ items,
                                                        uint32_t null_item) {
	const int64_t stride = (int64_t)gridDim.x * MSX_BLOCK;
	for (int64_t i = (int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x; i < n; i += stride) {
		uint2 it = make_uint2(null_item, null_item);
		const int32_t t = tid[i];
		if (t >= 0) {                                        // :42
			if (covered) covered[t] = 1;                     // :45-49
			const int64_t t_beg = cov_off[t], t_len = cov_off[t + 1] - t_beg;
			int64_t p = pos[i];
			const uint32_t ks = cigar_off[i], ke = cigar_off[i + 1];
			int64_t run_start = -1;
			bool first = true;
			auto mark = [&](int64_t s, int64_t e) {
				if (s < 0) s = 0;
				if (e > t_len) e = t_len;
				if (e <= s) return;
				if (first) {
					first = false;
					it = make_uint2((uint32_t)(t_beg + s) << 1, (uint32_t)(t_beg + e) << 1 | 1u);
				} else { atomicAdd(&diff[t_beg + s], 1); atomicAdd(&diff[t_beg + e], -1); }
			};
			for (uint32_t k = ks; k < ke; ++k) {
				const uint32_t op = cigar[k] & 0xf, w = cigar[k] >> 4;
				if (op == MSX_OP_MATCH || op == MSX_OP_EQUAL || op == MSX_OP_DIFF) {   // :63-74
					if (run_start < 0) run_start = p;
					p += w;
				} else if (op == MSX_OP_DEL || op == MSX_OP_REF_SKIP) {                // :75-78
					if (run_start >= 0 && w > 0) { mark(run_start, p); run_start = -1; }
					p += w;
				}
			}
			if (run_start >= 0) mark(run_start, p);
		}
		items[i] = it;
	}
}

// first item of every tile among the sorted keys (lower bounds, one thread per tile; tile_start[n_tiles] = n)
__global__ __launch_bounds__(MSX_BLOCK) void k_cov_tile_starts(const uint32_t *__restrict__ ikey, int64_t n_items,
                                                               int64_t n_tiles, uint32_t *__restrict__ tile_start) {
	const int64_t t = (int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x;
	if (t > n_tiles) return;
	int64_t lo = 0, hi = n_items;
	while (lo < hi) {
		const int64_t mid = (lo + hi) >> 1;
		if ((int64_t)(ikey[mid] >> (CV_TILE_SHIFT + 1)) < t) lo = mid + 1; else hi = mid;
	}
	tile_start[t] = (uint32_t)lo;
}

// One workgroup per CHUNK of CV_CHUNK sorted items (equal work whatever the skew: the hottest reference of a
// metagenome receives millions of marks, all in one tile), tile by tile inside the chunk: the tile's items of
// this chunk are added into an LDS image (ds_add), and the image onto the tile's cells with atomic adds of
// consecutive cells (a tile can be shared with the neighbouring chunks; 256 contiguous bytes per wave
// instruction is the shape the memory-side adders take at full rate, and rows of zeros are skipped).
#define CV_CHUNK 8192
__global__ __launch_bounds__(MSX_BLOCK) void k_cov_tile(const uint32_t *__restrict__ items, int64_t n_tiles, const uint32_t *__restrict__ tile_start,
                                                        int64_t total_cells, int32_t *__restrict__ diff) {
	__shared__ int32_t s_d[CV_TILE];
	const int64_t n = (int64_t)tile_start[n_tiles];          // the empty slots sort behind the last tile
	const int64_t lo_c = (int64_t)blockIdx.x * CV_CHUNK;
	if (lo_c >= n) return;
	const int64_t hi_c = lo_c + CV_CHUNK < n ? lo_c + CV_CHUNK : n;
	const uint32_t t_first = items[lo_c] >> (CV_TILE_SHIFT + 1), t_last = items[hi_c - 1] >> (CV_TILE_SHIFT + 1);
	for (uint32_t t = t_first; t <= t_last; ++t) {
		const int64_t a = (int64_t)tile_start[t] > lo_c ? (int64_t)tile_start[t] : lo_c;
		const int64_t b = (int64_t)tile_start[t + 1] < hi_c ? (int64_t)tile_start[t + 1] : hi_c;
		if (a >= b) continue;                                   // (workgroup-uniform)
		for (uint32_t q = threadIdx.x; q < CV_TILE; q += MSX_BLOCK) s_d[q] = 0;
		__syncthreads();
		for (int64_t q = a + threadIdx.x; q < b; q += MSX_BLOCK) {
			const uint32_t v = items[q];
			atomicAdd(&s_d[(v >> 1) & (CV_TILE - 1)], (v & 1u) ? -1 : 1);
		}
		__syncthreads();
		const int64_t c0 = (int64_t)t << CV_TILE_SHIFT;
		for (uint32_t q = threadIdx.x; q < CV_TILE; q += MSX_BLOCK) {
			const int32_t d = s_d[q];
			if (__ballot(d != 0) != 0ull && c0 + q <= total_cells) atomicAdd(&diff[c0 + q], d);
		}
		__syncthreads();
	}
}

extern "C" int msx_coverage_accumulate(msx_ctx *ctx, const msx_batch *b, const int64_t *cov_off, int32_t n_targets,
                                       int64_t total_len, int32_t *cov, uint8_t *covered) {
	if (!ctx || !b || !cov_off || !cov) return MSX_ERR_ARG;
	if (!b->pos || !b->tid || !b->cigar_off || !b->cigar)
		return msx_fail(ctx, MSX_ERR_ARG, "msx_coverage_accumulate needs tid, pos and cigar arrays");
	msx_join(ctx);
	if (b->n_records == 0) return MSX_OK;
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	static const int64_t binned_from = [] {
		const char *e = getenv("MSX_COV_BINNED_FROM");          // records per batch from which the binned path is used
		return e ? atoll(e) : (int64_t)(2 << 20);
	}();
	// (total_len = cov_off[n_targets], which the caller summed itself: reading it back cost a stream synchronisation per batch)
	const int64_t total_cells = (b->n_records >= binned_from && n_targets > 0) ? total_len : 0;
	// an item is one word (cell << 1 | sign) and tile n_tiles marks an empty slot: depth arrays of 2^31 cells
	// and more take the direct path
	if (total_cells > 0 && total_cells + 2 * (int64_t)CV_TILE < ((int64_t)1 << 31)) {
		const int64_t n_items = 2 * b->n_records;
		const int64_t n_tiles = (total_cells + 1 + CV_TILE - 1) >> CV_TILE_SHIFT;
		if (n_items >= ((int64_t)1 << 32)) return msx_fail(ctx, MSX_ERR_ARG, "msx_coverage_accumulate: batch too large");
		int bits = 1;
		while (((int64_t)1 << bits) <= n_tiles) bits++;          // tiles 0 .. n_tiles (n_tiles = an empty slot)
		int rc;
		for (int q = 0; q < 2; q++)
			if ((rc = msx_reserve(ctx, &ctx->cv_key[q], (size_t)(n_items + 64) * 4))) return rc;
		msx_time_begin(ctx, MSX_K_COVERAGE);
		hipLaunchKernelGGL(k_cov_emit, dim3(msx_grid_x(ctx, b->n_records, MSX_BLOCK, 8)), dim3(MSX_BLOCK), 0, ctx->stream,
		                   b->n_records, b->tid, b->pos, b->cigar_off, b->cigar, cov_off, cov, covered,
		                   (uint2 *)ctx->cv_key[0].p, (uint32_t)n_tiles << (CV_TILE_SHIFT + 1));
		int sel = 0;
		if ((rc = msx_sort_keys32(ctx, (uint32_t *)ctx->cv_key[0].p, (uint32_t *)ctx->cv_key[1].p, n_items, CV_TILE_SHIFT + 1,
		                          bits, &ctx->cv_hist, &ctx->cv_off, &sel)))
			return rc;
		if ((rc = msx_reserve(ctx, &ctx->cv_start, (size_t)(n_tiles + 8) * 4))) return rc;
		hipLaunchKernelGGL(k_cov_tile_starts, dim3((unsigned)((n_tiles + 1 + MSX_BLOCK - 1) / MSX_BLOCK)), dim3(MSX_BLOCK), 0,
		                   ctx->stream, (const uint32_t *)ctx->cv_key[sel].p, n_items, n_tiles, (uint32_t *)ctx->cv_start.p);
		hipLaunchKernelGGL(k_cov_tile, dim3((unsigned)((n_items + CV_CHUNK - 1) / CV_CHUNK)), dim3(MSX_BLOCK), 0, ctx->stream,
		                   (const uint32_t *)ctx->cv_key[sel].p, n_tiles, (const uint32_t *)ctx->cv_start.p, total_cells, cov);
		msx_time_end(ctx);
		MSX_HIP(ctx, hipGetLastError());
		return MSX_OK;
	}
	msx_time_begin(ctx, MSX_K_COVERAGE);
	hipLaunchKernelGGL(k_coverage_pileup, dim3(msx_grid(ctx, b->n_records, MSX_BLOCK)), dim3(MSX_BLOCK), 0,
	                   ctx->stream, b->n_records, b->tid, b->pos, b->cigar_off, b->cigar, cov_off, cov, covered);
	msx_time_end(ctx);
	MSX_HIP(ctx, hipGetLastError());
	return MSX_OK;
}

extern "C" int msx_coverage_finish(msx_ctx *ctx, int32_t *cov, int64_t total_len) {
	if (!ctx || !cov || total_len < 0) return MSX_ERR_ARG;
	msx_join(ctx);
	if (total_len == 0) return MSX_OK;
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	return msx_scan_inclusive_u32(ctx, (uint32_t *)cov, total_len);
}

// mWriteCoverageSummaryToStream (msam_coverage.c:188-219) prints, per target, the fraction of positions with a depth
// other than 0 and the mean depth: two sums over the target's cells.  One wave per target; the sums leave the device,
// the depths need not (a million references of 4.5 kb are 18 GB of them).
__global__ __launch_bounds__(64) void k_coverage_summary(const int32_t *__restrict__ cov, const int64_t *__restrict__ off,
                                                         int32_t n_targets, int64_t *__restrict__ touched, int64_t *__restrict__ sum) {
	const int32_t t = (int32_t)blockIdx.x;
	if (t >= n_targets) return;
	const int64_t lo = off[t], hi = off[t + 1];
	int64_t a = 0, b = 0;
	for (int64_t i = lo + threadIdx.x; i < hi; i += 64) {
		const int32_t v = cov[i];
		a += v != 0;
		b += v;
	}
	for (int step = 32; step >= 1; step >>= 1) {
		a += __shfl_xor(a, step);
		b += __shfl_xor(b, step);
	}
	if (threadIdx.x == 0) { touched[t] = a; sum[t] = b; }
}

extern "C" int msx_coverage_summary(msx_ctx *ctx, const int32_t *cov, const int64_t *cov_off, int32_t n_targets,
                                    int64_t *touched_host, int64_t *sum_host) {
	if (!ctx || !cov || !cov_off || n_targets < 0 || !touched_host || !sum_host) return MSX_ERR_ARG;
	msx_join(ctx);
	if (n_targets == 0) return MSX_OK;
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	int rc;
	const size_t bytes = (size_t)n_targets * 8;
	if ((rc = msx_reserve(ctx, &ctx->cv_start, 2 * bytes))) return rc;
	int64_t *d_t = (int64_t *)ctx->cv_start.p, *d_s = d_t + n_targets;
	hipLaunchKernelGGL(k_coverage_summary, dim3((unsigned)n_targets), dim3(64), 0, ctx->stream, cov, cov_off, n_targets, d_t, d_s);
	MSX_HIP(ctx, hipGetLastError());
	MSX_HIP(ctx, hipMemcpyAsync(touched_host, d_t, bytes, hipMemcpyDeviceToHost, ctx->stream));
	MSX_HIP(ctx, hipMemcpyAsync(sum_host, d_s, bytes, hipMemcpyDeviceToHost, ctx->stream));
	MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return MSX_OK;
}
