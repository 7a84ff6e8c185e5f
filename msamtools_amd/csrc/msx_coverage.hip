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

// ---------------------------------------------------------------------------
// A whole sample in one batch: depths written once (msx_coverage_depths).
//
// The streamed path above keeps a difference array in global memory -- zeroed by the caller, marked batch after
// batch, summed in place at the end: the 1 GB array of the c4 workload is touched three times and the marks arrive
// as atomic adds.  When the batch IS the sample the depths of a tile can be finished where its marks are gathered:
//   k_cov_emit2     a record's first run as two items in its own slots -- item = sign << S | cell: the sign sits ABOVE
//                   the tile number, so the sort leaves all +1 marks by tile, then all -1 marks by tile, and the
//                   running depth at a tile's first cell is (+1 marks in front of the tile) - (-1 marks in front of
//                   it): two positions in the sorted array, no pass over the marks; further runs of a record (after
//                   a D or N) go to one of 64 overflow lists (one LDS-staged append per workgroup, so no same-address
//                   traffic); the first radix pass's digit counts of the workgroup's 4096 items are left in the
//                   sort's table on the way (one read of the items saved)
//   radix sort      by (sign, tile), keys only
//   k_cov_starts2   where every (sign, tile) begins
//   k_cov_heavy_*   tiles with more marks than a workgroup should walk alone (hot references of a skewed community)
//                   are pre-reduced: every chunk of the sorted array that touches such a tile adds its LDS image to
//                   the tile's image in a side buffer (atomic adds of consecutive cells)
//   k_cov_depths    one workgroup per 8 K-cell tile: marks into an LDS image (or the side buffer's), inclusive scan,
//                   + the tile's incoming depth, written out as 16-byte vectors -- the depth array's only touch.
// Integer work, HBM-bound: no MFMA.
// ---------------------------------------------------------------------------
#define CV2_REC 2048                    // records per workgroup of k_cov_emit2 (= MSX_SORT_TILE / 2)
#define CV2_LISTS 64                    // overflow lists
#define CV2_STAGE 2048                  // overflow items a workgroup can stage
#define CV2_HEAVY 32768u                // marks of one sign in a tile from which the tile is pre-reduced
#define CV2_HEAVY_CAP 2048              // side-buffer slots
struct cv2_state {
	uint32_t list_n[CV2_LISTS];         // items in every overflow list
	uint32_t overflow;                  // a list or a stage was too small: the caller takes the streamed path
	uint32_t n_heavy;
	uint32_t n_hchunks;                 // chunks of the sorted items that touch a pre-reduced tile
};

__global__ __launch_bounds__(MSX_BLOCK) void k_cov_emit2(int64_t n, const int32_t *__restrict__ tid, const int32_t *__restrict__ pos,
                                                         const uint32_t *__restrict__ cigar_off, const uint32_t *__restrict__ cigar,
                                                         const int64_t *__restrict__ cov_off, uint8_t *__restrict__ covered,
                                                         uint32_t *__restrict__ items, int sign_shift, int64_t list_base, uint32_t list_cap,
                                                         cv2_state *__restrict__ st, uint32_t *__restrict__ hist, int64_t hist_tiles, int hist_shift) {
	__shared__ uint32_t s_extra[CV2_STAGE];
	__shared__ uint32_t s_n, s_base;
	__shared__ uint32_t s_cnt[MSX_BLOCK / 64][256];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	if (threadIdx.x == 0) s_n = 0;
	for (int q = 0; q < 4; q++) s_cnt[w][lane + 64 * q] = 0;
	__syncthreads();
	const uint32_t sgn = 1u << sign_shift;
	for (int q = 0; q < CV2_REC / MSX_BLOCK; q++) {
		const int64_t i = (int64_t)blockIdx.x * CV2_REC + (int64_t)q * MSX_BLOCK + threadIdx.x;
		if (i >= n) break;
		uint2 it = make_uint2(0xffffffffu, 0xffffffffu);
		const int32_t t = tid[i];
		if (t >= 0) {                                        // msam_coverage.c:42
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
				const uint32_t a = (uint32_t)(t_beg + s), b = (uint32_t)(t_beg + e) | sgn;
				if (first) {
					first = false;
					it = make_uint2(a, b);
				} else {
					const uint32_t k = atomicAdd(&s_n, 2u);
					if (k + 2u <= CV2_STAGE) { s_extra[k] = a; s_extra[k + 1u] = b; }
				}
			};
			for (uint32_t k = ks; k < ke; ++k) {
				const uint32_t op = cigar[k] & 0xf, wd = cigar[k] >> 4;
				if (op == MSX_OP_MATCH || op == MSX_OP_EQUAL || op == MSX_OP_DIFF) {   // :63-74
					if (run_start < 0) run_start = p;
					p += wd;
				} else if (op == MSX_OP_DEL || op == MSX_OP_REF_SKIP) {                // :75-78
					if (run_start >= 0 && wd > 0) { mark(run_start, p); run_start = -1; }
					p += wd;
				}
			}
			if (run_start >= 0) mark(run_start, p);
		}
		reinterpret_cast<uint2 *>(items)[i] = it;
		// the first radix pass's digits of the two items (an empty slot counts too: it is a key like any other)
		atomicAdd(&s_cnt[w][(it.x >> hist_shift) & 255u], 1u);
		atomicAdd(&s_cnt[w][(it.y >> hist_shift) & 255u], 1u);
	}
	__syncthreads();
	{
		const int d = threadIdx.x;
		hist[(int64_t)d * hist_tiles + blockIdx.x] = s_cnt[0][d] + s_cnt[1][d] + s_cnt[2][d] + s_cnt[3][d];
	}
	const uint32_t ne = s_n;
	if (ne == 0) return;
	if (ne > CV2_STAGE) { if (threadIdx.x == 0) st->overflow = 1; return; }
	const uint32_t li = blockIdx.x & (CV2_LISTS - 1);
	if (threadIdx.x == 0) s_base = atomicAdd(&st->list_n[li], ne);
	__syncthreads();
	const uint32_t b0 = s_base;
	if (b0 + ne > list_cap) { if (threadIdx.x == 0) st->overflow = 1; return; }
	for (uint32_t k = threadIdx.x; k < ne; k += MSX_BLOCK) items[list_base + (int64_t)li * list_cap + b0 + k] = s_extra[k];
}

// the unused part of every overflow list: empty slots
__global__ __launch_bounds__(MSX_BLOCK) void k_cov_fill_lists(uint32_t *__restrict__ items, int64_t list_base, uint32_t list_cap,
                                                              const cv2_state *__restrict__ st) {
	const uint32_t li = blockIdx.y;
	const uint32_t used = st->list_n[li] < list_cap ? st->list_n[li] : list_cap;
	for (uint32_t k = used + blockIdx.x * MSX_BLOCK + threadIdx.x; k < list_cap; k += gridDim.x * MSX_BLOCK)
		items[list_base + (int64_t)li * list_cap + k] = 0xffffffffu;
}

// start[s * (n_tiles + 1) + t]: first sorted item at or behind (sign s, tile t); t = n_tiles: where sign s ends
__global__ __launch_bounds__(MSX_BLOCK) void k_cov_starts2(const uint32_t *__restrict__ ikey, int64_t n_items, int64_t n_tiles, int sign_shift,
                                                           uint32_t *__restrict__ start) {
	const int64_t q = (int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x;
	if (q >= 2 * (n_tiles + 1)) return;
	const uint32_t s = q >= n_tiles + 1 ? 1u : 0u;
	const int64_t t = q - (int64_t)s * (n_tiles + 1);
	const uint64_t want = ((uint64_t)s << sign_shift) + ((uint64_t)t << CV_TILE_SHIFT);     // (t = n_tiles of sign 1 passes 32 bits: every real item is below)
	int64_t lo = 0, hi = n_items;
	while (lo < hi) {
		const int64_t mid = (lo + hi) >> 1;
		if ((uint64_t)ikey[mid] < want) lo = mid + 1; else hi = mid;
	}
	start[q] = (uint32_t)lo;
}

__global__ __launch_bounds__(MSX_BLOCK) void k_cov_heavy_list(const uint32_t *__restrict__ start, int64_t n_tiles, int32_t *__restrict__ slot_of,
                                                              cv2_state *__restrict__ st, uint32_t heavy_from) {
	const int64_t t = (int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x;
	if (t >= n_tiles) return;
	const uint32_t np = start[t + 1] - start[t], nm = start[n_tiles + 1 + t + 1] - start[n_tiles + 1 + t];
	int32_t slot = -1;
	if (np > heavy_from || nm > heavy_from) {
		const uint32_t k = atomicAdd(&st->n_heavy, 1u);
		if (k < CV2_HEAVY_CAP) slot = (int32_t)k; else st->overflow = 1;
	}
	slot_of[t] = slot;
}

// the chunks of the sorted array that hold marks of a pre-reduced tile.  Such a tile has more marks of one sign than a
// chunk holds, so inside a chunk it can only be the tile of the chunk's first or of its last item: one look at each.
__global__ __launch_bounds__(MSX_BLOCK) void k_cov_heavy_chunks(const uint32_t *__restrict__ items, const uint32_t *__restrict__ start, int64_t n_tiles,
                                                                int sign_shift, const int32_t *__restrict__ slot_of, cv2_state *__restrict__ st,
                                                                uint32_t *__restrict__ chunk_list) {
	const int64_t n = (int64_t)start[2 * n_tiles + 1];
	const int64_t c = (int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x;
	const int64_t lo = c * CV_CHUNK;
	bool hit = false;
	if (st->n_heavy != 0 && lo < n) {
		const int64_t hi = lo + CV_CHUNK < n ? lo + CV_CHUNK : n;
		const uint32_t smask = (1u << sign_shift) - 1u;
		hit = slot_of[(items[lo] & smask) >> CV_TILE_SHIFT] >= 0 || slot_of[(items[hi - 1] & smask) >> CV_TILE_SHIFT] >= 0;
	}
	const unsigned long long m = __ballot(hit);
	if (m) {                                                 // one append per wave
		const int lane = threadIdx.x & 63, lead = __ffsll((long long)m) - 1;
		uint32_t base = 0;
		if (lane == lead) base = atomicAdd(&st->n_hchunks, (uint32_t)__popcll(m));
		base = (uint32_t)__shfl((int)base, lead, 64);
		if (hit) chunk_list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)c;
	}
}

__global__ __launch_bounds__(MSX_BLOCK) void k_cov_heavy_zero(int32_t *__restrict__ side, const cv2_state *__restrict__ st) {
	if (blockIdx.x >= st->n_heavy || blockIdx.x >= CV2_HEAVY_CAP) return;
	int4 *p = reinterpret_cast<int4 *>(side + (size_t)blockIdx.x * CV_TILE);
	for (uint32_t q = threadIdx.x; q < CV_TILE / 4; q += MSX_BLOCK) p[q] = make_int4(0, 0, 0, 0);
}

// the listed chunks of the sorted items: the part that belongs to pre-reduced tiles is added to their side images
__global__ __launch_bounds__(MSX_BLOCK) void k_cov_heavy_add(const uint32_t *__restrict__ items, const uint32_t *__restrict__ start, int64_t n_tiles,
                                                             int sign_shift, const int32_t *__restrict__ slot_of, int32_t *__restrict__ side,
                                                             const cv2_state *__restrict__ st, const uint32_t *__restrict__ chunk_list) {
	__shared__ int32_t s_d[CV_TILE];
	const int64_t n = (int64_t)start[2 * n_tiles + 1];       // where the -1 marks end: the empty slots sort behind
	for (uint32_t ci = blockIdx.x; ci < st->n_hchunks; ci += gridDim.x) {
	const int64_t lo_c = (int64_t)chunk_list[ci] * CV_CHUNK;
	if (lo_c >= n) continue;
	const int64_t hi_c = lo_c + CV_CHUNK < n ? lo_c + CV_CHUNK : n;
	const uint32_t smask = (1u << sign_shift) - 1u;
	// the (sign, tile) pairs this chunk touches: walk them by their starts
	int64_t a = lo_c;
	while (a < hi_c) {
		const uint32_t v0 = items[a], s = v0 >> sign_shift, t = (v0 & smask) >> CV_TILE_SHIFT;
		const int64_t e_t = (int64_t)start[(int64_t)s * (n_tiles + 1) + t + 1];
		const int64_t b = e_t < hi_c ? e_t : hi_c;
		const int32_t slot = slot_of[t];
		if (slot >= 0) {                                         // (workgroup-uniform)
			for (uint32_t q = threadIdx.x; q < CV_TILE; q += MSX_BLOCK) s_d[q] = 0;
			__syncthreads();
			for (int64_t q = a + threadIdx.x; q < b; q += MSX_BLOCK) atomicAdd(&s_d[items[q] & (CV_TILE - 1)], s ? -1 : 1);
			__syncthreads();
			int32_t *img = side + (size_t)slot * CV_TILE;
			for (uint32_t q = threadIdx.x; q < CV_TILE; q += MSX_BLOCK) {
				const int32_t d = s_d[q];
				if (__ballot(d != 0) != 0ull) atomicAdd(&img[q], d);
			}
			__syncthreads();
		}
		a = b > a ? b : a + 1;
	}
	}
}

// one workgroup per tile: the tile's marks as an LDS image, its inclusive sum + the depth the tile starts from
__global__ __launch_bounds__(MSX_BLOCK) void k_cov_depths(const uint32_t *__restrict__ items, const uint32_t *__restrict__ start, int64_t n_tiles,
                                                          const int32_t *__restrict__ slot_of, const int32_t *__restrict__ side,
                                                          int64_t total_cells, int32_t *__restrict__ cov) {
	__shared__ int32_t s_d[CV_TILE];
	__shared__ int32_t s_w[MSX_BLOCK / 64];
	const int64_t t = blockIdx.x;
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const uint32_t ps = start[t], pe = start[t + 1], ms = start[n_tiles + 1 + t], me = start[n_tiles + 1 + t + 1];
	const int32_t carry = (int32_t)(ps - start[0]) - (int32_t)(ms - start[n_tiles + 1]);
	const int32_t slot = slot_of[t];
	if (slot >= 0) {
		const int4 *img = reinterpret_cast<const int4 *>(side + (size_t)slot * CV_TILE);
		for (uint32_t q = threadIdx.x; q < CV_TILE / 4; q += MSX_BLOCK) reinterpret_cast<int4 *>(s_d)[q] = img[q];
	} else {
		for (uint32_t q = threadIdx.x; q < CV_TILE / 4; q += MSX_BLOCK) reinterpret_cast<int4 *>(s_d)[q] = make_int4(0, 0, 0, 0);
		__syncthreads();
		for (uint32_t q = ps + threadIdx.x; q < pe; q += MSX_BLOCK) atomicAdd(&s_d[items[q] & (CV_TILE - 1)], 1);
		for (uint32_t q = ms + threadIdx.x; q < me; q += MSX_BLOCK) atomicAdd(&s_d[items[q] & (CV_TILE - 1)], -1);
	}
	__syncthreads();
	const int64_t c0 = t << CV_TILE_SHIFT;
	int32_t running = carry;
	for (uint32_t base = 0; base < CV_TILE; base += MSX_BLOCK * 4) {
		const uint32_t q = base + threadIdx.x * 4u;
		int4 v = *reinterpret_cast<const int4 *>(&s_d[q]);
		v.y += v.x; v.z += v.y; v.w += v.z;
		int32_t inc = v.w;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const int32_t u = __shfl_up(inc, o, 64);
			if (lane >= o) inc += u;
		}
		if (lane == 63) s_w[w] = inc;
		__syncthreads();
		int32_t woff = 0, tot = 0;
		for (int k = 0; k < MSX_BLOCK / 64; k++) { if (k < w) woff += s_w[k]; tot += s_w[k]; }
		const int32_t add = running + woff + inc - v.w;
		v.x += add; v.y += add; v.z += add; v.w += add;
		const int64_t c = c0 + q;
		if (c + 4 <= total_cells + 1) {                      // (non-temporal: see k_cov_depths3)
			typedef int v4i __attribute__((ext_vector_type(4)));
			v4i x = {v.x, v.y, v.z, v.w};
			__builtin_nontemporal_store(x, reinterpret_cast<v4i *>(&cov[c]));
		}
		else {
			if (c <= total_cells) cov[c] = v.x;
			if (c + 1 <= total_cells) cov[c + 1] = v.y;
			if (c + 2 <= total_cells) cov[c + 2] = v.z;
		}
		running += tot;
		__syncthreads();
	}
}

// ---------------------------------------------------------------------------------------------------------------------
// The same, one word per run (round 4, second form): an item is the PIECE of a run inside one 4 K-cell tile,
//   key = (tile & 255) << 24 | cell in tile << 12 | (length - 1),   sup = tile >> 8   (an 8-bit value beside the key)
// -- 28 bits of cell address and 12 of length do not fit one word, so the top byte of the tile number is a byte of its
// own beside the key: the first radix pass orders by that byte and drops it (from then on the position tells it: every
// byte value's run begins at a whole sort tile), the second orders every such run by the key's top byte
// (msx_sort_k32v8); 5 + 4 bytes move per run where the first form moves 8 + 8 (a +1 and a -1 mark, two passes).
// A run that crosses a tile boundary is cut there (the further
// pieces go to the overflow lists, like the runs behind a D or N): every tile then holds every piece that covers it,
// the depth at its first cell included, and no depth is carried from tile to tile.  Up to 255 * 2^20 cells (the byte 255
// marks an empty slot, which sorts behind everything); larger samples take the first form.
#define CV3_TILE_SHIFT 12
#define CV3_TILE (1u << CV3_TILE_SHIFT)
#define CV3_REC MSX_SORT_TILE           // records per workgroup of k_cov_emit3: one sort tile of own slots
#define CV3_STAGE 2048
#define CV3_MAX_TILES (255 * 256)
#define CV3_FLIGHT 4                    // records a thread of k_cov_emit3 has in flight (8: 86 registers, 5 waves per SIMD, 467 us against 381)
#define CV3_PICK(a) (u == 0 ? a[0] : u == 1 ? a[1] : u == 2 ? a[2] : a[3])

// (first cell, length) of every target in one 8-byte word: the record loop gathers it once per record
__global__ __launch_bounds__(MSX_BLOCK) void k_cov_targets3(const int64_t *__restrict__ cov_off, int32_t n_targets, uint2 *__restrict__ targets) {
	const int32_t t = (int32_t)(blockIdx.x * MSX_BLOCK + threadIdx.x);
	if (t < n_targets) targets[t] = make_uint2((uint32_t)cov_off[t], (uint32_t)(cov_off[t + 1] - cov_off[t]));
}

__global__ __launch_bounds__(MSX_BLOCK) void k_cov_emit3(int64_t n, const int32_t *__restrict__ tid, const int32_t *__restrict__ pos,
                                                         const uint32_t *__restrict__ cigar_off, const uint32_t *__restrict__ cigar,
                                                         const uint2 *__restrict__ targets, uint8_t *__restrict__ covered,
                                                         uint32_t *__restrict__ items, uint8_t *__restrict__ sups, int64_t list_base,
                                                         uint32_t list_cap, cv2_state *__restrict__ st, uint32_t *__restrict__ hist,
                                                         int64_t hist_tiles) {
	__shared__ uint32_t s_extra[CV3_STAGE];
	__shared__ uint8_t s_esup[CV3_STAGE];
	__shared__ uint32_t s_n, s_base;
	__shared__ uint32_t s_cnt[MSX_BLOCK / 64][256];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	if (threadIdx.x == 0) s_n = 0;
	for (int q = 0; q < 4; q++) s_cnt[w][lane + 64 * q] = 0;
	__syncthreads();
	// Four records per thread at a time: their fields (straight-line loads, indices clamped instead of branches), then what
	// those point at (the target's first cell and length, the first CIGAR word), then ONE copy of the walk run four times
	// over values picked out of the four -- with one record after the other every load of the chain was exposed and a wave
	// spent 9 K cycles per record; with the walk written out four times the code no longer fit and it got slower.
	for (int q0 = 0; q0 < CV3_REC / MSX_BLOCK; q0 += CV3_FLIGHT) {
		const int64_t i0 = (int64_t)blockIdx.x * CV3_REC + (int64_t)q0 * MSX_BLOCK + threadIdx.x;
		if (i0 >= n) {            // (the last workgroup's slots behind the records: empty keys, counted like any other)
#pragma unroll
			for (int u = 0; u < CV3_FLIGHT; u++) {
				const int64_t i = i0 + (int64_t)u * MSX_BLOCK;
				items[i] = 0xffffffffu;
				sups[i] = 0xffu;
				atomicAdd(&s_cnt[w][255], 1u);
			}
			continue;
		}
		int32_t t4[CV3_FLIGHT], p4[CV3_FLIGHT];
		uint32_t ks4[CV3_FLIGHT], ke4[CV3_FLIGHT], c04[CV3_FLIGHT];
		uint2 tt4[CV3_FLIGHT];
#pragma unroll
		for (int u = 0; u < CV3_FLIGHT; u++) {
			const int64_t i = i0 + (int64_t)u * MSX_BLOCK, ic = i < n ? i : n - 1;
			t4[u] = tid[ic];
			p4[u] = pos[ic];
			ks4[u] = cigar_off[ic];
			ke4[u] = cigar_off[ic + 1];
		}
#pragma unroll
		for (int u = 0; u < CV3_FLIGHT; u++) {
			if (i0 + (int64_t)u * MSX_BLOCK >= n) t4[u] = -1;
			tt4[u] = targets[t4[u] >= 0 ? t4[u] : 0];
			c04[u] = cigar[ke4[u] > ks4[u] ? ks4[u] : 0u];
		}
#pragma nounroll
		for (int u = 0; u < CV3_FLIGHT; u++) {
			const int64_t i = i0 + (int64_t)u * MSX_BLOCK;
			const int32_t t = CV3_PICK(t4);                      // (-1 behind the last record: an empty slot)
			uint32_t key = 0xffffffffu, sup = 0xffu;
			if (t >= 0) {                                        // msam_coverage.c:42
				if (covered) covered[t] = 1;                     // :45-49
				const uint2 tt = CV3_PICK(tt4);    // (first cell, length)
				const int64_t t_beg = tt.x, t_len = tt.y;
				int64_t p = CV3_PICK(p4);
				const uint32_t ks = CV3_PICK(ks4);
				const uint32_t ke = CV3_PICK(ke4);
				const uint32_t c0 = CV3_PICK(c04);
				int64_t run_start = -1;
				bool first = true;
				auto mark = [&](int64_t s, int64_t e) {
					if (s < 0) s = 0;
					if (e > t_len) e = t_len;
					if (e <= s) return;
					uint32_t a = (uint32_t)(t_beg + s);
					const uint32_t b = (uint32_t)(t_beg + e);
					while (a < b) {                                  // piece by piece, tile by tile
						const uint32_t tile_end = (a | (CV3_TILE - 1u)) + 1u, pe = b < tile_end ? b : tile_end;
						const uint32_t k1 = ((a >> CV3_TILE_SHIFT) & 255u) << 24 | (a & (CV3_TILE - 1u)) << 12 | (pe - a - 1u);
						const uint32_t s1 = a >> (CV3_TILE_SHIFT + 8);
						if (first) {
							first = false;
							key = k1; sup = s1;
						} else {
							const uint32_t k = atomicAdd(&s_n, 1u);
							if (k < CV3_STAGE) { s_extra[k] = k1; s_esup[k] = (uint8_t)s1; }
						}
						a = pe;
					}
				};
				for (uint32_t k = ks; k < ke; ++k) {
					const uint32_t cw = k == ks ? c0 : cigar[k];
					const uint32_t op = cw & 0xf, wd = cw >> 4;
					if (op == MSX_OP_MATCH || op == MSX_OP_EQUAL || op == MSX_OP_DIFF) {   // :63-74
						if (run_start < 0) run_start = p;
						p += wd;
					} else if (op == MSX_OP_DEL || op == MSX_OP_REF_SKIP) {                // :75-78
						if (run_start >= 0 && wd > 0) { mark(run_start, p); run_start = -1; }
						p += wd;
					}
				}
				if (run_start >= 0) mark(run_start, p);
			}
			items[i] = key;
			sups[i] = (uint8_t)sup;
			atomicAdd(&s_cnt[w][sup], 1u);                       // the first radix pass's digit (an empty slot is a key like any other)
		}
	}
	__syncthreads();
	if (hist) {      // (msx_coverage_collect appends batch after batch: the table's stride is not known yet, the sort counts)
		const int d = threadIdx.x;
		hist[(int64_t)d * hist_tiles + blockIdx.x] = s_cnt[0][d] + s_cnt[1][d] + s_cnt[2][d] + s_cnt[3][d];
	}
	const uint32_t ne = s_n;
	if (ne == 0) return;
	if (ne > CV3_STAGE) { if (threadIdx.x == 0) st->overflow = 1; return; }
	const uint32_t li = blockIdx.x & (CV2_LISTS - 1);
	if (threadIdx.x == 0) s_base = atomicAdd(&st->list_n[li], ne);
	__syncthreads();
	const uint32_t b0 = s_base;
	if (b0 + ne > list_cap) { if (threadIdx.x == 0) st->overflow = 1; return; }
	for (uint32_t k = threadIdx.x; k < ne; k += MSX_BLOCK) {
		items[list_base + (int64_t)li * list_cap + b0 + k] = s_extra[k];
		sups[list_base + (int64_t)li * list_cap + b0 + k] = s_esup[k];
	}
}

__global__ __launch_bounds__(MSX_BLOCK) void k_cov_fill_lists3(uint32_t *__restrict__ items, uint8_t *__restrict__ sups, int64_t list_base,
                                                               uint32_t list_cap, const cv2_state *__restrict__ st) {
	const uint32_t li = blockIdx.y;
	const uint32_t used = st->list_n[li] < list_cap ? st->list_n[li] : list_cap;
	for (uint32_t k = used + blockIdx.x * MSX_BLOCK + threadIdx.x; k < list_cap; k += gridDim.x * MSX_BLOCK) {
		items[list_base + (int64_t)li * list_cap + k] = 0xffffffffu;
		sups[list_base + (int64_t)li * list_cap + k] = 0xffu;
	}
}

// lay[0 .. 256): where each top byte's bucket begins in the sorted items, lay[256 .. 512): how many items it holds
// start[2 t], start[2 t + 1]: the sorted items of tile t
__global__ __launch_bounds__(MSX_BLOCK) void k_cov_starts3(const uint32_t *__restrict__ items, const uint32_t *__restrict__ lay, int64_t n_tiles,
                                                           uint32_t *__restrict__ start) {
	const int64_t q = (int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x;
	if (q >= 2 * n_tiles) return;
	const uint32_t t = (uint32_t)(q >> 1), sup = t >> 8, want = (t & 255u) + (uint32_t)(q & 1);
	// (up to the bucket's last whole sort tile: the passes are not stable, so the empty slots that fill it -- all ones, digit
	//  255 -- lie anywhere among the items of the bucket's tile 255; k_cov_depths3 skips them)
	uint32_t lo = lay[sup], hi = lo + ((lay[256 + sup] + MSX_SORT_TILE - 1u) & ~(uint32_t)(MSX_SORT_TILE - 1u));
	while (lo < hi) {
		const uint32_t mid = lo + ((hi - lo) >> 1);
		if ((items[mid] >> 24) < want) lo = mid + 1; else hi = mid;
	}
	start[q] = lo;
}

// A tile with more items than a workgroup should walk alone is pre-reduced by several: its items in UNITS of CV3_UNIT, each
// unit one workgroup's share (k_cov_heavy_add3).  (Round 4's first cut listed the 8 K-item chunks of the sorted array that
// touch such a tile and flushed an image per chunk: 4096 atomics per 8192 items; units of 32 K flush a quarter of that.)
#define CV3_UNIT 32768u
__global__ __launch_bounds__(MSX_BLOCK) void k_cov_heavy_list3(const uint32_t *__restrict__ start, int64_t n_tiles, int32_t *__restrict__ slot_of,
                                                               cv2_state *__restrict__ st, uint32_t heavy_from, uint4 *__restrict__ units,
                                                               uint32_t unit_cap) {
	const int64_t t = (int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x;
	if (t >= n_tiles) return;
	int32_t slot = -1;
	const uint32_t lo = start[2 * t], hi = start[2 * t + 1];
	if (hi - lo > heavy_from) {
		const uint32_t k = atomicAdd(&st->n_heavy, 1u);
		const uint32_t nu = (hi - lo + CV3_UNIT - 1u) / CV3_UNIT;
		const uint32_t ub = atomicAdd(&st->n_hchunks, nu);
		if (k < CV2_HEAVY_CAP && ub + nu <= unit_cap) {
			slot = (int32_t)k;
			for (uint32_t u = 0; u < nu; u++) {
				const uint32_t a = lo + u * CV3_UNIT;
				units[ub + u] = make_uint4((uint32_t)k, a, hi - a < CV3_UNIT ? hi : a + CV3_UNIT, 0u);
			}
		} else {
			// no image slot (or no room in the unit list) for this tile: its workgroup of k_cov_depths3 walks the items alone --
			// slower, as exact.  The unit numbers drawn above must not stay unwritten: k_cov_heavy_add3 walks the list up to
			// n_hchunks, so they become units without items.
			st->overflow = 1;
			for (uint32_t u = 0; u < nu && ub + u < unit_cap; u++) units[ub + u] = make_uint4(0u, 0u, 0u, 0u);
		}
	}
	slot_of[t] = slot;
}

__global__ __launch_bounds__(MSX_BLOCK) void k_cov_heavy_zero3(int32_t *__restrict__ side, const cv2_state *__restrict__ st) {
	if (blockIdx.x >= st->n_heavy || blockIdx.x >= CV2_HEAVY_CAP) return;
	int4 *p = reinterpret_cast<int4 *>(side + (size_t)blockIdx.x * CV3_TILE);
	for (uint32_t q = threadIdx.x; q < CV3_TILE / 4; q += MSX_BLOCK) p[q] = make_int4(0, 0, 0, 0);
}

// an item's two marks in a tile's image: +1 where the piece begins, -1 behind its last cell if that is inside the tile
// (all ones: an empty slot -- no piece reaches beyond its tile, so no piece reads like that)
__device__ __forceinline__ void cv3_mark(int32_t *s_d, uint32_t v) {
	if (v == 0xffffffffu) return;
	const uint32_t c = (v >> 12) & (CV3_TILE - 1u), e = c + (v & 4095u) + 1u;
	atomicAdd(&s_d[c], 1);
	if (e < CV3_TILE) atomicAdd(&s_d[e], -1);
}

__global__ __launch_bounds__(MSX_BLOCK) void k_cov_heavy_add3(const uint32_t *__restrict__ items, int32_t *__restrict__ side,
                                                              const cv2_state *__restrict__ st, const uint4 *__restrict__ units,
                                                              uint32_t unit_cap) {
	__shared__ int32_t s_d[CV3_TILE];
	const uint32_t n_units = st->n_hchunks < unit_cap ? st->n_hchunks : unit_cap;
	for (uint32_t ui = blockIdx.x; ui < n_units; ui += gridDim.x) {
		const uint4 u = units[ui];                               // (image slot, first item, behind the last item)
		for (uint32_t q = threadIdx.x; q < CV3_TILE; q += MSX_BLOCK) s_d[q] = 0;
		__syncthreads();
		{
			// eight items per thread in flight, then their marks (one after the other the loads were the whole of the time)
			uint32_t q = u.y + threadIdx.x;
			for (; q + 7u * MSX_BLOCK < u.z; q += 8u * MSX_BLOCK) {
				uint32_t v[8];
#pragma unroll
				for (int j = 0; j < 8; j++) v[j] = items[q + (uint32_t)j * MSX_BLOCK];
#pragma unroll
				for (int j = 0; j < 8; j++) cv3_mark(s_d, v[j]);
			}
			for (; q < u.z; q += MSX_BLOCK) cv3_mark(s_d, items[q]);
		}
		__syncthreads();
		int32_t *img = side + (size_t)u.x * CV3_TILE;
		for (uint32_t q = threadIdx.x; q < CV3_TILE; q += MSX_BLOCK) {
			const int32_t d = s_d[q];
			if (__ballot(d != 0) != 0ull) atomicAdd(&img[q], d);
		}
		__syncthreads();
	}
}

// one workgroup per tile: the pieces' marks as an LDS image (or the pre-reduced image), its inclusive sum, 16-byte stores --
// non-temporal ones (nt): the gigabyte of depths is not read again here and need not pass through the L2s' write-back
// (c4: 290 -> 190 us for this kernel; the same hint on the emit kernel's and the radix passes' stores changed nothing:
// what they write is read by the next kernel)
__global__ __launch_bounds__(MSX_BLOCK) void k_cov_depths3(const uint32_t *__restrict__ items, const uint32_t *__restrict__ start,
                                                           const int32_t *__restrict__ slot_of, const int32_t *__restrict__ side,
                                                           int64_t total_cells, int32_t *__restrict__ cov, int nt, int acc) {
	// acc: cov[] already holds depths (of the batches msx_coverage_collect had to pile up the streamed way) -- added to
	__shared__ int32_t s_d[CV3_TILE];
	__shared__ int32_t s_w[MSX_BLOCK / 64];
	const int64_t t = blockIdx.x;
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const uint32_t ps = start[2 * t], pe = start[2 * t + 1];
	const int32_t slot = slot_of[t];
	if (slot >= 0) {
		const int4 *img = reinterpret_cast<const int4 *>(side + (size_t)slot * CV3_TILE);
		for (uint32_t q = threadIdx.x; q < CV3_TILE / 4; q += MSX_BLOCK) reinterpret_cast<int4 *>(s_d)[q] = img[q];
	} else {
		for (uint32_t q = threadIdx.x; q < CV3_TILE / 4; q += MSX_BLOCK) reinterpret_cast<int4 *>(s_d)[q] = make_int4(0, 0, 0, 0);
		__syncthreads();
		uint32_t q = ps + threadIdx.x;
		for (; q + 3u * MSX_BLOCK < pe; q += 4u * MSX_BLOCK) {      // (four loads in flight)
			uint32_t v[4];
#pragma unroll
			for (int j = 0; j < 4; j++) v[j] = items[q + (uint32_t)j * MSX_BLOCK];
#pragma unroll
			for (int j = 0; j < 4; j++) cv3_mark(s_d, v[j]);
		}
		for (; q < pe; q += MSX_BLOCK) cv3_mark(s_d, items[q]);
	}
	__syncthreads();
	const int64_t c0 = t << CV3_TILE_SHIFT;
	int32_t running = 0;
	for (uint32_t base = 0; base < CV3_TILE; base += MSX_BLOCK * 4) {
		const uint32_t q = base + threadIdx.x * 4u;
		int4 v = *reinterpret_cast<const int4 *>(&s_d[q]);
		v.y += v.x; v.z += v.y; v.w += v.z;
		int32_t inc = v.w;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const int32_t u = __shfl_up(inc, o, 64);
			if (lane >= o) inc += u;
		}
		if (lane == 63) s_w[w] = inc;
		__syncthreads();
		int32_t woff = 0, tot = 0;
		for (int k = 0; k < MSX_BLOCK / 64; k++) { if (k < w) woff += s_w[k]; tot += s_w[k]; }
		const int32_t add = running + woff + inc - v.w;
		v.x += add; v.y += add; v.z += add; v.w += add;
		const int64_t c = c0 + q;
		if (acc) {
			if (c <= total_cells) v.x += cov[c];
			if (c + 1 <= total_cells) v.y += cov[c + 1];
			if (c + 2 <= total_cells) v.z += cov[c + 2];
			if (c + 3 <= total_cells) v.w += cov[c + 3];
		}
		if (c + 4 <= total_cells + 1) {
			if (nt) {
				typedef int v4i __attribute__((ext_vector_type(4)));
				v4i x = {v.x, v.y, v.z, v.w};
				__builtin_nontemporal_store(x, reinterpret_cast<v4i *>(&cov[c]));
			}
			else *reinterpret_cast<int4 *>(&cov[c]) = v;
		}
		else {
			if (c <= total_cells) cov[c] = v.x;
			if (c + 1 <= total_cells) cov[c + 1] = v.y;
			if (c + 2 <= total_cells) cov[c + 2] = v.z;
		}
		running += tot;
		__syncthreads();
	}
}

// From the items of a sample (own slots and overflow lists, empty slots all ones / 0xff) to its depths: the two passes, where
// every tile's items lie, the pre-reduction of the hot tiles, the depth kernel.  items: msx_sort_k32v8_bound(n_items) + 64
// words (the sorted items end up there), items1 as many; counted: sort tiles whose first-pass digit counts the emit kernel
// left in cv_hist (stride = the sort tiles of n_items).  acc: cov[] holds depths already, add to them.  *heavy_overflow: a
// hot tile had to be walked by one workgroup (no image slot left) -- the depths are right all the same.
static int cov_pieces_finish(msx_ctx *ctx, uint32_t *items, const uint8_t *sups, uint32_t *items1, int64_t n_items, int64_t counted,
                             int64_t total_len, int32_t *cov, int acc, int *heavy_overflow) {
	int rc;
	const int64_t n_tiles = (total_len + 1 + CV3_TILE - 1) >> CV3_TILE_SHIFT;
	const int64_t n_ub = msx_sort_k32v8_bound(n_items);
	const uint32_t unit_cap = (uint32_t)(n_ub / CV3_UNIT + CV2_HEAVY_CAP + 64);      // (a tile's last unit may be short)
	if ((rc = msx_reserve(ctx, &ctx->cv_start, (size_t)(2 * n_tiles + n_tiles + 512 + 64) * 4 + sizeof(cv2_state) + 64 + (size_t)(unit_cap + 8) * 16 + 64)))
		return rc;
	if ((rc = msx_reserve(ctx, &ctx->cv_side, (size_t)CV2_HEAVY_CAP * CV_TILE * 4))) return rc;
	uint32_t *start = (uint32_t *)ctx->cv_start.p;
	int32_t *slot_of = (int32_t *)(start + 2 * n_tiles);
	uint32_t *lay = (uint32_t *)(slot_of + n_tiles);
	cv2_state *st = (cv2_state *)(((uintptr_t)(lay + 512) + 63) & ~(uintptr_t)63);
	uint4 *units = (uint4 *)(((uintptr_t)(st + 1) + 15) & ~(uintptr_t)15);
	MSX_HIP(ctx, hipMemsetAsync(st, 0, sizeof(cv2_state), ctx->stream));
	if ((rc = msx_sort_k32v8(ctx, items, sups, items1, n_items, 24, &ctx->cv_hist, &ctx->cv_off, counted, 255, lay))) return rc;
	const uint32_t heavy_from = CV2_HEAVY;
	hipLaunchKernelGGL(k_cov_starts3, dim3((unsigned)((2 * n_tiles + MSX_BLOCK - 1) / MSX_BLOCK)), dim3(MSX_BLOCK), 0, ctx->stream,
	                   (const uint32_t *)items, (const uint32_t *)lay, n_tiles, start);
	hipLaunchKernelGGL(k_cov_heavy_list3, dim3((unsigned)((n_tiles + MSX_BLOCK - 1) / MSX_BLOCK)), dim3(MSX_BLOCK), 0, ctx->stream,
	                   (const uint32_t *)start, n_tiles, slot_of, st, heavy_from, units, unit_cap);
	hipLaunchKernelGGL(k_cov_heavy_zero3, dim3(CV2_HEAVY_CAP), dim3(MSX_BLOCK), 0, ctx->stream, (int32_t *)ctx->cv_side.p, (const cv2_state *)st);
	hipLaunchKernelGGL(k_cov_heavy_add3, dim3(unit_cap < 2048u ? unit_cap : 2048u), dim3(MSX_BLOCK), 0, ctx->stream, (const uint32_t *)items,
	                   (int32_t *)ctx->cv_side.p, (const cv2_state *)st, (const uint4 *)units, unit_cap);
	hipLaunchKernelGGL(k_cov_depths3, dim3((unsigned)n_tiles), dim3(MSX_BLOCK), 0, ctx->stream, (const uint32_t *)items, (const uint32_t *)start,
	                   (const int32_t *)slot_of, (const int32_t *)ctx->cv_side.p, total_len, cov, acc ? 0 : 1, acc);
	MSX_HIP(ctx, hipGetLastError());
	if (heavy_overflow) {
		cv2_state h;
		MSX_HIP(ctx, hipMemcpyAsync(&h, st, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
		MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		*heavy_overflow = (int)h.overflow;
	}
	return MSX_OK;
}

// the items of n records into items[0 ..) / sups[0 ..): own slots (whole sort tiles), then CV2_LISTS overflow lists of
// list_cap; *n_items_out what that comes to.  hist (or null) / hist_tiles: the first pass's digit counts left for the sort.
static inline void cov_pieces_geometry(int64_t n, int64_t *n_wg, int64_t *own, uint32_t *list_cap, int64_t *n_items) {
	*n_wg = (n + CV3_REC - 1) / CV3_REC;
	*own = *n_wg * MSX_SORT_TILE;
	*list_cap = (uint32_t)(((n / 8 + CV2_LISTS - 1) / CV2_LISTS + 1024) & ~(int64_t)1023);     // (a kilobyte of items at least)
	*n_items = *own + (int64_t)CV2_LISTS * *list_cap;
}
static int cov_pieces_emit(msx_ctx *ctx, const msx_batch *b, const int64_t *cov_off, int32_t n_targets, uint8_t *covered, uint32_t *items,
                           uint8_t *sups, cv2_state *st, uint32_t *hist, int64_t hist_tiles, bool targets_ready) {
	int rc;
	const int64_t n = b->n_records;
	int64_t n_wg, own, n_items;
	uint32_t list_cap;
	cov_pieces_geometry(n, &n_wg, &own, &list_cap, &n_items);
	if ((rc = msx_reserve(ctx, &ctx->cv_targets, (size_t)n_targets * 8 + 64))) return rc;
	uint2 *targets = (uint2 *)ctx->cv_targets.p;
	MSX_HIP(ctx, hipMemsetAsync(st, 0, sizeof(cv2_state), ctx->stream));
	// (the last workgroup's tile need not be full: the emit kernel fills the slots behind the records with empty keys)
	if (!targets_ready)
		hipLaunchKernelGGL(k_cov_targets3, dim3((unsigned)((n_targets + MSX_BLOCK - 1) / MSX_BLOCK)), dim3(MSX_BLOCK), 0, ctx->stream, cov_off,
		                   n_targets, targets);
	hipLaunchKernelGGL(k_cov_emit3, dim3((unsigned)n_wg), dim3(MSX_BLOCK), 0, ctx->stream, n, b->tid, b->pos, b->cigar_off, b->cigar,
	                   (const uint2 *)targets, covered, items, sups, own, list_cap, st, hist, hist_tiles);
	hipLaunchKernelGGL(k_cov_fill_lists3, dim3(16, CV2_LISTS), dim3(MSX_BLOCK), 0, ctx->stream, items, sups, own, list_cap, st);
	MSX_HIP(ctx, hipGetLastError());
	return MSX_OK;
}

// returns MSX_OK with *done = 1 when the depths are written, *done = 0 when this form does not apply (or its lists ran full)
static int cov_depths_pieces(msx_ctx *ctx, const msx_batch *b, const int64_t *cov_off, int32_t n_targets, int64_t total_len, int32_t *cov,
                             uint8_t *covered, int *done) {
	*done = 0;
	const int64_t n = b->n_records;
	const int64_t n_tiles = (total_len + 1 + CV3_TILE - 1) >> CV3_TILE_SHIFT;
	if (n < (1 << 16) || n_targets <= 0 || n_tiles > CV3_MAX_TILES || n >= ((int64_t)1 << 31) - (1 << 24) || getenv("MSX_COV_MARKS") ||
	    getenv("MSX_COV_STREAMED"))
		return MSX_OK;
	int rc;
	int64_t n_wg, own, n_items;
	uint32_t list_cap;
	cov_pieces_geometry(n, &n_wg, &own, &list_cap, &n_items);
	const int64_t n_ub = msx_sort_k32v8_bound(n_items);       // (every bucket of the sorted items begins at a whole sort tile)
	// cv_key[0]: the items and their top bytes, later the sorted items; cv_key[1]: the items between the passes
	if ((rc = msx_reserve(ctx, &ctx->cv_key[0], (size_t)(n_ub + 64) * 4 + (size_t)n_items + 64 + sizeof(cv2_state) + 64))) return rc;
	if ((rc = msx_reserve(ctx, &ctx->cv_key[1], (size_t)(n_ub + 64) * 4))) return rc;
	int64_t sort_tiles = 0;
	if ((rc = msx_sort_k32v8_reserve(ctx, n_items, &ctx->cv_hist, &ctx->cv_off, &sort_tiles))) return rc;
	uint32_t *items = (uint32_t *)ctx->cv_key[0].p, *items1 = (uint32_t *)ctx->cv_key[1].p;
	uint8_t *sups = (uint8_t *)(items + n_ub + 64);
	cv2_state *st = (cv2_state *)(((uintptr_t)(sups + n_items + 64) + 63) & ~(uintptr_t)63);      // (the emit kernel's: list fill, overflow)
	msx_time_begin(ctx, MSX_K_COVERAGE);
	if ((rc = cov_pieces_emit(ctx, b, cov_off, n_targets, covered, items, sups, st, (uint32_t *)ctx->cv_hist.p, sort_tiles, false))) return rc;
	const int64_t counted = n_wg;                              // (every workgroup left the digit counts of its whole tile)
	// Did the pieces fit?  A batch whose lists ran full is piled up again the other ways -- and must not go on here: what the
	// workgroups that found no room left unwritten is whatever the buffers held before (the passes take any keys, but nothing
	// behind them is meant to read a stale word as an item).  The copy waits for the emit kernel; the finish is enqueued after it.
	if (!ctx->cvc_flag) MSX_HIP(ctx, hipHostMalloc((void **)&ctx->cvc_flag, 64, hipHostMallocDefault));
	MSX_HIP(ctx, hipMemcpyAsync(ctx->cvc_flag, &st->overflow, 4, hipMemcpyDeviceToHost, ctx->stream));
	MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (*ctx->cvc_flag) { msx_time_end(ctx); return MSX_OK; }
	int heavy = 0;
	rc = cov_pieces_finish(ctx, items, sups, items1, n_items, counted, total_len, cov, 0, &heavy);
	msx_time_end(ctx);
	if (rc) return rc;
	*done = !heavy;
	return MSX_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The same for a sample that arrives batch after batch (the command line; round 5): msx_coverage_collect puts a batch's
// pieces behind those of the batches before it -- 5 bytes per record stay on the device, a 50 M-read sample keeps 300 MB --
// and msx_coverage_collect_finish sorts and sums them once, writing the depth array once: what msx_coverage_depths does
// for one batch, without the batch having to be the sample.  What this form does not take goes the streamed way inside
// the same two calls: a sample of more than 255 x 2^20 cells (every batch), a batch whose overflow lists ran full or
// that would take the items beyond 2^31 (that batch: marks in cov[], zeroed then; the finish sums them and ADDS the tiles'
// depths to them).
// ---------------------------------------------------------------------------------------------------------------------
static int cov_grow_keep(msx_ctx *ctx, msx_buf *b, size_t keep_bytes, size_t want_bytes) {
	if (want_bytes <= b->cap && b->p) return MSX_OK;
	size_t cap = 2 * want_bytes + 4096;                  // (doubling: a sample's batches arrive without its size being known)
	void *np = nullptr;
	hipError_t e = hipMalloc(&np, cap);
	if (e != hipSuccess) return msx_fail(ctx, MSX_ERR_NOMEM, "hipMalloc(%zu) failed: %s", cap, hipGetErrorString(e));
	if (msx_poison_on()) MSX_HIP(ctx, hipMemsetAsync(np, 0xa5, cap, ctx->stream));        // (tests: msx_reserve)
	if (b->p) {
		if (keep_bytes) MSX_HIP(ctx, hipMemcpyAsync(np, b->p, keep_bytes, hipMemcpyDeviceToDevice, ctx->stream));
		MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		(void)hipFree(b->p);
	}
	b->p = np;
	b->cap = cap;
	return MSX_OK;
}

extern "C" int msx_coverage_collect(msx_ctx *ctx, const msx_batch *b, const int64_t *cov_off, int32_t n_targets, int64_t total_len,
                                    int32_t *cov, uint8_t *covered) {
	if (!ctx || !b || !cov_off || !cov || total_len < 0) return MSX_ERR_ARG;
	if (!b->pos || !b->tid || !b->cigar_off || !b->cigar)
		return msx_fail(ctx, MSX_ERR_ARG, "msx_coverage_collect needs tid, pos and cigar arrays");
	msx_join(ctx);
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	msx_cov_collect &C = ctx->cvc;
	int rc;
	if (!C.active) {
		const int64_t n_tiles = (total_len + 1 + CV3_TILE - 1) >> CV3_TILE_SHIFT;
		C = msx_cov_collect();
		C.active = true;
		C.total_len = total_len;
		C.n_targets = n_targets;
		C.cov = cov;
		C.pieces = n_targets > 0 && n_tiles <= CV3_MAX_TILES && !getenv("MSX_COV_MARKS") && !getenv("MSX_COV_STREAMED");
	} else if (C.total_len != total_len || C.n_targets != n_targets || C.cov != cov) {
		return msx_fail(ctx, MSX_ERR_ARG, "msx_coverage_collect: another depth array than the one the sample began with");
	}
	const int64_t n = b->n_records;
	if (n <= 0) return MSX_OK;
	bool streamed = !C.pieces;
	if (!streamed) {
		int64_t n_wg, own, n_items;
		uint32_t list_cap;
		cov_pieces_geometry(n, &n_wg, &own, &list_cap, &n_items);
		if (C.n_items + n_items >= ((int64_t)1 << 31) - (1 << 24)) streamed = true;       // (the sort's positions are 32-bit words)
		else {
			const size_t at = (size_t)C.n_items, need = at + (size_t)n_items;
			if ((rc = cov_grow_keep(ctx, &ctx->cvc_items, at * 4, need * 4 + 256))) return rc;
			if ((rc = cov_grow_keep(ctx, &ctx->cvc_sups, at, need + 256))) return rc;
			if ((rc = msx_reserve(ctx, &ctx->cv_side, (size_t)CV2_HEAVY_CAP * CV_TILE * 4))) return rc;
			cv2_state *st = (cv2_state *)ctx->cv_side.p;            // (the emit kernel's state: the side images are the finish's)
			msx_time_begin(ctx, MSX_K_COVERAGE);
			// (the targets' table is written again for every batch -- n_targets words, microseconds -- rather than trusted to have
			//  survived whatever else the context was asked for in between: msx_coverage_depths uses the same buffer)
			if ((rc = cov_pieces_emit(ctx, b, cov_off, n_targets, covered, (uint32_t *)ctx->cvc_items.p + at, (uint8_t *)ctx->cvc_sups.p + at, st,
			                          nullptr, 0, false)))
				return rc;
			msx_time_end(ctx);
			// the one thing the host must know before the batch's arrays may be given back: did its pieces fit (a page-locked word)
			if (!ctx->cvc_flag) MSX_HIP(ctx, hipHostMalloc((void **)&ctx->cvc_flag, 64, hipHostMallocDefault));
			MSX_HIP(ctx, hipMemcpyAsync(ctx->cvc_flag, &st->overflow, 4, hipMemcpyDeviceToHost, ctx->stream));
			MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
			if (getenv("MSX_COV_DEBUG")) {
				cv2_state h;
				MSX_HIP(ctx, hipMemcpy(&h, st, sizeof h, hipMemcpyDeviceToHost));
				uint32_t mx = 0;
				for (int q = 0; q < CV2_LISTS; q++) mx = h.list_n[q] > mx ? h.list_n[q] : mx;
				fprintf(stderr, "# collect: batch of %lld records at item %lld (+%lld), lists of %u (fullest %u), overflow %u\n", (long long)n,
				        (long long)C.n_items, (long long)n_items, list_cap, mx, h.overflow);
			}
			if (!*ctx->cvc_flag) { C.n_items += n_items; C.n_batches++; return MSX_OK; }
			streamed = true;                                        // (its items are not kept: the next batch writes over them)
		}
	}
	if (!C.cov_zeroed) {
		MSX_HIP(ctx, hipMemsetAsync(cov, 0, (size_t)(total_len + 1) * 4, ctx->stream));
		C.cov_zeroed = true;
	}
	C.n_streamed++;
	return msx_coverage_accumulate(ctx, b, cov_off, n_targets, total_len, cov, covered);
}

extern "C" int msx_coverage_collect_finish(msx_ctx *ctx, int32_t *cov, int64_t total_len, int64_t *n_batches_streamed) {
	if (!ctx || !cov || total_len < 0) return MSX_ERR_ARG;
	msx_join(ctx);
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	msx_cov_collect &C = ctx->cvc;
	if (C.active && (C.cov != cov || C.total_len != total_len))
		return msx_fail(ctx, MSX_ERR_ARG, "msx_coverage_collect_finish: another depth array than the one the sample began with");
	int rc = MSX_OK;
	const bool marks = C.active && C.cov_zeroed;
	if (n_batches_streamed) *n_batches_streamed = C.active ? C.n_streamed : 0;
	if (marks) rc = msx_coverage_finish(ctx, cov, total_len);               // the streamed batches' marks -> depths
	if (!rc && C.active && C.n_items > 0) {
		const int64_t n_ub = msx_sort_k32v8_bound(C.n_items);
		if (!(rc = cov_grow_keep(ctx, &ctx->cvc_items, (size_t)C.n_items * 4, (size_t)(n_ub + 64) * 4)) &&
		    !(rc = msx_reserve(ctx, &ctx->cv_key[1], (size_t)(n_ub + 64) * 4)) &&
		    !(rc = msx_sort_k32v8_reserve(ctx, C.n_items, &ctx->cv_hist, &ctx->cv_off, nullptr))) {
			msx_time_begin(ctx, MSX_K_COVERAGE);
			rc = cov_pieces_finish(ctx, (uint32_t *)ctx->cvc_items.p, (const uint8_t *)ctx->cvc_sups.p, (uint32_t *)ctx->cv_key[1].p, C.n_items, 0,
			                       total_len, cov, marks ? 1 : 0, nullptr);
			msx_time_end(ctx);
		}
	} else if (!rc && !marks && total_len >= 0) {
		MSX_HIP(ctx, hipMemsetAsync(cov, 0, (size_t)(total_len + 1) * 4, ctx->stream));       // a sample without a record
	}
	C = msx_cov_collect();             // (the items' buffers stay with the context: the next sample writes over them)
	return rc;
}

extern "C" int msx_coverage_depths(msx_ctx *ctx, const msx_batch *b, const int64_t *cov_off, int32_t n_targets, int64_t total_len,
                                   int32_t *cov, uint8_t *covered) {
	if (!ctx || !b || !cov_off || !cov || total_len < 0) return MSX_ERR_ARG;
	if (!b->pos || !b->tid || !b->cigar_off || !b->cigar)
		return msx_fail(ctx, MSX_ERR_ARG, "msx_coverage_depths needs tid, pos and cigar arrays");
	msx_join(ctx);
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	int rc;
	const int64_t n = b->n_records;
	const int64_t n_tiles = (total_len + 1 + CV_TILE - 1) >> CV_TILE_SHIFT;
	int tile_bits = 1;
	while (((int64_t)1 << tile_bits) < n_tiles + 1) tile_bits++;
	const int sign_shift = CV_TILE_SHIFT + tile_bits;          // item = sign << sign_shift | cell; an empty slot: all ones
	// what this path is for: a batch that is large against the depth array, items of one word.  Everything else takes
	// the streamed path (zero, accumulate, finish) -- as does a batch whose overflow lists run full.
	{
		int done = 0;
		if ((rc = cov_depths_pieces(ctx, b, cov_off, n_targets, total_len, cov, covered, &done))) return rc;
		if (done) return MSX_OK;
	}
	bool fused = n >= (1 << 16) && n_targets > 0 && sign_shift <= 30 && 2 * n < ((int64_t)1 << 31) && !getenv("MSX_COV_STREAMED");
	if (fused) {
		const int64_t n_wg = (n + CV2_REC - 1) / CV2_REC;
		const int64_t own = n_wg * MSX_SORT_TILE;                  // the records' own slots, rounded up to whole sort tiles
		const uint32_t list_cap = (uint32_t)(((n / 8 + CV2_LISTS - 1) / CV2_LISTS + 1023) & ~(int64_t)1023);
		const int64_t n_items = own + (int64_t)CV2_LISTS * list_cap;
		for (int q = 0; q < 2; q++)
			if ((rc = msx_reserve(ctx, &ctx->cv_key[q], (size_t)(n_items + 64) * 4))) return rc;
		const int64_t n_chunks = (n_items + CV_CHUNK - 1) / CV_CHUNK;
		if ((rc = msx_reserve(ctx, &ctx->cv_start, (size_t)(2 * (n_tiles + 1) + n_tiles + 64) * 4 + sizeof(cv2_state) + 64 +
		                                                (size_t)(n_chunks / 32 + 2 + n_chunks + 64) * 4)))
			return rc;
		if ((rc = msx_reserve(ctx, &ctx->cv_side, (size_t)CV2_HEAVY_CAP * CV_TILE * 4))) return rc;
		int64_t sort_tiles = 0;
		if ((rc = msx_sort_keys32_reserve(ctx, n_items, &ctx->cv_hist, &ctx->cv_off, &sort_tiles))) return rc;
		uint32_t *start = (uint32_t *)ctx->cv_start.p;
		int32_t *slot_of = (int32_t *)(start + 2 * (n_tiles + 1));
		cv2_state *st = (cv2_state *)(((uintptr_t)(slot_of + n_tiles) + 63) & ~(uintptr_t)63);
		uint32_t *items = (uint32_t *)ctx->cv_key[0].p;
		uint32_t *chunk_list = (uint32_t *)(st + 1);
		MSX_HIP(ctx, hipMemsetAsync(st, 0, sizeof(cv2_state), ctx->stream));
		msx_time_begin(ctx, MSX_K_COVERAGE);
		if (own > 2 * n)             // (the last workgroup's tile is not full: its tail holds empty slots)
			MSX_HIP(ctx, hipMemsetAsync(items + 2 * n, 0xff, (size_t)(own - 2 * n) * 4, ctx->stream));
		hipLaunchKernelGGL(k_cov_emit2, dim3((unsigned)n_wg), dim3(MSX_BLOCK), 0, ctx->stream, n, b->tid, b->pos, b->cigar_off, b->cigar,
		                   cov_off, covered, items, sign_shift, own, list_cap, st, (uint32_t *)ctx->cv_hist.p, sort_tiles, CV_TILE_SHIFT);
		hipLaunchKernelGGL(k_cov_fill_lists, dim3(16, CV2_LISTS), dim3(MSX_BLOCK), 0, ctx->stream, items, own, list_cap, st);
		// (as in the pieces form: lists that ran full leave stale words behind -- keys the sort does not order on their upper bits,
		//  so that k_cov_starts2's searches return anything; found by a test that came after large batches had used the buffers)
		if (!ctx->cvc_flag) MSX_HIP(ctx, hipHostMalloc((void **)&ctx->cvc_flag, 64, hipHostMallocDefault));
		MSX_HIP(ctx, hipMemcpyAsync(ctx->cvc_flag, &st->overflow, 4, hipMemcpyDeviceToHost, ctx->stream));
		MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		fused = *ctx->cvc_flag == 0u;
		if (!fused) msx_time_end(ctx);
		if (fused) {
		int sel = 0;
		// (the last workgroup's histogram counted only the records it has: the tail's empty slots are added by ... nothing --
		//  so that tile is counted by the sort itself)
		const int64_t counted = (own > 2 * n) ? n_wg - 1 : n_wg;
		if ((rc = msx_sort_keys32(ctx, items, (uint32_t *)ctx->cv_key[1].p, n_items, CV_TILE_SHIFT, tile_bits + 1, &ctx->cv_hist, &ctx->cv_off,
		                          &sel, counted)))
			return rc;
		const uint32_t *sorted = (const uint32_t *)ctx->cv_key[sel].p;
		const uint32_t heavy_from = CV2_HEAVY;
		hipLaunchKernelGGL(k_cov_starts2, dim3((unsigned)((2 * (n_tiles + 1) + MSX_BLOCK - 1) / MSX_BLOCK)), dim3(MSX_BLOCK), 0, ctx->stream,
		                   sorted, n_items, n_tiles, sign_shift, start);
		hipLaunchKernelGGL(k_cov_heavy_list, dim3((unsigned)((n_tiles + MSX_BLOCK - 1) / MSX_BLOCK)), dim3(MSX_BLOCK), 0, ctx->stream,
		                   (const uint32_t *)start, n_tiles, slot_of, st, heavy_from);
		hipLaunchKernelGGL(k_cov_heavy_chunks, dim3((unsigned)((n_chunks + MSX_BLOCK - 1) / MSX_BLOCK)), dim3(MSX_BLOCK), 0, ctx->stream, sorted,
		                   (const uint32_t *)start, n_tiles, sign_shift, (const int32_t *)slot_of, st, chunk_list);
		hipLaunchKernelGGL(k_cov_heavy_zero, dim3(CV2_HEAVY_CAP), dim3(MSX_BLOCK), 0, ctx->stream, (int32_t *)ctx->cv_side.p, (const cv2_state *)st);
		hipLaunchKernelGGL(k_cov_heavy_add, dim3((unsigned)(n_chunks < 1024 ? n_chunks : 1024)), dim3(MSX_BLOCK), 0, ctx->stream, sorted,
		                   (const uint32_t *)start, n_tiles, sign_shift, (const int32_t *)slot_of, (int32_t *)ctx->cv_side.p, (const cv2_state *)st,
		                   (const uint32_t *)chunk_list);
		hipLaunchKernelGGL(k_cov_depths, dim3((unsigned)n_tiles), dim3(MSX_BLOCK), 0, ctx->stream, sorted, (const uint32_t *)start, n_tiles,
		                   (const int32_t *)slot_of, (const int32_t *)ctx->cv_side.p, total_len, cov);
		msx_time_end(ctx);
		MSX_HIP(ctx, hipGetLastError());
		cv2_state h;
		MSX_HIP(ctx, hipMemcpyAsync(&h, st, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
		MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		if (!h.overflow) return MSX_OK;
		}
	}
	// the streamed path on one batch
	MSX_HIP(ctx, hipMemsetAsync(cov, 0, (size_t)(total_len + 1) * 4, ctx->stream));
	if ((rc = msx_coverage_accumulate(ctx, b, cov_off, n_targets, total_len, cov, covered))) return rc;
	if ((rc = msx_coverage_finish(ctx, cov, total_len))) return rc;
	MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
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

// msx_runtime_warmup: this translation unit's code object loaded onto the device ahead of its first launch (the runtime loads a
// module when one of its kernels is first asked for: 2-10 ms each, otherwise paid by the first batches of a command)
void msx_touch_coverage(void) {
	hipFuncAttributes attr;
	(void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_coverage_pileup));
}
