// msx_coverage.hip -- device side of `msamtools coverage` (BASELINE.json
// configs[3]): per-base depth of every reference (msam_coverage.c:33-87, 106-139).
#include "msx_internal.h"

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

extern "C" int msx_coverage_accumulate(msx_ctx *ctx, const msx_batch *b, const int64_t *cov_off, int32_t n_targets,
                                       int32_t *cov, uint8_t *covered) {
	(void)n_targets;
	if (!ctx || !b || !cov_off || !cov) return MSX_ERR_ARG;
	if (!b->pos || !b->tid || !b->cigar_off || !b->cigar)
		return msx_fail(ctx, MSX_ERR_ARG, "msx_coverage_accumulate needs tid, pos and cigar arrays");
	if (b->n_records == 0) return MSX_OK;
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	msx_time_begin(ctx, MSX_K_COVERAGE);
	hipLaunchKernelGGL(k_coverage_pileup, dim3(msx_grid(ctx, b->n_records, MSX_BLOCK)), dim3(MSX_BLOCK), 0,
	                   ctx->stream, b->n_records, b->tid, b->pos, b->cigar_off, b->cigar, cov_off, cov, covered);
	msx_time_end(ctx);
	MSX_HIP(ctx, hipGetLastError());
	return MSX_OK;
}

extern "C" int msx_coverage_finish(msx_ctx *ctx, int32_t *cov, int64_t total_len) {
	if (!ctx || !cov || total_len < 0) return MSX_ERR_ARG;
	if (total_len == 0) return MSX_OK;
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	return msx_scan_inclusive_u32(ctx, (uint32_t *)cov, total_len);
}
