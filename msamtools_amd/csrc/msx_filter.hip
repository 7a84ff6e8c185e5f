// msx_filter.hip -- device side of `msamtools filter`:
//   (k_aln_stats_flat, the per-record CIGAR/MD walk with the -l/-p/-z predicates, is in msx_stats.hip)
//   k_besthit_select    per-pool best-hit / unique-best-hit selection, mate aware
//                                                 (msam_filter.c:192-263)
//   k_emit_*            the order in which the reference calls mSamWrite
//                                                 (msam_filter.c:235-244, mBamVector.c:343-348)
// All integer work, HBM-bound; no MFMA.  One wave64 lane per pool.
#include "msx_internal.h"
#include "msx_count.h"
#include "msx_md.h"
#include "msx_stats.h"

#include <climits>
#include <cstdlib>

// ---------------------------------------------------------------------------
// best-hit selection: one lane per pool.
// ---------------------------------------------------------------------------
struct SelectArgs {
	int64_t n, n_groups;
	const uint32_t *group_off;
	const uint16_t *flag;
	const uint8_t *rflags;
	const uint8_t *pool;      // null: pooled = mapped (plain --besthit, msam_filter.c:104)
	int32_t pool_is_code;     // pool bytes are MSX_PC_* codes: FLAG and the aux bits need not be fetched
	const int32_t *as;        // AS to compare (as_out after --rescore)
	int32_t rescored;         // every mapped pooled record has AS (msam_filter.c:167)
	int32_t unique_only;      // --uniqhit
	uint8_t *keep;            // [n]
	uint32_t *gcount;         // [n_groups] records written per pool
	msx_dev_status *st;
};

// One lane per pool.  Pools are short (1 + Poisson(4) hits per read), so a lane
// fetches its pool's FLAGs, pool bytes, aux bits and AS values (<= 12 records) with
// ~27 independent loads issued back to back -- one memory round trip per pool
// instead of one per record -- realigns the sub-dword windows with alignbit /
// alignbyte so that record r sits at a compile-time position, and runs both
// passes (max + tie count per mate class; keep codes) out of registers.  Longer
// pools take the per-record loop.
#define BH_WIN 12

struct BhAcc {
	int32_t b0, b1, b2;          // best AS per mate class: neither bit, READ1, READ2
	uint32_t n0, n1, n2;         // ties
	uint32_t paired;
	uint32_t noas0, noas1, noas2;   // first participating record without AS
};

__device__ __forceinline__ void bh_count(BhAcc &c, uint32_t i, uint32_t fl, bool pooled, bool has, int32_t sc) {
	if (!pooled) return;
	const uint32_t cls = fl & MSX_F_MATES;
	c.paired |= cls;                                                 // mBamPoolIsPaired :196-204
	if (cls == 0) {
		if (!has) { if (c.noas0 == 0xffffffffu) c.noas0 = i; }
		else if (sc > c.b0) { c.b0 = sc; c.n0 = 1; } else if (sc == c.b0) c.n0++;
	} else if (cls == 0x40u) {
		if (!has) { if (c.noas1 == 0xffffffffu) c.noas1 = i; }
		else if (sc > c.b1) { c.b1 = sc; c.n1 = 1; } else if (sc == c.b1) c.n1++;
	} else if (cls == 0x80u) {
		if (!has) { if (c.noas2 == 0xffffffffu) c.noas2 = i; }
		else if (sc > c.b2) { c.b2 = sc; c.n2 = 1; } else if (sc == c.b2) c.n2++;
	}
}

__device__ __forceinline__ uint8_t bh_keep(const BhAcc &c, bool w0, bool w1, bool w2, uint32_t fl, bool pooled,
                                           bool has, int32_t sc) {
	if (!pooled || !has) return 0;
	const uint32_t cls = fl & MSX_F_MATES;
	if (cls == 0 && w0 && sc == c.b0) return 1;
	if (cls == 0x40u && w1 && sc == c.b1) return 1;
	if (cls == 0x80u && w2 && sc == c.b2) return 2;
	return 0;
}

// COUNT: the fused `filter | profile` form.  The pool's winners are in registers (k1/k2
// masks) when the selection is done, so the insert accounting of msx_count.h runs right
// here instead of in a second kernel that would fetch group_off and the keep codes again.
template <bool COUNT>
__global__ __launch_bounds__(MSX_BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_besthit_select(SelectArgs A, CountArgs P) {
	__shared__ uint32_t s_c[3][MSX_BLOCK / 64];
	__shared__ int32_t s_key[COUNT ? UI_TBL : 1];
	__shared__ uint32_t s_val[COUNT ? UI_TBL : 1];
	BlockCounts bc = {0u, 0u, 0u};
	if (COUNT) count_block_begin(P, s_key, s_val);
	const int64_t stride = (int64_t)gridDim.x * MSX_BLOCK;
	int64_t g = (int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x;
	// the next pool's bounds are fetched while this one is processed (one round trip less per pool)
	uint32_t s_nx = 0, e_nx = 0;
	if (g < A.n_groups) { s_nx = A.group_off[g]; e_nx = A.group_off[g + 1]; }
	for (; g < A.n_groups; g += stride) {
		const uint32_t s = s_nx, e = e_nx;
		if (g + stride < A.n_groups) { s_nx = A.group_off[g + stride]; e_nx = A.group_off[g + stride + 1]; }
		const uint32_t len = e - s;
		BhAcc c = {INT_MIN, INT_MIN, INT_MIN, 0u, 0u, 0u, 0u, 0xffffffffu, 0xffffffffu, 0xffffffffu};
		uint32_t cnt = 0;
		if (len <= BH_WIN && (uint64_t)(s & ~3u) + 16u <= (uint64_t)A.n && (uint64_t)s + BH_WIN <= (uint64_t)A.n) {
			// ---- bulk loads (independent of each other) ----
			const bool coded = A.pool_is_code != 0;      // (kernel argument: a scalar branch)
			const uint32_t *f32p = reinterpret_cast<const uint32_t *>(A.flag) + (s >> 1);
			uint32_t fw[7] = {0u, 0u, 0u, 0u, 0u, 0u, 0u};
			const uint32_t *r32p = reinterpret_cast<const uint32_t *>(A.rflags) + (s >> 2);
			uint32_t rw[4] = {0u, 0u, 0u, 0u}, pw[4] = {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u};
			if (!coded) {
#pragma unroll
				for (int q = 0; q < 7; q++)        // only the dwords the pool reaches into
					fw[q] = (2u * ((s >> 1) + (uint32_t)q) < e) ? f32p[q] : 0u;
#pragma unroll
				for (int q = 0; q < 4; q++) rw[q] = (4u * ((s >> 2) + (uint32_t)q) < e) ? r32p[q] : 0u;
			}
			if (A.pool) {
				const uint32_t *p32p = reinterpret_cast<const uint32_t *>(A.pool) + (s >> 2);
#pragma unroll
				for (int q = 0; q < 4; q++) pw[q] = (4u * ((s >> 2) + (uint32_t)q) < e) ? p32p[q] : 0u;
			}
			int32_t sc[BH_WIN];
#pragma unroll
			for (int r = 0; r < BH_WIN; r++) sc[r] = ((uint32_t)r < len) ? A.as[s + r] : INT_MIN;
			// ---- realign so that record r is at a fixed position ----
			const uint32_t fsh = 16u * (s & 1u), bsh = s & 3u;
			uint32_t fa[6], ra[3], pa[3];
#pragma unroll
			for (int q = 0; q < 6; q++) fa[q] = __builtin_amdgcn_alignbit(fw[q + 1], fw[q], fsh);
#pragma unroll
			for (int q = 0; q < 3; q++) {
				ra[q] = __builtin_amdgcn_alignbyte(rw[q + 1], rw[q], bsh);
				pa[q] = __builtin_amdgcn_alignbyte(pw[q + 1], pw[q], bsh);
			}
			// ---- who takes part: code = mate bits | 0x100 (participates) | 0x200 (has AS) ----
			uint32_t code[BH_WIN];
			uint32_t paired = 0;
#pragma unroll
			for (int r = 0; r < BH_WIN; r++) {
				const uint32_t pb = (pa[r >> 2] >> (8 * (r & 3))) & 0xffu;
				if (coded) {
					// the stats kernel has already decided participation and looked at FLAG and the aux bits
					code[r] = ((uint32_t)r < len && (pb & MSX_PC_IN))
					              ? ((pb & MSX_F_MATES) | 0x100u | ((pb & MSX_PC_HAS_AS) ? 0x200u : 0u)) : 0u;
				} else {
					const uint32_t fl = (fa[r >> 1] >> (16 * (r & 1))) & 0xffffu;
					const uint32_t rf = (ra[r >> 2] >> (8 * (r & 3))) & 0xffu;
					const bool part = ((uint32_t)r < len) && (A.pool ? (pb != 0) : !(fl & MSX_F_UNMAP));
					const bool has = (rf & MSX_HAS_AS) || (A.rescored && !(fl & MSX_F_UNMAP));
					code[r] = part ? ((fl & MSX_F_MATES) | 0x100u | (has ? 0x200u : 0u)) : 0u;
				}
				paired |= code[r] & MSX_F_MATES;                                 // mBamPoolIsPaired :196-204
			}
			// a paired pool is judged per mate (READ1 -> A, READ2 -> B), an unpaired one as a whole (A)
			const uint32_t selA = paired ? 0x140u : 0x100u, selB = paired ? 0x180u : 0xffffu;
			int32_t bA = INT_MIN, bB = INT_MIN;
			uint32_t hasA = 0, hasB = 0, noas = 0;                                // bit r = record r
#pragma unroll
			for (int r = 0; r < BH_WIN; r++) {
				const uint32_t cl = code[r] & 0x1c0u;
				const bool inA = (cl == selA), inB = (cl == selB), has = (code[r] & 0x200u) != 0;
				hasA |= (inA && has) ? (1u << r) : 0u;
				hasB |= (inB && has) ? (1u << r) : 0u;
				noas |= ((inA || inB) && !has) ? (1u << r) : 0u;
				const int32_t sa = (inA && has) ? sc[r] : INT_MIN, sb = (inB && has) ? sc[r] : INT_MIN;
				bA = sa > bA ? sa : bA;
				bB = sb > bB ? sb : bB;
			}
			uint32_t eqA = 0, eqB = 0;
#pragma unroll
			for (int r = 0; r < BH_WIN; r++) {
				eqA |= (sc[r] == bA) ? (1u << r) : 0u;
				eqB |= (sc[r] == bB) ? (1u << r) : 0u;
			}
			eqA &= hasA;                                                          // the records holding the best score
			eqB &= hasB;
			const uint32_t nA = (uint32_t)__popc(eqA), nB = (uint32_t)__popc(eqB);
			const bool wA = nA > 0 && (!A.unique_only || nA == 1);                // :232-233
			const bool wB = nB > 0 && (!A.unique_only || nB == 1);
			// keep codes: 1 = written in the first pass (unpaired winners, READ1 winners), 2 = READ2 winners
			const uint32_t k1 = wA ? eqA : 0u, k2 = wB ? eqB : 0u;
#pragma unroll
			for (int r = 0; r < BH_WIN; r++)
				if ((uint32_t)r < len) A.keep[s + r] = (uint8_t)(((k1 >> r) & 1u) | (((k2 >> r) & 1u) << 1));
			cnt = (uint32_t)__popc(k1 | k2);
			if (COUNT) {
				PoolAcc v;
				pool_begin(P, v, s);
				pool_visit_masks(P, v, s, k1, k2);
				pool_finish(P, g, v, s_key, s_val, bc);
			}
			if (noas) {
				const uint32_t first = s + (uint32_t)__ffs((int)noas) - 1u;
				if (paired) c.noas1 = first; else c.noas0 = first;
			}
			c.paired = paired;
		} else {
			for (uint32_t i = s; i < e; ++i) {
				const uint32_t fl = A.flag[i];
				const bool pooled = A.pool ? (A.pool[i] != 0) : !(fl & MSX_F_UNMAP);
				const bool has = A.pool_is_code ? ((A.pool[i] & MSX_PC_HAS_AS) != 0)
				                                : ((A.rflags[i] & MSX_HAS_AS) || (A.rescored && !(fl & MSX_F_UNMAP)));
				bh_count(c, i, fl, pooled, has, A.as[i]);
			}
			const bool w0 = !c.paired && c.n0 > 0 && (!A.unique_only || c.n0 == 1);
			const bool w1 = c.paired && c.n1 > 0 && (!A.unique_only || c.n1 == 1);
			const bool w2 = c.paired && c.n2 > 0 && (!A.unique_only || c.n2 == 1);
			for (uint32_t i = s; i < e; ++i) {
				const uint32_t fl = A.flag[i];
				const bool pooled = A.pool ? (A.pool[i] != 0) : !(fl & MSX_F_UNMAP);
				const bool has = A.pool_is_code ? ((A.pool[i] & MSX_PC_HAS_AS) != 0)
				                                : ((A.rflags[i] & MSX_HAS_AS) || (A.rescored && !(fl & MSX_F_UNMAP)));
				const uint8_t k = bh_keep(c, w0, w1, w2, fl, pooled, has, A.as[i]);
				A.keep[i] = k;
				cnt += (k != 0);
			}
			if (COUNT) {
				// long pool: the keep codes this lane just wrote, in output order
				PoolAcc v;
				pool_begin(P, v, s);
				for (uint32_t pass = 1; pass <= 2; ++pass)
					for (uint32_t i = s; i < e; ++i)
						if (A.keep[i] == pass) pool_visit(P, v, P.tid[i]);
				pool_finish(P, g, v, s_key, s_val, bc);
			}
		}
		// msam_filter.c:219-221: a participating record without AS is fatal
		const uint32_t bad = c.paired ? (c.noas1 < c.noas2 ? c.noas1 : c.noas2) : c.noas0;
		if (bad != 0xffffffffu) atomicMin(&A.st->first_no_as, (unsigned long long)bad);
		A.gcount[g] = cnt;
	}
	if (COUNT) count_block_end(P, s_key, s_val, s_c, bc);
}

// emit order for pools: all pass-1 records of a pool, then its pass-2 records.
// The keep codes of a pool (<= 20 records) arrive as six independent aligned dword
// loads -- one memory round trip per pool instead of one per record.
__global__ __launch_bounds__(MSX_BLOCK) void k_emit_groups(int64_t n_records, int64_t n_groups,
                                                           const uint32_t *__restrict__ group_off,
                                                           const uint8_t *__restrict__ keep,
                                                           const uint32_t *__restrict__ gbase,
                                                           int32_t *__restrict__ emit_idx, msx_dev_status *st) {
	const int64_t stride = (int64_t)gridDim.x * MSX_BLOCK;
	for (int64_t g = (int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x; g < n_groups; g += stride) {
		const uint32_t s = group_off[g], e = group_off[g + 1];
		uint32_t o = gbase[g];
		const uint32_t cnt = gbase[g + 1] - o;
		if (!cnt || !emit_idx) continue;
		if (e - s <= 20u && (uint64_t)(s & ~3u) + 24u <= (uint64_t)n_records) {
			const uint32_t base = s & ~3u, sh = s - base, len = e - s;
			const uint32_t *kw = reinterpret_cast<const uint32_t *>(keep) + (base >> 2);
			uint32_t w[6];
#pragma unroll
			for (int q = 0; q < 6; q++) w[q] = (base + 4u * (uint32_t)q < e) ? kw[q] : 0u;   // only the dwords the pool reaches into
			uint32_t m1 = 0, m2 = 0;
#pragma unroll
			for (int q = 0; q < 6; q++) {
#pragma unroll
				for (int bq = 0; bq < 4; bq++) {
					const uint32_t r = (uint32_t)(q * 4 + bq) - sh;
					const uint32_t kc = (w[q] >> (8 * bq)) & 0xffu;
					if ((uint32_t)(q * 4 + bq) >= sh && r < len) {
						m1 |= (kc == 1u ? 1u : 0u) << r;
						m2 |= (kc == 2u ? 1u : 0u) << r;
					}
				}
			}
			while (m1) { const uint32_t b = (uint32_t)__ffs((int)m1) - 1u; m1 &= m1 - 1u; emit_idx[o++] = (int32_t)(s + b); }
			while (m2) { const uint32_t b = (uint32_t)__ffs((int)m2) - 1u; m2 &= m2 - 1u; emit_idx[o++] = (int32_t)(s + b); }
		} else {
			uint32_t n2 = 0;
			for (uint32_t i = s; i < e; ++i) {
				const uint8_t k = keep[i];
				if (k == 1) emit_idx[o++] = (int32_t)i;
				n2 += (k == 2);
			}
			if (n2)
				for (uint32_t i = s; i < e; ++i)
					if (keep[i] == 2) emit_idx[o++] = (int32_t)i;
		}
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) st->n_emit = gbase[n_groups];
}

// plain filter (no best-hit): output = pooled records in input order.
// count per 2048-record chunk -> scan -> ordered fill.
#define EMIT_CHUNK 2048
__global__ __launch_bounds__(MSX_BLOCK) void k_emit_count(int64_t n, const uint8_t *__restrict__ keep,
                                                          uint32_t *__restrict__ ccount) {
	__shared__ uint32_t s_w[4];
	const int64_t base = (int64_t)blockIdx.x * EMIT_CHUNK;
	uint32_t c = 0;
	for (int k = 0; k < EMIT_CHUNK / MSX_BLOCK; k++) {
		int64_t i = base + k * MSX_BLOCK + threadIdx.x;
		if (i < n) c += keep[i] != 0;
	}
	for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
	if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
	__syncthreads();
	if (threadIdx.x == 0) ccount[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ __launch_bounds__(MSX_BLOCK) void k_emit_fill(int64_t n, int64_t n_chunks, const uint8_t *__restrict__ keep,
                                                         const uint32_t *__restrict__ cbase,
                                                         int32_t *__restrict__ emit_idx, msx_dev_status *st) {
	__shared__ uint32_t s_w[4];
	const int64_t base = (int64_t)blockIdx.x * EMIT_CHUNK;
	uint32_t run = cbase[blockIdx.x];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	for (int k = 0; k < EMIT_CHUNK / MSX_BLOCK; k++) {
		int64_t i = base + k * MSX_BLOCK + threadIdx.x;
		const bool kp = (i < n) && keep[i] != 0;
		const unsigned long long bal = __ballot(kp);
		if (lane == 0) s_w[w] = (uint32_t)__popcll(bal);
		__syncthreads();
		uint32_t before = 0, tot = 0;
		for (int q = 0; q < 4; q++) {
			if (q < w) before += s_w[q];
			tot += s_w[q];
		}
		if (kp && emit_idx)
			emit_idx[run + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = (int32_t)i;
		run += tot;
		__syncthreads();
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) st->n_emit = cbase[n_chunks];
}

__global__ void k_status_init(msx_dev_status *st) {
	st->first_no_mdnm = ~0ull;
	st->first_no_as = ~0ull;
	st->n_emit = 0;
}

// ---------------------------------------------------------------------------
// host entry points
// ---------------------------------------------------------------------------
static int filter_choice(const msx_filter_params *p) {
	int c = 0;
	if (p->min_length > 0) c |= 1;   // msam_filter.c:79-81
	if (p->ppt != 0) c |= 2;
	if (p->max_clip < 100) c |= 4;
	return c;
}

// The statistics kernel: k_aln_stats_flat (msx_stats.hip), one wave per 128 records
static void launch_stats(msx_ctx *ctx, FilterArgs &A) {
	const uintptr_t al8 = (uintptr_t)A.cigar_off | (uintptr_t)A.md_off;
	A.wide_ok = ((al8 & 7u) == 0 && ((uintptr_t)A.flag & 3u) == 0 && ((uintptr_t)A.rflags & 1u) == 0 &&
	             ((uintptr_t)A.pool & 1u) == 0) ? 1 : 0;
	msx_launch_aln_stats_flat(ctx, A, msx_grid_x(ctx, A.n, 128 * (MSX_BLOCK / 64), 4));
}

static void fill_args(FilterArgs &A, const msx_batch *b) {
	A.n = b->n_records;
	A.flag = b->flag;
	A.rflags = b->rflags;
	A.cigar_off = b->cigar_off;
	A.cigar = b->cigar;
	A.md_off = b->md_off;
	A.md = b->md;
	A.nm = b->nm;
	A.as = b->as;
}

static int filter_enqueue_impl(msx_ctx *ctx, const msx_batch *b, const msx_filter_params *p, const msx_filter_out *out,
                               msx_profile *prof) {
	if (!ctx || !b || !p || !out || !out->keep) return msx_fail(ctx, MSX_ERR_ARG, "msx_filter_enqueue: null argument");
	if (prof && (!b->group_off || !b->tid))
		return msx_fail(ctx, MSX_ERR_ARG, "msx_filter_profile_enqueue needs tid and group_off");
	const int choice = filter_choice(p);
	const bool best = p->besthit || p->uniqhit;
	if (choice == 0 && !best)
		return msx_fail(ctx, MSX_ERR_NO_FILTER,
		                "'filter' command requires atleast one of --ppt, -l, -p, -z, --besthit or --uniqhit");
	if (best && !b->group_off)
		return msx_fail(ctx, MSX_ERR_ARG, "--besthit/--uniqhit need msx_batch.group_off (QNAME pools)");
	if (p->rescore && !out->as_out)
		return msx_fail(ctx, MSX_ERR_ARG, "--rescore needs msx_filter_out.as_out");
	if (b->n_records > 0x7fffffffLL) return msx_fail(ctx, MSX_ERR_ARG, "batch too large");
	if (!b->flag || !b->rflags) return msx_fail(ctx, MSX_ERR_ARG, "msx_filter_enqueue: flag/rflags missing");
	if ((choice != 0 || p->rescore) && (!b->cigar_off || !b->cigar || !b->md_off || !b->md || !b->nm))
		return msx_fail(ctx, MSX_ERR_ARG, "msx_filter_enqueue: -l/-p/-z/--rescore need cigar, md and nm arrays");
	if (best && !b->as) return msx_fail(ctx, MSX_ERR_ARG, "msx_filter_enqueue: --besthit/--uniqhit need the as array");
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	const int64_t n = b->n_records;
	hipLaunchKernelGGL(k_status_init, dim3(1), dim3(1), 0, ctx->stream, ctx->d_status);
	ctx->filter_pending = true;
	if (n == 0) return MSX_OK;

	const bool need_stats = choice != 0 || p->rescore;
	uint8_t *pool = nullptr;
	int rc;
	if (need_stats) {
		FilterArgs A = {};
		fill_args(A, b);
		A.min_length = p->min_length;
		A.ppt = p->ppt;
		A.max_clip = p->max_clip;
		A.choice = choice;
		A.rescore = p->rescore;
		A.invert = p->invert;
		A.keep_unmapped = p->keep_unmapped;
		A.as_out = p->rescore ? out->as_out : nullptr;
		A.st = ctx->d_status;
		if (best) {
			if ((rc = msx_reserve(ctx, &ctx->pool_code, (size_t)n))) return rc;
			pool = (uint8_t *)ctx->pool_code.p;
			A.pool_as_code = 1;
		} else {
			pool = out->keep;   // mWriteBamPool: keep == pooled
		}
		A.pool = pool;
		msx_time_begin(ctx, MSX_K_ALN_STATS);
		launch_stats(ctx, A);
		msx_time_end(ctx);
	}
	if (best) {
		const int64_t ng = b->n_groups;
		if ((rc = msx_reserve(ctx, &ctx->gcount, (size_t)(ng + 8) * 4))) return rc;
		if ((rc = msx_reserve(ctx, &ctx->gbase, (size_t)(ng + 8) * 4))) return rc;
		SelectArgs S = {};
		S.n = n;
		S.n_groups = ng;
		S.group_off = b->group_off;
		S.flag = b->flag;
		S.rflags = b->rflags;
		S.pool = pool;
		S.pool_is_code = (pool && need_stats) ? 1 : 0;
		S.as = p->rescore ? out->as_out : b->as;
		S.rescored = p->rescore;
		S.unique_only = p->uniqhit ? 1 : 0;   // msam_filter.c:88-91: --uniqhit wins
		S.keep = out->keep;
		S.gcount = (uint32_t *)ctx->gcount.p;
		S.st = ctx->d_status;
		CountArgs P = {};
		bool by_part = false;
		if (prof && ng > 0 && (rc = msx_profile_count_prepare(ctx, prof, b, out->keep, &P, &by_part))) return rc;
		msx_time_begin(ctx, MSX_K_BESTHIT);
		if (prof && ng > 0)
			hipLaunchKernelGGL(k_besthit_select<true>, dim3(msx_grid_x(ctx, ng, MSX_BLOCK, 4)), dim3(MSX_BLOCK), 0,
			                   ctx->stream, S, P);
		else
			hipLaunchKernelGGL(k_besthit_select<false>, dim3(msx_grid_x(ctx, ng, MSX_BLOCK, 4)), dim3(MSX_BLOCK), 0,
			                   ctx->stream, S, P);
		msx_time_end(ctx);
		// three independent chains from here: counting the unique-insert keys (side lane 0, inside
		// msx_profile_count_finish), appending the multi-mapper lists (main stream), and filter's own
		// output order (side lane 1); all three are latency-bound and overlap well
		const bool forked = prof && ng > 0 && msx_fork(ctx);
		if (prof && ng > 0 && (rc = msx_profile_count_finish(ctx, prof, b, by_part))) { msx_join(ctx); return rc; }
		if (forked) msx_lane_enter(ctx, 1);
		rc = msx_scan_u32(ctx, S.gcount, (uint32_t *)ctx->gbase.p, ng);
		if (!rc) {
			msx_time_begin(ctx, MSX_K_EMIT);
			hipLaunchKernelGGL(k_emit_groups, dim3(msx_grid(ctx, ng, MSX_BLOCK)), dim3(MSX_BLOCK), 0, ctx->stream,
			                   n, ng, b->group_off, (const uint8_t *)out->keep, (const uint32_t *)ctx->gbase.p,
			                   out->emit_idx, ctx->d_status);
			msx_time_end(ctx);
		}
		msx_join(ctx);
		if (rc) return rc;
	} else {
		const int64_t nc = (n + EMIT_CHUNK - 1) / EMIT_CHUNK;
		if ((rc = msx_reserve(ctx, &ctx->gcount, (size_t)(nc + 8) * 4))) return rc;
		if ((rc = msx_reserve(ctx, &ctx->gbase, (size_t)(nc + 8) * 4))) return rc;
		msx_time_begin(ctx, MSX_K_EMIT);
		hipLaunchKernelGGL(k_emit_count, dim3((unsigned)nc), dim3(MSX_BLOCK), 0, ctx->stream, n,
		                   (const uint8_t *)out->keep, (uint32_t *)ctx->gcount.p);
		msx_time_end(ctx);
		if ((rc = msx_scan_u32(ctx, (const uint32_t *)ctx->gcount.p, (uint32_t *)ctx->gbase.p, nc))) return rc;
		msx_time_begin(ctx, MSX_K_EMIT);
		hipLaunchKernelGGL(k_emit_fill, dim3((unsigned)nc), dim3(MSX_BLOCK), 0, ctx->stream, n, nc,
		                   (const uint8_t *)out->keep, (const uint32_t *)ctx->gbase.p, out->emit_idx,
		                   ctx->d_status);
		msx_time_end(ctx);
		// without best-hit selection there is no per-pool kernel to fuse into: plain sequence
		if (prof && (rc = msx_profile_accumulate(ctx, prof, b, out->keep))) return rc;
	}
	MSX_HIP(ctx, hipGetLastError());
	return MSX_OK;
}

extern "C" int msx_filter_enqueue(msx_ctx *ctx, const msx_batch *b, const msx_filter_params *p,
                                  const msx_filter_out *out) {
	return filter_enqueue_impl(ctx, b, p, out, nullptr);
}

extern "C" int msx_filter_profile_enqueue(msx_ctx *ctx, const msx_batch *b, const msx_filter_params *p,
                                          const msx_filter_out *out, msx_profile *prof) {
	if (!prof) return msx_fail(ctx, MSX_ERR_ARG, "msx_filter_profile_enqueue: null profile");
	return filter_enqueue_impl(ctx, b, p, out, prof);
}

extern "C" int msx_filter_finish(msx_ctx *ctx, msx_filter_status *status) {
	if (!ctx) return MSX_ERR_ARG;
	if (status) { status->n_emit = 0; status->err_record = -1; }
	if (!ctx->filter_pending) return msx_fail(ctx, MSX_ERR_ARG, "msx_filter_finish without msx_filter_enqueue");
	ctx->filter_pending = false;
	MSX_HIP(ctx, hipMemcpyAsync(ctx->h_status, ctx->d_status, sizeof(msx_dev_status), hipMemcpyDeviceToHost,
	                            ctx->stream));
	MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	const msx_dev_status s = *ctx->h_status;
	if (status) status->n_emit = (int64_t)s.n_emit;
	// the reference dies at the first offending record in stream order; stats
	// errors are raised while reading (before the pool is written)
	if (s.first_no_mdnm != ~0ull && (s.first_no_as == ~0ull || s.first_no_mdnm <= s.first_no_as)) {
		if (status) status->err_record = (int64_t)s.first_no_mdnm;
		return msx_fail(ctx, MSX_ERR_NO_MD_NM,
		                "Either NM or MD must be present in SAM/BAM input for 'filter' command. "
		                "Type 'msamtools filter -h' for details.");
	}
	if (s.first_no_as != ~0ull) {
		if (status) status->err_record = (int64_t)s.first_no_as;
		return msx_fail(ctx, MSX_ERR_NO_AS,
		                "Required field AS not found in SAM/BAM input. Type 'msamtools -h' for details.");
	}
	return MSX_OK;
}

extern "C" int msx_aln_stats(msx_ctx *ctx, const msx_batch *b, int32_t *length, int32_t *qlen, int32_t *qclip,
                             int32_t *edit, uint8_t *status) {
	if (!ctx || !b) return MSX_ERR_ARG;
	if (!length || !qlen || !qclip || !edit)
		return msx_fail(ctx, MSX_ERR_ARG, "msx_aln_stats: length/query_length/query_clip/edit are all required");
	if (!b->flag || !b->rflags || !b->cigar_off || !b->cigar || !b->md_off || !b->md || !b->nm)
		return msx_fail(ctx, MSX_ERR_ARG, "msx_aln_stats: flag, rflags, cigar, md and nm arrays are required");
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	hipLaunchKernelGGL(k_status_init, dim3(1), dim3(1), 0, ctx->stream, ctx->d_status);
	if (b->n_records == 0) return MSX_OK;
	FilterArgs A = {};
	fill_args(A, b);
	A.max_clip = 100;
	A.o_len = length;
	A.o_qlen = qlen;
	A.o_qclip = qclip;
	A.o_edit = edit;
	A.o_status = status;
	A.st = ctx->d_status;
	msx_time_begin(ctx, MSX_K_ALN_STATS);
	launch_stats(ctx, A);
	msx_time_end(ctx);
	MSX_HIP(ctx, hipGetLastError());
	MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return MSX_OK;
}
