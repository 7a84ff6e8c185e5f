// msx_profile.hip -- device side of `msamtools profile`:
//   k_insert_count   per-pool unique / multi-mapper accounting
//                    (mEstimateInsertCountOnFile/OnPool, msam_profile.c:65-243)
//   k_multi_compact  appends the pools' distinct-feature lists to the compact
//                    multi-mapper CSR (global->multi_mappers, msam_profile.c:36-39,107-121,184-186)
// (the proportional-sharing iteration itself lives in msx_prop.hip)
// Integer counters are exact (u32 atomics); the proportional iteration is
// double precision with order-free atomic accumulation (<= 1e-6 relative to the
// reference's sequential order, as BASELINE.json allows).
#include "msx_internal.h"
#include "msx_count.h"

#include <cstdlib>

__global__ __launch_bounds__(MSX_BLOCK) void k_insert_count(CountArgs A) {
	__shared__ uint32_t s_c[3][MSX_BLOCK / 64];
	__shared__ int32_t s_key[UI_TBL];
	__shared__ uint32_t s_val[UI_TBL];
	count_block_begin(A, s_key, s_val);
	BlockCounts c = {0u, 0u, 0u};
	const int64_t stride = (int64_t)gridDim.x * MSX_BLOCK;
	for (int64_t g = (int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x; g < A.n_groups; g += stride) {
		const uint32_t s = A.group_off[g], e = A.group_off[g + 1];
		if (pool_follows(A, g) || pool_follows(A, g + 1)) {      // member of a chain of pools: k_insert_chains
			if (A.pinfo) A.pinfo[g] = MSX_PINFO_NONE;
			continue;
		}
		PoolAcc v;
		pool_begin(A, v, s);
		// the stream: the batch itself, or filter's output order (all pass-1 records of the
		// pool, then its pass-2 records)
		if (!A.keep) {
			for (uint32_t i = s; i < e; ++i) pool_visit(A, v, A.tid[i]);
		} else if (e - s <= 20u && (uint64_t)(s & ~3u) + 24u <= (uint64_t)A.n_records) {
			// the pool's keep codes arrive as up to six independent aligned dword loads (one round
			// trip instead of one per record)
			const uint32_t base = s & ~3u, sh = s - base, len = e - s;
			const uint32_t *kw = reinterpret_cast<const uint32_t *>(A.keep) + (base >> 2);
			uint32_t w[6];
#pragma unroll
			for (int q = 0; q < 6; q++) w[q] = (base + 4u * (uint32_t)q < e) ? kw[q] : 0u;   // only the dwords the pool reaches into
			uint32_t m1 = 0, m2 = 0;
#pragma unroll
			for (int q = 0; q < 6; q++) {
#pragma unroll
				for (int bq = 0; bq < 4; bq++) {
					const uint32_t r = (uint32_t)(q * 4 + bq) - sh;          // record index inside the pool
					const uint32_t kc = (w[q] >> (8 * bq)) & 0xffu;
					if ((uint32_t)(q * 4 + bq) >= sh && r < len) {
						m1 |= (kc == 1u ? 1u : 0u) << r;
						m2 |= (kc == 2u ? 1u : 0u) << r;
					}
				}
			}
			pool_visit_masks(A, v, s, m1, m2);
		} else {
			for (uint32_t pass = 1; pass <= 2; ++pass)
				for (uint32_t i = s; i < e; ++i)
					if (A.keep[i] == pass) pool_visit(A, v, A.tid[i]);
		}
		pool_finish(A, g, v, s_key, s_val, c);
	}
	count_block_end(A, s_key, s_val, s_c, c);
}

// Chains of filter pools that are ONE insert for profile (CountArgs.chain_flag, msx_count.h): one lane per chain
// head walks the records its pools write -- pool by pool, first-pass records, then second-pass records: the order of
// filter's output (msam_filter.c:247-263), which is the order profile's pool holds them in (msam_profile.c:131-145
// lists distinct features by first appearance) -- and accounts for them as one pool.  Rare by construction (a record
// of another name, unmapped, between the alignments of one read), so no attempt at coalescing.
__global__ __launch_bounds__(MSX_BLOCK) void k_insert_chains(CountArgs A) {
	__shared__ uint32_t s_c[3][MSX_BLOCK / 64];
	__shared__ int32_t s_key[UI_TBL];
	__shared__ uint32_t s_val[UI_TBL];
	count_block_begin(A, s_key, s_val);
	BlockCounts c = {0u, 0u, 0u};
	const int64_t stride = (int64_t)gridDim.x * MSX_BLOCK;
	for (int64_t g = (int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x; g < A.n_groups; g += stride) {
		if (pool_follows(A, g) || !pool_follows(A, g + 1)) continue;      // not the head of a chain
		PoolAcc v;
		pool_begin(A, v, A.group_off[g]);
		for (int64_t h = g;; ++h) {
			const uint32_t s = A.group_off[h], e = A.group_off[h + 1];
			if (!A.keep) {
				for (uint32_t i = s; i < e; ++i) pool_visit(A, v, A.tid[i]);
			} else {
				for (uint32_t pass = 1; pass <= 2; ++pass)
					for (uint32_t i = s; i < e; ++i)
						if (A.keep[i] == pass) pool_visit(A, v, A.tid[i]);
			}
			if (!pool_follows(A, h + 1)) break;
		}
		pool_finish(A, g, v, s_key, s_val, c);
	}
	count_block_end(A, s_key, s_val, s_c, c);
}

void msx_profile_count_chains(msx_ctx *ctx, const CountArgs &A) {
	msx_time_begin(ctx, MSX_K_INSERT_COUNT);
	hipLaunchKernelGGL(k_insert_chains, dim3(msx_grid(ctx, A.n_groups, MSX_BLOCK)), dim3(MSX_BLOCK), 0, ctx->stream, A);
	msx_time_end(ctx);
}

// csr_tot = {n_lists, n_entries} running totals of the compact CSR.
// One workgroup per chunk of MSX_PINFO_CHUNK pools, eight consecutive pools per thread: the positions of a
// pool's list and of its entries are the chunk's base (msx_scan_pinfo_chunks) plus a scan inside the
// workgroup of (1 << 32 | nd) over the pools that hold a list.
__global__ __launch_bounds__(MSX_BLOCK) void k_multi_compact(int64_t n_groups, const uint32_t *__restrict__ group_off,
                                                             const uint32_t *__restrict__ pinfo,
                                                             const unsigned long long *__restrict__ chunk_base,
                                                             const int32_t *__restrict__ tmp_fid,
                                                             const unsigned long long *__restrict__ csr_tot,
                                                             uint32_t *__restrict__ m_off, int32_t *__restrict__ m_fid) {
	__shared__ unsigned long long s_w[MSX_BLOCK / 64];
	__shared__ unsigned long long s_pos[MSX_PINFO_CHUNK];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int64_t g0 = (int64_t)blockIdx.x * MSX_PINFO_CHUNK + (int64_t)threadIdx.x * 8;
	uint32_t info[8];
	if (g0 + 8 <= n_groups) {
		const uint4 a = *reinterpret_cast<const uint4 *>(pinfo + g0), b = *reinterpret_cast<const uint4 *>(pinfo + g0 + 4);
		info[0] = a.x; info[1] = a.y; info[2] = a.z; info[3] = a.w; info[4] = b.x; info[5] = b.y; info[6] = b.z; info[7] = b.w;
	} else {
#pragma unroll
		for (int k = 0; k < 8; k++) info[k] = g0 + k < n_groups ? pinfo[g0 + k] : MSX_PINFO_NONE;
	}
	unsigned long long v[8], sum = 0;
#pragma unroll
	for (int k = 0; k < 8; k++) {
		v[k] = ((info[k] & MSX_PINFO_LIST) && info[k] != MSX_PINFO_NONE) ? ((1ull << 32) | (info[k] & ~MSX_PINFO_LIST)) : 0ull;
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
	// the positions go through LDS so that the copying below runs one pool per lane, neighbours side by side
	// (eight consecutive pools per thread is right for the scan and wrong for the stores: 0.41 against 0.18 ms)
#pragma unroll
	for (int k = 0; k < 8; k++) {
		s_pos[threadIdx.x * 8 + k] = v[k] ? run : ~0ull;
		run += v[k];
	}
	__syncthreads();
	const unsigned long long base_l = csr_tot[0], base_e = csr_tot[1];
	const int64_t c0 = (int64_t)blockIdx.x * MSX_PINFO_CHUNK;
#pragma unroll
	for (int r = 0; r < 8; r++) {
		const int q = r * MSX_BLOCK + threadIdx.x;
		const unsigned long long sc = s_pos[q];
		if (sc != ~0ull) {
			const int64_t g = c0 + q;
			const uint32_t nd = pinfo[g] & ~MSX_PINFO_LIST;
			const unsigned long long li = base_l + (sc >> 32), ei = base_e + (sc & 0xffffffffull);
			m_off[li] = (uint32_t)ei;
			const int32_t *src = tmp_fid + group_off[g];
			for (uint32_t j = 0; j < nd; ++j) m_fid[ei + j] = src[j];
		}
	}
}

__global__ void k_multi_advance(int64_t n_chunks, const unsigned long long *__restrict__ chunk_base,
                                unsigned long long *csr_tot, uint32_t *m_off) {
	const unsigned long long tot = chunk_base[n_chunks];
	const unsigned long long nl = csr_tot[0] + (tot >> 32), ne = csr_tot[1] + (tot & 0xffffffffull);
	csr_tot[0] = nl;
	csr_tot[1] = ne;
	m_off[nl] = (uint32_t)ne;     // CSR sentinel
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
int msx_grow_keep(msx_ctx *ctx, msx_buf *b, size_t bytes) {
	if (bytes <= b->cap && b->p) return MSX_OK;
	size_t want = bytes + bytes / 2 + 4096;
	void *np = nullptr;
	hipError_t e = hipMalloc(&np, want);
	if (e != hipSuccess) return msx_fail(ctx, MSX_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
	if (msx_poison_on()) MSX_HIP(ctx, hipMemsetAsync(np, 0xa5, want, ctx->stream));      // (tests: msx_reserve)
	if (b->p) {
		MSX_HIP(ctx, hipMemcpyAsync(np, b->p, b->cap, hipMemcpyDeviceToDevice, ctx->stream));
		MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		MSX_HIP(ctx, hipFree(b->p));
	}
	b->p = np;
	b->cap = want;
	return MSX_OK;
}

extern "C" void msx_profile_destroy(msx_ctx *ctx, msx_profile *p) {
	if (!p) return;
	if (ctx) msx_join(ctx);
	if (ctx && ctx->stream) (void)hipStreamSynchronize(ctx->stream);
	void *ptrs[] = {p->fmap, p->ui, p->d, p->dq, p->counters, p->U, p->a_in_recip ? nullptr : (void *)p->a, p->share, p->delta, p->iter_state,
	                p->m_off.p, p->m_fid.p, p->csr_tot, p->partial, p->purged_local,
	                p->t_key[0].p, p->t_key[1].p, p->rs_hist.p, p->rs_off.p, p->ck_hist.p, p->ck_off.p,
	                p->recip.p, p->runs.p, p->owned.p, p->part_key.p, p->part_val.p, p->m_off_alt.p, p->m_fid_alt.p,
	                p->head.p, p->hpos.p, p->d_tot,
	                p->t_val64[0].p, p->t_val64[1].p, p->gl_idx.p};
	for (void *q : ptrs)
		if (q) (void)hipFree(q);
	delete p;
}

extern "C" int msx_profile_create(msx_ctx *ctx, msx_profile **out, int32_t n_features, int32_t share_type,
                                  const int32_t *fmap, int32_t n_targets) {
	if (!ctx || !out) return MSX_ERR_ARG;
	msx_join(ctx);
	*out = nullptr;
	if (share_type < MSX_MULTI_ADD_ALL || share_type > MSX_MULTI_IGNORE)
		return msx_fail(ctx, MSX_ERR_SHARE_TYPE, "Do not understand share_type=%d", share_type);
	if (n_features < 0) return msx_fail(ctx, MSX_ERR_ARG, "n_features < 0");
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	msx_profile *p = new msx_profile();
	p->n_features = n_features;
	p->n_targets = fmap ? n_targets : n_features;
	p->share_type = share_type;
	const size_t nf = (size_t)(n_features > 0 ? n_features : 1);
	bool ok = hipMalloc((void **)&p->ui, nf * 4) == hipSuccess &&
	          hipMalloc((void **)&p->counters, 4 * 4) == hipSuccess &&
	          hipMalloc((void **)&p->U, nf * 8) == hipSuccess && hipMalloc((void **)&p->a, nf * 8) == hipSuccess &&
	          hipMalloc((void **)&p->share, nf * 8) == hipSuccess &&
	          hipMalloc((void **)&p->delta, 20 * 8) == hipSuccess &&
	          hipMalloc((void **)&p->iter_state, 4 * 4) == hipSuccess &&
	          hipMalloc((void **)&p->csr_tot, 2 * 8) == hipSuccess &&
	          hipMalloc((void **)&p->d_tot, 8 * 8) == hipSuccess &&
	          hipMalloc((void **)&p->partial, (size_t)(msx_apply_blocks(n_features) + 2 * msx_share_waves(ctx) / MSX_BLOCK + 16) * 8) == hipSuccess &&
	          hipMalloc((void **)&p->purged_local, 4) == hipSuccess;
	if (ok && share_type == MSX_MULTI_SHARE_EQUAL)
		ok = hipMalloc((void **)&p->d, nf * 8) == hipSuccess && hipMalloc((void **)&p->dq, nf * 8) == hipSuccess;
	if (ok && fmap && n_targets > 0) {
		ok = hipMalloc((void **)&p->fmap, (size_t)n_targets * 4) == hipSuccess &&
		     hipMemcpy(p->fmap, fmap, (size_t)n_targets * 4, hipMemcpyHostToDevice) == hipSuccess;
	}
	if (!ok) {
		int rc = msx_fail(ctx, MSX_ERR_NOMEM, "msx_profile_create: allocation failed: %s",
		                  hipGetErrorString(hipGetLastError()));
		msx_profile_destroy(ctx, p);
		return rc;
	}
	int rc = msx_grow_keep(ctx, &p->m_off, 4096);
	if (!rc) rc = msx_grow_keep(ctx, &p->m_fid, 4096);
	if (!rc) rc = msx_profile_reset(ctx, p);
	if (rc) {
		msx_profile_destroy(ctx, p);
		return rc;
	}
	*out = p;
	return MSX_OK;
}

extern "C" int msx_profile_reset(msx_ctx *ctx, msx_profile *p) {
	if (!ctx || !p) return MSX_ERR_ARG;
	msx_join(ctx);
	const size_t nf = (size_t)(p->n_features > 0 ? p->n_features : 1);
	MSX_HIP(ctx, hipMemsetAsync(p->ui, 0, nf * 4, ctx->stream));
	if (p->d) MSX_HIP(ctx, hipMemsetAsync(p->d, 0, nf * 8, ctx->stream));
	if (p->dq) MSX_HIP(ctx, hipMemsetAsync(p->dq, 0, nf * 8, ctx->stream));
	p->dq_dirty = false;
	MSX_HIP(ctx, hipMemsetAsync(p->counters, 0, 16, ctx->stream));
	MSX_HIP(ctx, hipMemsetAsync(p->csr_tot, 0, 16, ctx->stream));
	MSX_HIP(ctx, hipMemsetAsync(p->d_tot, 0, 64, ctx->stream));
	MSX_HIP(ctx, hipMemsetAsync(p->purged_local, 0, 4, ctx->stream));
	MSX_HIP(ctx, hipMemsetAsync(p->iter_state, 0, 16, ctx->stream));
	MSX_HIP(ctx, hipMemsetAsync(p->delta, 0, 160, ctx->stream));
	MSX_HIP(ctx, hipMemsetAsync(p->m_off.p, 0, 4, ctx->stream));
	p->lists_ub = p->entries_ub = 0;
	p->iter_k = 0;
	p->recip_valid = false;
	p->begun = false;
	p->transposed_valid = false;
	return MSX_OK;
}

// msx_profile_accumulate in two halves, so that the fused filter + profile call can run the
// per-pool accounting inside its best-hit kernel: prepare fills the kernel arguments (and grows
// the stores), finish counts the staged keys and appends the pools' lists to the multi-mapper CSR.
int msx_profile_count_prepare(msx_ctx *ctx, msx_profile *p, const msx_batch *b, const uint8_t *keep, CountArgs *out,
                              bool *by_part_out) {
	const int64_t n = b->n_records, ng = b->n_groups;
	const bool prop = p->share_type == MSX_MULTI_SHARE_PROPORTIONAL;
	int rc;
	if ((rc = msx_reserve(ctx, &ctx->tmp_fid, (size_t)n * 4))) return rc;
	if (prop) {
		// the bounds grow by the whole batch (the host never waits for the true counts); once they near the
		// 32-bit offsets of the store -- a file of billions of records, few of them multi-mapped -- they are
		// folded back to what the device has really stored (one 16-byte read-back)
		if (p->entries_ub + n > 0xffffff00LL || p->lists_ub + ng > 0xffffff00LL) {
			unsigned long long t[2] = {0, 0};
			MSX_HIP(ctx, hipMemcpyAsync(t, p->csr_tot, 16, hipMemcpyDeviceToHost, ctx->stream));
			MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
			p->lists_ub = (int64_t)t[0];
			p->entries_ub = (int64_t)t[1];
		}
		p->lists_ub += ng;
		p->entries_ub += n;
		if (p->entries_ub > 0xffffff00LL) return msx_fail(ctx, MSX_ERR_ARG, "multi-mapper CSR exceeds 2^32 entries");
		if ((rc = msx_grow_keep(ctx, &p->m_off, (size_t)(p->lists_ub + 2) * 4))) return rc;
		if ((rc = msx_grow_keep(ctx, &p->m_fid, (size_t)(p->entries_ub + 2) * 4))) return rc;
		p->transposed_valid = false;
	}
	CountArgs A = {};
	A.n_groups = ng;
	A.n_records = n;
	A.group_off = b->group_off;
	A.tid = b->tid;
	A.keep = keep;
	A.fmap = p->fmap;
	A.share_type = p->share_type;
	A.ui = p->ui;
	A.d = p->d;
	A.dq = p->dq;
	if (p->dq) p->dq_dirty = true;
	A.counters = p->counters;
	A.tmp_fid = (int32_t *)ctx->tmp_fid.p;
	if (b->pool_rule == MSX_POOLS_FILTER) {
		if (!b->flag) return msx_fail(ctx, MSX_ERR_ARG, "msx_batch.pool_rule = MSX_POOLS_FILTER needs the flag array");
		A.chain_flag = b->flag;
	}
	{
		// Staging-table size (measured on MI355X): with ~1 M features a large table is needed to
		// catch the hot references among the cold ones (2048: 1.5 ms vs 256: 2.7 ms at 20 M pools);
		// with ~10 k features every slot is hot and the end-of-kernel flush dominates (256: 0.29 ms
		// vs 2048: 0.56 ms at 2 M pools).
		int tbl = p->n_features > 100000 ? UI_TBL : 256;
#ifdef MSX_DEBUG_SWITCHES
		if (const char *e = getenv("MSX_UI_TBL")) {          // (libmsamtools_amd_dbg only: how the two sizes were chosen)
			int v = atoi(e);
			if (v >= 1) { tbl = 1; while (tbl < v && tbl < UI_TBL) tbl <<= 1; }
		}
#endif
		A.tbl_mask = (uint32_t)tbl - 1u;
	}
	// uniquely mapped inserts: counted by partition when there are enough pools to pay for the six
	// extra launches (measured on MI355X: 2 M pools / 10 k features 0.51 -> 0.17 ms, 20 M pools /
	// 1 M features 0.63 -> 0.23 ms; smaller batches not measured and left on the staging tables).
	// MSX_COUNT_BY_PARTITION=0/1 overrides.
	bool by_part = p->n_features <= MSX_COUNT_KEYS_MAX_FEATURES && ng >= (1 << 20);
	if (const char *e = getenv("MSX_COUNT_BY_PARTITION"))
		by_part = atoi(e) != 0 && p->n_features <= MSX_COUNT_KEYS_MAX_FEATURES;
	if (by_part && (rc = msx_reserve(ctx, &ctx->ukey2, (size_t)(ng + 8) * 4))) return rc;
	A.count_keys = by_part ? 1 : 0;
	if (by_part || prop) {
		if ((rc = msx_reserve(ctx, &ctx->pinfo, (size_t)(ng + 8) * 4))) return rc;
		A.pinfo = (uint32_t *)ctx->pinfo.p;
	}
	*out = A;
	*by_part_out = by_part;
	return MSX_OK;
}

// The two chains that follow the per-pool accounting are independent of each other (one reads the
// unique-insert keys and updates ui[], the other appends the multi-mapped pools' lists to the store):
// a caller that has forked runs the first on a side lane.
int msx_profile_count_finish(msx_ctx *ctx, msx_profile *p, const msx_batch *b, bool by_part) {
	const int64_t ng = b->n_groups;
	const bool prop = p->share_type == MSX_MULTI_SHARE_PROPORTIONAL;
	int rc;
	if (by_part) {
		msx_lane_enter(ctx, 0);
		rc = msx_count_keys(ctx, p, (const uint32_t *)ctx->pinfo.p, (uint32_t *)ctx->ukey2.p, ng, 2u);
		msx_lane_leave(ctx);
		if (rc) return rc;
	}
	if (prop) {
		const unsigned long long *chunk_base = nullptr;
		const int64_t n_chunks = (ng + MSX_PINFO_CHUNK - 1) / MSX_PINFO_CHUNK;
		if ((rc = msx_scan_pinfo_chunks(ctx, (const uint32_t *)ctx->pinfo.p, ng, &chunk_base))) return rc;
		msx_time_begin(ctx, MSX_K_MULTI_COMPACT);
		hipLaunchKernelGGL(k_multi_compact, dim3((unsigned)n_chunks), dim3(MSX_BLOCK), 0, ctx->stream, ng,
		                   b->group_off, (const uint32_t *)ctx->pinfo.p, chunk_base, (const int32_t *)ctx->tmp_fid.p,
		                   (const unsigned long long *)p->csr_tot, (uint32_t *)p->m_off.p, (int32_t *)p->m_fid.p);
		hipLaunchKernelGGL(k_multi_advance, dim3(1), dim3(1), 0, ctx->stream, n_chunks, chunk_base, p->csr_tot,
		                   (uint32_t *)p->m_off.p);
		msx_time_end(ctx);
	}
	MSX_HIP(ctx, hipGetLastError());
	return MSX_OK;
}

extern "C" int msx_profile_accumulate(msx_ctx *ctx, msx_profile *p, const msx_batch *b, const uint8_t *keep) {
	if (!ctx || !p || !b) return MSX_ERR_ARG;
	if (!b->group_off || !b->tid) return msx_fail(ctx, MSX_ERR_ARG, "msx_profile_accumulate needs tid and group_off");
	msx_join(ctx);
	if (b->n_records == 0 || b->n_groups == 0) return MSX_OK;
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	CountArgs A;
	bool by_part = false;
	int rc;
	if ((rc = msx_profile_count_prepare(ctx, p, b, keep, &A, &by_part))) return rc;
	msx_time_begin(ctx, MSX_K_INSERT_COUNT);
	hipLaunchKernelGGL(k_insert_count, dim3(msx_grid(ctx, b->n_groups, MSX_BLOCK)), dim3(MSX_BLOCK), 0, ctx->stream, A);
	msx_time_end(ctx);
	if (A.chain_flag) msx_profile_count_chains(ctx, A);
	return msx_profile_count_finish(ctx, p, b, by_part);
}

// ---- one process, several contexts: the inserts of one sample counted on several devices ----------------
// --multi equal: what the pools added to dq[] in units of 1/MSX_EQ_L becomes part of d[] (msx_count.h)
__global__ __launch_bounds__(MSX_BLOCK) void k_fold_equal(int32_t nf, unsigned long long *__restrict__ dq, double *__restrict__ d) {
	const int64_t stride = (int64_t)gridDim.x * MSX_BLOCK;
	for (int64_t i = (int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x; i < nf; i += stride) {
		const unsigned long long q = dq[i];
		if (q) { d[i] += (double)q / (double)MSX_EQ_L; dq[i] = 0; }
	}
}
int msx_profile_fold_equal(msx_ctx *ctx, msx_profile *p) {
	if (!p->dq || !p->dq_dirty) return MSX_OK;
	hipLaunchKernelGGL(k_fold_equal, dim3(msx_grid(ctx, p->n_features, MSX_BLOCK)), dim3(MSX_BLOCK), 0, ctx->stream, p->n_features,
	                   p->dq, p->d);
	p->dq_dirty = false;
	MSX_HIP(ctx, hipGetLastError());
	return MSX_OK;
}

// (--multi equal: the integer shares dq[] are added AS INTEGERS and folded into d[] once, by whoever reads d[] next -- N contexts'
// sums are then one context's, bit for bit; folding each side first added N rounded quotients instead of rounding one)
__global__ __launch_bounds__(MSX_BLOCK) void k_merge_counts(int32_t nf, const uint32_t *__restrict__ ui_src, uint32_t *__restrict__ ui,
                                                            const double *__restrict__ d_src, double *__restrict__ d,
                                                            const unsigned long long *__restrict__ dq_src, unsigned long long *__restrict__ dq,
                                                            const uint32_t *__restrict__ cnt_src, uint32_t *__restrict__ cnt) {
	const int64_t stride = (int64_t)gridDim.x * MSX_BLOCK;
	for (int64_t i = (int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x; i < nf; i += stride) {
		ui[i] += ui_src[i];
		if (d) d[i] += d_src[i];
		if (dq) dq[i] += dq_src[i];
	}
	if (blockIdx.x == 0 && threadIdx.x < 3) cnt[threadIdx.x] += cnt_src[threadIdx.x];     // inserts, uniq, multi
}

// m_off[L + j] = off_src[j] + E for j = 0 .. n (the sentinel included), then the totals
__global__ __launch_bounds__(MSX_BLOCK) void k_merge_offsets(int64_t n, const uint32_t *__restrict__ off_src,
                                                             unsigned long long *csr_tot, uint32_t *__restrict__ m_off) {
	const unsigned long long L = csr_tot[0], E = csr_tot[1];
	const int64_t stride = (int64_t)gridDim.x * MSX_BLOCK;
	for (int64_t j = (int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x; j <= n; j += stride) m_off[L + j] = off_src[j] + (uint32_t)E;
}
__global__ void k_merge_totals(unsigned long long n_lists, unsigned long long n_entries, unsigned long long *csr_tot) {
	csr_tot[0] += n_lists;
	csr_tot[1] += n_entries;
}

extern "C" int msx_profile_merge(msx_ctx *ctx, msx_profile *p, msx_ctx *src_ctx, msx_profile *q) {
	if (!ctx || !p || !src_ctx || !q) return MSX_ERR_ARG;
	if (p == q) return msx_fail(ctx, MSX_ERR_ARG, "msx_profile_merge: source and destination are the same profile");
	if (p->n_features != q->n_features || p->share_type != q->share_type)
		return msx_fail(ctx, MSX_ERR_ARG, "msx_profile_merge: the two profiles differ in features or --multi mode");
	msx_join(ctx);
	msx_join(src_ctx);
	// everything the source has enqueued must have landed
	MSX_HIP(src_ctx, hipSetDevice(src_ctx->device));
	MSX_HIP(src_ctx, hipStreamSynchronize(src_ctx->stream));
	unsigned long long t[2] = {0, 0};
	MSX_HIP(src_ctx, hipMemcpy(t, q->csr_tot, 16, hipMemcpyDeviceToHost));
	const int64_t nl = (int64_t)t[0], ne = (int64_t)t[1];
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	const size_t nf = (size_t)(p->n_features > 0 ? p->n_features : 1);
	// staging on the destination device: ui, d, dq, counters, offsets of the source
	const bool with_dq = q->dq && p->dq && q->dq_dirty;
	const size_t b_ui = nf * 4, b_d = q->d ? nf * 8 : 0, b_dq = with_dq ? nf * 8 : 0, b_off = (size_t)(nl + 1) * 4;
	const size_t o_d = (b_ui + 15) & ~(size_t)15, o_dq = o_d + ((b_d + 15) & ~(size_t)15), o_cnt = o_dq + ((b_dq + 15) & ~(size_t)15), o_off = o_cnt + 16;
	char *stage = nullptr;
	if (hipMalloc((void **)&stage, o_off + b_off + 16) != hipSuccess)
		return msx_fail(ctx, MSX_ERR_NOMEM, "msx_profile_merge: staging allocation failed");
	const bool same = ctx->device == src_ctx->device;
	auto copy = [&](void *dst, const void *src, size_t bytes) -> hipError_t {
		if (!bytes) return hipSuccess;
		return same ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream)
		            : hipMemcpyPeerAsync(dst, ctx->device, src, src_ctx->device, bytes, ctx->stream);
	};
	int rc = MSX_OK;
	hipError_t e = copy(stage, q->ui, b_ui);
	if (e == hipSuccess) e = copy(stage + o_d, q->d, b_d);
	if (e == hipSuccess) e = copy(stage + o_dq, q->dq, b_dq);
	if (e == hipSuccess) e = copy(stage + o_cnt, q->counters, 16);
	if (e == hipSuccess) e = copy(stage + o_off, q->m_off.p, b_off);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_merge_counts, dim3(msx_grid(ctx, p->n_features, MSX_BLOCK)), dim3(MSX_BLOCK), 0, ctx->stream,
		                   p->n_features, (const uint32_t *)stage, p->ui, (const double *)(stage + o_d), p->d ? p->d : nullptr,
		                   (const unsigned long long *)(stage + o_dq), with_dq ? p->dq : nullptr,
		                   (const uint32_t *)(stage + o_cnt), p->counters);
		if (with_dq) p->dq_dirty = true;
		if (p->share_type == MSX_MULTI_SHARE_PROPORTIONAL && nl > 0) {
			// fold the destination's bounds back to what it really holds, then make room for the source's lists
			unsigned long long mine[2] = {0, 0};
			e = hipMemcpyAsync(mine, p->csr_tot, 16, hipMemcpyDeviceToHost, ctx->stream);
			if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
			if (e == hipSuccess) {
				p->lists_ub = (int64_t)mine[0] + nl;
				p->entries_ub = (int64_t)mine[1] + ne;
				if (p->entries_ub > 0xffffff00LL) rc = msx_fail(ctx, MSX_ERR_ARG, "multi-mapper CSR exceeds 2^32 entries");
				if (!rc) rc = msx_grow_keep(ctx, &p->m_off, (size_t)(p->lists_ub + 2) * 4);
				if (!rc) rc = msx_grow_keep(ctx, &p->m_fid, (size_t)(p->entries_ub + 2) * 4);
				if (!rc) {
					e = copy((int32_t *)p->m_fid.p + mine[1], q->m_fid.p, (size_t)ne * 4);
					hipLaunchKernelGGL(k_merge_offsets, dim3(msx_grid(ctx, nl + 1, MSX_BLOCK)), dim3(MSX_BLOCK), 0, ctx->stream, nl,
					                   (const uint32_t *)(stage + o_off), p->csr_tot, (uint32_t *)p->m_off.p);
					hipLaunchKernelGGL(k_merge_totals, dim3(1), dim3(1), 0, ctx->stream, (unsigned long long)nl, (unsigned long long)ne,
					                   p->csr_tot);
					p->transposed_valid = false;
				}
			}
		}
	}
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	(void)hipFree(stage);
	if (rc) return rc;
	if (e != hipSuccess) return msx_fail(ctx, MSX_ERR_HIP, "msx_profile_merge failed: %s", hipGetErrorString(e));
	return MSX_OK;
}

extern "C" int msx_profile_accumulators(msx_ctx *ctx, msx_profile *p, uint32_t **ui, double **d, uint32_t **counters) {
	if (!ctx || !p) return MSX_ERR_ARG;
	msx_join(ctx);
	{ int frc = msx_profile_fold_equal(ctx, p); if (frc) return frc; }
	if (ui) *ui = p->ui;
	if (d) *d = p->d;
	if (counters) *counters = p->counters;
	return MSX_OK;
}

extern "C" int msx_profile_abundance_dev(msx_ctx *ctx, msx_profile *p, double **a) {
	if (!ctx || !p || !a) return MSX_ERR_ARG;
	msx_join(ctx);
	*a = p->a;
	return MSX_OK;
}

extern "C" int msx_profile_multi_size(msx_ctx *ctx, msx_profile *p, int64_t *n_lists, int64_t *n_entries) {
	if (!ctx || !p) return MSX_ERR_ARG;
	msx_join(ctx);
	unsigned long long t[2] = {0, 0};
	MSX_HIP(ctx, hipMemcpyAsync(t, p->csr_tot, 16, hipMemcpyDeviceToHost, ctx->stream));
	MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (n_lists) *n_lists = (int64_t)t[0];
	if (n_entries) *n_entries = (int64_t)t[1];
	return MSX_OK;
}

extern "C" int msx_profile_shared_size(msx_ctx *ctx, msx_profile *p, int64_t *n_lists, int64_t *n_entries) {
	if (!ctx || !p) return MSX_ERR_ARG;
	msx_join(ctx);
	unsigned long long t[2] = {0, 0};
	MSX_HIP(ctx, hipMemcpyAsync(t, p->d_tot, 16, hipMemcpyDeviceToHost, ctx->stream));
	MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (n_lists) *n_lists = (int64_t)t[0];
	if (n_entries) *n_entries = (int64_t)t[1];
	return MSX_OK;
}

// msx_runtime_warmup: this translation unit's code object loaded onto the device ahead of its first launch (the runtime loads a
// module when one of its kernels is first asked for: 2-10 ms each, otherwise paid by the first batches of a command)
void msx_touch_profile(void) {
	hipFuncAttributes attr;
	(void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_insert_count));
}
