// msx_count.h -- per-pool insert accounting shared by k_insert_count (msx_profile.hip)
// and the fused best-hit + count kernel (msx_filter.hip):
// mEstimateInsertCountOnFile/OnPool, msam_profile.c:65-243.
#ifndef MSX_COUNT_H
#define MSX_COUNT_H

#include "msx_internal.h"

#define MSX_EQ_L 5354228880ull        // lcm(1..24); 2^64 / (L / 3) = 1.0e10 additions before a word could wrap: more than uint32 inserts
#define MSX_EQ_KMAX 24u

struct CountArgs {
	int64_t n_groups;
	int64_t n_records;
	const uint32_t *group_off;
	const int32_t *tid;
	const uint8_t *keep;      // null: all records; else filter's keep codes (1 then 2 = output order)
	const int32_t *fmap;      // null: identity
	int32_t share_type;
	uint32_t *ui;
	double *d;
	// --multi equal: the 1/k of a pool with k <= MSX_EQ_KMAX distinct features is added as the INTEGER MSX_EQ_L / k into dq[]
	// (MSX_EQ_L = lcm(1..24): exact, and integer atomics do not care about their order); msx_profile_fold_equal turns
	// dq[] into d[] with one division per feature before anything reads d[].  Floating-point atomics in arrival order gave
	// sums whose last bits changed from run to run (tests/test_gpu_determinism.py); pools of more than 24 distinct
	// features still take that route.
	unsigned long long *dq;
	uint32_t *counters;       // {inserts, uniq, multi, purged}
	int32_t *tmp_fid;         // [n_records] scratch: a multi-mapped pool g's distinct features at tmp_fid[group_off[g]..]
	uint32_t tbl_mask;        // LDS staging table size - 1 (power of two, <= UI_TBL)
	// one word per pool for the passes that follow (null: not needed):
	//   feature id            uniquely mapped insert whose +2 is still to be counted (count_keys mode)
	//   MSX_PINFO_LIST | nd   multi-mapped insert whose nd distinct features go to the multi-mapper store
	//   MSX_PINFO_NONE        nothing to do
	uint32_t *pinfo;
	int32_t count_keys;       // unique inserts are counted afterwards by msx_count_keys, not by ui_add here
	// msx_batch.pool_rule == MSX_POOLS_FILTER: group_off are the pools of msam_filter.c:120-125,170, and the stream
	// profile reads is filter's output re-pooled by QNAME (msam_profile.c:223-232).  A filter pool that begins with
	// an UNMAPPED record was opened by that record's name while prev_read kept the name of the last mapped record;
	// the mapped records that follow in it carry that earlier name, so the pool's output belongs to the insert of
	// the pool before it.  An insert is then a *chain*: a pool that begins with a mapped record (or pool 0) and the
	// pools after it that begin with an unmapped one.  Chain members are left out by the per-pool kernels and
	// counted as one pool by k_insert_chains.  chain_flag = FLAG array (null: every pool is its own insert).
	const uint16_t *chain_flag;
};

// pool g continues the insert of pool g - 1
__device__ __forceinline__ bool pool_follows(const CountArgs &A, int64_t g) {
	if (!A.chain_flag || g <= 0 || g >= A.n_groups) return false;
	const uint32_t s = A.group_off[g];
	return s < A.group_off[g + 1] && (A.chain_flag[s] & MSX_F_UNMAP) != 0;
}
#define MSX_PINFO_LIST 0x80000000u
#define MSX_PINFO_NONE 0xffffffffu

// Per-workgroup staging of the per-reference adds in LDS: a small open-addressed
// table (feature -> pending count).  Hot references (a few references receive a
// large share of all inserts) collapse to one global atomic per workgroup; an
// add that finds no slot within 4 probes goes straight to global memory.
#define UI_TBL 2048
#define UI_EMPTY (-1)

__device__ __forceinline__ void ui_add(int32_t *s_key, uint32_t *s_val, uint32_t *ui, int32_t fid, uint32_t v,
                                       uint32_t mask) {
	uint32_t h = ((uint32_t)fid * 2654435761u) >> 21;   // 11 bits
#pragma unroll
	for (int probe = 0; probe < 4; ++probe) {
		const uint32_t slot = (h + probe) & mask;
		const int32_t old = atomicCAS(&s_key[slot], UI_EMPTY, fid);
		if (old == UI_EMPTY || old == fid) {
			atomicAdd(&s_val[slot], v);
			return;
		}
	}
	atomicAdd(&ui[fid], v);
}

// what one pool's walk accumulates: records with a reference, distinct features in
// first-appearance order (msam_profile.c:131-145; the first four in registers)
struct PoolAcc {
	uint32_t nvalid, nd;
	int32_t f0, f1, f2, f3;
	int32_t *lst;
};

__device__ __forceinline__ void pool_begin(const CountArgs &A, PoolAcc &v, uint32_t s) {
	v.nvalid = 0; v.nd = 0;
	v.f0 = v.f1 = v.f2 = v.f3 = -1;
	v.lst = A.tmp_fid + s;
}

// one record of the stream profile sees, given its tid
__device__ __forceinline__ void pool_visit(const CountArgs &A, PoolAcc &v, int32_t t) {
	// (written with selects: the lanes of a wave are at different points of their pools, and every branch here
	//  is an exec-mask round trip per lane group; only the fifth and later distinct features of a pool branch)
	const bool valid = t != -1;                      // msam_profile.c:223-225
	int32_t fid = t;
	if (A.fmap) fid = valid ? A.fmap[t] : -1;        // (kernel argument: a uniform branch)
	v.nvalid += valid ? 1u : 0u;
	bool seen = (v.nd > 0 && fid == v.f0) || (v.nd > 1 && fid == v.f1) || (v.nd > 2 && fid == v.f2) ||
	            (v.nd > 3 && fid == v.f3);
	if (valid && !seen && v.nd > 4)
		for (uint32_t k = 4; k < v.nd; ++k)
			if (v.lst[k] == fid) { seen = true; break; }
	const bool fresh = valid && !seen;
	v.f0 = (fresh && v.nd == 0) ? fid : v.f0;
	v.f1 = (fresh && v.nd == 1) ? fid : v.f1;
	v.f2 = (fresh && v.nd == 2) ? fid : v.f2;
	v.f3 = (fresh && v.nd == 3) ? fid : v.f3;
	// the first four stay in registers until the pool is done (pool_finish writes them out, and only
	// for a pool with two or more -- the scratch list of any other pool is never read)
	if (fresh && v.nd >= 4) v.lst[v.nd] = fid;
	v.nd += fresh ? 1u : 0u;
}

// The records of the pool at s whose bit is set in m1, then those in m2 (filter's output
// order: first-pass records, then second-pass records).  Memory-level parallelism: the
// tids of the first eight are fetched with independent loads before any is looked at.
__device__ __forceinline__ void pool_visit_masks(const CountArgs &A, PoolAcc &v, uint32_t s, uint32_t m1, uint32_t m2) {
	uint32_t idx[8];
	int32_t tv[8];
	uint32_t rest1 = m1, rest2 = m2;
#pragma unroll
	for (int q = 0; q < 8; q++) {
		uint32_t bpos = 0xffffffffu;
		if (rest1) { bpos = (uint32_t)__ffs((int)rest1) - 1u; rest1 &= rest1 - 1u; }
		else if (rest2) { bpos = (uint32_t)__ffs((int)rest2) - 1u; rest2 &= rest2 - 1u; }
		idx[q] = bpos;
	}
#pragma unroll
	for (int q = 0; q < 8; q++) tv[q] = (idx[q] != 0xffffffffu) ? A.tid[s + idx[q]] : -1;
#pragma unroll
	for (int q = 0; q < 8; q++) {
		if (__ballot(idx[q] != 0xffffffffu) == 0ull) break;   // (wave-uniform) no lane has a q-th kept record
		pool_visit(A, v, tv[q]);
	}
	// more than eight kept records: the rest one by one
	while (rest1) { const uint32_t bpos = (uint32_t)__ffs((int)rest1) - 1u; rest1 &= rest1 - 1u; pool_visit(A, v, A.tid[s + bpos]); }
	while (rest2) { const uint32_t bpos = (uint32_t)__ffs((int)rest2) - 1u; rest2 &= rest2 - 1u; pool_visit(A, v, A.tid[s + bpos]); }
}

struct BlockCounts {
	uint32_t ins, uniq, multi;
};

// the pool's insert: unique / multi-mapper accounting
__device__ __forceinline__ void pool_finish(const CountArgs &A, int64_t g, const PoolAcc &v, int32_t *s_key,
                                            uint32_t *s_val, BlockCounts &c) {
	uint32_t info = MSX_PINFO_NONE;
	if (v.nd >= 2) {
		v.lst[0] = v.f0; v.lst[1] = v.f1;
		if (v.nd > 2) v.lst[2] = v.f2;
		if (v.nd > 3) v.lst[3] = v.f3;
	}
	if (v.nvalid > 0) {
		c.ins++;                                          // one insert per pool (:230,:237)
		if (v.nd == 1) {                                  // :75-78, :87-91, :152-159
			if (A.count_keys) info = (uint32_t)v.f0;
			else ui_add(s_key, s_val, A.ui, v.f0, 2u, A.tbl_mask);
			c.uniq++;
		} else {
			c.multi++;                                    // :95, :162
			switch (A.share_type) {
			case MSX_MULTI_ADD_ALL:                       // :99-102, :169-173
				for (uint32_t k = 0; k < v.nd; ++k) ui_add(s_key, s_val, A.ui, v.lst[k], 2u, A.tbl_mask);
				break;
			case MSX_MULTI_SHARE_EQUAL:
				if (v.nvalid == 2) {                      // :103-106 (integer halves)
					ui_add(s_key, s_val, A.ui, v.f0, 1u, A.tbl_mask);
					ui_add(s_key, s_val, A.ui, v.f1, 1u, A.tbl_mask);
				} else {                                  // :175-182
					if (A.dq && v.nd <= MSX_EQ_KMAX) {
						const unsigned long long q = MSX_EQ_L / (unsigned long long)v.nd;
						for (uint32_t k = 0; k < v.nd; ++k) atomicAdd(&A.dq[v.lst[k]], q);
					} else {
						const double share = 1.0 / (double)v.nd;
						for (uint32_t k = 0; k < v.nd; ++k) atomicAdd(&A.d[v.lst[k]], share);
					}
				}
				break;
			case MSX_MULTI_SHARE_PROPORTIONAL:            // :107-121, :184-186
				info = MSX_PINFO_LIST | v.nd;
				break;
			default:                                      // MULTI_IGNORE
				break;
			}
		}
	}
	if (A.pinfo) A.pinfo[g] = info;
}

// workgroup prologue / epilogue around the pools loop
__device__ __forceinline__ void count_block_begin(const CountArgs &A, int32_t *s_key, uint32_t *s_val) {
	const int tbl = (int)A.tbl_mask + 1;
	for (int i = threadIdx.x; i < tbl; i += MSX_BLOCK) { s_key[i] = UI_EMPTY; s_val[i] = 0; }
	__syncthreads();
}

__device__ __forceinline__ void count_block_end(const CountArgs &A, int32_t *s_key, uint32_t *s_val,
                                                uint32_t (*s_c)[MSX_BLOCK / 64], BlockCounts c) {
	// one set of counter atomics per workgroup
	for (int d = 32; d > 0; d >>= 1) {
		c.ins += __shfl_down(c.ins, d, 64);
		c.uniq += __shfl_down(c.uniq, d, 64);
		c.multi += __shfl_down(c.multi, d, 64);
	}
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	if (lane == 0) { s_c[0][w] = c.ins; s_c[1][w] = c.uniq; s_c[2][w] = c.multi; }
	__syncthreads();
	if (threadIdx.x < 3) {
		uint32_t v = s_c[threadIdx.x][0] + s_c[threadIdx.x][1] + s_c[threadIdx.x][2] + s_c[threadIdx.x][3];
		if (v) atomicAdd(&A.counters[threadIdx.x], v);
	}
	// flush the staged adds (the barrier above ordered every ui_add before this)
	const int tbl = (int)A.tbl_mask + 1;
	for (int i = threadIdx.x; i < tbl; i += MSX_BLOCK) {
		const uint32_t v = s_val[i];
		if (v) atomicAdd(&A.ui[s_key[i]], v);
	}
}

// host side (msx_profile.hip)
int msx_profile_count_prepare(msx_ctx *ctx, msx_profile *p, const msx_batch *b, const uint8_t *keep, CountArgs *out,
                              bool *by_part_out);
int msx_profile_count_finish(msx_ctx *ctx, msx_profile *p, const msx_batch *b, bool by_part);
// --multi equal: d[] += dq[] / MSX_EQ_L, dq[] = 0 (a no-op when nothing was added since the last fold); on ctx->stream, behind
// msx_join -- every reader of d[] calls it first (accumulators, merge, all-reduce of the counts, prop_begin)
int msx_profile_fold_equal(msx_ctx *ctx, msx_profile *p);
void msx_profile_count_chains(msx_ctx *ctx, const CountArgs &A);      // k_insert_chains, after the per-pool kernel

#endif
