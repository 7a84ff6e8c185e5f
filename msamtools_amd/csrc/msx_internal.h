// msx_internal.h -- shared internals of libmsamtools_amd.so (HIP, gfx950 only).
#ifndef MSX_INTERNAL_H
#define MSX_INTERNAL_H

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/msamtools_amd.h"

// MSX_GUARD=1 (tests; device-side AddressSanitizer is not to be had on this pool): every device allocation of the library gets
// 512 guard bytes in front and behind, filled with 0xC3 and looked at again when the allocation is freed and whenever
// msx_debug_guard_check is called -- a kernel that writes past the end (or before the start) of what it was given aborts the
// process with the allocation's size and the first damaged byte instead of spoiling a neighbour.  With the guard on,
// msx_reserve hands out exactly what was asked for (no slack), so "past the end" means past the request.  Reads are not seen
// (MSX_POISON covers unwritten words).  Every translation unit allocates through these two names (msx_guard.hip has the bodies).
hipError_t msx_guard_malloc(void **p, size_t n);
hipError_t msx_guard_free(void *p);
bool msx_guard_on();
#ifndef MSX_GUARD_IMPL
#define hipMalloc(p, n) msx_guard_malloc((void **)(p), (n))
#define hipFree(p) msx_guard_free((void *)(p))
#endif

#define MSX_BLOCK 256          // 4 wave64 per workgroup
#define MSX_WAVE 64

// BAM constants (SAMv1 4.2)
#define MSX_OP_MATCH 0
#define MSX_OP_INS 1
#define MSX_OP_DEL 2
#define MSX_OP_REF_SKIP 3
#define MSX_OP_SOFT_CLIP 4
#define MSX_OP_HARD_CLIP 5
#define MSX_OP_PAD 6
#define MSX_OP_EQUAL 7
#define MSX_OP_DIFF 8
#define MSX_F_UNMAP 4u
#define MSX_F_MATES 0xC0u      // BAM_FREAD1 | BAM_FREAD2

// kernel ids for the timing table
enum msx_kid {
	MSX_K_ALN_STATS = 0,   // k_aln_stats_flat
	MSX_K_BESTHIT,         // k_besthit_select
	MSX_K_EMIT,            // k_emit_groups / k_emit_count + k_emit_fill
	MSX_K_INSERT_COUNT,    // k_insert_count; the partition count of the unique-insert keys (msx_count_keys)
	MSX_K_MULTI_COMPACT,   // k_multi_compact + k_multi_advance
	MSX_K_GENERAL_RECIP,   // k_general_recip
	MSX_K_SHARE_REDUCE,    // k_share_reduce
	MSX_K_PARTIAL_REDUCE,  // k_partial_reduce
	MSX_K_PROP_APPLY,      // k_prop_begin / k_prop_apply + k_prop_finish / k_prop_purged
	MSX_K_LIST_ORDER,      // k_list_key, k_dup_mark, k_uniq_gather, k_entry_weight, k_part_index
	MSX_K_RS_HIST,         // k_rs_hist
	MSX_K_RS_SCATTER,      // k_rs_scatter
	MSX_K_COVERAGE,        // k_coverage_pileup
	MSX_K_SCAN,            // k_scan_reduce + k_scan_apply (one bracket per scan call)
	MSX_K_SYNTH,           // generator kernels
	MSX_K_COUNT
};

// device-side status words of one filter call
struct msx_dev_status {
	unsigned long long first_no_mdnm;  // min record index, ~0ull if none
	unsigned long long first_no_as;
	unsigned long long n_emit;
};

struct msx_buf {
	void *p = nullptr;
	size_t cap = 0;
};

// msx_coverage_collect: a sample's coverage items gathered batch after batch (msx_coverage.hip)
struct msx_cov_collect {
	bool active = false;
	bool pieces = false;            // the sample takes the one-word-per-piece form (else every batch is piled up the streamed way)
	bool cov_zeroed = false;        // cov[] holds marks of streamed batches (zeroed when the first of them came)
	int64_t n_items = 0;            // items kept so far (own slots and overflow lists of every batch, empty slots included)
	int64_t n_batches = 0, n_streamed = 0;
	int64_t total_len = 0;
	int32_t n_targets = 0;
	int32_t *cov = nullptr;
};

struct msx_timed {
	int kid;
	hipEvent_t a, b;
	// algorithmic bytes of the bracket: fixed + per_item * items, items = cap or, with a device-side
	// count, min(cap, mul * ceil(*n_ptr / div)) -- resolved when the figure is asked for
	int64_t bytes_fixed = 0, per_item = 0, cap = 0, div = 1, mul = 1;
	const unsigned long long *n_ptr = nullptr;
};

// A side lane: a second HIP stream with its own scan workspace, so that a chain of kernels that
// does not depend on what the main stream is doing can run beside it (the chains after the
// best-hit kernel are latency-bound and leave most of the chip idle when run one after another).
struct msx_lane {
	hipStream_t stream = nullptr;
	msx_buf scan_l1, scan_l2, scan_l3;
	hipEvent_t done = nullptr;
};
#define MSX_SIDE_LANES 2

struct msx_dist;     // msx_dist.hip: RCCL communicator of this rank
#define MSX_MAX_SLICES 4

struct msx_ctx {
	int device = 0;
	msx_dist *dist = nullptr;          // null: single GPU
	hipStream_t stream = nullptr;      // the stream every launch helper uses: the main one, or a side lane's
	                                   // between msx_lane_enter and msx_lane_leave
	hipStream_t main_stream = nullptr;
	msx_lane side[MSX_SIDE_LANES];
	hipEvent_t ev_fork = nullptr;
	int in_lane = -1;                  // side lane currently entered, -1 = main
	bool forked = false;
	bool lanes_ok = false;             // side lanes created; MSX_SERIAL=1 keeps everything on the main stream
	int lanes_state = 0;               // 0: to be made at the first fork, 1: made (or failed: lanes_ok says), -1: not wanted
	std::string err;
	int num_cu = 256;
	int blocks_per_cu = 8;            // grid cap of the grid-stride kernels (MSX_BLOCKS_PER_CU overrides)
	// workspace, grown on demand and kept
	msx_buf pool_code, gcount, gbase, scan_l1, scan_l2, scan_l3, pinfo, moff, tmp_fid, ukey2;
	msx_buf cv_key[2], cv_hist, cv_off, cv_start, cv_side;   // coverage: binned pile-up items; images of pre-reduced tiles
	msx_buf cv_targets, cvc_items, cvc_sups;                 // (first cell, length) per target; msx_coverage_collect's items
	msx_cov_collect cvc;
	uint32_t *cvc_flag = nullptr;      // page-locked: a batch's overflow flag on its way to the host
	msx_buf df_slots, df_size, df_tok;              // msx_deflate.hip: block slots, sizes + offsets, token scratch of the resident waves
	hipStream_t df_last = nullptr;     // the stream the encoder's scratch was last used on (ONE set per context: a launch on
	hipEvent_t df_done = nullptr;      // another stream waits for df_done first; growing the scratch drains df_last)
	bool df_used = false;
	// msx_inflate.hip: the lane-parallel inflater's match lists (one per resident workgroup) and the list of blocks it hands
	// back to the serial kernel -- one set per stream that launches it (launches on one stream follow each other; the command
	// line inflates the batch sent ahead on a stream of its own beside the main one)
	struct inf_set { hipStream_t stream = nullptr; msx_buf matches, retry; bool used = false; } inf[4];
	msx_dev_status *d_status = nullptr;
	msx_dev_status *h_status = nullptr;  // pinned
	bool filter_pending = false;
	// timing
	bool timing = false;
	std::vector<msx_timed> timed;
	std::vector<int> timed_open;     // stack of open brackets (nesting allowed)
	std::vector<hipEvent_t> event_pool;
};

struct msx_event {
	hipEvent_t ev = nullptr;
	bool recorded = false;
};

int msx_fail(msx_ctx *ctx, int code, const char *fmt, ...);
// fork / join around independent chains (no-ops returning false when timing is on or lanes are off):
//   if (msx_fork(ctx)) ...; msx_lane_enter(ctx, i); <launches>; msx_lane_leave(ctx); ...; msx_join(ctx);
bool msx_fork(msx_ctx *ctx);
void msx_lane_enter(msx_ctx *ctx, int lane);
void msx_lane_leave(msx_ctx *ctx);
void msx_join(msx_ctx *ctx);
int msx_reserve(msx_ctx *ctx, msx_buf *b, size_t bytes);
// MSX_POISON=1 (tests): fresh workspace holds 0xa5 bytes, not the zeros a new allocation tends to hold -- a kernel that reads a
// word nobody wrote then reads garbage here too.  Read the same way by every allocator, at every call (a test may switch it on
// in the middle of a process).
static inline bool msx_poison_on() { const char *e = getenv("MSX_POISON"); return e && atoi(e) != 0; }
// msx_inflate.hip: inflate + CRC check of n_blocks BGZF blocks on `stream`, waves_per_cu waves per compute unit (0: all the LDS holds); d_n_bad[0] (zeroed by the caller) counts the refused, d_n_bad[1] is the launch's ticket counter
int msx_bgzf_inflate_launch(msx_ctx *ctx, hipStream_t stream, int waves_per_cu, const uint8_t *d_comp, size_t comp_len,
                            const msx_bgzf_block *d_blocks, int64_t n_blocks, uint8_t *d_out, uint32_t *d_status, uint32_t *d_n_bad);
// msx_deflate.hip: the byte string in[0 .. total) -- total = *d_total (a device word) if given, else n_cap -- framed as stored BGZF
// blocks: block k at k * (0xff00 + 31), all full but the last; the launch is sized by n_cap
int msx_bgzf_store_launch(msx_ctx *ctx, hipStream_t stream, const uint8_t *d_in, const uint32_t *d_total, size_t n_cap, uint8_t *d_out);
extern "C" int64_t msx_bgzf_bound(int64_t n_bytes, int level);
// ... deflated (level >= 1) into BGZF blocks back to back in d_out, on `stream` (no use of the context's scan workspace: the
// stream may run beside the context's own); *d_out_total (device) = the stream's length
int msx_bgzf_deflate_launch(msx_ctx *ctx, hipStream_t stream, const uint8_t *d_in, const uint32_t *d_total, size_t n_cap, uint8_t *d_out,
                            uint32_t *d_out_total, int level);
extern thread_local std::string msx_tls_err;

#define MSX_HIP(ctx, call)                                                              \
	do {                                                                                \
		hipError_t e_ = (call);                                                         \
		if (e_ != hipSuccess)                                                           \
			return msx_fail((ctx), MSX_ERR_HIP, "%s failed: %s (%s:%d)", #call,         \
			                hipGetErrorString(e_), __FILE__, __LINE__);                 \
	} while (0)

// RAII-less timing bracket: records events around a launch when enabled
void msx_time_begin(msx_ctx *ctx, int kid);
void msx_time_end(msx_ctx *ctx);
// algorithmic bytes of the innermost open bracket (no-op unless timing is on)
void msx_time_bytes(msx_ctx *ctx, int64_t fixed, int64_t per_item, int64_t items_cap,
                    const unsigned long long *n_ptr = nullptr, int64_t div = 1, int64_t mul = 1);
#define MSX_TIMED(ctx, kid, stmt)    \
	do {                             \
		msx_time_begin((ctx), (kid)); \
		stmt;                        \
		msx_time_end((ctx));         \
	} while (0)

// Grid of a grid-stride kernel: enough 256-thread workgroups to fill the chip
// (8 per CU = 32 waves), times `over` for kernels whose tail/load balance gains
// from more, smaller shares (measured: stats and best-hit kernels, x4).
static inline int msx_grid_x(msx_ctx *ctx, int64_t items, int per_block, int over) {
	int64_t nb = (items + per_block - 1) / per_block;
	int64_t cap = (int64_t)ctx->num_cu * ctx->blocks_per_cu * over;
	if (nb > cap) nb = cap;
	if (nb < 1) nb = 1;
	return (int)nb;
}
static inline int msx_grid(msx_ctx *ctx, int64_t items, int per_block) { return msx_grid_x(ctx, items, per_block, 1); }

// exclusive scan: out[0..m] (m+1 entries, out[m] = total), u32
int msx_scan_u32(msx_ctx *ctx, const uint32_t *in, uint32_t *out, int64_t m);
// exclusive u64 scan of (1 << 32 | nd) over the pools whose word is MSX_PINFO_LIST | nd (msx_count.h), 0 for the others
int msx_scan_pinfo(msx_ctx *ctx, const uint32_t *pinfo, uint64_t *out, int64_t m);
// the same scan's chunk sums only (exclusive, chunks of MSX_PINFO_CHUNK pools; base[n_chunks] = total); *base_out lives in the scan workspace
#define MSX_PINFO_CHUNK 2048
int msx_scan_pinfo_chunks(msx_ctx *ctx, const uint32_t *pinfo, int64_t m, const unsigned long long **base_out,
                          const unsigned long long *n_ptr = nullptr);
int msx_scan_inclusive_u32(msx_ctx *ctx, uint32_t *data, int64_t m);
// exclusive u32 scan whose data length is known on the device only: min(m, mul * ceil(*n_ptr / div)) items
int msx_scan_u32_len(msx_ctx *ctx, const uint32_t *in, uint32_t *out, int64_t m, const unsigned long long *n_ptr,
                     int64_t div, int64_t mul);

// profile state (one sample)
#define PROP_MAX_BLOCKS 2048
struct msx_profile {
	int32_t n_features = 0, n_targets = 0, share_type = 0;
	int32_t *fmap = nullptr;          // device [n_targets] or null
	uint32_t *ui = nullptr;           // [n_features] global->ui_insert_count
	double *d = nullptr;              // [n_features] global->d_insert_count (EQUAL)
	unsigned long long *dq = nullptr; // [n_features] (EQUAL) the 1/k shares not yet folded into d[], in units of 1/MSX_EQ_L (msx_count.h)
	bool dq_dirty = false;            // dq[] holds something: msx_profile_fold_equal before d[] is read
	uint32_t *counters = nullptr;     // [4] {inserts, uniq, multi, purged}
	double *U = nullptr, *a = nullptr;   // [n_features] U(i), a(i,k)
	double *share = nullptr;          // [n_features] sum over multi-mappers of 1/S (all-reduced across ranks)
	double *delta = nullptr;          // [20] device, delta[k]
	int32_t *iter_state = nullptr;    // [4]: {done flag, iterations, -, -}
	unsigned long long *csr_tot = nullptr;   // device {n_lists, n_entries}
	double *partial = nullptr;        // device [workgroups of k_prop_apply]: their sums of diff^2
	uint32_t *purged_local = nullptr; // device [1]
	// multi-mapper store, list-major CSR (global->multi_mappers)
	msx_buf m_off;                    // u32 [n_lists+1]
	msx_buf m_fid;                    // i32 [n_entries]
	int64_t lists_ub = 0, entries_ub = 0;    // host upper bounds (capacity / grid sizing)
	// feature-major view built once per finalize by a stable radix sort
	msx_buf t_key[2];                 // ping-pong keys of the radix sorts (list keys, then entry keys)
	msx_buf t_val64[2];               // ping-pong 64-bit values travelling with them: set signatures (msx_prop.hip)
	msx_buf gl_idx;                   // u32 [general lists]: numbers of the lists k_general_recip handles
	msx_buf recip;                    // f64: a[] (n_features, padded to 8) and behind it recip[n_lists], w/S of the general lists
	bool a_in_recip = false;          // msx_prop_build has moved a[] into `recip` (one buffer descriptor for both)
	double *recip_ptr = nullptr;
	uint32_t recip_at = 0;            // bytes from a[] to recip[]
	msx_buf rs_hist, rs_off;          // radix-sort histograms
	msx_buf ck_hist, ck_off;          // the same for msx_count_keys, which runs on a side lane while the store is being built
	msx_buf part_key, part_val;       // boundary partials of k_share_reduce (2 per wave)
	msx_buf runs, owned;              // runs of partial slots (feature, first slot, count) and the bitmap of the features that own one
	msx_buf m_off_alt, m_fid_alt;         // derived store: renumbered, duplicate lists merged
	msx_buf head, hpos;                   // dedupe scratch (head: one word per sorted list, msx_count.h's encoding); hpos[u+1]-hpos[u] = weight of merged list u
	unsigned long long *d_tot = nullptr;  // device {lists, entries, general lists, short runs, long runs of partial slots} of the derived store
	int sorted_buf = 0;               // which ping-pong buffer holds the sorted pairs
	int key_bits = 0;                 // bits of the feature id in an entry key (the list weight sits above)
	bool transposed_valid = false;
	// slices of an iteration's local half (msx_profile_prop_local_slice): wave and feature cuts, read once per store
	int slice_n = 0;
	bool slice_valid = false;
	int64_t slice_wave[MSX_MAX_SLICES + 1] = {0};
	uint32_t slice_key[MSX_MAX_SLICES + 1] = {0}, slice_feat[MSX_MAX_SLICES + 1] = {0};
	int iter_k = 0;
	bool recip_valid = false;   // recip[] of the general lists belongs to the current a[]
	bool begun = false;
};

// proportional-sharing engine (msx_prop.hip)
int msx_prop_build(msx_ctx *ctx, msx_profile *p);            // feature-major view of the multi-mapper store
int msx_prop_iteration(msx_ctx *ctx, msx_profile *p, bool complete);   // share[f] = sum_j w_j/S_j over this rank's lists
int msx_prop_apply_launch(msx_ctx *ctx, msx_profile *p, int k, bool fused);
int64_t msx_share_waves(msx_ctx *ctx);
int64_t msx_apply_blocks(int32_t nf);
int msx_prop_purged_launch(msx_ctx *ctx, msx_profile *p, uint32_t *out_dev);
int msx_grow_keep(msx_ctx *ctx, msx_buf *b, size_t bytes);
#define MSX_SORT_TILE 4096      // keys per workgroup tile of a radix pass (msx_prop.hip: RS_TILE)
int msx_sort_keys32_reserve(msx_ctx *ctx, int64_t n, msx_buf *hist, msx_buf *off, int64_t *n_tiles_out);
int msx_sort_keys32(msx_ctx *ctx, uint32_t *k0, uint32_t *k1, int64_t n, int shift0, int bits, msx_buf *hist, msx_buf *off,
                    int *sel, int64_t counted_tiles = 0);
int64_t msx_sort_k32v8_bound(int64_t n);
int msx_sort_k32v8_reserve(msx_ctx *ctx, int64_t n, msx_buf *hist, msx_buf *off, int64_t *n_tiles_out);
int msx_sort_k32v8(msx_ctx *ctx, uint32_t *k0, const uint8_t *v0, uint32_t *k1, int64_t n, int shift, msx_buf *hist, msx_buf *off,
                   int64_t counted_tiles, int skip_bucket, uint32_t *lay);
// msx_dist.hip: in-place all-reduce(sum) on the ctx stream; no-ops without a communicator
int msx_dist_allreduce_share(msx_ctx *ctx, msx_profile *p);
// share[first, first + count) on the communicator's SIDE stream, behind what the context's stream holds now (slices); the
// context's stream is made to wait for every side all-reduce issued so far by msx_dist_side_join
int msx_dist_allreduce_share_side(msx_ctx *ctx, msx_profile *p, int32_t first, int32_t count);
int msx_dist_side_join(msx_ctx *ctx);
int msx_dist_allreduce_u32(msx_ctx *ctx, uint32_t *dev, size_t count);
// ui[key] += add for every key < 0x80000000 of keys[0..n) by partition + LDS counting (n_features <= 2 M)
int msx_count_keys(msx_ctx *ctx, msx_profile *p, const uint32_t *keys, uint32_t *key2, int64_t n, uint32_t add);
#define MSX_COUNT_KEYS_MAX_FEATURES (256 * 8192)

// (msx_runtime_warmup) every translation unit's code object loaded ahead of its first launch
void msx_touch_scan(void);
void msx_touch_filter(void);
void msx_touch_stats(void);
void msx_touch_profile(void);
void msx_touch_prop(void);
void msx_touch_coverage(void);
void msx_touch_unpack(void);
void msx_touch_inflate(void);
void msx_touch_deflate(void);

#endif
