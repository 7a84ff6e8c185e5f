// msx_scan.hip -- exclusive prefix sum over u32 / u64 (reduce-then-scan, 2048
// items per workgroup).  Used for the emit order of filter's output stream
// (u32), the radix passes, and for compacting multi-mapper lists, where one u64
// scan carries (list count << 32 | entry count) and reads the 4-byte per-pool
// words directly.  HBM-bound: reads the input twice, writes once.
// (Round 6 tried ONE pass of chained tiles -- a ticket per tile, sums and prefixes published under device-scope release /
// acquire, 64 predecessors per look back: correct (tests/test_gpu_scan.py) and slow here: 260 us per scan of the c3 step's
// 10^8 items against ~45, the step 6.9 ms against 3.95 -- a tile's walk back over the thousands of tiles in flight costs a
// memory round trip per look across the eight XCDs' L2s -- and the command line, whose batches it was meant for, did not
// gain: profiles/round6/scan_chain.md.)
#include "msx_internal.h"

#define SCAN_ITEMS 8
#define SCAN_CHUNK (MSX_BLOCK * SCAN_ITEMS)   // 2048

template <typename T>
__device__ __forceinline__ T wave_incl_scan(T v) {
	const int lane = threadIdx.x & 63;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		T o = __shfl_up(v, d, 64);
		if (lane >= d) v += o;
	}
	return v;
}

// block-wide exclusive scan of one value per thread; returns the exclusive
// prefix, *total = block sum.  s_w needs 4 slots.
template <typename T>
__device__ __forceinline__ T block_excl_scan(T v, T *s_w, T *total) {
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	T incl = wave_incl_scan<T>(v);
	if (lane == 63) s_w[w] = incl;
	__syncthreads();
	T base = 0, tot = 0;
#pragma unroll
	for (int i = 0; i < MSX_BLOCK / 64; i++) {
		T x = s_w[i];
		if (i < w) base += x;
		tot += x;
	}
	__syncthreads();
	*total = tot;
	return base + incl - v;
}

template <typename T>
__device__ __forceinline__ void load8(const T *__restrict__ in, int64_t base, int64_t m, T (&v)[SCAN_ITEMS]) {
	if (base + SCAN_ITEMS <= m) {
		// 16-byte vector loads (chunk bases are multiples of 2048 items)
		constexpr int PER = 16 / sizeof(T);
		using V = typename std::conditional<sizeof(T) == 4, uint4, ulonglong2>::type;
		const V *p = reinterpret_cast<const V *>(in + base);
#pragma unroll
		for (int q = 0; q < SCAN_ITEMS / PER; q++) {
			V x = p[q];
			const T *e = reinterpret_cast<const T *>(&x);
#pragma unroll
			for (int r = 0; r < PER; r++) v[q * PER + r] = e[r];
		}
	} else {
#pragma unroll
		for (int k = 0; k < SCAN_ITEMS; k++) v[k] = (base + k < m) ? in[base + k] : (T)0;
	}
}

// PINFO: the input is the per-pool word of msx_count.h (u32); a pool with a multi-mapper list of
// nd features counts as (1 << 32 | nd), every other pool as 0 -- the scan then carries
// (list count << 32 | entry count) without the 8-byte-per-pool array ever being written.
template <typename T, bool PINFO>
__device__ __forceinline__ void load8x(const void *__restrict__ in, int64_t base, int64_t m, T (&v)[SCAN_ITEMS]) {
	if (PINFO) {
		uint32_t w[SCAN_ITEMS];
		load8<uint32_t>(reinterpret_cast<const uint32_t *>(in), base, m, w);
#pragma unroll
		for (int k = 0; k < SCAN_ITEMS; k++)
			v[k] = ((w[k] & 0x80000000u) && w[k] != 0xffffffffu && base + k < m)
			           ? (T)((1ull << 32) | (unsigned long long)(w[k] & 0x7fffffffu)) : (T)0;
	} else {
		load8<T>(reinterpret_cast<const T *>(in), base, m, v);
	}
}

// Device-side length: the launches are sized for the host's upper bound m, but only the first
// min(m, mul * ceil(*n_ptr / div)) items hold data (the rest would be zeros); workgroups beyond
// that touch no memory.  n_ptr == nullptr: all m items.
struct ScanLen {
	const unsigned long long *n_ptr;
	int64_t div, mul;
};

__device__ __forceinline__ int64_t scan_len(const ScanLen &L, int64_t m) {
	if (!L.n_ptr) return m;
	const int64_t n = (int64_t)*L.n_ptr;
	const int64_t e = L.mul * ((n + L.div - 1) / L.div);
	return e < m ? e : m;
}

template <typename T, bool PINFO = false>
__global__ __launch_bounds__(MSX_BLOCK) void k_scan_reduce(const void *__restrict__ in, int64_t m_ub, ScanLen L,
                                                           T *__restrict__ partial) {
	__shared__ T s_w[4];
	const int64_t m = scan_len(L, m_ub);
	if ((int64_t)blockIdx.x * SCAN_CHUNK >= m) {
		if (threadIdx.x == 0) partial[blockIdx.x] = 0;
		return;
	}
	int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
	T v[SCAN_ITEMS];
	load8x<T, PINFO>(in, base, m, v);
	T s = 0;
#pragma unroll
	for (int k = 0; k < SCAN_ITEMS; k++) s += v[k];
	T tot;
	(void)block_excl_scan<T>(s, s_w, &tot);
	if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// carry == nullptr: single-chunk top level.  Writes out[i] for i < m and, from
// the last block, out[m] = grand total.
template <typename T, bool INCL, bool PINFO = false>
__global__ __launch_bounds__(MSX_BLOCK) void k_scan_apply(const void *in, T *out, int64_t m_ub, ScanLen L,
                                                          const T *__restrict__ carry) {
	__shared__ T s_w[4];
	const int64_t m = scan_len(L, m_ub);
	if ((int64_t)blockIdx.x * SCAN_CHUNK >= m && (blockIdx.x > 0 || m_ub > 0)) {
		// nothing but zeros from here on: only the grand total is still owed (at the host's index)
		if (!INCL && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[m_ub] = carry ? carry[blockIdx.x] : (T)0;
		return;
	}
	int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
	T v[SCAN_ITEMS];
	load8x<T, PINFO>(in, base, m, v);
	T s = 0;
#pragma unroll
	for (int k = 0; k < SCAN_ITEMS; k++) s += v[k];
	T tot;
	T ex = block_excl_scan<T>(s, s_w, &tot);
	T c = carry ? carry[blockIdx.x] : (T)0;
	T run = c + ex;
	T o[SCAN_ITEMS];
#pragma unroll
	for (int k = 0; k < SCAN_ITEMS; k++) {
		if (INCL) { run += v[k]; o[k] = run; }
		else { o[k] = run; run += v[k]; }
	}
	if (base + SCAN_ITEMS <= m) {
		constexpr int PER = 16 / sizeof(T);
		using V = typename std::conditional<sizeof(T) == 4, uint4, ulonglong2>::type;
		V *q = reinterpret_cast<V *>(out + base);
#pragma unroll
		for (int j = 0; j < SCAN_ITEMS / PER; j++) {
			V x;
			T *e = reinterpret_cast<T *>(&x);
#pragma unroll
			for (int r = 0; r < PER; r++) e[r] = o[j * PER + r];
			q[j] = x;
		}
	} else {
#pragma unroll
		for (int k = 0; k < SCAN_ITEMS; k++)
			if (base + k < m) out[base + k] = o[k];
	}
	if (!INCL && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[m_ub] = c + tot;
}

template <typename T, bool INCL = false, bool PINFO = false>
static int scan_rec(msx_ctx *ctx, const void *in, T *out, int64_t m, int level, ScanLen L = ScanLen{nullptr, 1, 1}) {
	int64_t nb = (m + SCAN_CHUNK - 1) / SCAN_CHUNK;
	if (nb < 1) nb = 1;
	if (nb == 1) {
		hipLaunchKernelGGL((k_scan_apply<T, INCL, PINFO>), dim3(1), dim3(MSX_BLOCK), 0, ctx->stream, in, out, m, L,
		                   (const T *)nullptr);
		return MSX_OK;
	}
	if (level > 2) return msx_fail(ctx, MSX_ERR_ARG, "scan: input too large");
	msx_buf *lv = level == 0 ? &ctx->scan_l1 : level == 1 ? &ctx->scan_l2 : &ctx->scan_l3;
	// partial sums [nb] followed by their exclusive scan [nb+1]
	int rc = msx_reserve(ctx, lv, (size_t)(2 * nb + 8) * sizeof(T));
	if (rc) return rc;
	T *partial = (T *)lv->p;
	T *pscan = partial + ((nb + 3) & ~(int64_t)3);
	hipLaunchKernelGGL((k_scan_reduce<T, PINFO>), dim3((unsigned)nb), dim3(MSX_BLOCK), 0, ctx->stream, in, m, L, partial);
	rc = scan_rec<T, false, false>(ctx, partial, pscan, nb, level + 1);
	if (rc) return rc;
	hipLaunchKernelGGL((k_scan_apply<T, INCL, PINFO>), dim3((unsigned)nb), dim3(MSX_BLOCK), 0, ctx->stream, in, out, m, L,
	                   (const T *)pscan);
	return MSX_OK;
}

int msx_scan_u32(msx_ctx *ctx, const uint32_t *in, uint32_t *out, int64_t m) {
	msx_time_begin(ctx, MSX_K_SCAN);
	msx_time_bytes(ctx, 0, 12, m);                       // reduce pass reads, apply pass reads and writes
	int rc = scan_rec<uint32_t>(ctx, in, out, m, 0);
	msx_time_end(ctx);
	if (rc) return rc;
	MSX_HIP(ctx, hipGetLastError());
	return MSX_OK;
}

// the same with a device-side length: items at and beyond min(m, mul * ceil(*n_ptr / div)) count as zeros and
// are neither read nor written; out[m] still receives the total
int msx_scan_u32_len(msx_ctx *ctx, const uint32_t *in, uint32_t *out, int64_t m, const unsigned long long *n_ptr,
                     int64_t div, int64_t mul) {
	msx_time_begin(ctx, MSX_K_SCAN);
	msx_time_bytes(ctx, 0, 12, m, n_ptr, div, mul);
	int rc = scan_rec<uint32_t>(ctx, in, out, m, 0, ScanLen{n_ptr, div, mul});
	msx_time_end(ctx);
	if (rc) return rc;
	MSX_HIP(ctx, hipGetLastError());
	return MSX_OK;
}

// inclusive, in place (each workgroup reads its 2048 items into registers before writing them)
int msx_scan_inclusive_u32(msx_ctx *ctx, uint32_t *data, int64_t m) {
	msx_time_begin(ctx, MSX_K_SCAN);
	msx_time_bytes(ctx, 0, 12, m);
	int rc = scan_rec<uint32_t, true>(ctx, data, data, m, 0);
	msx_time_end(ctx);
	if (rc) return rc;
	MSX_HIP(ctx, hipGetLastError());
	return MSX_OK;
}

// The chunk sums of the same scan only: base[c] = (lists << 32 | entries) of the pools before chunk c, chunks of
// MSX_PINFO_CHUNK pools, base[n_chunks] = the total.  The consumer (k_multi_compact, k_uniq_gather) scans inside
// its chunk itself -- the 8 bytes per pool of the full scan are neither written nor read.  n_ptr: the words at
// and beyond *n_ptr do not exist (device-side length), m is then the host's upper bound.
int msx_scan_pinfo_chunks(msx_ctx *ctx, const uint32_t *pinfo, int64_t m, const unsigned long long **base_out,
                          const unsigned long long *n_ptr) {
	static_assert(MSX_PINFO_CHUNK == SCAN_CHUNK, "k_multi_compact's chunk is the scan's");
	using T = unsigned long long;
	int64_t nb = (m + SCAN_CHUNK - 1) / SCAN_CHUNK;
	if (nb < 1) nb = 1;
	int rc = msx_reserve(ctx, &ctx->scan_l1, (size_t)(2 * nb + 8) * sizeof(T));
	if (rc) return rc;
	T *partial = (T *)ctx->scan_l1.p;
	T *pscan = partial + ((nb + 3) & ~(int64_t)3);
	msx_time_begin(ctx, MSX_K_SCAN);
	msx_time_bytes(ctx, 0, 4, m, n_ptr);                 // 4-byte words in, 8 bytes per 2048 of them out
	hipLaunchKernelGGL((k_scan_reduce<T, true>), dim3((unsigned)nb), dim3(MSX_BLOCK), 0, ctx->stream, (const void *)pinfo, m,
	                   ScanLen{n_ptr, 1, 1}, partial);
	rc = scan_rec<T, false, false>(ctx, partial, pscan, nb, 1);
	msx_time_end(ctx);
	if (rc) return rc;
	*base_out = pscan;
	MSX_HIP(ctx, hipGetLastError());
	return MSX_OK;
}

int msx_scan_pinfo(msx_ctx *ctx, const uint32_t *pinfo, uint64_t *out, int64_t m) {
	msx_time_begin(ctx, MSX_K_SCAN);
	msx_time_bytes(ctx, 0, 16, m);                       // 4-byte words in (twice), 8-byte sums out
	int rc = scan_rec<unsigned long long, false, true>(ctx, pinfo, (unsigned long long *)out, m, 0);
	msx_time_end(ctx);
	if (rc) return rc;
	MSX_HIP(ctx, hipGetLastError());
	return MSX_OK;
}


#ifdef MSX_DEBUG_SWITCHES
// (libmsamtools_amd_dbg only; tests/test_gpu_scan.py) the scan by itself on device arrays: exclusive (out[n] = total: n + 1 words)
// or inclusive in place; n_dev: a device-side length (null: n), as msx_scan_u32_len takes it
extern "C" int msx_debug_scan_u32(msx_ctx *ctx, const uint32_t *d_in, uint32_t *d_out, int64_t n, int inclusive, const unsigned long long *n_dev) {
	if (!ctx || n < 0) return MSX_ERR_ARG;
	msx_join(ctx);
	int rc = inclusive ? msx_scan_inclusive_u32(ctx, d_out, n) : n_dev ? msx_scan_u32_len(ctx, d_in, d_out, n, n_dev, 1, 1) : msx_scan_u32(ctx, d_in, d_out, n);
	if (rc) return rc;
	MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return MSX_OK;
}
#endif

// msx_runtime_warmup: this translation unit's code object loaded onto the device ahead of its first launch (the runtime loads a
// module when one of its kernels is first asked for: 2-10 ms each, otherwise paid by the first batches of a command)
void msx_touch_scan(void) {
	hipFuncAttributes attr;
	(void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_scan_reduce<uint32_t, false>));
}
