// msx_synth.hip -- synthetic workload generator (device kernels + host twin)

#include "msx_internal.h"
#include "msx_synth.h"

#include <cstdlib>
#include <cstring>

// ---------------------------------------------------------------------------
// generator: three passes, one lane per QNAME group
//   hits   -> h[g]                 -> scan -> group_off
//   sizes  -> n_cigar, md_len / record -> scans -> cigar_off, md_off
//   fill   -> every SoA array
// ---------------------------------------------------------------------------
__device__ __host__ static inline uint32_t mates_mask(const msx_synth_model *m, uint64_t g, uint32_t h) {
	uint32_t mask = 0;
	for (uint32_t k = 1; k < h; k++)
		if (msx_synth_mate(m, g, k) == 2) mask |= 1u << k;
	return mask;
}

__global__ __launch_bounds__(MSX_BLOCK) void k_synth_hits(msx_synth_model m, int64_t first, int64_t ng,
                                                          uint32_t *__restrict__ h) {
	const int64_t stride = (int64_t)gridDim.x * MSX_BLOCK;
	for (int64_t g = (int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x; g < ng; g += stride)
		h[g] = (uint32_t)msx_synth_hits(&m, (uint64_t)(first + g));
}

__global__ __launch_bounds__(MSX_BLOCK) void k_synth_sizes(msx_synth_model m, int64_t first, int64_t ng,
                                                           const uint32_t *__restrict__ group_off,
                                                           uint32_t *__restrict__ ncig, uint32_t *__restrict__ nmd) {
	const int64_t stride = (int64_t)gridDim.x * MSX_BLOCK;
	for (int64_t g = (int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x; g < ng; g += stride) {
		const uint32_t s = group_off[g], h = group_off[g + 1] - s;
		const uint64_t gg = (uint64_t)(first + g);
		const uint32_t mask = mates_mask(&m, gg, h);
		for (uint32_t k = 0; k < h; k++) {
			msx_synth_rec r;
			msx_synth_record(&m, gg, k, h, mask, &r);
			ncig[s + k] = r.n_cigar;
			nmd[s + k] = r.md_len;
		}
	}
}

struct SynthOut {
	uint16_t *flag;
	uint8_t *rflags;
	int32_t *tid, *pos, *nm, *as;
	const uint32_t *cigar_off, *md_off;
	uint32_t *cigar;
	uint8_t *md;
};

__global__ __launch_bounds__(MSX_BLOCK) void k_synth_fill(msx_synth_model m, int64_t first, int64_t ng,
                                                          const uint32_t *__restrict__ group_off, SynthOut o) {
	const int64_t stride = (int64_t)gridDim.x * MSX_BLOCK;
	for (int64_t g = (int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x; g < ng; g += stride) {
		const uint32_t s = group_off[g], h = group_off[g + 1] - s;
		const uint64_t gg = (uint64_t)(first + g);
		const uint32_t mask = mates_mask(&m, gg, h);
		for (uint32_t k = 0; k < h; k++) {
			msx_synth_rec r;
			msx_synth_record(&m, gg, k, h, mask, &r);
			const uint32_t i = s + k;
			o.flag[i] = r.flag;
			o.rflags[i] = (uint8_t)(MSX_HAS_MD | MSX_HAS_NM | MSX_HAS_AS);
			o.tid[i] = r.tid;
			o.pos[i] = r.pos;
			o.nm[i] = r.nm;
			o.as[i] = r.as;
			const uint32_t c0 = o.cigar_off[i];
			for (uint32_t q = 0; q < r.n_cigar; q++) o.cigar[c0 + q] = r.cigar[q];
			const uint32_t m0 = o.md_off[i];
			for (uint32_t q = 0; q < r.md_len; q++) o.md[m0 + q] = r.md[q];
		}
	}
}

static msx_synth_model make_model(const msx_synth_params *sp) {
	msx_synth_model m;
	m.seed = sp->seed;
	m.n_refs = sp->n_refs;
	m.lambda = sp->mean_extra_hits;
	m.log2n_q16 = msx_log2_q16((uint32_t)sp->n_refs);
	return m;
}

static int check_params(msx_ctx *ctx, const msx_synth_params *sp) {
	if (!sp || sp->n_groups < 0 || sp->n_refs < 1 || sp->mean_extra_hits < 0 || sp->mean_extra_hits > 4 ||
	    sp->first_group < 0)
		return msx_fail(ctx, MSX_ERR_ARG, "msx_synth: bad parameters (n_refs >= 1, 0 <= mean_extra_hits <= 4)");
	if (sp->n_groups * (int64_t)MSX_SYNTH_MAX_HITS > 0x7fffffffLL * 4)
		return msx_fail(ctx, MSX_ERR_ARG, "msx_synth: too many groups for one batch");
	return MSX_OK;
}

template <typename T>
static int dalloc(msx_ctx *ctx, T **p, size_t count) {
	*p = nullptr;
	hipError_t e = hipMalloc((void **)p, (count ? count : 1) * sizeof(T) + 64);
	if (e != hipSuccess) return msx_fail(ctx, MSX_ERR_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
	return MSX_OK;
}

extern "C" int msx_synth_device(msx_ctx *ctx, const msx_synth_params *sp, msx_batch *dev, msx_synth_sizes *sizes) {
	if (!ctx || !dev) return MSX_ERR_ARG;
	msx_join(ctx);
	int rc = check_params(ctx, sp);
	if (rc) return rc;
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	memset(dev, 0, sizeof(*dev));
	const msx_synth_model m = make_model(sp);
	const int64_t ng = sp->n_groups;
	uint32_t *h = nullptr, *goff = nullptr, *ncig = nullptr, *nmd = nullptr;
	uint32_t *coff = nullptr, *moff = nullptr;
	uint32_t last[1];
#define GEN_FAIL(code)                   \
	do {                                 \
		int c_ = (code);                 \
		if (h) hipFree(h);               \
		if (ncig) hipFree(ncig);         \
		if (nmd) hipFree(nmd);           \
		msx_batch_free(ctx, dev);        \
		return c_;                       \
	} while (0)
	if ((rc = dalloc(ctx, &h, (size_t)ng + 8))) return rc;
	if ((rc = dalloc(ctx, &goff, (size_t)ng + 8))) GEN_FAIL(rc);
	dev->group_off = goff;
	msx_time_begin(ctx, MSX_K_SYNTH);
	hipLaunchKernelGGL(k_synth_hits, dim3(msx_grid(ctx, ng, MSX_BLOCK)), dim3(MSX_BLOCK), 0, ctx->stream, m,
	                   sp->first_group, ng, h);
	msx_time_end(ctx);
	if ((rc = msx_scan_u32(ctx, h, goff, ng))) GEN_FAIL(rc);
	if (hipMemcpyAsync(last, goff + ng, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
	    hipStreamSynchronize(ctx->stream) != hipSuccess)
		GEN_FAIL(msx_fail(ctx, MSX_ERR_HIP, "synth: %s", hipGetErrorString(hipGetLastError())));
	const int64_t n = last[0];
	if (n > 0x7fffffffLL) GEN_FAIL(msx_fail(ctx, MSX_ERR_ARG, "synth: batch exceeds 2^31-1 records"));
	if ((rc = dalloc(ctx, &ncig, (size_t)n + 8)) || (rc = dalloc(ctx, &nmd, (size_t)n + 8))) GEN_FAIL(rc);
	if ((rc = dalloc(ctx, &coff, (size_t)n + 8))) GEN_FAIL(rc);
	dev->cigar_off = coff;
	if ((rc = dalloc(ctx, &moff, (size_t)n + 8))) GEN_FAIL(rc);
	dev->md_off = moff;
	msx_time_begin(ctx, MSX_K_SYNTH);
	hipLaunchKernelGGL(k_synth_sizes, dim3(msx_grid(ctx, ng, MSX_BLOCK)), dim3(MSX_BLOCK), 0, ctx->stream, m,
	                   sp->first_group, ng, (const uint32_t *)goff, ncig, nmd);
	msx_time_end(ctx);
	if ((rc = msx_scan_u32(ctx, ncig, coff, n)) || (rc = msx_scan_u32(ctx, nmd, moff, n))) GEN_FAIL(rc);
	uint32_t tc = 0, tm = 0;
	if (hipMemcpyAsync(&tc, coff + n, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
	    hipMemcpyAsync(&tm, moff + n, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
	    hipStreamSynchronize(ctx->stream) != hipSuccess)
		GEN_FAIL(msx_fail(ctx, MSX_ERR_HIP, "synth: %s", hipGetErrorString(hipGetLastError())));
	SynthOut o;
	uint32_t *cig = nullptr;
	uint8_t *md = nullptr;
	if ((rc = dalloc(ctx, &o.flag, (size_t)n))) GEN_FAIL(rc);
	dev->flag = o.flag;
	if ((rc = dalloc(ctx, &o.rflags, (size_t)n))) GEN_FAIL(rc);
	dev->rflags = o.rflags;
	if ((rc = dalloc(ctx, &o.tid, (size_t)n))) GEN_FAIL(rc);
	dev->tid = o.tid;
	if ((rc = dalloc(ctx, &o.pos, (size_t)n))) GEN_FAIL(rc);
	dev->pos = o.pos;
	if ((rc = dalloc(ctx, &o.nm, (size_t)n))) GEN_FAIL(rc);
	dev->nm = o.nm;
	if ((rc = dalloc(ctx, &o.as, (size_t)n))) GEN_FAIL(rc);
	dev->as = o.as;
	if ((rc = dalloc(ctx, &cig, (size_t)tc + 4))) GEN_FAIL(rc);
	dev->cigar = cig;
	if ((rc = dalloc(ctx, &md, (size_t)tm + 16))) GEN_FAIL(rc);
	dev->md = md;
	o.cigar_off = coff;
	o.md_off = moff;
	o.cigar = cig;
	o.md = md;
	msx_time_begin(ctx, MSX_K_SYNTH);
	hipLaunchKernelGGL(k_synth_fill, dim3(msx_grid(ctx, ng, MSX_BLOCK)), dim3(MSX_BLOCK), 0, ctx->stream, m,
	                   sp->first_group, ng, (const uint32_t *)goff, o);
	msx_time_end(ctx);
	if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess)
		GEN_FAIL(msx_fail(ctx, MSX_ERR_HIP, "synth fill: %s", hipGetErrorString(hipGetLastError())));
	hipFree(h);
	hipFree(ncig);
	hipFree(nmd);
#undef GEN_FAIL
	dev->n_records = n;
	dev->n_groups = ng;
	if (sizes) {
		sizes->n_records = n;
		sizes->n_cigar = tc;
		sizes->n_md = tm;
	}
	return MSX_OK;
}

// ---- host twin -------------------------------------------------------------

extern "C" void msx_synth_host_free(msx_batch *b) {
	if (!b) return;
	const void *ptrs[] = {b->flag, b->rflags, b->tid, b->pos, b->cigar_off, b->cigar,
	                      b->md_off, b->md, b->nm, b->as, b->group_off, b->qname_hash};
	for (const void *p : ptrs) free((void *)p);
	memset(b, 0, sizeof(*b));
}

extern "C" int msx_synth_host(const msx_synth_params *sp, msx_batch *hb, msx_synth_sizes *sizes) {
	if (!hb) return MSX_ERR_ARG;
	int rc = check_params(nullptr, sp);
	if (rc) return rc;
	memset(hb, 0, sizeof(*hb));
	const msx_synth_model m = make_model(sp);
	const int64_t ng = sp->n_groups;
	uint32_t *goff = (uint32_t *)malloc(((size_t)ng + 1) * 4);
	if (!goff) return msx_fail(nullptr, MSX_ERR_NOMEM, "malloc failed");
	int64_t n = 0;
	for (int64_t g = 0; g < ng; g++) {
		goff[g] = (uint32_t)n;
		n += msx_synth_hits(&m, (uint64_t)(sp->first_group + g));
	}
	goff[ng] = (uint32_t)n;
	const size_t nn = (size_t)(n ? n : 1);
	uint16_t *flag = (uint16_t *)malloc(nn * 2);
	uint8_t *rfl = (uint8_t *)malloc(nn);
	int32_t *tid = (int32_t *)malloc(nn * 4), *pos = (int32_t *)malloc(nn * 4);
	int32_t *nm = (int32_t *)malloc(nn * 4), *as = (int32_t *)malloc(nn * 4);
	// sizing pass first: fresh pages are expensive, so allocate exactly
	uint32_t *coff = (uint32_t *)malloc((nn + 1) * 4), *moff = (uint32_t *)malloc((nn + 1) * 4);
	if (!coff || !moff) {
		free(goff); free(coff); free(moff);
		return msx_fail(nullptr, MSX_ERR_NOMEM, "malloc failed");
	}
	uint32_t tc = 0, tm = 0;
	for (int64_t g = 0; g < ng; g++) {
		const uint32_t s = goff[g], h = goff[g + 1] - s;
		const uint64_t gg = (uint64_t)(sp->first_group + g);
		const uint32_t mask = mates_mask(&m, gg, h);
		for (uint32_t k = 0; k < h; k++) {
			msx_synth_rec r;
			msx_synth_record(&m, gg, k, h, mask, &r);
			coff[s + k] = tc;
			moff[s + k] = tm;
			tc += r.n_cigar;
			tm += r.md_len;
		}
	}
	coff[n] = tc;
	moff[n] = tm;
	uint32_t *cig = (uint32_t *)malloc((size_t)tc * 4 + 16);
	uint8_t *md = (uint8_t *)malloc((size_t)tm + 16);
	hb->group_off = goff; hb->flag = flag; hb->rflags = rfl; hb->tid = tid; hb->pos = pos;
	hb->nm = nm; hb->as = as; hb->cigar_off = coff; hb->md_off = moff; hb->cigar = cig; hb->md = md;
	if (!flag || !rfl || !tid || !pos || !nm || !as || !coff || !moff || !cig || !md) {
		msx_synth_host_free(hb);
		return msx_fail(nullptr, MSX_ERR_NOMEM, "malloc failed");
	}
	for (int64_t g = 0; g < ng; g++) {
		const uint32_t s = goff[g], h = goff[g + 1] - s;
		const uint64_t gg = (uint64_t)(sp->first_group + g);
		const uint32_t mask = mates_mask(&m, gg, h);
		for (uint32_t k = 0; k < h; k++) {
			msx_synth_rec r;
			msx_synth_record(&m, gg, k, h, mask, &r);
			const uint32_t i = s + k;
			flag[i] = r.flag;
			rfl[i] = (uint8_t)(MSX_HAS_MD | MSX_HAS_NM | MSX_HAS_AS);
			tid[i] = r.tid; pos[i] = r.pos; nm[i] = r.nm; as[i] = r.as;
			for (uint32_t q = 0; q < r.n_cigar; q++) cig[coff[i] + q] = r.cigar[q];
			memcpy(md + moff[i], r.md, r.md_len);
		}
	}
	hb->n_records = n;
	hb->n_groups = ng;
	if (sizes) {
		sizes->n_records = n;
		sizes->n_cigar = tc;
		sizes->n_md = tm;
	}
	return MSX_OK;
}
