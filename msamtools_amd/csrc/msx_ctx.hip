// msx_ctx.hip -- context, workspace, error text, timing table, raw memory helpers.
#include "msx_internal.h"

#include <cstdlib>
#include <cstring>
#include <mutex>

thread_local std::string msx_tls_err;

int msx_fail(msx_ctx *ctx, int code, const char *fmt, ...) {
	char buf[1024];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	if (ctx) ctx->err = buf;
	msx_tls_err = buf;
	return code;
}

int msx_reserve(msx_ctx *ctx, msx_buf *b, size_t bytes) {
	if (bytes <= b->cap && b->p) return MSX_OK;
	size_t want = bytes + bytes / 8 + 256;   // slack so streaming batches settle quickly
	if (msx_guard_on()) want = bytes ? bytes : 16;            // (MSX_GUARD: the guard sits right behind the request)
	if (b->p) {
		MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		MSX_HIP(ctx, hipFree(b->p));
		b->p = nullptr;
		b->cap = 0;
	}
	hipError_t e = hipMalloc(&b->p, want);
	if (e != hipSuccess) {
		b->p = nullptr;
		return msx_fail(ctx, MSX_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
	}
	b->cap = want;
	if (msx_poison_on()) { MSX_HIP(ctx, hipMemsetAsync(b->p, 0xa5, want, ctx->stream)); MSX_HIP(ctx, hipStreamSynchronize(ctx->stream)); }
	return MSX_OK;
}

static void free_buf(msx_buf *b) {
	if (b->p) (void)hipFree(b->p);
	b->p = nullptr;
	b->cap = 0;
}

extern "C" int msx_abi_version(void) { return MSX_ABI_VERSION; }

extern "C" const char *msx_last_error(const msx_ctx *ctx) {
	return ctx ? ctx->err.c_str() : msx_tls_err.c_str();
}

// Every code object of the library onto the device, once per process and device, under one lock -- and msx_ctx_create does
// not return before it has happened.  The runtime loads a code object when one of its kernels is first asked for; two threads
// asking at once is what it does not take: the warm-up thread's hipFuncGetAttributes beside a device thread's first launch
// aborted the process now and then on inputs small enough for the two to meet ("Cannot find Symbol with name: k_status_init",
// hip_global.cpp; seen twice in some four hundred runs of the command-line tests), and two device threads of a several-context
// run could have met the same way.  With every module loaded before any context exists, no launch loads one.
// MSX_NO_MODULE_WARMUP=1: as before round 5 (modules loaded by the first launches).
static void msx_modules_load(int device_id) {
	static std::mutex mu;
	static bool ready[64];
	std::lock_guard<std::mutex> lk(mu);
	if (device_id < 0 || device_id >= 64 || ready[device_id]) return;
	if (!getenv("MSX_NO_MODULE_WARMUP")) {
		msx_touch_unpack(); msx_touch_inflate(); msx_touch_stats(); msx_touch_filter(); msx_touch_scan(); msx_touch_profile();
		msx_touch_deflate(); msx_touch_prop(); msx_touch_coverage();
		(void)hipGetLastError();
	}
	ready[device_id] = true;
}

extern "C" int msx_runtime_warmup(int device_id) {
	int ndev = 0;
	hipError_t e = hipGetDeviceCount(&ndev);
	if (e != hipSuccess || ndev <= 0)
		return msx_fail(nullptr, MSX_ERR_NO_DEVICE, "no HIP device available (%s); libmsamtools_amd has no CPU fallback",
		                e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
	if (device_id < 0 || device_id >= ndev) return msx_fail(nullptr, MSX_ERR_ARG, "device %d out of range (0..%d)", device_id, ndev - 1);
	if (hipSetDevice(device_id) != hipSuccess || hipFree(nullptr) != hipSuccess)      // (hipFree(0): the primary context, now)
		return msx_fail(nullptr, MSX_ERR_HIP, "runtime start-up failed: %s", hipGetErrorString(hipGetLastError()));
	msx_modules_load(device_id);
	return MSX_OK;
}

extern "C" int msx_ctx_create(msx_ctx **out, int device_id) {
	if (!out) return msx_fail(nullptr, MSX_ERR_ARG, "msx_ctx_create: null output pointer");
	*out = nullptr;
	int ndev = 0;
	hipError_t e = hipGetDeviceCount(&ndev);
	if (e != hipSuccess || ndev <= 0)
		return msx_fail(nullptr, MSX_ERR_NO_DEVICE,
		                "no HIP device available (%s); libmsamtools_amd has no CPU fallback",
		                e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
	if (device_id < 0 || device_id >= ndev)
		return msx_fail(nullptr, MSX_ERR_ARG, "device %d out of range (0..%d)", device_id, ndev - 1);
	hipDeviceProp_t prop;
	e = hipGetDeviceProperties(&prop, device_id);
	if (e != hipSuccess)
		return msx_fail(nullptr, MSX_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
		return msx_fail(nullptr, MSX_ERR_NO_DEVICE,
		                "device %d is %s; this library carries gfx950 (MI355X) code objects only",
		                device_id, prop.gcnArchName);
	msx_ctx *ctx = new msx_ctx();
	ctx->device = device_id;
	ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	if (const char *e = getenv("MSX_BLOCKS_PER_CU")) {
		int v = atoi(e);
		if (v >= 1 && v <= 64) ctx->blocks_per_cu = v;
	}
	if (const char *e = getenv("MSX_SCHED")) {       // how the host waits for the device: spin | yield | block (HIP's default: auto)
		const unsigned f = !strcmp(e, "spin") ? hipDeviceScheduleSpin : !strcmp(e, "yield") ? hipDeviceScheduleYield
		                 : !strcmp(e, "block") ? hipDeviceScheduleBlockingSync : hipDeviceScheduleAuto;
		(void)hipSetDevice(device_id);
		(void)hipSetDeviceFlags(f);
	}
	if (hipSetDevice(device_id) != hipSuccess ||
	    hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
	    hipMalloc((void **)&ctx->d_status, sizeof(msx_dev_status)) != hipSuccess ||
	    hipHostMalloc((void **)&ctx->h_status, sizeof(msx_dev_status), hipHostMallocDefault) != hipSuccess) {
		int rc = msx_fail(nullptr, MSX_ERR_HIP, "context setup failed: %s",
		                  hipGetErrorString(hipGetLastError()));
		delete ctx;
		return rc;
	}
	ctx->main_stream = ctx->stream;
	msx_modules_load(device_id);          // (waits for a warm-up thread that is in the middle of it; does it if nobody has)
	{
		// (the side lanes' streams are made when a step first forks: a stream is 7.5 ms -- its hardware queue and the queue's
		// 173 MB of host memory -- and a caller may not want them at all: msx_ctx_set_lanes)
		const char *ser = getenv("MSX_SERIAL");
		ctx->lanes_state = (ser && atoi(ser) != 0) ? -1 : 0;
	}
	*out = ctx;
	return MSX_OK;
}

// Side lanes on or off for this context (on: the default unless MSX_SERIAL=1).  A step over 10^8 records overlaps its independent
// scans and compactions on them (the c3 step 4.00 against 4.05 ms); the command line's batches of a million records do not gain
// what the two streams cost to make and to fork and join per batch (filter -b over 100 M records: pipeline 0.40 -> 0.37 s without).
extern "C" int msx_ctx_set_lanes(msx_ctx *ctx, int on) {
	if (!ctx) return MSX_ERR_ARG;
	if (ctx->forked || ctx->in_lane >= 0) return msx_fail(ctx, MSX_ERR_ARG, "msx_ctx_set_lanes inside a forked step");
	if (!on) { ctx->lanes_ok = false; ctx->lanes_state = -1; }
	else if (ctx->lanes_state < 0) ctx->lanes_state = 0;          // (made at the next fork)
	return MSX_OK;
}

// ---- side lanes -------------------------------------------------------------------------------
static void lanes_make(msx_ctx *ctx) {
	bool ok = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) == hipSuccess;
	for (int i = 0; ok && i < MSX_SIDE_LANES; i++)
		ok = hipStreamCreateWithFlags(&ctx->side[i].stream, hipStreamNonBlocking) == hipSuccess &&
		     hipEventCreateWithFlags(&ctx->side[i].done, hipEventDisableTiming) == hipSuccess;
	ctx->lanes_ok = ok;          // without them everything simply runs on the main stream
	ctx->lanes_state = 1;
}
bool msx_fork(msx_ctx *ctx) {
	if (ctx->lanes_state == 0 && !ctx->timing) lanes_make(ctx);
	if (!ctx->lanes_ok || ctx->timing || ctx->forked) return false;
	if (hipEventRecord(ctx->ev_fork, ctx->main_stream) != hipSuccess) return false;
	for (int i = 0; i < MSX_SIDE_LANES; i++)
		if (hipStreamWaitEvent(ctx->side[i].stream, ctx->ev_fork, 0) != hipSuccess) return false;
	ctx->forked = true;
	return true;
}

static void swap_scan_ws(msx_ctx *ctx, msx_lane &l) {
	std::swap(ctx->scan_l1, l.scan_l1);
	std::swap(ctx->scan_l2, l.scan_l2);
	std::swap(ctx->scan_l3, l.scan_l3);
}

void msx_lane_enter(msx_ctx *ctx, int lane) {
	if (!ctx->forked || ctx->in_lane >= 0) return;
	ctx->in_lane = lane;
	ctx->stream = ctx->side[lane].stream;
	swap_scan_ws(ctx, ctx->side[lane]);
}

void msx_lane_leave(msx_ctx *ctx) {
	if (ctx->in_lane < 0) return;
	swap_scan_ws(ctx, ctx->side[ctx->in_lane]);
	ctx->stream = ctx->main_stream;
	ctx->in_lane = -1;
}

void msx_join(msx_ctx *ctx) {
	// (the head of nearly every entry point: the calling thread may have last worked on another context's device)
	(void)hipSetDevice(ctx->device);
	if (!ctx->forked) return;
	msx_lane_leave(ctx);
	for (int i = 0; i < MSX_SIDE_LANES; i++) {
		(void)hipEventRecord(ctx->side[i].done, ctx->side[i].stream);
		(void)hipStreamWaitEvent(ctx->main_stream, ctx->side[i].done, 0);
	}
	ctx->forked = false;
}

extern "C" void msx_ctx_destroy(msx_ctx *ctx) {
	if (!ctx) return;
	(void)hipSetDevice(ctx->device);
	msx_join(ctx);
	msx_dist_finalize(ctx);
	if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
	for (int i = 0; i < MSX_SIDE_LANES; i++) {
		msx_lane &l = ctx->side[i];
		if (l.stream) { (void)hipStreamSynchronize(l.stream); (void)hipStreamDestroy(l.stream); }
		if (l.done) (void)hipEventDestroy(l.done);
		free_buf(&l.scan_l1); free_buf(&l.scan_l2); free_buf(&l.scan_l3);
	}
	if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
	if (ctx->df_used && ctx->df_last) (void)hipStreamSynchronize(ctx->df_last);
	if (ctx->df_done) (void)hipEventDestroy(ctx->df_done);
	if (ctx->cvc_flag) (void)hipHostFree(ctx->cvc_flag);
	for (auto &t : ctx->timed) {
		(void)hipEventDestroy(t.a);
		(void)hipEventDestroy(t.b);
	}
	for (auto &ev : ctx->event_pool) (void)hipEventDestroy(ev);
	msx_buf *bufs[] = {&ctx->pool_code, &ctx->gcount, &ctx->gbase, &ctx->scan_l1, &ctx->scan_l2,
	                   &ctx->scan_l3, &ctx->pinfo, &ctx->moff, &ctx->tmp_fid, &ctx->ukey2,
	                   &ctx->cv_key[0], &ctx->cv_key[1], &ctx->cv_hist, &ctx->cv_off, &ctx->cv_start, &ctx->cv_side, &ctx->cv_targets, &ctx->cvc_items, &ctx->cvc_sups, &ctx->df_slots, &ctx->df_size,
	                   &ctx->df_tok,
	                   &ctx->inf[0].matches, &ctx->inf[0].retry, &ctx->inf[1].matches, &ctx->inf[1].retry, &ctx->inf[2].matches, &ctx->inf[2].retry,
	                   &ctx->inf[3].matches, &ctx->inf[3].retry};
	for (auto *b : bufs) free_buf(b);
	if (ctx->d_status) (void)hipFree(ctx->d_status);
	if (ctx->h_status) (void)hipHostFree(ctx->h_status);
	if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
	delete ctx;
}

extern "C" void *msx_ctx_stream(msx_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" int msx_ctx_sync(msx_ctx *ctx) {
	if (!ctx) return MSX_ERR_ARG;
	msx_join(ctx);
	MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return MSX_OK;
}

// ---- timing ---------------------------------------------------------------

static const char *k_names[MSX_K_COUNT] = {
    "k_aln_stats_flat", "k_besthit_select", "k_emit_order", "k_insert_count", "k_multi_compact",
    "k_general_recip", "k_share_reduce", "k_partial_reduce", "k_prop_apply", "k_list_order", "k_rs_hist", "k_rs_scatter",
    "k_coverage_pileup", "scan", "synth"};

static hipEvent_t get_event(msx_ctx *ctx) {
	if (!ctx->event_pool.empty()) {
		hipEvent_t e = ctx->event_pool.back();
		ctx->event_pool.pop_back();
		return e;
	}
	hipEvent_t e = nullptr;
	(void)hipEventCreate(&e);
	return e;
}

void msx_time_begin(msx_ctx *ctx, int kid) {
	if (!ctx->timing) return;
	msx_timed t;
	t.kid = kid;
	t.a = get_event(ctx);
	t.b = get_event(ctx);
	(void)hipEventRecord(t.a, ctx->stream);
	ctx->timed_open.push_back((int)ctx->timed.size());
	ctx->timed.push_back(t);
}

void msx_time_bytes(msx_ctx *ctx, int64_t fixed, int64_t per_item, int64_t items_cap, const unsigned long long *n_ptr,
                    int64_t div, int64_t mul) {
	if (!ctx->timing || ctx->timed_open.empty()) return;
	msx_timed &t = ctx->timed[ctx->timed_open.back()];
	t.bytes_fixed = fixed; t.per_item = per_item; t.cap = items_cap;
	t.n_ptr = n_ptr; t.div = div > 0 ? div : 1; t.mul = mul > 0 ? mul : 1;
}

void msx_time_end(msx_ctx *ctx) {
	if (!ctx->timing || ctx->timed_open.empty()) return;
	const int idx = ctx->timed_open.back();
	ctx->timed_open.pop_back();
	(void)hipEventRecord(ctx->timed[idx].b, ctx->stream);
}

extern "C" int msx_timing_enable(msx_ctx *ctx, int on) {
	if (!ctx) return MSX_ERR_ARG;
	msx_join(ctx);
	ctx->timing = on != 0;
	return MSX_OK;
}

extern "C" int msx_timing_reset(msx_ctx *ctx) {
	if (!ctx) return MSX_ERR_ARG;
	msx_join(ctx);
	MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	for (auto &t : ctx->timed) {
		ctx->event_pool.push_back(t.a);
		ctx->event_pool.push_back(t.b);
	}
	ctx->timed.clear();
	ctx->timed_open.clear();
	return MSX_OK;
}

extern "C" int msx_timing_get(msx_ctx *ctx, const char *name, double *ms_total, int64_t *launches) {
	if (!ctx || !name) return MSX_ERR_ARG;
	msx_join(ctx);
	int kid = -1;
	for (int i = 0; i < MSX_K_COUNT; i++)
		if (strcmp(name, k_names[i]) == 0) kid = i;
	if (kid < 0) return msx_fail(ctx, MSX_ERR_ARG, "unknown kernel name '%s'", name);
	MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	double tot = 0;
	int64_t cnt = 0;
	for (auto &t : ctx->timed) {
		if (t.kid != kid) continue;
		float ms = 0;
		if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
			tot += ms;
			cnt++;
		}
	}
	if (ms_total) *ms_total = tot;
	if (launches) *launches = cnt;
	return MSX_OK;
}

// Sum of the algorithmic bytes the library attached to the timed launches of `name` (scans and radix
// passes, whose lengths live on the device); 0 for kernels it does not price.
extern "C" int msx_timing_get_bytes(msx_ctx *ctx, const char *name, int64_t *bytes_total) {
	if (!ctx || !name || !bytes_total) return MSX_ERR_ARG;
	msx_join(ctx);
	int kid = -1;
	for (int i = 0; i < MSX_K_COUNT; i++)
		if (strcmp(name, k_names[i]) == 0) kid = i;
	if (kid < 0) return msx_fail(ctx, MSX_ERR_ARG, "unknown kernel name '%s'", name);
	MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	int64_t tot = 0;
	for (auto &t : ctx->timed) {
		if (t.kid != kid) continue;
		int64_t items = t.cap;
		if (t.n_ptr) {
			unsigned long long nv = 0;
			MSX_HIP(ctx, hipMemcpy(&nv, t.n_ptr, 8, hipMemcpyDeviceToHost));
			const int64_t it = t.mul * (((int64_t)nv + t.div - 1) / t.div);
			if (it < items) items = it;
		}
		tot += t.bytes_fixed + t.per_item * items;
	}
	*bytes_total = tot;
	return MSX_OK;
}

// ---- raw memory helpers -----------------------------------------------------

extern "C" int msx_dev_alloc(msx_ctx *ctx, void **ptr, size_t bytes) {
	if (!ctx || !ptr) return MSX_ERR_ARG;
	*ptr = nullptr;
	hipError_t e = hipMalloc(ptr, bytes ? bytes : 16);
	if (e != hipSuccess)
		return msx_fail(ctx, MSX_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
	if (msx_poison_on()) { MSX_HIP(ctx, hipMemsetAsync(*ptr, 0xa5, bytes ? bytes : 16, ctx->stream)); MSX_HIP(ctx, hipStreamSynchronize(ctx->stream)); }   // (tests: msx_reserve)
	return MSX_OK;
}

extern "C" void msx_dev_free(msx_ctx *ctx, void *ptr) {
	(void)ctx;
	if (ptr) (void)hipFree(ptr);
}

extern "C" int msx_dev_zero(msx_ctx *ctx, void *ptr, size_t bytes) {
	if (!ctx) return MSX_ERR_ARG;
	msx_join(ctx);
	if (bytes) MSX_HIP(ctx, hipMemsetAsync(ptr, 0, bytes, ctx->stream));
	return MSX_OK;
}

extern "C" int msx_dev_to_host(msx_ctx *ctx, void *host, const void *dev, size_t bytes) {
	if (!ctx) return MSX_ERR_ARG;
	msx_join(ctx);
	if (bytes) {
		MSX_HIP(ctx, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
		MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	}
	return MSX_OK;
}

extern "C" int msx_host_to_dev(msx_ctx *ctx, void *dev, const void *host, size_t bytes) {
	if (!ctx) return MSX_ERR_ARG;
	msx_join(ctx);
	if (bytes) {
		MSX_HIP(ctx, hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, ctx->stream));
		MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	}
	return MSX_OK;
}

// ---- batches ---------------------------------------------------------------

template <typename T>
static int up(msx_ctx *ctx, const T *src, size_t count, size_t pad_elems, const T **dst) {
	*dst = nullptr;
	if (!src) return MSX_OK;
	void *p = nullptr;
	size_t bytes = (count + pad_elems) * sizeof(T);
	hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
	if (e != hipSuccess)
		return msx_fail(ctx, MSX_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
	*dst = (const T *)p;
	if (count) MSX_HIP(ctx, hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
	return MSX_OK;
}

extern "C" void msx_batch_free(msx_ctx *ctx, msx_batch *dev) {
	if (!dev) return;
	if (ctx && ctx->stream) (void)hipStreamSynchronize(ctx->stream);
	const void *ptrs[] = {dev->flag, dev->rflags, dev->tid, dev->pos, dev->cigar_off, dev->cigar,
	                      dev->md_off, dev->md, dev->nm, dev->as, dev->group_off, dev->qname_hash};
	for (const void *p : ptrs)
		if (p) (void)hipFree((void *)p);
	memset(dev, 0, sizeof(*dev));
}

extern "C" int msx_batch_upload(msx_ctx *ctx, const msx_batch *h, msx_batch *d) {
	if (!ctx || !h || !d) return MSX_ERR_ARG;
	msx_join(ctx);
	memset(d, 0, sizeof(*d));
	size_t n = (size_t)h->n_records;
	if (h->n_records < 0 || h->n_records > 0x7fffffffLL)
		return msx_fail(ctx, MSX_ERR_ARG, "batch of %lld records exceeds the 2^31-1 per-batch limit",
		                (long long)h->n_records);
	// arrays a caller does not need may be NULL (e.g. `profile` reads tid and group_off only);
	// each compute entry point checks for the arrays it reads
	size_t n_cig = (n && h->cigar_off) ? h->cigar_off[n] : 0, n_md = (n && h->md_off) ? h->md_off[n] : 0;
	d->n_records = h->n_records;
	d->n_groups = h->group_off ? h->n_groups : 0;
	d->pool_rule = h->pool_rule;
	int rc;
#define UP(field, count, pad) \
	if ((rc = up(ctx, h->field, (count), (pad), &d->field)) != MSX_OK) { msx_batch_free(ctx, d); return rc; }
	UP(flag, n, 0)
	UP(rflags, n, 0)
	UP(tid, n, 0)
	UP(pos, n, 0)
	UP(cigar_off, n + 1, 0)
	UP(cigar, n_cig, 4)
	UP(md_off, n + 1, 0)
	UP(md, n_md, 16)       // the stats kernel reads MD as aligned dwords
	UP(nm, n, 0)
	UP(as, n, 0)
	UP(group_off, h->group_off ? (size_t)h->n_groups + 1 : 0, 0)
	UP(qname_hash, n, 0)
#undef UP
	MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return MSX_OK;
}

// ---- stages: device buffers reused batch after batch (the command line's streaming path) -------------
struct msx_stage {
	msx_buf arr[12];          // flag rflags tid pos cigar_off cigar md_off md nm as group_off qname_hash
	msx_buf keep, emit, as_out;
};

extern "C" int msx_stage_create(msx_ctx *ctx, msx_stage **stage) {
	if (!ctx || !stage) return MSX_ERR_ARG;
	*stage = new msx_stage();
	return MSX_OK;
}

extern "C" void msx_stage_destroy(msx_ctx *ctx, msx_stage *st) {
	if (!st) return;
	if (ctx && ctx->stream) (void)hipStreamSynchronize(ctx->stream);
	for (auto &b : st->arr) free_buf(&b);
	free_buf(&st->keep);
	free_buf(&st->emit);
	free_buf(&st->as_out);
	delete st;
}

template <typename T>
static int stage_up(msx_ctx *ctx, msx_buf *buf, const T *src, size_t count, size_t pad_elems, const T **dst) {
	*dst = nullptr;
	if (!src) return MSX_OK;
	int rc = msx_reserve(ctx, buf, (count + pad_elems) * sizeof(T) + 16);
	if (rc) return rc;
	*dst = (const T *)buf->p;
	if (count) MSX_HIP(ctx, hipMemcpyAsync(buf->p, src, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
	return MSX_OK;
}

extern "C" int msx_stage_upload(msx_ctx *ctx, msx_stage *st, const msx_batch *h, msx_batch *d) {
	if (!ctx || !st || !h || !d) return MSX_ERR_ARG;
	msx_join(ctx);
	memset(d, 0, sizeof(*d));
	const size_t n = (size_t)h->n_records;
	if (h->n_records < 0 || h->n_records > 0x7fffffffLL)
		return msx_fail(ctx, MSX_ERR_ARG, "batch of %lld records exceeds the 2^31-1 per-batch limit", (long long)h->n_records);
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	const size_t n_cig = (n && h->cigar_off) ? h->cigar_off[n] : 0, n_md = (n && h->md_off) ? h->md_off[n] : 0;
	d->n_records = h->n_records;
	d->n_groups = h->group_off ? h->n_groups : 0;
	d->pool_rule = h->pool_rule;
	int rc;
#define UP(i, field, count, pad) if ((rc = stage_up(ctx, &st->arr[i], h->field, (count), (pad), &d->field)) != MSX_OK) return rc;
	UP(0, flag, n, 2)
	UP(1, rflags, n, 2)
	UP(2, tid, n, 0)
	UP(3, pos, n, 0)
	UP(4, cigar_off, n + 1, 2)
	UP(5, cigar, n_cig, 4)
	UP(6, md_off, n + 1, 2)
	UP(7, md, n_md, 16)
	UP(8, nm, n, 0)
	UP(9, as, n, 0)
	UP(10, group_off, h->group_off ? (size_t)h->n_groups + 1 : 0, 0)
	UP(11, qname_hash, n, 0)
#undef UP
	return MSX_OK;
}

extern "C" int msx_stage_outputs(msx_ctx *ctx, msx_stage *st, int64_t n_records, int want_as, msx_filter_out *out) {
	if (!ctx || !st || !out || n_records < 0) return MSX_ERR_ARG;
	msx_join(ctx);
	const size_t n = (size_t)(n_records > 0 ? n_records : 1);
	int rc;
	if ((rc = msx_reserve(ctx, &st->keep, n + 16))) return rc;
	if ((rc = msx_reserve(ctx, &st->emit, 4 * n + 16))) return rc;
	if (want_as && (rc = msx_reserve(ctx, &st->as_out, 4 * n + 16))) return rc;
	out->keep = (uint8_t *)st->keep.p;
	out->emit_idx = (int32_t *)st->emit.p;
	out->as_out = want_as ? (int32_t *)st->as_out.p : nullptr;
	return MSX_OK;
}

extern "C" int msx_host_alloc(msx_ctx *ctx, void **ptr, size_t bytes) {
	if (!ctx || !ptr) return MSX_ERR_ARG;
	*ptr = nullptr;
	hipError_t e = hipHostMalloc(ptr, bytes ? bytes : 16, hipHostMallocDefault);
	if (e != hipSuccess) return msx_fail(ctx, MSX_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
	return MSX_OK;
}

extern "C" void msx_host_free(msx_ctx *ctx, void *ptr) {
	(void)ctx;
	if (ptr) (void)hipHostFree(ptr);
}

extern "C" int msx_host_register(msx_ctx *ctx, void *ptr, size_t bytes) {
	if (!ctx || !ptr || !bytes) return MSX_ERR_ARG;
	hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterPortable);    // (pinned for every device: one process may drive several contexts)
	if (e != hipSuccess) return msx_fail(ctx, MSX_ERR_HIP, "hipHostRegister(%zu) failed: %s", bytes, hipGetErrorString(e));
	return MSX_OK;
}

extern "C" int msx_host_unregister(msx_ctx *ctx, void *ptr) {
	if (!ctx || !ptr) return MSX_ERR_ARG;
	hipError_t e = hipHostUnregister(ptr);
	if (e != hipSuccess) return msx_fail(ctx, MSX_ERR_HIP, "hipHostUnregister failed: %s", hipGetErrorString(e));
	return MSX_OK;
}

extern "C" int msx_dev_to_host_async(msx_ctx *ctx, void *host, const void *dev, size_t bytes) {
	if (!ctx) return MSX_ERR_ARG;
	msx_join(ctx);
	if (bytes) MSX_HIP(ctx, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
	return MSX_OK;
}

// ---- stream markers ------------------------------------------------------------------------------
extern "C" int msx_event_create(msx_ctx *ctx, msx_event **out) {
	if (!ctx || !out) return MSX_ERR_ARG;
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	msx_event *e = new msx_event();
	if (hipEventCreateWithFlags(&e->ev, hipEventDisableTiming) != hipSuccess) {
		delete e;
		return msx_fail(ctx, MSX_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(hipGetLastError()));
	}
	*out = e;
	return MSX_OK;
}

extern "C" int msx_event_record(msx_ctx *ctx, msx_event *e) {
	if (!ctx || !e) return MSX_ERR_ARG;
	MSX_HIP(ctx, hipEventRecord(e->ev, ctx->stream));
	e->recorded = true;
	return MSX_OK;
}

extern "C" int msx_event_wait(msx_ctx *ctx, msx_event *e) {
	if (!ctx || !e) return MSX_ERR_ARG;
	if (e->recorded) MSX_HIP(ctx, hipEventSynchronize(e->ev));
	return MSX_OK;
}

extern "C" void msx_event_destroy(msx_ctx *ctx, msx_event *e) {
	(void)ctx;
	if (!e) return;
	if (e->ev) (void)hipEventDestroy(e->ev);
	delete e;
}
