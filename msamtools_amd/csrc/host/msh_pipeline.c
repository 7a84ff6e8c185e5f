/*
 * msh_pipeline.c -- the decode | device | encode pipeline under the commands: batch slots, the decode stage (BGZF blocks
 * inflated here or handed to the device compressed, the speculative record chase, SAM text), page-locking, queues.
 * Counterpart of the read loop msam_helper.c:246-268 feeding msam_filter.c:119-186 / msam_profile.c:222-234.
 */
#include "msh_cli.h"

/* Fill `b` with up to `target` records; with pools, stop at the first pool
 * boundary at or after the target so no pool straddles two batches. */
void fill_filter_batch(reader *rd, rbatch *b, size_t target, int pools, int want_stats) {
	rb_clear(b);
	if (pools) rb_mark_group(b);
	for (;;) {
		const uint8_t *r;
		const char *q;
		int flush;
		if (!rd->have_pending) {
			if (rd->eof || msh_read(rd->in, &rd->rec) < 0) { rd->eof = 1; return; }
			rd->have_pending = 1;
		}
		r = (const uint8_t *)rd->rec.s;
		q = REC_QNAME(r);
		flush = rd->have_prev && strcmp(q, rd->prev_read) != 0;          /* :120-121 */
		if (pools) {
			if (flush && b->n >= target) return;                          /* record stays pending */
			if (flush && b->n > b->group_off[b->n_groups - 1]) rb_mark_group(b);
		} else if (b->n >= target) {
			return;
		}
		if (!(REC_FLAG(r) & 4)) {                                         /* :170, mapped records only */
			strcpy(rd->prev_read, q);
			rd->have_prev = 1;
		}
		rb_append(b, r, rd->rec.l, want_stats);
		rd->have_pending = 0;
	}
}

/* ---- bulk path for BAM input ---------------------------------------------------
 * The inflated BAM bytes form one contiguous span; record boundaries are found
 * with one cheap serial walk, everything else (aux scan for MD/NM/AS, pool
 * boundaries, SoA fill) runs on all host threads.  A batch ends at the last pool
 * boundary of the scanned records; the open pool stays in the span.
 * mode: 0 = no pools (plain -l/-p/-z), 1 = filter pools (msam_filter.c:120-125,
 * 170), 2 = profile pools (msam_profile.c:223-232). */
typedef struct {
	rbatch *b;
	const uint8_t *base;
	size_t n;
	int mode, want_stats, unmapped_visible;
	const char *carry_name;      /* QNAME of the last mapped / tid != -1 record of earlier batches */
} pack_job;

/* Does the record take part in the pool rule of `mode`?
 *   1  msam_filter.c:120-125,170: every record is compared, only a MAPPED one renames the read (rule_sees = 1 always)
 *   2  msam_profile.c:223-232: records with tid == -1 are skipped entirely
 *   3  profile's rule over the records filter can write when pools do not shape its output (no best hit): a record
 *      with tid == -1 is invisible to profile, an unmapped one is never written (unless -k -v, unmapped_visible) */
static inline int rec_names_pool(const uint8_t *r, int mode, int uv) {
	if (mode == 1) return !(REC_FLAG(r) & 4);
	if (mode == 2) return REC_TID(r) != -1;
	return REC_TID(r) != -1 && (uv || !(REC_FLAG(r) & 4));
}
static inline int rec_rule_sees(const uint8_t *r, int mode, int uv) { return mode == 1 ? 1 : rec_names_pool(r, mode, uv); }

/* QNAMEs of two records of the batch differ?  The earlier one may lie in another thread's range and not have been
 * through msh_rec_check yet: nothing is read beyond what the records' own lengths allow (a corrupt record is reported
 * by the thread that owns it; here it only must not send a string compare past the buffer). */
static int rec_name_differs(const rbatch *b, const uint8_t *base, size_t i, const uint8_t *r, const char *pn_rec, size_t pn_len_max) {
	const size_t len = b->rec_off[i + 1] - b->rec_off[i] - 4;
	const size_t lq = REC_LQNAME(r);
	size_t lp = 0;
	(void)base;
	if (32 + lq > len || lq == 0) return 1;
	while (lp < pn_len_max && pn_rec[lp]) lp++;
	if (lp + 1 != lq) return 1;
	return memcmp(REC_QNAME(r), pn_rec, lp) != 0;
}

static void pack_scan(void *arg, int tid, int nth) {
	pack_job *J = (pack_job *)arg;
	rbatch *b = J->b;
	size_t lo = J->n * (size_t)tid / (size_t)nth, hi = J->n * (size_t)(tid + 1) / (size_t)nth, i;
	/* QNAME of the nearest earlier record that counts for the pool rule (mapped for filter, tid != -1 for
	 * profile): found once by walking back from this thread's first record, then carried forward */
	const char *pn = NULL;
	size_t pn_max = 255;          /* bytes that may be read at pn */
	if (J->mode != 0) {
		size_t j = lo;
		while (j > 0) {
			const uint8_t *pr = J->base + b->rec_off[j - 1] + 4;
			const size_t plen = b->rec_off[j] - b->rec_off[j - 1] - 4;
			if (plen >= 32 && rec_names_pool(pr, J->mode, J->unmapped_visible)) {
				pn = REC_QNAME(pr);
				pn_max = plen - 32 < 255 ? plen - 32 : 255;
				break;
			}
			j--;
		}
		if (!pn) pn = J->carry_name;
	}
	for (i = lo; i < hi; i++) {
		const uint8_t *r = J->base + b->rec_off[i] + 4;
		size_t len = b->rec_off[i + 1] - b->rec_off[i] - 4;
		const uint8_t *p, *end = r + len, *md = NULL, *nm = NULL, *as = NULL;
		uint8_t bd = 0;
		msh_rec_check(r, len);
		b->flag[i] = (uint16_t)REC_FLAG(r);
		b->tid[i] = REC_TID(r);
		b->pos[i] = REC_POS(r);
		for (p = REC_AUX(r); p + 3 <= end; p += 2 + msh_aux_size(p + 2, end)) {
			if (p[0] == 'M' && p[1] == 'D' && !md) md = p + 2;
			else if (p[0] == 'N' && p[1] == 'M' && !nm) nm = p + 2;
			else if (p[0] == 'A' && p[1] == 'S' && !as) as = p + 2;
		}
		b->rflags[i] = (uint8_t)((md ? MSX_HAS_MD : 0) | (nm ? MSX_HAS_NM : 0) | (as ? MSX_HAS_AS : 0));
		b->nm[i] = nm ? (int32_t)msh_aux2i(nm) : 0;
		b->as[i] = as ? (int32_t)msh_aux2i(as) : 0;
		if (J->want_stats) {
			size_t ml = (md && *md == 'Z') ? strlen((const char *)md + 1) : 0;
			{ uint32_t nc_; (void)msh_real_cigar(r, len, &nc_, NULL); b->cigar_off[i + 1] = nc_; }    /* counts (the real CIGAR's: CG:B:I); prefix-summed afterwards */
			b->md_off[i + 1] = (uint32_t)ml;
			b->md_rel[i] = ml ? (uint32_t)(md + 1 - r) : 0;
		} else {
			b->cigar_off[i + 1] = 0;
			b->md_off[i + 1] = 0;
		}
		if (J->mode != 0 && rec_rule_sees(r, J->mode, J->unmapped_visible))
			bd = (pn && rec_name_differs(b, J->base, i, r, pn, pn_max)) ? 1 : 0;
		b->bound[i] = bd;
		if (J->mode != 0 && rec_names_pool(r, J->mode, J->unmapped_visible)) { pn = REC_QNAME(r); pn_max = len - 32 < 255 ? len - 32 : 255; }
	}
}

static void pack_copy(void *arg, int tid, int nth) {
	pack_job *J = (pack_job *)arg;
	rbatch *b = J->b;
	size_t lo = J->n * (size_t)tid / (size_t)nth, hi = J->n * (size_t)(tid + 1) / (size_t)nth, i;
	for (i = lo; i < hi; i++) {
		const uint8_t *r = J->base + b->rec_off[i] + 4;
		uint32_t nc = b->cigar_off[i + 1] - b->cigar_off[i], ml = b->md_off[i + 1] - b->md_off[i];
		if (nc) { uint32_t n_; memcpy(b->cigar + b->cigar_off[i], msh_real_cigar(r, b->rec_off[i + 1] - b->rec_off[i] - 4, &n_, NULL), 4 * (size_t)nc); }
		if (ml) memcpy(b->md + b->md_off[i], r + b->md_rel[i], ml);
	}
}

void fill_batch_bulk(reader *rd, rbatch *b, size_t target, int mode, int want_stats) {
	msh_in *in = rd->in;
	size_t len = 0, off = 0, n = 0, n_batch, i;
	const uint8_t *span;
	pack_job J;
	if (rd->consume_pending) { msh_span_consume(in, rd->consume_pending); rd->consume_pending = 0; }
	b->n = 0;
	b->n_groups = 0;
	for (;;) {
		/* 1. record boundaries (serial: a pointer chase over block_size fields) */
		for (;;) {
			span = msh_span(in, &len);
			while (off + 4 <= len) {
				int32_t bs = le32(span + off);
				if (bs < 32) mDie("Corrupt BAM record (block_size %d)", bs);
				if (off + 4 + (size_t)bs > len) break;
				b->n = n;
				rb_reserve(b);
				b->rec_off[n++] = off;
				off += 4 + (size_t)bs;
			}
			if (n > target || rd->eof) break;
			if (!msh_span_fill(in)) rd->eof = 1;
		}
		if (rd->eof && off != len) mDie("Truncated BAM record");
		b->n = n;
		rb_reserve(b);
		b->rec_off[n] = off;
		if (n == 0) { rd->done = 1; b->n = 0; return; }
		/* 2. parallel: aux scan, SoA scalars, pool boundaries */
		J.b = b; J.base = span; J.n = n; J.mode = mode; J.want_stats = want_stats; J.unmapped_visible = 0;
		J.carry_name = rd->have_prev ? rd->prev_read : NULL;
		msh_parallel(msh_threads(), pack_scan, &J);
		/* 3. where the batch ends: the last pool boundary (the open pool waits for more data) */
		n_batch = n;
		if (mode != 0 && !rd->eof) {
			size_t k = n;
			while (k > 1 && !b->bound[k - 1]) k--;
			n_batch = k - 1;
			if (n_batch == 0) {          /* one pool fills the whole span: read more */
				target = n + target;
				if (!msh_span_fill(in)) rd->eof = 1;
				continue;
			}
		}
		break;
	}
	/* 4. offsets and pools (serial prefix sums, a few ms per million records) */
	b->cigar_off[0] = 0;
	b->md_off[0] = 0;
	for (i = 0; i < n_batch; i++) {
		b->cigar_off[i + 1] += b->cigar_off[i];
		b->md_off[i + 1] += b->md_off[i];
	}
	if (mode != 0) {
		b->n = 0;
		rb_mark_group(b);
		for (i = 1; i < n_batch; i++)
			if (b->bound[i]) { b->n = i; rb_mark_group(b); }
	}
	if (want_stats) {
		size_t nc = b->cigar_off[n_batch], nm = b->md_off[n_batch];
		if (nc + 4 > b->cigar_cap) { b->cigar_cap = nc + nc / 4 + 1024; b->cigar = (uint32_t *)realloc(b->cigar, b->cigar_cap * 4); msh_huge_hint(b->cigar, b->cigar_cap * 4); }
		if (nm + 16 > b->md_cap) { b->md_cap = nm + nm / 4 + 4096; b->md = (uint8_t *)realloc(b->md, b->md_cap); msh_huge_hint(b->md, b->md_cap); }
		if (!b->cigar || !b->md) mDie("Out of memory");
		J.n = n_batch;
		msh_parallel(msh_threads(), pack_copy, &J);
	}
	b->n = n_batch;
	b->base = span;
	/* carry the grouping state into the next batch */
	for (i = n_batch; i > 0; i--) {
		const uint8_t *r = span + b->rec_off[i - 1] + 4;
		if (mode == 2 ? (REC_TID(r) != -1) : !(REC_FLAG(r) & 4)) {
			strcpy(rd->prev_read, REC_QNAME(r));
			rd->have_prev = 1;
			break;
		}
	}
	rd->consume_pending = b->rec_off[n_batch];
	if (rd->eof && n_batch == n) rd->done = 1;
}
#define PIPE_SLOTS_COMP 6               /* slots when the batches arrive compressed (a slot is 40 MB of payloads then) */
#define BGZF_INFLATE_MAX ((size_t)1024 * 65536)   /* msh_inflate_append appends at most one batch of blocks (msh_io.c: BGZF_BATCH x BGZF_MAX) */
static void pq_init(pq *q) { pthread_mutex_init(&q->mu, NULL); pthread_cond_init(&q->cv, NULL); q->n = 0; }
void pq_push(pq *q, int v) {
	pthread_mutex_lock(&q->mu);
	q->item[q->n++] = v;
	pthread_cond_signal(&q->cv);
	pthread_mutex_unlock(&q->mu);
}
int pq_pop(pq *q) {
	int v, i;
	pthread_mutex_lock(&q->mu);
	while (q->n == 0) pthread_cond_wait(&q->cv, &q->mu);
	v = q->item[0];
	for (i = 1; i < q->n; i++) q->item[i - 1] = q->item[i];
	q->n--;
	pthread_mutex_unlock(&q->mu);
	return v;
}
int pq_try_pop(pq *q) {
	int v = PQ_NONE, i;
	pthread_mutex_lock(&q->mu);
	if (q->n > 0) {
		v = q->item[0];
		for (i = 1; i < q->n; i++) q->item[i - 1] = q->item[i];
		q->n--;
	}
	pthread_mutex_unlock(&q->mu);
	return v;
}

/* the next stretch of BAM record bytes of the input, whatever its format: inflated BGZF blocks, or SAM text parsed
 * into records (the reference reads both through sam_read1: msam_helper.c:246-268; its validation harness feeds .sam) */
static size_t pipe_append(msh_in *in, uint8_t **buf, size_t *len, size_t *cap) {
	return msh_is_bam(in) ? msh_inflate_append(in, buf, len, cap) : msh_sam_append(in, buf, len, cap);
}

static size_t env_size(const char *name, size_t dflt) {
	const char *e = getenv(name);
	long long v = e ? strtoll(e, NULL, 10) : 0;
	return v > 0 ? (size_t)v : dflt;
}

void *xmalloc(size_t n) {
	void *p = malloc(n ? n : 1);
	if (!p) mDie("Out of memory");
	msh_huge_hint(p, n);
	return p;
}

void pipe_init(pipe_t *P, msh_in *in, int mode, int want_stats, int n_consumers) {
	int i;
	memset(P, 0, sizeof *P);
	P->in = in;
	P->n_consumers = n_consumers < 1 ? 1 : n_consumers;
	P->n_slots = (int)env_size("MSX_SLOTS", PIPE_SLOTS + 1) + P->n_consumers - 1;
	if (P->n_slots < 2) P->n_slots = 2;
	if (P->n_slots > PIPE_SLOTS_MAX) P->n_slots = PIPE_SLOTS_MAX;
	P->hdr = msh_header(in);
	P->mode = mode;
	P->want_stats = want_stats;
	P->batch_bytes = P->batch_bytes_cfg = env_size("MSX_BATCH_BYTES", (size_t)96 << 20);
	P->cap_rec = env_size("MSX_BATCH_RECORDS", (size_t)3 << 20);
	if (P->cap_rec < COORD_ORDER_CHECK_RECORDS + 1024) P->cap_rec = COORD_ORDER_CHECK_RECORDS + 1024;
	P->cap_cig = want_stats ? 2 * P->cap_rec : 4;
	P->cap_md = want_stats ? 16 * P->cap_rec : 16;
	pq_init(&P->q_free); pq_init(&P->q_dev); pq_init(&P->q_out);
	pthread_mutex_init(&P->ob_mu, NULL);
	pthread_cond_init(&P->ob_cv, NULL);
	for (i = 0; i < PIPE_SLOTS_MAX; i++) P->slot[i].ob = -1;
	pthread_mutex_init(&P->first_mu, NULL);
	pthread_mutex_init(&P->baton_mu, NULL);
	pthread_cond_init(&P->baton_cv, NULL);
	P->baton_seq = (size_t)-1;
	pthread_cond_init(&P->first_cv, NULL);
	for (i = 0; i < P->n_slots; i++) {
		pslot *s = &P->slot[i];
		rbatch *b = &s->b;
		const size_t c = P->cap_rec + 8;
		b->cap = c;
		b->rec_off = (size_t *)xmalloc((c + 1) * sizeof(size_t));
		b->flag = (uint16_t *)xmalloc(c * 2);
		b->rflags = (uint8_t *)xmalloc(c);
		b->tid = (int32_t *)xmalloc(c * 4);
		b->pos = (int32_t *)xmalloc(c * 4);
		b->nm = (int32_t *)xmalloc(c * 4);
		b->as = (int32_t *)xmalloc(c * 4);
		b->cigar_off = (uint32_t *)xmalloc((c + 1) * 4);
		b->md_off = (uint32_t *)xmalloc((c + 1) * 4);
		b->md_rel = (uint32_t *)xmalloc(c * 4);
		b->bound = (uint8_t *)xmalloc(c);
		b->cigar_cap = P->cap_cig; b->cigar = (uint32_t *)xmalloc(b->cigar_cap * 4);
		b->md_cap = P->cap_md; b->md = (uint8_t *)xmalloc(b->md_cap);
		b->group_cap = c + 1; b->group_off = (uint32_t *)xmalloc(b->group_cap * 4);
		s->emit = (int32_t *)xmalloc(c * 4);
		if (getenv("MSX_TIMING") && atoi(getenv("MSX_TIMING")) >= 2)
			fprintf(stderr, "# slot %d: SoA arrays from %p (rec_off) to %p (emit + %zu)\n", i, (void *)b->rec_off, (void *)s->emit, c * 4);
		pq_push(&P->q_free, i);
	}
}

/* An I/O buffer of the device-unpack path: anonymous memory advised for huge pages.  Page-locking costs next to nothing
 * once the pages exist (0.4 ms for 80 MB, scripts/micro/pin_rate.hip) -- what takes the time is faulting them in, and
 * that needs no HIP call and holds no lock of the runtime: io_populate does it (MADV_POPULATE_WRITE: contents untouched,
 * so a buffer the decode stage is already filling may be populated) on the pin threads while HIP is starting up. */
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
uint8_t *io_alloc(size_t bytes) {
	const size_t al = (size_t)2 << 20, len = (bytes + al - 1) / al * al + al;
	uint8_t *m = (uint8_t *)mmap(NULL, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0), *p;
	if (m == MAP_FAILED) mDie("Out of memory");
	p = (uint8_t *)(((uintptr_t)m + al - 1) / al * al);
#ifdef MADV_HUGEPAGE
	(void)madvise(p, len - (size_t)(p - m), MADV_HUGEPAGE);
#endif
	return p;                              /* (never unmapped: the buffers live as long as the process) */
}
void io_populate(uint8_t *p, size_t bytes) {
	if (getenv("MSX_NO_POPULATE")) return;
	(void)madvise(p, bytes, MADV_POPULATE_WRITE);      /* (EINVAL on kernels before 5.14: page-locking faults the pages in then) */
}

/* Page-locking a slot's byte buffer and allocating its page-locked output buffer takes tens of milliseconds per
 * slot -- on a thread of its own, in the order the decode stage will use the slots for raw batches (1, 2, ..., 0), so
 * that neither the first batch nor HIP start-up waits for it; the device thread waits for the one slot it is about to use. */
static void *pin_thread(void *arg) {
	struct pin_arg_s *A = (struct pin_arg_s *)arg;
	pipe_t *P = (pipe_t *)A->P;
	int k;
	int j;
	/* phase 1, no HIP involved (runs from pipe_enable_raw on, beside HIP start-up): the pages */
	for (k = 1 + A->first, j = A->first; k <= P->n_slots || j < PIPE_OBUFS; k += A->step, j += A->step) {
		if (k <= P->n_slots) io_populate(P->slot[k % P->n_slots].rbuf, P->slot[k % P->n_slots].rcap);
		if (P->with_obuf && j < PIPE_OBUFS) io_populate(P->ob[j], P->ob_cap[j]);
	}
	/* phase 2, once a context exists: page-lock them, in the order they will be needed */
	pthread_mutex_lock(&P->pin_mu);
	while (!P->pin_ctx && !P->pin_quit) pthread_cond_wait(&P->pin_cv, &P->pin_mu);
	pthread_mutex_unlock(&P->pin_mu);
	if (!P->pin_ctx) return NULL;
	g_ctx = P->pin_ctx;
	for (k = 1 + A->first, j = A->first; k <= P->n_slots || j < PIPE_OBUFS; k += A->step, j += A->step) {
		if (k <= P->n_slots) {
			pslot *s = &P->slot[k % P->n_slots];
			if (!getenv("MSX_NO_PIN")) MSX(msx_host_register(g_ctx, s->rbuf, s->rcap));
			pthread_mutex_lock(&P->pin_mu);
			s->pin_ready = 1;
			pthread_cond_broadcast(&P->pin_cv);
			pthread_mutex_unlock(&P->pin_mu);
		}
		if (P->with_obuf && j < PIPE_OBUFS) {
			if (!getenv("MSX_NO_PIN")) MSX(msx_host_register(g_ctx, P->ob[j], P->ob_cap[j]));
			ob_release(P, j);
		}
	}
	/* phase 3 (filter -b over a long input: comp_ramp): once the command is under way, a larger buffer for every slot --
	 * allocated, faulted in and page-locked here, taken by the decode stage the next time it holds the slot (pipe_fill).  A
	 * command that ends before that never pays for them; one whose buffers are late goes on with the small batches. */
	if (P->comp_ramp) {
		for (;;) {
			struct timespec ts = {0, 1000000};
			if (__atomic_load_n(&P->pin_quit, __ATOMIC_ACQUIRE)) return NULL;
			if (__atomic_load_n(&P->n_filled, __ATOMIC_ACQUIRE) >= (size_t)P->big_from) break;
			nanosleep(&ts, NULL);
		}
		for (k = A->first; k < P->n_slots; k += A->step) {
			uint8_t *b;
			if (__atomic_load_n(&P->pin_quit, __ATOMIC_ACQUIRE)) return NULL;
			b = io_alloc(P->big_rcap);
			io_populate(b, P->big_rcap);
			if (!getenv("MSX_NO_PIN")) MSX(msx_host_register(g_ctx, b, P->big_rcap));
			__atomic_store_n(&P->big_rbuf[k], b, __ATOMIC_RELEASE);
		}
	}
	return NULL;
}
/* the threads are started by pipe_enable_raw (they populate the buffers); the device thread hands them its context here */
void pin_start(pipe_t *P, int with_obuf) {
	(void)with_obuf;
	if (!P->pin_started || P->pin_ctx) return;
	pthread_mutex_lock(&P->pin_mu);
	if (!P->pin_ctx) {                               /* (the first context to come, of several) */
		P->pin_ctx = g_ctx;
		pthread_cond_broadcast(&P->pin_cv);
	}
	pthread_mutex_unlock(&P->pin_mu);
}
static void pin_spawn(pipe_t *P) {
	int t;
	P->pin_started = 1;
	P->n_pin = getenv("MSX_PIN_THREADS") ? atoi(getenv("MSX_PIN_THREADS")) : 2;
	if (P->n_pin < 1) P->n_pin = 1;
	if (P->n_pin > P->n_slots) P->n_pin = P->n_slots;
	pthread_mutex_init(&P->pin_mu, NULL);
	pthread_cond_init(&P->pin_cv, NULL);
	for (t = 0; t < P->n_pin; t++) {
		P->pin_args[t].P = P; P->pin_args[t].first = t; P->pin_args[t].step = P->n_pin;
		if (pthread_create(&P->pin_th[t], NULL, pin_thread, &P->pin_args[t]) != 0) mDie("pthread_create failed");
	}
}
void pin_join(pipe_t *P) {
	int t;
	if (!P->pin_started) return;
	pthread_mutex_lock(&P->pin_mu);
	/* (every device thread calls this; a thread is joined once -- four contexts joining the same threads hung one run in 48) */
	while (P->pin_joined == 1) pthread_cond_wait(&P->pin_cv, &P->pin_mu);
	if (P->pin_joined == 2) { pthread_mutex_unlock(&P->pin_mu); return; }
	P->pin_joined = 1;
	P->pin_quit = 1;                                  /* (threads that were never given a context) */
	pthread_cond_broadcast(&P->pin_cv);
	pthread_mutex_unlock(&P->pin_mu);
	for (t = 0; t < P->n_pin; t++) pthread_join(P->pin_th[t], NULL);
	pthread_mutex_lock(&P->pin_mu);
	P->pin_joined = 2;
	pthread_cond_broadcast(&P->pin_cv);
	pthread_mutex_unlock(&P->pin_mu);
}
/* MSX_TRACE=1: one line per hand-over between the pipeline's threads (who waits for what), for hangs */
int msh_trace_on(void) {
	static int on = -1;          /* (every thread would store the same value: relaxed atomics keep the sanitizer's books straight) */
	int v = __atomic_load_n(&on, __ATOMIC_RELAXED);
	if (v < 0) { v = getenv("MSX_TRACE") != NULL; __atomic_store_n(&on, v, __ATOMIC_RELAXED); }
	return v;
}
#define TRACE(...) do { if (msh_trace_on()) { fprintf(stderr, "# trace %.3f: ", now_s()); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); } } while (0)
int ob_acquire(pipe_t *P, size_t seq) {
	const int i = (int)(seq % PIPE_OBUFS);
	TRACE("batch %zu wants output buffer %d (state %d, %zu batches written)", seq, i, P->ob_state[i], P->ob_written);
	pthread_mutex_lock(&P->ob_mu);
	/* only the PIPE_OBUFS batches the writer will take next may hold a buffer: each of them has a buffer number of its
	 * own, so the batch the writer is waiting for always finds its buffer free (a batch further behind that came first
	 * would otherwise sit on it: seen as a hang with three contexts) */
	while (P->ob_state[i] != 1 || seq >= P->ob_written + PIPE_OBUFS) pthread_cond_wait(&P->ob_cv, &P->ob_mu);
	P->ob_state[i] = 2;
	pthread_mutex_unlock(&P->ob_mu);
	return i;
}
void ob_release(pipe_t *P, int i) {
	TRACE("output buffer %d released", i);
	pthread_mutex_lock(&P->ob_mu);
	P->ob_state[i] = 1;
	pthread_cond_broadcast(&P->ob_cv);
	pthread_mutex_unlock(&P->ob_mu);
}
/* the writer is done with batch seq (whether it held a buffer or not) */
void ob_written(pipe_t *P, size_t seq) {
	pthread_mutex_lock(&P->ob_mu);
	P->ob_written = seq + 1;
	pthread_cond_broadcast(&P->ob_cv);
	pthread_mutex_unlock(&P->ob_mu);
}
void pin_wait(pipe_t *P, pslot *s) {
	pthread_mutex_lock(&P->pin_mu);
	while (!s->pin_ready) pthread_cond_wait(&P->pin_cv, &P->pin_mu);
	pthread_mutex_unlock(&P->pin_mu);
}

/* a raw slot's bytes to the device and the record walk over them.  Compressed slots are inflated there; a batch with a
 * block the device inflater refuses is inflated here instead -- by the reader's own decoder and zlib, whose diagnostics
 * are the command's -- and handed over inflated. */
void unpack_slot_enqueue(pipe_t *P, pslot *s, msx_unpack *unpack, const msx_unpack_params *up) {
	if (P->n_consumers > 1) {
		/* Several contexts share the one stream: this batch's blocks go up and are inflated at once (a stream of their
		 * own), its walk waits for the carry of the batch before it -- which another context may still be walking. */
		if (s->comp && s->n_blk > 0) MSX(msx_unpack_prefetch_bgzf(g_ctx, unpack, s->rbuf, s->rlen, s->blk, s->n_blk));
		TRACE("batch %zu waits for the carry (baton at %zu)", s->seq, P->baton_seq);
		pthread_mutex_lock(&P->baton_mu);
		while (P->baton_seq != s->seq) pthread_cond_wait(&P->baton_cv, &P->baton_mu);
		TRACE("batch %zu has the carry (%zu bytes, fresh %d)", s->seq, P->baton.l, P->baton_fresh);
		if (P->baton_fresh) MSX(msx_unpack_seed(g_ctx, unpack, (const uint8_t *)P->baton.s, P->baton.l, P->baton_has_name ? P->baton_name : NULL));
		pthread_mutex_unlock(&P->baton_mu);
	}
	if (!s->comp) MSX(msx_unpack_enqueue(g_ctx, unpack, s->rbuf, s->rlen, up));
	else MSX(msx_unpack_enqueue_bgzf(g_ctx, unpack, s->rbuf, s->rlen, s->blk, s->n_blk, up));
}
/* (several contexts) the walk of batch s is done: its carry to whoever walks the next batch */
static void baton_pass(pipe_t *P, pslot *s, msx_unpack *unpack) {
	size_t n = 0;
	MSX(msx_unpack_carry(g_ctx, unpack, NULL, 0, &n, NULL, NULL));
	pthread_mutex_lock(&P->baton_mu);
	P->baton.l = 0;
	ks_reserve(&P->baton, n + 1);
	MSX(msx_unpack_carry(g_ctx, unpack, (uint8_t *)P->baton.s, n, &n, P->baton_name, &P->baton_has_name));
	P->baton.l = n;
	P->baton_seq = s->seq + 1;
	P->baton_fresh = 1;
	TRACE("batch %zu passes the carry on (%zu bytes)", s->seq, n);
	pthread_cond_broadcast(&P->baton_cv);
	pthread_mutex_unlock(&P->baton_mu);
}
void unpack_slots_ahead(pipe_t *P, msx_unpack *unpack, ahead_q *A) {
	static int depth = 0;
	if (!depth) { const char *e = getenv("MSX_INFLATE_AHEAD"); depth = e && atoi(e) == 2 ? 2 : 1; }
	if (!P->comp_mode || getenv("MSX_NO_INFLATE_AHEAD")) return;
	while (A->n < depth && !A->closed) {
		const int nx = pq_try_pop(&P->q_dev);
		if (nx == PQ_NONE) return;
		A->item[A->n++] = nx;
		if (nx >= 0 && P->slot[nx].raw && P->slot[nx].comp && P->slot[nx].n_blk > 0) {
			pin_wait(P, &P->slot[nx]);
			__atomic_add_fetch(&P->n_ahead, 1, __ATOMIC_RELAXED);
			MSX(msx_unpack_prefetch_bgzf(g_ctx, unpack, P->slot[nx].rbuf, P->slot[nx].rlen, P->slot[nx].blk, P->slot[nx].n_blk));
		} else {
			A->closed = 1;               /* (what follows it must not overtake it on the device) */
		}
	}
}
int ahead_pop(ahead_q *A) {
	int v;
	if (A->n == 0) return PQ_NONE;
	v = A->item[0];
	A->item[0] = A->item[1];
	if (--A->n == 0) A->closed = 0;
	return v;
}
void unpack_slot_finish(pipe_t *P, pslot *s, msx_unpack *unpack, const msx_unpack_params *up, msx_unpack_result *ur, msx_batch *db) {
	int rc = msx_unpack_finish(g_ctx, unpack, ur, db);
	if (s->comp) __atomic_add_fetch(&P->n_comp_done, 1, __ATOMIC_RELAXED);
	if (rc == MSX_ERR_INFLATE && s->comp) {
		static __thread uint8_t *fb = NULL;
		static __thread size_t fb_cap = 0;
		if (fb_cap < s->inflated + 64) { fb_cap = s->inflated + 64; fb = (uint8_t *)realloc(fb, fb_cap); if (!fb) mDie("Out of memory"); }
		msh_inflate_table(s->rbuf, s->blk, s->n_blk, fb);
		__atomic_add_fetch(&P->n_host_inflated, 1, __ATOMIC_RELAXED);
		MSX(msx_unpack_enqueue(g_ctx, unpack, fb, s->inflated, up));
		rc = msx_unpack_finish(g_ctx, unpack, ur, db);
	}
	if (rc != MSX_OK) mDie("%s", msx_last_error(g_ctx));
	if (P->n_consumers > 1) baton_pass(P, s, unpack);
}

/* device unpack: every slot gets a buffer of fixed size for the inflated bytes (page-locked by pin_thread) */
void pipe_enable_raw(pipe_t *P, int with_obuf) {
	int i;
	P->raw_mode = 1;
	P->with_obuf = with_obuf;
	/* BAM input: the blocks stay compressed until they are on the device (MSX_HOST_INFLATE=1: inflate here).  A batch is
	 * as many blocks as the device inflates at a time -- one wave per block, eight per compute unit (msx_inflate.hip) --
	 * or what fits the slot's buffer, whichever comes first. */
	P->comp_mode = msh_is_bam(P->in) && !getenv("MSX_HOST_INFLATE");
	/* (MSX_BATCH_BYTES, the inflated size of a batch, translates into blocks) */
	/* filter on one device (with_obuf): 4096 blocks once the command is under way -- batch after batch pays the same four
	 * dozen small launches of the record walk and the filter, each slowed to ~50 us beside the inflater's and the encoder's
	 * resident waves, so a batch of twice the records costs the device stage little more (the 400 M-record file: 1.28 ->
	 * 1.19 s) -- and 2048 for the first eight, which decide when the first output leaves, and for as long as the larger
	 * buffers are not ready (pin_thread, phase 3: a 100 M-record file paid 20-30 ms for buffers of that size allocated at
	 * start-up).  -bu stays at 2048 (its output buffers would have to grow with the batches); profile and coverage too:
	 * they lost a little with larger ones (profiles/round6/batch_geometry.log). */
	/* MSX_COMP_RAMP_FROM=<batch>: the batch from which on twice MSX_COMP_BLOCKS are taken (0: never), whatever the other
	 * settings say (tests: a ramp within a small file; the larger buffers are then made at once, twice MSX_COMP_BYTES) */
	P->comp_ramp = 0;
	if (with_obuf == 2 && P->n_consumers == 1) {
		const char *re = getenv("MSX_COMP_RAMP_FROM");
		if (re) P->comp_ramp = atoi(re) > 0 ? atoi(re) : 0;
		else if (!getenv("MSX_COMP_BLOCKS") && !getenv("MSX_BATCH_BYTES") && !getenv("MSX_COMP_BYTES")) P->comp_ramp = 8;
	}
	P->comp_blocks = (int)env_size("MSX_COMP_BLOCKS", getenv("MSX_BATCH_BYTES") ? P->batch_bytes_cfg / 65280 : 2048);
	P->big_rcap = getenv("MSX_COMP_BYTES") ? 2 * env_size("MSX_COMP_BYTES", (size_t)40 << 20) : (size_t)64 << 20;
	if (P->big_rcap < ((size_t)4 << 20)) P->big_rcap = (size_t)4 << 20;
	/* when the larger buffers are made: at once for a file of 3 GB and more; otherwise when 32 batches have shown that the
	 * input is long (a 100 M-record file -- 13 batches -- lost up to 0.1 s to half a gigabyte page-locked beside its last batches) */
	P->big_from = (getenv("MSX_COMP_RAMP_FROM") || msh_in_bytes(P->in) >= ((int64_t)3 << 30)) ? (P->comp_ramp < 4 ? 1 : 4) : 32;
	if (P->comp_blocks < 1) P->comp_blocks = 1;
	if (P->comp_blocks > (1 << 16)) P->comp_blocks = 1 << 16;
	/* compressed batches need nothing of a slot but its buffer of payloads: two more of them (batch 0, walked on the host,
	 * has taken its slot -- one of the first -- by the time these are used) */
	if (P->comp_mode && !getenv("MSX_SLOTS") && P->n_consumers == 1)
		while (P->n_slots < PIPE_SLOTS_COMP && P->n_slots < PIPE_SLOTS_MAX) pq_push(&P->q_free, P->n_slots++);
	for (i = 0; i < P->n_slots; i++) {
		pslot *s = &P->slot[i];
		if (P->comp_mode) {
			s->rcap = env_size("MSX_COMP_BYTES", (size_t)40 << 20);
			if (s->rcap < ((size_t)2 << 20)) s->rcap = (size_t)2 << 20;
			s->blk = (msx_bgzf_block *)xmalloc((size_t)P->comp_blocks * (P->comp_ramp ? 2 : 1) * sizeof(msx_bgzf_block));
			/* (the output buffer starts at half of what the blocks inflate to -- page-locking is paid per byte, at start-up --
			 * and is replaced by a larger one when a batch keeps more: filter_dev_thread) */
			P->ocap_cfg = (size_t)P->comp_blocks * 32768 + ((size_t)8 << 20);
		} else {
			s->rcap = P->batch_bytes_cfg + BGZF_INFLATE_MAX + 4096;
			P->ocap_cfg = s->rcap;
		}
		s->rbuf = io_alloc(s->rcap);
	}
	for (i = 0; with_obuf && i < PIPE_OBUFS; i++) {
		P->ob_cap[i] = P->ocap_cfg;
		P->ob[i] = io_alloc(P->ob_cap[i]);
	}
	pin_spawn(P);
}

/* ---- record boundaries: a speculative parallel chase -------------------------------------------
 * The block_size chain is serial by nature.  Here every worker guesses a record start near the
 * beginning of its segment (a header that looks like one, followed by two more that do) and walks
 * its segment from there; the segments are then stitched in order: where a worker's first offset
 * is not the true one the stitcher walks on by itself until both chains meet (from any true start
 * the chain is the true chain).  Guesses only decide how much of the walk ran in parallel. */
#ifdef MSX_DEBUG_SWITCHES
static int chase_sloppy = -1;      /* MSX_CHASE_SLOPPY=1 (msamtools-dbg only; tests): accept almost anything as a record start, so
                                      that most guesses are wrong and the stitcher has to repair them */
#endif
static int rec_plausible(const uint8_t *u, size_t off, size_t len, int32_t nt) {
	const uint8_t *r;
	int32_t bs, tid, pos, mtid, mpos, ls;
	uint32_t lq, nc, k;
	if (off + 36 > len) return 0;
	bs = le32(u + off);
	if (bs < 32 || bs > (64 << 20)) return 0;
#ifdef MSX_DEBUG_SWITCHES
	{
		int sl = __atomic_load_n(&chase_sloppy, __ATOMIC_RELAXED);      /* (every thread would compute the same value) */
		if (sl < 0) { sl = getenv("MSX_CHASE_SLOPPY") != NULL; __atomic_store_n(&chase_sloppy, sl, __ATOMIC_RELAXED); }
		if (sl) return bs < 4096;
	}
#endif
	r = u + off + 4;
	tid = REC_TID(r); pos = REC_POS(r); mtid = le32(r + 20); mpos = le32(r + 24);
	if (tid < -1 || tid >= nt || mtid < -1 || mtid >= nt || pos < -1 || mpos < -1) return 0;
	lq = REC_LQNAME(r); nc = REC_NCIGAR(r); ls = REC_LSEQ(r);
	if (lq < 1 || ls < 0) return 0;
	if (32ull + lq + 4ull * nc + ((uint64_t)ls + 1) / 2 + (uint64_t)ls > (uint64_t)bs) return 0;
	if (off + 4 + 32 + lq > len) return 1;
	if (r[32 + lq - 1] != 0) return 0;
	for (k = 0; k + 1 < lq; k++)
		if (r[32 + k] < 33 || r[32 + k] > 126) return 0;
	return 1;
}

typedef struct {
	pipe_t *P;
	const uint8_t *u;
	size_t len;
	int nseg;
} chase_job;

static void chase_worker(void *arg, int k, int nth) {
	chase_job *J = (chase_job *)arg;
	pipe_t *P = J->P;
	const uint8_t *u = J->u;
	const size_t len = J->len, lo = len * (size_t)k / (size_t)J->nseg, hi = len * (size_t)(k + 1) / (size_t)J->nseg;
	size_t off = lo, n = 0, *list = P->seg_list[k];
	(void)nth;
	if (k > 0) {
		const int32_t nt = P->hdr->n_targets;
		const size_t lim = lo + ((size_t)1 << 20) < hi ? lo + ((size_t)1 << 20) : hi;
		int found = 0;
		for (; off < lim; off++) {
			size_t o2, o3;
			if (!rec_plausible(u, off, len, nt)) continue;
			o2 = off + 4 + (size_t)le32(u + off);
			if (o2 + 36 <= len) {
				if (!rec_plausible(u, o2, len, nt)) continue;
				o3 = o2 + 4 + (size_t)le32(u + o2);
				if (o3 + 36 <= len && !rec_plausible(u, o3, len, nt)) continue;
			}
			found = 1;
			break;
		}
		if (!found) { P->seg_cnt[k] = 0; P->seg_end[k] = (size_t)-1; return; }
	}
	while (off < hi && off + 4 <= len) {
		const int32_t bs = le32(u + off);
		if (bs < 32 || off + 4 + (size_t)bs > len) break;      /* a cut record, or not a record at all (wrong guess) */
		list[n++] = off;
		off += 4 + (size_t)bs;
	}
	P->seg_cnt[k] = n;
	P->seg_end[k] = off;
}

typedef struct {
	pipe_t *P;
	rbatch *b;
	size_t *base;          /* first output index of every segment's part */
	int nseg;
} chase_copy_job;

static void chase_copy_worker(void *arg, int k, int nth) {
	chase_copy_job *J = (chase_copy_job *)arg;
	pipe_t *P = J->P;
	size_t o = J->base[k];
	(void)nth;
	if (P->seg_extra_n[k]) { memcpy(J->b->rec_off + o, P->seg_extra[k], P->seg_extra_n[k] * sizeof(size_t)); o += P->seg_extra_n[k]; }
	if (P->seg_cnt[k] > P->seg_from[k])
		memcpy(J->b->rec_off + o, P->seg_list[k] + P->seg_from[k], (P->seg_cnt[k] - P->seg_from[k]) * sizeof(size_t));
}

/* offsets of the complete records of u[0, len) into b->rec_off (at most max_rec of them);
 * returns their number, *tail = offset of the first byte not covered by them */
static size_t chase_records(pipe_t *P, rbatch *b, const uint8_t *u, size_t len, size_t max_rec, size_t *tail) {
	int nseg = msh_threads(), k;
	chase_job J;
	chase_copy_job C;
	size_t cur = 0, total = 0, *base;
	if ((size_t)nseg > len / ((size_t)1 << 20) + 1) nseg = (int)(len / ((size_t)1 << 20) + 1);
	if (nseg > P->nseg_cap) {
		P->seg_list = (size_t **)realloc(P->seg_list, sizeof(size_t *) * (size_t)nseg);
		P->seg_extra = (size_t **)realloc(P->seg_extra, sizeof(size_t *) * (size_t)nseg);
		for (k = P->nseg_cap; k < nseg; k++) { P->seg_list[k] = NULL; P->seg_extra[k] = NULL; }
		P->seg_cnt = (size_t *)realloc(P->seg_cnt, sizeof(size_t) * (size_t)nseg);
		P->seg_end = (size_t *)realloc(P->seg_end, sizeof(size_t) * (size_t)nseg);
		P->seg_from = (size_t *)realloc(P->seg_from, sizeof(size_t) * (size_t)nseg);
		P->seg_extra_n = (size_t *)realloc(P->seg_extra_n, sizeof(size_t) * (size_t)nseg);
		P->nseg_cap = nseg;
	}
	for (k = 0; k < nseg; k++) {
		const size_t seg = len / (size_t)nseg + 2;
		P->seg_list[k] = (size_t *)realloc(P->seg_list[k], (seg / 36 + 4) * sizeof(size_t));
		P->seg_extra[k] = (size_t *)realloc(P->seg_extra[k], (seg / 36 + 4) * sizeof(size_t));
		if (!P->seg_list[k] || !P->seg_extra[k]) mDie("Out of memory");
	}
	J.P = P; J.u = u; J.len = len; J.nseg = nseg;
	msh_parallel(nseg, chase_worker, &J);
	/* stitch */
	base = (size_t *)xmalloc(sizeof(size_t) * (size_t)(nseg + 1));
	for (k = 0; k < nseg; k++) {
		const size_t hi = len * (size_t)(k + 1) / (size_t)nseg;
		const size_t *list = P->seg_list[k];
		const size_t cnt = P->seg_cnt[k];
		size_t j = 0, ne = 0;
		int joined = 0;
		base[k] = total;
		P->seg_extra_n[k] = 0;
		P->seg_from[k] = cnt;
		if (cur < hi) {
			while (j < cnt && list[j] < cur) j++;
			if (j < cnt && list[j] == cur) {
				joined = 1;
			} else {
				/* the guess was off: walk from the true position until the chains meet */
				while (cur < hi && cur + 4 <= len) {
					const int32_t bs = le32(u + cur);
					if (bs < 32) mDie("Corrupt BAM record (block_size %d)", bs);
					if (cur + 4 + (size_t)bs > len) break;
					P->seg_extra[k][ne++] = cur;
					cur += 4 + (size_t)bs;
					while (j < cnt && list[j] < cur) j++;
					if (j < cnt && list[j] == cur) { joined = 1; break; }
				}
			}
			P->seg_extra_n[k] = ne;
			if (joined) { P->seg_from[k] = j; cur = P->seg_end[k]; }
		}
		total += ne + (P->seg_cnt[k] - P->seg_from[k]);
	}
	base[nseg] = total;
	if (cur + 4 <= len) {                 /* the chain stopped inside the buffer: a cut record, or garbage */
		const int32_t bs = le32(u + cur);
		if (bs < 32) mDie("Corrupt BAM record (block_size %d)", bs);
	}
	*tail = cur;
	if (total > max_rec) {
		/* more records than a slot holds: keep the first max_rec (rare: only with very short records) */
		size_t keep = max_rec, kk;
		for (k = 0; k < nseg; k++) {
			const size_t have = base[k + 1] - base[k];
			if (base[k] >= keep) { P->seg_extra_n[k] = 0; P->seg_from[k] = P->seg_cnt[k]; continue; }
			if (base[k] + have <= keep) continue;
			kk = keep - base[k];               /* entries of this segment to keep */
			if (kk <= P->seg_extra_n[k]) {
				*tail = P->seg_extra[k][kk];     /* (kk < extra_n, or the first list entry follows) */
				if (kk == P->seg_extra_n[k]) *tail = P->seg_list[k][P->seg_from[k]];
				P->seg_extra_n[k] = kk; P->seg_cnt[k] = P->seg_from[k];
			} else {
				const size_t jj = P->seg_from[k] + (kk - P->seg_extra_n[k]);
				*tail = P->seg_list[k][jj];
				P->seg_cnt[k] = jj;
			}
		}
		total = keep;
	}
	C.P = P; C.b = b; C.base = base; C.nseg = nseg;
	msh_parallel(nseg, chase_copy_worker, &C);
	b->rec_off[total] = *tail;
	free(base);
	return total;
}

/* counts -> offsets (cigar_off, md_off) and pool starts (group_off), two passes over per-thread ranges */
typedef struct {
	rbatch *b;
	size_t n;
	int mode, pass;
	uint64_t sum_c[MSH_POOL_MAX], sum_m[MSH_POOL_MAX], sum_g[MSH_POOL_MAX];
} offs_job;

static void offs_worker(void *arg, int tid, int nth) {
	offs_job *O = (offs_job *)arg;
	rbatch *b = O->b;
	const size_t lo = O->n * (size_t)tid / (size_t)nth, hi = O->n * (size_t)(tid + 1) / (size_t)nth;
	size_t i;
	if (O->pass == 0) {
		uint64_t c = 0, m = 0, g = 0;
		for (i = lo; i < hi; i++) {
			c += b->cigar_off[i + 1];
			m += b->md_off[i + 1];
			g += (O->mode != 0 && (i == 0 || b->bound[i]));
		}
		O->sum_c[tid] = c; O->sum_m[tid] = m; O->sum_g[tid] = g;
	} else {
		uint64_t c = O->sum_c[tid], m = O->sum_m[tid], g = O->sum_g[tid];
		for (i = lo; i < hi; i++) {
			if (O->mode != 0 && (i == 0 || b->bound[i])) b->group_off[g++] = (uint32_t)i;
			c += b->cigar_off[i + 1];
			m += b->md_off[i + 1];
			b->cigar_off[i + 1] = (uint32_t)c;
			b->md_off[i + 1] = (uint32_t)m;
		}
	}
}

/* one batch into slot s: returns the number of records (0 = end of stream) */
static size_t pipe_fill(pipe_t *P, pslot *s) {
	rbatch *b = &s->b;
	size_t n = 0, tail = 0, n_batch, i;
	int must_read = 0;      /* what is here is not a batch yet (no whole record, one pool only): the next bytes are waited for */
	pack_job J;
	s->ulen = 0;
	/* SAM text: the slot's buffer at the size a batch will take, in one piece and before anything is written to it -- it used to
	 * grow by halves under the parser (16 MB -> 182 MB in seven steps, each a move of the mapping): every page of it a 4 KB page,
	 * 1.2 GB of them over the slots, and their tear-down 0.3 s of the command's exit */
	if (!msh_is_bam(P->in) && s->ucap < P->carry.l + P->batch_bytes + ((size_t)40 << 20)) {
		const size_t nc = P->carry.l + P->batch_bytes + ((size_t)40 << 20);
		uint8_t *nb = (uint8_t *)malloc(nc);
		if (!nb) mDie("Out of memory");
		msh_huge_hint(nb, nc);
		free(s->ubuf);
		s->ubuf = nb;
		s->ucap = nc;
	}
	if (P->carry.l) {
		if (P->carry.l + 64 > s->ucap) { s->ucap = P->carry.l + P->batch_bytes + 64; s->ubuf = (uint8_t *)realloc(s->ubuf, s->ucap); if (!s->ubuf) mDie("Out of memory"); msh_huge_hint(s->ubuf, s->ucap); }
		memcpy(s->ubuf, P->carry.s, P->carry.l);
		s->ulen = P->carry.l;
		P->carry.l = 0;
	}
	/* device unpack: batch 0 -- the one batch walked on the host, for the preflight -- is kept to the preflight window
	 * (it is decoded, uploaded from pageable memory and filtered while everything else waits for it) */
	if (P->raw_mode && !P->have_first && P->batch_bytes == P->batch_bytes_cfg) {
		P->batch_bytes = (size_t)12 << 20;
		if (P->batch_bytes > P->batch_bytes_cfg) P->batch_bytes = P->batch_bytes_cfg;
		msh_inflate_limit(192);
	}
	for (;;) {
		size_t want = P->batch_bytes;
		double tq = now_s(), tq2;
		/* a batch also ends where the producer goes quiet: the reference writes pool by pool (msam_filter.c:120-125,186).  (The
		 * first batch still holds the preflight's window -- 100 000 records, as the reference's own look-ahead does,
		 * msam_helper.c:295-484: what is here when the producer pauses is looked at, and if it is less, waited for) */
		while (!P->in_eof && s->ulen < want) {
			if (!must_read && s->ulen > 0 && msh_idle_ms() > 0 && !msh_input_ready(P->in, msh_idle_ms())) break;
			must_read = 0;
			if (!pipe_append(P->in, &s->ubuf, &s->ulen, &s->ucap)) P->in_eof = 1;
		}
		tq2 = now_s(); P->t_inflate += tq2 - tq; tq = tq2;
		if (s->ulen == 0) return 0;
		n = chase_records(P, b, s->ubuf, s->ulen, P->cap_rec, &tail);
		tq2 = now_s(); P->t_chase += tq2 - tq; tq = tq2;
		if (P->in_eof && n < P->cap_rec && tail != s->ulen) mDie("Truncated BAM record");
		if (!P->have_first && n < COORD_ORDER_CHECK_RECORDS && !P->in_eof) {   /* the preflight window (msam_helper.c:4-6) */
			if (s->ulen >= want) P->batch_bytes += P->batch_bytes;
			must_read = 1;
			continue;
		}
		if (n == 0) {
			if (P->in_eof) return 0;
			if (s->ulen >= want) P->batch_bytes += P->batch_bytes;            /* a record larger than the batch: read on */
			must_read = 1;
			continue;
		}
		/* aux scan, SoA scalars, pool boundaries */
		J.b = b; J.base = s->ubuf; J.n = n; J.mode = P->mode; J.want_stats = P->want_stats; J.unmapped_visible = P->unmapped_visible;
		J.carry_name = P->have_prev ? P->prev_read : NULL;
		msh_parallel(msh_threads(), pack_scan, &J);
		P->t_scan += now_s() - tq;
		n_batch = n;
		if (P->mode != 0 && !(P->in_eof && tail == s->ulen && n < P->cap_rec)) {
			size_t k = n;
			while (k > 1 && !b->bound[k - 1]) k--;
			n_batch = k - 1;
			if (P->cut_mapped && n_batch > 0) {
				/* filter | profile in one process: a pool that begins with an unmapped record belongs to the insert of
				 * the pool before it (msx_batch.pool_rule), so the batch should end in front of a pool that begins with
				 * a mapped one; looked for in the batch's second half (a long tail of unmapped records must not make
				 * the batch grow without bound -- cut inside it, only a QNAME that reappears behind it could notice) */
				size_t q = k;
				while (q - 1 > n / 2 && !(b->bound[q - 1] && !(b->flag[q - 1] & 4))) q--;
				if (q - 1 > n / 2) n_batch = q - 1;
			}
			if (n_batch == 0) {          /* one pool fills the whole batch: take more bytes */
				if (P->in_eof && tail == s->ulen) { n_batch = n; break; }
				if (n >= P->cap_rec) mDie("A single QNAME group exceeds the batch capacity (%zu records); raise MSX_BATCH_RECORDS", P->cap_rec);
				if (s->ulen >= want) P->batch_bytes += P->batch_bytes / 2;
				must_read = 1;
				continue;
			}
		}
		break;
	}
	/* offsets and pools: per-thread sums, a short serial pass over the threads, per-thread fill */
	{ double tser = now_s();
	{
		offs_job O;
		int nth = msh_threads(), t;
		size_t cut = n_batch;
		if ((size_t)nth > n_batch / 65536 + 1) nth = (int)(n_batch / 65536 + 1);
		O.b = b; O.n = n_batch; O.mode = P->mode; O.pass = 0;
		msh_parallel(nth, offs_worker, &O);
		{
			uint64_t c = 0, m = 0, g = 0;
			for (t = 0; t < nth; t++) {
				uint64_t tc = O.sum_c[t], tm = O.sum_m[t], tg = O.sum_g[t];
				O.sum_c[t] = c; O.sum_m[t] = m; O.sum_g[t] = g;
				c += tc; m += tm; g += tg;
			}
			if (c > 0xfffffff0ull || m > 0xfffffff0ull) mDie("CIGAR/MD payload of a batch exceeds 2^32 bytes; lower MSX_BATCH_BYTES");
			b->n_groups = (size_t)g;
		}
		O.pass = 1;
		msh_parallel(nth, offs_worker, &O);
		b->cigar_off[0] = 0;
		b->md_off[0] = 0;
		/* cut where the payload arrays are full (rare): at the last pool boundary that still fits */
		if (b->cigar_off[n_batch] + 4 > b->cigar_cap || b->md_off[n_batch] + 16 > b->md_cap) {
			size_t k = n_batch;
			while (k > 0 && (b->cigar_off[k] + 4 > b->cigar_cap || b->md_off[k] + 16 > b->md_cap)) k--;
			if (P->mode != 0) { while (k > 1 && !b->bound[k]) k--; if (!b->bound[k]) k = 0; }
			if (k == 0) mDie("CIGAR/MD payload of one QNAME group exceeds the batch capacity; raise MSX_BATCH_RECORDS");
			cut = k;
			if (P->mode != 0) { size_t gq = 0, q; for (q = 0; q < cut; q++) gq += (q == 0 || b->bound[q]); b->n_groups = gq; }
		}
		n_batch = cut;
	}
	P->t_serial += now_s() - tser; tser = now_s();
	if (P->want_stats) {
		J.n = n_batch;
		msh_parallel(msh_threads(), pack_copy, &J);
	}
	P->t_copy += now_s() - tser; }
	b->n = n_batch;
	b->base = s->ubuf;
	P->have_first = 1;
	msh_inflate_limit(0);
	for (i = n_batch; i > 0; i--) {       /* grouping state for the next batch */
		const uint8_t *r = s->ubuf + b->rec_off[i - 1] + 4;
		if (P->mode == 0 || rec_names_pool(r, P->mode, P->unmapped_visible)) {
			strcpy(P->prev_read, REC_QNAME(r));
			P->have_prev = 1;
			break;
		}
	}
	/* what lies behind the batch goes to the next one */
	P->carry.l = 0;
	if (b->rec_off[n_batch] < s->ulen) ks_put(&P->carry, s->ubuf + b->rec_off[n_batch], s->ulen - b->rec_off[n_batch]);
	return n_batch;
}

void *pipe_decode_thread(void *arg) {
	pipe_t *P = (pipe_t *)arg;
	for (;;) {
		double t0 = now_s(), t1;
		const int si = pq_pop(&P->q_free);
		pslot *s = &P->slot[si];
		size_t n;
		t1 = now_s();
		P->t_wait_free += t1 - t0;
		s->raw = 0;
		if (P->raw_mode && P->n_filled >= 1) {
			/* device unpack: inflate only.  The first raw slot takes along what batch 0's host-side cut left over. */
			double tq = now_s();
			if (P->raw_done || (!P->raw_started && P->in_eof && P->carry.l == 0)) {
				n = 0;
			} else {
				s->raw = 1;
				s->rlen = 0;
				s->has_seed = 0;
				if (!P->raw_started) {
					P->raw_started = 1;
					s->has_seed = 1;
					s->seed.l = 0;
					if (P->carry.l) ks_put(&s->seed, P->carry.s, P->carry.l);
					P->carry.l = 0;
					s->seed_has_name = P->have_prev;
					if (P->have_prev) strcpy(s->seed_name, P->prev_read);
				}
				/* (batch 0 may have grown batch_bytes to reach the preflight window; the configured size holds from here
				 * on.  The buffer is page-locked and must not move: msh_inflate_append appends one batch of blocks at most,
				 * so there is always room for the next call) */
				/* an input whose blocks the device inflater keeps refusing (the first three batches, every one of them: an
				 * encoder whose streams it does not decode) is inflated here from then on -- in batches the slots' buffers
				 * hold -- instead of being tried on the device and inflated here batch by batch */
				if (P->comp_mode && !P->comp_given_up) {
					const size_t refused = __atomic_load_n(&P->n_host_inflated, __ATOMIC_RELAXED);
					/* (+ 1: the batch the device stage is inflating here right now has been counted as done, not yet as refused) */
					if (refused >= 3 && refused + 1 >= __atomic_load_n(&P->n_comp_done, __ATOMIC_RELAXED)) P->comp_given_up = 1;
				}
				s->comp = P->comp_mode && !P->comp_given_up;
				s->n_blk = 0;
				s->inflated = 0;
				if (P->comp_mode && P->comp_given_up) {
					/* blocks per call: what the batch size asks for, and no more than the slot's buffer holds (a small
					 * MSX_COMP_BYTES: without this the loop below never ran and the slot went out empty, for ever) */
					size_t per = P->batch_bytes_cfg / 65536 + 1 < 128 ? P->batch_bytes_cfg / 65536 + 1 : 128;
					const size_t before = s->rlen;
					if (s->rcap / 65536 < per + 2) per = s->rcap / 65536 > 2 ? s->rcap / 65536 - 2 : 1;
					msh_inflate_limit((int)per);
					while (!P->in_eof && s->rlen < P->batch_bytes_cfg && s->rlen + (per + 1) * 65536 + 64 <= s->rcap) {
						if (s->rlen > before && msh_idle_ms() > 0 && !msh_input_ready(P->in, msh_idle_ms())) break;
						if (!pipe_append(P->in, &s->rbuf, &s->rlen, &s->rcap)) P->in_eof = 1;
					}
					msh_inflate_limit(0);
					if (s->rlen == before && !P->in_eof) mDie("The batch buffers are too small for a BGZF block (MSX_COMP_BYTES)");
				} else if (P->comp_mode) {
					int cb = P->comp_blocks;
					if (P->comp_ramp && P->n_filled >= (size_t)P->comp_ramp) {
						/* (the larger buffer, if the pin thread has it ready: nothing of the device reads a free slot's old one) */
						uint8_t *big = __atomic_load_n(&P->big_rbuf[si], __ATOMIC_ACQUIRE);
						if (big && s->rbuf != big && s->rlen == 0) { s->rbuf = big; s->rcap = P->big_rcap; }
						if (s->rbuf == big) cb = 2 * P->comp_blocks;
					}
					while (!P->in_eof && s->n_blk < cb && (s->rcap - s->rlen) / (65536 + 1024) > 0) {
						if (s->n_blk > 0 && msh_idle_ms() > 0 && !msh_input_ready(P->in, msh_idle_ms())) break;      /* (the producer has gone quiet) */
						if (!msh_raw_append(P->in, s->rbuf, s->rcap, &s->rlen, s->blk, &s->n_blk, cb, &s->inflated)) P->in_eof = 1;
					}
				} else
				while (!P->in_eof && s->rlen < P->batch_bytes_cfg && s->rlen + BGZF_INFLATE_MAX + 64 <= s->rcap) {
					if (s->rlen > 0 && msh_idle_ms() > 0 && !msh_input_ready(P->in, msh_idle_ms())) break;
					if (!pipe_append(P->in, &s->rbuf, &s->rlen, &s->rcap)) P->in_eof = 1;
				}
				s->last = P->in_eof;
				if (s->last) P->raw_done = 1;
				n = 1;                       /* (a slot: possibly without bytes, its `last` flag flushes the device's carry) */
			}
			P->t_inflate += now_s() - tq;
		} else {
			n = pipe_fill(P, s);
		}
		P->t_decode += now_s() - t1;
		s->eof = n == 0;
		if (n == 0 || P->n_filled == 0) {
			/* whoever opens the output (preflight on batch 0's records, header) need not wait for the device stage */
			pthread_mutex_lock(&P->first_mu);
			if (P->first_state == 0) { P->first_state = n == 0 ? 2 : 1; P->first_slot = si; }
			pthread_cond_broadcast(&P->first_cv);
			pthread_mutex_unlock(&P->first_mu);
		}
		if (n == 0) {
			msh_release_input(P->in);
			/* end of the stream: one token per consumer (P->n_filled is final from here on) */
			int c;
			for (c = 0; c < P->n_consumers; c++) pq_push(&P->q_dev, PQ_END);
			return NULL;
		}
		s->seq = P->n_filled;
		TRACE("decode: batch %zu ready (slot %d, raw %d, comp %d, %d blocks, last %d)", s->seq, si, s->raw, s->comp, s->n_blk, s->last);
		if (s->raw && s->has_seed) {          /* the first raw batch is seeded from batch 0's leftovers, not from a baton */
			pthread_mutex_lock(&P->baton_mu);
			P->baton_seq = s->seq;
			P->baton_fresh = 0;
			pthread_cond_broadcast(&P->baton_cv);
			pthread_mutex_unlock(&P->baton_mu);
		}
		__atomic_store_n(&P->n_filled, P->n_filled + 1, __ATOMIC_RELEASE);
		pq_push(&P->q_dev, si);
	}
}

/* page-lock the slot's SoA arrays (they never move): uploads become asynchronous DMA.  Not their whole capacity -- 3 M records,
 * 173 MB a slot, of which a 96 MB batch of records with SEQ/QUAL fills a seventh: what the batch at hand needs and a quarter
 * more, locked again (larger) only when a later batch outgrows it.  (Whole capacities were 0.9 GB of 4 KB pages over the slots of
 * a SAM-text command: 20-40 ms each to lock on the device thread, 0.3 s to let go of after the command's last line.) */
void pipe_pin_slot(pipe_t *P, pslot *s) {
	rbatch *b = &s->b;
	/* (device unpack: only batch 0 takes the host-side walk -- of its arrays just what it uses is page-locked: 170 MB for
	 * one upload would cost more than the upload saves, and from pageable memory the dozen copies took 77 ms) */
	size_t c, c_cig, c_md, c_grp;
	if (getenv("MSX_NO_PIN")) return;
	if (s->pinned && (P->raw_mode || (b->n + 8 <= s->pin_rec && (P->mode == 0 || b->n_groups + 8 <= s->pin_grp) &&
	                                  (!P->want_stats || ((size_t)b->cigar_off[b->n] + 8 <= s->pin_cig && (size_t)b->md_off[b->n] + 64 <= s->pin_md)))))
		return;
	if (s->pinned) {                   /* a batch larger than any before it in this slot: let go, lock more */
		MSX(msx_host_unregister(g_ctx, b->flag));
		MSX(msx_host_unregister(g_ctx, b->rflags));
		MSX(msx_host_unregister(g_ctx, b->tid));
		if (P->mode != 2) {
			MSX(msx_host_unregister(g_ctx, b->nm));
			MSX(msx_host_unregister(g_ctx, b->as));
			MSX(msx_host_unregister(g_ctx, s->emit));
			if (s->as_out) MSX(msx_host_unregister(g_ctx, s->as_out));
		}
		if (P->want_stats) {
			MSX(msx_host_unregister(g_ctx, b->cigar_off));
			MSX(msx_host_unregister(g_ctx, b->md_off));
			MSX(msx_host_unregister(g_ctx, b->cigar));
			MSX(msx_host_unregister(g_ctx, b->md));
		}
		if (P->mode != 0) MSX(msx_host_unregister(g_ctx, b->group_off));
	}
	s->pinned = 1;
	if (P->raw_mode) {
		c = b->n + 8; c_grp = b->n_groups + 8;
		c_cig = P->want_stats ? (size_t)b->cigar_off[b->n] + 8 : 0; c_md = P->want_stats ? (size_t)b->md_off[b->n] + 64 : 0;
	} else {
		c = b->n + b->n / 4 + 65536; if (c > b->cap) c = b->cap;
		c_cig = P->want_stats ? (size_t)b->cigar_off[b->n] + (size_t)b->cigar_off[b->n] / 4 + 65536 : 0; if (c_cig > b->cigar_cap) c_cig = b->cigar_cap;
		c_md = P->want_stats ? (size_t)b->md_off[b->n] + (size_t)b->md_off[b->n] / 4 + 65536 : 0; if (c_md > b->md_cap) c_md = b->md_cap;
		c_grp = b->n_groups + b->n_groups / 4 + 65536; if (c_grp > b->group_cap) c_grp = b->group_cap;
	}
	s->pin_rec = c; s->pin_cig = c_cig; s->pin_md = c_md; s->pin_grp = c_grp;
	MSX(msx_host_register(g_ctx, b->flag, c * 2));
	MSX(msx_host_register(g_ctx, b->rflags, c));
	MSX(msx_host_register(g_ctx, b->tid, c * 4));
	if (P->mode != 2) {
		MSX(msx_host_register(g_ctx, b->nm, c * 4));
		MSX(msx_host_register(g_ctx, b->as, c * 4));
		MSX(msx_host_register(g_ctx, s->emit, c * 4));
		if (s->as_out) MSX(msx_host_register(g_ctx, s->as_out, c * 4));
	}
	if (P->want_stats) {
		MSX(msx_host_register(g_ctx, b->cigar_off, (c + 1) * 4));
		MSX(msx_host_register(g_ctx, b->md_off, (c + 1) * 4));
		MSX(msx_host_register(g_ctx, b->cigar, c_cig * 4));
		MSX(msx_host_register(g_ctx, b->md, c_md));
	}
	if (P->mode != 0) MSX(msx_host_register(g_ctx, b->group_off, c_grp * 4));
}
