/*
 * msh_io.c -- BGZF/BAM and SAM-text I/O for the host command line.
 * Written from the SAM/BAM specification (SAMv1 sections 1.3-1.5, 4.1-4.2);
 * replaces what the reference gets from htslib through msam_helper.c:196-293.
 */
#define _GNU_SOURCE
#include "msh.h"

#include <ctype.h>
#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <stdarg.h>
#include <sys/mman.h>
#include <errno.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>
#include <zlib.h>

/* ------------------------------------------------------------------------ */
/* errors, strings                                                            */
/* ------------------------------------------------------------------------ */
pthread_t msh_main_thread;
int msh_main_thread_set;

void mDie(const char *fmt, ...) {
	va_list ap;
	fflush(stdout);
	fprintf(stderr, "Fatal Error: ");
	va_start(ap, fmt);
	vfprintf(stderr, fmt, ap);
	va_end(ap);
	fprintf(stderr, "\n");
	/* a fatal error raised on a worker, reader, writer or device thread: the other threads are still running (a
	 * device thread may be inside the HIP runtime, which exit()'s handlers would tear down under it) -- leave at
	 * once, with the diagnostic and everything written so far flushed.  The only output streams are stdout (flushed above)
	 * and stderr; NOT fflush(NULL): that locks every open stream in turn, the input's among them, and a reader sitting in a
	 * read of a pipe holds its stream's lock for as long as the other end is silent -- the error would wait with it */
	fflush(stderr);
	if (msh_main_thread_set) _exit(EXIT_FAILURE);       /* (the command line: device, reader or writer threads may be running) */
	exit(EXIT_FAILURE);
}

void mQuit(const char *fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vfprintf(stderr, fmt, ap);
	va_end(ap);
	fprintf(stderr, "\n");
	exit(EXIT_FAILURE);
}

void ks_reserve(kstr *k, size_t extra) {
	if (k->l + extra + 1 > k->m) {
		size_t m = k->m ? k->m : 64;
		while (m < k->l + extra + 1) m += m >> 1;
		k->s = (char *)realloc(k->s, m);
		if (!k->s) mDie("Out of memory");
		k->m = m;
	}
}
void ks_put(kstr *k, const void *p, size_t n) {
	ks_reserve(k, n);
	memcpy(k->s + k->l, p, n);
	k->l += n;
	k->s[k->l] = 0;
}
void ks_puts(kstr *k, const char *s) { ks_put(k, s, strlen(s)); }
void ks_putc(kstr *k, int c) {
	char ch = (char)c;
	ks_put(k, &ch, 1);
}
void ks_printf(kstr *k, const char *fmt, ...) {
	va_list ap;
	char buf[512];
	int n;
	va_start(ap, fmt);
	n = vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	if (n < (int)sizeof buf) {
		ks_put(k, buf, (size_t)n);
	} else {
		ks_reserve(k, (size_t)n + 1);
		va_start(ap, fmt);
		vsnprintf(k->s + k->l, (size_t)n + 1, fmt, ap);
		va_end(ap);
		k->l += (size_t)n;
	}
}

static void put_le32(kstr *k, uint32_t v) {
	uint8_t b[4] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)};
	ks_put(k, b, 4);
}
static void put_le16(kstr *k, uint32_t v) {
	uint8_t b[2] = {(uint8_t)v, (uint8_t)(v >> 8)};
	ks_put(k, b, 2);
}

/* ------------------------------------------------------------------------ */
/* header                                                                     */
/* ------------------------------------------------------------------------ */
static void hdr_add_target(msh_hdr *h, const char *name, size_t nl, uint32_t len) {
	h->target_name = (char **)realloc(h->target_name, sizeof(char *) * (size_t)(h->n_targets + 1));
	h->target_len = (uint32_t *)realloc(h->target_len, sizeof(uint32_t) * (size_t)(h->n_targets + 1));
	h->target_name[h->n_targets] = (char *)malloc(nl + 1);
	memcpy(h->target_name[h->n_targets], name, nl);
	h->target_name[h->n_targets][nl] = 0;
	h->target_len[h->n_targets] = len;
	h->n_targets++;
}

/* reference dictionary from @SQ lines (SAM text input) */
static void hdr_targets_from_text(msh_hdr *h) {
	const char *p = h->text.s, *end = h->text.s ? h->text.s + h->text.l : NULL;
	while (p && p < end) {
		const char *nl = memchr(p, '\n', (size_t)(end - p));
		const char *le = nl ? nl : end;
		if (le - p >= 3 && p[0] == '@' && p[1] == 'S' && p[2] == 'Q') {
			const char *q = p + 3, *sn = NULL;
			size_t snl = 0;
			uint32_t ln = 0;
			while (q < le) {
				const char *t = q + 1, *te;
				if (*q != '\t') { q++; continue; }
				te = memchr(t, '\t', (size_t)(le - t));
				if (!te) te = le;
				if (te - t > 3 && t[2] == ':') {
					if (t[0] == 'S' && t[1] == 'N') { sn = t + 3; snl = (size_t)(te - t - 3); }
					if (t[0] == 'L' && t[1] == 'N') ln = (uint32_t)strtoul(t + 3, NULL, 10);
				}
				q = te;
			}
			if (sn) hdr_add_target(h, sn, snl, ln);
		}
		p = nl ? nl + 1 : end;
	}
}

char *msh_hdr_sort_order(const msh_hdr *h) {
	const char *p = h->text.s, *end = h->text.s ? h->text.s + h->text.l : NULL;
	while (p && p < end) {
		const char *nl = memchr(p, '\n', (size_t)(end - p));
		const char *le = nl ? nl : end;
		if (le - p >= 3 && p[0] == '@' && p[1] == 'H' && p[2] == 'D') {
			const char *q = p + 3;
			while (q < le) {
				if (*q == '\t' && le - q > 4 && q[1] == 'S' && q[2] == 'O' && q[3] == ':') {
					const char *v = q + 4, *ve = memchr(v, '\t', (size_t)(le - v));
					size_t n;
					char *out;
					if (!ve) ve = le;
					n = (size_t)(ve - v);
					out = (char *)malloc(n + 1);
					memcpy(out, v, n);
					out[n] = 0;
					return out;
				}
				q++;
			}
			return NULL;
		}
		p = nl ? nl + 1 : end;
	}
	return NULL;
}

/* name -> tid for SAM text in (sam_hdr_name2tid: htslib keeps a hash of the @SQ names).  A one-entry cache per thread in
 * front of an open-addressing table built on first use, under a lock, for the header it is first asked about (a process
 * reads one input); any other header takes the linear probe.  The first @SQ line of a name wins, as the probe's order has it.
 * (Round 5: the cache used to be one static word shared by the parsing threads -- a data race ThreadSanitizer reported --
 * in front of the linear probe alone: O(references) per record whose reference differs from the one before.) */
static struct {
	pthread_mutex_t mu;
	const msh_hdr *owner;     /* published last, with release order: readers that see it see the table */
	int32_t *slot;            /* tid + 1, 0 = empty */
	uint32_t mask;
} g_n2t = {PTHREAD_MUTEX_INITIALIZER, NULL, NULL, 0};
static uint32_t n2t_hash(const char *s) {
	uint32_t h = 2166136261u;
	while (*s) { h ^= (uint8_t)*s++; h *= 16777619u; }
	return h ^ (h >> 15);
}
static void n2t_build(const msh_hdr *h) {
	pthread_mutex_lock(&g_n2t.mu);
	if (!g_n2t.owner) {
		uint32_t cap = 16;
		int32_t i;
		while (cap < 2u * (uint32_t)h->n_targets) cap <<= 1;
		g_n2t.slot = (int32_t *)calloc(cap, sizeof(int32_t));
		if (!g_n2t.slot) mDie("out of memory");
		g_n2t.mask = cap - 1;
		for (i = 0; i < h->n_targets; i++) {
			uint32_t k = n2t_hash(h->target_name[i]) & g_n2t.mask;
			while (g_n2t.slot[k] && strcmp(h->target_name[g_n2t.slot[k] - 1], h->target_name[i]) != 0) k = (k + 1) & g_n2t.mask;
			if (!g_n2t.slot[k]) g_n2t.slot[k] = i + 1;
		}
		__atomic_store_n(&g_n2t.owner, h, __ATOMIC_RELEASE);
	}
	pthread_mutex_unlock(&g_n2t.mu);
}
int32_t msh_hdr_name2tid(const msh_hdr *h, const char *name) {
	static __thread int32_t last = 0;
	const msh_hdr *owner;
	int32_t i;
	if (last < h->n_targets && strcmp(h->target_name[last], name) == 0) return last;
	owner = __atomic_load_n(&g_n2t.owner, __ATOMIC_ACQUIRE);
	if (!owner && h->n_targets > 8) { n2t_build(h); owner = __atomic_load_n(&g_n2t.owner, __ATOMIC_ACQUIRE); }
	if (owner == h) {
		uint32_t k = n2t_hash(name) & g_n2t.mask;
		while (g_n2t.slot[k]) {
			if (strcmp(h->target_name[g_n2t.slot[k] - 1], name) == 0) return last = g_n2t.slot[k] - 1;
			k = (k + 1) & g_n2t.mask;
		}
		return -1;
	}
	for (i = 0; i < h->n_targets; i++)
		if (strcmp(h->target_name[i], name) == 0) { last = i; return i; }
	return -1;
}

/* sam_hdr_add_pg (htslib 1.24 header.c), as the reference calls it at
 * msam_helper.c:170-178: ID made unique ("name", "name.1", ...), one new @PG per
 * existing chain end with PP pointing at it.  Tag order: ID, PP, then the
 * caller's PN VN CL DS. */
void msh_hdr_add_pg(kstr *text, const char *name, const char *vn, const char *cl, const char *ds) {
	/* collect existing @PG IDs and PPs */
	char **ids = NULL, **pps = NULL;
	int n = 0, i, j;
	const char *p = text->s, *end = text->s ? text->s + text->l : NULL;
	while (p && p < end) {
		const char *nl = memchr(p, '\n', (size_t)(end - p));
		const char *le = nl ? nl : end;
		if (le - p >= 3 && p[0] == '@' && p[1] == 'P' && p[2] == 'G') {
			const char *q = p + 3;
			char *id = NULL, *pp = NULL;
			while (q < le) {
				if (*q == '\t' && le - q > 4 && q[3] == ':') {
					const char *v = q + 4, *ve = memchr(v, '\t', (size_t)(le - v));
					if (!ve) ve = le;
					if (q[1] == 'I' && q[2] == 'D') id = strndup(v, (size_t)(ve - v));
					if (q[1] == 'P' && q[2] == 'P') pp = strndup(v, (size_t)(ve - v));
				}
				q++;
			}
			ids = (char **)realloc(ids, sizeof(char *) * (size_t)(n + 1));
			pps = (char **)realloc(pps, sizeof(char *) * (size_t)(n + 1));
			ids[n] = id ? id : strdup("");
			pps[n] = pp;
			n++;
		}
		p = nl ? nl + 1 : end;
	}
	{
		int n_old = n, added = 0, suffix = 0;
		/* chain ends = IDs nobody names as PP */
		for (i = 0; i < n_old || (n_old == 0 && added == 0); i++) {
			char idbuf[256];
			int is_end = 1, clash;
			if (n_old > 0) {
				for (j = 0; j < n_old; j++)
					if (pps[j] && strcmp(pps[j], ids[i]) == 0) is_end = 0;
				if (!is_end) continue;
			}
			do {   /* unique ID */
				if (suffix == 0) snprintf(idbuf, sizeof idbuf, "%s", name);
				else snprintf(idbuf, sizeof idbuf, "%s.%d", name, suffix);
				clash = 0;
				for (j = 0; j < n; j++)
					if (strcmp(ids[j], idbuf) == 0) clash = 1;
				if (clash) suffix++;
			} while (clash);
			if (text->l && text->s[text->l - 1] != '\n') ks_putc(text, '\n');
			ks_printf(text, "@PG\tID:%s", idbuf);
			if (n_old > 0) ks_printf(text, "\tPP:%s", ids[i]);
			ks_printf(text, "\tPN:%s\tVN:%s\tCL:%s\tDS:%s\n", name, vn, cl, ds);
			ids = (char **)realloc(ids, sizeof(char *) * (size_t)(n + 1));
			pps = (char **)realloc(pps, sizeof(char *) * (size_t)(n + 1));
			ids[n] = strdup(idbuf);
			pps[n] = NULL;
			n++;
			added++;
			if (n_old == 0) break;
		}
	}
	for (i = 0; i < n; i++) { free(ids[i]); free(pps[i]); }
	free(ids);
	free(pps);
}

/* ------------------------------------------------------------------------ */
/* aux fields                                                                 */
/* ------------------------------------------------------------------------ */
static size_t aux_type_size(int t) {
	switch (t) {
	case 'A': case 'c': case 'C': return 1;
	case 's': case 'S': return 2;
	case 'i': case 'I': case 'f': return 4;
	case 'd': return 8;
	default: return 0;
	}
}

/* size of the aux field whose type byte is at t (type byte included); a field that does not end inside the
 * record is fatal, like any other damage to a record */
size_t msh_aux_size(const uint8_t *t, const uint8_t *end) {
	int ty = *t;
	size_t fs = aux_type_size(ty);
	if (fs) {
		if ((size_t)(end - t) < 1 + fs) mDie("Corrupt aux field of type '%c' in BAM record", ty);
		return 1 + fs;
	}
	if (ty == 'Z' || ty == 'H') {
		const uint8_t *z = (const uint8_t *)memchr(t + 1, 0, (size_t)(end - t - 1));
		if (!z) mDie("Corrupt aux field of type '%c' in BAM record", ty);
		return (size_t)(z - t) + 1;
	}
	if (ty == 'B' && end - t >= 6) {
		size_t es = aux_type_size(t[1]);
		uint32_t cnt = (uint32_t)le32(t + 2);
		if (es && (size_t)(end - t - 6) / es >= cnt) return 1 + 1 + 4 + es * cnt;
	}
	mDie("Corrupt aux field of type '%c' in BAM record", ty);
	return 0;
}

/* the fixed part of a BAM record and the variable-length fields it announces lie inside the record, and the
 * read name is a string */
void msh_rec_check(const uint8_t *r, size_t len) {
	if (len >= 32) {
		const size_t lq = REC_LQNAME(r), nc = REC_NCIGAR(r);
		const int32_t ls = REC_LSEQ(r);
		if (lq >= 1 && ls >= 0 && 32 + lq + 4 * nc + ((size_t)ls + 1) / 2 + (size_t)ls <= len && r[32 + lq - 1] == 0) return;
	}
	mDie("Corrupt BAM record (its fields do not fit its length)");
}

const uint8_t *msh_aux_get(const uint8_t *rec, size_t len, const char tag[2]) {
	const uint8_t *p = REC_AUX(rec), *end = rec + len;
	while (p + 3 <= end) {
		if (p[0] == (uint8_t)tag[0] && p[1] == (uint8_t)tag[1]) return p + 2;
		p += 2 + msh_aux_size(p + 2, end);
	}
	return NULL;
}

/* The CIGAR the reference computes from (htslib's bam_tag2cigar under sam_read1, msam_helper.c:246-268): a CIGAR of more than
 * 65535 operations is stored as the placeholder <l_seq>S<reference length>N with the real one in a CG:B:I tag (SAMv1 4.2.2),
 * and the reader swaps it in -- for a mapped record whose first operation is S of l_seq bases and whose first CG tag is an
 * array of I / i with at least n_cigar elements (fewer than 2^29).  Returns the words to use and their number; *cg_tag (if
 * asked for) is the tag's first byte when the swap applies, NULL otherwise.  The record's bytes pass through as they are. */
const uint8_t *msh_real_cigar(const uint8_t *r, size_t len, uint32_t *n_out, const uint8_t **cg_tag) {
	const uint32_t n = REC_NCIGAR(r);
	const uint8_t *cig = REC_CIGAR(r), *cg;
	uint32_t c0, cnt;
	*n_out = n;
	if (cg_tag) *cg_tag = NULL;
	if (n == 0 || REC_TID(r) < 0 || REC_POS(r) < 0) return cig;
	c0 = (uint32_t)le32(cig);
	if ((c0 & 15) != 4 || (c0 >> 4) != (uint32_t)REC_LSEQ(r)) return cig;
	cg = msh_aux_get(r, len, "CG");
	if (!cg || cg[0] != 'B' || (cg[1] != 'I' && cg[1] != 'i')) return cig;
	cnt = (uint32_t)le32(cg + 2);
	if (cnt < n || cnt >= (1u << 29)) return cig;
	*n_out = cnt;
	if (cg_tag) *cg_tag = cg - 2;
	return cg + 6;
}

int64_t msh_aux2i(const uint8_t *s) {
	switch (*s) {
	case 'c': return (int8_t)s[1];
	case 'C': return s[1];
	case 's': return (int16_t)le16(s + 1);
	case 'S': return le16(s + 1);
	case 'i': return le32(s + 1);
	case 'I': return (uint32_t)le32(s + 1);
	default: return 0;
	}
}

/* ------------------------------------------------------------------------ */
/* threads                                                                    */
/* ------------------------------------------------------------------------ */
#define MSH_MAX_THREADS 128
int msh_threads(void) {
	static int cached = 0;
	if (!cached) {
		const char *e = getenv("MSX_THREADS");
		long n = e ? strtol(e, NULL, 10) : sysconf(_SC_NPROCESSORS_ONLN);
		if (!e) {
			/* what this process may actually use: its affinity mask and the cgroup's CPU quota */
			cpu_set_t set;
			FILE *f;
			if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0 && CPU_COUNT(&set) < n) n = CPU_COUNT(&set);
			if ((f = fopen("/sys/fs/cgroup/cpu.max", "r")) != NULL) {
				long long quota = 0, period = 0;
				if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0 && (quota + period - 1) / period < n)
					n = (long)((quota + period - 1) / period);
				fclose(f);
			}
			if (n > 96) n = 96;                /* beyond that the stages of this pipeline stop gaining */
		}
		if (n < 1) n = 1;
		if (n > MSH_MAX_THREADS) n = MSH_MAX_THREADS;
		cached = (int)n;
	}
	return cached;
}

/* A persistent pool: the pipeline calls msh_parallel thousands of times per file, from several
 * stage threads at once.  A job is a counter of thread slots; workers and the caller itself claim
 * slots until none are left, the caller then waits for the stragglers. */
typedef struct pf_job {
	msh_pf fn;
	void *arg;
	int nth;
	int next;                 /* next unclaimed slot (under pool.mu) */
	int done;                 /* finished slots (under pool.mu) */
	pthread_cond_t fin;
	struct pf_job *link;
} pf_job;

static struct {
	pthread_mutex_t mu;
	pthread_cond_t work;
	pf_job *head, *tail;      /* jobs with unclaimed slots */
	int started;
} pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, NULL, NULL, 0};

static void pool_run_slot(pf_job *j, int slot) {     /* called without the lock */
	j->fn(j->arg, slot, j->nth);
	pthread_mutex_lock(&pool.mu);
	if (++j->done == j->nth) pthread_cond_signal(&j->fin);
	pthread_mutex_unlock(&pool.mu);
}

static void *pool_worker(void *unused) {
	(void)unused;
	pthread_mutex_lock(&pool.mu);
	for (;;) {
		pf_job *j;
		int slot;
		while (!pool.head) pthread_cond_wait(&pool.work, &pool.mu);
		j = pool.head;
		slot = j->next++;
		if (j->next == j->nth) {             /* fully claimed: off the list */
			pool.head = j->link;
			if (!pool.head) pool.tail = NULL;
		}
		pthread_mutex_unlock(&pool.mu);
		pool_run_slot(j, slot);
		pthread_mutex_lock(&pool.mu);
	}
	return NULL;
}

static void pool_start(void) {
	int i, n = msh_threads() - 1;
	pool.started = 1;
	for (i = 0; i < n; i++) {
		pthread_t th;
		pthread_attr_t at;
		pthread_attr_init(&at);
		pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
		if (pthread_create(&th, &at, pool_worker, NULL) != 0) mDie("pthread_create failed");
		pthread_attr_destroy(&at);
	}
}

void msh_parallel(int nth, msh_pf fn, void *arg) {
	pf_job job;
	if (nth > MSH_MAX_THREADS) nth = MSH_MAX_THREADS;
	if (nth <= 1) { fn(arg, 0, 1); return; }
	job.fn = fn; job.arg = arg; job.nth = nth; job.next = 1; job.done = 0; job.link = NULL;   /* slot 0 is the caller's */
	pthread_cond_init(&job.fin, NULL);
	pthread_mutex_lock(&pool.mu);
	if (!pool.started) pool_start();
	if (pool.tail) pool.tail->link = &job; else pool.head = &job;
	pool.tail = &job;
	pthread_cond_broadcast(&pool.work);
	pthread_mutex_unlock(&pool.mu);
	pool_run_slot(&job, 0);
	/* help with whatever of this job is still unclaimed, then wait for the rest */
	pthread_mutex_lock(&pool.mu);
	while (job.next < job.nth) {
		int slot = job.next++;
		if (job.next == job.nth) {
			pf_job **pp = &pool.head, *prev = NULL;
			while (*pp && *pp != &job) { prev = *pp; pp = &(*pp)->link; }
			if (*pp == &job) {
				*pp = job.link;
				if (pool.tail == &job) pool.tail = prev;
			}
		}
		pthread_mutex_unlock(&pool.mu);
		pool_run_slot(&job, slot);
		pthread_mutex_lock(&pool.mu);
	}
	while (job.done < job.nth) pthread_cond_wait(&job.fin, &pool.mu);
	pthread_mutex_unlock(&pool.mu);
	pthread_cond_destroy(&job.fin);
}

/* ------------------------------------------------------------------------ */
/* CRC-32 of a BGZF payload.  zlib's table-driven crc32 runs at about a          */
/* gigabyte per second and core, and every byte that enters or leaves as BAM      */
/* passes through it; with carry-less multiplication (the folding method of      */
/* Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ") */
/* the same value costs a tenth of that.  Constants for the reflected IEEE        */
/* polynomial: x^(4*128+64), x^(4*128), x^(128+64), x^128, x^64 mod P, then P and  */
/* the Barrett quotient.  Anything the fast path does not take (short buffers,   */
/* the last bytes, other CPUs) goes to zlib.                                      */
/* ------------------------------------------------------------------------ */
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("pclmul,sse4.1")))
static uint32_t crc32_fold(const uint8_t *buf, size_t len, uint32_t crc) {   /* len >= 64, a multiple of 16; crc pre-inverted */
	const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596LL, 0x0154442bd4LL);
	const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009eLL, 0x01751997d0LL);
	const __m128i k5k0 = _mm_set_epi64x(0x0000000000LL, 0x0163cd6124LL);
	const __m128i poly = _mm_set_epi64x(0x01f7011641LL, 0x01db710641LL);
	__m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
	x1 = _mm_loadu_si128((const __m128i *)(buf + 0x00));
	x2 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
	x3 = _mm_loadu_si128((const __m128i *)(buf + 0x20));
	x4 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
	x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
	x0 = k1k2;
	buf += 64;
	len -= 64;
	while (len >= 64) {                  /* four lanes of 128 bits, folded 512 bits ahead */
		x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
		x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
		x7 = _mm_clmulepi64_si128(x3, x0, 0x00);
		x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
		x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
		x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
		x3 = _mm_clmulepi64_si128(x3, x0, 0x11);
		x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
		y5 = _mm_loadu_si128((const __m128i *)(buf + 0x00));
		y6 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
		y7 = _mm_loadu_si128((const __m128i *)(buf + 0x20));
		y8 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
		x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5);
		x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
		x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7);
		x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
		buf += 64;
		len -= 64;
	}
	x0 = k3k4;                           /* the four lanes into one */
	x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
	x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
	x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
	x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
	x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
	x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
	x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
	x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
	x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
	while (len >= 16) {                  /* single 128-bit folds */
		x2 = _mm_loadu_si128((const __m128i *)buf);
		x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
		x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
		x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
		buf += 16;
		len -= 16;
	}
	x2 = _mm_clmulepi64_si128(x1, x0, 0x10);     /* 128 -> 64 bits */
	x3 = _mm_setr_epi32(~0, 0, ~0, 0);
	x1 = _mm_srli_si128(x1, 8);
	x1 = _mm_xor_si128(x1, x2);
	x0 = k5k0;
	x2 = _mm_srli_si128(x1, 4);
	x1 = _mm_and_si128(x1, x3);
	x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
	x1 = _mm_xor_si128(x1, x2);
	x0 = poly;                           /* Barrett reduction to 32 bits */
	x2 = _mm_and_si128(x1, x3);
	x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
	x2 = _mm_and_si128(x2, x3);
	x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
	x1 = _mm_xor_si128(x1, x2);
	return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif

uint32_t msh_crc32(const void *p, size_t n) {
	const uint8_t *s = (const uint8_t *)p;
	uLong c = crc32(0L, NULL, 0);
#if defined(__x86_64__)
	static int fast_flag = -1;           /* (every thread would compute the same value: relaxed accesses, no lock) */
	int fast = __atomic_load_n(&fast_flag, __ATOMIC_RELAXED);
	if (fast < 0) {
		fast = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1") && !getenv("MSX_NO_PCLMUL");
		__atomic_store_n(&fast_flag, fast, __ATOMIC_RELAXED);
	}
	if (fast && n >= 64) {
		const size_t m = n & ~(size_t)15;
		c = (uLong)(~crc32_fold(s, m, ~(uint32_t)c) & 0xffffffffu);
		s += m;
		n -= m;
	}
#endif
	while (n) {                          /* (zlib takes an unsigned int at a time) */
		const size_t k = n > 0x40000000u ? 0x40000000u : n;
		c = crc32(c, s, (uInt)k);
		s += k;
		n -= k;
	}
	return (uint32_t)c;
}

/* ------------------------------------------------------------------------ */
/* BGZF reader: batches of raw blocks inflated in parallel into one           */
/* contiguous "span" of BAM bytes                                             */
/* ------------------------------------------------------------------------ */
#define BGZF_MAX 65536
#define BGZF_BATCH 1024
/* blocks inflated per call: MSX_INFLATE_BLOCKS lowers it (tests: many small batches through the pipeline) */
static int bgzf_blocks_limit;          /* msh_inflate_limit: a caller's own, temporary limit (0: none) */
static int bgzf_batch_blocks(void) {
	static int v = 0;
	if (!v) {
		const char *e = getenv("MSX_INFLATE_BLOCKS");
		const long n = e ? strtol(e, NULL, 10) : 0;
		v = (n >= 1 && n < BGZF_BATCH) ? (int)n : BGZF_BATCH;
	}
	return (bgzf_blocks_limit > 0 && bgzf_blocks_limit < v) ? bgzf_blocks_limit : v;
}
void msh_inflate_limit(int blocks) { bgzf_blocks_limit = blocks; }
#define RD_NBUF 3
#define RD_HEAD (BGZF_MAX + 1024)    /* headroom in front of a ring buffer's data: the tail of the block its predecessor cut */

typedef struct {
	FILE *fp;
	const uint8_t *map;                  /* a regular file is mapped: blocks are inflated straight out of the page cache */
	size_t map_len, map_pos;
	size_t map_released;                 /* pages of the mapping in front of this offset have been given back */
	/* otherwise (a pipe): a thread of its own keeps draining the descriptor with read(2) into a ring of raw
	 * buffers -- the writer at the other end never waits for this process to finish parsing a batch -- and
	 * the blocks are parsed in place.  A block cut by a buffer's end is completed in the headroom in front of
	 * the next buffer's data. */
	uint8_t *cbuf;                       /* the buffer being parsed (one of rd_buf[]) */
	size_t cbeg, cend, ccap;             /* unparsed raw bytes of cbuf; capacity of a ring buffer's data area */
	int fd;
	int cur;                             /* ring slot cbuf points into, -1: none */
	uint8_t *rd_buf[RD_NBUF];
	size_t rd_len[RD_NBUF];
	int rd_full[RD_NBUF];
	int rd_head, rd_eof, rd_wait, rd_started;
	size_t rd_prefill;
	const uint8_t *rd_pre;               /* the rd_prefill bytes msh_open read to tell BAM from compressed SAM text */
	pthread_t rd_thr;
	pthread_mutex_t rd_mu;
	pthread_cond_t rd_cv_full, rd_cv_free;
	const uint8_t *cptr[BGZF_BATCH];     /* where each raw block starts */
	size_t coff[BGZF_BATCH + 1];
	size_t uoff[BGZF_BATCH + 1];         /* where each block inflates to, relative to dst */
	uint8_t *dst;
	int nblk, eof;
	/* the span: inflated, not yet consumed bytes */
	uint8_t *span;
	size_t span_beg, span_end, span_cap;
} bgz_in;

/* one block's DEFLATE stream into `out` (isize bytes expected, CRC-32 `crc`) */
static void inflate_payload(const uint8_t *data, size_t dlen, uint8_t *out, uint32_t isize, uint32_t crc) {
	/* one stream per thread, reset between blocks: initialising one per block means an allocation per
	 * block, and with a hundred threads those serialise inside the allocator */
	static __thread z_stream zs;
	static __thread int zs_ready = 0;
	static int fast_flag = -1;
	int fast = __atomic_load_n(&fast_flag, __ATOMIC_RELAXED);
	if (isize == 0) return;
	if (fast < 0) {
		fast = !getenv("MSX_NO_FAST_INFLATE");
		__atomic_store_n(&fast_flag, fast, __ATOMIC_RELAXED);
	}
	/* the decoder of msh_inflate.c first (twice zlib's speed on BAM records); whatever it does not vouch for,
	 * and whatever fails the CRC afterwards, is decoded again by zlib, whose verdict stands */
	if (fast && msh_fast_inflate(data, dlen, out, isize) && msh_crc32(out, isize) == crc) return;
	if (!zs_ready) {
		memset(&zs, 0, sizeof zs);
		if (inflateInit2(&zs, -15) != Z_OK) mDie("zlib inflateInit2 failed");
		zs_ready = 1;
	} else if (inflateReset(&zs) != Z_OK) {
		mDie("zlib inflateReset failed");
	}
	zs.next_in = (Bytef *)data;
	zs.avail_in = (uInt)dlen;
	zs.next_out = out;
	zs.avail_out = isize;
	if (inflate(&zs, Z_FINISH) != Z_STREAM_END || zs.total_out != isize) mDie("Corrupt BGZF block (inflate failed)");
	if (msh_crc32(out, isize) != crc) mDie("Corrupt BGZF block (CRC mismatch)");
}

static void inflate_block(bgz_in *b, int i) {
	const uint8_t *c = b->cptr[i];
	size_t clen = b->coff[i + 1] - b->coff[i];
	uint32_t xlen = le16(c + 10);
	inflate_payload(c + 12 + xlen, clen - 12 - xlen - 8, b->dst + b->uoff[i], (uint32_t)(b->uoff[i + 1] - b->uoff[i]),
	                (uint32_t)le32(c + clen - 8));
}

static void inflate_worker(void *arg, int tid, int nth) {
	bgz_in *b = (bgz_in *)arg;
	int i;
	for (i = tid; i < b->nblk; i += nth) inflate_block(b, i);
}

/* the draining thread: fills free ring buffers in order; a buffer is handed over when it is full, at end of
 * input, or -- so that a slow producer does not hold a batch back -- as soon as the parser is waiting and
 * there is a megabyte to give it */
static void *bgz_reader_main(void *arg) {
	bgz_in *b = (bgz_in *)arg;
	int slot = 0;
	for (;;) {
		size_t n = 0;
		int eof = 0;
		pthread_mutex_lock(&b->rd_mu);
		while (b->rd_full[slot]) pthread_cond_wait(&b->rd_cv_free, &b->rd_mu);
		pthread_mutex_unlock(&b->rd_mu);
		if (b->rd_prefill) {             /* the bytes msh_open looked at */
			memcpy(b->rd_buf[slot] + RD_HEAD, b->rd_pre, b->rd_prefill);
			n = b->rd_prefill;
			b->rd_prefill = 0;
		}
		while (n < b->ccap) {
			ssize_t k = read(b->fd, b->rd_buf[slot] + RD_HEAD + n, b->ccap - n);
			if (k < 0 && errno == EINTR) continue;
			if (k < 0) mDie("Read failed");
			if (k == 0) { eof = 1; break; }
			n += (size_t)k;
			if (n >= ((size_t)1 << 20) && __atomic_load_n(&b->rd_wait, __ATOMIC_RELAXED)) break;   /* (a hint: no ordering needed) */
		}
		pthread_mutex_lock(&b->rd_mu);
		b->rd_len[slot] = n;
		b->rd_full[slot] = 1;
		if (eof) b->rd_eof = 1;
		pthread_cond_signal(&b->rd_cv_full);
		pthread_mutex_unlock(&b->rd_mu);
		if (eof) break;
		slot = (slot + 1) % RD_NBUF;
	}
	return NULL;
}

/* The buffer being parsed is used up (what is left of it, less than a block, is carried over): give it back
 * and take the next one.  Returns 0 at the end of the input, *left = the bytes that were carried to nowhere. */
static int bgz_next_buffer(bgz_in *b, size_t *left_out) {
	const size_t left = b->cur >= 0 ? b->cend - b->cbeg : 0;
	uint8_t tail[RD_HEAD];
	int slot;
	if (!b->rd_started) {
		int i;
		for (i = 0; i < RD_NBUF; i++)
			if (!(b->rd_buf[i] = (uint8_t *)malloc(RD_HEAD + b->ccap))) mDie("Out of memory");
		pthread_mutex_init(&b->rd_mu, NULL);
		pthread_cond_init(&b->rd_cv_full, NULL);
		pthread_cond_init(&b->rd_cv_free, NULL);
		if (pthread_create(&b->rd_thr, NULL, bgz_reader_main, b) != 0) mDie("Cannot start the reader thread");
		b->rd_started = 1;
	}
	if (left) memcpy(tail, b->cbuf + b->cbeg, left);
	*left_out = left;
	pthread_mutex_lock(&b->rd_mu);
	if (b->cur >= 0) {
		b->rd_full[b->cur] = 0;
		pthread_cond_signal(&b->rd_cv_free);
		b->cur = -1;
	}
	slot = b->rd_head;
	__atomic_store_n(&b->rd_wait, 1, __ATOMIC_RELAXED);
	while (!b->rd_full[slot] && !b->rd_eof) pthread_cond_wait(&b->rd_cv_full, &b->rd_mu);   /* (the last buffer and rd_eof are set together) */
	__atomic_store_n(&b->rd_wait, 0, __ATOMIC_RELAXED);
	if (!b->rd_full[slot]) {             /* the reader has handed over its last buffer, and that one is behind us */
		pthread_mutex_unlock(&b->rd_mu);
		return 0;
	}
	pthread_mutex_unlock(&b->rd_mu);
	b->rd_head = (slot + 1) % RD_NBUF;
	b->cur = slot;
	b->cbuf = b->rd_buf[slot];
	b->cbeg = RD_HEAD - left;
	b->cend = RD_HEAD + b->rd_len[slot];
	if (left) memcpy(b->cbuf + b->cbeg, tail, left);
	return 1;
}

/* read the next batch of raw blocks into cbuf; returns the number of bytes they inflate to (0 at EOF) */
static size_t bgz_read_blocks(bgz_in *b) {
	size_t off = 0, total = 0;
	b->nblk = 0;
	if (b->eof) return 0;
	b->uoff[0] = 0;
	while (b->nblk < bgzf_batch_blocks()) {
		uint32_t bsize, isize;
		const uint8_t *blk;
		if (b->map) {
			/* mapped file: nothing is copied here, the inflating threads read the pages themselves */
			const uint8_t *h = b->map + b->map_pos;
			const size_t left = b->map_len - b->map_pos;
			uint32_t xlen, p = 0;
			int found = 0;
			if (left == 0) { b->eof = 1; break; }
			if (left < 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4))
				mDie("Input is not BGZF-compressed BAM (bad block header)");
			xlen = le16(h + 10);
			if (12 + (size_t)xlen > left) mDie("Truncated BGZF block");
			bsize = 0;
			while (p + 4 <= xlen) {
				uint32_t sl = le16(h + 12 + p + 2);
				if (h[12 + p] == 'B' && h[12 + p + 1] == 'C' && sl == 2) { bsize = le16(h + 12 + p + 4) + 1; found = 1; }
				p += 4 + sl;
			}
			if (!found) mDie("BGZF block without BC subfield");
			if (bsize < 12 + xlen + 8 || bsize > BGZF_MAX + 1024) mDie("Corrupt BGZF block size");
			if (bsize > left) mDie("Truncated BGZF block");
			blk = h;
			b->map_pos += bsize;
		} else {
			/* whole blocks out of the raw buffer; when the next block is not complete in it, the batch ends here if
			 * it has blocks (they point into this buffer, which therefore stays), otherwise the next buffer is taken */
			int got_block = 0;
			bsize = 0;
			for (;;) {
				const size_t have = b->cur >= 0 ? b->cend - b->cbeg : 0;
				if (have >= 18) {
					const uint8_t *h = b->cbuf + b->cbeg;
					const uint32_t xlen = le16(h + 10);
					if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4))
						mDie("Input is not BGZF-compressed BAM (bad block header)");
					if (have >= 12 + (size_t)xlen) {
						uint32_t p = 0;
						int found = 0;
						while (p + 4 <= xlen) {
							uint32_t sl = le16(h + 12 + p + 2);
							if (h[12 + p] == 'B' && h[12 + p + 1] == 'C' && sl == 2) { bsize = le16(h + 12 + p + 4) + 1; found = 1; }
							p += 4 + sl;
						}
						if (!found) mDie("BGZF block without BC subfield");
						if (bsize < 12 + xlen + 8 || bsize > BGZF_MAX + 1024) mDie("Corrupt BGZF block size");
						if (have >= bsize) { got_block = 1; break; }
					}
				}
				if (b->nblk > 0) break;
				{
					size_t left = 0;
					if (!bgz_next_buffer(b, &left)) {
						if (left == 0) { b->eof = 1; break; }
						mDie(left < 18 ? "Input is not BGZF-compressed BAM (bad block header)" : "Truncated BGZF block");
					}
				}
			}
			if (!got_block) break;
			blk = b->cbuf + b->cbeg;
			b->cbeg += bsize;
		}
		isize = (uint32_t)le32(blk + bsize - 4);
		if (isize > BGZF_MAX) mDie("Corrupt BGZF block (ISIZE %u)", isize);
		b->cptr[b->nblk] = blk;
		b->coff[b->nblk] = off;
		off += bsize;
		total += isize;
		b->nblk++;
		b->coff[b->nblk] = off;
		b->uoff[b->nblk] = total;
	}
	return total;
}

/* The blocks in front of the read position have been inflated or copied: their pages of the mapping are given back now
 * (the page cache keeps them; the process's page tables do not: a 1.6 GB mapping is 400 000 entries to tear down when
 * the process ends, and the decode stage has time to spare). */
static void bgz_release_consumed(bgz_in *b) {
	const size_t pg = 4096, lo = (b->map_released + pg - 1) / pg * pg, hi = b->map_pos / pg * pg;
	if (!b->map || hi <= lo || hi - lo < ((size_t)8 << 20)) return;
#ifdef MADV_DONTNEED
	(void)madvise((void *)(b->map + lo), hi - lo, MADV_DONTNEED);
#endif
	b->map_released = hi;
}

/* read the next batch of raw blocks and append their inflated bytes to the span; 0 at EOF */
static int bgz_fill(bgz_in *b) {
	size_t total = bgz_read_blocks(b);
	if (b->nblk == 0) return 0;
	/* make room: compact the unconsumed bytes to the front when that frees enough, else grow */
	if (b->span_end + total > b->span_cap) {
		size_t live = b->span_end - b->span_beg;
		if (live + total > b->span_cap) {
			size_t cap = b->span_cap ? b->span_cap : ((size_t)4 << 20);
			uint8_t *ns;
			while (cap < live + total) cap += cap >> 1;
			ns = (uint8_t *)malloc(cap);
			if (!ns) mDie("Out of memory");
			if (live) memcpy(ns, b->span + b->span_beg, live);
			free(b->span);
			b->span = ns;
			b->span_cap = cap;
		} else if (live) {
			memmove(b->span, b->span + b->span_beg, live);
		}
		b->span_beg = 0;
		b->span_end = live;
	}
	b->dst = b->span + b->span_end;
	msh_parallel(msh_threads() < b->nblk ? msh_threads() : b->nblk, inflate_worker, b);
	b->span_end += total;
	return 1;
}

/* ------------------------------------------------------------------------ */
/* SAM text <-> BAM record                                                    */
/* ------------------------------------------------------------------------ */
static const char SEQ_NT16[] = "=ACMGRSVTWYHKDBN";
static const char CIGAR_OPS[] = "MIDNSHP=XB";

static int reg2bin(int64_t beg, int64_t end) {   /* SAMv1 5.3 */
	--end;
	if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
	if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
	if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
	if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
	if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
	return 0;
}

static void aux_put_int(kstr *rec, int64_t v) {   /* smallest fitting type, as htslib's SAM parser */
	if (v < 0) {
		if (v >= -128) { ks_putc(rec, 'c'); ks_putc(rec, (int)(v & 0xff)); }
		else if (v >= -32768) { ks_putc(rec, 's'); put_le16(rec, (uint32_t)(v & 0xffff)); }
		else { ks_putc(rec, 'i'); put_le32(rec, (uint32_t)v); }
	} else {
		if (v <= 255) { ks_putc(rec, 'C'); ks_putc(rec, (int)v); }
		else if (v <= 65535) { ks_putc(rec, 'S'); put_le16(rec, (uint32_t)v); }
		else { ks_putc(rec, 'I'); put_le32(rec, (uint32_t)v); }
	}
}

void msh_sam_parse(const msh_hdr *h, char *line, kstr *rec) {
	char *f[12], *p = line, *aux = NULL;
	int nf = 0, i;
	int32_t tid, mtid, pos, mpos, tlen;
	uint32_t flag, mapq, n_cigar = 0, l_seq, n_long = 0;
	int64_t reflen = 0;
	size_t qn_len, core_at;
	static __thread kstr long_cigar;
	while (nf < 11) {
		char *t = strchr(p, '\t');
		f[nf++] = p;
		if (!t) { p = NULL; break; }
		*t = 0;
		p = t + 1;
	}
	if (nf < 11) mDie("Malformed SAM record (fewer than 11 fields)");
	aux = p;
	flag = (uint32_t)strtoul(f[1], NULL, 10);
	tid = strcmp(f[2], "*") == 0 ? -1 : msh_hdr_name2tid(h, f[2]);
	if (tid < 0 && strcmp(f[2], "*") != 0) mDie("Unknown reference name '%s' in SAM record", f[2]);
	pos = (int32_t)strtol(f[3], NULL, 10) - 1;
	mapq = (uint32_t)strtoul(f[4], NULL, 10);
	if (strcmp(f[6], "=") == 0) mtid = tid;
	else if (strcmp(f[6], "*") == 0) mtid = -1;
	else mtid = msh_hdr_name2tid(h, f[6]);
	mpos = (int32_t)strtol(f[7], NULL, 10) - 1;
	tlen = (int32_t)strtol(f[8], NULL, 10);
	qn_len = strlen(f[0]);
	if (qn_len > 254) mDie("QNAME longer than 254 characters");
	l_seq = strcmp(f[9], "*") == 0 ? 0 : (uint32_t)strlen(f[9]);
	rec->l = 0;
	core_at = rec->l;
	ks_reserve(rec, 32);
	memset(rec->s, 0, 32);
	rec->l = 32;
	ks_put(rec, f[0], qn_len + 1);
	if (strcmp(f[5], "*") != 0) {
		char *c = f[5];
		const size_t cigar_at = rec->l;
		while (*c) {
			char *e;
			unsigned long len = strtoul(c, &e, 10);
			const char *op = strchr(CIGAR_OPS, *e);
			if (e == c || !*e || !op) mDie("Malformed CIGAR '%s'", f[5]);
			put_le32(rec, (uint32_t)(len << 4 | (uint32_t)(op - CIGAR_OPS)));
			{
				int o = (int)(op - CIGAR_OPS);
				if (o == 0 || o == 2 || o == 3 || o == 7 || o == 8) reflen += (int64_t)len;
			}
			n_cigar++;
			c = e + 1;
		}
		if (n_cigar > 65535) {
			/* more operations than BAM's 16-bit count holds: the placeholder <l_seq>S<reference length>N in the CIGAR's place and
			 * the real one in a CG:B:I tag behind the other optional fields -- what htslib's bam_write1 stores (SAMv1 4.2.2) and
			 * its reader swaps back (msh_real_cigar) */
			ks_reserve(&long_cigar, 4 * (size_t)n_cigar);
			memcpy(long_cigar.s, rec->s + cigar_at, 4 * (size_t)n_cigar);
			long_cigar.l = 4 * (size_t)n_cigar;
			n_long = n_cigar;
			rec->l = cigar_at;
			put_le32(rec, l_seq << 4 | 4u);
			put_le32(rec, (uint32_t)reflen << 4 | 3u);
			n_cigar = 2;
		}
	}
	{   /* SEQ, 4-bit packed */
		uint32_t k;
		for (k = 0; k + 1 < l_seq; k += 2) {
			const char *a = strchr(SEQ_NT16, toupper((unsigned char)f[9][k]));
			const char *b = strchr(SEQ_NT16, toupper((unsigned char)f[9][k + 1]));
			ks_putc(rec, (int)(((a ? a - SEQ_NT16 : 15) << 4) | (b ? b - SEQ_NT16 : 15)));
		}
		if (l_seq & 1) {
			const char *a = strchr(SEQ_NT16, toupper((unsigned char)f[9][l_seq - 1]));
			ks_putc(rec, (int)((a ? a - SEQ_NT16 : 15) << 4));
		}
		if (strcmp(f[10], "*") == 0) {
			for (k = 0; k < l_seq; k++) ks_putc(rec, 0xff);
		} else {
			if (strlen(f[10]) != l_seq) mDie("SEQ and QUAL of different length");
			for (k = 0; k < l_seq; k++) ks_putc(rec, f[10][k] - 33);
		}
	}
	while (aux && *aux) {   /* TAG:TYPE:VALUE */
		char *t = strchr(aux, '\t');
		if (t) *t = 0;
		if (strlen(aux) < 5 || aux[2] != ':' || aux[4] != ':') mDie("Malformed SAM optional field '%s'", aux);
		ks_put(rec, aux, 2);
		switch (aux[3]) {
		case 'A': ks_putc(rec, 'A'); ks_putc(rec, aux[5]); break;
		case 'i': aux_put_int(rec, strtoll(aux + 5, NULL, 10)); break;
		case 'f': {
			float fl = strtof(aux + 5, NULL);
			uint32_t u;
			memcpy(&u, &fl, 4);
			ks_putc(rec, 'f');
			put_le32(rec, u);
			break;
		}
		case 'Z': case 'H': ks_putc(rec, aux[3]); ks_put(rec, aux + 5, strlen(aux + 5) + 1); break;
		case 'B': {
			char sub = aux[5], *c = aux + 6;
			size_t cnt_at;
			uint32_t cnt = 0;
			ks_putc(rec, 'B');
			ks_putc(rec, sub);
			cnt_at = rec->l;
			put_le32(rec, 0);
			while (*c == ',') {
				c++;
				if (sub == 'f') {
					float fl = strtof(c, &c);
					uint32_t u;
					memcpy(&u, &fl, 4);
					put_le32(rec, u);
				} else {
					long long v = strtoll(c, &c, 10);
					size_t es = aux_type_size(sub);
					if (es == 1) ks_putc(rec, (int)(v & 0xff));
					else if (es == 2) put_le16(rec, (uint32_t)(v & 0xffff));
					else put_le32(rec, (uint32_t)v);
				}
				cnt++;
			}
			rec->s[cnt_at] = (char)cnt; rec->s[cnt_at + 1] = (char)(cnt >> 8);
			rec->s[cnt_at + 2] = (char)(cnt >> 16); rec->s[cnt_at + 3] = (char)(cnt >> 24);
			break;
		}
		default: mDie("Unknown SAM optional field type '%c'", aux[3]);
		}
		aux = t ? t + 1 : NULL;
	}
	if (n_long) {
		ks_put(rec, "CGBI", 4);
		put_le32(rec, n_long);
		ks_put(rec, long_cigar.s, long_cigar.l);
	}
	{   /* fixed-length core */
		uint8_t *c = (uint8_t *)rec->s + core_at;
		int64_t end = pos + (reflen > 0 ? reflen : 1);
		uint32_t bin = (uint32_t)reg2bin(pos < 0 ? 0 : pos, end < 1 ? 1 : end);
		uint32_t v[8];
		v[0] = (uint32_t)tid; v[1] = (uint32_t)pos;
		v[2] = (uint32_t)(qn_len + 1) | mapq << 8 | bin << 16;
		v[3] = n_cigar | flag << 16;
		v[4] = l_seq; v[5] = (uint32_t)mtid; v[6] = (uint32_t)mpos; v[7] = (uint32_t)tlen;
		for (i = 0; i < 8; i++) {
			c[4 * i] = (uint8_t)v[i]; c[4 * i + 1] = (uint8_t)(v[i] >> 8);
			c[4 * i + 2] = (uint8_t)(v[i] >> 16); c[4 * i + 3] = (uint8_t)(v[i] >> 24);
		}
	}
}

void msh_sam_format(const msh_hdr *h, const uint8_t *r, size_t len, kstr *o) {
	int32_t tid = (msh_rec_check(r, len), REC_TID(r)), mtid = le32(r + 20);
	uint32_t n_cigar = REC_NCIGAR(r), l_seq = (uint32_t)REC_LSEQ(r), k;
	const uint8_t *cig = REC_CIGAR(r), *seq = cig + 4 * n_cigar, *qual = seq + (l_seq + 1) / 2;
	const uint8_t *p = qual + l_seq, *end = r + len, *cg_tag = NULL;
	cig = msh_real_cigar(r, len, &n_cigar, &cg_tag);     /* (a long CIGAR kept in CG:B:I is printed in its place, the tag left out: what htslib's reader hands sam_format1) */
	ks_puts(o, REC_QNAME(r));
	ks_printf(o, "\t%u\t", REC_FLAG(r));
	ks_puts(o, tid >= 0 && tid < h->n_targets ? h->target_name[tid] : "*");
	ks_printf(o, "\t%lld\t%u\t", (long long)REC_POS(r) + 1, REC_MAPQ(r));
	if (n_cigar == 0) ks_putc(o, '*');
	for (k = 0; k < n_cigar; k++) {
		uint32_t c = (uint32_t)le32(cig + 4 * k);
		ks_printf(o, "%u%c", c >> 4, (c & 15) < 10 ? CIGAR_OPS[c & 15] : '?');
	}
	ks_putc(o, '\t');
	if (mtid < 0) ks_putc(o, '*');
	else if (mtid == tid) ks_putc(o, '=');
	else ks_puts(o, mtid < h->n_targets ? h->target_name[mtid] : "*");
	ks_printf(o, "\t%lld\t%d\t", (long long)le32(r + 24) + 1, le32(r + 28));
	if (l_seq == 0) ks_putc(o, '*');
	else {
		ks_reserve(o, l_seq);
		for (k = 0; k < l_seq; k++) o->s[o->l++] = SEQ_NT16[(seq[k >> 1] >> ((~k & 1) << 2)) & 15];
		o->s[o->l] = 0;
	}
	ks_putc(o, '\t');
	if (l_seq == 0 || qual[0] == 0xff) ks_putc(o, '*');
	else {
		ks_reserve(o, l_seq);
		for (k = 0; k < l_seq; k++) o->s[o->l++] = (char)(qual[k] + 33);
		o->s[o->l] = 0;
	}
	while (p + 3 <= end) {
		int ty = p[2];
		const size_t fsz = msh_aux_size(p + 2, end);       /* (checks that the field ends inside the record) */
		if (p == cg_tag) { p += 2 + fsz; continue; }
		ks_putc(o, '\t');
		ks_put(o, p, 2);
		switch (ty) {
		case 'A': ks_printf(o, ":A:%c", p[3]); break;
		case 'c': case 'C': case 's': case 'S': case 'i': case 'I':
			ks_printf(o, ":i:%lld", (long long)msh_aux2i(p + 2)); break;
		case 'f': { float fl; memcpy(&fl, p + 3, 4); ks_printf(o, ":f:%g", fl); break; }
		case 'd': { double d; memcpy(&d, p + 3, 8); ks_printf(o, ":d:%g", d); break; }
		case 'Z': case 'H': ks_printf(o, ":%c:", ty); ks_puts(o, (const char *)p + 3); break;
		case 'B': {
			int sub = p[3];
			uint32_t cnt = (uint32_t)le32(p + 4), q;
			size_t es = aux_type_size(sub);
			const uint8_t *e = p + 8;
			ks_printf(o, ":B:%c", sub);
			for (q = 0; q < cnt; q++, e += es) {
				if (sub == 'f') { float fl; memcpy(&fl, e, 4); ks_printf(o, ",%g", fl); }
				else {
					uint8_t tmp[5];
					tmp[0] = (uint8_t)sub;
					memcpy(tmp + 1, e, es);
					ks_printf(o, ",%lld", (long long)msh_aux2i(tmp));
				}
			}
			break;
		}
		default: mDie("Corrupt aux field of type '%c' in BAM record", ty);
		}
		p += 2 + fsz;
	}
}

/* ------------------------------------------------------------------------ */
/* input                                                                      */
/* ------------------------------------------------------------------------ */
struct msh_in {
	FILE *fp;
	int is_bam;
	msh_hdr hdr;
	bgz_in bz;
	/* SAM text */
	char *line;
	size_t line_cap;
	kstr pending;        /* first record line, read while scanning the header */
	int has_pending;
	/* SAM text through the pipelined reader (msh_sam_append) */
	char *tbuf;
	size_t tcap, tlen;   /* tbuf[0, tlen): text read but not parsed yet (an incomplete last line) */
	int text_eof;
	/* gzip / bgzip-compressed SAM text (htslib's sam_open reads it like any other SAM): a thread inflates the stream into a
	 * pipe, fp is the pipe's reading end and everything downstream sees plain text */
	uint8_t *pre;        /* what msh_open read ahead of a gzip stream (at most PRE_MAX bytes) */
	size_t npre;
	FILE *gz_src;        /* the compressed stream itself */
	int gz_wfd;
	pthread_t gz_thr;
	int gz_started;
	int gz_err;          /* the decompressor gave up: gz_errmsg says why (set before it closes the pipe) */
	char gz_errmsg[200];
};

static void gz_text_check(msh_in *in);
int msh_is_bam(const msh_in *in) { return in->is_bam; }

/* ensure at least n unconsumed bytes in the span (BAM); returns 0 if EOF comes first */
static int span_need(msh_in *in, size_t n) {
	while (in->bz.span_end - in->bz.span_beg < n)
		if (!bgz_fill(&in->bz)) return 0;
	return 1;
}

int msh_span_fill(msh_in *in) { return bgz_fill(&in->bz); }

/* the input has been read to its end: whatever of the mapping is still in the page tables goes now, on the caller's thread --
 * not when the process ends, where taking the mapping apart is part of the command's wall time */
void msh_release_input(msh_in *in) {
	bgz_in *b = &in->bz;
#ifdef MADV_DONTNEED
	if (b->map && b->map_len) (void)madvise((void *)b->map, b->map_len, MADV_DONTNEED);
#endif
	b->map_released = b->map_len;
}


const uint8_t *msh_span(msh_in *in, size_t *len) {
	*len = in->bz.span_end - in->bz.span_beg;
	return in->bz.span + in->bz.span_beg;
}

void msh_span_consume(msh_in *in, size_t n) { in->bz.span_beg += n; }

/* The pipelined reader owns its batch buffers: append the inflated bytes of the next batch of blocks
 * to *buf (grown as needed; *len bytes in use).  What msh_open left in the span goes first.  Returns the
 * number of bytes appended, 0 at EOF. */
size_t msh_inflate_append(msh_in *in, uint8_t **buf, size_t *len, size_t *cap) {
	bgz_in *b = &in->bz;
	size_t total, live = b->span_end - b->span_beg;
	if (live) {
		total = live;
	} else {
		total = bgz_read_blocks(b);
		if (b->nblk == 0) return 0;
	}
	if (*len + total + 64 > *cap) {
		size_t nc = *cap ? *cap : ((size_t)16 << 20);
		while (nc < *len + total + 64) nc += nc >> 1;
		*buf = (uint8_t *)realloc(*buf, nc);
		if (!*buf) mDie("Out of memory");
		*cap = nc;
	}
	if (live) {
		memcpy(*buf + *len, b->span + b->span_beg, live);
		b->span_beg = b->span_end = 0;
	} else {
		b->dst = *buf + *len;
		msh_parallel(msh_threads() < b->nblk ? msh_threads() : b->nblk, inflate_worker, b);
		bgz_release_consumed(b);
	}
	*len += total;
	return total;
}

/* The device inflater's feed (msx_unpack_enqueue_bgzf): the DEFLATE payloads of the next blocks, copied back to back
 * behind buf[0, *len), and their table behind blk[0, *n) -- at most max_blocks in the table, at most cap bytes in the
 * buffer.  Returns the number of blocks appended: 0 at the end of the input, or when nothing more fits. */
typedef struct { bgz_in *b; uint8_t *dst; const size_t *poff; } rawcopy_job;
static void rawcopy_worker(void *arg, int tid, int nth) {
	rawcopy_job *J = (rawcopy_job *)arg;
	bgz_in *b = J->b;
	int i;
	for (i = tid; i < b->nblk; i += nth) {
		const uint8_t *c = b->cptr[i];
		const uint32_t xlen = le16(c + 10);
		memcpy(J->dst + J->poff[i], c + 12 + xlen, J->poff[i + 1] - J->poff[i]);
	}
}
int msh_raw_append(msh_in *in, uint8_t *buf, size_t cap, size_t *len, msx_bgzf_block *blk, int *n, int max_blocks, size_t *out_total) {
	bgz_in *b = &in->bz;
	static size_t poff[BGZF_BATCH + 1];
	rawcopy_job J;
	size_t room = cap > *len ? (cap - *len) / (BGZF_MAX + 1024) : 0;
	int want = max_blocks - *n, i, added = 0, save = bgzf_blocks_limit;
	if (b->span_end != b->span_beg) mDie("msh_raw_append: inflated bytes pending");
	if ((size_t)want > room) want = (int)room;
	if (want <= 0) return 0;
	bgzf_blocks_limit = want;
	(void)bgz_read_blocks(b);
	bgzf_blocks_limit = save;
	if (b->nblk == 0) return 0;
	poff[0] = 0;
	for (i = 0; i < b->nblk; i++) {
		const size_t clen = b->coff[i + 1] - b->coff[i];
		const uint32_t xlen = le16(b->cptr[i] + 10);
		poff[i + 1] = poff[i] + (clen - 12 - xlen - 8);
	}
	J.b = b; J.dst = buf + *len; J.poff = poff;
	msh_parallel(msh_threads() < b->nblk ? msh_threads() : b->nblk, rawcopy_worker, &J);
	bgz_release_consumed(b);
	for (i = 0; i < b->nblk; i++) {
		const uint8_t *c = b->cptr[i];
		const size_t clen = b->coff[i + 1] - b->coff[i];
		const uint32_t isize = (uint32_t)(b->uoff[i + 1] - b->uoff[i]);
		msx_bgzf_block *q;
		if (isize == 0) continue;                 /* (an empty block -- the end-of-file marker -- has nothing to say) */
		q = &blk[*n];
		q->in_off = *len + poff[i];
		q->in_len = (uint32_t)(poff[i + 1] - poff[i]);
		q->out_off = *out_total;
		q->out_len = isize;
		q->crc32 = (uint32_t)le32(c + clen - 8);
		q->reserved_ = 0;
		*out_total += isize;
		(*n)++;
		added++;
	}
	*len += poff[b->nblk];
	return added ? added : msh_raw_append(in, buf, cap, len, blk, n, max_blocks, out_total);   /* (only empty blocks: read on) */
}

/* what the device refused: the blocks of a table inflated here, with this reader's diagnostics */
typedef struct { const uint8_t *comp; const msx_bgzf_block *blk; int n; uint8_t *out; } tabinf_job;
static void tabinf_worker(void *arg, int tid, int nth) {
	tabinf_job *J = (tabinf_job *)arg;
	int i;
	for (i = tid; i < J->n; i += nth)
		inflate_payload(J->comp + J->blk[i].in_off, J->blk[i].in_len, J->out + J->blk[i].out_off, J->blk[i].out_len, J->blk[i].crc32);
}
void msh_inflate_table(const uint8_t *comp, const msx_bgzf_block *blk, int n, uint8_t *out) {
	tabinf_job J;
	J.comp = comp; J.blk = blk; J.n = n; J.out = out;
	if (n > 0) msh_parallel(msh_threads() < n ? msh_threads() : n, tabinf_worker, &J);
}

/* SAM text for the pipelined reader: the next chunk of lines, parsed on all threads into BAM records
 * ([block_size | record] back to back, in input order) and appended to *buf.  A chunk is SAM_CHUNK bytes of text: at
 * most twice that in BAM bytes (a record's binary form exceeds its text by the fixed core at most).  Returns the number
 * of bytes appended, 0 at the end of the input. */
#define SAM_CHUNK_MAX ((size_t)16 << 20)
static size_t sam_chunk_bytes(void) {           /* MSX_SAM_CHUNK lowers it (tests: many small batches) */
	static size_t v = 0;
	if (!v) {
		const char *e = getenv("MSX_SAM_CHUNK");
		const long long n = e ? strtoll(e, NULL, 10) : 0;
		v = (n >= 4096 && (size_t)n < SAM_CHUNK_MAX) ? (size_t)n : SAM_CHUNK_MAX;
	}
	return v;
}
#define SAM_CHUNK sam_chunk_bytes()
typedef struct {
	const msh_hdr *h;
	char *text;
	size_t lo[MSH_MAX_THREADS + 1];       /* line-aligned ranges of the chunk, one per thread */
	kstr out[MSH_MAX_THREADS];
} sam_job;

static void sam_worker(void *arg, int tid, int nth) {
	sam_job *J = (sam_job *)arg;
	char *p = J->text + J->lo[tid], *end = J->text + J->lo[tid + 1];
	kstr rec = {0, 0, 0}, *o = &J->out[tid];
	(void)nth;
	o->l = 0;
	while (p < end) {
		char *nl = (char *)memchr(p, '\n', (size_t)(end - p));
		char *stop = nl ? nl : end;
		size_t n = (size_t)(stop - p);
		*stop = 0;
		while (n > 0 && p[n - 1] == '\r') p[--n] = 0;
		if (n > 0) {
			uint8_t b4[4];
			msh_sam_parse(J->h, p, &rec);
			b4[0] = (uint8_t)rec.l; b4[1] = (uint8_t)(rec.l >> 8); b4[2] = (uint8_t)(rec.l >> 16); b4[3] = (uint8_t)(rec.l >> 24);
			ks_put(o, b4, 4);
			ks_put(o, rec.s, rec.l);
		}
		p = stop + 1;
	}
	free(rec.s);
}

size_t msh_sam_append(msh_in *in, uint8_t **buf, size_t *len, size_t *cap) {
	static sam_job J;
	size_t end, total = 0;
	int nth = msh_threads(), t;
	if (in->is_bam) mDie("msh_sam_append on BAM input");
	for (;;) {
		const size_t need = in->tlen + (in->has_pending ? in->pending.l + 1 : 0) + SAM_CHUNK + 2;
		if (in->tcap < need) {
			in->tcap = need;
			in->tbuf = (char *)realloc(in->tbuf, in->tcap);
			if (!in->tbuf) mDie("Out of memory");
		}
		if (in->has_pending) {                 /* the first record line, read while the header was scanned */
			in->has_pending = 0;
			memcpy(in->tbuf + in->tlen, in->pending.s, in->pending.l);
			in->tlen += in->pending.l;
			in->tbuf[in->tlen++] = '\n';
		}
		if (!in->text_eof) {
			const size_t got = fread(in->tbuf + in->tlen, 1, SAM_CHUNK, in->fp);
			in->tlen += got;
			if (got < SAM_CHUNK) { in->text_eof = 1; gz_text_check(in); }
		}
		if (in->tlen == 0) return 0;
		/* the chunk ends behind its last newline; at the end of the input the rest is a line as well */
		end = in->tlen;
		if (!in->text_eof) {
			while (end > 0 && in->tbuf[end - 1] != '\n') end--;
			if (end == 0) continue;            /* one line longer than the chunk: read on */
		}
		break;
	}
	if (nth > MSH_MAX_THREADS) nth = MSH_MAX_THREADS;
	if ((size_t)nth > end / 65536 + 1) nth = (int)(end / 65536 + 1);
	J.h = &in->hdr;
	J.text = in->tbuf;
	J.lo[0] = 0;
	for (t = 1; t < nth; t++) {
		size_t q = end * (size_t)t / (size_t)nth;
		if (q < J.lo[t - 1]) q = J.lo[t - 1];
		while (q < end && q > 0 && in->tbuf[q - 1] != '\n') q++;
		J.lo[t] = q;
	}
	J.lo[nth] = end;
	if (end == in->tlen) { in->tbuf[end] = 0; }      /* (room for the terminator of an unterminated last line) */
	msh_parallel(nth, sam_worker, &J);
	for (t = 0; t < nth; t++) total += J.out[t].l;
	if (*len + total + 64 > *cap) {
		size_t nc = *cap ? *cap : ((size_t)16 << 20);
		while (nc < *len + total + 64) nc += nc >> 1;
		*buf = (uint8_t *)realloc(*buf, nc);
		if (!*buf) mDie("Out of memory");
		*cap = nc;
	}
	for (t = 0; t < nth; t++) {
		memcpy(*buf + *len, J.out[t].s, J.out[t].l);
		*len += J.out[t].l;
	}
	memmove(in->tbuf, in->tbuf + end, in->tlen - end);
	in->tlen -= end;
	if (total == 0 && (in->tlen > 0 || !in->text_eof)) return msh_sam_append(in, buf, len, cap);   /* (a chunk of empty lines) */
	return total;
}

#define PRE_MAX 65536
/* the first bytes a gzip stream inflates to (at most n_out), from its first n_in bytes; returns how many came out */
static size_t gz_peek(const uint8_t *in_bytes, size_t n_in, uint8_t *out, size_t n_out) {
	z_stream zs;
	size_t got = 0;
	memset(&zs, 0, sizeof zs);
	if (inflateInit2(&zs, 15 + 32) != Z_OK) return 0;
	zs.next_in = (Bytef *)in_bytes; zs.avail_in = (uInt)n_in;
	zs.next_out = out; zs.avail_out = (uInt)n_out;
	for (;;) {
		const int rc = inflate(&zs, Z_SYNC_FLUSH);
		got = n_out - zs.avail_out;
		if (rc == Z_STREAM_END && got < n_out && zs.avail_in > 0) {      /* an empty member in front (BGZF allows them) */
			if (inflateReset(&zs) != Z_OK) break;
			continue;
		}
		break;
	}
	inflateEnd(&zs);
	return got;
}

static void gz_write_all(int fd, const uint8_t *p, size_t n) {
	while (n) {
		const ssize_t k = write(fd, p, n);
		if (k < 0 && errno == EINTR) continue;
		if (k <= 0) mDie("Write failed");            /* (the reading end is this process's own: it never goes away first) */
		p += k; n -= (size_t)k;
	}
}

/* compressed SAM text: every gzip member of the stream (plain gzip has one, bgzip one per block), inflated into the pipe */
static void *gz_text_main(void *arg) {
	msh_in *in = (msh_in *)arg;
	const int fd = fileno(in->gz_src);
	const size_t ICAP = (size_t)1 << 20, OCAP = (size_t)4 << 20;
	uint8_t *ibuf = (uint8_t *)malloc(ICAP), *obuf = (uint8_t *)malloc(OCAP);
	z_stream zs;
	int at_member_start = 1, eof = 0;
	if (!ibuf || !obuf) mDie("Out of memory");
	memset(&zs, 0, sizeof zs);
	if (inflateInit2(&zs, 15 + 32) != Z_OK) mDie("inflateInit2 failed");
	zs.next_in = in->pre; zs.avail_in = (uInt)in->npre;
	for (;;) {
		if (zs.avail_in == 0 && !eof) {
			ssize_t k;
			do k = read(fd, ibuf, ICAP); while (k < 0 && errno == EINTR);
			if (k < 0) { snprintf(in->gz_errmsg, sizeof in->gz_errmsg, "Read failed"); goto fail; }
			if (k == 0) eof = 1;
			zs.next_in = ibuf; zs.avail_in = (uInt)k;
		}
		if (zs.avail_in == 0 && eof) {
			if (!at_member_start) { snprintf(in->gz_errmsg, sizeof in->gz_errmsg, "Truncated gzip stream in SAM input"); goto fail; }
			break;
		}
		zs.next_out = obuf; zs.avail_out = (uInt)OCAP;
		{
			const int rc = inflate(&zs, Z_NO_FLUSH);
			if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) {
				snprintf(in->gz_errmsg, sizeof in->gz_errmsg, "Corrupt gzip stream in SAM input (%s)", zs.msg ? zs.msg : "zlib error");
				goto fail;
			}
			at_member_start = 0;
			gz_write_all(in->gz_wfd, obuf, OCAP - zs.avail_out);
			if (rc == Z_STREAM_END) {                    /* the next member, if any */
				if (inflateReset(&zs) != Z_OK) { snprintf(in->gz_errmsg, sizeof in->gz_errmsg, "inflateReset failed"); goto fail; }
				at_member_start = 1;
			}
		}
	}
	if (0) {
fail:
		/* Not mDie from here: it flushes every stream, and the reader sits inside a read of the pipe's stream with that
		 * stream's lock held, waiting for bytes this thread would never send -- a deadlock (found by the damaged-input test).
		 * The reason is left for the reader, which sees the end of the pipe next and dies with it (gz_text_check). */
		__atomic_store_n(&in->gz_err, 1, __ATOMIC_RELEASE);
	}
	inflateEnd(&zs);
	free(ibuf);
	free(obuf);
	close(in->gz_wfd);
	return NULL;
}

/* at the end of the text: was it the stream's end, or the decompressor's? */
static void gz_text_check(msh_in *in) {
	if (in->gz_started && __atomic_load_n(&in->gz_err, __ATOMIC_ACQUIRE)) mDie("%s", in->gz_errmsg);
}


msh_in *msh_open(const char *path) {
	msh_in *in = (msh_in *)calloc(1, sizeof(*in));
	int c0, c1;
	if (!in) mDie("Out of memory");
	in->fp = strcmp(path, "-") == 0 ? stdin : fopen(path, "rb");
	if (!in->fp) mDie("Cannot open %s for reading", path);
	setvbuf(in->fp, NULL, _IOFBF, (size_t)4 << 20);
	{
		/* The first two bytes tell BAM from SAM text.  They are read with read(2), before stdio has touched the
		 * descriptor: a BAM stream from a pipe is then read without stdio (and its second copy) altogether. */
		uint8_t two[2];
		size_t n2 = 0;
		while (n2 < 2) {
			ssize_t k = read(fileno(in->fp), two + n2, 2 - n2);
			if (k < 0 && errno == EINTR) continue;
			if (k <= 0) break;
			n2 += (size_t)k;
		}
		c0 = n2 > 0 ? two[0] : EOF;
		c1 = n2 > 1 ? two[1] : EOF;
		in->is_bam = (c0 == 0x1f && c1 == 0x8b);
		if (!in->is_bam) {
			if (c1 != EOF) ungetc(c1, in->fp);
			if (c0 != EOF) ungetc(c0, in->fp);   /* two-byte pushback works on glibc full-buffered streams */
			if (c0 == 'C' && c1 == 'R') {
				/* htslib would read CRAM here (given the reference sequences); this reader has BAM and SAM text only -- said,
				 * instead of a complaint about the fields of a "SAM line" (a CRAM file begins "CRAM", then a binary version) */
				int c[5], k, nc = 0;
				while (nc < 5 && (c[nc] = getc(in->fp)) != EOF) nc++;
				if (nc == 5 && c[2] == 'A' && c[3] == 'M' && c[4] >= 1 && c[4] <= 4)
					mDie("%s is a CRAM file: CRAM input is not supported (samtools view -b makes BAM of it)", path);
				for (k = nc - 1; k >= 0; k--) ungetc(c[k], in->fp);
			}
		} else {
			/* a gzip stream: BAM (BGZF whose first bytes inflate to "BAM\1") or compressed SAM text, which htslib's sam_open
			 * reads like any other SAM (msam_helper.c:203-215 opens with "r" / "rb" and lets it detect the format) */
			uint8_t head[4];
			in->pre = (uint8_t *)malloc(PRE_MAX);
			if (!in->pre) mDie("Out of memory");
			in->pre[0] = 0x1f; in->pre[1] = 0x8b;
			in->npre = 2;
			while (in->npre < PRE_MAX) {
				ssize_t k = read(fileno(in->fp), in->pre + in->npre, PRE_MAX - in->npre);
				if (k < 0 && errno == EINTR) continue;
				if (k <= 0) break;
				in->npre += (size_t)k;
			}
			const size_t nh = gz_peek(in->pre, in->npre, head, 4);
			if (nh >= 1 && !(nh == 4 && memcmp(head, "BAM\1", 4) == 0)) {
				int pfd[2];
				if (pipe(pfd) != 0) mDie("pipe failed");
#ifdef F_SETPIPE_SZ
				(void)fcntl(pfd[1], F_SETPIPE_SZ, 1 << 20);
#endif
				in->is_bam = 0;
				in->gz_src = in->fp;
				in->gz_wfd = pfd[1];
				in->fp = fdopen(pfd[0], "rb");
				if (!in->fp) mDie("fdopen failed");
				setvbuf(in->fp, NULL, _IOFBF, (size_t)4 << 20);
				if (pthread_create(&in->gz_thr, NULL, gz_text_main, in) != 0) mDie("pthread_create failed");
				in->gz_started = 1;
			}
		}
	}
	if (in->is_bam) {
		const uint8_t *p;
		int32_t l_text, n_ref, i;
		size_t at;
		in->bz.fp = in->fp;
		{
			struct stat sb;
			if (in->fp != stdin && fstat(fileno(in->fp), &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0 &&
			    !getenv("MSX_NO_MMAP")) {
				void *m = mmap(NULL, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fileno(in->fp), 0);
				if (m != MAP_FAILED) {
					static const uint8_t eof_block[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0,
					                                      0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
					in->bz.map = (const uint8_t *)m;
					in->bz.map_len = (size_t)sb.st_size;
					(void)madvise(m, (size_t)sb.st_size, MADV_SEQUENTIAL);
					/* htslib looks for the BGZF end-of-file marker of a seekable BAM when it reads the header and warns if it is
					 * missing (the reference's stderr then carries this line); the records are read all the same */
					if (sb.st_size < 28 || memcmp(in->bz.map + sb.st_size - 28, eof_block, 28) != 0)
						fprintf(stderr, "[W::bam_hdr_read] EOF marker is absent. The input is probably truncated\n");
				}
			}
		}
		if (!in->bz.map) {
			in->bz.ccap = (size_t)BGZF_BATCH * (BGZF_MAX + 1024);
			in->bz.fd = fileno(in->fp);
			in->bz.cur = -1;
			in->bz.rd_prefill = in->npre;                       /* the bytes looked at above */
			in->bz.rd_pre = in->pre;
#ifdef F_SETPIPE_SZ
			(void)fcntl(fileno(in->fp), F_SETPIPE_SZ, 1 << 20);      /* a pipe from `msamtools filter -bu`: fewer, larger reads */
#endif
		}
		if (!span_need(in, 12)) mDie("Cannot read header from %s", path);
		p = in->bz.span + in->bz.span_beg;
		if (memcmp(p, "BAM\1", 4) != 0) mDie("Cannot read header from %s", path);
		l_text = le32(p + 4);
		if (l_text < 0 || !span_need(in, 12 + (size_t)l_text)) mDie("Cannot read header from %s", path);
		p = in->bz.span + in->bz.span_beg;
		ks_put(&in->hdr.text, p + 8, strnlen((const char *)p + 8, (size_t)l_text));
		n_ref = le32(p + 8 + l_text);
		at = 12 + (size_t)l_text;
		for (i = 0; i < n_ref; i++) {
			int32_t l_name;
			if (!span_need(in, at + 4)) mDie("Cannot read header from %s", path);
			l_name = le32(in->bz.span + in->bz.span_beg + at);
			if (l_name <= 0 || !span_need(in, at + 8 + (size_t)l_name)) mDie("Cannot read header from %s", path);
			p = in->bz.span + in->bz.span_beg + at;
			hdr_add_target(&in->hdr, (const char *)p + 4, strnlen((const char *)p + 4, (size_t)l_name),
			               (uint32_t)le32(p + 4 + l_name));
			at += 8 + (size_t)l_name;
		}
		msh_span_consume(in, at);
	} else {
		ssize_t n;
		while ((n = getline(&in->line, &in->line_cap, in->fp)) > 0) {
			if (in->line[0] != '@') {
				ks_put(&in->pending, in->line, (size_t)n);
				in->has_pending = 1;
				break;
			}
			ks_put(&in->hdr.text, in->line, (size_t)n);
			if (in->line[n - 1] != '\n') ks_putc(&in->hdr.text, '\n');
		}
		if (n <= 0) gz_text_check(in);
		hdr_targets_from_text(&in->hdr);
	}
	return in;
}

const msh_hdr *msh_header(msh_in *in) { return &in->hdr; }

int msh_read(msh_in *in, kstr *rec) {
	if (in->is_bam) {
		int32_t bs;
		if (!span_need(in, 4)) {
			if (in->bz.span_end != in->bz.span_beg) mDie("Truncated BAM record");
			return -1;
		}
		bs = le32(in->bz.span + in->bz.span_beg);
		if (bs < 32) mDie("Corrupt BAM record (block_size %d)", bs);
		if (!span_need(in, 4 + (size_t)bs)) mDie("Truncated BAM record");
		rec->l = 0;
		ks_put(rec, in->bz.span + in->bz.span_beg + 4, (size_t)bs);
		msh_span_consume(in, 4 + (size_t)bs);
		return 0;
	} else {
		char *ln;
		ssize_t n;
		for (;;) {
			if (in->has_pending) {
				in->has_pending = 0;
				ln = in->pending.s;
				n = (ssize_t)in->pending.l;
			} else {
				n = getline(&in->line, &in->line_cap, in->fp);
				if (n <= 0) { gz_text_check(in); return -1; }
				ln = in->line;
			}
			while (n > 0 && (ln[n - 1] == '\n' || ln[n - 1] == '\r')) ln[--n] = 0;
			if (n == 0) continue;
			msh_sam_parse(&in->hdr, ln, rec);
			return 0;
		}
	}
}

void msh_close(msh_in *in) {
	int i;
	if (!in) return;
	if (in->gz_started) {
		/* an input read to its end: the decompressor has closed its side and returns.  One left earlier still has text to
		 * hand over: neither end of its pipe is closed under it (a write into a closed pipe is a signal) -- the process is on
		 * its way out in that case */
		if (in->text_eof || feof(in->fp)) {
			pthread_join(in->gz_thr, NULL);
			if (in->gz_src && in->gz_src != stdin) fclose(in->gz_src);
		} else {
			in->fp = NULL;
		}
	}
	if (in->fp && in->fp != stdin) fclose(in->fp);
	if (__atomic_load_n(&g_n2t.owner, __ATOMIC_ACQUIRE) == &in->hdr) {      /* (no thread parses this input any more) */
		pthread_mutex_lock(&g_n2t.mu);
		free(g_n2t.slot);
		g_n2t.slot = NULL;
		__atomic_store_n(&g_n2t.owner, (const msh_hdr *)NULL, __ATOMIC_RELEASE);
		pthread_mutex_unlock(&g_n2t.mu);
	}
	for (i = 0; i < in->hdr.n_targets; i++) free(in->hdr.target_name[i]);
	free(in->hdr.target_name);
	free(in->hdr.target_len);
	free(in->hdr.text.s);
	if (in->bz.rd_started && in->bz.rd_eof) {           /* (a reader still waiting for input is left to the process's end) */
		int i;
		pthread_join(in->bz.rd_thr, NULL);
		for (i = 0; i < RD_NBUF; i++) free(in->bz.rd_buf[i]);
	}
	if (in->bz.map) munmap((void *)in->bz.map, in->bz.map_len);
	free(in->bz.span);
	free(in->line);
	free(in->pending.s);
	free(in->pre);
	free(in);
}

/* ------------------------------------------------------------------------ */
/* output                                                                     */
/* ------------------------------------------------------------------------ */
struct msh_out {
	FILE *fp;
	int fd;              /* >= 0 (BAM output): written with write/writev, whole chunks of blocks per call */
	int is_pipe;         /* fd is a FIFO: finished blocks are handed over by reference (vmsplice), see msh_write_many */
	/* finished chunks of blocks are written by a thread of their own, so that the next chunk is built meanwhile */
	int wr_on, wr_n, wr_head, wr_busy, wr_quit;
	struct wchunk *wr_q[2];
	pthread_t wr_thr;
	pthread_mutex_t wr_mu;
	pthread_cond_t wr_cv_put, wr_cv_got;
	int mode;
	const msh_hdr *hdr;
	kstr line;
	uint8_t *ubuf;       /* BGZF payload being filled */
	uint32_t ulen;
	int level;
};
#define BGZF_PAYLOAD 0xff00
#define WCHUNK_BLOCKS 2048          /* blocks per chunk handed to the writer thread */
#define WSLOT (BGZF_MAX + 1024)     /* bytes reserved per block in a chunk */

static uint32_t bgzf_compress(uint8_t *out, const uint8_t *in, uint32_t n, int level);

/* a chunk of finished BGZF blocks on its way out */
struct wchunk {
	uint8_t *slots;          /* nblk blocks, WSLOT apart */
	size_t slots_bytes, slots_cap, nblk;
	uint32_t *slot_len;
	int mapped;              /* slots is an anonymous mapping of its own (vmsplice) rather than heap memory */
	size_t flat_len;         /* != 0: slots holds flat_len bytes of finished blocks back to back (msh_write_framed) */
};

/* Slot arrays of file output are reused: a fresh 136 MB allocation per chunk meant a page fault (and a zeroed page) for
 * every 4 KB written.  (Pipe output keeps its fresh mappings: handed-over pages must never be written again.) */
static struct { uint8_t *buf[4]; size_t bytes[4]; int n; pthread_mutex_t mu; } slot_pool = {{0}, {0}, 0, PTHREAD_MUTEX_INITIALIZER};
static uint8_t *slots_get(size_t bytes, size_t *got) {
	uint8_t *p = NULL;
	int i;
	pthread_mutex_lock(&slot_pool.mu);
	for (i = 0; i < slot_pool.n; i++)
		if (slot_pool.bytes[i] >= bytes) {
			p = slot_pool.buf[i]; *got = slot_pool.bytes[i];
			slot_pool.buf[i] = slot_pool.buf[slot_pool.n - 1]; slot_pool.bytes[i] = slot_pool.bytes[slot_pool.n - 1];
			slot_pool.n--;
			break;
		}
	pthread_mutex_unlock(&slot_pool.mu);
	if (!p) {
		/* anonymous memory advised for huge pages: a 136 MB array touched once per 4 KB page is 35 000 faults to fill and
		 * as many pages to give back when the process ends */
		const size_t al = (size_t)2 << 20;
		*got = bytes < (size_t)WCHUNK_BLOCKS * WSLOT ? (size_t)WCHUNK_BLOCKS * WSLOT : bytes;
		*got = (*got + al - 1) / al * al;
		p = (uint8_t *)mmap(NULL, *got, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
		if (p == (uint8_t *)MAP_FAILED) return NULL;
#ifdef MADV_HUGEPAGE
		(void)madvise(p, *got, MADV_HUGEPAGE);
#endif
	}
	return p;
}
static void slots_put(uint8_t *p, size_t bytes) {
	pthread_mutex_lock(&slot_pool.mu);
	if (slot_pool.n < 4) { slot_pool.buf[slot_pool.n] = p; slot_pool.bytes[slot_pool.n] = bytes; slot_pool.n++; p = NULL; }
	pthread_mutex_unlock(&slot_pool.mu);
	if (p) munmap(p, bytes);
}

static void chunk_write(msh_out *o, struct wchunk *c) {
	size_t q;
	if (c->flat_len) {                   /* one run of bytes (a pipe: handed over by reference, like the slots below) */
		uint8_t *p = c->slots;
		size_t want = c->flat_len;
		while (want) {
			struct iovec iv;
			ssize_t got;
			iv.iov_base = p; iv.iov_len = want;
			if (c->mapped && __atomic_load_n(&o->is_pipe, __ATOMIC_RELAXED)) {
				got = vmsplice(o->fd, &iv, 1, 0);
				if (got < 0 && (errno == EINVAL || errno == ENOSYS || errno == EBADF)) { __atomic_store_n(&o->is_pipe, 0, __ATOMIC_RELAXED); continue; }
			} else {
				got = write(o->fd, p, want);
			}
			if (got < 0 && errno == EINTR) continue;
			if (got <= 0) mDie("Write failed");
			p += got; want -= (size_t)got;
		}
		munmap(c->slots, c->slots_bytes);
		free(c);
		return;
	}
	/* the chunk's blocks in order, up to 512 of them per system call */
	for (q = 0; q < c->nblk;) {
		struct iovec iv[512];
		int niv = 0, v = 0;
		size_t want = 0;
		ssize_t got;
		for (; q < c->nblk && niv < 512; q++, niv++) {
			iv[niv].iov_base = c->slots + q * (BGZF_MAX + 1024);
			iv[niv].iov_len = c->slot_len[q];
			want += c->slot_len[q];
		}
		while (want) {
			if (c->mapped && __atomic_load_n(&o->is_pipe, __ATOMIC_RELAXED)) {
				got = vmsplice(o->fd, iv + v, (unsigned long)(niv - v), 0);
				if (got < 0 && (errno == EINVAL || errno == ENOSYS || errno == EBADF)) { __atomic_store_n(&o->is_pipe, 0, __ATOMIC_RELAXED); continue; }   /* not here: copy */
			} else {
				got = writev(o->fd, iv + v, niv - v);
			}
			if (got < 0 && errno == EINTR) continue;
			if (got <= 0) mDie("Write failed");
			want -= (size_t)got;
			while (got > 0 && (size_t)got >= iv[v].iov_len) { got -= (ssize_t)iv[v].iov_len; v++; }
			if (got > 0) { iv[v].iov_base = (uint8_t *)iv[v].iov_base + got; iv[v].iov_len -= (size_t)got; }
		}
	}
	if (c->mapped) munmap(c->slots, c->slots_bytes); else slots_put(c->slots, c->slots_cap);
	free(c->slot_len);
	free(c);
}

static void *writer_main(void *arg) {
	msh_out *o = (msh_out *)arg;
	pthread_mutex_lock(&o->wr_mu);
	for (;;) {
		struct wchunk *c;
		while (o->wr_n == 0 && !o->wr_quit) pthread_cond_wait(&o->wr_cv_put, &o->wr_mu);
		if (o->wr_n == 0) break;
		c = o->wr_q[o->wr_head];
		o->wr_head = (o->wr_head + 1) % 2;
		o->wr_n--;
		o->wr_busy = 1;
		pthread_cond_broadcast(&o->wr_cv_got);
		pthread_mutex_unlock(&o->wr_mu);
		chunk_write(o, c);
		pthread_mutex_lock(&o->wr_mu);
		o->wr_busy = 0;
		pthread_cond_broadcast(&o->wr_cv_got);
	}
	pthread_mutex_unlock(&o->wr_mu);
	return NULL;
}

/* everything handed to the writer thread so far is in the descriptor */
static void writer_drain(msh_out *o) {
	if (!o->wr_on) return;
	pthread_mutex_lock(&o->wr_mu);
	while (o->wr_n > 0 || o->wr_busy) pthread_cond_wait(&o->wr_cv_got, &o->wr_mu);
	pthread_mutex_unlock(&o->wr_mu);
}

static void writer_put(msh_out *o, struct wchunk *c) {
	if (!o->wr_on) {
		pthread_mutex_init(&o->wr_mu, NULL);
		pthread_cond_init(&o->wr_cv_put, NULL);
		pthread_cond_init(&o->wr_cv_got, NULL);
		if (pthread_create(&o->wr_thr, NULL, writer_main, o) != 0) mDie("Cannot start the writer thread");
		o->wr_on = 1;
	}
	pthread_mutex_lock(&o->wr_mu);
	while (o->wr_n == 2) pthread_cond_wait(&o->wr_cv_got, &o->wr_mu);
	o->wr_q[(o->wr_head + o->wr_n) % 2] = c;
	o->wr_n++;
	pthread_cond_signal(&o->wr_cv_put);
	pthread_mutex_unlock(&o->wr_mu);
}

static void out_bytes(msh_out *o, const void *p, size_t n) {
	writer_drain(o);                     /* (bytes written here follow whatever the writer thread still holds) */
	if (o->fd >= 0) {
		const uint8_t *s = (const uint8_t *)p;
		while (n) {
			ssize_t k = write(o->fd, s, n);
			if (k < 0 && errno == EINTR) continue;
			if (k <= 0) mDie("Write failed");
			s += k; n -= (size_t)k;
		}
	} else if (fwrite(p, 1, n, o->fp) != n) {
		mDie("Write failed");
	}
}

static void bgz_flush_block(msh_out *o) {
	uint8_t out[BGZF_MAX + 1024];
	uint32_t total = bgzf_compress(out, o->ubuf, o->ulen, o->level);
	out_bytes(o, out, total);
	o->ulen = 0;
}

static void bgz_write(msh_out *o, const void *p, size_t n) {
	const uint8_t *s = (const uint8_t *)p;
	while (n) {
		size_t room = BGZF_PAYLOAD - o->ulen, k = n < room ? n : room;
		memcpy(o->ubuf + o->ulen, s, k);
		o->ulen += (uint32_t)k;
		s += k;
		n -= k;
		if (o->ulen == BGZF_PAYLOAD) bgz_flush_block(o);
	}
}

msh_out *msh_out_open(FILE *fp, int mode, const msh_hdr *h, const char *hdr_text) {
	msh_out *o = (msh_out *)calloc(1, sizeof(*o));
	if (!o) mDie("Out of memory");
	o->fp = fp;
	o->mode = mode;
	o->hdr = h;
	o->fd = -1;
	setvbuf(fp, NULL, _IOFBF, 1 << 20);
	if (mode == MSH_OUT_BAM || mode == MSH_OUT_UBAM) {
		fflush(fp);
		o->fd = fileno(fp);              /* nothing of the BAM stream goes through stdio */
#ifdef F_SETPIPE_SZ
		(void)fcntl(o->fd, F_SETPIPE_SZ, 1 << 20);       /* a pipe into `msamtools profile -`: fewer, larger transfers */
#endif
		{
			struct stat st;
			const char *e = getenv("MSX_VMSPLICE");
			o->is_pipe = fstat(o->fd, &st) == 0 && S_ISFIFO(st.st_mode) && !(e && atoi(e) == 0);
		}
	}
	if (mode == MSH_OUT_BAM || mode == MSH_OUT_UBAM) {
		kstr b = {0, 0, 0};
		int32_t i;
		size_t tl = strlen(hdr_text);
		o->ubuf = (uint8_t *)malloc(BGZF_MAX);
		o->level = mode == MSH_OUT_UBAM ? 0 : Z_DEFAULT_COMPRESSION;
		if (mode == MSH_OUT_BAM) {           /* MSX_BGZF_LEVEL=1..9: trade file size for speed (-b is deflate-bound: level 6 by default, as htslib) */
			const char *e = getenv("MSX_BGZF_LEVEL");
			const int lv = e ? atoi(e) : 0;
			if (lv >= 1 && lv <= 9) o->level = lv;
		}
		ks_put(&b, "BAM\1", 4);
		put_le32(&b, (uint32_t)tl);
		ks_put(&b, hdr_text, tl);
		put_le32(&b, (uint32_t)h->n_targets);
		for (i = 0; i < h->n_targets; i++) {
			size_t nl = strlen(h->target_name[i]) + 1;
			put_le32(&b, (uint32_t)nl);
			ks_put(&b, h->target_name[i], nl);
			put_le32(&b, h->target_len[i]);
		}
		bgz_write(o, b.s, b.l);
		if (o->ulen) bgz_flush_block(o);      /* header in its own block(s), as htslib does */
		free(b.s);
	} else if (mode == MSH_OUT_SAM_HDR) {
		out_bytes(o, hdr_text, strlen(hdr_text));
	}
	return o;
}

void msh_write(msh_out *o, const uint8_t *rec, size_t len) {
	if (o->mode == MSH_OUT_BAM || o->mode == MSH_OUT_UBAM) {
		uint8_t b4[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
		if (o->ulen + 4 + len > BGZF_PAYLOAD && o->ulen) bgz_flush_block(o);   /* keep records whole when they fit */
		bgz_write(o, b4, 4);
		bgz_write(o, rec, len);
	} else {
		o->line.l = 0;
		msh_sam_format(o->hdr, rec, len, &o->line);
		ks_putc(&o->line, '\n');
		out_bytes(o, o->line.s, o->line.l);
	}
}

/* ---- bulk, multi-threaded writer ------------------------------------------------
 * Writes the records base + rec_off[idx[k]] (each preceded by its 4-byte
 * block_size, as in the BAM stream) for k = 0..n-1.  BAM: the records are packed
 * greedily into BGZF blocks (whole records per block), blocks are deflated in
 * parallel and written in order.  SAM: lines are formatted in parallel. */
#define WCHUNK_LINES 262144

typedef struct {
	msh_out *o;
	const uint8_t *base;
	const size_t *rec_off;
	const int32_t *idx;
	/* BAM */
	size_t nblk;
	const size_t *first;    /* [nblk+1] first emitted-record index of each block */
	uint8_t *slots;         /* nblk * SLOT bytes */
	uint32_t *slot_len;
	/* SAM */
	size_t lo, hi;
	kstr *lines;            /* one per thread */
} wjob;

static const uint8_t BGZF_HEAD[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};

static void bgzf_finish(uint8_t *out, uint32_t clen, uint32_t crc, uint32_t n) {
	const uint32_t total = 18 + clen + 8;
	memcpy(out, BGZF_HEAD, 16);
	out[16] = (uint8_t)((total - 1) & 0xff);
	out[17] = (uint8_t)((total - 1) >> 8);
	out[18 + clen + 0] = (uint8_t)crc; out[18 + clen + 1] = (uint8_t)(crc >> 8);
	out[18 + clen + 2] = (uint8_t)(crc >> 16); out[18 + clen + 3] = (uint8_t)(crc >> 24);
	out[18 + clen + 4] = (uint8_t)n; out[18 + clen + 5] = (uint8_t)(n >> 8);
	out[18 + clen + 6] = (uint8_t)(n >> 16); out[18 + clen + 7] = (uint8_t)(n >> 24);
}

static uint32_t bgzf_compress(uint8_t *out, const uint8_t *in, uint32_t n, int level) {
	/* one deflate state per thread (a quarter of a megabyte each): created once, reset per block */
	static __thread z_stream zs;
	static __thread int zs_level = -100;
	uint32_t clen;
	if (zs_level != level) {
		if (zs_level != -100) deflateEnd(&zs);
		memset(&zs, 0, sizeof zs);
		if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) mDie("zlib deflateInit2 failed");
		zs_level = level;
	} else if (deflateReset(&zs) != Z_OK) {
		mDie("zlib deflateReset failed");
	}
	zs.next_in = (Bytef *)in;
	zs.avail_in = n;
	zs.next_out = out + 18;
	zs.avail_out = WSLOT - 18 - 8;
	if (deflate(&zs, Z_FINISH) != Z_STREAM_END) mDie("BGZF deflate failed");
	clen = (uint32_t)zs.total_out;
	bgzf_finish(out, clen, msh_crc32(in, n), n);
	return 18 + clen + 8;
}

static void wbam_worker(void *arg, int tid, int nth) {
	wjob *w = (wjob *)arg;
	const int stored = w->o->level == 0;
	static __thread uint8_t *payload = NULL;
	size_t k, r;
	if (!stored && !payload && !(payload = (uint8_t *)malloc(BGZF_MAX))) mDie("Out of memory");
	for (k = (size_t)tid; k < w->nblk; k += (size_t)nth) {
		uint8_t *slot = w->slots + k * WSLOT;
		/* -u: one stored deflate block, the records gathered straight into their place */
		uint8_t *dst = stored ? slot + 18 + 5 : payload;
		uint32_t n = 0;
		for (r = w->first[k]; r < w->first[k + 1]; r++) {
			size_t i = (size_t)w->idx[r], sz = w->rec_off[i + 1] - w->rec_off[i];
			memcpy(dst + n, w->base + w->rec_off[i], sz);
			n += (uint32_t)sz;
		}
		if (stored) {
			slot[18] = 1; slot[19] = (uint8_t)n; slot[20] = (uint8_t)(n >> 8); slot[21] = (uint8_t)~n; slot[22] = (uint8_t)(~n >> 8);
			bgzf_finish(slot, 5 + n, msh_crc32(dst, n), n);
			w->slot_len[k] = 18 + 5 + n + 8;
		} else {
			w->slot_len[k] = bgzf_compress(slot, payload, n, w->o->level);
		}
	}
}

static void wsam_worker(void *arg, int tid, int nth) {
	wjob *w = (wjob *)arg;
	size_t n = w->hi - w->lo, a = w->lo + n * (size_t)tid / (size_t)nth, b = w->lo + n * (size_t)(tid + 1) / (size_t)nth, r;
	kstr *ln = &w->lines[tid];
	ln->l = 0;
	for (r = a; r < b; r++) {
		size_t i = (size_t)w->idx[r];
		msh_sam_format(w->o->hdr, w->base + w->rec_off[i] + 4, w->rec_off[i + 1] - w->rec_off[i] - 4, ln);
		ks_putc(ln, '\n');
	}
}

void msh_write_many(msh_out *o, const uint8_t *base, const size_t *rec_off, const int32_t *idx, size_t n) {
	int nth = msh_threads();
	wjob w;
	size_t r;
	if (n == 0) return;
	memset(&w, 0, sizeof w);
	w.o = o; w.base = base; w.rec_off = rec_off; w.idx = idx;
	if (o->mode == MSH_OUT_BAM || o->mode == MSH_OUT_UBAM) {
		for (r = 0; r < n; r++)          /* a record larger than one block: leave everything to the serial writer */
			if (rec_off[(size_t)idx[r] + 1] - rec_off[(size_t)idx[r]] > BGZF_PAYLOAD) {
				for (r = 0; r < n; r++) {
					size_t i = (size_t)idx[r];
					msh_write(o, base + rec_off[i] + 4, rec_off[i + 1] - rec_off[i] - 4);
				}
				return;
			}
		{
			/* plan: whole records per block, greedily; the last, partly filled block stays in the writer's buffer.
			 * What the previous call left there becomes block 0 of this call's first chunk. */
			size_t cap = 1024, nb = 0, cur = 0, done, *first = (size_t *)malloc((cap + 2) * sizeof(size_t));
			uint8_t *carry = NULL;
			uint32_t carry_len = 0;
			if (!first) mDie("Out of memory");
			if (o->ulen) {
				if (!(carry = (uint8_t *)malloc(BGZF_MAX + 1024))) mDie("Out of memory");
				carry_len = bgzf_compress(carry, o->ubuf, o->ulen, o->level);
				o->ulen = 0;
			}
			first[0] = 0;
			for (r = 0; r < n; r++) {
				size_t sz = rec_off[(size_t)idx[r] + 1] - rec_off[(size_t)idx[r]];
				if (cur + sz > BGZF_PAYLOAD) {
					if (nb + 2 > cap) { cap *= 2; first = (size_t *)realloc(first, (cap + 2) * sizeof(size_t)); if (!first) mDie("Out of memory"); }
					first[++nb] = r;
					cur = 0;
				}
				cur += sz;
			}
			for (r = first[nb]; r < n; r++) {
				size_t i = (size_t)idx[r], sz = rec_off[i + 1] - rec_off[i];
				memcpy(o->ubuf + o->ulen, base + rec_off[i], sz);
				o->ulen += (uint32_t)sz;
			}
			if (nb == 0 && carry) {              /* nothing but the carried block to write */
				out_bytes(o, carry, carry_len);
				free(carry);
				carry = NULL;
			}
			/* Chunks of up to WCHUNK_BLOCKS blocks: built by all threads, then handed to the writer thread, which
			 * writes them in order while the next chunk (of this call or the next) is built.
			 * Into a pipe the finished blocks are not copied but handed over by reference (vmsplice): the kernel
			 * pins their pages for the reader.  Such pages must never be written again, so every chunk gets a
			 * fresh anonymous mapping that is unmapped as soon as it has been handed over -- the pipe's
			 * references keep the pages alive until they are read, whatever this process does meanwhile. */
			for (done = 0; done < nb;) {
				struct wchunk *c = (struct wchunk *)calloc(1, sizeof(*c));
				const size_t extra = carry ? 1 : 0;
				size_t take = nb - done < WCHUNK_BLOCKS - extra ? nb - done : WCHUNK_BLOCKS - extra;
				if (!c) mDie("Out of memory");
				c->nblk = take + extra;
				c->slots_bytes = c->nblk * WSLOT;
				c->mapped = __atomic_load_n(&o->is_pipe, __ATOMIC_RELAXED);
				c->slots = c->mapped ? (uint8_t *)mmap(NULL, c->slots_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0)
				                     : slots_get(c->slots_bytes, &c->slots_cap);
				c->slot_len = (uint32_t *)malloc(c->nblk * sizeof(uint32_t));
				if (!c->slots || c->slots == (uint8_t *)MAP_FAILED || !c->slot_len) mDie("Out of memory");
				if (carry) {
					memcpy(c->slots, carry, carry_len);
					c->slot_len[0] = carry_len;
					free(carry);
					carry = NULL;
				}
				w.nblk = take;
				w.first = first + done;
				w.slots = c->slots + extra * WSLOT;
				w.slot_len = c->slot_len + extra;
				msh_parallel(nth < (int)w.nblk ? nth : (int)w.nblk, wbam_worker, &w);
				writer_put(o, c);
				done += take;
			}
			free(first);
		}
	} else {
		int t;
		w.lines = (kstr *)calloc((size_t)nth, sizeof(kstr));
		for (w.lo = 0; w.lo < n; w.lo = w.hi) {
			w.hi = w.lo + WCHUNK_LINES < n ? w.lo + WCHUNK_LINES : n;
			msh_parallel(nth, wsam_worker, &w);
			for (t = 0; t < nth; t++)
				if (w.lines[t].l) out_bytes(o, w.lines[t].s, w.lines[t].l);
		}
		for (t = 0; t < nth; t++) free(w.lines[t].s);
		free(w.lines);
	}
}

/* The same for a ready-made record stream (records with their block_size prefixes, back to back -- what
 * msx_unpack_emit returns): payloads of BGZF_PAYLOAD bytes cut where they fall (a record may straddle two blocks, as
 * the format allows), blocks built by all threads, written in order by the writer thread. */
typedef struct {
	msh_out *o;
	const uint8_t *bytes;
	size_t nblk;
	uint8_t *slots;
	uint32_t *slot_len;
} sjob;

static void wstream_worker(void *arg, int tid, int nth) {
	sjob *w = (sjob *)arg;
	const int stored = w->o->level == 0;
	size_t k;
	for (k = (size_t)tid; k < w->nblk; k += (size_t)nth) {
		uint8_t *slot = w->slots + k * WSLOT;
		const uint8_t *src = w->bytes + k * BGZF_PAYLOAD;
		const uint32_t n = BGZF_PAYLOAD;
		if (stored) {
			memcpy(slot + 18 + 5, src, n);
			slot[18] = 1; slot[19] = (uint8_t)n; slot[20] = (uint8_t)(n >> 8); slot[21] = (uint8_t)~n; slot[22] = (uint8_t)(~n >> 8);
			bgzf_finish(slot, 5 + n, msh_crc32(src, n), n);
			w->slot_len[k] = 18 + 5 + n + 8;
		} else {
			w->slot_len[k] = bgzf_compress(slot, src, n, w->o->level);
		}
	}
}

void msh_write_stream(msh_out *o, const uint8_t *bytes, size_t n) {
	const int nth = msh_threads();
	size_t nb, done;
	if (n == 0) return;
	if (o->mode != MSH_OUT_BAM && o->mode != MSH_OUT_UBAM) {          /* text: record by record */
		size_t p = 0;
		while (p + 4 <= n) {
			const size_t len = (size_t)(uint32_t)le32(bytes + p);
			msh_write(o, bytes + p + 4, len);
			p += 4 + len;
		}
		return;
	}
	uint8_t *carry = NULL;
	uint32_t carry_len = 0;
	if (o->ulen) {                           /* top up the block the previous call left open */
		const size_t room = BGZF_PAYLOAD - o->ulen, k = n < room ? n : room;
		memcpy(o->ubuf + o->ulen, bytes, k);
		o->ulen += (uint32_t)k;
		bytes += k;
		n -= k;
		if (o->ulen == BGZF_PAYLOAD) {       /* full: it travels as block 0 of this call's first chunk */
			if (!(carry = (uint8_t *)malloc(BGZF_MAX + 1024))) mDie("Out of memory");
			carry_len = bgzf_compress(carry, o->ubuf, o->ulen, o->level);
			o->ulen = 0;
		}
	}
	nb = n / BGZF_PAYLOAD;
	if (nb == 0 && carry) {
		out_bytes(o, carry, carry_len);
		free(carry);
		carry = NULL;
	}
	for (done = 0; done < nb;) {
		struct wchunk *c = (struct wchunk *)calloc(1, sizeof(*c));
		sjob w;
		const size_t extra = carry ? 1 : 0;
		const size_t take = nb - done < WCHUNK_BLOCKS - extra ? nb - done : WCHUNK_BLOCKS - extra;
		if (!c) mDie("Out of memory");
		c->nblk = take + extra;
		c->slots_bytes = c->nblk * WSLOT;
		c->mapped = __atomic_load_n(&o->is_pipe, __ATOMIC_RELAXED);
		c->slots = c->mapped ? (uint8_t *)mmap(NULL, c->slots_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0)
		                     : slots_get(c->slots_bytes, &c->slots_cap);
		c->slot_len = (uint32_t *)malloc(c->nblk * sizeof(uint32_t));
		if (!c->slots || c->slots == (uint8_t *)MAP_FAILED || !c->slot_len) mDie("Out of memory");
		if (carry) {
			memcpy(c->slots, carry, carry_len);
			c->slot_len[0] = carry_len;
			free(carry);
			carry = NULL;
		}
		w.o = o; w.bytes = bytes + done * BGZF_PAYLOAD; w.nblk = take; w.slots = c->slots + extra * WSLOT; w.slot_len = c->slot_len + extra;
		msh_parallel(nth < (int)take ? nth : (int)take, wstream_worker, &w);
		writer_put(o, c);
		done += take;
	}
	if (n > nb * BGZF_PAYLOAD) {             /* the rest waits in the open block */
		const size_t rest = n - nb * BGZF_PAYLOAD;
		memcpy(o->ubuf, bytes + nb * BGZF_PAYLOAD, rest);
		o->ulen = (uint32_t)rest;
	}
}

/* Finished BGZF blocks, back to back, as the device framed them (msx_unpack_emit_gather_bgzf): nothing is copied or
 * summed here.  The block a host-side writer call left open goes out first, as a short block of its own.
 * Regular file: written from where they are by this one thread (pwrite()s of disjoint ranges from 4 / 8 / 16 threads
 * measured 11.6 / 11.7 / 10.9 GB/s against 12.1 for one write(): the inode lock -- profiles/round4/write_rate.log; that path,
 * MSX_WRITE_THREADS, was taken out in round 6).  Pipe: the bytes are copied once, by all threads, into a fresh
 * mapping that is handed over by reference (the caller's buffer is page-locked and reused, so it cannot be). */
typedef struct { const uint8_t *src; uint8_t *dst; size_t n; } fjob;
static void framed_copy_worker(void *arg, int tid, int nth) {
	const fjob *j = (const fjob *)arg;
	const size_t lo = j->n * (size_t)tid / (size_t)nth, hi = j->n * (size_t)(tid + 1) / (size_t)nth;
	memcpy(j->dst + lo, j->src + lo, hi - lo);
}

void msh_write_framed(msh_out *o, const uint8_t *blocks, size_t n) {
	fjob J;
	if (n == 0) return;
	if (o->mode != MSH_OUT_BAM && o->mode != MSH_OUT_UBAM) mDie("msh_write_framed: not a BAM output");
	if (o->ulen) bgz_flush_block(o);
	writer_drain(o);
	memset(&J, 0, sizeof J);
	J.src = blocks; J.n = n;
	if (__atomic_load_n(&o->is_pipe, __ATOMIC_RELAXED)) {
		struct wchunk *c = (struct wchunk *)calloc(1, sizeof(*c));
		int nth = msh_threads();
		if (!c) mDie("Out of memory");
		c->slots_bytes = (n + 4095) & ~(size_t)4095;
		c->slots = (uint8_t *)mmap(NULL, c->slots_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
		if (c->slots == (uint8_t *)MAP_FAILED) mDie("Out of memory");
		c->mapped = 1;
		c->flat_len = n;
		J.dst = c->slots;
		if ((size_t)nth > n / 65536 + 1) nth = (int)(n / 65536 + 1);
		msh_parallel(nth, framed_copy_worker, &J);
		writer_put(o, c);
		return;
	}
	while (n) {
		ssize_t k = write(o->fd, blocks, n);
		if (k < 0 && errno == EINTR) continue;
		if (k <= 0) mDie("Write failed");
		blocks += k; n -= (size_t)k;
	}
}

/* everything handed over so far is in the descriptor; the open block is NOT written (a fatal error follows: the
 * reference dies with its last buffer unwritten too) */
void msh_out_drain(msh_out *o) {
	if (!o) return;
	writer_drain(o);
	fflush(o->fp);
}

void msh_out_close(msh_out *o) {
	if (!o) return;
	if (o->mode == MSH_OUT_BAM || o->mode == MSH_OUT_UBAM) {
		static const uint8_t eof_block[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0,
		                                      0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
		if (o->ulen) bgz_flush_block(o);
		out_bytes(o, eof_block, 28);
	}
	if (o->wr_on) {
		writer_drain(o);
		pthread_mutex_lock(&o->wr_mu);
		o->wr_quit = 1;
		pthread_cond_signal(&o->wr_cv_put);
		pthread_mutex_unlock(&o->wr_mu);
		pthread_join(o->wr_thr, NULL);
	}
	fflush(o->fp);
	free(o->ubuf);
	free(o->line.s);
	free(o);
}
