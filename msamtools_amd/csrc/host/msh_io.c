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
#include "msh_io_int.h"

/* ------------------------------------------------------------------------ */
/* errors, strings                                                            */
/* ------------------------------------------------------------------------ */
pthread_t msh_main_thread;
int msh_main_thread_set;
void (*msh_exit_hook)(int rc);     /* msh_main.c, MSX_DETACH=1: called with the exit code by whatever is about to _exit */

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
	if (msh_exit_hook) msh_exit_hook(EXIT_FAILURE);
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

/* A large heap block asked to be backed by huge pages (where the system leaves that to madvise): the 2 MB-aligned part of it.
 * The text path's buffers -- parsed records per thread, a batch's bytes, the SoA arrays -- are a couple of gigabytes; in 4 KB
 * pages they cost 700 000 page faults while the first batches are parsed and half a second of tear-down after the
 * command's last line (SAM text, 20 M records: 1.81 s, 0.48 of them after exit). */
void msh_huge_hint(void *p, size_t n) {
#ifdef MADV_HUGEPAGE
	const uintptr_t al = (uintptr_t)2 << 20;
	const uintptr_t a = ((uintptr_t)p + al - 1) & ~(al - 1), b = ((uintptr_t)p + n) & ~(al - 1);
	if (p && n >= ((size_t)8 << 20) && b > a) (void)madvise((void *)a, (size_t)(b - a), MADV_HUGEPAGE);
#else
	(void)p; (void)n;
#endif
}

void ks_reserve(kstr *k, size_t extra) {
	if (k->l + extra + 1 > k->m) {
		size_t m = k->m ? k->m : 64;
		while (m < k->l + extra + 1) m += m >> 1;
		k->s = (char *)realloc(k->s, m);
		if (!k->s) mDie("Out of memory");
		k->m = m;
		msh_huge_hint(k->s, m);
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

void msh_put_le32(kstr *k, uint32_t v) {
	uint8_t b[4] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)};
	ks_put(k, b, 4);
}
void msh_put_le16(kstr *k, uint32_t v) {
	uint8_t b[2] = {(uint8_t)v, (uint8_t)(v >> 8)};
	ks_put(k, b, 2);
}

/* ------------------------------------------------------------------------ */
/* header                                                                     */
/* ------------------------------------------------------------------------ */
void msh_hdr_add_target(msh_hdr *h, const char *name, size_t nl, uint32_t len) {
	h->target_name = (char **)realloc(h->target_name, sizeof(char *) * (size_t)(h->n_targets + 1));
	h->target_len = (uint32_t *)realloc(h->target_len, sizeof(uint32_t) * (size_t)(h->n_targets + 1));
	h->target_name[h->n_targets] = (char *)malloc(nl + 1);
	memcpy(h->target_name[h->n_targets], name, nl);
	h->target_name[h->n_targets][nl] = 0;
	h->target_len[h->n_targets] = len;
	h->n_targets++;
}

/* reference dictionary from @SQ lines (SAM text input) */
void msh_hdr_targets_from_text(msh_hdr *h) {
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
			if (sn) msh_hdr_add_target(h, sn, snl, ln);
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
void msh_hdr_forget_names(const msh_hdr *h) {
	if (__atomic_load_n(&g_n2t.owner, __ATOMIC_ACQUIRE) == h) {      /* (no thread parses this input any more) */
		pthread_mutex_lock(&g_n2t.mu);
		free(g_n2t.slot);
		g_n2t.slot = NULL;
		__atomic_store_n(&g_n2t.owner, (const msh_hdr *)NULL, __ATOMIC_RELEASE);
		pthread_mutex_unlock(&g_n2t.mu);
	}
}
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
size_t msh_aux_type_size(int t) {
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
	size_t fs = msh_aux_type_size(ty);
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
		size_t es = msh_aux_type_size(t[1]);
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

