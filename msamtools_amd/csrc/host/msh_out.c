/*
 * msh_out.c -- output: SAM text, BAM in stored or deflated BGZF blocks, device-framed blocks written as they are (htslib's
 * sam_write1 / bgzf_write under msam_helper.c:270-272, modes msam_filter.c:464-470).  Split out of msh_io.c in round 6.
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
		msh_put_le32(&b, (uint32_t)tl);
		ks_put(&b, hdr_text, tl);
		msh_put_le32(&b, (uint32_t)h->n_targets);
		for (i = 0; i < h->n_targets; i++) {
			size_t nl = strlen(h->target_name[i]) + 1;
			msh_put_le32(&b, (uint32_t)nl);
			ks_put(&b, h->target_name[i], nl);
			msh_put_le32(&b, h->target_len[i]);
		}
		/* (on all threads: a million @SQ lines are 70 MB of header, and deflating them block after block on this thread held
		 * the writer for 0.15 s of a 0.67 s command while the device stage ran out of output buffers behind it) */
		msh_write_stream(o, (const uint8_t *)b.s, b.l);
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

/* Nothing written so far stays in this process: the open BGZF block goes out as a short block of its own, stdio's buffer
 * is flushed.  Called by a writer that has to wait for its next batch (a slow producer must see its output). */
void msh_out_flush(msh_out *o) {
	if (!o) return;
	if ((o->mode == MSH_OUT_BAM || o->mode == MSH_OUT_UBAM) && o->ulen) bgz_flush_block(o);
	writer_drain(o);
	fflush(o->fp);
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

