/*
 * msh_in.c -- input: the BGZF container (blocks read, inflated in parallel into one span of BAM bytes, or handed on
 * compressed for the device), BAM records, SAM text and gzip'd SAM text behind one reader (htslib's sam_open / sam_read1 under
 * msam_helper.c:196-268).  Split out of msh_io.c in round 6.
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
#include <poll.h>
#include <time.h>

/* A producer that trickles (an aligner writing SAM or BAM into the pipe as it goes): the reference writes a read's alignments
 * as soon as the next read's first record has arrived (msam_filter.c:120-125,186); a batch pipeline that waits for its
 * 96 MB would sit on them.  So a batch also ends when the input has had nothing to give for this long (MSX_IDLE_MS,
 * default 50; 0: never): the reader thread hands over what it holds, the text reader returns what it has, and
 * msh_input_ready tells the decode stage not to wait for more. */
int msh_idle_ms(void) {
	static int v = -1;
	int c = __atomic_load_n(&v, __ATOMIC_RELAXED);
	if (c < 0) {
		const char *e = getenv("MSX_IDLE_MS");
		c = e ? atoi(e) : 50;
		if (c < 0) c = 0;
		__atomic_store_n(&v, c, __ATOMIC_RELAXED);
	}
	return c;
}
static int fd_readable_within(int fd, int ms) {       /* 1: bytes (or the end of the input) can be read now */
	struct pollfd pf;
	int r;
	pf.fd = fd; pf.events = POLLIN; pf.revents = 0;
	do r = poll(&pf, 1, ms); while (r < 0 && errno == EINTR);
	return r != 0;
}

/* ------------------------------------------------------------------------ */
/* BGZF reader: batches of raw blocks inflated in parallel into one           */
/* contiguous "span" of BAM bytes                                             */
/* ------------------------------------------------------------------------ */
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

/* one block's DEFLATE stream into `out` (isize bytes expected, CRC-32 `crc`); NULL, or what is wrong with it */
static const char *inflate_payload_try(const uint8_t *data, size_t dlen, uint8_t *out, uint32_t isize, uint32_t crc) {
	/* one stream per thread, reset between blocks: initialising one per block means an allocation per
	 * block, and with a hundred threads those serialise inside the allocator */
	static __thread z_stream zs;
	static __thread int zs_ready = 0;
	static int fast_flag = -1;
	int fast = __atomic_load_n(&fast_flag, __ATOMIC_RELAXED);
	if (isize == 0) return NULL;
	if (fast < 0) {
		fast = !getenv("MSX_NO_FAST_INFLATE");
		__atomic_store_n(&fast_flag, fast, __ATOMIC_RELAXED);
	}
	/* the decoder of msh_inflate.c first (twice zlib's speed on BAM records); whatever it does not vouch for,
	 * and whatever fails the CRC afterwards, is decoded again by zlib, whose verdict stands */
	if (fast && msh_fast_inflate(data, dlen, out, isize) && msh_crc32(out, isize) == crc) return NULL;
	if (!zs_ready) {
		memset(&zs, 0, sizeof zs);
		if (inflateInit2(&zs, -15) != Z_OK) return "zlib inflateInit2 failed";
		zs_ready = 1;
	} else if (inflateReset(&zs) != Z_OK) {
		return "zlib inflateReset failed";
	}
	zs.next_in = (Bytef *)data;
	zs.avail_in = (uInt)dlen;
	zs.next_out = out;
	zs.avail_out = isize;
	if (inflate(&zs, Z_FINISH) != Z_STREAM_END || zs.total_out != isize) return "Corrupt BGZF block (inflate failed)";
	if (msh_crc32(out, isize) != crc) return "Corrupt BGZF block (CRC mismatch)";
	return NULL;
}
static void inflate_payload(const uint8_t *data, size_t dlen, uint8_t *out, uint32_t isize, uint32_t crc) {
	const char *err = inflate_payload_try(data, dlen, out, isize, crc);
	if (err) mDie("%s", err);
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
			ssize_t k;
			/* (the parser is waiting and the producer has gone quiet: what is here goes over now) */
			if (n > 0 && msh_idle_ms() > 0 && __atomic_load_n(&b->rd_wait, __ATOMIC_RELAXED) && !fd_readable_within(b->fd, msh_idle_ms())) break;
			k = read(b->fd, b->rd_buf[slot] + RD_HEAD + n, b->ccap - n);
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
	kstr carry_text;     /* text read but not parsed yet: the line the last chunk cut (at first: what the header scan left) */
	int text_in_carry;
	int text_eof;
	int idle_hit;        /* the last chunk came early: the producer had gone quiet (msh_input_ready) */
	char *tr_buf[2];     /* the reader thread's two chunks (text_reader_main) */
	size_t tr_len[2];
	int tr_full[2], tr_idle[2], tr_head, tr_eof, tr_started;
	pthread_t tr_thr;
	pthread_mutex_t tr_mu;
	pthread_cond_t tr_cv_full, tr_cv_free;
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
/* bytes of the input file, if it is a regular file (-1: a pipe, a terminal, ...) */
int64_t msh_in_bytes(const msh_in *in) {
	struct stat st;
	if (!in || !in->fp || fstat(fileno(in->fp), &st) != 0 || !S_ISREG(st.st_mode)) return -1;
	return (int64_t)st.st_size;
}

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
		msh_huge_hint(*buf, nc);
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

/* The text's reader: a thread of its own keeps the descriptor drained into two chunk buffers (16 MB each, a megabyte of headroom
 * in front for the line the chunk before it cut), so that reading chunk k + 1 -- one thread's read(2), 8-10 GB/s from a pipe --
 * runs beside the parsing of chunk k on all threads.  (Round 5 read, parsed and copied one after the other: 2.1 GB/s of text
 * on 16 cores, a third of it the read, most of the rest ONE thread copying the parsed records together.)  A chunk is handed
 * over when it is full, at the end of the input, or when the producer has been quiet for msh_idle_ms(). */
#define TR_HEAD ((size_t)1 << 20)
static void *text_reader_main(void *arg) {
	msh_in *in = (msh_in *)arg;
	const int fd = fileno(in->fp), idle = msh_idle_ms();
	const size_t chunk = SAM_CHUNK;
	int slot = 0;
	for (;;) {
		size_t n = 0;
		int eof = 0, went_idle = 0;
		pthread_mutex_lock(&in->tr_mu);
		while (in->tr_full[slot]) pthread_cond_wait(&in->tr_cv_free, &in->tr_mu);
		pthread_mutex_unlock(&in->tr_mu);
		while (n < chunk) {
			ssize_t k;
			if (n > 0 && idle > 0 && !fd_readable_within(fd, idle)) { went_idle = 1; break; }
			k = read(fd, in->tr_buf[slot] + TR_HEAD + n, chunk - n);
			if (k < 0 && errno == EINTR) continue;
			if (k < 0) mDie("Read failed");
			if (k == 0) { eof = 1; break; }
			n += (size_t)k;
		}
		pthread_mutex_lock(&in->tr_mu);
		in->tr_len[slot] = n;
		in->tr_idle[slot] = went_idle;
		in->tr_full[slot] = 1;
		if (eof) in->tr_eof = 1;
		pthread_cond_signal(&in->tr_cv_full);
		pthread_mutex_unlock(&in->tr_mu);
		if (eof) break;
		slot ^= 1;
	}
	return NULL;
}

typedef struct { uint8_t *dst; const kstr *out; const size_t *off; } samcopy_job;
static void samcopy_worker(void *arg, int tid, int nth) {
	const samcopy_job *j = (const samcopy_job *)arg;
	(void)nth;
	if (j->out[tid].l) memcpy(j->dst + j->off[tid], j->out[tid].s, j->out[tid].l);
}

size_t msh_sam_append(msh_in *in, uint8_t **buf, size_t *len, size_t *cap) {
	static sam_job J;
	size_t end, total = 0, tlen, off[MSH_MAX_THREADS + 1];
	char *text;
	int nth = msh_threads(), t, slot, last;
	if (in->is_bam) mDie("msh_sam_append on BAM input");
	if (in->text_eof) return 0;
	if (!in->tr_started) {
		/* what the header scan left behind -- the first record line, and whatever stdio still holds -- leads the text;
		 * from here on the descriptor is read by the thread above, never through stdio again */
		const size_t pend = (size_t)(in->fp->_IO_read_end - in->fp->_IO_read_ptr);
		int i;
		in->carry_text.l = 0;
		if (in->has_pending) {
			in->has_pending = 0;
			ks_put(&in->carry_text, in->pending.s, in->pending.l);
			ks_putc(&in->carry_text, '\n');
		}
		if (pend) {
			ks_reserve(&in->carry_text, pend);
			in->carry_text.l += fread(in->carry_text.s + in->carry_text.l, 1, pend, in->fp);
		}
		for (i = 0; i < 2; i++) {
			if (!(in->tr_buf[i] = (char *)malloc(TR_HEAD + SAM_CHUNK + 2))) mDie("Out of memory");
			msh_huge_hint(in->tr_buf[i], TR_HEAD + SAM_CHUNK + 2);
		}
		pthread_mutex_init(&in->tr_mu, NULL);
		pthread_cond_init(&in->tr_cv_full, NULL);
		pthread_cond_init(&in->tr_cv_free, NULL);
#ifdef F_SETPIPE_SZ
		(void)fcntl(fileno(in->fp), F_SETPIPE_SZ, 1 << 20);      /* a pipe from the aligner: fewer, larger transfers (the most an unprivileged process may ask for) */
#endif
		if (pthread_create(&in->tr_thr, NULL, text_reader_main, in) != 0) mDie("Cannot start the reader thread");
		in->tr_started = 1;
	}
	for (;;) {
		char *data;
		size_t n;
		/* the next chunk (or the end of the input) */
		pthread_mutex_lock(&in->tr_mu);
		slot = in->tr_head;
		while (!in->tr_full[slot] && !in->tr_eof) pthread_cond_wait(&in->tr_cv_full, &in->tr_mu);
		if (!in->tr_full[slot]) {            /* (the reader's last chunk is behind us) */
			pthread_mutex_unlock(&in->tr_mu);
			in->text_eof = 1;
			gz_text_check(in);
			return 0;
		}
		last = in->tr_eof && !in->tr_full[slot ^ 1];
		pthread_mutex_unlock(&in->tr_mu);
		data = in->tr_buf[slot] + TR_HEAD;
		n = in->tr_len[slot];
		in->idle_hit = in->tr_idle[slot];
		if (in->carry_text.l <= TR_HEAD) {
			text = data - in->carry_text.l;
			if (in->carry_text.l) memcpy(text, in->carry_text.s, in->carry_text.l);
			tlen = in->carry_text.l + n;
			in->carry_text.l = 0;
			in->text_in_carry = 0;
		} else {                             /* a line of more than a megabyte: joined in the carry itself */
			ks_put(&in->carry_text, data, n);
			ks_reserve(&in->carry_text, 2);
			text = in->carry_text.s;
			tlen = in->carry_text.l;
			in->text_in_carry = 1;
		}
		/* the chunk ends behind its last newline; at the end of the input the rest is a line as well */
		end = tlen;
		if (!last) while (end > 0 && text[end - 1] != '\n') end--;
		if (end == 0 && !last) {             /* no whole line yet: keep it, take the next chunk */
			if (!in->text_in_carry) { in->carry_text.l = 0; ks_put(&in->carry_text, text, tlen); }
			pthread_mutex_lock(&in->tr_mu);
			in->tr_full[slot] = 0;
			in->tr_head = slot ^ 1;
			pthread_cond_signal(&in->tr_cv_free);
			pthread_mutex_unlock(&in->tr_mu);
			continue;
		}
		break;
	}
	if (nth > MSH_MAX_THREADS) nth = MSH_MAX_THREADS;
	if ((size_t)nth > end / 65536 + 1) nth = (int)(end / 65536 + 1);
	J.h = &in->hdr;
	J.text = text;
	J.lo[0] = 0;
	for (t = 1; t < nth; t++) {
		size_t q = end * (size_t)t / (size_t)nth;
		if (q < J.lo[t - 1]) q = J.lo[t - 1];
		while (q < end && q > 0 && text[q - 1] != '\n') q++;
		J.lo[t] = q;
	}
	J.lo[nth] = end;
	if (end == tlen) text[end] = 0;           /* (room for the terminator of an unterminated last line) */
	if (last && tlen > 0 && text[tlen - 1] != '\n') gz_text_check(in);      /* (cut by a decompressor that gave up? its verdict comes first) */
	msh_parallel(nth, sam_worker, &J);
	for (t = 0; t < nth; t++) { off[t] = total; total += J.out[t].l; }
	if (*len + total + 64 > *cap) {
		size_t nc = *cap ? *cap : ((size_t)16 << 20);
		while (nc < *len + total + 64) nc += nc >> 1;
		*buf = (uint8_t *)realloc(*buf, nc);
		msh_huge_hint(*buf, nc);
		if (!*buf) mDie("Out of memory");
		*cap = nc;
	}
	{
		samcopy_job C;
		C.dst = *buf + *len; C.out = J.out; C.off = off;
		msh_parallel(nth, samcopy_worker, &C);
		*len += total;
	}
	/* the cut line waits for the next chunk; the chunk's buffer goes back to the reader */
	{
		const size_t rest = tlen - end;
		if (in->text_in_carry) { memmove(in->carry_text.s, text + end, rest); in->carry_text.l = rest; }
		else { in->carry_text.l = 0; if (rest) ks_put(&in->carry_text, text + end, rest); }
	}
	pthread_mutex_lock(&in->tr_mu);
	in->tr_full[slot] = 0;
	in->tr_head = slot ^ 1;
	pthread_cond_signal(&in->tr_cv_free);
	pthread_mutex_unlock(&in->tr_mu);
	if (last) { in->text_eof = 1; gz_text_check(in); }
	if (total == 0 && !in->text_eof) return msh_sam_append(in, buf, len, cap);   /* (a chunk of empty lines) */
	return total;
}

/* Can the next append make progress without waiting longer than `ms` for the producer?  (A mapped file: always.  BAM through
 * a pipe: a whole block is in the buffer being parsed, or the reader thread has the next buffer, or the input has ended -- else
 * wait that long for the reader, who hands over what it has once the producer goes quiet.  Text: a line is here, or the
 * descriptor is readable.)  The decode stage asks this before it adds to a batch that already holds records. */
int msh_input_ready(msh_in *in, int ms) {
	if (in->is_bam) {
		bgz_in *b = &in->bz;
		int ok;
		if (b->map || b->eof) return 1;
		if (b->span_end > b->span_beg) return 1;
		if (b->cur >= 0 && b->cend - b->cbeg >= 18) {
			const uint8_t *h = b->cbuf + b->cbeg;
			const size_t have = b->cend - b->cbeg;
			const uint32_t xlen = le16(h + 10);
			uint32_t p = 0, bsize = 0;
			if (have >= 12 + (size_t)xlen)
				while (p + 4 <= xlen) {
					const uint32_t sl = le16(h + 12 + p + 2);
					if (h[12 + p] == 'B' && h[12 + p + 1] == 'C' && sl == 2) bsize = le16(h + 12 + p + 4) + 1;
					p += 4 + sl;
				}
			if (bsize && have >= bsize) return 1;
		}
		if (!b->rd_started) return 1;
		pthread_mutex_lock(&b->rd_mu);
		ok = b->rd_full[b->rd_head] || b->rd_eof;
		if (!ok && ms > 0) {
			struct timespec ts;
			clock_gettime(CLOCK_REALTIME, &ts);
			ts.tv_nsec += (long)(ms % 1000) * 1000000L;
			ts.tv_sec += ms / 1000 + ts.tv_nsec / 1000000000L;
			ts.tv_nsec %= 1000000000L;
			__atomic_store_n(&b->rd_wait, 1, __ATOMIC_RELAXED);
			while (!b->rd_full[b->rd_head] && !b->rd_eof)
				if (pthread_cond_timedwait(&b->rd_cv_full, &b->rd_mu, &ts) != 0) break;
			__atomic_store_n(&b->rd_wait, 0, __ATOMIC_RELAXED);
			ok = b->rd_full[b->rd_head] || b->rd_eof;
		}
		pthread_mutex_unlock(&b->rd_mu);
		return ok;
	}
	if (in->text_eof || in->has_pending) return 1;
	if (in->idle_hit) { in->idle_hit = 0; return 0; }
	if (!in->tr_started) return 1;
	{
		int ok;
		pthread_mutex_lock(&in->tr_mu);
		ok = in->tr_full[in->tr_head] || in->tr_eof;
		if (!ok && ms > 0) {
			struct timespec ts;
			clock_gettime(CLOCK_REALTIME, &ts);
			ts.tv_nsec += (long)(ms % 1000) * 1000000L;
			ts.tv_sec += ms / 1000 + ts.tv_nsec / 1000000000L;
			ts.tv_nsec %= 1000000000L;
			while (!in->tr_full[in->tr_head] && !in->tr_eof)
				if (pthread_cond_timedwait(&in->tr_cv_full, &in->tr_mu, &ts) != 0) break;
			ok = in->tr_full[in->tr_head] || in->tr_eof;
		}
		pthread_mutex_unlock(&in->tr_mu);
		return ok;
	}
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

/* a BGZF member at p (n bytes on hand): its size, 0 if what begins there is not one (a gzip member of another kind, or
 * damage), -1 if more bytes are needed to tell */
static long bgzf_member_at(const uint8_t *p, size_t n) {
	uint32_t xlen, o;
	if (n < 12) return -1;
	if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 0;
	xlen = le16(p + 10);
	if (n < 12 + (size_t)xlen) return -1;
	for (o = 0; o + 4 <= xlen; ) {
		const uint8_t *f = p + 12 + o;
		const uint32_t slen = le16(f + 2);
		if (f[0] == 'B' && f[1] == 'C' && slen == 2 && o + 6 <= xlen) {
			const long bsize = (long)le16(f + 4) + 1;
			return bsize >= (long)(12 + xlen + 8) ? bsize : 0;
		}
		o += 4 + slen;
	}
	return 0;
}

/* bgzip'd text: the members on hand inflated side by side on the reader's pool (the BAM path's host decoder: msh_inflate.c,
 * then zlib), in order into the pipe */
#define GZT_BLOCKS 512
typedef struct { const uint8_t *src; const msx_bgzf_block *blk; int n; uint8_t *out; const char *err; } gzt_job;
static void gzt_worker(void *arg, int tid, int nth) {
	gzt_job *J = (gzt_job *)arg;
	int i;
	for (i = tid; i < J->n; i += nth) {
		const char *e = inflate_payload_try(J->src + J->blk[i].in_off, J->blk[i].in_len, J->out + J->blk[i].out_off, J->blk[i].out_len, J->blk[i].crc32);
		if (e) __atomic_store_n(&J->err, e, __ATOMIC_RELAXED);
	}
}

/* compressed SAM text: every gzip member of the stream (plain gzip has one, bgzip one per block), inflated into the pipe.
 * BGZF members say how long they are and what they inflate to: as many as are on hand are inflated at once; anything else
 * goes through one zlib stream, member after member. */
static void *gz_text_main(void *arg) {
	msh_in *in = (msh_in *)arg;
	const int fd = fileno(in->gz_src);
	const size_t ICAP = (size_t)GZT_BLOCKS * 65536 + PRE_MAX, OCAP = (size_t)GZT_BLOCKS * 65536;
	uint8_t *ibuf = (uint8_t *)malloc(ICAP), *obuf = (uint8_t *)malloc(OCAP);
	msh_huge_hint(ibuf, ICAP); msh_huge_hint(obuf, OCAP);
	msx_bgzf_block *blk = (msx_bgzf_block *)malloc(GZT_BLOCKS * sizeof(msx_bgzf_block));
	size_t have = 0, at = 0;                  /* ibuf[at, have): read, not yet inflated */
	z_stream zs;
	int zs_on = 0, at_member_start = 1, eof = 0;
	if (!ibuf || !obuf || !blk) mDie("Out of memory");
	memcpy(ibuf, in->pre, in->npre);
	have = in->npre;
	/* ---- BGZF members, as long as that is what comes ---- */
	for (;;) {
		int n = 0;
		size_t out_total = 0, p = at;
		long m = 0;
		while (n < GZT_BLOCKS && (m = bgzf_member_at(ibuf + p, have - p)) > 0 && p + (size_t)m <= have) {
			const uint8_t *c = ibuf + p;
			const uint32_t xlen = le16(c + 10), isize = (uint32_t)le32(c + m - 4);
			if (isize > 65536u) { m = 0; break; }      /* (no BGZF block inflates to more: some other gzip member) */
			blk[n].in_off = p + 12 + xlen;
			blk[n].in_len = (uint32_t)((size_t)m - 12 - xlen - 8);
			blk[n].out_off = out_total;
			blk[n].out_len = isize;
			blk[n].crc32 = (uint32_t)le32(c + m - 8);
			blk[n].reserved_ = 0;
			out_total += isize;
			p += (size_t)m;
			n++;
		}
		if (n > 0) {
			gzt_job J;
			J.src = ibuf; J.blk = blk; J.n = n; J.out = obuf; J.err = NULL;
			msh_parallel(msh_threads() < n ? msh_threads() : n, gzt_worker, &J);
			if (J.err) { snprintf(in->gz_errmsg, sizeof in->gz_errmsg, "Corrupt gzip stream in SAM input (%s)", J.err); goto fail; }
			gz_write_all(in->gz_wfd, obuf, out_total);
			at = p;
			continue;
		}
		if (m == 0 && have > p) break;             /* something else begins here: the zlib loop takes over */
		/* a member is cut by what has been read (or nothing is left): more bytes */
		if (eof) {
			if (have > at) { snprintf(in->gz_errmsg, sizeof in->gz_errmsg, "Truncated gzip stream in SAM input"); goto fail; }
			goto done;
		}
		memmove(ibuf, ibuf + at, have - at);
		have -= at; at = 0;
		{
			ssize_t k;
			do k = read(fd, ibuf + have, ICAP - have); while (k < 0 && errno == EINTR);
			if (k < 0) { snprintf(in->gz_errmsg, sizeof in->gz_errmsg, "Read failed"); goto fail; }
			if (k == 0) eof = 1;
			have += (size_t)k;
		}
	}
	/* ---- any gzip member: one stream ---- */
	memset(&zs, 0, sizeof zs);
	if (inflateInit2(&zs, 15 + 32) != Z_OK) mDie("inflateInit2 failed");
	zs_on = 1;
	zs.next_in = ibuf + at; zs.avail_in = (uInt)(have - at);
	for (;;) {
		if (zs.avail_in == 0 && !eof) {
			ssize_t k;
			do k = read(fd, ibuf, ICAP < ((size_t)1 << 20) ? ICAP : ((size_t)1 << 20)); while (k < 0 && errno == EINTR);
			if (k < 0) { snprintf(in->gz_errmsg, sizeof in->gz_errmsg, "Read failed"); goto fail; }
			if (k == 0) eof = 1;
			zs.next_in = ibuf; zs.avail_in = (uInt)k;
		}
		if (zs.avail_in == 0 && eof) {
			if (!at_member_start) { snprintf(in->gz_errmsg, sizeof in->gz_errmsg, "Truncated gzip stream in SAM input"); goto fail; }
			break;
		}
		zs.next_out = obuf; zs.avail_out = (uInt)((size_t)4 << 20);
		{
			const int rc = inflate(&zs, Z_NO_FLUSH);
			if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) {
				snprintf(in->gz_errmsg, sizeof in->gz_errmsg, "Corrupt gzip stream in SAM input (%s)", zs.msg ? zs.msg : "zlib error");
				goto fail;
			}
			at_member_start = 0;
			gz_write_all(in->gz_wfd, obuf, ((size_t)4 << 20) - zs.avail_out);
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
done:
	if (zs_on) inflateEnd(&zs);
	free(ibuf);
	free(obuf);
	free(blk);
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
			msh_hdr_add_target(&in->hdr, (const char *)p + 4, strnlen((const char *)p + 4, (size_t)l_name),
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
		msh_hdr_targets_from_text(&in->hdr);
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
				/* (a last line without its newline: the text's end -- or where a decompressor gave up, whose verdict comes first) */
				if (in->line[n - 1] != '\n') gz_text_check(in);
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
	if (in->tr_started && in->tr_eof) {                 /* (a reader still waiting for input is left to the process's end) */
		pthread_join(in->tr_thr, NULL);
		free(in->tr_buf[0]);
		free(in->tr_buf[1]);
	}
	free(in->carry_text.s);
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
	msh_hdr_forget_names(&in->hdr);
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

