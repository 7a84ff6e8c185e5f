/*
 * msh_io.c -- BGZF/BAM and SAM-text I/O for the host command line.
 * Written from the SAM/BAM specification (SAMv1 sections 1.3-1.5, 4.1-4.2);
 * replaces what the reference gets from htslib through msam_helper.c:196-293.
 */
#include "msh.h"

#include <ctype.h>
#include <pthread.h>
#include <stdarg.h>
#include <unistd.h>
#include <zlib.h>

/* ------------------------------------------------------------------------ */
/* errors, strings                                                            */
/* ------------------------------------------------------------------------ */
void mDie(const char *fmt, ...) {
	va_list ap;
	fflush(stdout);
	fprintf(stderr, "Fatal Error: ");
	va_start(ap, fmt);
	vfprintf(stderr, fmt, ap);
	va_end(ap);
	fprintf(stderr, "\n");
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

int32_t msh_hdr_name2tid(const msh_hdr *h, const char *name) {
	/* linear probe with a one-entry cache is enough for the SAM-text path (fixtures, pipes) */
	static int32_t last = 0;
	int32_t i;
	if (last < h->n_targets && strcmp(h->target_name[last], name) == 0) return last;
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

size_t msh_aux_size(const uint8_t *t, const uint8_t *end) {
	int ty = *t;
	size_t fs = aux_type_size(ty);
	if (fs) return 1 + fs;
	if (ty == 'Z' || ty == 'H') {
		const uint8_t *z = memchr(t + 1, 0, (size_t)(end - t - 1));
		return z ? (size_t)(z - t) + 1 : (size_t)(end - t);
	}
	if (ty == 'B') {
		size_t es = aux_type_size(t[1]);
		uint32_t cnt = (uint32_t)le32(t + 2);
		return 1 + 1 + 4 + es * cnt;
	}
	mDie("Corrupt aux field of type '%c' in BAM record", ty);
	return 0;
}

const uint8_t *msh_aux_get(const uint8_t *rec, size_t len, const char tag[2]) {
	const uint8_t *p = REC_AUX(rec), *end = rec + len;
	while (p + 3 <= end) {
		if (p[0] == (uint8_t)tag[0] && p[1] == (uint8_t)tag[1]) return p + 2;
		p += 2 + msh_aux_size(p + 2, end);
	}
	return NULL;
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
/* BGZF reader with batched, multi-threaded inflate                           */
/* ------------------------------------------------------------------------ */
#define BGZF_MAX 65536
#define BGZF_BATCH 128

typedef struct {
	FILE *fp;
	uint8_t *cbuf;                       /* BGZF_BATCH compressed blocks back to back   */
	size_t coff[BGZF_BATCH + 1];
	uint8_t *ubuf;                       /* BGZF_BATCH * 64 KiB inflated                 */
	uint32_t ulen[BGZF_BATCH];
	int nblk, cur;                       /* blocks in the batch, block being consumed    */
	uint32_t upos;                       /* offset in the current block                  */
	int eof, nthreads;
} bgz_in;

typedef struct {
	bgz_in *b;
	int first, step;
} bgz_job;

static void inflate_block(bgz_in *b, int i) {
	const uint8_t *c = b->cbuf + b->coff[i];
	size_t clen = b->coff[i + 1] - b->coff[i];
	uint32_t xlen = le16(c + 10);
	const uint8_t *data = c + 12 + xlen;
	size_t dlen = clen - 12 - xlen - 8;
	uint32_t isize = (uint32_t)le32(c + clen - 4);
	z_stream zs;
	memset(&zs, 0, sizeof zs);
	if (isize > BGZF_MAX) mDie("Corrupt BGZF block (ISIZE %u)", isize);
	zs.next_in = (Bytef *)data;
	zs.avail_in = (uInt)dlen;
	zs.next_out = b->ubuf + (size_t)i * BGZF_MAX;
	zs.avail_out = BGZF_MAX;
	if (inflateInit2(&zs, -15) != Z_OK) mDie("zlib inflateInit2 failed");
	if (isize && inflate(&zs, Z_FINISH) != Z_STREAM_END) mDie("Corrupt BGZF block (inflate failed)");
	inflateEnd(&zs);
	if (isize && zs.total_out != isize) mDie("Corrupt BGZF block (size mismatch)");
	if ((uint32_t)crc32(crc32(0L, NULL, 0), b->ubuf + (size_t)i * BGZF_MAX, isize) != (uint32_t)le32(c + clen - 8))
		mDie("Corrupt BGZF block (CRC mismatch)");
	b->ulen[i] = isize;
}

static void *inflate_worker(void *arg) {
	bgz_job *j = (bgz_job *)arg;
	int i;
	for (i = j->first; i < j->b->nblk; i += j->step) inflate_block(j->b, i);
	return NULL;
}

static int host_threads(void) {
	const char *e = getenv("MSX_THREADS");
	long n = e ? strtol(e, NULL, 10) : sysconf(_SC_NPROCESSORS_ONLN);
	if (n < 1) n = 1;
	if (n > 32) n = 32;
	return (int)n;
}

/* read the next batch of raw blocks and inflate them; returns 0 at EOF */
static int bgz_fill(bgz_in *b) {
	size_t off = 0;
	int t;
	b->nblk = 0;
	b->cur = 0;
	b->upos = 0;
	if (b->eof) return 0;
	while (b->nblk < BGZF_BATCH) {
		uint8_t *h = b->cbuf + off;
		size_t got = fread(h, 1, 18, b->fp);
		uint32_t bsize;
		if (got == 0) { b->eof = 1; break; }
		if (got != 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4))
			mDie("Input is not BGZF-compressed BAM (bad block header)");
		{   /* locate the BC subfield (normally the only one) */
			uint32_t xlen = le16(h + 10);
			if (xlen == 6 && h[12] == 'B' && h[13] == 'C') {
				bsize = le16(h + 16) + 1;
			} else {
				uint8_t extra[65536];
				uint32_t p = 0;
				int found = 0;
				memcpy(extra, h + 12, 6);
				if (xlen > 6 && fread(extra + 6, 1, xlen - 6, b->fp) != xlen - 6) mDie("Truncated BGZF block");
				memcpy(h + 12, extra, xlen);
				bsize = 0;
				while (p + 4 <= xlen) {
					uint32_t sl = le16(extra + p + 2);
					if (extra[p] == 'B' && extra[p + 1] == 'C' && sl == 2) { bsize = le16(extra + p + 4) + 1; found = 1; }
					p += 4 + sl;
				}
				if (!found) mDie("BGZF block without BC subfield");
				got = 12 + xlen;
			}
		}
		if (bsize < got || bsize > BGZF_MAX + 1024) mDie("Corrupt BGZF block size");
		if (fread(h + got, 1, bsize - got, b->fp) != bsize - got) mDie("Truncated BGZF block");
		b->coff[b->nblk] = off;
		off += bsize;
		b->nblk++;
		b->coff[b->nblk] = off;
	}
	if (b->nblk == 0) return 0;
	t = b->nthreads < b->nblk ? b->nthreads : b->nblk;
	if (t <= 1) {
		int i;
		for (i = 0; i < b->nblk; i++) inflate_block(b, i);
	} else {
		pthread_t th[32];
		bgz_job job[32];
		int i;
		for (i = 0; i < t; i++) {
			job[i].b = b; job[i].first = i; job[i].step = t;
			if (pthread_create(&th[i], NULL, inflate_worker, &job[i]) != 0) mDie("pthread_create failed");
		}
		for (i = 0; i < t; i++) pthread_join(th[i], NULL);
	}
	return 1;
}

/* returns bytes copied (< n only at EOF) */
static size_t bgz_read(bgz_in *b, void *dst, size_t n) {
	size_t done = 0;
	while (done < n) {
		uint32_t avail;
		if (b->cur >= b->nblk) {
			if (!bgz_fill(b)) break;
		}
		avail = b->ulen[b->cur] - b->upos;
		if (avail == 0) { b->cur++; b->upos = 0; continue; }
		if (avail > n - done) avail = (uint32_t)(n - done);
		memcpy((uint8_t *)dst + done, b->ubuf + (size_t)b->cur * BGZF_MAX + b->upos, avail);
		b->upos += avail;
		done += avail;
	}
	return done;
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
	uint32_t flag, mapq, n_cigar = 0, l_seq;
	int64_t reflen = 0;
	size_t qn_len, core_at;
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
	int32_t tid = REC_TID(r), mtid = le32(r + 20);
	uint32_t n_cigar = REC_NCIGAR(r), l_seq = (uint32_t)REC_LSEQ(r), k;
	const uint8_t *cig = REC_CIGAR(r), *seq = cig + 4 * n_cigar, *qual = seq + (l_seq + 1) / 2;
	const uint8_t *p = qual + l_seq, *end = r + len;
	ks_puts(o, REC_QNAME(r));
	ks_printf(o, "\t%u\t", REC_FLAG(r));
	ks_puts(o, tid >= 0 && tid < h->n_targets ? h->target_name[tid] : "*");
	ks_printf(o, "\t%d\t%u\t", REC_POS(r) + 1, REC_MAPQ(r));
	if (n_cigar == 0) ks_putc(o, '*');
	for (k = 0; k < n_cigar; k++) {
		uint32_t c = (uint32_t)le32(cig + 4 * k);
		ks_printf(o, "%u%c", c >> 4, (c & 15) < 10 ? CIGAR_OPS[c & 15] : '?');
	}
	ks_putc(o, '\t');
	if (mtid < 0) ks_putc(o, '*');
	else if (mtid == tid) ks_putc(o, '=');
	else ks_puts(o, mtid < h->n_targets ? h->target_name[mtid] : "*");
	ks_printf(o, "\t%d\t%d\t", le32(r + 24) + 1, le32(r + 28));
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
		p += 2 + msh_aux_size(p + 2, end);
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
};

msh_in *msh_open(const char *path) {
	msh_in *in = (msh_in *)calloc(1, sizeof(*in));
	int c0, c1;
	if (!in) mDie("Out of memory");
	in->fp = strcmp(path, "-") == 0 ? stdin : fopen(path, "rb");
	if (!in->fp) mDie("Cannot open %s for reading", path);
	setvbuf(in->fp, NULL, _IOFBF, 1 << 20);
	c0 = fgetc(in->fp);
	c1 = c0 == EOF ? EOF : fgetc(in->fp);
	if (c1 != EOF) ungetc(c1, in->fp);
	if (c0 != EOF) ungetc(c0, in->fp);   /* two-byte pushback works on glibc full-buffered streams */
	in->is_bam = (c0 == 0x1f && c1 == 0x8b);
	if (in->is_bam) {
		uint8_t magic[8];
		int32_t l_text, n_ref, i;
		in->bz.fp = in->fp;
		in->bz.cbuf = (uint8_t *)malloc((size_t)BGZF_BATCH * (BGZF_MAX + 1024));
		in->bz.ubuf = (uint8_t *)malloc((size_t)BGZF_BATCH * BGZF_MAX);
		in->bz.nthreads = host_threads();
		if (!in->bz.cbuf || !in->bz.ubuf) mDie("Out of memory");
		if (bgz_read(&in->bz, magic, 8) != 8 || memcmp(magic, "BAM\1", 4) != 0)
			mDie("Cannot read header from %s", path);
		l_text = le32(magic + 4);
		ks_reserve(&in->hdr.text, (size_t)l_text + 1);
		if (bgz_read(&in->bz, in->hdr.text.s, (size_t)l_text) != (size_t)l_text) mDie("Cannot read header from %s", path);
		in->hdr.text.l = strnlen(in->hdr.text.s, (size_t)l_text);
		in->hdr.text.s[in->hdr.text.l] = 0;
		if (bgz_read(&in->bz, magic, 4) != 4) mDie("Cannot read header from %s", path);
		n_ref = le32(magic);
		for (i = 0; i < n_ref; i++) {
			uint8_t b4[4];
			char name[65536];
			int32_t l_name;
			if (bgz_read(&in->bz, b4, 4) != 4) mDie("Cannot read header from %s", path);
			l_name = le32(b4);
			if (l_name <= 0 || l_name > 65535 || bgz_read(&in->bz, name, (size_t)l_name) != (size_t)l_name)
				mDie("Cannot read header from %s", path);
			if (bgz_read(&in->bz, b4, 4) != 4) mDie("Cannot read header from %s", path);
			hdr_add_target(&in->hdr, name, strnlen(name, (size_t)l_name), (uint32_t)le32(b4));
		}
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
		hdr_targets_from_text(&in->hdr);
	}
	return in;
}

const msh_hdr *msh_header(msh_in *in) { return &in->hdr; }

int msh_read(msh_in *in, kstr *rec) {
	if (in->is_bam) {
		uint8_t b4[4];
		size_t got = bgz_read(&in->bz, b4, 4);
		int32_t bs;
		if (got == 0) return -1;
		if (got != 4) mDie("Truncated BAM record");
		bs = le32(b4);
		if (bs < 32) mDie("Corrupt BAM record (block_size %d)", bs);
		rec->l = 0;
		ks_reserve(rec, (size_t)bs);
		if (bgz_read(&in->bz, rec->s, (size_t)bs) != (size_t)bs) mDie("Truncated BAM record");
		rec->l = (size_t)bs;
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
				if (n <= 0) return -1;
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
	if (in->fp && in->fp != stdin) fclose(in->fp);
	for (i = 0; i < in->hdr.n_targets; i++) free(in->hdr.target_name[i]);
	free(in->hdr.target_name);
	free(in->hdr.target_len);
	free(in->hdr.text.s);
	free(in->bz.cbuf);
	free(in->bz.ubuf);
	free(in->line);
	free(in->pending.s);
	free(in);
}

/* ------------------------------------------------------------------------ */
/* output                                                                     */
/* ------------------------------------------------------------------------ */
struct msh_out {
	FILE *fp;
	int mode;
	const msh_hdr *hdr;
	kstr line;
	uint8_t *ubuf;       /* BGZF payload being filled */
	uint32_t ulen;
	int level;
};
#define BGZF_PAYLOAD 0xff00

static void bgz_flush_block(msh_out *o) {
	uint8_t out[BGZF_MAX + 1024];
	z_stream zs;
	uint32_t clen, crc, total;
	static const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
	memset(&zs, 0, sizeof zs);
	if (deflateInit2(&zs, o->level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) mDie("zlib deflateInit2 failed");
	zs.next_in = o->ubuf;
	zs.avail_in = o->ulen;
	zs.next_out = out + 18;
	zs.avail_out = sizeof out - 18 - 8;
	if (deflate(&zs, Z_FINISH) != Z_STREAM_END) mDie("BGZF deflate failed");
	clen = (uint32_t)zs.total_out;
	deflateEnd(&zs);
	memcpy(out, head, 16);
	total = 18 + clen + 8;
	out[16] = (uint8_t)((total - 1) & 0xff);
	out[17] = (uint8_t)((total - 1) >> 8);
	crc = (uint32_t)crc32(crc32(0L, NULL, 0), o->ubuf, o->ulen);
	out[18 + clen + 0] = (uint8_t)crc; out[18 + clen + 1] = (uint8_t)(crc >> 8);
	out[18 + clen + 2] = (uint8_t)(crc >> 16); out[18 + clen + 3] = (uint8_t)(crc >> 24);
	out[18 + clen + 4] = (uint8_t)o->ulen; out[18 + clen + 5] = (uint8_t)(o->ulen >> 8);
	out[18 + clen + 6] = (uint8_t)(o->ulen >> 16); out[18 + clen + 7] = (uint8_t)(o->ulen >> 24);
	if (fwrite(out, 1, total, o->fp) != total) mDie("Write failed");
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
	setvbuf(fp, NULL, _IOFBF, 1 << 20);
	if (mode == MSH_OUT_BAM || mode == MSH_OUT_UBAM) {
		kstr b = {0, 0, 0};
		int32_t i;
		size_t tl = strlen(hdr_text);
		o->ubuf = (uint8_t *)malloc(BGZF_MAX);
		o->level = mode == MSH_OUT_UBAM ? 0 : Z_DEFAULT_COMPRESSION;
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
		fputs(hdr_text, fp);
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
		if (fwrite(o->line.s, 1, o->line.l, o->fp) != o->line.l) mDie("Write failed");
	}
}

void msh_out_close(msh_out *o) {
	if (!o) return;
	if (o->mode == MSH_OUT_BAM || o->mode == MSH_OUT_UBAM) {
		static const uint8_t eof_block[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0,
		                                      0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
		if (o->ulen) bgz_flush_block(o);
		fwrite(eof_block, 1, 28, o->fp);
	}
	fflush(o->fp);
	free(o->ubuf);
	free(o->line.s);
	free(o);
}
