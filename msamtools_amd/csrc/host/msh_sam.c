/*
 * msh_sam.c -- SAM text <-> BAM record (SAMv1 1.4-1.5, 4.2): what the reference's readers and writers get from htslib's
 * sam_parse1 / sam_format1 under sam_read1 / sam_write1 (msam_helper.c:246-272).  Split out of msh_io.c in round 6.
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
		else if (v >= -32768) { ks_putc(rec, 's'); msh_put_le16(rec, (uint32_t)(v & 0xffff)); }
		else { ks_putc(rec, 'i'); msh_put_le32(rec, (uint32_t)v); }
	} else {
		if (v <= 255) { ks_putc(rec, 'C'); ks_putc(rec, (int)v); }
		else if (v <= 65535) { ks_putc(rec, 'S'); msh_put_le16(rec, (uint32_t)v); }
		else { ks_putc(rec, 'I'); msh_put_le32(rec, (uint32_t)v); }
	}
}

static uint8_t nt16[256];
static pthread_once_t nt16_once = PTHREAD_ONCE_INIT;
static void nt16_fill(void) {
	int c;
	for (c = 0; c < 256; c++) {
		const char *a = strchr(SEQ_NT16, toupper(c));
		nt16[c] = (uint8_t)((a && c) ? a - SEQ_NT16 : 15);
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
			msh_put_le32(rec, (uint32_t)(len << 4 | (uint32_t)(op - CIGAR_OPS)));
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
			msh_put_le32(rec, l_seq << 4 | 4u);
			msh_put_le32(rec, (uint32_t)reflen << 4 | 3u);
			n_cigar = 2;
		}
	}
	{   /* SEQ, 4-bit packed, and QUAL: a table look-up per base and one pass per string (a strchr per base and a ks_putc per
	     * byte were most of the parser's time: 322 MB/s of text per core; the reference's documented workflow feeds SAM text) */
		const size_t nseq = ((size_t)l_seq + 1) / 2;
		uint8_t *o;
		const uint8_t *q = (const uint8_t *)f[9];
		uint32_t k;
		pthread_once(&nt16_once, nt16_fill);       /* (the parser runs on every thread of the pool) */
		ks_reserve(rec, nseq + l_seq + 8);
		o = (uint8_t *)rec->s + rec->l;
		for (k = 0; k + 1 < l_seq; k += 2) *o++ = (uint8_t)(nt16[q[k]] << 4 | nt16[q[k + 1]]);
		if (l_seq & 1) *o++ = (uint8_t)(nt16[q[l_seq - 1]] << 4);
		if (strcmp(f[10], "*") == 0) {
			memset(o, 0xff, l_seq);
		} else {
			const uint8_t *ql = (const uint8_t *)f[10];
			if (strlen(f[10]) != l_seq) mDie("SEQ and QUAL of different length");
			for (k = 0; k < l_seq; k++) o[k] = (uint8_t)(ql[k] - 33);
		}
		rec->l += nseq + l_seq;
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
			msh_put_le32(rec, u);
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
			msh_put_le32(rec, 0);
			while (*c == ',') {
				c++;
				if (sub == 'f') {
					float fl = strtof(c, &c);
					uint32_t u;
					memcpy(&u, &fl, 4);
					msh_put_le32(rec, u);
				} else {
					long long v = strtoll(c, &c, 10);
					size_t es = msh_aux_type_size(sub);
					if (es == 1) ks_putc(rec, (int)(v & 0xff));
					else if (es == 2) msh_put_le16(rec, (uint32_t)(v & 0xffff));
					else msh_put_le32(rec, (uint32_t)v);
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
		msh_put_le32(rec, n_long);
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
			size_t es = msh_aux_type_size(sub);
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

