/* msx_cpu_e2e.c -- the reference's execution model, end to end, on host cores: BAM file in, filtered BAM + profile out.
 *
 * Test / measurement infrastructure (bench.py's `cpu_baseline.e2e` leg); nothing in the product links or runs it.
 * The reference is one thread that reads a record (sam_read1: BGZF inflate + CRC, record walk; msam_helper.c:246-268),
 * looks up MD / NM / AS (bam_aux_get, msam_filter.c:146-162), filters / pools / selects (msam_filter.c:98-263), writes
 * (sam_write1: BGZF deflate, msam_helper.c:270-272) and, in the second process of the pipe, counts and shares
 * (msam_profile.c:65-425).  It cannot be built on the GPU box (htslib, argtable2), so this is a port:
 *   decode   every BGZF block inflated with zlib and CRC-checked, the record chain walked, one aux scan per record
 *            for MD / NM / AS, the SoA arrays of the oracle filled
 *   compute  orc_filter (-l 80 -p 95 -z 80 --besthit) and orc_profile (--multi proportional) of oracle/msx_oracle.c
 *   encode   the selected records gathered into BGZF blocks of 0xff00 bytes, deflated with zlib at the given level
 *            (0: stored, as -bu), CRC, written to the output descriptor
 * on the first `max_records` records of the file (whole pools: the prefix ends at a QNAME boundary).
 * threads = 1 mirrors the reference (CPU-1 of BASELINE.md section 2); threads = N inflates, scans, filters (N shards cut
 * at QNAME boundaries) and deflates on N threads, the record chain and the profile stay serial (CPU-N).
 * Prints one line of JSON.
 * usage: msx_cpu_e2e <in.bam> <max_records> <threads> <out.bam> <level> */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <zlib.h>

#include "msx_oracle.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static void die(const char *m) { fprintf(stderr, "msx_cpu_e2e: %s\n", m); exit(1); }
static uint32_t le32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint32_t le16(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8; }

/* ---- a parallel for over [0, n) in contiguous shares ---- */
typedef void (*pf_fn)(void *arg, int64_t lo, int64_t hi, int tid);
typedef struct { pf_fn fn; void *arg; int64_t lo, hi; int tid; } pf_job;
static void *pf_main(void *a) { pf_job *j = (pf_job *)a; j->fn(j->arg, j->lo, j->hi, j->tid); return NULL; }
static void parallel_for(int nth, int64_t n, pf_fn fn, void *arg) {
	pthread_t th[256];
	pf_job jb[256];
	int t;
	if (nth <= 1 || n < 2) { fn(arg, 0, n, 0); return; }
	for (t = 0; t < nth; t++) {
		jb[t].fn = fn; jb[t].arg = arg; jb[t].tid = t; jb[t].lo = n * t / nth; jb[t].hi = n * (t + 1) / nth;
		pthread_create(&th[t], NULL, pf_main, &jb[t]);
	}
	for (t = 0; t < nth; t++) pthread_join(th[t], NULL);
}

/* ---- decode ---- */
typedef struct { const uint8_t *in; size_t in_len, out_off; uint32_t isize, crc; } blk_t;
typedef struct { blk_t *b; uint8_t *out; } inf_job;
static void inflate_range(void *arg, int64_t lo, int64_t hi, int tid) {
	inf_job *J = (inf_job *)arg;
	int64_t k;
	z_stream zs;
	(void)tid;
	memset(&zs, 0, sizeof zs);
	if (inflateInit2(&zs, -15) != Z_OK) die("inflateInit2");
	for (k = lo; k < hi; k++) {
		blk_t *b = &J->b[k];
		inflateReset(&zs);
		zs.next_in = (Bytef *)b->in; zs.avail_in = (uInt)b->in_len;
		zs.next_out = J->out + b->out_off; zs.avail_out = b->isize;
		if (b->isize && (inflate(&zs, Z_FINISH) != Z_STREAM_END || zs.total_out != b->isize)) die("inflate failed");
		if ((uint32_t)crc32(0, J->out + b->out_off, b->isize) != b->crc) die("CRC mismatch");
	}
	inflateEnd(&zs);
}

typedef struct {
	const uint8_t *u;
	const uint64_t *rec;        /* [n + 1] record offsets in u (each at its block_size word) */
	uint16_t *flag; uint8_t *rflags; int32_t *tid, *pos, *nm, *as;
	uint32_t *ncig, *mdlen, *qlen;
	const uint8_t **cigp, **mdp;
} scan_job;
static size_t aux_len(const uint8_t *t, const uint8_t *end) {
	switch (*t) {
	case 'A': case 'c': case 'C': return 2;
	case 's': case 'S': return 3;
	case 'i': case 'I': case 'f': return 5;
	case 'Z': case 'H': { const uint8_t *z = memchr(t + 1, 0, (size_t)(end - t - 1)); if (!z) die("corrupt aux"); return (size_t)(z - t) + 1; }
	case 'B': {
		size_t es = (t[1] == 'c' || t[1] == 'C') ? 1 : (t[1] == 's' || t[1] == 'S') ? 2 : 4;
		return 6 + es * le32(t + 2);
	}
	default: die("corrupt aux type"); return 0;
	}
}
static int32_t aux_int(const uint8_t *t) {
	switch (*t) {
	case 'c': return (int8_t)t[1];
	case 'C': return t[1];
	case 's': return (int16_t)le16(t + 1);
	case 'S': return (int32_t)le16(t + 1);
	case 'i': case 'I': return (int32_t)le32(t + 1);
	default: return 0;
	}
}
static void scan_range(void *arg, int64_t lo, int64_t hi, int tid) {
	scan_job *J = (scan_job *)arg;
	int64_t i;
	(void)tid;
	for (i = lo; i < hi; i++) {
		const uint8_t *r = J->u + J->rec[i] + 4, *end = J->u + J->rec[i + 1];
		const uint32_t lq = r[8], nc = le16(r + 12), ls = le32(r + 16);
		const uint8_t *p = r + 32 + lq + 4 * (size_t)nc + (ls + 1) / 2 + ls;
		uint8_t rf = 0;
		J->tid[i] = (int32_t)le32(r); J->pos[i] = (int32_t)le32(r + 4); J->flag[i] = (uint16_t)le16(r + 14);
		J->qlen[i] = lq - 1; J->ncig[i] = nc; J->cigp[i] = r + 32 + lq;
		J->nm[i] = 0; J->as[i] = 0; J->mdlen[i] = 0; J->mdp[i] = NULL;
		while (p + 3 <= end) {           /* bam_aux_get x 3, as one pass: first occurrence of each tag */
			const size_t sz = aux_len(p + 2, end);
			if (p[0] == 'M' && p[1] == 'D' && !(rf & ORC_HAS_MD)) { rf |= ORC_HAS_MD; if (p[2] == 'Z') { J->mdp[i] = p + 3; J->mdlen[i] = (uint32_t)sz - 2; } }
			else if (p[0] == 'N' && p[1] == 'M' && !(rf & ORC_HAS_NM)) { rf |= ORC_HAS_NM; J->nm[i] = aux_int(p + 2); }
			else if (p[0] == 'A' && p[1] == 'S' && !(rf & ORC_HAS_AS)) { rf |= ORC_HAS_AS; J->as[i] = aux_int(p + 2); }
			p += 2 + sz;
		}
		J->rflags[i] = rf;
	}
}

/* ---- compute: filter on shards cut at QNAME boundaries ---- */
typedef struct {
	const orc_records *all;
	const orc_filter_params *fp;
	const int64_t *cut;         /* [nth + 1] record index where every shard begins */
	int32_t **emit; int64_t *n_emit;
} filt_job;
static void filter_shard(void *arg, int64_t lo, int64_t hi, int tid) {
	filt_job *J = (filt_job *)arg;
	int64_t s;
	(void)tid;
	for (s = lo; s < hi; s++) {
		const int64_t a = J->cut[s], b = J->cut[s + 1];
		orc_records r = *J->all;
		int64_t err = -1, k;
		r.n = b - a;
		r.qname_off = J->all->qname_off + a; r.flag = J->all->flag + a; r.rflags = J->all->rflags + a; r.tid = J->all->tid + a;
		r.pos = J->all->pos + a; r.cigar_off = J->all->cigar_off + a; r.md_off = J->all->md_off + a; r.nm = J->all->nm + a; r.as = J->all->as + a;
		J->emit[s] = (int32_t *)malloc((size_t)(r.n + 1) * 4);
		if (orc_filter(&r, J->fp, J->emit[s], &J->n_emit[s], NULL, &err) != ORC_OK) die("orc_filter failed");
		for (k = 0; k < J->n_emit[s]; k++) J->emit[s][k] += (int32_t)a;
	}
}

/* ---- encode ---- */
typedef struct { const uint8_t *payload; size_t n_payload; uint8_t *slots; uint32_t *slot_len; int level; } def_job;
#define PAYLOAD 0xff00u
#define SLOT (PAYLOAD + 1024u)
static void deflate_range(void *arg, int64_t lo, int64_t hi, int tid) {
	def_job *J = (def_job *)arg;
	static const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
	z_stream zs;
	int64_t k;
	(void)tid;
	memset(&zs, 0, sizeof zs);
	if (J->level > 0 && deflateInit2(&zs, J->level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) die("deflateInit2");
	for (k = lo; k < hi; k++) {
		const uint8_t *src = J->payload + (size_t)k * PAYLOAD;
		const uint32_t n = (uint32_t)(J->n_payload - (size_t)k * PAYLOAD < PAYLOAD ? J->n_payload - (size_t)k * PAYLOAD : PAYLOAD);
		uint8_t *o = J->slots + (size_t)k * SLOT;
		uint32_t clen, crc = (uint32_t)crc32(0, src, n), total;
		if (J->level > 0) {
			deflateReset(&zs);
			zs.next_in = (Bytef *)src; zs.avail_in = n; zs.next_out = o + 18; zs.avail_out = SLOT - 26;
			if (deflate(&zs, Z_FINISH) != Z_STREAM_END) die("deflate failed");
			clen = (uint32_t)zs.total_out;
		} else {
			o[18] = 1; o[19] = (uint8_t)n; o[20] = (uint8_t)(n >> 8); o[21] = (uint8_t)~n; o[22] = (uint8_t)(~n >> 8);
			memcpy(o + 23, src, n);
			clen = 5 + n;
		}
		total = 18 + clen + 8;
		memcpy(o, head, 16);
		o[16] = (uint8_t)((total - 1) & 0xff); o[17] = (uint8_t)((total - 1) >> 8);
		memcpy(o + 18 + clen, &crc, 4); memcpy(o + 18 + clen + 4, &n, 4);
		J->slot_len[k] = total;
	}
	if (J->level > 0) deflateEnd(&zs);
}

int main(int argc, char **argv) {
	if (argc < 6) { fprintf(stderr, "usage: msx_cpu_e2e <in.bam> <max_records> <threads> <out.bam> <level>\n"); return 2; }
	const int64_t max_rec = atoll(argv[2]);
	const int nth = atoi(argv[3]) < 1 ? 1 : atoi(argv[3]) > 256 ? 256 : atoi(argv[3]);
	const int level = atoi(argv[5]);
	int fd = open(argv[1], O_RDONLY);
	struct stat st;
	if (fd < 0 || fstat(fd, &st) != 0) die("cannot open input");
	const uint8_t *file = (const uint8_t *)mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
	if (file == (const uint8_t *)MAP_FAILED) die("mmap");
	int ofd = open(argv[4], O_WRONLY | O_CREAT | O_TRUNC, 0644);
	if (ofd < 0) die("cannot open output");
	double t0 = now(), t_dec, t_cmp, t_enc;

	/* ---------------- decode ---------------- */
	/* blocks: enough of them for the header and max_rec records (found out as we go: inflate in rounds) */
	size_t nb = 0, cap_b = 1 << 16, fpos = 0, total_out = 0;
	blk_t *B = (blk_t *)malloc(cap_b * sizeof(blk_t));
	uint8_t *u = NULL;
	size_t u_cap = 0, u_len = 0;
	int32_t n_ref = 0;
	size_t rec0 = 0;             /* first record's offset in u */
	uint64_t *rec = NULL;
	int64_t n = 0, rec_cap = 0;
	size_t walk = 0;             /* u is walked up to here */
	int have_header = 0, done = 0;
	while (!done) {
		/* the next round of blocks (4096 at a time), inflated by all threads */
		const size_t first = nb;
		while (nb - first < 4096 && fpos + 28 <= (size_t)st.st_size) {
			const uint8_t *h = file + fpos;
			const uint32_t bsize = le16(h + 16) + 1u, xlen = le16(h + 10);
			if (h[0] != 0x1f || h[1] != 0x8b || fpos + bsize > (size_t)st.st_size) die("not a BGZF block");
			if (nb == cap_b) { cap_b *= 2; B = (blk_t *)realloc(B, cap_b * sizeof(blk_t)); }
			B[nb].in = h + 12 + xlen; B[nb].in_len = bsize - 12 - xlen - 8;
			B[nb].crc = le32(h + bsize - 8); B[nb].isize = le32(h + bsize - 4); B[nb].out_off = total_out;
			total_out += B[nb].isize;
			fpos += bsize;
			nb++;
		}
		if (nb == first) break;
		if (total_out + 64 > u_cap) { u_cap = total_out * 2 + (64 << 20); u = (uint8_t *)realloc(u, u_cap); if (!u) die("out of memory"); }
		{ inf_job J = {B + first, u}; parallel_for(nth, (int64_t)(nb - first), inflate_range, &J); }
		u_len = total_out;
		if (!have_header) {
			size_t p;
			int32_t k;
			if (u_len < 12 || memcmp(u, "BAM\1", 4)) die("not a BAM file");
			p = 8 + le32(u + 4);
			if (p + 4 > u_len) continue;
			n_ref = (int32_t)le32(u + p); p += 4;
			for (k = 0; k < n_ref; k++) { if (p + 4 > u_len) break; p += 4 + le32(u + p) + 4; if (p > u_len) break; }
			if (k < n_ref || p > u_len) continue;        /* the header is longer than what has been inflated so far */
			have_header = 1; rec0 = p; walk = p;
		}
		/* the record chain is serial (block_size by block_size), as sam_read1's is */
		while (walk + 4 <= u_len) {
			const uint32_t bs = le32(u + walk);
			if (walk + 4 + bs > u_len) break;
			if (n + 2 > rec_cap) { rec_cap = rec_cap ? rec_cap * 2 : (1 << 20); rec = (uint64_t *)realloc(rec, (size_t)rec_cap * 8); }
			rec[n++] = walk;
			walk += 4 + bs;
			if (n >= max_rec + 4096) { done = 1; break; }      /* (a little more than asked for: the prefix is cut at a QNAME boundary below) */
		}
	}
	if (!have_header) die("truncated header");
	rec[n] = walk;
	/* the prefix ends where a QNAME ends */
	if (n > max_rec) {
		int64_t k = max_rec;
		while (k < n) {
			const uint8_t *a = u + rec[k - 1] + 4, *b = u + rec[k] + 4;
			if (a[8] != b[8] || memcmp(a + 32, b + 32, a[8])) break;
			k++;
		}
		n = k;
	}
	/* one aux scan per record, SoA fill */
	orc_records R;
	memset(&R, 0, sizeof R);
	scan_job S;
	memset(&S, 0, sizeof S);
	S.u = u; S.rec = rec;
	S.flag = (uint16_t *)malloc((size_t)n * 2 + 2); S.rflags = (uint8_t *)malloc((size_t)n + 1);
	S.tid = (int32_t *)malloc((size_t)n * 4 + 4); S.pos = (int32_t *)malloc((size_t)n * 4 + 4); S.nm = (int32_t *)malloc((size_t)n * 4 + 4); S.as = (int32_t *)malloc((size_t)n * 4 + 4);
	S.ncig = (uint32_t *)malloc((size_t)n * 4 + 4); S.mdlen = (uint32_t *)malloc((size_t)n * 4 + 4); S.qlen = (uint32_t *)malloc((size_t)n * 4 + 4);
	S.cigp = (const uint8_t **)malloc((size_t)n * 8 + 8); S.mdp = (const uint8_t **)malloc((size_t)n * 8 + 8);
	parallel_for(nth, n, scan_range, &S);
	uint32_t *cigar_off = (uint32_t *)malloc((size_t)(n + 1) * 4), *md_off = (uint32_t *)malloc((size_t)(n + 1) * 4), *qname_off = (uint32_t *)malloc((size_t)(n + 1) * 4);
	{
		int64_t i;
		uint32_t c = 0, m = 0, q = 0;
		for (i = 0; i < n; i++) { cigar_off[i] = c; md_off[i] = m; qname_off[i] = q; c += S.ncig[i]; m += S.mdlen[i]; q += S.qlen[i]; }
		cigar_off[n] = c; md_off[n] = m; qname_off[n] = q;
		uint32_t *cigar = (uint32_t *)malloc((size_t)c * 4 + 4);
		uint8_t *md = (uint8_t *)malloc((size_t)m + 1);
		char *qname = (char *)malloc((size_t)q + 1);
		for (i = 0; i < n; i++) {
			memcpy(cigar + cigar_off[i], S.cigp[i], 4 * (size_t)S.ncig[i]);
			if (S.mdlen[i]) memcpy(md + md_off[i], S.mdp[i], S.mdlen[i]);
			memcpy(qname + qname_off[i], u + rec[i] + 4 + 32, S.qlen[i]);
		}
		R.n = n; R.qname_off = qname_off; R.qname = qname; R.flag = S.flag; R.rflags = S.rflags; R.tid = S.tid; R.pos = S.pos;
		R.cigar_off = cigar_off; R.cigar = cigar; R.md_off = md_off; R.md = md; R.nm = S.nm; R.as = S.as;
	}
	t_dec = now() - t0;

	/* ---------------- compute ---------------- */
	t0 = now();
	orc_filter_params fp = {80, 950, 20, 0, 0, 0, 1, 0};
	int64_t cut[257];
	int32_t *emit_s[256];
	int64_t n_emit_s[256], n_emit = 0;
	int s;
	cut[0] = 0;
	for (s = 1; s < nth; s++) {
		int64_t k = n * s / nth;
		if (k <= cut[s - 1]) k = cut[s - 1];
		while (k > cut[s - 1] && k < n) {            /* forward to the next QNAME boundary */
			const uint8_t *a = u + rec[k - 1] + 4, *b = u + rec[k] + 4;
			if (a[8] != b[8] || memcmp(a + 32, b + 32, a[8])) break;
			k++;
		}
		cut[s] = k;
	}
	cut[nth] = n;
	{ filt_job J = {&R, &fp, cut, emit_s, n_emit_s}; parallel_for(nth, nth, filter_shard, &J); }
	for (s = 0; s < nth; s++) n_emit += n_emit_s[s];
	int32_t *emit = (int32_t *)malloc((size_t)(n_emit + 1) * 4);
	{ int64_t k = 0; for (s = 0; s < nth; s++) { memcpy(emit + k, emit_s[s], (size_t)n_emit_s[s] * 4); k += n_emit_s[s]; free(emit_s[s]); } }
	double *ab = (double *)calloc((size_t)n_ref + 1, sizeof(double));
	orc_profile_stats ps;
	memset(&ps, 0, sizeof ps);
	if (orc_profile(&R, emit, n_emit, NULL, n_ref, ORC_MULTI_SHARE_PROPORTIONAL, ab, NULL, &ps) != ORC_OK) die("orc_profile failed");
	t_cmp = now() - t0;

	/* ---------------- encode ---------------- */
	t0 = now();
	size_t out_bytes = 0, written = 0;
	{
		int64_t k;
		for (k = 0; k < n_emit; k++) out_bytes += (size_t)(rec[emit[k] + 1] - rec[emit[k]]);
		uint8_t *payload = (uint8_t *)malloc(rec0 + out_bytes + 64);
		size_t p = rec0;
		memcpy(payload, u, rec0);                      /* the header goes out as it came in */
		for (k = 0; k < n_emit; k++) { const size_t l = (size_t)(rec[emit[k] + 1] - rec[emit[k]]); memcpy(payload + p, u + rec[emit[k]], l); p += l; }
		const size_t nblk = (p + PAYLOAD - 1) / PAYLOAD;
		uint8_t *slots = (uint8_t *)malloc(nblk * (size_t)SLOT);
		uint32_t *slot_len = (uint32_t *)malloc(nblk * 4 + 4);
		def_job J = {payload, p, slots, slot_len, level};
		size_t b;
		parallel_for(nth, (int64_t)nblk, deflate_range, &J);
		for (b = 0; b < nblk; b++) {
			size_t off = 0;
			while (off < slot_len[b]) { ssize_t w = write(ofd, slots + b * (size_t)SLOT + off, slot_len[b] - off); if (w <= 0) die("write failed"); off += (size_t)w; }
			written += slot_len[b];
		}
		static const uint8_t eof_block[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
		if (write(ofd, eof_block, 28) != 28) die("write failed");
		written += 28;
	}
	close(ofd);
	t_enc = now() - t0;
	{
		const double wall = t_dec + t_cmp + t_enc;
		printf("{\"records\": %lld, \"records_out\": %lld, \"threads\": %d, \"bgzf_level_out\": %d, \"decode_s\": %.4f, \"compute_s\": %.4f, "
		       "\"encode_s\": %.4f, \"seconds\": %.4f, \"M_alignments_per_s\": %.3f, \"inserts\": %u, \"iterations\": %d, \"bytes_out\": %zu, "
		       "\"inflated_MB\": %.1f}\n",
		       (long long)n, (long long)n_emit, nth, level, t_dec, t_cmp, t_enc, wall, (double)n / wall / 1e6, ps.insert_count, ps.iterations, written,
		       (double)walk / 1e6);
	}
	return 0;
}
