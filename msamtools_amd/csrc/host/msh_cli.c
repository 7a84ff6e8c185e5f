/*
 * msh_cli.c -- `msamtools filter` and `msamtools profile` on the MI355X path.
 *
 * Keeps the reference's command-line surface (msam_filter.c:304-347,
 * msam_profile.c:554-570), its validation messages and their stdout/stderr
 * split, the QNAME-grouping preflight (msam_helper.c:295-484), the @PG /
 * '#' provenance lines (msam_helper.c:139-184) and the profile
 * post-processing + text format (msam_profile.c:858-983, mMatrix.c:359-376).
 * Records are decoded on the host, packed into structure-of-arrays batches cut
 * at QNAME-pool boundaries and handed to libmsamtools_amd.so; the library
 * returns the indices of the records to write, in the reference's order.
 */
#include "msh.h"

#include <errno.h>
#include <fcntl.h>
#include <getopt.h>
#include <pthread.h>
#include <sys/mman.h>
#include <math.h>
#include <time.h>
#include <unistd.h>
#include <zlib.h>

#define QNAME_GROUP_CHECK_RECORDS 10000      /* msam_helper.c:4-6 */
#define COORD_ORDER_CHECK_RECORDS 100000
#define COORD_ORDER_MIN_RECORDS 10000

/* the context the calling thread works on: one device thread per GPU, each with its own (MSX_DEVICES) */
static __thread msx_ctx *g_ctx;

/* stage timers, printed to stderr when MSX_TIMING is set */
static double now_s(void) {
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static double t_decode, t_upload, t_gpu, t_fetch, t_write;
#define TIC double t0_ = now_s()
#define TOC(acc) do { double t1_ = now_s(); (acc) += t1_ - t0_; t0_ = t1_; } while (0)

/* Everything has been written: leave without tearing down the HIP runtime, the page-locked arenas and
 * a gigabyte of batch buffers (a quarter of a second on a one-second run).  MSX_CLEAN_EXIT=1 keeps the
 * orderly shutdown (leak checks). */
static double g_t_main;     /* now_s() at the head of main (MSX_TIMING) */

static void fast_exit(void) {
	if (getenv("MSX_TIMING")) {
		/* what the stage timers do not see: from exec to main (loader, static initialisers) and, after this line,
		 * the kernel taking the address space apart (mapped input, pinned buffers) */
		struct timespec ts;
		double cpu = 0;
		if (clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts) == 0) cpu = (double)ts.tv_sec + ts.tv_nsec * 1e-9;
		fprintf(stderr, "# process: %.3f s from main to exit, %.3f s of CPU time in all threads\n", now_s() - g_t_main, cpu);
	}
	if (getenv("MSX_CLEAN_EXIT")) return;
	fflush(stdout);
	fflush(stderr);
	_exit(0);
}

#define MSX(call) do { if ((call) != MSX_OK) mDie("%s", msx_last_error(g_ctx)); } while (0)

/* several GPUs: one process per GPU, started with RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT in
 * the environment (torchrun, mpirun wrappers, a shell loop).  `profile` then reads ITS shard of the sample --
 * "{rank}" in the input path is replaced by the rank; shards are cut at QNAME boundaries, e.g. by splitting the
 * name-grouped BAM -- and the ranks exchange counts and, per sharing iteration, the increment vector over RCCL
 * (msx_profile_finalize_dist_enqueue); rank 0 writes the profile. */
static int dist_world(void) { const char *e = getenv("WORLD_SIZE"); int v = e ? atoi(e) : 1; return v > 1 ? v : 1; }
static int dist_rank(void) { const char *e = getenv("RANK"); return e ? atoi(e) : 0; }
/* Running as one rank of several is something the caller asks for -- "{rank}" in the input path, or MSX_DIST=1 --
 * never something inferred from a WORLD_SIZE that happens to be in the environment (a SLURM job, a torchrun
 * parent): a rank that read the WHOLE file would have its counts multiplied by the number of ranks. */
static int g_dist;

static void ctx_open_dev(int id) {
	if (msx_ctx_create(&g_ctx, id) != MSX_OK) mDie("%s", msx_last_error(NULL));
}
static void ctx_open(void) {
	const char *dev = getenv("MSX_DEVICE"), *lr = getenv("LOCAL_RANK");
	ctx_open_dev(dev ? atoi(dev) : (g_dist && lr ? atoi(lr) : 0));
}

/* One process, several GPUs: MSX_DEVICES="0,1,2,3" (device ids, one context and one device thread each; the decode
 * stage deals the batches of the one input to them and the writer puts filter's output back in input order --
 * SURVEY.md 8e).  Unset: one device (MSX_DEVICE, default 0). */
#define MSH_MAX_DEVICES 16
static int device_list(int *ids) {
	const char *e = getenv("MSX_DEVICES");
	int n = 0;
	if (e && *e && !g_dist) {
		const char *p = e;
		while (*p && n < MSH_MAX_DEVICES) {
			char *end;
			long v = strtol(p, &end, 10);
			if (end == p || v < 0) mDie("MSX_DEVICES: expected a comma-separated list of device ids, got '%s'", e);
			ids[n++] = (int)v;
			p = *end == ',' ? end + 1 : end;
			if (*end && *end != ',') mDie("MSX_DEVICES: expected a comma-separated list of device ids, got '%s'", e);
		}
	}
	if (n == 0) {
		const char *dev = getenv("MSX_DEVICE"), *lr = getenv("LOCAL_RANK");
		ids[n++] = dev ? atoi(dev) : (g_dist && lr ? atoi(lr) : 0);
	}
	return n;
}

static size_t batch_target(void) {
	const char *e = getenv("MSX_BATCH_RECORDS");
	long n = e ? strtol(e, NULL, 10) : (1L << 21);
	return n < 1 ? 1 : (size_t)n;
}

/* stringify_argv() as used by mBuildCommandLine (msam_helper.c:59-76) */
static char *command_line(int argc, char *argv[]) {
	kstr k = {0, 0, 0};
	int i;
	char *p;
	ks_puts(&k, PROGRAM);
	for (i = 0; i < argc; i++) {
		ks_putc(&k, ' ');
		ks_puts(&k, argv[i]);
	}
	for (p = k.s; *p; p++)
		if (*p == '\t') *p = ' ';
	return k.s;
}

/* ------------------------------------------------------------------------ */
/* record batch: BAM blobs + the SoA view the kernels read                    */
/* ------------------------------------------------------------------------ */
typedef struct {
	size_t n, cap;
	kstr blob;                 /* SAM-text path: [block_size | record] back to back          */
	const uint8_t *base;       /* records live at base + rec_off[i] + 4 (blob or the BAM span) */
	size_t *rec_off;           /* [n+1] offsets of the block_size fields                     */
	uint32_t *md_rel;          /* bulk path scratch: offset of the MD string inside the record */
	uint8_t *bound;            /* bulk path scratch: record starts a pool                    */
	uint16_t *flag;
	uint8_t *rflags;
	int32_t *tid, *pos, *nm, *as;
	uint32_t *cigar_off, *md_off;
	uint32_t *cigar; size_t cigar_cap;
	uint8_t *md; size_t md_cap;
	uint32_t *group_off; size_t n_groups, group_cap;
} rbatch;

static void rb_reserve(rbatch *b) {
	if (b->n + 2 > b->cap) {
		size_t c = b->cap ? b->cap * 2 : 65536;
		b->rec_off = (size_t *)realloc(b->rec_off, (c + 1) * sizeof(size_t));
		b->flag = (uint16_t *)realloc(b->flag, c * 2);
		b->rflags = (uint8_t *)realloc(b->rflags, c);
		b->tid = (int32_t *)realloc(b->tid, c * 4);
		b->pos = (int32_t *)realloc(b->pos, c * 4);
		b->nm = (int32_t *)realloc(b->nm, c * 4);
		b->as = (int32_t *)realloc(b->as, c * 4);
		b->cigar_off = (uint32_t *)realloc(b->cigar_off, (c + 1) * 4);
		b->md_off = (uint32_t *)realloc(b->md_off, (c + 1) * 4);
		b->md_rel = (uint32_t *)realloc(b->md_rel, c * 4);
		b->bound = (uint8_t *)realloc(b->bound, c);
		if (!b->md_rel || !b->bound) mDie("Out of memory");
		if (!b->rec_off || !b->flag || !b->rflags || !b->tid || !b->pos || !b->nm || !b->as || !b->cigar_off || !b->md_off)
			mDie("Out of memory");
		b->cap = c;
	}
}

static void rb_clear(rbatch *b) {
	b->n = 0;
	b->blob.l = 0;
	b->n_groups = 0;
	rb_reserve(b);
	b->rec_off[0] = 0;
	b->cigar_off[0] = 0;
	b->md_off[0] = 0;
}

#define RB_REC(b, i) ((b)->base + (b)->rec_off[i] + 4)
#define RB_LEN(b, i) ((b)->rec_off[(i) + 1] - (b)->rec_off[i] - 4)

static void rb_mark_group(rbatch *b) {   /* a pool starts at the record about to be appended */
	if (b->n_groups + 2 > b->group_cap) {
		b->group_cap = b->group_cap ? b->group_cap * 2 : 65536;
		b->group_off = (uint32_t *)realloc(b->group_off, b->group_cap * 4);
		if (!b->group_off) mDie("Out of memory");
	}
	b->group_off[b->n_groups++] = (uint32_t)b->n;
}

/* one pass over the aux block for MD, NM, AS (first occurrence wins, as bam_aux_get) */
static void rb_append(rbatch *b, const uint8_t *r, size_t len, int want_stats) {
	size_t i = b->n;
	uint32_t nc = (msh_rec_check(r, len), REC_NCIGAR(r));
	const uint8_t *p, *end = r + len, *md = NULL, *nm = NULL, *as = NULL;
	rb_reserve(b);
	{
		uint8_t b4[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
		ks_put(&b->blob, b4, 4);
	}
	ks_put(&b->blob, r, len);
	b->base = (const uint8_t *)b->blob.s;
	b->rec_off[i + 1] = b->blob.l;
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
	if (want_stats) {
		size_t ml = (md && *md == 'Z') ? strlen((const char *)md + 1) : 0;
		if (b->cigar_off[i] + nc + 4 > b->cigar_cap) {
			b->cigar_cap = (b->cigar_cap ? b->cigar_cap * 2 : 1 << 20) + nc;
			b->cigar = (uint32_t *)realloc(b->cigar, b->cigar_cap * 4);
		}
		if (b->md_off[i] + ml + 16 > b->md_cap) {
			b->md_cap = (b->md_cap ? b->md_cap * 2 : 1 << 22) + ml;
			b->md = (uint8_t *)realloc(b->md, b->md_cap);
		}
		if (!b->cigar || !b->md) mDie("Out of memory");
		memcpy(b->cigar + b->cigar_off[i], REC_CIGAR(r), 4 * (size_t)nc);
		if (ml) memcpy(b->md + b->md_off[i], md + 1, ml);
		b->cigar_off[i + 1] = b->cigar_off[i] + nc;
		b->md_off[i + 1] = b->md_off[i] + (uint32_t)ml;
	} else {
		b->cigar_off[i + 1] = b->cigar_off[i];
		b->md_off[i + 1] = b->md_off[i];
	}
	b->n++;
}

static void rb_host_view(rbatch *b, msx_batch *h, int with_groups) {
	static uint32_t zero_c[4];
	static uint8_t zero_m[16];
	memset(h, 0, sizeof(*h));
	h->n_records = (int64_t)b->n;
	h->flag = b->flag; h->rflags = b->rflags; h->tid = b->tid; h->pos = b->pos;
	h->cigar_off = b->cigar_off; h->cigar = b->cigar ? b->cigar : zero_c;
	h->md_off = b->md_off; h->md = b->md ? b->md : zero_m;
	h->nm = b->nm; h->as = b->as;
	if (with_groups) {
		b->group_off[b->n_groups] = (uint32_t)b->n;       /* sentinel */
		h->group_off = b->group_off;
		h->n_groups = (int64_t)b->n_groups;
	}
}

/* ------------------------------------------------------------------------ */
/* QNAME grouping preflight (msam_helper.c:78-137, 295-484) on the first      */
/* records of the stream; `first` holds at least COORD_ORDER_CHECK_RECORDS    */
/* records unless the input is shorter.                                       */
/* ------------------------------------------------------------------------ */
typedef enum { QN_NOT_REQUIRED = 0, QN_HEADER_CONFIRMED, QN_SAMPLE_OK, QN_SAMPLE_WARNING } qn_status;
typedef struct {
	qn_status status;
	size_t qname_records_checked, input_records_checked, mapped_records_checked;
} qn_result;

static void qn_format(const qn_result *r, char *buf, size_t n) {
	switch (r->status) {
	case QN_NOT_REQUIRED: snprintf(buf, n, "QNAME grouping check: not required for this operation"); break;
	case QN_HEADER_CONFIRMED: snprintf(buf, n, "QNAME grouping check: confirmed by input header SO:queryname"); break;
	case QN_SAMPLE_OK:
		if (r->input_records_checked < QNAME_GROUP_CHECK_RECORDS)
			snprintf(buf, n, "QNAME grouping check: no QNAME grouping violation detected in all %zu records",
			         r->qname_records_checked);
		else
			snprintf(buf, n, "QNAME grouping check: no QNAME grouping violation detected in first %zu records",
			         r->qname_records_checked);
		break;
	default:
		snprintf(buf, n,
		         "QNAME grouping check: WARNING - no QNAME grouping violation detected in first %zu records; "
		         "%zu mapped records among the first %zu input records were consistent with coordinate ordering",
		         r->qname_records_checked, r->mapped_records_checked, r->input_records_checked);
	}
}

/* open-addressing set of closed QNAMEs -> last record number of the group */
typedef struct { const char *name; size_t last; } qn_slot;
static uint64_t str_hash(const char *s) {
	uint64_t h = 1469598103934665603ull;
	while (*s) { h ^= (uint8_t)*s++; h *= 1099511628211ull; }
	return h;
}

static qn_result qn_check(const msh_hdr *hdr, const rbatch *first) {
	qn_result res = {QN_SAMPLE_OK, 0, 0, 0};
	char *so = msh_hdr_sort_order(hdr);
	size_t i, cap = 1 << 15, limit;
	qn_slot *tab;
	const char *cur = NULL;
	size_t cur_first = 0;
	int coordinate_ordered = 1, coordinate_relevant = 0, have_prev = 0;
	int32_t prev_tid = -1, prev_pos = -1;
	if (so) {                                              /* header declarations are authoritative */
		if (strcmp(so, "queryname") == 0) { res.status = QN_HEADER_CONFIRMED; free(so); return res; }
		if (strcmp(so, "coordinate") == 0)
			mDie("Input SAM/BAM declares 'SO:coordinate', but this operation requires records to be grouped by QNAME.\n"
			     "             Please name-sort the input, for example with "
			     "'samtools sort -n input.bam -o input.name_sorted.bam'.");
		free(so);
	}
	tab = (qn_slot *)calloc(cap, sizeof(qn_slot));
	limit = first->n < COORD_ORDER_CHECK_RECORDS ? first->n : COORD_ORDER_CHECK_RECORDS;
	for (i = 0; i < limit; i++) {
		const uint8_t *r = RB_REC(first, i);
		const char *q = REC_QNAME(r);
		size_t recno = i + 1;
		res.input_records_checked++;
		if (recno <= QNAME_GROUP_CHECK_RECORDS) {
			res.qname_records_checked++;
			if (!cur) { cur = q; cur_first = recno; }
			else if (strcmp(q, cur) != 0) {
				uint64_t h = str_hash(cur) & (cap - 1);
				while (tab[h].name && strcmp(tab[h].name, cur) != 0) h = (h + 1) & (cap - 1);
				tab[h].name = cur;
				tab[h].last = recno - 1;
				h = str_hash(q) & (cap - 1);
				while (tab[h].name && strcmp(tab[h].name, q) != 0) h = (h + 1) & (cap - 1);
				if (tab[h].name)
					mDie("SAM/BAM file is not grouped by QNAME. Read '%s' reappears at record %zu after its previous "
					     "group ended at record %zu (%zu intervening records). Please name-sort the input, for example "
					     "with 'samtools sort -n input.bam -o input.name_sorted.bam'.",
					     q, recno, tab[h].last, recno - tab[h].last - 1);
				cur = q;
				cur_first = recno;
			}
		}
		if (!(first->flag[i] & 4) && first->tid[i] >= 0) {
			res.mapped_records_checked++;
			if (have_prev && (first->tid[i] < prev_tid || (first->tid[i] == prev_tid && first->pos[i] < prev_pos)))
				coordinate_ordered = 0;
			prev_tid = first->tid[i];
			prev_pos = first->pos[i];
			have_prev = 1;
		}
		if (first->flag[i] & (0x1 | 0x100 | 0x800)) coordinate_relevant = 1;
	}
	(void)cur_first;
	free(tab);
	if (coordinate_ordered && coordinate_relevant && res.mapped_records_checked >= COORD_ORDER_MIN_RECORDS) {
		char w[1024];
		res.status = QN_SAMPLE_WARNING;
		qn_format(&res, w, sizeof w);
		fprintf(stderr, "WARNING: %s\n", w);
	}
	return res;
}

/* ------------------------------------------------------------------------ */
/* filter                                                                     */
/* ------------------------------------------------------------------------ */
static void filter_help(FILE *out) {
	fprintf(out,
	        "Usage:\n------\n\n%s filter [-buhSkv] <bamfile> [--help] [-l <int>] [-p <int>] [--ppt=<int>] [-z <int>] "
	        "[--rescore] [--besthit] [--uniqhit]\n"
	        "\nGeneral options:\n----------------\n\n"
	        "These options specify the input/output formats of BAM/SAM files \n(same meaning as in 'samtools view'):\n"
	        "  -b                        output BAM (default: false)\n"
	        "  -u                        uncompressed BAM output (force -b) (default: false)\n"
	        "  -h                        print header for the SAM output (default: false)\n"
	        "  -S                        input is SAM (default: false)\n"
	        "  <bamfile>                 input SAM/BAM file\n"
	        "  --help                    print this help and exit\n\n"
	        "Specific options:\n-----------------\n\n"
	        "  -l <int>                  min. length of alignment (default: 0)\n"
	        "  -p <int>                  min. sequence identity of alignment, in percentage, integer between 0 and 100; "
	        "requires MD or NM field to be present (default: 0)\n"
	        "  --ppt=<int>               min/max sequence identity of alignment, in parts per thousand, integer between "
	        "-1000 and 1000; requires MD or NM field to be present (default: 0)\n"
	        "  -z <int>                  min. percent of the query that must be aligned, between 0 and 100 (default: 0)\n"
	        "  -k, --keep_unmapped       report unmapped reads, when filtering using upper-limit thresholds (default: false)\n"
	        "  -v, --invert              invert the effect of the filter (default: false)\n"
	        "  --rescore                 rescore alignments using MD or NM fields, in that order (default: false)\n\n"
	        "Special filters:\n----------------\n\n"
	        "  --besthit                 keep all highest scoring hit(s) per read (default: false)\n"
	        "  --uniqhit                 keep only one highest scoring hit per read, only if it is unique (default: false)\n"
	        "\nOne-process pipe (MI355X build):\n--------------------------------\n\n"
	        "  --profile-out=<file>      also write the profile `msamtools profile -` would estimate from this command's output\n"
	        "                            (needs --label; takes profile's --genome --total --mincount --unit --pandas --no-pandas --nolen --multi)\n",
	        PROGRAM);
}

typedef struct {
	msh_in *in;
	kstr rec;
	int have_pending;            /* rec holds a record read but not yet batched */
	int eof;
	/* R5 grouping state (msam_filter.c:107,117-125,170) */
	char prev_read[256];
	int have_prev;
	/* bulk (BAM) path */
	size_t consume_pending;      /* span bytes used by the batch being processed */
	int done;
} reader;

/* Fill `b` with up to `target` records; with pools, stop at the first pool
 * boundary at or after the target so no pool straddles two batches. */
static void fill_filter_batch(reader *rd, rbatch *b, size_t target, int pools, int want_stats) {
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
			b->cigar_off[i + 1] = REC_NCIGAR(r);           /* counts; prefix-summed afterwards */
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
		if (nc) memcpy(b->cigar + b->cigar_off[i], REC_CIGAR(r), 4 * (size_t)nc);
		if (ml) memcpy(b->md + b->md_off[i], r + b->md_rel[i], ml);
	}
}

static void fill_batch_bulk(reader *rd, rbatch *b, size_t target, int mode, int want_stats) {
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
		if (nc + 4 > b->cigar_cap) { b->cigar_cap = nc + nc / 4 + 1024; b->cigar = (uint32_t *)realloc(b->cigar, b->cigar_cap * 4); }
		if (nm + 16 > b->md_cap) { b->md_cap = nm + nm / 4 + 4096; b->md = (uint8_t *)realloc(b->md, b->md_cap); }
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

/* ------------------------------------------------------------------------ */
/* The pipelined BAM path: decode | device | encode run as three stages on    */
/* their own threads over a ring of batch slots, the parallel parts of every  */
/* stage on the shared worker pool (counterpart of the loop msam_filter.c:    */
/* 119-186 / msam_helper.c:246-272, whose read, compute and write are one     */
/* thread).  A slot owns the inflated bytes of its batch, so the writer can   */
/* still copy records out of batch i while batch i+1 is being decoded; the    */
/* bytes of the pool left open at a batch's end are carried into the next.    */
/* ------------------------------------------------------------------------ */
#define PIPE_SLOTS 3                    /* with one device; one more per further device */
#define PIPE_SLOTS_MAX (PIPE_SLOTS + MSH_MAX_DEVICES)
#define PIPE_OBUFS 4                    /* page-locked output buffers of the device-unpack path */
#define PIPE_SLOTS_COMP 6               /* slots when the batches arrive compressed (a slot is 40 MB of payloads then) */
#define MSH_POOL_MAX 128
#define PQ_END (-1)                     /* queue item: end of the stream (one per consumer) */
#define BGZF_INFLATE_MAX ((size_t)1024 * 65536)   /* msh_inflate_append appends at most one batch of blocks (msh_io.c: BGZF_BATCH x BGZF_MAX) */

typedef struct {
	pthread_mutex_t mu;
	pthread_cond_t cv;
	int item[2 * PIPE_SLOTS_MAX + 4], n;
} pq;
static void pq_init(pq *q) { pthread_mutex_init(&q->mu, NULL); pthread_cond_init(&q->cv, NULL); q->n = 0; }
static void pq_push(pq *q, int v) {
	pthread_mutex_lock(&q->mu);
	q->item[q->n++] = v;
	pthread_cond_signal(&q->cv);
	pthread_mutex_unlock(&q->mu);
}
static int pq_pop(pq *q) {
	int v, i;
	pthread_mutex_lock(&q->mu);
	while (q->n == 0) pthread_cond_wait(&q->cv, &q->mu);
	v = q->item[0];
	for (i = 1; i < q->n; i++) q->item[i - 1] = q->item[i];
	q->n--;
	pthread_mutex_unlock(&q->mu);
	return v;
}

#define PQ_NONE (-2)                    /* pq_try_pop: the queue is empty */
static int pq_try_pop(pq *q) {
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

typedef struct {
	rbatch b;                  /* fixed-capacity SoA (page-locked by the device stage) + rec_off; b.base = ubuf */
	uint8_t *ubuf;             /* inflated BAM bytes of this batch */
	size_t ulen, ucap;
	int32_t *emit;             /* filter: indices of the records to write, in output order */
	int32_t *as_out;           /* --rescore: the AS every record carries on output */
	int64_t n_emit;
	size_t seq;                /* number of the batch in the input: the writer's order */
	int eof;                   /* end-of-stream marker */
	int pinned;
	/* device unpack (msx_unpack): the slot carries inflated bytes only, cut anywhere */
	int raw, last;             /* raw: ubuf[0, ulen) is all there is; last: nothing follows */
	int has_seed, seed_has_name;
	kstr seed;                 /* what the host-side reader left over in front of the first raw slot */
	char seed_name[256];
	uint8_t *rbuf;             /* raw slots: the inflated bytes, in a buffer of fixed size (page-locked by pin_thread) */
	size_t rcap, rlen;
	/* device inflate (msx_unpack_enqueue_bgzf): rbuf holds the blocks' DEFLATE payloads instead, blk their table */
	int comp, n_blk;
	msx_bgzf_block *blk;
	size_t inflated;           /* bytes the table's blocks inflate to */
	int pin_ready;             /* rbuf is page-locked and obuf allocated (pin_mu) */
	int ob;                    /* the output buffer this batch holds (pipe_t.ob[]), -1: none */
	uint8_t *obuf;             /* = P->ob[ob]: filter's output records of the batch (msx_unpack_emit_fetch), page-locked */
	msx_event *ev_out;         /* ... are there once this has been waited for */
	msx_ctx *ev_ctx;
	size_t ocap, olen;
	int framed;                /* obuf holds finished BGZF blocks (msx_unpack_emit_gather_bgzf), not a bare record stream */
	int fatal;                 /* the batch holds a record the reference dies at: fatal_msg, after the pools before it */
	char fatal_msg[512];
} pslot;

typedef struct {
	msh_in *in;
	const msh_hdr *hdr;
	int mode, want_stats;      /* pool rule (0 none / 1 filter / 2 profile / 3 profile's rule over what filter can write), cigar+md wanted */
	int unmapped_visible;      /* mode 3: -k -v with a PPT >= 0 filter writes unmapped records (msam_filter.c:132-138) */
	int cut_mapped;            /* prefer batch ends in front of a pool that begins with a MAPPED record (an insert's first pool) */
	int n_slots, n_consumers;
	int raw_mode;              /* batches after the first are unpacked on the device: the decode stage only inflates */
	size_t n_host_inflated;        /* batches the device inflater refused */
	size_t n_comp_done;            /* compressed batches the device stage is through with */
	int comp_given_up;             /* (decode thread) the input's blocks are inflated on the host from here on */
	size_t n_ahead;                /* batches whose blocks were sent and inflated ahead (msx_unpack_prefetch_bgzf) */
	int comp_mode, comp_blocks;    /* ... and inflated there as well: the decode stage only copies the blocks' payloads */
	size_t ocap_cfg;
	/* output buffers are a pool of their own: a slot is the decode stage's unit (a buffer of compressed blocks), and the
	 * decoder must not run out of slots because the writer still holds the outputs of earlier batches */
	uint8_t *ob[PIPE_OBUFS];
	size_t ob_cap[PIPE_OBUFS];
	pq q_ob;
	int raw_started, raw_done;
	int first_state, first_slot;   /* 0: batch 0 not decoded yet; 1: it is, in slot first_slot; 2: the input holds no record (first_mu) */
	int out_opened;                /* the preflight has passed and the output is open (first_mu) */
	pthread_mutex_t first_mu;
	pthread_cond_t first_cv;
	int with_obuf;             /* filter: there are output buffers as well */
	int pin_started, pin_quit;
	msx_ctx *pin_ctx;
	pthread_t pin_th[PIPE_SLOTS_MAX];
	int n_pin;
	struct pin_arg_s { void *P; int first, step; } pin_args[PIPE_SLOTS_MAX];
	pthread_mutex_t pin_mu;
	pthread_cond_t pin_cv;
	size_t n_filled;           /* batches handed on so far */
	size_t batch_bytes, batch_bytes_cfg, cap_rec, cap_cig, cap_md;
	pslot slot[PIPE_SLOTS_MAX];
	pq q_free, q_dev, q_out;
	/* decode state */
	kstr carry;                /* bytes of the open pool (and of a cut record) left by the previous batch */
	char prev_read[256];
	int have_prev, in_eof, have_first;
	size_t **seg_list, *seg_cnt, *seg_end, *seg_from, *seg_extra_n;
	size_t **seg_extra;
	int nseg_cap;
	double t_decode, t_wait_free;
	double t_inflate, t_chase, t_scan, t_serial, t_copy;    /* inside t_decode */
} pipe_t;

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

static void *xmalloc(size_t n) {
	void *p = malloc(n ? n : 1);
	if (!p) mDie("Out of memory");
	return p;
}

static void pipe_init(pipe_t *P, msh_in *in, int mode, int want_stats, int n_consumers) {
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
	pq_init(&P->q_free); pq_init(&P->q_dev); pq_init(&P->q_out); pq_init(&P->q_ob);
	for (i = 0; i < PIPE_SLOTS_MAX; i++) P->slot[i].ob = -1;
	pthread_mutex_init(&P->first_mu, NULL);
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
static uint8_t *io_alloc(size_t bytes) {
	const size_t al = (size_t)2 << 20, len = (bytes + al - 1) / al * al + al;
	uint8_t *m = (uint8_t *)mmap(NULL, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0), *p;
	if (m == MAP_FAILED) mDie("Out of memory");
	p = (uint8_t *)(((uintptr_t)m + al - 1) / al * al);
#ifdef MADV_HUGEPAGE
	(void)madvise(p, len - (size_t)(p - m), MADV_HUGEPAGE);
#endif
	return p;                              /* (never unmapped: the buffers live as long as the process) */
}
static void io_populate(uint8_t *p, size_t bytes) {
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
			pq_push(&P->q_ob, j);
		}
	}
	return NULL;
}
/* the threads are started by pipe_enable_raw (they populate the buffers); the device thread hands them its context here */
static void pin_start(pipe_t *P, int with_obuf) {
	(void)with_obuf;
	if (!P->pin_started || P->pin_ctx) return;
	pthread_mutex_lock(&P->pin_mu);
	P->pin_ctx = g_ctx;
	pthread_cond_broadcast(&P->pin_cv);
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
static void pin_join(pipe_t *P) {
	int t;
	if (!P->pin_started) return;
	pthread_mutex_lock(&P->pin_mu);
	P->pin_quit = 1;                                  /* (threads that were never given a context) */
	pthread_cond_broadcast(&P->pin_cv);
	pthread_mutex_unlock(&P->pin_mu);
	for (t = 0; t < P->n_pin; t++) pthread_join(P->pin_th[t], NULL);
}
static void pin_wait(pipe_t *P, pslot *s) {
	pthread_mutex_lock(&P->pin_mu);
	while (!s->pin_ready) pthread_cond_wait(&P->pin_cv, &P->pin_mu);
	pthread_mutex_unlock(&P->pin_mu);
}

/* a raw slot's bytes to the device and the record walk over them.  Compressed slots are inflated there; a batch with a
 * block the device inflater refuses is inflated here instead -- by the reader's own decoder and zlib, whose diagnostics
 * are the command's -- and handed over inflated. */
static void unpack_slot_enqueue(pipe_t *P, pslot *s, msx_unpack *unpack, const msx_unpack_params *up) {
	(void)P;
	if (!s->comp) MSX(msx_unpack_enqueue(g_ctx, unpack, s->rbuf, s->rlen, up));
	else MSX(msx_unpack_enqueue_bgzf(g_ctx, unpack, s->rbuf, s->rlen, s->blk, s->n_blk, up));
}
/* The batches behind this one, as far as the decode stage has them ready (two at most): their blocks start their way up
 * and are inflated beside the work on this batch.  `ahead` is the caller's queue of slots taken off q_dev ahead of their
 * turn (in order; an end-of-stream token or a slot that cannot be sent ahead closes it). */
typedef struct { int item[2], n, closed; } ahead_q;
static void unpack_slots_ahead(pipe_t *P, msx_unpack *unpack, ahead_q *A) {
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
static int ahead_pop(ahead_q *A) {
	int v;
	if (A->n == 0) return PQ_NONE;
	v = A->item[0];
	A->item[0] = A->item[1];
	if (--A->n == 0) A->closed = 0;
	return v;
}
static void unpack_slot_finish(pipe_t *P, pslot *s, msx_unpack *unpack, const msx_unpack_params *up, msx_unpack_result *ur, msx_batch *db) {
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
}

/* device unpack: every slot gets a buffer of fixed size for the inflated bytes (page-locked by pin_thread) */
static void pipe_enable_raw(pipe_t *P, int with_obuf) {
	int i;
	P->raw_mode = 1;
	P->with_obuf = with_obuf;
	/* BAM input: the blocks stay compressed until they are on the device (MSX_HOST_INFLATE=1: inflate here).  A batch is
	 * as many blocks as the device inflates at a time -- one wave per block, eight per compute unit (msx_inflate.hip) --
	 * or what fits the slot's buffer, whichever comes first. */
	P->comp_mode = msh_is_bam(P->in) && !getenv("MSX_HOST_INFLATE");
	/* (MSX_BATCH_BYTES, the inflated size of a batch, translates into blocks) */
	P->comp_blocks = (int)env_size("MSX_COMP_BLOCKS", getenv("MSX_BATCH_BYTES") ? P->batch_bytes_cfg / 65280 : 2048);
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
			s->blk = (msx_bgzf_block *)xmalloc((size_t)P->comp_blocks * sizeof(msx_bgzf_block));
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
static int chase_sloppy = -1;      /* MSX_CHASE_SLOPPY=1 (tests): accept almost anything as a record start, so that
                                      most guesses are wrong and the stitcher has to repair them */
static int rec_plausible(const uint8_t *u, size_t off, size_t len, int32_t nt) {
	const uint8_t *r;
	int32_t bs, tid, pos, mtid, mpos, ls;
	uint32_t lq, nc, k;
	if (off + 36 > len) return 0;
	bs = le32(u + off);
	if (bs < 32 || bs > (64 << 20)) return 0;
	{
		int sl = __atomic_load_n(&chase_sloppy, __ATOMIC_RELAXED);      /* (every thread would compute the same value) */
		if (sl < 0) { sl = getenv("MSX_CHASE_SLOPPY") != NULL; __atomic_store_n(&chase_sloppy, sl, __ATOMIC_RELAXED); }
		if (sl) return bs < 4096;
	}
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
	pack_job J;
	s->ulen = 0;
	if (P->carry.l) {
		if (P->carry.l + 64 > s->ucap) { s->ucap = P->carry.l + P->batch_bytes + 64; s->ubuf = (uint8_t *)realloc(s->ubuf, s->ucap); if (!s->ubuf) mDie("Out of memory"); }
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
		while (!P->in_eof && s->ulen < want)
			if (!pipe_append(P->in, &s->ubuf, &s->ulen, &s->ucap)) P->in_eof = 1;
		tq2 = now_s(); P->t_inflate += tq2 - tq; tq = tq2;
		if (s->ulen == 0) return 0;
		n = chase_records(P, b, s->ubuf, s->ulen, P->cap_rec, &tail);
		tq2 = now_s(); P->t_chase += tq2 - tq; tq = tq2;
		if (P->in_eof && n < P->cap_rec && tail != s->ulen) mDie("Truncated BAM record");
		if (!P->have_first && n < COORD_ORDER_CHECK_RECORDS && !P->in_eof) {   /* the preflight window (msam_helper.c:4-6) */
			P->batch_bytes += P->batch_bytes;
			continue;
		}
		if (n == 0) {
			if (P->in_eof) return 0;
			P->batch_bytes += P->batch_bytes;            /* a record larger than the batch: read on */
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
				P->batch_bytes += P->batch_bytes / 2;
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

static void *pipe_decode_thread(void *arg) {
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
					while (!P->in_eof && s->rlen < P->batch_bytes_cfg && s->rlen + (per + 1) * 65536 + 64 <= s->rcap)
						if (!pipe_append(P->in, &s->rbuf, &s->rlen, &s->rcap)) P->in_eof = 1;
					msh_inflate_limit(0);
					if (s->rlen == before && !P->in_eof) mDie("The batch buffers are too small for a BGZF block (MSX_COMP_BYTES)");
				} else if (P->comp_mode) {
					while (!P->in_eof && s->n_blk < P->comp_blocks && (s->rcap - s->rlen) / (65536 + 1024) > 0)
						if (!msh_raw_append(P->in, s->rbuf, s->rcap, &s->rlen, s->blk, &s->n_blk, P->comp_blocks, &s->inflated)) P->in_eof = 1;
				} else
				while (!P->in_eof && s->rlen < P->batch_bytes_cfg && s->rlen + BGZF_INFLATE_MAX + 64 <= s->rcap)
					if (!pipe_append(P->in, &s->rbuf, &s->rlen, &s->rcap)) P->in_eof = 1;
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
			/* end of the stream: one token per consumer (P->n_filled is final from here on) */
			int c;
			for (c = 0; c < P->n_consumers; c++) pq_push(&P->q_dev, PQ_END);
			return NULL;
		}
		s->seq = P->n_filled;
		__atomic_store_n(&P->n_filled, P->n_filled + 1, __ATOMIC_RELEASE);
		pq_push(&P->q_dev, si);
	}
}

/* page-lock the slot's SoA arrays once (they never move): uploads become asynchronous DMA */
static void pipe_pin_slot(pipe_t *P, pslot *s) {
	rbatch *b = &s->b;
	/* (device unpack: only batch 0 takes the host-side walk -- of its arrays just what it uses is page-locked: 170 MB for
	 * one upload would cost more than the upload saves, and from pageable memory the dozen copies took 77 ms) */
	const size_t c = P->raw_mode ? b->n + 8 : b->cap;
	if (s->pinned || getenv("MSX_NO_PIN")) return;
	s->pinned = 1;
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
		MSX(msx_host_register(g_ctx, b->cigar, P->raw_mode ? ((size_t)b->cigar_off[b->n] + 8) * 4 : b->cigar_cap * 4));
		MSX(msx_host_register(g_ctx, b->md, P->raw_mode ? (size_t)b->md_off[b->n] + 64 : b->md_cap));
	}
	if (P->mode != 0) MSX(msx_host_register(g_ctx, b->group_off, P->raw_mode ? (b->n_groups + 8) * 4 : b->group_cap * 4));
}

/* --rescore: drop the first AS and append AS:i (msam_filter.c:162-167) */
static void rescore_record(const uint8_t *r, size_t len, int32_t score, kstr *out) {
	const uint8_t *as = msh_aux_get(r, len, "AS");
	out->l = 0;
	if (as) {
		size_t sz = msh_aux_size(as, r + len);
		ks_put(out, r, (size_t)(as - 2 - r));
		ks_put(out, as + sz, (size_t)(r + len - (as + sz)));
	} else {
		ks_put(out, r, len);
	}
	ks_put(out, "ASi", 3);
	{
		uint8_t b4[4] = {(uint8_t)score, (uint8_t)((uint32_t)score >> 8), (uint8_t)((uint32_t)score >> 16),
		                 (uint8_t)((uint32_t)score >> 24)};
		ks_put(out, b4, 4);
	}
}

/* ------------------------------------------------------------------------ */
/* profile                                                                    */
/* ------------------------------------------------------------------------ */
static void profile_help(FILE *out) {
	fprintf(out,
	        "Usage:\n------\n\n%s profile [-S] <bamfile> [--help] -o <file> --label=<string> [--genome=<string>] "
	        "[--total=<int>] [--mincount=<int>] [--unit=<string>] [--pandas] [--no-pandas] [--nolen] [--multi=<string>]\n"
	        "\nGeneral options:\n----------------\n\n"
	        "These options specify the input/output formats of BAM/SAM files \n(same meaning as in 'samtools view'):\n"
	        "  -S                        input is SAM (default: false)\n"
	        "  <bamfile>                 input SAM/BAM file\n"
	        "  --help                    print this help and exit\n\n"
	        "Specific options:\n-----------------\n\n"
	        "  -o <file>                 name of output file (required)\n"
	        "  --label=<string>          label to use for the profile; typically the sample id (required)\n"
	        "  --genome=<string>         tab-delimited genome definition file - 'genome-id<tab>seq-id' (default: none)\n"
	        "  --total=<int>             number of high-quality inserts (mate-pairs/paired-ends) that were input to the aligner (default: unknown)\n"
	        "  --mincount=<int>          minimum number of inserts mapped to a feature, below which the feature is counted as absent (default: 0)\n"
	        "  --unit=<string>           unit of abundance to report {ab | rel | fpkm | tpm} (default: rel)\n"
	        "  --pandas                  print two columns (ID, sample-label) as header compatible with python pandas (default)\n"
	        "  --no-pandas               use legacy profile header without the ID column\n"
	        "  --nolen                   do not normalize the abundance (only relevant for ab or rel) for sequence length (default: normalize)\n"
	        "  --multi=<string>          how to deal with multi-mappers {all | equal | proportional | ignore} (default: proportional)\n",
	        PROGRAM);
}

/* mPrintInsertStats / mPrintInsertStatsDouble (msam_profile.c:434-499); the text goes to a buffer */
static void print_stats_int(kstr *s, int left, const char *type, int number, int total, const char *post) {
	int width = 7;
	if (total > 0) width = (int)(1 + log10(total));
	ks_printf(s, "# ");
	if (left) ks_printf(s, "%-20s: ", type); else ks_printf(s, "%20s: ", type);
	if (strcmp(type, "Total inserts") == 0 && number == -1) ks_printf(s, "%*s (", width, "NA");
	else ks_printf(s, "%*d (", width, number);
	if (total > 0) ks_printf(s, "%6.2f", 100.0 * number / total); else ks_printf(s, "%6s", "NA");
	ks_printf(s, "%%)");
	if (post) ks_printf(s, " %s\n", post); else ks_printf(s, "\n");
}
static void print_stats_dbl(kstr *s, int left, const char *type, double number, int total, const char *post) {
	ks_printf(s, "# ");
	if (left) ks_printf(s, "%-20s: ", type); else ks_printf(s, "%20s: ", type);
	ks_printf(s, "%10.7g (", number);
	if (total > 0) ks_printf(s, "%6.2f", 100.0 * number / total); else ks_printf(s, "%6s", "NA");
	ks_printf(s, "%%)");
	if (post) ks_printf(s, " %s\n", post); else ks_printf(s, "\n");
}



/* ---- what `profile` and `filter --profile-out` share: options, features, the report ------------------------ */
typedef struct {
	const char *out, *label, *genome, *unit, *multi;
	int n_out, n_label, n_total, n_mincount, pandas, nopandas, nolen;
	long v_total, v_mincount;
	/* derived (prof_opts_derive) */
	int share_type, unit_type, length_normalize, total_inserts;
} prof_opts;

typedef struct {
	int32_t n_features, *fmap;
	char **name;
	uint32_t *len;
} prof_feat;

/* msam_profile.c:712-755: --multi and --unit by prefix match, length normalisation */
static void prof_opts_derive(prof_opts *o) {
	int i;
	o->total_inserts = o->n_total > 0 ? (int)o->v_total : -1;
	o->share_type = MSX_MULTI_SHARE_PROPORTIONAL;                     /* :712-728, prefix match */
	if (o->multi) {
		const char *types[5] = {"", "all", "equal", "proportional", "ignore"};
		o->share_type = -1;
		for (i = 1; i <= 4; i++)
			if (strncmp(o->multi, types[i], strlen(o->multi)) == 0) { o->share_type = i; break; }
		if (o->share_type == -1) mDie("Do not understand --multi=%s", o->multi);
	}
	o->unit_type = 1;                                                 /* :732-748 */
	if (o->unit) {
		const char *types[5] = {"", "relative", "fpkm", "tpm", "abundance"};
		o->unit_type = -1;
		for (i = 1; i <= 4; i++)
			if (strncmp(o->unit, types[i], strlen(o->unit)) == 0) { o->unit_type = i; break; }
		if (o->unit_type == -1) mDie("Do not understand --unit=%s", o->unit);
	}
	o->length_normalize = 1;
	if (o->unit_type == 1 || o->unit_type == 4) o->length_normalize = (o->nolen == 0);   /* :752-755 */
}

static void prof_features(const prof_opts *o, const msh_hdr *hdr, prof_feat *F) {
	F->fmap = NULL;
	if (o->genome) {
		F->fmap = msh_genome_map(o->genome, hdr, &F->n_features, &F->name, &F->len);     /* :757-852 */
	} else {
		F->n_features = hdr->n_targets;
		F->name = hdr->target_name;
		F->len = hdr->target_len;
	}
}

/* The profile's text, "%s\t%.8g\n" per feature (mMatrix.c:359-376), is formatted and gzip-compressed by all threads:
 * every thread's share of the lines becomes a gzip member of its own, the members are written in order.  A gzip file
 * of several members decompresses to the concatenation (RFC 1952 2.2; zcat, zlib's gzread, Python and R read it as
 * one text).  With a million features the single gzprintf stream of the reference took a third of a second here.
 * MSX_GZ_SINGLE=1: one member. */
typedef struct {
	const prof_feat *F;
	const double *row;
	int32_t n;
	kstr text[MSH_POOL_MAX], gz[MSH_POOL_MAX];
} report_job;

/* one gzip member from a text buffer */
static void gz_member(const kstr *in, kstr *out) {
	z_stream zs;
	size_t bound;
	memset(&zs, 0, sizeof zs);
	/* level 4 rather than gzip's 6: the million-line profile of the bench compresses in 25 ms instead of 59 per thread and
	 * comes out 5 % larger (7.1 MB instead of 6.8); MSX_GZ_LEVEL=6 for the reference's level */
	static int level = 0;
	if (!level) { const char *e = getenv("MSX_GZ_LEVEL"); level = e && atoi(e) >= 1 && atoi(e) <= 9 ? atoi(e) : 4; }
	if (deflateInit2(&zs, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) mDie("deflateInit2 failed");
	bound = deflateBound(&zs, (uLong)in->l) + 64;
	out->l = 0;
	ks_reserve(out, bound);
	zs.next_in = (Bytef *)in->s; zs.avail_in = (uInt)in->l;
	zs.next_out = (Bytef *)out->s; zs.avail_out = (uInt)bound;
	if (deflate(&zs, Z_FINISH) != Z_STREAM_END) mDie("deflate failed");
	out->l = bound - zs.avail_out;
	deflateEnd(&zs);
}

static void report_worker(void *arg, int tid, int nth) {
	report_job *J = (report_job *)arg;
	const int32_t lo = (int32_t)((int64_t)J->n * tid / nth), hi = (int32_t)((int64_t)J->n * (tid + 1) / nth);
	kstr *k = &J->text[tid];
	int32_t i;
	for (i = lo; i < hi; i++) ks_printf(k, "%s\t%.8g\n", J->F->name[i], J->row[1 + i]);
	gz_member(k, &J->gz[tid]);
}

static void fd_write_all(int fd, const void *p, size_t n) {
	const uint8_t *s = (const uint8_t *)p;
	while (n) {
		ssize_t k = write(fd, s, n);
		if (k < 0 && errno == EINTR) continue;
		if (k <= 0) mDie("Write failed");
		s += k; n -= (size_t)k;
	}
}

/* msam_profile.c:858-983 + mMatrix.c:137-179,359-376: post-processing and the text, in the reference's order.
 * row[0] = Unknown, row[1 + i] = abundance of feature i as mInsertCountToAbundanceMatrix left it. */
static void profile_report(const prof_opts *o, const prof_feat *F, const msx_profile_stats *st, double *row,
                           const qn_result *qn, const char *cl) {
	const int32_t n_features = F->n_features;
	int total_inserts = o->total_inserts, mapped_inserts = (int)st->insert_count, i;
	double purged_insert_equivalent = 0, purged_inserts, effective_inserts;
	char qmsg[1024];
	kstr head = {0, 0, 0};
	int fd;
	if (o->share_type == MSX_MULTI_SHARE_PROPORTIONAL) {
		int k;
		for (k = 1; k <= st->iterations; k++)
			fprintf(stderr, "#     PropSharing Iteration: %2d; DELTA^2=%g%s\n", k, st->delta[k],
			        (k == st->iterations && st->converged) ? ". CONVERGED!" : "");
		fprintf(stderr, "# End   PropSharing!\n");
		fprintf(stderr, "# Purged %d inserts that mapped to features without unique inserts.\n",
		        (int)st->purged_insert_count);
	}
	row[0] = 0.0;
	if (o->n_mincount > 0) {                                          /* :858-869 */
		int mincount = (int)o->v_mincount;
		for (i = 1; i < n_features + 1; i++)
			if (row[i] < mincount) { purged_insert_equivalent += row[i]; row[i] = 0; }
		fprintf(stderr, "# Purged %.7g insert-equivalents from low-abundance features based on --mincount.\n",
		        purged_insert_equivalent);
	}
	if (total_inserts > 0 && total_inserts < mapped_inserts) {        /* :873-876 */
		fprintf(stderr, "# Ignoring 'unknown' fraction, as total inserts (%d) < mapped inserts (%d)!\n", total_inserts,
		        mapped_inserts);
		total_inserts = -1;
	}
	fd = strcmp(o->out, "-") == 0 ? fileno(stdout) : open(o->out, O_WRONLY | O_CREAT | O_TRUNC, 0666);   /* :879-883 */
	if (fd < 0) mDie("Cannot open %s for writing", o->out);
	qn_format(qn, qmsg, sizeof qmsg);
	ks_printf(&head, "# msamtools version: %s\n", MSH_VERSION);           /* msam_helper.c:145-148 */
	ks_printf(&head, "# msamtools git commit: %s\n", MSH_GIT_COMMIT);
	ks_printf(&head, "# Command line: %s\n", cl);
	ks_printf(&head, "# %s\n", qmsg);
	purged_inserts = st->purged_insert_count + purged_insert_equivalent;   /* :889-903 */
	effective_inserts = mapped_inserts - purged_inserts;
	if (o->share_type == MSX_MULTI_IGNORE) effective_inserts -= st->multi_mapper_count;
	print_stats_int(&head, 1, "Total inserts", total_inserts, total_inserts, NULL);
	print_stats_int(&head, 1, "Mapped inserts", mapped_inserts, total_inserts, NULL);
	print_stats_int(&head, 0, "- Multiple mapped ", (int)st->multi_mapper_count, total_inserts, NULL);
	print_stats_int(&head, 0, "- Uniquely mapped ", (int)st->uniq_mapper_count, total_inserts, NULL);
	print_stats_dbl(&head, 1, "Purged inserts", purged_inserts, total_inserts,
	                "due to ambiguous mapping or low abundance features");
	print_stats_dbl(&head, 1, "Effective inserts", effective_inserts, total_inserts, NULL);
	if (total_inserts <= 0) ks_printf(&head, "# Estimated seq. length for 'Unknown': NA\n");
	if (total_inserts > 0) {                                          /* :906-934 */
		row[0] = total_inserts - mapped_inserts + purged_inserts;
		if (o->share_type == MSX_MULTI_IGNORE) row[0] += st->multi_mapper_count;
		if (o->length_normalize) {
			int count = 0;
			uint64_t sum = 0;
			uint32_t unknown_size;
			for (i = 0; i < n_features; i++) { sum += F->len[i]; count++; }
			unknown_size = (uint32_t)(sum / (uint64_t)count);
			ks_printf(&head, "# Estimated seq. length for 'Unknown': %dbp\n", unknown_size);
			row[0] = 1.0 * row[0] / unknown_size;
		} else {
			ks_printf(&head, "# Estimated seq. length for 'Unknown': NA\n");
		}
	}
	if (o->length_normalize)                                          /* :937-947 */
		for (i = 0; i < n_features; i++) row[1 + i] /= F->len[i];
	switch (o->unit_type) {                                           /* :950-975, mMatrix.c:137-179 */
	case 2: {
		double d = total_inserts > 0 ? 1.0E9 / total_inserts : 1.0E9 / mapped_inserts;
		for (i = 0; i < n_features + 1; i++) row[i] *= d;
		break;
	}
	case 3:
	case 1: {
		double sum = 0;
		for (i = 0; i < n_features + 1; i++) sum += row[i];
		for (i = 0; i < n_features + 1; i++) row[i] /= sum;
		if (o->unit_type == 3)
			for (i = 0; i < n_features + 1; i++) row[i] *= 1.0E6;
		break;
	}
	default: break;
	}
	if (o->nopandas == 0) ks_printf(&head, "ID\t");                       /* mMatrix.c:359-376 */
	ks_printf(&head, "%s\n", o->label);
	ks_printf(&head, "Unknown\t%.8g\n", row[0]);
	{
		static report_job J;
		int nth = msh_threads(), t;
		kstr hz = {0, 0, 0};
		if (nth > MSH_POOL_MAX) nth = MSH_POOL_MAX;
		if (n_features < 4096 || getenv("MSX_GZ_SINGLE")) nth = 1;
		memset(&J, 0, sizeof J);
		J.F = F; J.row = row; J.n = n_features;
		if (nth == 1) {
			/* one member, head and features together */
			J.text[0] = head;
			head.s = NULL; head.l = head.m = 0;
			msh_parallel(1, report_worker, &J);
			fd_write_all(fd, J.gz[0].s, J.gz[0].l);
		} else {
			msh_parallel(nth, report_worker, &J);
			gz_member(&head, &hz);
			fd_write_all(fd, hz.s, hz.l);
			for (t = 0; t < nth; t++) fd_write_all(fd, J.gz[t].s, J.gz[t].l);
		}
		for (t = 0; t < nth; t++) { free(J.text[t].s); free(J.gz[t].s); }
		free(hz.s);
		free(head.s);
	}
	if (fd != fileno(stdout) && close(fd) != 0) mDie("Write failed");
}

/* The inserts of one sample counted on several devices of this process (or on this rank of several): everything
 * onto the first context, then mInsertCountToAbundanceMatrix (msam_profile.c:248-425) there.  Leaves g_ctx = ctx[0]. */
static void profile_combine_and_finalize(msx_ctx **ctx, msx_profile **prof, int n_dev, int share_type, double *row,
                                         msx_profile_stats *st) {
	int k;
	g_ctx = ctx[0];
	for (k = 1; k < n_dev; k++)
		if (msx_profile_merge(ctx[0], prof[0], ctx[k], prof[k]) != MSX_OK) mDie("%s", msx_last_error(ctx[0]));
	if (share_type == MSX_MULTI_SHARE_PROPORTIONAL) fprintf(stderr, "# Start PropSharing:\n");
	if (g_dist) {
		/* this rank's counts are a shard's: sum them over the ranks, iterate with the increment all-reduced */
		MSX(msx_profile_finalize_dist_enqueue(g_ctx, prof[0]));
		MSX(msx_profile_fetch(g_ctx, prof[0], row + 1, st));
	} else {
		MSX(msx_profile_finalize(g_ctx, prof[0], row + 1, st));
	}
}

/* the communicator of a rank-per-process run is made BEFORE the input is read: a rank that finishes reading minutes
 * after another would otherwise find its peers' rendezvous timed out */
static void dist_begin(void) {
	if (g_dist) MSX(msx_dist_init_env(g_ctx));
}

/* ---- filter over the pipeline ----------------------------------------------------------------------- */
typedef struct fshared fshared;
typedef struct {
	fshared *S;
	int dev_id, index;
	msx_ctx *ctx;
	msx_profile *prof;          /* filter --profile-out: this device's part of the sample */
	pthread_t th;
	double t_ctx, t_upload, t_gpu, t_fetch, t_wait;
	size_t n_prefetched;
	double t_ctx_end;
	double t_end[64];            /* MSX_TIMING: when the first batches left this stage */
	int n_end;
} fdev_t;

struct fshared {
	pipe_t *P;
	const msx_filter_params *fp;
	int pools, out_mode, argc, n_dev, n_done;
	char **argv;
	msh_out *out;               /* created by the device thread that sees batch 0, after its preflight */
	pthread_mutex_t mu;
	qn_result qn;
	const prof_opts *po;        /* filter --profile-out, else NULL */
	const prof_feat *pf;
	/* one device: its thread finalizes and writes the profile as soon as the last batch is accumulated, beside the
	 * writer's last batches (profile_done); not when a batch held a record the reference dies at (any_fatal) */
	int profile_done, any_fatal;
	int dev_frame, dev_level;   /* -bu / -b: the device hands down finished BGZF blocks (stored / deflated), msx_unpack_emit_gather_bgzf */
	double t_finalized, t_reported;
	fdev_t dev[MSH_MAX_DEVICES];
};

/* preflight on the first records (msam_filter.c:478-482), then the header with its @PG line */
static void filter_open_output(fshared *F, const rbatch *first) {
	pipe_t *P = F->P;
	qn_result qn = {QN_NOT_REQUIRED, 0, 0, 0};
	char qmsg[1024], ds[1300], *cl;
	kstr htext = {0, 0, 0};
	rbatch empty;
	memset(&empty, 0, sizeof empty);
	if (F->pools || F->po) qn = qn_check(P->hdr, first ? first : &empty);     /* (profile checks always: msam_profile.c:708) */
	F->qn = qn;
	if (!F->pools) { qn_result nr = {QN_NOT_REQUIRED, 0, 0, 0}; qn_format(&nr, qmsg, sizeof qmsg); }
	else qn_format(&qn, qmsg, sizeof qmsg);
	cl = command_line(F->argc, F->argv);
	snprintf(ds, sizeof ds, "git=%s; %s", MSH_GIT_COMMIT, qmsg);      /* msam_helper.c:159-164 */
	if (P->hdr->text.l) ks_put(&htext, P->hdr->text.s, P->hdr->text.l);
	msh_hdr_add_pg(&htext, PROGRAM, MSH_VERSION, cl, ds);
	F->out = msh_out_open(stdout, F->out_mode, P->hdr, htext.s);
	free(cl);
	free(htext.s);
}

/* A record the reference's loop dies at (no MD and no NM where statistics are needed: msam_filter.c:150-152; a
 * participating record without AS: :219-221).  The reference has by then written every pool it had completed; the batch
 * API reports the error for the whole batch.  So the batch is filtered once more, cut in front of the pool that holds the
 * offending record, that output goes to the writer, and the writer -- when it reaches this batch, every earlier one
 * written -- dies with the reference's message.  (Not reproduced: a paired pool whose READ1 pass the reference had
 * already written when its READ2 pass met the record without AS.)
 * Returns the number of records in front of the offending pool, *g = the number of pools. */
static int64_t fatal_prefix(const uint32_t *group_off, int64_t n_groups, int64_t err_record, int64_t *g) {
	int64_t lo = 0, hi = n_groups;            /* the last pool that starts at or before err_record */
	while (lo + 1 < hi) {
		const int64_t mid = (lo + hi) >> 1;
		if ((int64_t)group_off[mid] <= err_record) lo = mid; else hi = mid;
	}
	*g = lo;
	return (int64_t)group_off[lo];
}

static void *filter_dev_thread(void *arg) {
	fdev_t *D = (fdev_t *)arg;
	fshared *F = D->S;
	pipe_t *P = F->P;
	msx_stage *stage = NULL;
	msx_unpack *unpack = NULL;
	int pending = PQ_NONE;             /* a slot taken off the queue ahead of its turn (its bytes are being sent up) */
	ahead_q ahead = {{PQ_NONE, PQ_NONE}, 0, 0};
	const int prefetch_on = getenv("MSX_PREFETCH") != NULL;
	{
		double t0 = now_s();
		/* HIP start-up runs beside the decoding of the first batch.  Should it fail, the input's own faults are
		 * reported first (the preflight runs on the writer thread; the reference checks the input before anything else) */
		if (msx_ctx_create(&g_ctx, D->dev_id) != MSX_OK) {
			pthread_mutex_lock(&P->first_mu);
			while (!P->out_opened) pthread_cond_wait(&P->first_cv, &P->first_mu);
			pthread_mutex_unlock(&P->first_mu);
			mDie("%s", msx_last_error(NULL));
		}
		D->ctx = g_ctx;
		MSX(msx_stage_create(g_ctx, &stage));
		if (F->po)
			MSX(msx_profile_create(g_ctx, &D->prof, F->pf->n_features, F->po->share_type, F->pf->fmap, P->hdr->n_targets));
		D->t_ctx = now_s() - t0;
		D->t_ctx_end = now_s();
		if (P->raw_mode) { MSX(msx_unpack_create(g_ctx, &unpack)); pin_start(P, 1); }
	}
	for (;;) {
		double t0 = now_s(), t1;
		const int si = pending != PQ_NONE ? pending : ahead.n ? ahead_pop(&ahead) : pq_pop(&P->q_dev);
		pslot *s;
		rbatch *b;
		msx_batch hb, db;
		msx_filter_out fo;
		msx_filter_status st;
		pending = PQ_NONE;
		t1 = now_s();
		D->t_wait += t1 - t0;
		if (si == PQ_END) break;
		s = &P->slot[si];
		b = &s->b;
		s->fatal = 0;
		if (s->raw) {
			/* the record walk on the device: inflated bytes up, filter's output records back */
			msx_unpack_params up;
			msx_unpack_result ur;
			int64_t nb = 0;
			pin_start(P, 1);
			pin_wait(P, s);
			if (s->has_seed) MSX(msx_unpack_seed(g_ctx, unpack, (const uint8_t *)s->seed.s, s->seed.l, s->seed_has_name ? s->seed_name : NULL));
			memset(&up, 0, sizeof up);
			up.pool_mode = P->mode; up.unmapped_visible = P->unmapped_visible; up.want_aux = 1; up.want_stats = P->want_stats;
			up.n_targets = P->hdr->n_targets; up.last = s->last; up.cut_mapped = P->cut_mapped;
			unpack_slot_enqueue(P, s, unpack, &up);
			if (F->n_dev == 1) unpack_slots_ahead(P, unpack, &ahead);
			unpack_slot_finish(P, s, unpack, &up, &ur, &db);
			if (F->n_dev == 1) unpack_slots_ahead(P, unpack, &ahead);        /* (not decoded yet a moment ago?) */
			/* MSX_PREFETCH=1: the next batch, if it is decoded already, starts its way up now -- behind the bytes this
			 * batch carried over -- and travels while this one is filtered and its output gathered and fetched.  Off by
			 * default: measured, it changes nothing (upload phase 0.35-0.40 s of the 100 M-record run either way).  The
			 * copies are not what that phase spends its time on (rocprofv3: 2.3 ms of upload per 130 MB batch at 56 GB/s,
			 * 48 GB/s each way when both directions are busy -- scripts/micro/pcie_rate.hip -- in a phase of 6.5 ms):
			 * the device thread synchronises five times per batch and has to be scheduled again each time on a host whose
			 * granted CPUs are all busy inflating. */
			if (F->n_dev == 1 && prefetch_on && !P->comp_mode) {
				pending = pq_try_pop(&P->q_dev);
				if (pending >= 0 && P->slot[pending].raw && !P->slot[pending].has_seed && !P->slot[pending].comp) {
					pin_wait(P, &P->slot[pending]);
					MSX(msx_unpack_prefetch(g_ctx, unpack, P->slot[pending].rbuf, P->slot[pending].rlen));
					D->n_prefetched++;
				}
			}
			D->t_upload += now_s() - t1; t1 = now_s();
			b->n = (size_t)ur.n_records;
			s->n_emit = 0;
			s->olen = 0;
			if (ur.n_records > 0) {
				MSX(msx_stage_outputs(g_ctx, stage, ur.n_records, 0, &fo));
				if (D->prof) MSX(msx_filter_profile_enqueue(g_ctx, &db, F->fp, &fo, D->prof));
				else MSX(msx_filter_enqueue(g_ctx, &db, F->fp, &fo));
				if (msx_filter_finish(g_ctx, &st) != MSX_OK) {
					s->fatal = 1;
					F->any_fatal = 1;
					snprintf(s->fatal_msg, sizeof s->fatal_msg, "%s", msx_last_error(g_ctx));
					st.n_emit = 0;
					if (P->mode == 1 && st.err_record >= 0 && ur.n_groups > 0) {
						uint32_t *go = (uint32_t *)xmalloc(((size_t)ur.n_groups + 1) * 4);
						int64_t g = 0, npre;
						MSX(msx_dev_to_host(g_ctx, go, db.group_off, ((size_t)ur.n_groups + 1) * 4));
						npre = fatal_prefix(go, ur.n_groups, st.err_record, &g);
						free(go);
						if (npre > 0) {
							db.n_records = npre; db.n_groups = g;
							MSX(msx_filter_enqueue(g_ctx, &db, F->fp, &fo));
							if (msx_filter_finish(g_ctx, &st) != MSX_OK) st.n_emit = 0;
						}
					}
				}
				D->t_gpu += now_s() - t1; t1 = now_s();
				s->n_emit = st.n_emit;
				if (F->n_dev == 1) unpack_slots_ahead(P, unpack, &ahead);
				/* gather on the device, make room here if this batch keeps more than any before it, and let the bytes travel
				 * while the next batch is worked on: the writer waits for s->ev_out */
				s->framed = F->dev_frame;
				if (F->dev_frame) MSX(msx_unpack_emit_gather_bgzf(g_ctx, unpack, fo.emit_idx, st.n_emit, F->dev_level, &nb, NULL));
				else MSX(msx_unpack_emit_gather(g_ctx, unpack, fo.emit_idx, st.n_emit, &nb));
				if (nb > 0) {
					s->ob = pq_pop(&P->q_ob);                /* (waits for the writer when all of them are on their way out) */
					if ((size_t)nb + 64 > P->ob_cap[s->ob]) {          /* (the old one stays mapped and page-locked: rare) */
						P->ob_cap[s->ob] = (size_t)nb + (size_t)nb / 4 + ((size_t)4 << 20);
						P->ob[s->ob] = io_alloc(P->ob_cap[s->ob]);
						io_populate(P->ob[s->ob], P->ob_cap[s->ob]);
						MSX(msx_host_register(g_ctx, P->ob[s->ob], P->ob_cap[s->ob]));
					}
					s->obuf = P->ob[s->ob];
					s->ocap = P->ob_cap[s->ob];
				}
				if (!s->ev_out) MSX(msx_event_create(g_ctx, &s->ev_out));
				MSX(msx_unpack_emit_fetch(g_ctx, unpack, s->obuf, s->ocap, s->ev_out));
				s->ev_ctx = g_ctx;
				s->olen = (size_t)nb;
				D->t_fetch += now_s() - t1;
			}
			if (D->n_end < 64) D->t_end[D->n_end++] = now_s();
			pq_push(&P->q_out, si);
			continue;
		}
		pipe_pin_slot(P, s);
		rb_host_view(b, &hb, P->mode != 0);
		hb.pool_rule = (F->pools && F->po) ? MSX_POOLS_FILTER : MSX_POOLS_PROFILE;
		MSX(msx_stage_upload(g_ctx, stage, &hb, &db));
		MSX(msx_stage_outputs(g_ctx, stage, (int64_t)b->n, F->fp->rescore, &fo));
		/* (the I/O buffers of the batches behind batch 0 were populated while HIP started up; page-locking them now is a
		 * matter of a millisecond -- it used to be tens, holding the runtime's lock, with this batch's allocations and
		 * copies in line behind it: 77 ms for a 12 MB batch) */
		if (s->seq == 0 && P->raw_mode) pin_start(P, 1);
		if (s->seq == 0 && getenv("MSX_TIMING")) fprintf(stderr, "# batch 0: popped +%.0f ms after the context, uploaded +%.0f\n", (t1 - (t0 - 0)) * 0 + (t1 - D->t_ctx_end) * 1e3, (now_s() - D->t_ctx_end) * 1e3);
		D->t_upload += now_s() - t1; t1 = now_s();
		if (D->prof) MSX(msx_filter_profile_enqueue(g_ctx, &db, F->fp, &fo, D->prof));
		else MSX(msx_filter_enqueue(g_ctx, &db, F->fp, &fo));
		if (msx_filter_finish(g_ctx, &st) != MSX_OK) {          /* the reference's own mDie texts; see fatal_prefix */
			s->fatal = 1;
			F->any_fatal = 1;
			snprintf(s->fatal_msg, sizeof s->fatal_msg, "%s", msx_last_error(g_ctx));
			st.n_emit = 0;
			if (P->mode == 1 && st.err_record >= 0 && b->n_groups > 0) {
				int64_t g = 0;
				const int64_t npre = fatal_prefix(b->group_off, (int64_t)b->n_groups, st.err_record, &g);
				if (npre > 0) {
					db.n_records = npre; db.n_groups = g;
					MSX(msx_filter_enqueue(g_ctx, &db, F->fp, &fo));
					if (msx_filter_finish(g_ctx, &st) != MSX_OK) st.n_emit = 0;
				}
			}
		}
		if (s->seq == 0 && getenv("MSX_TIMING")) fprintf(stderr, "# batch 0: kernels done +%.0f ms after the context\n", (now_s() - D->t_ctx_end) * 1e3);
		D->t_gpu += now_s() - t1; t1 = now_s();
		s->n_emit = st.n_emit;
		MSX(msx_dev_to_host(g_ctx, s->emit, fo.emit_idx, 4 * (size_t)st.n_emit));
		if (F->fp->rescore) MSX(msx_dev_to_host(g_ctx, s->as_out, fo.as_out, 4 * b->n));
		D->t_fetch += now_s() - t1;
		if (D->n_end < 64) D->t_end[D->n_end++] = now_s();
		pq_push(&P->q_out, si);
	}
	MSX(msx_ctx_sync(g_ctx));
	/* the last device thread to finish closes the writer's queue (and opens the output of an empty input) */
	pthread_mutex_lock(&F->mu);
	if (++F->n_done == F->n_dev) pq_push(&P->q_out, PQ_END);
	pthread_mutex_unlock(&F->mu);
	if (F->po && F->n_dev == 1 && !F->any_fatal && D->prof && __atomic_load_n(&P->n_filled, __ATOMIC_ACQUIRE) > 0) {
		msx_ctx *ctxs[1] = {g_ctx};
		msx_profile *profs[1] = {D->prof};
		msx_profile_stats pst;
		double *row = (double *)calloc((size_t)F->pf->n_features + 1, sizeof(double));
		char *cl = command_line(F->argc, F->argv);
		/* (the preflight's verdict is part of the report: the writer thread has it once the output is open) */
		pthread_mutex_lock(&P->first_mu);
		while (!P->out_opened) pthread_cond_wait(&P->first_cv, &P->first_mu);
		pthread_mutex_unlock(&P->first_mu);
		profile_combine_and_finalize(ctxs, profs, 1, F->po->share_type, row, &pst);
		F->t_finalized = now_s();
		profile_report(F->po, F->pf, &pst, row, &F->qn, cl);
		F->t_reported = now_s();
		free(row);
		free(cl);
		F->profile_done = 1;
	}
	pin_join(P);
	msx_stage_destroy(g_ctx, stage);
	msx_unpack_destroy(g_ctx, unpack);
	return NULL;
}

/* the writer's side of q_out: the slot that holds batch `seq` (device threads finish in any order), or PQ_END */
static int pq_pop_seq(pq *q, const pipe_t *P, size_t seq) {
	int i, v = PQ_END - 1;
	pthread_mutex_lock(&q->mu);
	for (;;) {
		int end = 0;
		for (i = 0; i < q->n; i++) {
			if (q->item[i] == PQ_END) { end = 1; continue; }
			if (P->slot[q->item[i]].seq == seq) break;
		}
		if (i < q->n) {
			v = q->item[i];
			for (i = i + 1; i < q->n; i++) q->item[i - 1] = q->item[i];
			q->n--;
			break;
		}
		if (end) { v = PQ_END; break; }       /* (pushed after every batch: nothing more can arrive) */
		pthread_cond_wait(&q->cv, &q->mu);
	}
	pthread_mutex_unlock(&q->mu);
	return v;
}

/* --rescore: the emitted records of a batch rewritten (first AS dropped, AS:i appended: msam_filter.c:160-168),
 * in parallel into one buffer the block writer then reads from */
typedef struct {
	const pslot *s;
	int pass;
	size_t *off;             /* [n_emit + 1] offsets into blob (each record with its 4-byte length) */
	uint8_t *blob;
} rescore_job;

static void rescore_worker(void *arg, int tid, int nth) {
	rescore_job *J = (rescore_job *)arg;
	const pslot *s = J->s;
	const rbatch *b = &s->b;
	const size_t n = (size_t)s->n_emit, lo = n * (size_t)tid / (size_t)nth, hi = n * (size_t)(tid + 1) / (size_t)nth;
	size_t i;
	kstr tmp = {0, 0, 0};
	for (i = lo; i < hi; i++) {
		const size_t k = (size_t)s->emit[i];
		const uint8_t *r = RB_REC(b, k);
		const size_t len = RB_LEN(b, k);
		const int mapped = !(b->flag[k] & 4);
		if (J->pass == 0) {
			size_t out_len = len;
			if (mapped) {
				const uint8_t *as = msh_aux_get(r, len, "AS");
				out_len = len - (as ? 2 + msh_aux_size(as, r + len) : 0) + 7;
			}
			J->off[i] = 4 + out_len;
		} else {
			uint8_t *o = J->blob + J->off[i];
			const uint8_t *src = r;
			size_t l = len;
			if (mapped) { rescore_record(r, len, s->as_out[k], &tmp); src = (const uint8_t *)tmp.s; l = tmp.l; }
			o[0] = (uint8_t)l; o[1] = (uint8_t)(l >> 8); o[2] = (uint8_t)(l >> 16); o[3] = (uint8_t)(l >> 24);
			memcpy(o + 4, src, l);
		}
	}
	free(tmp.s);
}

static int filter_pipelined(msh_in *in, const msx_filter_params *fp, int pools, int want_stats, int out_mode, int argc,
                            char *argv[], const prof_opts *po) {
	static pipe_t P;
	static fshared F;
	prof_feat pf;
	pthread_t th_dec;
	double t_start = now_s(), tw = 0, t_wait = 0, t_tail[4] = {0, 0, 0, 0};
	size_t n_in = 0, n_out = 0, n_batches = 0, seq = 0;
	int dev_ids[MSH_MAX_DEVICES], k;
	rescore_job RJ;
	int32_t *ident = NULL;
	size_t ident_cap = 0;
	const int unmapped_written = fp->keep_unmapped && fp->ppt >= 0 && fp->invert &&
	                             ((fp->min_length > 0) || fp->ppt != 0 || fp->max_clip < 100);   /* msam_filter.c:132-138 */
	memset(&F, 0, sizeof F);
	memset(&RJ, 0, sizeof RJ);
	memset(&pf, 0, sizeof pf);
	F.n_dev = device_list(dev_ids);
	/* pools: filter's own rule when best-hit selection needs them; with --profile-out and no best hit, profile's rule
	 * over the records filter can write (filter's output does not depend on pools then) */
	pipe_init(&P, in, pools ? 1 : (po ? 3 : 0), want_stats, F.n_dev);
	P.unmapped_visible = unmapped_written;
	P.cut_mapped = pools && po;
	/* From the second batch on the record walk runs on the device (msx_unpack): one context, records written as they
	 * are (no --rescore), BAM out.  The first batch takes the host-side walk: the preflight reads its records.
	 * MSX_HOST_UNPACK=1 keeps every batch on the host. */
	if (F.n_dev == 1 && !fp->rescore && (out_mode == MSH_OUT_BAM || out_mode == MSH_OUT_UBAM) && !getenv("MSX_HOST_UNPACK"))
		pipe_enable_raw(&P, 1);
	if (fp->rescore)
		for (k = 0; k < P.n_slots; k++) P.slot[k].as_out = (int32_t *)xmalloc((P.cap_rec + 8) * 4);
	if (po) prof_features(po, P.hdr, &pf);
	/* the BGZF layer of the output on the device: stored blocks for -bu, DEFLATE for -b (MSX_HOST_FRAME=1: frame / zlib-deflate
	 * on the host cores, as round 3 did; MSX_HOST_DEFLATE=1: only -b's deflate) */
	F.dev_frame = (out_mode == MSH_OUT_UBAM || (out_mode == MSH_OUT_BAM && !getenv("MSX_HOST_DEFLATE"))) && !getenv("MSX_HOST_FRAME");
	F.dev_level = out_mode == MSH_OUT_UBAM ? 0 : 6;
	F.P = &P; F.fp = fp; F.pools = pools; F.out_mode = out_mode; F.argc = argc; F.argv = argv; F.po = po; F.pf = &pf;
	pthread_mutex_init(&F.mu, NULL);
	if (pthread_create(&th_dec, NULL, pipe_decode_thread, &P) != 0) mDie("pthread_create failed");
	for (k = 0; k < F.n_dev; k++) {
		F.dev[k].S = &F; F.dev[k].dev_id = dev_ids[k]; F.dev[k].index = k;
		if (pthread_create(&F.dev[k].th, NULL, filter_dev_thread, &F.dev[k]) != 0) mDie("pthread_create failed");
	}
	{
		/* the preflight on batch 0's records (msam_filter.c:478-482) and the header, here -- beside the device
		 * stage's start-up and its work on batch 0, not in front of it (a million @SQ lines are 45 MB of header) */
		pthread_mutex_lock(&P.first_mu);
		while (P.first_state == 0) pthread_cond_wait(&P.first_cv, &P.first_mu);
		pthread_mutex_unlock(&P.first_mu);
		filter_open_output(&F, P.first_state == 1 ? &P.slot[P.first_slot].b : NULL);
		pthread_mutex_lock(&P.first_mu);
		P.out_opened = 1;
		pthread_cond_broadcast(&P.first_cv);
		pthread_mutex_unlock(&P.first_mu);
	}
	for (;;) {                                   /* this thread is the encode stage: batches in input order */
		double t0 = now_s(), t1;
		const int si = pq_pop_seq(&P.q_out, &P, seq);
		pslot *s;
		t1 = now_s();
		t_wait += t1 - t0;
		if (si == PQ_END) break;
		s = &P.slot[si];
		if (s->raw) {
			if (s->ev_out && s->olen) { if (msx_event_wait(s->ev_ctx, s->ev_out) != MSX_OK) mDie("%s", msx_last_error(s->ev_ctx)); }
			if (s->framed) msh_write_framed(F.out, s->obuf, s->olen);
			else msh_write_stream(F.out, s->obuf, s->olen);
			if (s->ob >= 0) { pq_push(&P.q_ob, s->ob); s->ob = -1; }
		} else if (!fp->rescore) {
			msh_write_many(F.out, s->b.base, s->b.rec_off, s->emit, (size_t)s->n_emit);
		} else if (s->n_emit > 0) {
			const size_t n = (size_t)s->n_emit;
			size_t i, tot = 0;
			RJ.s = s;
			RJ.off = (size_t *)realloc(RJ.off, (n + 1) * sizeof(size_t));
			if (n > ident_cap) {
				ident_cap = n + n / 4 + 1024;
				ident = (int32_t *)realloc(ident, ident_cap * 4);
				if (!ident) mDie("Out of memory");
				for (i = 0; i < ident_cap; i++) ident[i] = (int32_t)i;
			}
			RJ.pass = 0;
			msh_parallel(msh_threads(), rescore_worker, &RJ);
			for (i = 0; i < n; i++) { const size_t z = RJ.off[i]; RJ.off[i] = tot; tot += z; }
			RJ.off[n] = tot;
			RJ.blob = (uint8_t *)realloc(RJ.blob, tot + 16);
			if (!RJ.off || !RJ.blob) mDie("Out of memory");
			RJ.pass = 1;
			msh_parallel(msh_threads(), rescore_worker, &RJ);
			msh_write_many(F.out, RJ.blob, RJ.off, ident, n);
		}
		if (s->fatal) {                      /* every earlier batch and the pools in front of the record are written */
			msh_out_drain(F.out);
			mDie("%s", s->fatal_msg);
		}
		n_batches++;
		seq++;
		n_in += s->b.n;
		n_out += (size_t)s->n_emit;
		tw += now_s() - t1;
		pq_push(&P.q_free, si);
	}
	t_tail[0] = now_s();
	pthread_join(th_dec, NULL);
	for (k = 0; k < F.n_dev; k++) pthread_join(F.dev[k].th, NULL);
	msh_out_close(F.out);
	t_tail[1] = now_s();
	if (po && F.profile_done) {
		t_tail[2] = F.t_finalized; t_tail[3] = F.t_reported;
		g_ctx = F.dev[0].ctx;
	} else if (po) {
		/* the other half of `filter ... | profile -`, without the pipe, the second decode and the second process */
		msx_ctx *ctxs[MSH_MAX_DEVICES];
		msx_profile *profs[MSH_MAX_DEVICES];
		msx_profile_stats st;
		double *row = (double *)calloc((size_t)pf.n_features + 1, sizeof(double));
		char *cl = command_line(argc, argv);
		for (k = 0; k < F.n_dev; k++) { ctxs[k] = F.dev[k].ctx; profs[k] = F.dev[k].prof; }
		profile_combine_and_finalize(ctxs, profs, F.n_dev, po->share_type, row, &st);
		t_tail[2] = now_s();
		profile_report(po, &pf, &st, row, &F.qn, cl);
		t_tail[3] = now_s();
		free(row);
		free(cl);
	} else {
		g_ctx = F.dev[0].ctx;
	}
	if (getenv("MSX_TIMING")) {
		double t_ctx = 0, t_up = 0, t_gpu = 0, t_fetch = 0, t_dw = 0;
		for (k = 0; k < F.n_dev; k++) {
			t_ctx += F.dev[k].t_ctx; t_up += F.dev[k].t_upload; t_gpu += F.dev[k].t_gpu; t_fetch += F.dev[k].t_fetch; t_dw += F.dev[k].t_wait;
		}
		fprintf(stderr, "# batches: %zu (%zu sent ahead)%s\n", n_batches, P.comp_mode ? P.n_ahead : F.dev[0].n_prefetched,
		        P.comp_mode ? "; BGZF blocks inflated on the device" : "");
		if (P.n_host_inflated) fprintf(stderr, "# %zu batches inflated on the host (blocks the device refused)%s\n", P.n_host_inflated,
		                               P.comp_given_up ? "; the device was not asked any more after that" : "");
		if (F.dev[0].n_end) {
			int q;
			fprintf(stderr, "# device stage, batches done at (ms):");
			for (q = 0; q < F.dev[0].n_end; q++) fprintf(stderr, " %.0f", (F.dev[0].t_end[q] - t_start) * 1e3);
			fprintf(stderr, "; writer done %.0f, output closed %.0f, profile finalized %.0f, written %.0f\n", (t_tail[0] - t_start) * 1e3,
			        (t_tail[1] - t_start) * 1e3, (t_tail[2] - t_start) * 1e3, (t_tail[3] - t_start) * 1e3);
		}
		fprintf(stderr, "# decode stage: inflate %.3f, record chase %.3f, aux scan %.3f, offsets+pools (serial) %.3f, payload copy %.3f s\n",
		        P.t_inflate, P.t_chase, P.t_scan, P.t_serial, P.t_copy);
		fprintf(stderr, "# filter pipeline: wall %.3f s; decode %.3f s (+%.3f waiting for a free slot); device: start-up %.3f, "
		        "upload %.3f, kernels %.3f, fetch %.3f (+%.3f waiting for a batch); encode+write %.3f s (+%.3f waiting); "
		        "%zu records in, %zu out, %d threads, %d device%s\n",
		        now_s() - t_start, P.t_decode, P.t_wait_free, t_ctx, t_up, t_gpu, t_fetch, t_dw, tw, t_wait,
		        n_in, n_out, msh_threads(), F.n_dev, F.n_dev > 1 ? "s" : "");
	}
	fast_exit();
	for (k = 0; k < F.n_dev; k++) {
		if (F.dev[k].prof) msx_profile_destroy(F.dev[k].ctx, F.dev[k].prof);
		msx_ctx_destroy(F.dev[k].ctx);
	}
	return 0;
}

int msam_filter_main(int argc, char *argv[]) {
	static const struct option lopts[] = {
	    {"help", no_argument, 0, 1000},       {"ppt", required_argument, 0, 1001},
	    {"rescore", no_argument, 0, 1002},    {"besthit", no_argument, 0, 1003},
	    {"uniqhit", no_argument, 0, 1004},    {"keep_unmapped", no_argument, 0, 'k'},
	    {"invert", no_argument, 0, 'v'},
	    /* `filter ... | profile -` in one process (additive; the reference's surface is unchanged): profile's options */
	    {"profile-out", required_argument, 0, 1100}, {"label", required_argument, 0, 1101},
	    {"genome", required_argument, 0, 1102},      {"total", required_argument, 0, 1103},
	    {"mincount", required_argument, 0, 1104},    {"unit", required_argument, 0, 1105},
	    {"pandas", no_argument, 0, 1106},            {"no-pandas", no_argument, 0, 1107},
	    {"nolen", no_argument, 0, 1108},             {"multi", required_argument, 0, 1109},
	    {0, 0, 0, 0}};
	prof_opts po;
	int tee;
	int o_b = 0, o_u = 0, o_h = 0, o_S = 0, o_help = 0, o_k = 0, o_v = 0, o_rescore = 0, o_best = 0, o_uniq = 0;
	int n_l = 0, n_p = 0, n_ppt = 0, n_z = 0, nerrors = 0, c;
	long v_l = 0, v_p = 0, v_ppt = 0, v_z = 0;
	msx_filter_params fp;
	const char *infile;
	char *cl;
	reader rd;
	rbatch b;
	qn_result qn = {QN_NOT_REQUIRED, 0, 0, 0};
	char qmsg[1024], ds[1300];
	kstr htext = {0, 0, 0}, tmp = {0, 0, 0};
	const msh_hdr *hdr;
	msh_out *out;
	int mode, pools, want_stats, choice, bulk;
	size_t target = batch_target();
	int32_t *emit = NULL, *as_out = NULL;
	size_t emit_cap = 0;

	(void)o_S;
	memset(&po, 0, sizeof po);
	opterr = 0;
	optind = 1;
	while ((c = getopt_long(argc, argv, "buhSkvl:p:z:", lopts, NULL)) != -1) {
		switch (c) {
		case 'b': o_b++; break;
		case 'u': o_u++; break;
		case 'h': o_h++; break;
		case 'S': o_S++; break;
		case 'k': o_k++; break;
		case 'v': o_v++; break;
		case 'l': n_l++; v_l = strtol(optarg, NULL, 10); break;
		case 'p': n_p++; v_p = strtol(optarg, NULL, 10); break;
		case 'z': n_z++; v_z = strtol(optarg, NULL, 10); break;
		case 1000: o_help++; break;
		case 1001: n_ppt++; v_ppt = strtol(optarg, NULL, 10); break;
		case 1002: o_rescore++; break;
		case 1003: o_best++; break;
		case 1004: o_uniq++; break;
		case 1100: po.n_out++; po.out = optarg; break;
		case 1101: po.n_label++; po.label = optarg; break;
		case 1102: po.genome = optarg; break;
		case 1103: po.n_total++; po.v_total = strtol(optarg, NULL, 10); break;
		case 1104: po.n_mincount++; po.v_mincount = strtol(optarg, NULL, 10); break;
		case 1105: po.unit = optarg; break;
		case 1106: po.pandas++; break;
		case 1107: po.nopandas++; break;
		case 1108: po.nolen++; break;
		case 1109: po.multi = optarg; break;
		default:
			fprintf(stderr, "%s: invalid option \"%s\"\n", PROGRAM, argv[optind - 1]);
			nerrors++;
		}
	}
	if (o_help > 0 || argc < 2) {                                     /* msam_filter.c:383-386 */
		filter_help(stdout);
		exit(EXIT_SUCCESS);
	}
	if (argc - optind < 1) { fprintf(stderr, "%s: missing option <bamfile>\n", PROGRAM); nerrors++; }
	if (nerrors > 0) {                                                /* :389-393 (stderr) */
		fprintf(stderr, "Use --help for usage instructions!\n");
		mQuit("");
	}
	if (argc - optind > 1) {                                          /* mMultipleFileError */
		fprintf(stderr, "Multiple input files not supported in filter.\n");
		fprintf(stderr, "Use 'samtools merge' to combine BAM/SAM files.\n");
		filter_help(stdout);
		mQuit("");
	}
#define BAIL(msg) do { fprintf(stdout, "%s\n", msg); filter_help(stdout); mQuit(""); } while (0)
	if (o_v > 0 && (o_best > 0 || o_uniq > 0)) BAIL("--invert cannot be combined with --besthit or --uniqhit");   /* :398-418 */
	else if (o_best > 0 && o_uniq > 0) BAIL("--besthit cannot be combined with --uniqhit");
	else if (n_p > 0 && n_ppt > 0) BAIL("-p cannot be combined with --ppt");
	else if (!n_l && !n_p && !n_ppt && !o_uniq && !o_best && !n_z)
		BAIL("--mode filter needs -l, -p, --ppt, -z, --besthit or --uniqhit");
	tee = po.n_out > 0;
	if (!tee && (po.n_label || po.genome || po.n_total || po.n_mincount || po.unit || po.pandas || po.nopandas || po.nolen || po.multi))
		BAIL("--label, --genome, --total, --mincount, --unit, --pandas, --no-pandas, --nolen and --multi need --profile-out");
	if (tee) {                                                        /* msam_profile.c:672-700 */
		if (po.n_label != 1 || po.n_out != 1) BAIL("--profile-out requires --label");
		if (strcmp(po.out, "-") == 0) BAIL("--profile-out cannot be '-': standard output carries the alignments");
		if (po.pandas > 0 && po.nopandas > 0) BAIL("--pandas and --no-pandas cannot be used together");
		if (po.n_total > 0 && po.v_total <= 0) BAIL("--total must be a positive integer");
		if (po.n_mincount > 0 && po.v_mincount < 0) BAIL("--mincount must be a non-negative integer");
	}
	memset(&fp, 0, sizeof fp);
	if (n_p > 0) {                                                    /* :420-457 */
		if (v_p < 0 || v_p > 100) BAIL("-p must be in the range [0,100]");
		fp.ppt = (int32_t)(10 * v_p);
	} else if (n_ppt > 0) {
		fp.ppt = (int32_t)v_ppt;
		if (fp.ppt < -1000 || fp.ppt > 1000) BAIL("--ppt must be in the range [-1000,1000]");
	}
	fp.max_clip = 100;
	if (n_z > 0) {
		fp.max_clip = (int32_t)(100 - v_z);
		if (fp.max_clip < 0 || fp.max_clip > 100) BAIL("-z must be in the range [0,100]");
	}
	if (n_l > 0) {
		fp.min_length = (int32_t)v_l;
		if (fp.min_length < 0) BAIL("-l must be a non-negative integer");
	}
#undef BAIL
	fp.rescore = o_rescore > 0;
	fp.invert = o_v > 0;
	fp.keep_unmapped = o_k > 0;
	fp.besthit = o_best > 0;
	fp.uniqhit = o_uniq > 0;
	mode = o_u ? MSH_OUT_UBAM : o_b ? MSH_OUT_BAM : o_h ? MSH_OUT_SAM_HDR : MSH_OUT_SAM;   /* :464-470 */
	pools = fp.besthit || fp.uniqhit;
	choice = (fp.min_length > 0) | (fp.ppt != 0) << 1 | (fp.max_clip < 100) << 2;
	want_stats = choice != 0 || fp.rescore;

	infile = argv[optind];
	memset(&rd, 0, sizeof rd);
	memset(&b, 0, sizeof b);
	rd.in = msh_open(infile);
	hdr = msh_header(rd.in);

	if (tee) prof_opts_derive(&po);
	bulk = msh_is_bam(rd.in);
	if (!getenv("MSX_SERIAL_IO")) {
		/* BAM or SAM text in: decode | device | encode as three overlapping stages */
		int rc = filter_pipelined(rd.in, &fp, pools, want_stats, mode, argc, argv, tee ? &po : NULL);
		msh_close(rd.in);
		return rc;
	}
	if (tee) mDie("--profile-out is not available with MSX_SERIAL_IO");
	/* MSX_SERIAL_IO (tests: the record-at-a-time reader as a second opinion): one batch at a time */
	/* first batch: large enough for the preflight window */
	{
		size_t t1 = target > COORD_ORDER_CHECK_RECORDS ? target : COORD_ORDER_CHECK_RECORDS;
		TIC;
		if (bulk) fill_batch_bulk(&rd, &b, t1, pools ? 1 : 0, want_stats);
		else fill_filter_batch(&rd, &b, t1, pools, want_stats);
		TOC(t_decode);
	}
	if (pools) qn = qn_check(hdr, &b);                                /* :478-482 */
	ctx_open();
	qn_format(&qn, qmsg, sizeof qmsg);
	cl = command_line(argc, argv);
	snprintf(ds, sizeof ds, "git=%s; %s", MSH_GIT_COMMIT, qmsg);      /* msam_helper.c:159-164 */
	if (hdr->text.l) ks_put(&htext, hdr->text.s, hdr->text.l);
	msh_hdr_add_pg(&htext, PROGRAM, MSH_VERSION, cl, ds);
	out = msh_out_open(stdout, mode, hdr, htext.s);

	for (;;) {
		if (b.n > 0) {
			msx_batch hb, db;
			msx_filter_out fo;
			msx_filter_status st;
			void *d_keep, *d_emit, *d_as = NULL;
			size_t i;
			int rc;
			TIC;
			rb_host_view(&b, &hb, pools);
			MSX(msx_batch_upload(g_ctx, &hb, &db));
			TOC(t_upload);
			MSX(msx_dev_alloc(g_ctx, &d_keep, b.n));
			MSX(msx_dev_alloc(g_ctx, &d_emit, 4 * b.n));
			if (fp.rescore) MSX(msx_dev_alloc(g_ctx, &d_as, 4 * b.n));
			fo.keep = (uint8_t *)d_keep; fo.emit_idx = (int32_t *)d_emit; fo.as_out = (int32_t *)d_as;
			MSX(msx_filter_enqueue(g_ctx, &db, &fp, &fo));
			rc = msx_filter_finish(g_ctx, &st);
			if (rc != MSX_OK) mDie("%s", msx_last_error(g_ctx));      /* the reference's own mDie texts */
			TOC(t_gpu);
			if ((size_t)st.n_emit > emit_cap || !emit) {
				emit_cap = (size_t)st.n_emit + 1024;
				emit = (int32_t *)realloc(emit, emit_cap * 4);
			}
			MSX(msx_dev_to_host(g_ctx, emit, d_emit, 4 * (size_t)st.n_emit));
			if (fp.rescore) {
				as_out = (int32_t *)realloc(as_out, 4 * b.n);
				MSX(msx_dev_to_host(g_ctx, as_out, d_as, 4 * b.n));
			}
			TOC(t_fetch);
			if (!fp.rescore) {
				msh_write_many(out, b.base, b.rec_off, emit, (size_t)st.n_emit);
				st.n_emit = 0;          /* nothing left for the per-record loop */
			}
			for (i = 0; i < (size_t)st.n_emit; i++) {
				size_t k = (size_t)emit[i];
				const uint8_t *r = RB_REC(&b, k);
				size_t len = RB_LEN(&b, k);
				if (fp.rescore && !(b.flag[k] & 4)) {
					rescore_record(r, len, as_out[k], &tmp);
					msh_write(out, (const uint8_t *)tmp.s, tmp.l);
				} else {
					msh_write(out, r, len);
				}
			}
			TOC(t_write);
			msx_dev_free(g_ctx, d_keep);
			msx_dev_free(g_ctx, d_emit);
			msx_dev_free(g_ctx, d_as);
			msx_batch_free(g_ctx, &db);
		}
		if (bulk ? rd.done : (rd.eof && !rd.have_pending)) break;
		{
			TIC;
			if (bulk) fill_batch_bulk(&rd, &b, target, pools ? 1 : 0, want_stats);
			else fill_filter_batch(&rd, &b, target, pools, want_stats);
			TOC(t_decode);
		}
	}
	msh_out_close(out);
	if (getenv("MSX_TIMING"))
		fprintf(stderr, "# filter stages: decode+pack %.3f s, upload %.3f s, gpu %.3f s, fetch %.3f s, write %.3f s\n",
		        t_decode, t_upload, t_gpu, t_fetch, t_write);
	msh_close(rd.in);
	msx_ctx_destroy(g_ctx);
	free(cl);
	return 0;
}

/* ---- profile over the pipeline: one device thread per GPU ---------------------------------------------------- */
typedef struct pshared pshared;
typedef struct {
	pshared *S;
	int dev_id;
	msx_ctx *ctx;
	msx_profile *prof;
	pthread_t th;
	double t_ctx, t_dev, t_wait;
	size_t n_in, n_batches;
} pdev_t;

struct pshared {
	pipe_t *P;
	const prof_opts *o;
	const prof_feat *F;
	qn_result qn;
	int n_dev;
	pdev_t dev[MSH_MAX_DEVICES];
};

static void *profile_dev_thread(void *arg) {
	pdev_t *D = (pdev_t *)arg;
	pshared *S = D->S;
	pipe_t *P = S->P;
	msx_stage *stage = NULL;
	msx_unpack *unpack = NULL;
	msx_event *ev[PIPE_SLOTS_MAX] = {NULL};
	int held = -1, q;                          /* held: slot whose uploads may still be in flight */
	ahead_q ahead = {{PQ_NONE, PQ_NONE}, 0, 0};  /* slots taken off the queue ahead of their turn */
	double t0 = now_s();
	ctx_open_dev(D->dev_id);                     /* HIP start-up runs beside the decoding of the first batch */
	D->ctx = g_ctx;
	if (D == &S->dev[0]) dist_begin();
	MSX(msx_stage_create(g_ctx, &stage));
	MSX(msx_profile_create(g_ctx, &D->prof, S->F->n_features, S->o->share_type, S->F->fmap, P->hdr->n_targets));   /* :855 */
	D->t_ctx = now_s() - t0;
	if (P->raw_mode) { MSX(msx_unpack_create(g_ctx, &unpack)); pin_start(P, 0); }
	for (;;) {
		double t1;
		int si;
		pslot *s;
		msx_batch hb, db;
		t0 = now_s();
		si = ahead.n ? ahead_pop(&ahead) : pq_pop(&P->q_dev);
		t1 = now_s();
		D->t_wait += t1 - t0;
		if (si == PQ_END) break;
		s = &P->slot[si];
		if (s->raw) {
			msx_unpack_params up;
			msx_unpack_result ur;
			if (held >= 0) { MSX(msx_event_wait(g_ctx, ev[held])); pq_push(&P->q_free, held); held = -1; }
			pin_start(P, 0);
			pin_wait(P, s);
			if (s->has_seed) MSX(msx_unpack_seed(g_ctx, unpack, (const uint8_t *)s->seed.s, s->seed.l, s->seed_has_name ? s->seed_name : NULL));
			memset(&up, 0, sizeof up);
			up.pool_mode = 2; up.n_targets = P->hdr->n_targets; up.last = s->last;
			unpack_slot_enqueue(P, s, unpack, &up);
			if (S->n_dev == 1) unpack_slots_ahead(P, unpack, &ahead);
			unpack_slot_finish(P, s, unpack, &up, &ur, &db);              /* (synchronises: the slot's bytes have left) */
			if (ur.n_records > 0) MSX(msx_profile_accumulate(g_ctx, D->prof, &db, NULL));
			D->n_in += (size_t)ur.n_records;
			D->n_batches++;
			pq_push(&P->q_free, si);
			D->t_dev += now_s() - t1;
			continue;
		}
		if (s->seq == 0 && P->raw_mode && !__atomic_load_n(&P->in_eof, __ATOMIC_RELAXED)) pin_start(P, 0);
		if (s->seq == 0) S->qn = qn_check(P->hdr, &s->b);            /* :708, always for profile */
		pipe_pin_slot(P, s);
		rb_host_view(&s->b, &hb, 1);
		hb.cigar_off = NULL; hb.cigar = NULL; hb.md_off = NULL; hb.md = NULL;   /* profile reads tid only */
		hb.nm = NULL; hb.as = NULL; hb.pos = NULL; hb.flag = NULL; hb.rflags = NULL;
		MSX(msx_stage_upload(g_ctx, stage, &hb, &db));
		/* the slot's page-locked arrays go back to the decoder once these copies have left -- a marker per slot,
		 * waited for one batch later, instead of a stream synchronisation per batch */
		if (!ev[si]) MSX(msx_event_create(g_ctx, &ev[si]));
		MSX(msx_event_record(g_ctx, ev[si]));
		MSX(msx_profile_accumulate(g_ctx, D->prof, &db, NULL));
		D->n_in += s->b.n;
		D->n_batches++;
		if (held >= 0) { MSX(msx_event_wait(g_ctx, ev[held])); pq_push(&P->q_free, held); }
		held = si;
		D->t_dev += now_s() - t1;
	}
	if (held >= 0) { MSX(msx_event_wait(g_ctx, ev[held])); pq_push(&P->q_free, held); }
	MSX(msx_ctx_sync(g_ctx));
	pin_join(P);
	msx_stage_destroy(g_ctx, stage);
	msx_unpack_destroy(g_ctx, unpack);
	for (q = 0; q < PIPE_SLOTS_MAX; q++) msx_event_destroy(g_ctx, ev[q]);
	return NULL;
}

int msam_profile_main(int argc, char *argv[]) {
	static const struct option lopts[] = {
	    {"help", no_argument, 0, 1000},        {"label", required_argument, 0, 1001},
	    {"genome", required_argument, 0, 1002},{"total", required_argument, 0, 1003},
	    {"mincount", required_argument, 0, 1004},{"unit", required_argument, 0, 1005},
	    {"pandas", no_argument, 0, 1006},      {"no-pandas", no_argument, 0, 1007},
	    {"nolen", no_argument, 0, 1008},       {"multi", required_argument, 0, 1009},
	    {0, 0, 0, 0}};
	prof_opts o;
	prof_feat F;
	int o_help = 0, nerrors = 0, c;
	msh_in *in;
	const msh_hdr *hdr;
	rbatch b;
	qn_result qn;
	msx_profile *prof = NULL;
	msx_profile_stats st;
	double *row;
	kstr rec = {0, 0, 0};
	char prev_read[256], *cl;
	int have_prev = 0, eof = 0, have_pending = 0, first = 1;
	size_t target = batch_target();
	static reader prof_rd;

	memset(&o, 0, sizeof o);
	opterr = 0;
	optind = 1;
	while ((c = getopt_long(argc, argv, "So:", lopts, NULL)) != -1) {
		switch (c) {
		case 'S': break;
		case 'o': o.n_out++; o.out = optarg; break;
		case 1000: o_help++; break;
		case 1001: o.n_label++; o.label = optarg; break;
		case 1002: o.genome = optarg; break;
		case 1003: o.n_total++; o.v_total = strtol(optarg, NULL, 10); break;
		case 1004: o.n_mincount++; o.v_mincount = strtol(optarg, NULL, 10); break;
		case 1005: o.unit = optarg; break;
		case 1006: o.pandas++; break;
		case 1007: o.nopandas++; break;
		case 1008: o.nolen++; break;
		case 1009: o.multi = optarg; break;
		default:
			fprintf(stdout, "%s: invalid option \"%s\"\n", PROGRAM, argv[optind - 1]);
			nerrors++;
		}
	}
	if (o_help > 0 || argc < 2) { profile_help(stdout); exit(EXIT_SUCCESS); }
	if (argc - optind < 1) { fprintf(stdout, "%s: missing option <bamfile>\n", PROGRAM); nerrors++; }
	if (o.n_out == 0) { fprintf(stdout, "%s: missing option -o <file>\n", PROGRAM); nerrors++; }
	if (o.n_label == 0) { fprintf(stdout, "%s: missing option --label=<string>\n", PROGRAM); nerrors++; }
	if (nerrors > 0) {                                                /* msam_profile.c:664-668 (stdout) */
		fprintf(stdout, "Use --help for usage instructions!\n");
		mQuit("");
	}
	if (argc - optind > 1) {
		fprintf(stderr, "Multiple input files not supported in profile.\n");
		fprintf(stderr, "Use 'samtools merge' to combine BAM/SAM files.\n");
		profile_help(stdout);
		mQuit("");
	}
#define BAIL(msg) do { fprintf(stdout, "%s\n", msg); profile_help(stdout); mQuit(""); } while (0)
	if (o.n_label != 1 || o.n_out != 1) BAIL("requires --label and -o");
	if (o.pandas > 0 && o.nopandas > 0) BAIL("--pandas and --no-pandas cannot be used together");
	if (o.n_total > 0 && (int)o.v_total <= 0) BAIL("--total must be a positive integer");
	if (o.n_mincount > 0 && o.v_mincount < 0) BAIL("--mincount must be a non-negative integer");
#undef BAIL

	{
		/* One rank of several (one process per GPU, RANK / WORLD_SIZE / MASTER_* in the environment): asked for with
		 * "{rank}" in the input path -- replaced by the rank: every rank reads ITS shard -- or MSX_DIST=1 (tests: the
		 * same path over a one-rank communicator).  A WORLD_SIZE that is merely present is refused, not obeyed: every
		 * rank would read the whole file and the all-reduce would multiply every count by the number of ranks. */
		const char *path = argv[optind], *ph = strstr(path, "{rank}"), *md = getenv("MSX_DIST");
		static char shard[4096];
		g_dist = (md && atoi(md) != 0) || getenv("MSX_FORCE_DIST") != NULL || (ph != NULL && dist_world() > 1);
		if (!g_dist && dist_world() > 1 && getenv("RANK") && !(md && atoi(md) == 0))
			mDie("WORLD_SIZE=%d is set but the input path has no \"{rank}\": as one rank of %d this command reads its own shard "
			     "(e.g. sample.shard{rank}.bam, cut at QNAME boundaries).  Unset WORLD_SIZE or set MSX_DIST=0 to run it as an "
			     "ordinary single process.", dist_world(), dist_world());
		if (ph && g_dist) {
			snprintf(shard, sizeof shard, "%.*s%d%s", (int)(ph - path), path, dist_rank(), ph + 6);
			path = shard;
		}
		in = msh_open(path);
	}
	hdr = msh_header(in);
	prof_opts_derive(&o);
	prof_features(&o, hdr, &F);

	/* mEstimateInsertCountOnFile (:204-243): pools by QNAME over records with tid != -1 */
	memset(&b, 0, sizeof b);
	memset(&qn, 0, sizeof qn);
	row = (double *)calloc((size_t)F.n_features + 1, sizeof(double));
	if (!getenv("MSX_SERIAL_IO")) {
		/* BAM or SAM text in: the decode stage on its own thread feeds one device thread per GPU */
		static pipe_t P;
		static pshared S;
		pthread_t th_dec;
		msx_ctx *ctxs[MSH_MAX_DEVICES];
		msx_profile *profs[MSH_MAX_DEVICES];
		int dev_ids[MSH_MAX_DEVICES], k;
		double t_start = now_s(), t_ctx = 0, t_dev = 0, t_wait = 0;
		size_t n_in = 0, n_batches = 0;
		memset(&S, 0, sizeof S);
		S.n_dev = device_list(dev_ids);
		pipe_init(&P, in, 2, 0, S.n_dev);
		if (S.n_dev == 1 && !getenv("MSX_HOST_UNPACK")) pipe_enable_raw(&P, 0);   /* the record walk of every batch but the first on the device */
		S.P = &P; S.o = &o; S.F = &F;
		if (pthread_create(&th_dec, NULL, pipe_decode_thread, &P) != 0) mDie("pthread_create failed");
		for (k = 0; k < S.n_dev; k++) {
			S.dev[k].S = &S; S.dev[k].dev_id = dev_ids[k];
			if (pthread_create(&S.dev[k].th, NULL, profile_dev_thread, &S.dev[k]) != 0) mDie("pthread_create failed");
		}
		pthread_join(th_dec, NULL);
		for (k = 0; k < S.n_dev; k++) {
			pthread_join(S.dev[k].th, NULL);
			ctxs[k] = S.dev[k].ctx; profs[k] = S.dev[k].prof;
			t_ctx += S.dev[k].t_ctx; t_dev += S.dev[k].t_dev; t_wait += S.dev[k].t_wait;
			n_in += S.dev[k].n_in; n_batches += S.dev[k].n_batches;
		}
		if (P.n_filled == 0) { rbatch e; memset(&e, 0, sizeof e); S.qn = qn_check(hdr, &e); }     /* an empty input is still checked (:708) */
		qn = S.qn;
		if (getenv("MSX_TIMING")) {
			fprintf(stderr, "# batches: %zu (%zu sent ahead)%s\n", n_batches, P.n_ahead, P.comp_mode ? "; BGZF blocks inflated on the device" : "");
			if (P.n_host_inflated) fprintf(stderr, "# %zu batches inflated on the host (blocks the device refused)%s\n", P.n_host_inflated,
		                               P.comp_given_up ? "; the device was not asked any more after that" : "");
			fprintf(stderr, "# decode stage: inflate %.3f, record chase %.3f, aux scan %.3f, offsets+pools (serial) %.3f, payload copy %.3f s\n",
			        P.t_inflate, P.t_chase, P.t_scan, P.t_serial, P.t_copy);
			fprintf(stderr, "# profile pipeline: wall %.3f s; decode %.3f s (+%.3f waiting for a free slot); device: start-up %.3f, "
			        "upload+accumulate %.3f (+%.3f waiting for a batch); %zu records, %d threads, %d device%s\n",
			        now_s() - t_start, P.t_decode, P.t_wait_free, t_ctx, t_dev, t_wait, n_in, msh_threads(), S.n_dev,
			        S.n_dev > 1 ? "s" : "");
		}
		profile_combine_and_finalize(ctxs, profs, S.n_dev, o.share_type, row, &st);
		prof = profs[0];
		goto finalized;
	}
	for (;;) {
		size_t tgt = first && target < COORD_ORDER_CHECK_RECORDS ? COORD_ORDER_CHECK_RECORDS : target;
		if (msh_is_bam(in)) {
			reader *prd = &prof_rd;
			prd->in = in;
			fill_batch_bulk(prd, &b, tgt, 2, 0);
			eof = prd->done;
			have_pending = 0;
			goto batch_ready;
		}
		rb_clear(&b);
		rb_mark_group(&b);
		for (;;) {
			const uint8_t *r;
			int valid, newgrp;
			if (!have_pending) {
				if (eof || msh_read(in, &rec) < 0) { eof = 1; break; }
				have_pending = 1;
			}
			r = (const uint8_t *)rec.s;
			valid = REC_TID(r) != -1;                                 /* :223-225 */
			newgrp = valid && have_prev && strcmp(REC_QNAME(r), prev_read) != 0;
			if (newgrp && b.n >= tgt) break;
			if (newgrp && b.n > b.group_off[b.n_groups - 1]) rb_mark_group(&b);
			if (valid) { strcpy(prev_read, REC_QNAME(r)); have_prev = 1; }
			rb_append(&b, r, rec.l, 0);
			have_pending = 0;
		}
batch_ready:
		if (first) {
			qn = qn_check(hdr, &b);                                   /* :708, always for profile */
			first = 0;
			ctx_open();
			dist_begin();
			MSX(msx_profile_create(g_ctx, &prof, F.n_features, o.share_type, F.fmap, hdr->n_targets));   /* :855 */
		}
		if (b.n > 0) {
			msx_batch hb, db;
			rb_host_view(&b, &hb, 1);
			hb.cigar_off = NULL; hb.cigar = NULL; hb.md_off = NULL; hb.md = NULL;   /* profile reads tid only */
			MSX(msx_batch_upload(g_ctx, &hb, &db));
			MSX(msx_profile_accumulate(g_ctx, prof, &db, NULL));
			MSX(msx_ctx_sync(g_ctx));
			msx_batch_free(g_ctx, &db);
		}
		if (eof && !have_pending) break;
	}
	/* mInsertCountToAbundanceMatrix (:248-425) */
	{
		msx_ctx *one = g_ctx;
		profile_combine_and_finalize(&one, &prof, 1, o.share_type, row, &st);
	}
finalized:
	if (g_dist && dist_rank() != 0) {             /* every rank holds the same result; rank 0 reports it */
		msx_dist_finalize(g_ctx);
		fast_exit();
		return 0;
	}
	cl = command_line(argc, argv);
	profile_report(&o, &F, &st, row, &qn, cl);

	fast_exit();
	msx_profile_destroy(g_ctx, prof);
	msx_ctx_destroy(g_ctx);
	msh_close(in);
	free(row);
	free(cl);
	return 0;
}

/* ------------------------------------------------------------------------ */
/* coverage (msam_coverage.c:143-390)                                         */
/* ------------------------------------------------------------------------ */
static void coverage_help(FILE *out) {
	fprintf(out,
	        "Usage:\n------\n\n%s coverage [-Sxz] <bamfile> [--help] -o <file> [--summary] [-w <int>]\n"
	        "\nGeneral options:\n----------------\n\n"
	        "These options specify the input/output formats of BAM/SAM files \n(same meaning as in 'samtools view'):\n"
	        "  -S                        input is SAM (default: false)\n"
	        "  <bamfile>                 input SAM/BAM file\n"
	        "  --help                    print this help and exit\n\n"
	        "Specific options:\n-----------------\n\n"
	        "  -o <file>                 name of output file (required)\n"
	        "  --summary                 do not report per-position coverage but report fraction of sequence covered (default: false)\n"
	        "  -x, --skipuncovered       do not report coverage for sequences without aligned reads (default: false)\n"
	        "  -w, --wordsize=<int>      number of words (coverage values) per line (default: 17)\n"
	        "  -z, --gzip                compress output file using gzip (default; option retained for backward compatibility)\n",
	        PROGRAM);
}

/* The coverage report, written like the profile's text: every thread formats its share of a round's targets and makes a
 * gzip member of it, the members are written in order (mWriteCoverageToStream / mWriteCoverageSummaryToStream,
 * msam_coverage.c:143-219).  A round is as many targets as hold COV_ROUND_CELLS positions: per-position text of a
 * million references would not fit the memory at once. */
#define COV_ROUND_CELLS ((int64_t)48 << 20)
typedef struct {
	const msh_hdr *hdr;
	const int64_t *off, *touched, *sum;
	const int32_t *cov;
	const uint8_t *covered;
	int summary, skip;
	long w;
	int32_t t_lo, t_hi;                  /* targets of the round */
	int32_t cut[MSH_POOL_MAX + 1];       /* and the threads' shares of them */
	kstr text[MSH_POOL_MAX], gz[MSH_POOL_MAX];
} covrep_job;
static void covrep_worker(void *arg, int th, int nth) {
	covrep_job *J = (covrep_job *)arg;
	kstr *k = &J->text[th];
	int32_t tid;
	(void)nth;
	k->l = 0;
	for (tid = J->cut[th]; tid < J->cut[th + 1]; tid++) {
		const int64_t tlen = J->hdr->target_len[tid];
		int64_t i;
		if (J->summary) {
			if (!J->covered[tid]) {
				if (!J->skip) ks_printf(k, "%s\t%d\t%d\n", J->hdr->target_name[tid], 0, 0);
			} else {
				ks_printf(k, "%s\t%.8f\t%.2f\n", J->hdr->target_name[tid], 1.0 * J->touched[tid] / tlen, 1.0 * J->sum[tid] / tlen);
			}
		} else {
			const int32_t *cv = J->cov + J->off[tid];
			if (!J->covered[tid] && J->skip) continue;
			ks_printf(k, ">%s\n", J->hdr->target_name[tid]);
			/* "%d%c" per position, without printf: twelve bytes at most each */
			ks_reserve(k, k->l + (size_t)(tlen > 0 ? tlen : 1) * 12 + 16);
			{
				char *o = k->s + k->l;
				long col = 0;
				for (i = 0; i < tlen; i++) {
					int32_t v = (J->covered[tid] && tlen > 0) ? cv[i] : 0;
					char tmp[12];
					int n = 0;
					uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
					if (v < 0) *o++ = '-';
					do { tmp[n++] = (char)('0' + u % 10u); u /= 10u; } while (u);
					while (n) *o++ = tmp[--n];
					col++;
					*o++ = (i == tlen - 1 || col % J->w == 0) ? '\n' : ' ';
				}
				if (tlen <= 0) { *o++ = '0'; *o++ = '\n'; }
				k->l = (size_t)(o - k->s);
			}
		}
	}
	J->gz[th].l = 0;
	if (k->l) gz_member(k, &J->gz[th]);
}

int msam_coverage_main(int argc, char *argv[]) {
	static const struct option lopts[] = {{"help", no_argument, 0, 1000},    {"summary", no_argument, 0, 1001},
	                                      {"skipuncovered", no_argument, 0, 'x'}, {"wordsize", required_argument, 0, 'w'},
	                                      {"gzip", no_argument, 0, 'z'},     {0, 0, 0, 0}};
	const char *o_out = NULL;
	int n_out = 0, o_summary = 0, o_skip = 0, o_help = 0, n_w = 0, nerrors = 0, c;
	long v_w = 17;
	msh_in *in;
	const msh_hdr *hdr;
	reader rd;
	rbatch b;
	int out_fd;
	int64_t *off, total;
	void *d_off, *d_cov, *d_covered;
	int32_t *cov, tid;
	uint8_t *covered;
	kstr rec = {0, 0, 0};
	size_t target = batch_target();
	static pipe_t P;
	pthread_t th_dec;
	int piped = 0;
	int64_t *cs_touched = NULL, *cs_sum = NULL;
	double t_cov[4] = {0, 0, 0, 0}, t_cov0 = now_s();

	opterr = 0;
	optind = 1;
	while ((c = getopt_long(argc, argv, "Sxzo:w:", lopts, NULL)) != -1) {
		switch (c) {
		case 'S': case 'z': break;
		case 'x': o_skip++; break;
		case 'o': n_out++; o_out = optarg; break;
		case 'w': n_w++; v_w = strtol(optarg, NULL, 10); break;
		case 1000: o_help++; break;
		case 1001: o_summary++; break;
		default:
			fprintf(stderr, "%s: invalid option \"%s\"\n", PROGRAM, argv[optind - 1]);
			nerrors++;
		}
	}
	if (o_help > 0 || argc < 2) { coverage_help(stdout); exit(EXIT_SUCCESS); }
	if (argc - optind < 1) { fprintf(stderr, "%s: missing option <bamfile>\n", PROGRAM); nerrors++; }
	if (n_out == 0) { fprintf(stderr, "%s: missing option -o <file>\n", PROGRAM); nerrors++; }
	if (nerrors > 0) {                                                /* msam_coverage.c:322-326 (stderr) */
		fprintf(stderr, "Use --help for usage instructions!\n");
		mQuit("");
	}
	if (n_w > 0 && v_w < 1) {                                         /* :332-339 */
		fprintf(stdout, "-w must be a non-zero positive integer\n");
		coverage_help(stdout);
		mQuit("");
	}
	if (n_out != 1) { fprintf(stdout, "requires -o\n"); coverage_help(stdout); mQuit(""); }
	out_fd = strcmp(o_out, "-") == 0 ? fileno(stdout) : open(o_out, O_WRONLY | O_CREAT | O_TRUNC, 0666);   /* :348-353 */
	if (out_fd < 0) mDie("Cannot open %s for writing", o_out);

	in = msh_open(argv[optind]);
	hdr = msh_header(in);
	/* BAM input goes through the pipeline of filter and profile: a decode thread hands over batches (the first one walked
	 * on the host, the others as compressed blocks), this thread is the device stage.  MSX_SERIAL_IO=1: the
	 * record-at-a-time reader below, one batch at a time. */
	piped = msh_is_bam(in) && !getenv("MSX_SERIAL_IO");
	if (piped) {
		pipe_init(&P, in, 0, 1, 1);
		if (!getenv("MSX_HOST_UNPACK")) pipe_enable_raw(&P, 0);
		if (pthread_create(&th_dec, NULL, pipe_decode_thread, &P) != 0) mDie("pthread_create failed");
	}
	ctx_open();
	off = (int64_t *)malloc(sizeof(int64_t) * ((size_t)hdr->n_targets + 1));
	off[0] = 0;
	for (tid = 0; tid < hdr->n_targets; tid++) off[tid + 1] = off[tid] + hdr->target_len[tid];
	total = off[hdr->n_targets];
	MSX(msx_dev_alloc(g_ctx, &d_off, sizeof(int64_t) * ((size_t)hdr->n_targets + 1)));
	MSX(msx_dev_alloc(g_ctx, &d_cov, 4 * (size_t)total + 8));
	MSX(msx_dev_alloc(g_ctx, &d_covered, (size_t)hdr->n_targets + 8));
	MSX(msx_host_to_dev(g_ctx, d_off, off, sizeof(int64_t) * ((size_t)hdr->n_targets + 1)));
	MSX(msx_dev_zero(g_ctx, d_cov, 4 * (size_t)total + 8));
	MSX(msx_dev_zero(g_ctx, d_covered, (size_t)hdr->n_targets + 8));

	/* mEstimateCoverageOnFile (:106-139): every alignment adds 1, pools do not matter */
	t_cov[0] = now_s();
	memset(&rd, 0, sizeof rd);
	memset(&b, 0, sizeof b);
	rd.in = in;
	if (piped) {
		msx_unpack *unpack = NULL;
		ahead_q ahead = {{PQ_NONE, PQ_NONE}, 0, 0};
		if (P.raw_mode) { MSX(msx_unpack_create(g_ctx, &unpack)); pin_start(&P, 0); }
		for (;;) {
			const int si = ahead.n ? ahead_pop(&ahead) : pq_pop(&P.q_dev);
			pslot *s;
			msx_batch hb, db;
			if (si == PQ_END) break;
			s = &P.slot[si];
			if (s->raw) {
				msx_unpack_params up;
				msx_unpack_result ur;
				pin_wait(&P, s);
				if (s->has_seed) MSX(msx_unpack_seed(g_ctx, unpack, (const uint8_t *)s->seed.s, s->seed.l, NULL));
				memset(&up, 0, sizeof up);
				up.pool_mode = 0; up.want_stats = 1; up.n_targets = hdr->n_targets; up.last = s->last;
				unpack_slot_enqueue(&P, s, unpack, &up);
				unpack_slots_ahead(&P, unpack, &ahead);
				unpack_slot_finish(&P, s, unpack, &up, &ur, &db);
				if (ur.n_records > 0)
					MSX(msx_coverage_accumulate(g_ctx, &db, (const int64_t *)d_off, hdr->n_targets, total, (int32_t *)d_cov,
					                            (uint8_t *)d_covered));
			} else if (s->b.n > 0) {
				rb_host_view(&s->b, &hb, 0);
				hb.md_off = NULL; hb.md = NULL; hb.nm = NULL; hb.as = NULL;
				MSX(msx_batch_upload(g_ctx, &hb, &db));
				MSX(msx_coverage_accumulate(g_ctx, &db, (const int64_t *)d_off, hdr->n_targets, total, (int32_t *)d_cov,
				                            (uint8_t *)d_covered));
				MSX(msx_ctx_sync(g_ctx));
				msx_batch_free(g_ctx, &db);
			}
			pq_push(&P.q_free, si);
		}
		pthread_join(th_dec, NULL);
		MSX(msx_ctx_sync(g_ctx));
		pin_join(&P);
		msx_unpack_destroy(g_ctx, unpack);
	} else
	for (;;) {
		if (msh_is_bam(in)) {
			fill_batch_bulk(&rd, &b, target, 0, 1);
		} else {
			rb_clear(&b);
			while (b.n < target && msh_read(in, &rec) == 0) rb_append(&b, (const uint8_t *)rec.s, rec.l, 1);
			if (b.n < target) rd.done = 1;
		}
		if (b.n > 0) {
			msx_batch hb, db;
			rb_host_view(&b, &hb, 0);
			hb.md_off = NULL; hb.md = NULL; hb.nm = NULL; hb.as = NULL;
			MSX(msx_batch_upload(g_ctx, &hb, &db));
			MSX(msx_coverage_accumulate(g_ctx, &db, (const int64_t *)d_off, hdr->n_targets, total, (int32_t *)d_cov,
			                            (uint8_t *)d_covered));
			MSX(msx_ctx_sync(g_ctx));
			msx_batch_free(g_ctx, &db);
		}
		if (rd.done) break;
	}
	t_cov[1] = now_s();
	MSX(msx_coverage_finish(g_ctx, (int32_t *)d_cov, total));
	MSX(msx_ctx_sync(g_ctx));
	t_cov[2] = now_s();
	covered = (uint8_t *)malloc((size_t)hdr->n_targets + 1);
	if (o_summary) {
		/* the summary needs two sums per target, not the depths: they are taken on the device */
		cs_touched = (int64_t *)calloc((size_t)hdr->n_targets + 1, sizeof(int64_t));
		cs_sum = (int64_t *)calloc((size_t)hdr->n_targets + 1, sizeof(int64_t));
		if (!covered || !cs_touched || !cs_sum) mDie("Out of memory");
		MSX(msx_coverage_summary(g_ctx, (const int32_t *)d_cov, (const int64_t *)d_off, hdr->n_targets, cs_touched, cs_sum));
		cov = NULL;
	} else {
		cov = (int32_t *)malloc(4 * (size_t)(total > 0 ? total : 1));
		if (!cov || !covered) mDie("Out of memory");
		MSX(msx_dev_to_host(g_ctx, cov, d_cov, 4 * (size_t)total));
	}
	MSX(msx_dev_to_host(g_ctx, covered, d_covered, (size_t)hdr->n_targets));
	t_cov[3] = now_s();

	{
		static covrep_job J;
		int nth = msh_threads(), t, any = 0;
		int32_t t0 = 0;
		if (nth > MSH_POOL_MAX) nth = MSH_POOL_MAX;
		memset(&J, 0, sizeof J);
		J.hdr = hdr; J.off = off; J.touched = cs_touched; J.sum = cs_sum; J.cov = cov; J.covered = covered;
		J.summary = o_summary; J.skip = o_skip; J.w = v_w;
		while (t0 < hdr->n_targets) {
			/* a round: targets [t0, t1) -- by positions for the per-position text, by lines for the summary */
			int32_t t1 = t0;
			int64_t cells = 0;
			while (t1 < hdr->n_targets && (cells == 0 || cells + (o_summary ? 64 : (int64_t)hdr->target_len[t1]) <= COV_ROUND_CELLS)) {
				cells += o_summary ? 64 : (int64_t)hdr->target_len[t1];
				t1++;
			}
			J.t_lo = t0; J.t_hi = t1;
			{   /* the threads' shares: equal positions (equal lines) */
				int use = nth;
				int32_t q = t0;
				int64_t acc = 0;
				if (t1 - t0 < use) use = t1 - t0;
				J.cut[0] = t0;
				for (t = 1; t < use; t++) {
					const int64_t want = cells * t / use;
					while (q < t1 && acc < want) { acc += o_summary ? 64 : (int64_t)hdr->target_len[q]; q++; }
					J.cut[t] = q;
				}
				J.cut[use] = t1;
				msh_parallel(use, covrep_worker, &J);
				for (t = 0; t < use; t++)
					if (J.gz[t].l) { fd_write_all(out_fd, J.gz[t].s, J.gz[t].l); any = 1; }
			}
			t0 = t1;
		}
		if (!any) {                      /* nothing to report: still a valid (empty) gzip file */
			kstr e = {0, 0, 0}, z = {0, 0, 0};
			gz_member(&e, &z);
			fd_write_all(out_fd, z.s, z.l);
			free(z.s);
		}
		for (t = 0; t < MSH_POOL_MAX; t++) { free(J.text[t].s); free(J.gz[t].s); }
		if (out_fd != fileno(stdout) && close(out_fd) != 0) mDie("Write failed");
	}
	if (getenv("MSX_TIMING"))
		fprintf(stderr, "# coverage: input and device set up %.3f s, records through %.3f, prefix sums %.3f, depths to the host %.3f, "
		        "report %.3f (%lld cells)\n", t_cov[0] - t_cov0, t_cov[1] - t_cov[0], t_cov[2] - t_cov[1], t_cov[3] - t_cov[2],
		        now_s() - t_cov[3], (long long)total);
	msx_dev_free(g_ctx, d_off);
	msx_dev_free(g_ctx, d_cov);
	msx_dev_free(g_ctx, d_covered);
	msx_ctx_destroy(g_ctx);
	msh_close(in);
	free(cov);
	free(covered);
	free(off);
	return 0;
}

/* ------------------------------------------------------------------------ */
/* msamtools.c:8-49                                                           */
/* ------------------------------------------------------------------------ */
static int usage(FILE *out) {
	fprintf(out, "\n");
	fprintf(out, "Program: %s (Metagenomics-related extension to samtools; MI355X filter/profile path)\n", PROGRAM);
	fprintf(out, "Version: %s (git %s; own BGZF/BAM reader, no htslib)\n", MSH_VERSION, MSH_GIT_COMMIT);
	fprintf(out, "\n");
	fprintf(out, "Usage:   %s <command> [options]\n\n", PROGRAM);
	fprintf(out, "Commands:\n");
	fprintf(out, " -- Filtering\n");
	fprintf(out, "     filter         filter alignments based on alignment statistics\n");
	fprintf(out, "\n");
	fprintf(out, " -- Profiling\n");
	fprintf(out, "     profile        estimate relative abundance profile of reference sequences or genomes in bam file\n");
	fprintf(out, "\n");
	fprintf(out, " -- Coverage\n");
	fprintf(out, "     coverage       estimate per-base or per-sequence read coverage of each reference sequence\n");
	fprintf(out, "\n");
	return 1;
}

/* Host I/O self-test (no GPU): `msamtools recode [-b|-u|-h] <file>` reads any
 * supported input and writes every record back out.  Not part of the
 * reference's surface; used by the test-suite to check the readers/writers. */
static int recode_main(int argc, char *argv[]) {
	int mode = MSH_OUT_SAM, i;
	const char *path = NULL;
	msh_in *in;
	msh_out *out;
	kstr rec = {0, 0, 0};
	for (i = 1; i < argc; i++) {
		if (strcmp(argv[i], "-b") == 0) mode = MSH_OUT_BAM;
		else if (strcmp(argv[i], "-u") == 0) mode = MSH_OUT_UBAM;
		else if (strcmp(argv[i], "-h") == 0) mode = MSH_OUT_SAM_HDR;
		else path = argv[i];
	}
	if (!path) mQuit("usage: %s recode [-b|-u|-h] <file>", PROGRAM);
	in = msh_open(path);
	out = msh_out_open(stdout, mode, msh_header(in), msh_header(in)->text.s ? msh_header(in)->text.s : "");
	while (msh_read(in, &rec) == 0) msh_write(out, (const uint8_t *)rec.s, rec.l);
	msh_out_close(out);
	msh_close(in);
	return 0;
}

typedef struct {
	const msx_batch *hb;
	int with_seq, pass;
	int64_t g0, g1, first_group;
	size_t r0;
	size_t *rec_off;          /* pass 0: size of record i (with its 4-byte length); then its offset in blob */
	int32_t *idx;
	uint8_t *blob;
	size_t blob_cap;
} synth_job;

static void synth_worker(void *arg, int tid, int nth) {
	synth_job *J = (synth_job *)arg;
	const msx_batch *hb = J->hb;
	const int64_t span = J->g1 - J->g0, ga = J->g0 + span * tid / nth, gb = J->g0 + span * (tid + 1) / nth;
	int64_t g, k;
	for (g = ga; g < gb; g++) {
		char qn[32];
		int ql = snprintf(qn, sizeof qn, "sim%08lld", (long long)(J->first_group + g));
		for (k = hb->group_off[g]; k < hb->group_off[g + 1]; k++) {
			const uint32_t nc = hb->cigar_off[k + 1] - hb->cigar_off[k], ml = hb->md_off[k + 1] - hb->md_off[k];
			const uint32_t l_seq = J->with_seq ? 100 : 0;
			const size_t len = 32 + (size_t)ql + 1 + 4 * (size_t)nc + (J->with_seq ? 150 : 0) + 4 + 3 + ml + 1 + 4;
			const size_t i = (size_t)k - J->r0;
			uint8_t *o;
			uint32_t v[8], q;
			if (J->pass == 0) { J->rec_off[i] = 4 + len; continue; }
			o = J->blob + J->rec_off[i];
			o[0] = (uint8_t)len; o[1] = (uint8_t)(len >> 8); o[2] = (uint8_t)(len >> 16); o[3] = (uint8_t)(len >> 24);
			o += 4;
			v[0] = (uint32_t)hb->tid[k]; v[1] = (uint32_t)hb->pos[k];
			v[2] = (uint32_t)(ql + 1) | 255u << 8 | 4680u << 16;
			v[3] = nc | (uint32_t)hb->flag[k] << 16;
			v[4] = l_seq; v[5] = (uint32_t)-1; v[6] = (uint32_t)-1; v[7] = 0;
			for (q = 0; q < 8; q++) { o[4*q] = (uint8_t)v[q]; o[4*q+1] = (uint8_t)(v[q] >> 8); o[4*q+2] = (uint8_t)(v[q] >> 16); o[4*q+3] = (uint8_t)(v[q] >> 24); }
			o += 32;
			memcpy(o, qn, (size_t)ql + 1); o += ql + 1;
			memcpy(o, hb->cigar + hb->cigar_off[k], 4 * (size_t)nc); o += 4 * (size_t)nc;
			if (J->with_seq) {
				for (q = 0; q < 50; q++) *o++ = (uint8_t)(0x12 + (int)((k + q) & 3) * 0x11);   /* A/C/G/T-ish nibbles */
				for (q = 0; q < 100; q++) *o++ = 40;
			}
			memcpy(o, "NMC", 3); o += 3; *o++ = (uint8_t)(hb->nm[k] & 0xff);
			memcpy(o, "MDZ", 3); o += 3; memcpy(o, hb->md + hb->md_off[k], ml); o += ml; *o++ = 0;
			memcpy(o, "ASc", 3); o += 3; *o++ = (uint8_t)(hb->as[k] & 0xff);
		}
	}
}

/* `msamtools synth --groups N --refs R [--seed S] [--seq] [-b|-u]`: writes the
 * library's deterministic synthetic alignment stream (BASELINE.md section 2
 * model) as a QNAME-grouped BAM, for end-to-end host-pipeline timing.  No GPU. */
static int synth_main(int argc, char *argv[]) {
	msx_synth_params sp = {13579, 100000, 10000, 4, 0};
	msx_synth_sizes sz;
	msx_batch hb;
	int mode = MSH_OUT_UBAM, with_seq = 0, i;
	msh_hdr hdr;
	msh_out *out;
	kstr rec = {0, 0, 0};
	int64_t g, k;
	for (i = 1; i < argc; i++) {
		if (strcmp(argv[i], "--groups") == 0 && i + 1 < argc) sp.n_groups = atoll(argv[++i]);
		else if (strcmp(argv[i], "--refs") == 0 && i + 1 < argc) sp.n_refs = atoi(argv[++i]);
		else if (strcmp(argv[i], "--seed") == 0 && i + 1 < argc) sp.seed = strtoull(argv[++i], NULL, 10);
		else if (strcmp(argv[i], "--seq") == 0) with_seq = 1;
		else if (strcmp(argv[i], "-b") == 0) mode = MSH_OUT_BAM;
		else if (strcmp(argv[i], "-u") == 0) mode = MSH_OUT_UBAM;
		else mQuit("usage: %s synth --groups N --refs R [--seed S] [--seq] [-b|-u]", PROGRAM);
	}
	if (msx_synth_host(&sp, &hb, &sz) != MSX_OK) mDie("%s", msx_last_error(NULL));
	memset(&hdr, 0, sizeof hdr);
	ks_puts(&hdr.text, "@HD\tVN:1.6\tSO:queryname\n");
	hdr.n_targets = sp.n_refs;
	hdr.target_name = (char **)malloc(sizeof(char *) * (size_t)sp.n_refs);
	hdr.target_len = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)sp.n_refs);
	for (i = 0; i < sp.n_refs; i++) {
		char nm[32];
		/* msx_synth_ref_len(): 400 + 12 hash bits of the tid; any length >= pos+150 is valid for the model */
		snprintf(nm, sizeof nm, "ref%07d", i);
		hdr.target_name[i] = strdup(nm);
		hdr.target_len[i] = 4496;
		ks_printf(&hdr.text, "@SQ\tSN:%s\tLN:%u\n", nm, hdr.target_len[i]);
	}
	out = msh_out_open(stdout, mode, &hdr, hdr.text.s);
	{
		/* records are built and compressed chunk by chunk on the worker pool */
		const int64_t chunk = 1 << 18;
		synth_job J;
		memset(&J, 0, sizeof J);
		J.hb = &hb; J.with_seq = with_seq; J.first_group = sp.first_group;
		for (g = 0; g < hb.n_groups; g += chunk) {
			const int64_t g1 = g + chunk < hb.n_groups ? g + chunk : hb.n_groups;
			const size_t r0 = hb.group_off[g], r1 = hb.group_off[g1], n = r1 - r0;
			size_t i, tot = 0;
			J.g0 = g; J.g1 = g1; J.r0 = r0;
			J.rec_off = (size_t *)realloc(J.rec_off, (n + 1) * sizeof(size_t));
			J.idx = (int32_t *)realloc(J.idx, (n + 1) * sizeof(int32_t));
			J.pass = 0;
			msh_parallel(msh_threads(), synth_worker, &J);          /* sizes */
			for (i = 0; i < n; i++) { size_t sz = J.rec_off[i]; J.rec_off[i] = tot; tot += sz; J.idx[i] = (int32_t)i; }
			J.rec_off[n] = tot;
			if (tot > J.blob_cap) { J.blob_cap = tot + tot / 4; J.blob = (uint8_t *)realloc(J.blob, J.blob_cap); if (!J.blob) mDie("Out of memory"); }
			J.pass = 1;
			msh_parallel(msh_threads(), synth_worker, &J);          /* bytes */
			msh_write_many(out, J.blob, J.rec_off, J.idx, n);
		}
		free(J.rec_off); free(J.idx); free(J.blob);
	}
	(void)rec; (void)k;
	msh_out_close(out);
	msx_synth_host_free(&hb);
	return 0;
}

/* hidden, host only: `msamtools pipetest <mode 0|1|2> <stats 0|1> <bam>` runs the pipeline's decode stage
 * alone and prints one line per batch boundary-independent digest: records, pools, and a hash over every
 * SoA field -- and the same computed with the record-at-a-time reader.  tests/test_host_cli.py compares. */
static uint64_t mix_u64(uint64_t h, uint64_t v) {
	h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
	return h;
}

static int pipetest_main(int argc, char *argv[]) {
	static pipe_t P;
	pthread_t th;
	msh_in *in;
	int mode, stats;
	uint64_t h = 0, hp = 0, pools = 0, recs = 0, batches = 0;
	if (argc < 4) mQuit("usage: %s pipetest <mode> <stats> <bam>", PROGRAM);
	mode = atoi(argv[1]); stats = atoi(argv[2]);
	in = msh_open(argv[3]);
	pipe_init(&P, in, mode, stats, 1);
	if (pthread_create(&th, NULL, pipe_decode_thread, &P) != 0) mDie("pthread_create failed");
	for (;;) {
		const int si = pq_pop(&P.q_dev);
		pslot *s;
		rbatch *b;
		size_t i, g;
		if (si == PQ_END) break;
		s = &P.slot[si];
		b = &s->b;
		batches++;
		for (i = 0; i < b->n; i++) {
			const uint8_t *r = RB_REC(b, i);
			size_t len = RB_LEN(b, i), k;
			h = mix_u64(h, len);
			for (k = 0; k < len; k += 8) { uint64_t v = 0; memcpy(&v, r + k, len - k < 8 ? len - k : 8); h = mix_u64(h, v); }
			h = mix_u64(h, b->flag[i]); h = mix_u64(h, b->rflags[i]); h = mix_u64(h, (uint32_t)b->tid[i]);
			h = mix_u64(h, (uint32_t)b->pos[i]); h = mix_u64(h, (uint32_t)b->nm[i]); h = mix_u64(h, (uint32_t)b->as[i]);
			if (stats) {
				for (k = b->cigar_off[i]; k < b->cigar_off[i + 1]; k++) h = mix_u64(h, b->cigar[k]);
				for (k = b->md_off[i]; k < b->md_off[i + 1]; k++) h = mix_u64(h, b->md[k]);
			}
		}
		if (mode != 0) {
			for (g = 0; g < b->n_groups; g++) hp = mix_u64(hp, (uint64_t)(recs + b->group_off[g]));
			pools += b->n_groups;
		}
		recs += b->n;
		pq_push(&P.q_free, si);
	}
	pthread_join(th, NULL);
	printf("pipeline records=%llu pools=%llu hash=%016llx pool_hash=%016llx batches=%llu\n", (unsigned long long)recs,
	       (unsigned long long)pools, (unsigned long long)h, (unsigned long long)hp, (unsigned long long)batches);
	msh_close(in);
	{   /* the same digest through msh_read and the per-record rules of msam_filter.c:120-125,170 / msam_profile.c:223-232 */
		rbatch b;
		kstr rec = {0, 0, 0};
		char prev[256];
		int have_prev = 0;
		uint64_t h2 = 0, hp2 = 0, pools2 = 0, recs2 = 0;
		memset(&b, 0, sizeof b);
		in = msh_open(argv[3]);
		while (msh_read(in, &rec) == 0) {
			const uint8_t *r = (const uint8_t *)rec.s;
			size_t len = rec.l, k;
			int counts = mode == 1 ? !(REC_FLAG(r) & 4) : (mode == 2 && REC_TID(r) != -1);
			int rule = mode == 1 || (mode == 2 && REC_TID(r) != -1);
			rb_clear(&b);
			rb_append(&b, r, len, stats);
			if (mode != 0 && (recs2 == 0 || (rule && have_prev && strcmp(REC_QNAME(r), prev) != 0))) {
				hp2 = mix_u64(hp2, recs2);
				pools2++;
			}
			h2 = mix_u64(h2, len);
			for (k = 0; k < len; k += 8) { uint64_t v = 0; memcpy(&v, r + k, len - k < 8 ? len - k : 8); h2 = mix_u64(h2, v); }
			h2 = mix_u64(h2, b.flag[0]); h2 = mix_u64(h2, b.rflags[0]); h2 = mix_u64(h2, (uint32_t)b.tid[0]);
			h2 = mix_u64(h2, (uint32_t)b.pos[0]); h2 = mix_u64(h2, (uint32_t)b.nm[0]); h2 = mix_u64(h2, (uint32_t)b.as[0]);
			if (stats) {
				for (k = 0; k < b.cigar_off[1]; k++) h2 = mix_u64(h2, b.cigar[k]);
				for (k = 0; k < b.md_off[1]; k++) h2 = mix_u64(h2, b.md[k]);
			}
			if (counts) { strcpy(prev, REC_QNAME(r)); have_prev = 1; }
			recs2++;
		}
		printf("serial   records=%llu pools=%llu hash=%016llx pool_hash=%016llx\n", (unsigned long long)recs2,
		       (unsigned long long)pools2, (unsigned long long)h2, (unsigned long long)hp2);
		msh_close(in);
		return (recs2 == recs && h2 == h && pools2 == pools && hp2 == hp) ? 0 : 1;
	}
}

/* hidden, host only: `msamtools digest <file>` prints the number of records and an order-sensitive 64-bit
 * digest of (QNAME, FLAG, tid, pos) over the record stream:  sum over records i = 0.. of (i + 1) * g(record i)
 * mod 2^64, g = a 64-bit mix of the three integers xor FNV-1a of the QNAME.  The tests and bench.py compute the
 * same figure from the oracle's emit list (tests/digest.py), so that an output of tens of millions of records
 * is compared with the oracle's -- which records, in which order -- without a text round trip. */
static uint64_t dg_mix(uint64_t x) {
	x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
	x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
	x ^= x >> 33;
	return x;
}
static uint64_t dg_record(const uint8_t *r) {
	const char *q = REC_QNAME(r);
	uint64_t h = 1469598103934665603ull;
	const uint64_t v = (uint64_t)REC_FLAG(r) + (uint64_t)(uint32_t)REC_TID(r) * 0x9e3779b97f4a7c15ull +
	                   (uint64_t)(uint32_t)REC_POS(r) * 0xc2b2ae3d27d4eb4full;
	while (*q) { h ^= (uint8_t)*q++; h *= 1099511628211ull; }
	return dg_mix(v) ^ h;
}
typedef struct { const rbatch *b; uint64_t first, part[MSH_POOL_MAX]; } digest_job;
static void digest_worker(void *arg, int tid, int nth) {
	digest_job *J = (digest_job *)arg;
	const rbatch *b = J->b;
	const size_t lo = b->n * (size_t)tid / (size_t)nth, hi = b->n * (size_t)(tid + 1) / (size_t)nth;
	uint64_t s = 0;
	size_t i;
	for (i = lo; i < hi; i++) {
		const uint8_t *r = RB_REC(b, i);
		msh_rec_check(r, RB_LEN(b, i));
		s += (J->first + (uint64_t)i + 1) * dg_record(r);
	}
	J->part[tid] = s;
}

static int digest_main(int argc, char *argv[]) {
	msh_in *in;
	uint64_t n = 0, h = 0;
	if (argc < 2) mQuit("usage: %s digest <file>", PROGRAM);
	in = msh_open(argv[1]);
	if (msh_is_bam(in)) {
		static pipe_t P;
		pthread_t th;
		pipe_init(&P, in, 0, 0, 1);
		if (pthread_create(&th, NULL, pipe_decode_thread, &P) != 0) mDie("pthread_create failed");
		for (;;) {
			const int si = pq_pop(&P.q_dev);
			pslot *s;
			digest_job J;
			int nth = msh_threads(), t;
			if (si == PQ_END) break;
			s = &P.slot[si];
			if (nth > MSH_POOL_MAX) nth = MSH_POOL_MAX;
			J.b = &s->b; J.first = n;
			msh_parallel(nth, digest_worker, &J);
			for (t = 0; t < nth; t++) h += J.part[t];
			n += s->b.n;
			pq_push(&P.q_free, si);
		}
		pthread_join(th, NULL);
	} else {
		kstr rec = {0, 0, 0};
		while (msh_read(in, &rec) == 0) { n++; h += n * dg_record((const uint8_t *)rec.s); }
	}
	printf("records=%llu digest=%016llx\n", (unsigned long long)n, (unsigned long long)h);
	msh_close(in);
	return 0;
}

/* hidden, host only: `msamtools restream [-b|-u] <file>` decodes the input through the pipeline's decode stage and
 * writes every batch with msh_write_stream -- the writer of device-unpacked batches: a ready-made record stream, cut
 * into BGZF payloads where they fall -- so that this writer is tested without a GPU. */
static int restream_main(int argc, char *argv[]) {
	static pipe_t P;
	pthread_t th;
	int mode = MSH_OUT_UBAM, i;
	const char *path = NULL;
	msh_in *in;
	msh_out *out;
	for (i = 1; i < argc; i++) {
		if (strcmp(argv[i], "-b") == 0) mode = MSH_OUT_BAM;
		else if (strcmp(argv[i], "-u") == 0) mode = MSH_OUT_UBAM;
		else path = argv[i];
	}
	if (!path) mQuit("usage: %s restream [-b|-u] <file>", PROGRAM);
	in = msh_open(path);
	out = msh_out_open(stdout, mode, msh_header(in), msh_header(in)->text.s ? msh_header(in)->text.s : "");
	pipe_init(&P, in, 0, 0, 1);
	if (pthread_create(&th, NULL, pipe_decode_thread, &P) != 0) mDie("pthread_create failed");
	for (;;) {
		const int si = pq_pop(&P.q_dev);
		pslot *s;
		if (si == PQ_END) break;
		s = &P.slot[si];
		if (s->b.n) msh_write_stream(out, s->b.base + s->b.rec_off[0], s->b.rec_off[s->b.n] - s->b.rec_off[0]);
		pq_push(&P.q_free, si);
	}
	pthread_join(th, NULL);
	msh_out_close(out);
	msh_close(in);
	return 0;
}

/* `msamtools rawtest [--blocks N] <file.bam>` (hidden, host only): the decode stage's feed of the device inflater --
 * msh_raw_append: block headers walked, DEFLATE payloads copied, table written -- checked without a device: every batch's
 * table is inflated by msh_inflate_table, and the length and CRC-32 of the whole record stream (everything behind the
 * BAM header) are printed for the test-suite to compare with an independent decompression. */
static int rawtest_main(int argc, char *argv[]) {
	const char *path = NULL;
	int max_blocks = 64, i;
	msh_in *in;
	uint8_t *comp, *out = NULL, *head = NULL;
	size_t cap = (size_t)8 << 20, out_cap = 0, hl = 0, hc = 0, total = 0, batches = 0, blocks = 0;
	msx_bgzf_block *blk;
	uLong crc = crc32(0L, NULL, 0);
	for (i = 1; i < argc; i++) {
		if (strcmp(argv[i], "--blocks") == 0 && i + 1 < argc) max_blocks = atoi(argv[++i]);
		else path = argv[i];
	}
	if (!path || max_blocks < 1) mQuit("usage: %s rawtest [--blocks N] <file.bam>", PROGRAM);
	in = msh_open(path);
	if (!msh_is_bam(in)) mQuit("rawtest: BAM input only");
	/* what msh_open has inflated beyond the header comes first (one call: the span's live bytes, or the next batch) */
	msh_inflate_limit(2);
	if (msh_inflate_append(in, &head, &hl, &hc)) { crc = crc32(crc, head, (uInt)hl); total += hl; }
	msh_inflate_limit(0);
	comp = (uint8_t *)xmalloc(cap);
	blk = (msx_bgzf_block *)xmalloc((size_t)max_blocks * sizeof *blk);
	for (;;) {
		size_t len = 0, inflated = 0;
		int n = 0;
		while (n < max_blocks && (cap - len) / (65536 + 1024) > 0)
			if (!msh_raw_append(in, comp, cap, &len, blk, &n, max_blocks, &inflated)) break;
		if (n == 0) break;
		if (inflated + 64 > out_cap) { out_cap = inflated + 64; out = (uint8_t *)realloc(out, out_cap); if (!out) mDie("Out of memory"); }
		msh_inflate_table(comp, blk, n, out);
		for (i = 0; i < n; i++)
			if (blk[i].out_off != (i ? blk[i - 1].out_off + blk[i - 1].out_len : 0) || blk[i].in_off + blk[i].in_len > len)
				mDie("rawtest: inconsistent table");
		{ size_t q = 0; while (q < inflated) { const size_t k = inflated - q > 0x40000000u ? 0x40000000u : inflated - q; crc = crc32(crc, out + q, (uInt)k); q += k; } }
		total += inflated;
		batches++;
		blocks += (size_t)n;
	}
	printf("bytes=%zu crc32=%08lx batches=%zu blocks=%zu\n", total, (unsigned long)crc, batches, blocks);
	msh_close(in);
	return 0;
}

int main(int argc, char *argv[]) {
	g_t_main = now_s();
	msh_main_thread = pthread_self();
	msh_main_thread_set = 1;
	if (argc < 2) return usage(stderr);
	if (strcmp(argv[1], "keyorder") == 0) {
		/* hidden, host only: names on stdin (one per line) -> the reference's key order on stdout */
		char line[8192];
		msh_keyset *k = msh_keyset_new();
		int32_t i;
		while (fgets(line, sizeof line, stdin)) {
			line[strcspn(line, "\n")] = 0;
			msh_keyset_put(k, line, 1);
		}
		for (i = 0; i < msh_keyset_size(k); i++) printf("%s\n", msh_keyset_key(k, msh_keyset_walk(k, i)));
		msh_keyset_free(k);
		return 0;
	}
	if (strcmp(argv[1], "recode") == 0) return recode_main(argc - 1, argv + 1);
	if (strcmp(argv[1], "pipetest") == 0) return pipetest_main(argc - 1, argv + 1);
	if (strcmp(argv[1], "digest") == 0) return digest_main(argc - 1, argv + 1);
	if (strcmp(argv[1], "restream") == 0) return restream_main(argc - 1, argv + 1);
	if (strcmp(argv[1], "rawtest") == 0) return rawtest_main(argc - 1, argv + 1);
	if (strcmp(argv[1], "synth") == 0) return synth_main(argc - 1, argv + 1);
	if (strcmp(argv[1], "filter") == 0) return msam_filter_main(argc - 1, argv + 1);
	else if (strcmp(argv[1], "profile") == 0) return msam_profile_main(argc - 1, argv + 1);
	else if (strcmp(argv[1], "coverage") == 0) return msam_coverage_main(argc - 1, argv + 1);
	else if (strcmp(argv[1], "help") == 0) { usage(stdout); return 0; }
	fprintf(stderr, "[msamtools] unrecognized command '%s'\n", argv[1]);
	usage(stderr);
	return 1;
}
