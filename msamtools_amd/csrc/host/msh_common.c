/*
 * msh_common.c -- what the commands share: the calling thread's context, stage timers, the device list, record batches
 * (BAM blobs + the SoA view the kernels read) and the QNAME-grouping preflight (msam_helper.c:78-137, 295-484).
 */
#include "msh_cli.h"

#define QNAME_GROUP_CHECK_RECORDS 10000      /* msam_helper.c:4-6 (COORD_ORDER_CHECK_RECORDS: msh_cli.h) */
#define COORD_ORDER_MIN_RECORDS 10000

#define COORD_ORDER_MIN_RECORDS 10000

/* the context the calling thread works on: one device thread per GPU, each with its own (MSX_DEVICES) */
__thread msx_ctx *g_ctx;

/* stage timers, printed to stderr when MSX_TIMING is set */
double now_s(void) {
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
double t_decode, t_upload, t_gpu, t_fetch, t_write;

/* Everything has been written: leave without tearing down the HIP runtime, the page-locked arenas and
 * a gigabyte of batch buffers (a quarter of a second on a one-second run).  MSX_CLEAN_EXIT=1 keeps the
 * orderly shutdown (leak checks). */
double g_t_main;     /* now_s() at the head of main (MSX_TIMING) */

/* MSX_TIMING=2: the mappings that hold more than 32 MB, at the checkpoints that call this (what the kernel has to take apart
 * when the process ends is part of the command's wall time) */
void mem_report(const char *tag) {
	const char *e = getenv("MSX_TIMING");
	FILE *f;
	char line[256], head[256] = "";
	long rss = 0, ahp = 0;
	char flags[128] = "";
	if (!e || atoi(e) < 2 || !(f = fopen("/proc/self/smaps", "r"))) return;
	fprintf(stderr, "# mappings at %s:\n", tag);
	while (fgets(line, sizeof line, f)) {
		if (!strstr(line, " kB") && strchr(line, '-') && ((line[0] >= '0' && line[0] <= '9') || (line[0] >= 'a' && line[0] <= 'f'))) {
			if (rss > 32768) fprintf(stderr, "#   %ld MB resident (%ld MB in huge pages; %s): %s", rss >> 10, ahp >> 10, flags, head);
			snprintf(head, sizeof head, "%s", line);
			rss = ahp = 0;
			flags[0] = 0;
		} else if (!strncmp(line, "Rss:", 4)) rss = atol(line + 4);
		else if (!strncmp(line, "AnonHugePages:", 14)) ahp = atol(line + 14);
		else if (!strncmp(line, "VmFlags:", 8)) { snprintf(flags, sizeof flags, "%.100s", line + 9); flags[strcspn(flags, "\n")] = 0; }
	}
	if (rss > 32768) fprintf(stderr, "#   %ld MB resident (%ld MB in huge pages; %s): %s", rss >> 10, ahp >> 10, flags, head);
	fclose(f);
}

void fast_exit(void) {
	if (getenv("MSX_TIMING")) {
		/* what the stage timers do not see: from exec to main (loader, static initialisers) and, after this line,
		 * the kernel taking the address space apart (mapped input, pinned buffers) */
		struct timespec ts;
		double cpu = 0;
		if (clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts) == 0) cpu = (double)ts.tv_sec + ts.tv_nsec * 1e-9;
		fprintf(stderr, "# process: %.3f s from main to exit, %.3f s of CPU time in all threads\n", now_s() - g_t_main, cpu);
		{   /* what the kernel will have to take apart */
			FILE *f = fopen("/proc/self/status", "r");
			char line[256];
			if (f) {
				fprintf(stderr, "# process memory at exit:");
				while (fgets(line, sizeof line, f))
					if (!strncmp(line, "VmHWM", 5) || !strncmp(line, "VmRSS", 5) || !strncmp(line, "RssAnon", 7) || !strncmp(line, "RssFile", 7) ||
					    !strncmp(line, "RssShmem", 8) || !strncmp(line, "VmPTE", 5) || !strncmp(line, "Threads", 7)) {
						line[strcspn(line, "\n")] = 0;
						for (char *q = line; *q; q++) if (*q == '\t') *q = ' ';
						fprintf(stderr, " %s;", line);
					}
				fprintf(stderr, "\n");
				fclose(f);
			}
			if (atoi(getenv("MSX_TIMING")) >= 2) mem_report("exit");
		}
	}
	if (getenv("MSX_CLEAN_EXIT")) { runtime_warmup_join(); return; }
	/* MSX_GUARD=1 (tests): nothing is freed on this way out, so the guard bytes around the library's device allocations are
	 * looked at here (include/msamtools_amd.h: msx_debug_guard_check) */
	if (getenv("MSX_GUARD") && msx_debug_guard_check() > 0) { fflush(NULL); if (msh_exit_hook) msh_exit_hook(70); _exit(70); }
	fflush(stdout);
	fflush(stderr);
	if (msh_exit_hook) msh_exit_hook(0);
	_exit(0);
}

/* several GPUs: one process per GPU, started with RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT in
 * the environment (torchrun, mpirun wrappers, a shell loop).  `profile` then reads ITS shard of the sample --
 * "{rank}" in the input path is replaced by the rank; shards are cut at QNAME boundaries, e.g. by splitting the
 * name-grouped BAM -- and the ranks exchange counts and, per sharing iteration, the increment vector over RCCL
 * (msx_profile_finalize_dist_enqueue); rank 0 writes the profile. */
int dist_world(void) { const char *e = getenv("WORLD_SIZE"); int v = e ? atoi(e) : 1; return v > 1 ? v : 1; }
int dist_rank(void) { const char *e = getenv("RANK"); return e ? atoi(e) : 0; }
/* Running as one rank of several is something the caller asks for -- "{rank}" in the input path, or MSX_DIST=1 --
 * never something inferred from a WORLD_SIZE that happens to be in the environment (a SLURM job, a torchrun
 * parent): a rank that read the WHOLE file would have its counts multiplied by the number of ranks. */
int g_dist;

/* The HIP runtime starts up (60-90 ms) on a thread of its own from the moment the options are known to be good, beside
 * the opening of the input and the parsing of its header (a million @SQ lines: 70 ms) -- the device threads' msx_ctx_create
 * then finds it up.  Nothing on the way THROUGH the command waits for this thread: a failure is reported by the
 * msx_ctx_create that follows, and the usual way out is _exit (mDie, fast_exit), which does not run the runtime's exit
 * handlers under it.  Where main() RETURNS instead (MSX_CLEAN_EXIT=1; `coverage`), the thread is joined first
 * (runtime_warmup_join): the runtime's exit handlers unload the code objects, and a warm-up still asking for a kernel's
 * attributes at that moment aborts the process ("Cannot find Symbol ...": seen on a small `profile` under MSX_CLEAN_EXIT).
 * MSX_NO_WARMUP=1: as before. */
static pthread_t g_warm_th;
static int g_warm_started;
static void *warmup_thread(void *arg) {
	(void)msx_runtime_warmup((int)(intptr_t)arg);
	return NULL;
}
void runtime_warmup_start(void) {
	int ids[MSH_MAX_DEVICES];
	if (getenv("MSX_NO_WARMUP") || g_warm_started) return;
	(void)device_list(ids);
	if (pthread_create(&g_warm_th, NULL, warmup_thread, (void *)(intptr_t)ids[0]) == 0) g_warm_started = 1;
}
void runtime_warmup_join(void) {
	if (!g_warm_started) return;
	g_warm_started = 0;
	pthread_join(g_warm_th, NULL);
}

void ctx_open_dev(int id) {
	if (msx_ctx_create(&g_ctx, id) != MSX_OK) mDie("%s", msx_last_error(NULL));
	/* (the command line's batches are a million records: the context's side lanes cost it more than they overlap) */
	{ const char *e = getenv("MSX_SERIAL"); if (!(e && atoi(e) == 0)) (void)msx_ctx_set_lanes(g_ctx, 0); }     /* (MSX_SERIAL=0: keep them) */
}
void ctx_open(void) {
	const char *dev = getenv("MSX_DEVICE"), *lr = getenv("LOCAL_RANK");
	ctx_open_dev(dev ? atoi(dev) : (g_dist && lr ? atoi(lr) : 0));
}
int device_list(int *ids) {
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

size_t batch_target(void) {
	const char *e = getenv("MSX_BATCH_RECORDS");
	long n = e ? strtol(e, NULL, 10) : (1L << 21);
	return n < 1 ? 1 : (size_t)n;
}

/* stringify_argv() as used by mBuildCommandLine (msam_helper.c:59-76) */
char *command_line(int argc, char *argv[]) {
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

void rb_reserve(rbatch *b) {
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

void rb_clear(rbatch *b) {
	b->n = 0;
	b->blob.l = 0;
	b->n_groups = 0;
	rb_reserve(b);
	b->rec_off[0] = 0;
	b->cigar_off[0] = 0;
	b->md_off[0] = 0;
}

void rb_mark_group(rbatch *b) {   /* a pool starts at the record about to be appended */
	if (b->n_groups + 2 > b->group_cap) {
		b->group_cap = b->group_cap ? b->group_cap * 2 : 65536;
		b->group_off = (uint32_t *)realloc(b->group_off, b->group_cap * 4);
		if (!b->group_off) mDie("Out of memory");
	}
	b->group_off[b->n_groups++] = (uint32_t)b->n;
}

/* one pass over the aux block for MD, NM, AS (first occurrence wins, as bam_aux_get) */
void rb_append(rbatch *b, const uint8_t *r, size_t len, int want_stats) {
	size_t i = b->n;
	uint32_t nc = (msh_rec_check(r, len), REC_NCIGAR(r));
	const uint8_t *cigw = msh_real_cigar(r, len, &nc, NULL);       /* (a long CIGAR kept in CG:B:I) */
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
		memcpy(b->cigar + b->cigar_off[i], cigw, 4 * (size_t)nc);
		if (ml) memcpy(b->md + b->md_off[i], md + 1, ml);
		b->cigar_off[i + 1] = b->cigar_off[i] + nc;
		b->md_off[i + 1] = b->md_off[i] + (uint32_t)ml;
	} else {
		b->cigar_off[i + 1] = b->cigar_off[i];
		b->md_off[i + 1] = b->md_off[i];
	}
	b->n++;
}

void rb_host_view(rbatch *b, msx_batch *h, int with_groups) {
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

void qn_format(const qn_result *r, char *buf, size_t n) {
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

qn_result qn_check(const msh_hdr *hdr, const rbatch *first) {
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
