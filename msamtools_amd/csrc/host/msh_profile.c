/*
 * msh_profile.c -- `msamtools profile` (msam_profile.c:554-990): options, --genome features, the device stage over the
 * pipeline, post-processing and the text report (mMatrix.c:359-376), shared with `filter --profile-out`.
 */
#include "msh_cli.h"
#include "msh_fmt.h"

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

/* msam_profile.c:712-755: --multi and --unit by prefix match, length normalisation */
void prof_opts_derive(prof_opts *o) {
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

void prof_features(const prof_opts *o, const msh_hdr *hdr, prof_feat *F) {
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
void gz_member(const kstr *in, kstr *out) {
	z_stream zs;
	size_t bound;
	memset(&zs, 0, sizeof zs);
	/* level 4 rather than gzip's 6: the million-line profile of the bench compresses in 25 ms instead of 59 per thread and
	 * comes out 5 % larger (7.1 MB instead of 6.8); MSX_GZ_LEVEL=6 for the reference's level */
	static int level_once = 0;       /* (the workers all find the same value) */
	int level = __atomic_load_n(&level_once, __ATOMIC_RELAXED);
	if (!level) { const char *e = getenv("MSX_GZ_LEVEL"); level = e && atoi(e) >= 1 && atoi(e) <= 9 ? atoi(e) : 4; __atomic_store_n(&level_once, level, __ATOMIC_RELAXED); }
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
	/* "%s\t%.8g\n" per feature (mMatrix.c:359-376), the number by msh_fmt_g8: printf's bytes at a third of its time */
	for (i = lo; i < hi; i++) {
		const char *nm = J->F->name[i];
		const size_t nl = strlen(nm);
		ks_reserve(k, nl + 40);
		memcpy(k->s + k->l, nm, nl);
		k->l += nl;
		k->s[k->l++] = '\t';
		k->l += (size_t)msh_fmt_g8(J->row[1 + i], k->s + k->l);
		k->s[k->l++] = '\n';
		k->s[k->l] = 0;
	}
	gz_member(k, &J->gz[tid]);
}

void fd_write_all(int fd, const void *p, size_t n) {
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
void profile_report(const prof_opts *o, const prof_feat *F, const msx_profile_stats *st, double *row,
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
		msh_fmt_init();
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
void profile_combine_and_finalize(msx_ctx **ctx, msx_profile **prof, int n_dev, int share_type, double *row,
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
		runtime_warmup_start();
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
		if ((S.n_dev == 1 || !getenv("MSX_MULTI_HOST_WALK")) && !getenv("MSX_HOST_UNPACK")) pipe_enable_raw(&P, 0);   /* the record walk of every batch but the first on the device */
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
