/*
 * msh_coverage.c -- `msamtools coverage` (msam_coverage.c:225-380) over the pipeline.
 */
#include "msh_cli.h"

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
	int64_t n_streamed = 0;

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

	runtime_warmup_start();
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
	/* (the depth array is not zeroed here: msx_coverage_collect keeps the batches' pieces on the device and
	 *  msx_coverage_collect_finish writes every cell once -- for what goes the streamed way the library zeroes it when the first
	 *  such batch comes; MSX_COV_STREAMED=1: zero, marks, prefix sum for every batch, as in rounds 2-4) */
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
					MSX(msx_coverage_collect(g_ctx, &db, (const int64_t *)d_off, hdr->n_targets, total, (int32_t *)d_cov,
					                            (uint8_t *)d_covered));
			} else if (s->b.n > 0) {
				rb_host_view(&s->b, &hb, 0);
				hb.md_off = NULL; hb.md = NULL; hb.nm = NULL; hb.as = NULL;
				MSX(msx_batch_upload(g_ctx, &hb, &db));
				MSX(msx_coverage_collect(g_ctx, &db, (const int64_t *)d_off, hdr->n_targets, total, (int32_t *)d_cov,
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
			MSX(msx_coverage_collect(g_ctx, &db, (const int64_t *)d_off, hdr->n_targets, total, (int32_t *)d_cov,
			                            (uint8_t *)d_covered));
			MSX(msx_ctx_sync(g_ctx));
			msx_batch_free(g_ctx, &db);
		}
		if (rd.done) break;
	}
	t_cov[1] = now_s();
	MSX(msx_coverage_collect_finish(g_ctx, (int32_t *)d_cov, total, &n_streamed));
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
		fprintf(stderr, "# coverage: input and device set up %.3f s, records through %.3f, depths from the pieces %.3f, depths to the host %.3f, "
		        "report %.3f (%lld cells; %lld batches piled up the streamed way)\n", t_cov[0] - t_cov0, t_cov[1] - t_cov[0],
		        t_cov[2] - t_cov[1], t_cov[3] - t_cov[2], now_s() - t_cov[3], (long long)total, (long long)n_streamed);
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
