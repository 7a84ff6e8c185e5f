/*
 * msh_filter.c -- `msamtools filter` (msam_filter.c:304-497): options and their validation messages, the device stage and
 * the writer stage over the pipeline, --rescore, --profile-out.
 */
#include "msh_cli.h"

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

/* ---- filter over the pipeline ----------------------------------------------------------------------- */
typedef struct fshared fshared;
typedef struct {
	fshared *S;
	int dev_id, index;
	msx_ctx *ctx;
	msx_profile *prof;          /* filter --profile-out: this device's part of the sample */
	pthread_t th;
	double t_ctx, t_upload, t_gpu, t_fetch, t_wait;
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
	int dev_overlap;            /* -b: the encoder on a stream of its own beside the next batch (msx_unpack_emit_bgzf_enqueue / _complete) */
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
 * written -- dies with the reference's message.  A record without AS is met when its pool is written
 * (msam_filter.c:219-221), and a paired pool is written READ1 pass first (:247-263): the batch is then cut BEHIND that pool
 * and filtered with msx_filter_params.fatal_pool_partial, which keeps of it what the reference had written -- the READ1
 * winners when only the READ2 pass meets such a record (fatal_refilter).
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

/* Plain filters carry no pools in their batches.  The reference holds the records it has kept since its last name change
 * in an open pool -- a record's name is compared with the last MAPPED record's (msam_filter.c:120-125,170), unmapped ones do
 * not move that name on (:132-138) -- and dies with them unwritten.  Where that pool begins, from the batch's records up to
 * the offending one (BAM records with their length words, one after the other).  (A pool that was opened in the batch
 * before cannot be taken back: those records are out.) */
static int64_t open_pool_start(const uint8_t *recs, int64_t err) {
	const uint8_t *p = recs, *prev = NULL;
	int64_t i, start = 0;
	for (i = 0; i <= err; i++) {
		const uint32_t len = (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
		const uint8_t *r = p + 4;
		if (prev && (r[8] != prev[8] || memcmp(r + 32, prev + 32, r[8]) != 0)) start = i;      /* written before record i is looked at */
		if (!(r[14] & 4)) prev = r;
		p += 4 + (size_t)len;
	}
	return start;
}

/* the batch once more, cut where the reference died (see fatal_prefix); db / fo as the first pass left them.  recs: the
 * batch's records on the host, or NULL with the unpacker that holds them on the device.  Leaves the number of records to
 * write in st->n_emit. */
static void fatal_refilter(const msx_filter_params *fp, int no_as, int has_pools, const uint32_t *group_off, int64_t n_groups,
                           const uint8_t *recs, msx_unpack *unpack, msx_batch *db, msx_filter_out *fo, msx_filter_status *st) {
	const int64_t err = st->err_record;
	msx_filter_params cut = *fp;
	int64_t g = 0, npre = err;
	st->n_emit = 0;
	if (err < 0) return;
	if (has_pools && n_groups > 0) {
		npre = fatal_prefix(group_off, n_groups, err, &g);
		if (no_as) {                           /* the offending pool stays in, judged as far as the reference got */
			npre = (int64_t)group_off[g + 1];
			g = g + 1;
			cut.fatal_pool_partial = 1;
		}
	} else if (recs) {
		npre = open_pool_start(recs, err);
	} else if (unpack) {                       /* the records in front of it, down from the device (a dying command can afford it) */
		int32_t *idx = (int32_t *)xmalloc(((size_t)err + 1) * 4);
		uint8_t *buf;
		int64_t i, nb = 0;
		for (i = 0; i <= err; i++) idx[i] = (int32_t)i;
		MSX(msx_host_to_dev(g_ctx, fo->emit_idx, idx, ((size_t)err + 1) * 4));
		MSX(msx_unpack_emit_gather(g_ctx, unpack, fo->emit_idx, err + 1, &nb));
		buf = (uint8_t *)xmalloc((size_t)nb + 64);
		MSX(msx_unpack_emit_fetch(g_ctx, unpack, buf, (size_t)nb + 64, NULL));
		npre = open_pool_start(buf, err);
		free(buf);
		free(idx);
	}
	if (npre <= 0) return;
	db->n_records = npre; db->n_groups = g;
	MSX(msx_filter_enqueue(g_ctx, db, &cut, fo));
	if (msx_filter_finish(g_ctx, st) != MSX_OK && !cut.fatal_pool_partial) st->n_emit = 0;
}

/* the gathered output of a device-unpacked batch (records, or finished BGZF blocks) on its way to the host: an output buffer
 * of the pool -- a larger one if this batch keeps more than any before it --, the copy on the unpacker's copy stream, the
 * event the writer waits for */
static void filter_emit_fetch(fdev_t *D, msx_unpack *unpack, pslot *s, int64_t nb) {
	pipe_t *P = D->S->P;
	if (nb > 0) {
		s->ob = ob_acquire(P, s->seq);           /* (the buffer of this batch's number: waits for the writer to have written the batch PIPE_OBUFS before it) */
		if ((size_t)nb + 64 > P->ob_cap[s->ob]) {          /* (the old one stays mapped and page-locked: rare) */
			P->ob_cap[s->ob] = (size_t)nb + (size_t)nb / 4 + ((size_t)4 << 20);
			P->ob[s->ob] = io_alloc(P->ob_cap[s->ob]);
			io_populate(P->ob[s->ob], P->ob_cap[s->ob]);
			MSX(msx_host_register(g_ctx, P->ob[s->ob], P->ob_cap[s->ob]));
		}
		s->obuf = P->ob[s->ob];
		s->ocap = P->ob_cap[s->ob];
	}
	/* (an event belongs to the device it was made on: one per slot and context) */
	if (!s->ev_out_by[D->index]) MSX(msx_event_create(g_ctx, &s->ev_out_by[D->index]));
	s->ev_out = s->ev_out_by[D->index];
	MSX(msx_unpack_emit_fetch(g_ctx, unpack, s->obuf, s->ocap, s->ev_out));
	s->ev_ctx = g_ctx;
	s->olen = (size_t)nb;
}
/* (-b, the encoder beside the next batch) the oldest batch handed to the encoder: wait for its blocks, send them down, pass
 * the batch on to the writer */
static void filter_emit_down(fdev_t *D, msx_unpack *unpack, int si) {
	pipe_t *P = D->S->P;
	const double t0 = now_s();
	int64_t nb = 0;
	MSX(msx_unpack_emit_bgzf_complete(g_ctx, unpack, &nb, NULL));
	filter_emit_fetch(D, unpack, &P->slot[si], nb);
	D->t_fetch += now_s() - t0;
	pq_push(&P->q_out, si);
}

void *filter_dev_thread(void *arg) {
	fdev_t *D = (fdev_t *)arg;
	fshared *F = D->S;
	pipe_t *P = F->P;
	msx_stage *stage = NULL;
	msx_unpack *unpack = NULL;
	int pending = PQ_NONE;             /* a slot taken off the queue ahead of its turn (its bytes are being sent up) */
	int held = -1;                     /* (-b) the batch whose records the encoder is working on */
	ahead_q ahead = {{PQ_NONE, PQ_NONE}, 0, 0};
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
		{ const char *e = getenv("MSX_SERIAL"); if (!(e && atoi(e) == 0)) (void)msx_ctx_set_lanes(g_ctx, 0); }     /* (batches of a million records: msh_common.c ctx_open_dev) */
		MSX(msx_stage_create(g_ctx, &stage));
		if (F->po)
			MSX(msx_profile_create(g_ctx, &D->prof, F->pf->n_features, F->po->share_type, F->pf->fmap, P->hdr->n_targets));
		D->t_ctx = now_s() - t0;
		D->t_ctx_end = now_s();
		if (P->raw_mode) { MSX(msx_unpack_create(g_ctx, &unpack)); pin_start(P, 1); }
	}
	for (;;) {
		double t0 = now_s(), t1;
		int si;
		pslot *s;
		rbatch *b;
		msx_batch hb, db;
		msx_filter_out fo;
		msx_filter_status st;
		if (pending != PQ_NONE) si = pending;
		else if (ahead.n) si = ahead_pop(&ahead);
		else if (held < 0) si = pq_pop(&P->q_dev);
		else if ((si = pq_try_pop(&P->q_dev)) == PQ_NONE) {
			/* nothing to work on beside the encoder: its batch goes on now.  (Waiting here with a batch in hand would also
			 * stall several contexts for good: the writer wants that batch first, and the decoder has no slot left to
			 * make the batch this thread waits for.) */
			filter_emit_down(D, unpack, held);
			held = -1;
			si = pq_pop(&P->q_dev);
		}
		pending = PQ_NONE;
		t1 = now_s();
		D->t_wait += t1 - t0;
		if (si == PQ_END) break;
		s = &P->slot[si];
		b = &s->b;
		s->fatal = 0;
		MSH_TRACE("device thread %d takes batch %zu (slot %d)", D->index, s->seq, si);
		if (s->seq <= 2) mem_report(s->seq == 0 ? "device thread takes batch 0" : s->seq == 1 ? "device thread takes batch 1" : "device thread takes batch 2");
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
			{
				/* (MSX_TIMING=2: the first device-walked batches step by step -- what the first of them pays beyond the others) */
				const int tt = s->seq <= 3 && getenv("MSX_TIMING") && atoi(getenv("MSX_TIMING")) >= 2;
				double ta = now_s(), tb, tc, td;
				unpack_slot_enqueue(P, s, unpack, &up);
				tb = now_s();
				if (F->n_dev == 1) unpack_slots_ahead(P, unpack, &ahead);
				tc = now_s();
				unpack_slot_finish(P, s, unpack, &up, &ur, &db);
				td = now_s();
				if (F->n_dev == 1) unpack_slots_ahead(P, unpack, &ahead);        /* (not decoded yet a moment ago?) */
				if (tt) fprintf(stderr, "# batch %zu on the device thread: pin+seed %.1f ms, enqueue %.1f, send ahead %.1f, finish %.1f, send ahead %.1f\n",
				                s->seq, (ta - t1) * 1e3, (tb - ta) * 1e3, (tc - tb) * 1e3, (td - tc) * 1e3, (now_s() - td) * 1e3);
			}
			D->t_upload += now_s() - t1; t1 = now_s();
			b->n = (size_t)ur.n_records;
			s->n_emit = 0;
			s->olen = 0;
			if (ur.n_records > 0) {
				MSX(msx_stage_outputs(g_ctx, stage, ur.n_records, 0, &fo));
				if (D->prof) MSX(msx_filter_profile_enqueue(g_ctx, &db, F->fp, &fo, D->prof));
				else MSX(msx_filter_enqueue(g_ctx, &db, F->fp, &fo));
				int frc;
				if ((frc = msx_filter_finish(g_ctx, &st)) != MSX_OK) {
					uint32_t *go = NULL;
					s->fatal = 1;
					F->any_fatal = 1;
					snprintf(s->fatal_msg, sizeof s->fatal_msg, "%s", msx_last_error(g_ctx));
					if (P->mode != 0 && ur.n_groups > 0 && st.err_record >= 0) {
						go = (uint32_t *)xmalloc(((size_t)ur.n_groups + 1) * 4);
						MSX(msx_dev_to_host(g_ctx, go, db.group_off, ((size_t)ur.n_groups + 1) * 4));
					}
					fatal_refilter(F->fp, frc == MSX_ERR_NO_AS, P->mode != 0, go, ur.n_groups, NULL, unpack, &db, &fo, &st);
					free(go);
				}
				if (s->seq <= 3 && getenv("MSX_TIMING") && atoi(getenv("MSX_TIMING")) >= 2)
					fprintf(stderr, "# batch %zu on the device thread: outputs + filter + its wait %.1f ms\n", s->seq, (now_s() - t1) * 1e3);
				D->t_gpu += now_s() - t1; t1 = now_s();
				s->n_emit = st.n_emit;
				if (F->n_dev == 1) unpack_slots_ahead(P, unpack, &ahead);
				/* gather on the device, make room here if this batch keeps more than any before it, and let the bytes travel
				 * while the next batch is worked on: the writer waits for s->ev_out */
				s->framed = F->dev_frame;
				if (F->dev_overlap) {
					/* -b: the encoder runs on a stream of its own beside the NEXT batch's walk and filter.  This batch is
					 * handed to it; the batch before, whose blocks are done by now, goes down and on to the writer. */
					MSX(msx_unpack_emit_bgzf_enqueue(g_ctx, unpack, fo.emit_idx, st.n_emit, F->dev_level));
					D->t_fetch += now_s() - t1;
					if (held >= 0) filter_emit_down(D, unpack, held);
					held = si;
					if (D->n_end < 64) D->t_end[D->n_end++] = now_s();
					continue;
				}
				if (F->dev_frame) MSX(msx_unpack_emit_gather_bgzf(g_ctx, unpack, fo.emit_idx, st.n_emit, F->dev_level, &nb, NULL));
				else MSX(msx_unpack_emit_gather(g_ctx, unpack, fo.emit_idx, st.n_emit, &nb));
				filter_emit_fetch(D, unpack, s, nb);
				D->t_fetch += now_s() - t1;
			}
			if (D->n_end < 64) D->t_end[D->n_end++] = now_s();
			MSH_TRACE("device thread %d hands batch %zu to the writer (%zu bytes)", D->index, s->seq, s->olen);
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
		{
			int frc;
			if ((frc = msx_filter_finish(g_ctx, &st)) != MSX_OK) {          /* the reference's own mDie texts; see fatal_prefix */
				s->fatal = 1;
				F->any_fatal = 1;
				snprintf(s->fatal_msg, sizeof s->fatal_msg, "%s", msx_last_error(g_ctx));
				fatal_refilter(F->fp, frc == MSX_ERR_NO_AS, P->mode != 0, b->group_off, (int64_t)b->n_groups, b->base + b->rec_off[0], NULL, &db, &fo, &st);
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
	if (held >= 0) filter_emit_down(D, unpack, held);
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
static int pq_pop_seq(pq *q, const pipe_t *P, size_t seq, msh_out *out) {
	int i, v = PQ_END - 1, flushed = 0;
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
		if (out && !flushed) {
			/* the writer has to wait for its next batch (a producer that trickles): what earlier batches left in the
			 * output's buffers -- an open BGZF block, stdio's share of the text -- goes out first (msh_out_flush) */
			flushed = 1;
			pthread_mutex_unlock(&q->mu);
			msh_out_flush(out);
			pthread_mutex_lock(&q->mu);
			continue;
		}
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
	mem_report("header read");
	pipe_init(&P, in, pools ? 1 : (po ? 3 : 0), want_stats, F.n_dev);
	mem_report("pipeline initialised");
	P.unmapped_visible = unmapped_written;
	P.cut_mapped = pools && po;
	/* From the second batch on the record walk runs on the device (msx_unpack): one context, records written as they
	 * are (no --rescore), BAM out.  The first batch takes the host-side walk: the preflight reads its records.
	 * MSX_HOST_UNPACK=1 keeps every batch on the host. */
	/* Several contexts (MSX_DEVICES) keep both: every context inflates and walks its batches, the stream's carry travels
	 * from the context that finished batch k to the one that walks batch k + 1 (unpack_slot_enqueue; MSX_MULTI_HOST_WALK=1:
	 * round 3's form, inflate and walk on the host for every batch). */
	if ((F.n_dev == 1 || !getenv("MSX_MULTI_HOST_WALK")) && !fp->rescore && (out_mode == MSH_OUT_BAM || out_mode == MSH_OUT_UBAM) &&
	    !getenv("MSX_HOST_UNPACK"))
		pipe_enable_raw(&P, out_mode == MSH_OUT_BAM ? 2 : 1);      /* (2: deflated output -- the batches may grow, msh_pipeline.c) */
	if (fp->rescore)
		for (k = 0; k < P.n_slots; k++) P.slot[k].as_out = (int32_t *)xmalloc((P.cap_rec + 8) * 4);
	if (po) prof_features(po, P.hdr, &pf);
	/* the BGZF layer of the output on the device: stored blocks for -bu, DEFLATE for -b (MSX_HOST_FRAME=1: frame / zlib-deflate
	 * on the host cores, as round 3 did; MSX_HOST_DEFLATE=1: only -b's deflate) */
	F.dev_frame = (out_mode == MSH_OUT_UBAM || (out_mode == MSH_OUT_BAM && !getenv("MSX_HOST_DEFLATE"))) && !getenv("MSX_HOST_FRAME");
	F.dev_level = out_mode == MSH_OUT_UBAM ? 0 : 6;           /* ("wb": htslib's default level; MSX_BGZF_LEVEL=1..9 as for the host's zlib) */
	if (F.dev_level && getenv("MSX_BGZF_LEVEL") && atoi(getenv("MSX_BGZF_LEVEL")) >= 1 && atoi(getenv("MSX_BGZF_LEVEL")) <= 9)
		F.dev_level = atoi(getenv("MSX_BGZF_LEVEL"));
	F.dev_overlap = F.dev_frame && F.dev_level > 0 && !getenv("MSX_DEFLATE_SYNC");     /* (MSX_DEFLATE_SYNC=1: one batch after the other) */
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
		const int si = pq_pop_seq(&P.q_out, &P, seq, seq > 0 ? F.out : NULL);     /* (behind batch 0: the output is open) */
		pslot *s;
		MSH_TRACE("writer has batch %zu (slot %d)", seq, si);
		t1 = now_s();
		t_wait += t1 - t0;
		if (si == PQ_END) break;
		s = &P.slot[si];
		if (s->raw) {
			if (s->ev_out && s->olen) { if (msx_event_wait(s->ev_ctx, s->ev_out) != MSX_OK) mDie("%s", msx_last_error(s->ev_ctx)); }
			if (s->framed) msh_write_framed(F.out, s->obuf, s->olen);
			else msh_write_stream(F.out, s->obuf, s->olen);
			if (s->ob >= 0) { ob_release(&P, s->ob); s->ob = -1; }
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
		ob_written(&P, seq);
		n_batches++;
		seq++;
		n_in += s->b.n;
		n_out += (size_t)s->n_emit;
		tw += now_s() - t1;
		pq_push(&P.q_free, si);
	}
	t_tail[0] = now_s();
	mem_report("writer done");
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
		fprintf(stderr, "# batches: %zu (%zu sent ahead)%s\n", n_batches, P.comp_mode ? P.n_ahead : (size_t)0,
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
	{   /* MSX_CLEAN_EXIT=1: an orderly shutdown (leak checks), timed piece by piece under MSX_TIMING */
		double tq = now_s();
		for (k = 0; k < F.n_dev; k++) {
			if (F.dev[k].prof) msx_profile_destroy(F.dev[k].ctx, F.dev[k].prof);
			if (getenv("MSX_TIMING")) { fprintf(stderr, "# clean exit: profile destroyed after %.3f s\n", now_s() - tq); tq = now_s(); }
			msx_ctx_destroy(F.dev[k].ctx);
			if (getenv("MSX_TIMING")) { fprintf(stderr, "# clean exit: context destroyed after %.3f s\n", now_s() - tq); tq = now_s(); }
		}
		msh_close(in);
		if (getenv("MSX_TIMING")) fprintf(stderr, "# clean exit: input closed after %.3f s; main returns %.3f s after it began\n", now_s() - tq, now_s() - g_t_main);
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
	runtime_warmup_start();
	rd.in = msh_open(infile);
	hdr = msh_header(rd.in);

	if (tee) prof_opts_derive(&po);
	bulk = msh_is_bam(rd.in);
	if (!getenv("MSX_SERIAL_IO")) {
		/* BAM or SAM text in: decode | device | encode as three overlapping stages */
		/* (under MSX_CLEAN_EXIT it closes the input itself, timed; otherwise the process has ended in there) */
		return filter_pipelined(rd.in, &fp, pools, want_stats, mode, argc, argv, tee ? &po : NULL);
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
