/*
 * msh_summary.c -- `msamtools summary` (msam_summary.c): one line of alignment statistics per primary mapped record, the
 * distribution of one of them (--stats), or the number of QNAME groups among the mapped records (--count).
 *
 * The statistics are those of bam_get_extended_summary (mBamVector.c:135-236).  They follow from what the device's
 * statistics kernel computes for `filter` (msx_aln_stats: length, query_length, query_clip, edit on the MD path:
 * mBamVector.c:40-133) when every record is sent along the MD path -- a record without MD with an empty MD string,
 * its NM not looked at, as the reference's summary never reads NM --:
 *     match        = length - edit                (M/=/X bases minus the MD mismatches; edit = mismatches + I + D bases)
 *     edit (here)  = edit + query_clip            (mismatch + qclip + gapopen + gapextend, :228: a gap of w bases opens once
 *                                                  and extends w - 1 times)
 *     glocal_len   = length + query_clip          (msam_summary.c:70)
 * What the device does not hand back is the record's span on the reference (bam_endpos, for --edge): the host takes it
 * from the CIGAR words the batch holds anyway.  The text is the host's, as everywhere.
 */
#include "msh_cli.h"

static void summary_help(FILE *out) {
	fprintf(out,
	        "Usage:\n------\n\n%s summary [-Sc] <bamfile> [--help] [-e <num>] [--stats=<string>]\n"
	        "\nGeneral options:\n----------------\n\n"
	        "These options specify the input/output formats of BAM/SAM files \n(same meaning as in 'samtools view'):\n"
	        "  -S                        input is SAM (default: false)\n"
	        "  <bamfile>                 input SAM/BAM file\n"
	        "  --help                    print this help and exit\n\n"
	        "Specific options:\n-----------------\n\n"
	        "  -e, --edge=<num>          ignore alignment if reads map to <num> bases at the edge of target sequence (default: 0)\n"
	        "  -c, --count               count number of inserts/QNAME groups in BAM file (default: false)\n"
	        "  --stats=<string>          {mapped|unmapped|edit|score} only report readcount distribution for specified stats, not read-level stats (default: none)\n\n"
	        "Description\n-----------\n"
	        "Prints summary of alignments in the given BAM/SAM file. By default, it prints\n"
	        "a summary line per alignment entry in the file. The summary is a tab-delimited\n"
	        "line with the following fields:\n"
	        "\tqname,aligned_qlen,target_name,glocal_align_len,matches,percent_identity\n"
	        "glocal_align_len includes the unaligned qlen mimicing a global alignment \n"
	        "in the query and local alignment in target, thus glocal.\n\n"
	        "With --stats option, summary is consolidated as distribution of read counts\n"
	        "for a given measure. \n"
	        "   --stats=mapped   - distribution for number of mapped query bases\n"
	        "   --stats=unmapped - distribution for number of unmapped query bases\n"
	        "   --stats=edit     - distribution for edit distances\n"
	        "   --stats=score    - distribution for score=match-edit\n",
	        PROGRAM);
}

#define SUMMARY_MAX_READ_LENGTH 4096          /* M_BAM_MAX_READ_LENGTH, mBamVector.h:28 */

/* the end of the record on the reference as htslib's bam_endpos gives it: pos + the bases its CIGAR consumes there
 * (M, D, N, =, X), 1 when that is 0 */
static int64_t record_endpos(const rbatch *b, size_t i) {
	int64_t rlen = 0;
	uint32_t k;
	if (!(b->flag[i] & 4))
		for (k = b->cigar_off[i]; k < b->cigar_off[i + 1]; k++) {
			const uint32_t op = b->cigar[k] & 0xf;
			if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += b->cigar[k] >> 4;
		}
	return (int64_t)b->pos[i] + (rlen ? rlen : 1);
}

int msam_summary_main(int argc, char *argv[]) {
	static const struct option lopts[] = {{"help", no_argument, 0, 1000},  {"edge", required_argument, 0, 'e'},
	                                      {"count", no_argument, 0, 'c'},  {"stats", required_argument, 0, 1001},
	                                      {0, 0, 0, 0}};
	int o_help = 0, o_S = 0, o_c = 0, n_e = 0, n_stats = 0, nerrors = 0, c, stat_mode = -1;
	long v_e = 0;
	const char *o_stats = NULL;
	uint32_t edge = 0;
	msh_in *in;
	const msh_hdr *hdr;
	reader rd;
	rbatch b;
	kstr rec = {0, 0, 0};
	size_t target = batch_target();

	(void)o_S;
	opterr = 0;
	optind = 1;
	while ((c = getopt_long(argc, argv, "Sce:", lopts, NULL)) != -1) {
		switch (c) {
		case 'S': o_S++; break;
		case 'c': o_c++; break;
		case 'e': {
			char *end;
			v_e = strtol(optarg, &end, 10);
			if (*end || end == optarg) { fprintf(stderr, "%s: invalid argument \"%s\" to option -e|--edge=<num>\n", PROGRAM, optarg); nerrors++; }
			n_e++;
			break;
		}
		case 1000: o_help++; break;
		case 1001: o_stats = optarg; n_stats++; break;
		default:
			fprintf(stderr, "%s: invalid option \"%s\"\n", PROGRAM, argv[optind - 1]);
			nerrors++;
		}
	}
	if (o_help > 0 || argc < 2) { summary_help(stdout); exit(EXIT_SUCCESS); }       /* msam_summary.c:205-208 */
	if (argc - optind < 1) { fprintf(stderr, "%s: missing option <bamfile>\n", PROGRAM); nerrors++; }
	if (nerrors > 0) {                                                              /* :211-215 */
		fprintf(stderr, "Use --help for usage instructions!\n");
		mQuit("");
	}
	if (n_e > 0) {                                                                  /* :222-230 */
		if (v_e < 0) { fprintf(stdout, "-e must be a positive integer\n"); summary_help(stdout); mQuit(""); }
		edge = (uint32_t)v_e;
	}
	if (n_stats > 0 && o_c > 0) { fprintf(stdout, "--stats cannot be combined with --count\n"); summary_help(stdout); mQuit(""); }
	if (n_e > 0 && o_c > 0) { fprintf(stdout, "-e cannot be combined with --count\n"); summary_help(stdout); mQuit(""); }

	if (n_stats > 0) {                                                              /* :254-268, after the input is open there */
		static const char *modes[] = {"mapped", "unmapped", "edit", "score"};
		int i;
		for (i = 0; i < 4; i++)
			if (strcmp(o_stats, modes[i]) == 0) stat_mode = i;
	}
	if (!o_c) runtime_warmup_start();
	in = msh_open(argv[optind]);
	hdr = msh_header(in);
	if (n_stats > 0 && stat_mode < 0) mDie("Do not understand %s as mode", o_stats);

	if (o_c) {
		/* mCountInserts (:19-40): mapped records whose QNAME is not the previous mapped record's.  Names and flags only:
		 * nothing for the device to do. */
		char prev[256] = "";
		long count = 0;
		while (msh_read(in, &rec) == 0) {
			const uint8_t *r = (const uint8_t *)rec.s;
			const uint16_t flag = (uint16_t)(r[14] | r[15] << 8);
			const char *name = (const char *)r + 32;
			if (flag & 4) continue;
			if (strcmp(name, prev) != 0) count++;
			strncpy(prev, name, sizeof prev - 1);
		}
		fprintf(stdout, "%ld\n", count);
		msh_close(in);
		free(rec.s);
		return 0;
	}

	ctx_open();
	memset(&rd, 0, sizeof rd);
	memset(&b, 0, sizeof b);
	rd.in = in;
	{
		long *dist = stat_mode >= 0 ? (long *)calloc(SUMMARY_MAX_READ_LENGTH + 1, sizeof(long)) : NULL;
		int32_t *h_len = NULL, *h_qlen = NULL, *h_qclip = NULL, *h_edit = NULL;
		size_t h_cap = 0;
		kstr line = {0, 0, 0};
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
				void *d_out;
				size_t i;
				/* every record along the MD path: the reference's summary never looks at NM, and a record without MD
				 * has no mismatches to count (mBamVector.c:200-219) -- its MD string is empty here */
				for (i = 0; i < b.n; i++) b.rflags[i] = (uint8_t)((b.rflags[i] | MSX_HAS_MD) & ~MSX_HAS_NM);
				rb_host_view(&b, &hb, 0);
				MSX(msx_batch_upload(g_ctx, &hb, &db));
				MSX(msx_dev_alloc(g_ctx, &d_out, 16 * b.n));
				{
					int32_t *d = (int32_t *)d_out;
					MSX(msx_aln_stats(g_ctx, &db, d, d + b.n, d + 2 * b.n, d + 3 * b.n, NULL));
				}
				if (b.n > h_cap) {
					h_cap = b.n + b.n / 4 + 1024;
					h_len = (int32_t *)realloc(h_len, 16 * h_cap);
					if (!h_len) mDie("Out of memory");
				}
				MSX(msx_dev_to_host(g_ctx, h_len, d_out, 16 * b.n));
				h_qlen = h_len + b.n; h_qclip = h_len + 2 * b.n; h_edit = h_len + 3 * b.n;
				msx_dev_free(g_ctx, d_out);
				msx_batch_free(g_ctx, &db);
				line.l = 0;
				for (i = 0; i < b.n; i++) {
					int64_t start, end;
					int32_t match, ext_edit, glocal, tid;
					if (b.flag[i] & 4) continue;                                  /* :56-57, :98-99 */
					if (b.flag[i] & 0x100) continue;                              /* :59-60 secondary */
					tid = b.tid[i];
					if (tid < 0 || tid >= hdr->n_targets)      /* (the reference reads target_len[tid] here whatever tid is) */
						mDie("Mapped record '%s' names no reference sequence", (const char *)RB_REC(&b, i) + 32);
					start = b.pos[i];
					end = record_endpos(&b, i);
					if (start < (int64_t)edge || (int64_t)hdr->target_len[tid] - end < (int64_t)edge) continue;    /* :62-63 */
					match = h_len[i] - h_edit[i];
					ext_edit = h_edit[i] + h_qclip[i];
					glocal = h_len[i] + h_qclip[i];
					if (stat_mode >= 0) {                                         /* :110-122 */
						const uint32_t stats[4] = {(uint32_t)match, (uint32_t)(h_qlen[i] - match), (uint32_t)ext_edit,
						                           (uint32_t)(match - ext_edit)};
						int idx = (int)stats[stat_mode];
						if (idx > SUMMARY_MAX_READ_LENGTH) idx = SUMMARY_MAX_READ_LENGTH;
						if (idx < 0) idx = 0;
						dist[idx]++;
					} else {                                                      /* :70-71 */
						char buf[96];
						const char *name = (const char *)RB_REC(&b, i) + 32;
						int l;
						ks_put(&line, name, strlen(name));
						l = snprintf(buf, sizeof buf, "\t%d\t", h_qlen[i]);
						ks_put(&line, buf, (size_t)l);
						ks_put(&line, hdr->target_name[tid], strlen(hdr->target_name[tid]));
						l = snprintf(buf, sizeof buf, "\t%d\t%d\t%.1f\n", glocal, match, 100.0 - 100.0 * ext_edit / glocal);
						ks_put(&line, buf, (size_t)l);
						if (line.l > ((size_t)1 << 20)) { fwrite(line.s, 1, line.l, stdout); line.l = 0; }
					}
				}
				if (line.l) fwrite(line.s, 1, line.l, stdout);
			}
			if (rd.done) break;
		}
		if (stat_mode >= 0) {                                                     /* :123-130 */
			int i;
			for (i = 0; i < SUMMARY_MAX_READ_LENGTH; i++)
				if (dist[i] > 0) fprintf(stdout, "%d\t%ld\n", i, dist[i]);
			if (dist[SUMMARY_MAX_READ_LENGTH] > 0) fprintf(stdout, "%d+\t%ld\n", SUMMARY_MAX_READ_LENGTH, dist[SUMMARY_MAX_READ_LENGTH]);
		}
		free(dist);
		free(h_len);
		free(line.s);
	}
	fflush(stdout);
	msh_close(in);
	free(rec.s);
	fast_exit();
	msx_ctx_destroy(g_ctx);
	return 0;
}
