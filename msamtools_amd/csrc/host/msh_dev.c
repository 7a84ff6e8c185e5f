/*
 * msh_dev.c -- `msamtools-dev`: developer and test commands that are NOT part of the reference's surface and not in the
 * product binary: synth (deterministic BAM generator), recode, digest, pipetest, restream, rawtest, keyorder.
 */
#include "msh_cli.h"

/* Host I/O self-test (no GPU): `msamtools recode [-b|-u|-h] <file>` reads any
 * supported input and writes every record back out.  Not part of the
 * reference's surface; used by the test-suite to check the readers/writers. */
int recode_main(int argc, char *argv[]) {
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
int synth_main(int argc, char *argv[]) {
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
		else if (strcmp(argv[i], "-h") == 0) mode = MSH_OUT_SAM_HDR;      /* SAM text with its header: what an aligner pipes into `filter -S` */
		else mQuit("usage: %s synth --groups N --refs R [--seed S] [--seq] [-b|-u|-h]", PROGRAM);
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

int pipetest_main(int argc, char *argv[]) {
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
/* --full: every byte of the record (block_size bytes: fixed fields, QNAME, CIGAR, SEQ, QUAL, aux) -- what byte-for-byte
 * equality of filter's output with the input records it selects means at a size no text comparison goes to */
static uint64_t dg_record_full(const uint8_t *r, size_t len) {
	uint64_t h = 0x9e3779b97f4a7c15ull ^ (uint64_t)len;
	size_t i = 0;
	for (; i + 8 <= len; i += 8) {
		uint64_t w;
		memcpy(&w, r + i, 8);
		h = (h ^ w) * 0xff51afd7ed558ccdull;
		h = (h << 29) | (h >> 35);
	}
	if (i < len) {
		uint64_t w = 0;
		memcpy(&w, r + i, len - i);
		h = (h ^ w) * 0xff51afd7ed558ccdull;
	}
	return dg_mix(h);
}
typedef struct {
	const rbatch *b;
	uint64_t first, part[MSH_POOL_MAX], cnt[MSH_POOL_MAX];
	int full;
	const uint32_t *rank;      /* --select: rank[i] = 1 + the place of input record i in the selection, 0 = not selected */
	size_t n_rank;
} digest_job;
static void digest_worker(void *arg, int tid, int nth) {
	digest_job *J = (digest_job *)arg;
	const rbatch *b = J->b;
	const size_t lo = b->n * (size_t)tid / (size_t)nth, hi = b->n * (size_t)(tid + 1) / (size_t)nth;
	uint64_t s = 0, c = 0;
	size_t i;
	for (i = lo; i < hi; i++) {
		const uint8_t *r = RB_REC(b, i);
		uint64_t place = J->first + (uint64_t)i + 1;
		msh_rec_check(r, RB_LEN(b, i));
		if (J->rank) {
			const uint64_t g = J->first + (uint64_t)i;
			if (g >= J->n_rank || J->rank[g] == 0) continue;
			place = J->rank[g];
		}
		c++;
		s += place * (J->full ? dg_record_full(r, RB_LEN(b, i)) : dg_record(r));
	}
	J->part[tid] = s;
	J->cnt[tid] = c;
}

/* digest [--full] [--select idx.u32] <file>: --select reads little-endian uint32 record indices (the oracle's emit list, in
 * emit order) and digests the selected records as if they had been written in that order -- the figure filter's output file
 * must give */
int digest_main(int argc, char *argv[]) {
	msh_in *in;
	uint64_t n = 0, h = 0, n_sel = 0;
	int full = 0, a = 1;
	const char *select = NULL;
	uint32_t *rank = NULL;
	size_t n_rank = 0;
	for (; a < argc - 1; a++) {
		if (strcmp(argv[a], "--full") == 0) full = 1;
		else if (strcmp(argv[a], "--select") == 0 && a + 1 < argc - 1) select = argv[++a];
		else break;
	}
	if (a != argc - 1) mQuit("usage: %s digest [--full] [--select idx.u32] <file>", PROGRAM);
	if (select) {
		FILE *f = fopen(select, "rb");
		uint32_t *idx;
		size_t k, m;
		long sz;
		if (!f) mDie("Cannot open %s for reading", select);
		fseek(f, 0, SEEK_END); sz = ftell(f); fseek(f, 0, SEEK_SET);
		m = (size_t)sz / 4;
		idx = (uint32_t *)xmalloc((m + 1) * 4);
		if (fread(idx, 4, m, f) != m) mDie("Cannot read %s", select);
		fclose(f);
		for (k = 0; k < m; k++) if ((size_t)idx[k] + 1 > n_rank) n_rank = (size_t)idx[k] + 1;
		rank = (uint32_t *)calloc(n_rank + 1, 4);
		if (!rank) mDie("out of memory");
		for (k = 0; k < m; k++) {
			if (rank[idx[k]]) mDie("%s: record %u selected twice", select, idx[k]);
			rank[idx[k]] = (uint32_t)(k + 1);
		}
		free(idx);
	}
	in = msh_open(argv[a]);
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
			J.b = &s->b; J.first = n; J.full = full; J.rank = rank; J.n_rank = n_rank;
			msh_parallel(nth, digest_worker, &J);
			for (t = 0; t < nth; t++) { h += J.part[t]; n_sel += J.cnt[t]; }
			n += s->b.n;
			pq_push(&P.q_free, si);
		}
		pthread_join(th, NULL);
	} else {
		kstr rec = {0, 0, 0};
		while (msh_read(in, &rec) == 0) {
			uint64_t place = ++n;
			if (rank) {
				if (n - 1 >= n_rank || rank[n - 1] == 0) continue;
				place = rank[n - 1];
			}
			n_sel++;
			h += place * (full ? dg_record_full((const uint8_t *)rec.s, rec.l) : dg_record((const uint8_t *)rec.s));
		}
		free(rec.s);
	}
	printf("records=%llu digest=%016llx\n", (unsigned long long)(rank ? n_sel : n), (unsigned long long)h);
	free(rank);
	msh_close(in);
	return 0;
}

/* hidden, host only: `msamtools restream [-b|-u] <file>` decodes the input through the pipeline's decode stage and
 * writes every batch with msh_write_stream -- the writer of device-unpacked batches: a ready-made record stream, cut
 * into BGZF payloads where they fall -- so that this writer is tested without a GPU. */
int restream_main(int argc, char *argv[]) {
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
int rawtest_main(int argc, char *argv[]) {
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
	if (argc < 2) {
		fprintf(stderr, "usage: msamtools-dev <synth|recode|digest|pipetest|restream|rawtest|keyorder> ...\n");
		return 1;
	}
	if (strcmp(argv[1], "keyorder") == 0) {
		/* host only: names on stdin (one per line) -> the reference's key order on stdout */
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
	fprintf(stderr, "[msamtools-dev] unrecognized command '%s'\n", argv[1]);
	return 1;
}
