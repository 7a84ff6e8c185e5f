/*
 * msx_oracle.c -- CPU oracle (TEST INFRASTRUCTURE, see msx_oracle.h).
 *
 * Scalar, single-threaded, written for fidelity to the reference's control
 * flow, not for speed.  All file:line citations are into /root/reference
 * (arumugamlab/msamtools v1.1.3).
 */
#include "msx_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* BAM constants (SAM spec section 4.2; htslib sam.h names in comments) */
#define OP_MATCH 0      /* BAM_CMATCH      */
#define OP_INS 1        /* BAM_CINS        */
#define OP_DEL 2        /* BAM_CDEL        */
#define OP_REF_SKIP 3   /* BAM_CREF_SKIP   */
#define OP_SOFT_CLIP 4  /* BAM_CSOFT_CLIP  */
#define OP_HARD_CLIP 5  /* BAM_CHARD_CLIP  */
#define OP_PAD 6        /* BAM_CPAD        */
#define OP_EQUAL 7      /* BAM_CEQUAL      */
#define OP_DIFF 8       /* BAM_CDIFF       */
#define F_UNMAP 4       /* BAM_FUNMAP      */
#define F_READ1 64      /* BAM_FREAD1      */
#define F_READ2 128     /* BAM_FREAD2      */

/* The reference does its sums and threshold products in `int`/int32_t; signed
 * overflow there is undefined behaviour that in practice wraps.  The oracle
 * (and the HIP kernels) make the wrap explicit. */
static int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }

/* ---------------------------------------------------------------------- */
/* mBamVector.c:23-38  bam_cigar2details                                    */
/* ---------------------------------------------------------------------- */
void orc_cigar2details(const uint32_t *cigar, uint32_t n_cigar,
                       int32_t *alen, int32_t *qlen, int32_t *qclip) {
	uint32_t k;
	*alen = *qlen = *qclip = 0;
	for (k = 0; k < n_cigar; ++k) {
		int op = cigar[k] & 0xf;
		int w = (int)(cigar[k] >> 4);
		if (op == OP_HARD_CLIP || op == OP_SOFT_CLIP) {
			*qclip = wadd(*qclip, w);
			*qlen = wadd(*qlen, w);
		} else if (!(op == OP_REF_SKIP || op == OP_PAD)) {
			*alen = wadd(*alen, w);
			if (op == OP_MATCH || op == OP_EQUAL || op == OP_DIFF || op == OP_INS)
				*qlen = wadd(*qlen, w);
		}
	}
}

/* ---------------------------------------------------------------------- */
/* htslib 1.24 kstring.c kstrtok(), restated (third-party, not in the       */
/* reference tree).  Splits on any byte of `sep`; EVERY separator ends a    */
/* token, so runs of separators yield empty tokens; aux->p is the end of    */
/* the token just returned.  Only the multi-character-separator branch is   */
/* needed (the reference passes "^0123456789").                             */
/* ---------------------------------------------------------------------- */
typedef struct {
	uint64_t tab[4];
	int finished;
	const char *p;
} orc_tokaux;

static const char *orc_kstrtok(const char *str, const char *sep, orc_tokaux *aux) {
	const unsigned char *p, *start;
	if (sep) {
		if (str == 0 && aux->finished) return 0;
		aux->finished = 0;
		aux->tab[0] = aux->tab[1] = aux->tab[2] = aux->tab[3] = 0;
		for (p = (const unsigned char *)sep; *p; ++p)
			aux->tab[*p >> 6] |= 1ull << (*p & 0x3f);
	}
	if (aux->finished) return 0;
	else if (str) start = (const unsigned char *)str, aux->finished = 0;
	else start = (const unsigned char *)aux->p + 1;
	for (p = start; *p; ++p)
		if (aux->tab[*p >> 6] >> (*p & 0x3f) & 1) break;
	aux->p = (const char *)p;
	if (*p == 0) aux->finished = 1;
	return (const char *)start;
}

/* ---------------------------------------------------------------------- */
/* mBamVector.c:40-133  bam_get_summary                                     */
/* ---------------------------------------------------------------------- */
void orc_get_summary(const uint32_t *cigar, uint32_t n_cigar, const char *md,
                     orc_summary *summary) {
	int32_t alen = 0, qlen = 0, qclip = 0;
	int32_t match = 0, mismatch = 0, edit = 0;
	uint32_t k;

	for (k = 0; k < n_cigar; ++k) {               /* :60-97 */
		int op = cigar[k] & 0xf;
		int w = (int)(cigar[k] >> 4);
		switch (op) {
		case OP_MATCH:
		case OP_EQUAL:
		case OP_DIFF:
			match = wadd(match, w);
			qlen = wadd(qlen, w);
			alen = wadd(alen, w);
			break;
		case OP_INS:
			qlen = wadd(qlen, w);
			/* FALL-THROUGH */
		case OP_DEL:
			edit = wadd(edit, w);
			alen = wadd(alen, w);
			break;
		case OP_HARD_CLIP:
		case OP_SOFT_CLIP:
			qclip = wadd(qclip, w);
			qlen = wadd(qlen, w);
			break;
		default:
			break;
		}
	}

	if (md) {                                     /* :101-120 */
		orc_tokaux aux;
		const char *p;
		aux.finished = 0;
		aux.p = 0;
		for (p = orc_kstrtok(md, "^0123456789", &aux); p; p = orc_kstrtok(0, 0, &aux)) {
			const char *x;
			if (p > md && p[-1] != '^')
				for (x = p; x < aux.p; x++)
					edit = wadd(edit, 1);
			/* :117 -- runs once per token (misleading indentation upstream) */
			mismatch = wadd(mismatch, 1);
		}
		match = wsub(match, mismatch);
	}

	summary->match = match;
	summary->mismatch = mismatch;
	summary->edit = edit;
	summary->query_length = qlen;
	summary->query_clip = qclip;
	summary->length = alen;
}

/* ---------------------------------------------------------------------- */
/* msam_filter.c:31-63 predicates, :71-85 dispatch                          */
/* ---------------------------------------------------------------------- */
static int filter_choice(const orc_filter_params *p) {
	int c = 0;
	if (p->min_length > 0) c |= 1;
	if (p->ppt != 0) c |= 2;
	if (p->max_clip < 100) c |= 4;
	return c;
}

int orc_filter_fails(const orc_summary *a, const orc_filter_params *p) {
	int c = filter_choice(p);
	int fl = a->length < p->min_length;                                        /* _FILTER_L */
	int fz = wmul(100, a->query_clip) > wmul(p->max_clip, a->query_length);   /* _FILTER_Z */
	int fp;
	if (p->ppt < 0)                                                            /* _FILTER_P */
		fp = wmul(1000, wsub(a->edit, a->length)) < wmul(a->length, p->ppt);
	else
		fp = wmul(1000, wsub(a->length, a->edit)) < wmul(a->length, p->ppt);
	/* filters[8] = {NULL, l, p, lp, z, lz, pz, lpz}; each is an || chain */
	return ((c & 1) && fl) || ((c & 2) && fp) || ((c & 4) && fz);
}

/* per-record view helpers */
static char *md_dup(const orc_records *r, int64_t i) {
	uint32_t s = r->md_off[i], e = r->md_off[i + 1];
	char *z = (char *)malloc((size_t)(e - s) + 1);
	memcpy(z, r->md + s, e - s);
	z[e - s] = 0;
	return z;
}

/* msam_filter.c:145-157: 0 ok, 1 neither MD nor NM */
static int record_stats(const orc_records *r, int64_t i, orc_summary *a) {
	const uint32_t *cig = r->cigar + r->cigar_off[i];
	uint32_t n_cigar = r->cigar_off[i + 1] - r->cigar_off[i];
	memset(a, 0, sizeof(*a));
	if (r->rflags[i] & ORC_HAS_MD) {
		char *md = md_dup(r, i);
		orc_get_summary(cig, n_cigar, md, a);
		free(md);
	} else {
		if (!(r->rflags[i] & ORC_HAS_NM)) return 1;
		orc_cigar2details(cig, n_cigar, &a->length, &a->query_length, &a->query_clip);
		a->edit = r->nm[i];
	}
	return 0;
}

void orc_aln_stats(const orc_records *r, int32_t *length, int32_t *qlen,
                   int32_t *qclip, int32_t *edit, uint8_t *status) {
	int64_t i;
	for (i = 0; i < r->n; i++) {
		orc_summary a;
		int st = record_stats(r, i, &a);
		if (length) length[i] = a.length;
		if (qlen) qlen[i] = a.query_length;
		if (qclip) qclip[i] = a.query_clip;
		if (edit) edit[i] = a.edit;
		if (status) status[i] = (uint8_t)st;
	}
}

static int same_name(const orc_records *r, int64_t a, int64_t b) {
	if (r->qname_off) {
		uint32_t la = r->qname_off[a + 1] - r->qname_off[a];
		uint32_t lb = r->qname_off[b + 1] - r->qname_off[b];
		return la == lb && memcmp(r->qname + r->qname_off[a], r->qname + r->qname_off[b], la) == 0;
	}
	return r->name_id[a] == r->name_id[b];
}

/* growable index pool standing in for mBamPool (mBamVector.c:286-348) */
typedef struct {
	int64_t *elem;
	int64_t size, limit;
} ipool;

static void ipool_push(ipool *p, int64_t v) {
	if (p->size == p->limit) {
		p->limit = p->limit ? 2 * p->limit : 64;   /* msam_filter.c:103, mBamVector.c:297-313 */
		p->elem = (int64_t *)realloc(p->elem, (size_t)p->limit * sizeof(int64_t));
	}
	p->elem[p->size++] = v;
}

/* ---------------------------------------------------------------------- */
/* writers: msam_filter.c:192-263, mBamVector.c:343-348                     */
/* ---------------------------------------------------------------------- */
typedef struct {
	const orc_records *r;
	const int32_t *as;        /* current AS per record (after rescore)  */
	const uint8_t *has_as;
	int32_t *emit_idx;
	int64_t n_emit;
	int err;
	int64_t err_record;
} wctx;

static void write_besthit_by_mate(wctx *w, const ipool *pool, uint32_t mate_flag, int unique_only) {
	int64_t i;
	int best_count = 0;
	int32_t best_score = INT32_MIN;
	for (i = 0; i < pool->size; i++) {
		int64_t e = pool->elem[i];
		int32_t score;
		if ((w->r->flag[e] & (F_READ1 | F_READ2)) != mate_flag) continue;
		if (!w->has_as[e]) {
			if (!w->err) { w->err = ORC_ERR_NO_AS; w->err_record = e; }
			return;
		}
		score = w->as[e];
		if (score > best_score) {
			best_score = score;
			best_count = 1;
		} else if (score == best_score) {
			best_count++;
		}
	}
	if (best_count == 0 || (unique_only && best_count != 1)) return;
	for (i = 0; i < pool->size; i++) {
		int64_t e = pool->elem[i];
		if ((w->r->flag[e] & (F_READ1 | F_READ2)) != mate_flag) continue;
		if (w->as[e] == best_score) w->emit_idx[w->n_emit++] = (int32_t)e;
	}
}

static void write_pool(wctx *w, const ipool *pool, const orc_filter_params *p) {
	int64_t i;
	if (w->err) return;
	if (p->uniqhit || p->besthit) {            /* :88-91: uniqhit wins */
		int unique_only = p->uniqhit ? 1 : 0;
		int paired = 0;
		for (i = 0; i < pool->size; i++)       /* mBamPoolIsPaired :196-204 */
			if (w->r->flag[pool->elem[i]] & (F_READ1 | F_READ2)) { paired = 1; break; }
		if (paired) {
			write_besthit_by_mate(w, pool, F_READ1, unique_only);
			if (!w->err) write_besthit_by_mate(w, pool, F_READ2, unique_only);
		} else {
			write_besthit_by_mate(w, pool, 0, unique_only);
		}
	} else {                                   /* mWriteBamPool */
		for (i = 0; i < pool->size; i++) w->emit_idx[w->n_emit++] = (int32_t)pool->elem[i];
	}
}

/* ---------------------------------------------------------------------- */
/* msam_filter.c:65-190  mFilterFileWrapper + mFilterFile                   */
/* ---------------------------------------------------------------------- */
int orc_filter(const orc_records *r, const orc_filter_params *p,
               int32_t *emit_idx, int64_t *n_emit, int32_t *as_out,
               int64_t *err_record) {
	int choice = filter_choice(p);
	int filter_active = choice != 0;                         /* filter != NULL */
	int need_alignment_stats = filter_active || p->rescore;  /* :104 */
	int64_t i, prev = -1;
	ipool pool = {0, 0, 0};
	int32_t *as_cur;
	uint8_t *has_as;
	wctx w;
	int rc = ORC_OK;

	*n_emit = 0;
	if (err_record) *err_record = -1;
	if (choice == 0 && !p->besthit && !p->uniqhit) return ORC_ERR_NO_FILTER;   /* :82-84 */

	as_cur = (int32_t *)malloc((size_t)(r->n ? r->n : 1) * sizeof(int32_t));
	has_as = (uint8_t *)malloc((size_t)(r->n ? r->n : 1));
	for (i = 0; i < r->n; i++) {
		as_cur[i] = r->as[i];
		has_as[i] = (r->rflags[i] & ORC_HAS_AS) ? 1 : 0;
	}
	w.r = r; w.as = as_cur; w.has_as = has_as; w.emit_idx = emit_idx;
	w.n_emit = 0; w.err = 0; w.err_record = -1;

	for (i = 0; i < r->n; i++) {                               /* :119 */
		orc_summary a;
		if (prev >= 0 && !same_name(r, i, prev)) {             /* :120-125 */
			write_pool(&w, &pool, p);
			if (w.err) break;
			pool.size = 0;
		}
		if (r->flag[i] & F_UNMAP) {                            /* :132-138 */
			if (filter_active && p->keep_unmapped != 0) {
				if (p->ppt >= 0 && p->invert == 1) ipool_push(&pool, i);
			}
			continue;
		}
		memset(&a, 0, sizeof(a));
		if (need_alignment_stats) {                            /* :145-157 */
			if (record_stats(r, i, &a)) {
				rc = ORC_ERR_NO_MD_NM;
				if (err_record) *err_record = i;
				break;
			}
		}
		if (p->rescore) {                                      /* :160-168 */
			int32_t score = wadd(wmul(wsub(a.length, a.edit), 1), wmul(a.edit, -1));
			as_cur[i] = score;
			has_as[i] = 1;
		}
		prev = i;                                              /* :170 */
		if (!filter_active || orc_filter_fails(&a, p) == p->invert)   /* :181 */
			ipool_push(&pool, i);
	}
	if (rc == ORC_OK && !w.err) write_pool(&w, &pool, p);      /* :186 */
	if (w.err) {
		rc = w.err;
		if (err_record) *err_record = w.err_record;
	}
	*n_emit = w.n_emit;
	if (as_out) memcpy(as_out, as_cur, (size_t)r->n * sizeof(int32_t));
	free(pool.elem);
	free(as_cur);
	free(has_as);
	return rc;
}

/* ---------------------------------------------------------------------- */
/* profile                                                                  */
/* ---------------------------------------------------------------------- */
typedef struct {
	int64_t *off;     /* [n_lists+1] */
	int32_t *fid;
	int64_t n_lists, cap_lists, n_fid, cap_fid;
} multi_csr;

static void multi_begin(multi_csr *m) {
	if (m->n_lists + 2 > m->cap_lists) {
		m->cap_lists = m->cap_lists ? 2 * m->cap_lists : 65536;   /* msam_profile.c:38 */
		m->off = (int64_t *)realloc(m->off, (size_t)m->cap_lists * sizeof(int64_t));
	}
	if (m->n_lists == 0) m->off[0] = 0;
}
static void multi_push_fid(multi_csr *m, int32_t f) {
	if (m->n_fid == m->cap_fid) {
		m->cap_fid = m->cap_fid ? 2 * m->cap_fid : 65536;
		m->fid = (int32_t *)realloc(m->fid, (size_t)m->cap_fid * sizeof(int32_t));
	}
	m->fid[m->n_fid++] = f;
}
static void multi_end(multi_csr *m) { m->off[++m->n_lists] = m->n_fid; }

typedef struct {
	const orc_records *r;
	const int32_t *fmap;
	int32_t share_type;
	uint32_t *ui;         /* global->ui_insert_count */
	double *d;            /* global->d_insert_count  */
	uint8_t *hit;         /* global->ub_target_hit   */
	multi_csr mm;         /* global->multi_mappers   */
	orc_profile_stats *st;
} pctx;

static int32_t fid_of(const pctx *c, int64_t rec) {
	int32_t tid = c->r->tid[rec];
	return c->fmap ? c->fmap[tid] : tid;
}

/* msam_profile.c:65-200  mEstimateInsertCountOnPool */
static void estimate_on_pool(pctx *c, const ipool *pool) {
	int64_t size = pool->size, i;
	if (size == 1) {                                           /* :75-78 */
		c->ui[fid_of(c, pool->elem[0])] += 2;
		c->st->uniq_mapper_count++;
		return;
	}
	if (size == 2) {                                           /* :80-127 */
		int32_t fid0 = fid_of(c, pool->elem[0]);
		int32_t fid1 = fid_of(c, pool->elem[1]);
		if (fid0 == fid1) {
			c->ui[fid0] += 2;
			c->st->uniq_mapper_count++;
			return;
		}
		c->st->multi_mapper_count++;
		switch (c->share_type) {
		case ORC_MULTI_IGNORE: break;
		case ORC_MULTI_ADD_ALL: c->ui[fid0] += 2; c->ui[fid1] += 2; break;
		case ORC_MULTI_SHARE_EQUAL: c->ui[fid0]++; c->ui[fid1]++; break;
		case ORC_MULTI_SHARE_PROPORTIONAL:
			multi_begin(&c->mm);
			multi_push_fid(&c->mm, fid0);
			multi_push_fid(&c->mm, fid1);
			multi_end(&c->mm);
			break;
		}
		return;
	}
	{                                                          /* default :129-198 */
		int64_t first = c->mm.n_fid, k, cnt;
		/* distinct features in first-appearance order (:136-142); the list
		 * is staged at the tail of the CSR and dropped again unless kept */
		for (i = 0; i < size; i++) {
			int32_t fid = fid_of(c, pool->elem[i]);
			if (!c->hit[fid]) {
				multi_push_fid(&c->mm, fid);
				c->hit[fid] = 1;
			}
		}
		cnt = c->mm.n_fid - first;
		for (k = first; k < c->mm.n_fid; k++) c->hit[c->mm.fid[k]] = 0;   /* :145 */
		if (cnt == 1) {                                        /* :152-159 */
			c->ui[c->mm.fid[first]] += 2;
			c->st->uniq_mapper_count++;
			c->mm.n_fid = first;
			return;
		}
		c->st->multi_mapper_count++;
		switch (c->share_type) {
		case ORC_MULTI_IGNORE: break;
		case ORC_MULTI_ADD_ALL:
			for (k = first; k < first + cnt; k++) c->ui[c->mm.fid[k]] += 2;
			break;
		case ORC_MULTI_SHARE_EQUAL: {
			double share = 1.0 / cnt;
			for (k = first; k < first + cnt; k++) c->d[c->mm.fid[k]] += share;
			break;
		}
		case ORC_MULTI_SHARE_PROPORTIONAL:
			/* keep the staged list: it is already at the CSR tail */
			multi_begin(&c->mm);
			multi_end(&c->mm);
			return;
		}
		c->mm.n_fid = first;
	}
}

int orc_profile(const orc_records *r, const int32_t *sel, int64_t n_sel,
                const int32_t *fmap, int32_t n_features, int32_t share_type,
                double *abundance, uint32_t *ui_out, orc_profile_stats *stats) {
	pctx c;
	ipool pool = {0, 0, 0};
	int64_t s, n_stream = sel ? n_sel : r->n, prev = -1;
	int32_t i;
	int64_t j;
	size_t nf = (size_t)(n_features > 0 ? n_features : 1);

	memset(stats, 0, sizeof(*stats));
	memset(&c, 0, sizeof(c));
	c.r = r; c.fmap = fmap; c.share_type = share_type; c.st = stats;
	c.ui = (uint32_t *)calloc(nf, sizeof(uint32_t));          /* mInitInsertCounts :23-40 */
	c.hit = (uint8_t *)calloc(nf, 1);
	c.d = (double *)calloc(nf, sizeof(double));

	/* mEstimateInsertCountOnFile :204-243 */
	for (s = 0; s < n_stream; s++) {
		int64_t rec = sel ? sel[s] : s;
		if (r->tid[rec] == -1) continue;                       /* :223-225 */
		if (prev >= 0 && !same_name(r, rec, prev)) {           /* :226-231 */
			estimate_on_pool(&c, &pool);
			pool.size = 0;
			stats->insert_count++;
		}
		prev = rec;                                            /* :232 */
		ipool_push(&pool, rec);                                /* :233 */
	}
	if (pool.size > 0) {                                       /* :235-238 */
		estimate_on_pool(&c, &pool);
		stats->insert_count++;
	}

	/* mInsertCountToAbundanceMatrix :248-425 */
	if (ui_out) memcpy(ui_out, c.ui, (size_t)n_features * sizeof(uint32_t));
	for (i = 0; i < n_features; i++) abundance[i] = 1.0 * c.ui[i] / 2;   /* :284-289 */

	switch (share_type) {
	case ORC_MULTI_IGNORE:
	case ORC_MULTI_ADD_ALL:
		break;
	case ORC_MULTI_SHARE_EQUAL:
		for (i = 0; i < n_features; i++) abundance[i] += c.d[i];          /* :303-308 */
		break;
	case ORC_MULTI_SHARE_PROPORTIONAL: {
		int k;
		double *t_k = (double *)malloc(nf * sizeof(double));
		double *t_km1 = (double *)malloc(nf * sizeof(double));
		double *increment = (double *)malloc(nf * sizeof(double));
		memcpy(t_k, abundance, (size_t)n_features * sizeof(double));     /* :326 */
		for (k = 1; k < 20; k++) {                                       /* :331 */
			double delta = 0;
			for (i = 0; i < n_features; i++) increment[i] = 0.0f;
			memcpy(t_km1, t_k, (size_t)n_features * sizeof(double));
			for (j = 0; j < c.mm.n_lists; j++) {                         /* :341-365 */
				const int32_t *elem = c.mm.fid + c.mm.off[j];
				int64_t sz = c.mm.off[j + 1] - c.mm.off[j], e;
				double sum = 0;
				for (e = 0; e < sz; e++) sum += t_k[elem[e]];
				if (sum > 0)
					for (e = 0; e < sz; e++) increment[elem[e]] += (t_k[elem[e]] / sum);
			}
			delta = 0;
			for (i = 0; i < n_features; i++) {                           /* :369-379 */
				double diff;
				t_k[i] = abundance[i] + increment[i];
				if (t_k[i] < 1e-20) t_k[i] = 0;
				diff = t_k[i] - t_km1[i];
				delta += diff * diff;
			}
			delta /= n_features;
			stats->iterations = k;
			stats->last_delta = delta;
			if (delta < 1e-10) {                                         /* :383 */
				stats->converged = 1;
				break;
			}
		}
		for (j = 0; j < c.mm.n_lists; j++) {                             /* :394-404 */
			const int32_t *elem = c.mm.fid + c.mm.off[j];
			int64_t sz = c.mm.off[j + 1] - c.mm.off[j], e;
			double sum = 0;
			for (e = 0; e < sz; e++) sum += t_k[elem[e]];
			if (sum == 0) stats->purged_insert_count++;
		}
		memcpy(abundance, t_k, (size_t)n_features * sizeof(double));     /* :391 */
		free(t_k); free(t_km1); free(increment);
		break;
	}
	default:
		break;
	}

	free(c.ui); free(c.hit); free(c.d); free(c.mm.off); free(c.mm.fid); free(pool.elem);
	return ORC_OK;
}

/* msam_profile.c:858-975 + mMatrix.c:137-179 */
void orc_profile_finish(double *values, int32_t n_features,
                        const uint32_t *feature_len, int32_t unit_type,
                        int32_t length_normalize, int32_t total_inserts,
                        int32_t mincount, int32_t share_type,
                        const orc_profile_stats *stats,
                        double *purged_inserts_out, double *effective_inserts_out) {
	int32_t i, ncols = n_features + 1;
	int mapped_inserts = (int)stats->insert_count;
	double purged_insert_equivalent = 0, purged_inserts, effective_inserts;

	values[0] = 0.0;                                         /* :421 */
	if (mincount >= 0) {                                     /* :858-869 */
		for (i = 1; i < ncols; i++) {
			if (values[i] < mincount) {
				purged_insert_equivalent += values[i];
				values[i] = 0;
			}
		}
	}
	if (total_inserts > 0 && total_inserts < mapped_inserts) total_inserts = -1;   /* :873-876 */

	purged_inserts = stats->purged_insert_count + purged_insert_equivalent;        /* :889-893 */
	effective_inserts = mapped_inserts - purged_inserts;
	if (share_type == ORC_MULTI_IGNORE) effective_inserts -= stats->multi_mapper_count;

	if (total_inserts > 0) {                                 /* :906-934 */
		values[0] = total_inserts - mapped_inserts + purged_inserts;
		if (share_type == ORC_MULTI_IGNORE) values[0] += stats->multi_mapper_count;
		if (length_normalize) {
			int count = 0;
			uint64_t sum = 0;
			uint32_t unknown_size;
			for (i = 0; i < n_features; i++) { sum += feature_len[i]; count++; }
			unknown_size = (uint32_t)(sum / (uint64_t)count);
			values[0] = 1.0 * values[0] / unknown_size;
		}
	}
	if (length_normalize)                                    /* :937-947 */
		for (i = 0; i < n_features; i++) values[1 + i] /= feature_len[i];

	switch (unit_type) {                                     /* :950-975 */
	case 2: {
		double d = (total_inserts > 0) ? 1.0E9 / total_inserts : 1.0E9 / mapped_inserts;
		for (i = 0; i < ncols; i++) values[i] *= d;
		break;
	}
	case 3:
	case 1: {
		double sum = 0;
		for (i = 0; i < ncols; i++) sum += values[i];
		for (i = 0; i < ncols; i++) values[i] /= sum;
		if (unit_type == 3)
			for (i = 0; i < ncols; i++) values[i] *= 1.0E6;
		break;
	}
	default:
		break;
	}
	if (purged_inserts_out) *purged_inserts_out = purged_inserts;
	if (effective_inserts_out) *effective_inserts_out = effective_inserts;
}

/* ---------------------------------------------------------------------- */
/* msam_coverage.c:33-87 per alignment; :106-139 adds 1 for every record    */
/* (pool grouping does not change the result: cov = 1 per alignment).       */
/* ---------------------------------------------------------------------- */
void orc_coverage(const orc_records *r, const int64_t *cov_off,
                  int32_t n_targets, int32_t *cov) {
	int64_t i;
	(void)n_targets;
	for (i = 0; i < r->n; i++) {
		int32_t tid = r->tid[i];
		int64_t pos;
		uint32_t k;
		int32_t *this_coverage;
		if (tid < 0) continue;                              /* :42 */
		this_coverage = cov + cov_off[tid];
		pos = r->pos[i];
		for (k = r->cigar_off[i]; k < r->cigar_off[i + 1]; k++) {
			int op = r->cigar[k] & 0xf;
			int w = (int)(r->cigar[k] >> 4), x;
			switch (op) {
			case OP_MATCH:
			case OP_EQUAL:
			case OP_DIFF:
				for (x = 0; x < w; x++) this_coverage[pos + x] += 1;
				pos += w;
				break;
			case OP_DEL:
			case OP_REF_SKIP:
				pos += w;
				break;
			default:
				break;
			}
		}
	}
}

/* ------------------------------------------------------------------------- */
/* zoeTools.c:202-363: the hash table whose key list orders the genomes       */
/* ------------------------------------------------------------------------- */
typedef struct { int32_t *item; int32_t n, cap; } orc_ivec;

static void ivec_push(orc_ivec *v, int32_t x) {
	if (v->n == v->cap) {
		v->cap = v->cap ? 2 * v->cap : 4;
		v->item = (int32_t *)realloc(v->item, sizeof(int32_t) * (size_t)v->cap);
	}
	v->item[v->n++] = x;
}

/* zoeHashFunc, :218-228 (plain char arithmetic, multipliers :202-210) */
static int orc_zoe_slot(int slots, const char *key) {
	static const double mult[7] = {3.1415926536, 2.7182818285, 1.6180339887, 1.7320508076,
	                               2.2360679775, 2.6457513111, 3.3166247904};
	double sum = 0;
	size_t i;
	for (i = 0; i < strlen(key); i++) sum += key[i] * mult[i % 7];
	return (int)(slots * (sum - floor(sum)));
}

int32_t orc_key_order(const char *const *names, int32_t n, int32_t *order) {
	int level = 1, slots = 4, s;                 /* zoeNewHash -> zoeExpandHash: level 1, pow(4, 1) slots (:304-314, :214-216) */
	orc_ivec *bucket = (orc_ivec *)calloc((size_t)slots, sizeof(orc_ivec));
	orc_ivec keys = {0, 0, 0};                   /* hash->keys: first-occurrence indices in list order */
	int32_t i, j, nk;
	for (i = 0; i < n; i++) {
		/* zoeSetHash :330-357 */
		orc_ivec *b = &bucket[orc_zoe_slot(slots, names[i])];
		int found = 0;
		for (j = 0; j < b->n; j++)
			if (strcmp(names[b->item[j]], names[i]) == 0) { found = 1; break; }
		if (found) continue;
		ivec_push(&keys, i);
		ivec_push(b, i);
		if ((float)keys.n / (float)slots >= 2.0f) {
			/* zoeExpandHash :230-279: one level up, old buckets re-inserted in slot order */
			const int old_slots = slots;
			orc_ivec *old = bucket;
			level++;
			slots = (int)pow(4, level);
			bucket = (orc_ivec *)calloc((size_t)slots, sizeof(orc_ivec));
			keys.n = 0;
			for (s = 0; s < old_slots; s++) {
				for (j = 0; j < old[s].n; j++) {
					const int32_t id = old[s].item[j];
					ivec_push(&keys, id);
					ivec_push(&bucket[orc_zoe_slot(slots, names[id])], id);
				}
				free(old[s].item);
			}
			free(old);
		}
	}
	nk = keys.n;
	for (i = 0; i < nk; i++) order[i] = keys.item[i];     /* zoeKeysOfHash :365 */
	for (s = 0; s < slots; s++) free(bucket[s].item);
	free(bucket);
	free(keys.item);
	return nk;
}

/* ------------------------------------------------------------------------- */
/* summary: mBamVector.c:135-236, msam_summary.c:19-135                       */
/* ------------------------------------------------------------------------- */
void orc_ext_summary_record(const uint32_t *cigar, uint32_t n_cigar, const char *md, int has_md, orc_ext_summary *out) {
	int32_t alen = 0, qlen = 0, qclip = 0, match = 0, mismatch = 0, gapopen = 0, gapextend = 0;
	uint32_t k;
	for (k = 0; k < n_cigar; ++k) {                          /* :155-196 */
		int op = (int)(cigar[k] & 0xf);
		int w = (int)(cigar[k] >> 4);
		switch (op) {
		case OP_MATCH: case OP_EQUAL: case OP_DIFF:
			match += w; qlen += w; alen += w;
			break;
		case OP_INS:
			qlen += w; gapopen++; gapextend += (w - 1); alen += w;
			break;
		case OP_DEL:
			gapopen++; gapextend += (w - 1); alen += w;
			break;
		case OP_HARD_CLIP: case OP_SOFT_CLIP:
			qclip += w; qlen += w;
			break;
		default:                                             /* N, P, anything else */
			break;
		}
	}
	if (has_md) {                                            /* :200-219: kstrtok(md, "^0123456789") */
		/* a token is a maximal run of bytes outside the separator set; it counts when it does not begin the string and
		 * the byte in front of it is not '^' -- every byte of it (the loop at :214-216 is the inner one here) */
		size_t i = 0, n = strlen(md);
		while (i < n) {
			size_t j;
			if (md[i] == '^' || (md[i] >= '0' && md[i] <= '9')) { i++; continue; }
			j = i;
			while (j < n && !(md[j] == '^' || (md[j] >= '0' && md[j] <= '9'))) j++;
			if (i > 0 && md[i - 1] != '^') mismatch += (int32_t)(j - i);
			i = j;
		}
		match -= mismatch;                                   /* :218 */
	}
	out->match = match; out->mismatch = mismatch; out->gapopen = gapopen; out->gapextend = gapextend;
	out->query_length = qlen; out->query_clip = qclip; out->length = alen;
	out->edit = mismatch + qclip + gapopen + gapextend;      /* :228 */
}

/* htslib bam_endpos: pos + bam_cigar2rlen (M, D, N, =, X consume the reference), a length of 0 counts as 1 */
static int64_t ext_endpos(const orc_records *r, int64_t i) {
	int64_t rlen = 0;
	uint32_t k;
	if (!(r->flag[i] & F_UNMAP))
		for (k = r->cigar_off[i]; k < r->cigar_off[i + 1]; k++) {
			int op = (int)(r->cigar[k] & 0xf);
			if (op == OP_MATCH || op == OP_DEL || op == OP_REF_SKIP || op == OP_EQUAL || op == OP_DIFF) rlen += r->cigar[k] >> 4;
		}
	if (rlen == 0) rlen = 1;
	return (int64_t)r->pos[i] + rlen;
}

/* the records mSummarizeAlignments / ...Stats look at (msam_summary.c:56-66, :98-108) */
static int summary_takes(const orc_records *r, const uint32_t *target_len, uint32_t edge, int64_t i) {
	int64_t start, end;
	if (r->flag[i] & F_UNMAP) return 0;
	if (r->flag[i] & 0x100) return 0;                        /* BAM_FSECONDARY */
	start = r->pos[i];
	end = ext_endpos(r, i);
	if (start < (int64_t)edge || (int64_t)target_len[r->tid[i]] - end < (int64_t)edge) return 0;
	return 1;
}

static void summary_of(const orc_records *r, int64_t i, orc_ext_summary *a) {
	char *md = NULL;
	int has_md = (r->rflags[i] & ORC_HAS_MD) != 0;
	if (has_md) {
		uint32_t l = r->md_off[i + 1] - r->md_off[i];
		md = (char *)malloc((size_t)l + 1);
		memcpy(md, r->md + r->md_off[i], l);
		md[l] = 0;
	}
	orc_ext_summary_record(r->cigar + r->cigar_off[i], r->cigar_off[i + 1] - r->cigar_off[i], md, has_md, a);
	free(md);
}

int64_t orc_summary_table(const orc_records *r, const uint32_t *target_len, uint32_t edge, int64_t *sel, int32_t *vals) {
	int64_t i, n = 0;
	for (i = 0; i < r->n; i++) {
		orc_ext_summary a;
		if (!summary_takes(r, target_len, edge, i)) continue;
		summary_of(r, i, &a);
		sel[n] = i;
		vals[4 * n] = a.query_length;
		vals[4 * n + 1] = a.length + a.query_clip;           /* glocal_len :70 */
		vals[4 * n + 2] = a.match;
		vals[4 * n + 3] = a.edit;
		n++;
	}
	return n;
}

void orc_summary_stats(const orc_records *r, const uint32_t *target_len, uint32_t edge, int stats_type, int64_t *dist) {
	int64_t i;
	memset(dist, 0, 4097 * sizeof(int64_t));
	for (i = 0; i < r->n; i++) {
		orc_ext_summary a;
		uint32_t stats[4];
		int idx;
		if (!summary_takes(r, target_len, edge, i)) continue;
		summary_of(r, i, &a);
		stats[0] = (uint32_t)a.match;                        /* :110-113 */
		stats[1] = (uint32_t)(a.query_length - a.match);
		stats[2] = (uint32_t)a.edit;
		stats[3] = (uint32_t)(a.match - a.edit);
		idx = (int)stats[stats_type];                        /* :120-122 */
		if (idx > 4096) idx = 4096;
		if (idx < 0) idx = 0;
		dist[idx]++;
	}
}

int64_t orc_summary_count(const orc_records *r) {
	int64_t i, prev = -1, count = 0;
	for (i = 0; i < r->n; i++) {
		if (r->flag[i] & F_UNMAP) continue;                  /* :30 */
		/* prev_read starts as "": a first mapped record with an empty name would not count (:21, :32) */
		if (prev < 0 ? (r->qname_off ? r->qname_off[i + 1] > r->qname_off[i] : 1) : !same_name(r, i, prev)) count++;
		prev = i;
	}
	return count;
}
