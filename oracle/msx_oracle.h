/*
 * msx_oracle.h -- CPU oracle for the msamtools filter -> profile hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference
 * algorithm (arumugamlab/msamtools v1.1.3), one function per reference
 * function, each citing the reference file:line it follows.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product (msamtools_amd/, include/) never links, imports or calls it.
 *
 * Parity pinning: the reference itself cannot be built in this image (it needs
 * htslib 1.24, argtable2 and autotools, none present; writing stand-in headers
 * is not allowed), so the oracle is pinned against the reference's OWN golden
 * expectations: every record list / count / abundance asserted by
 * tests/test_filter.sh, test_besthit.sh, test_profile.sh, test_integration.sh
 * and test_coverage.sh on the reference's fixtures (tests/golden/), plus the
 * SURVEY section 8c known-answer table for tests/tiny_aln.bam.
 * The one exception: zoeTools.c is self-contained, so `make ref` compiles it in
 * place into oracle/_ref/libzoe_ref.so; orc_key_order() is pinned by key-order
 * vectors that library produced (tests/golden/genome_order_vectors.json).
 * Third-party boundary restated here: htslib 1.24 kstrtok() (kstring.c) and
 * bam_aux2i() truncation semantics -- see orc_md_edit().
 * Not pinned by any reference test (stated in DESIGN.md): MD strings with '^'
 * deletions, CIGAR ops >= 9, records with both/neither mate bit in a paired
 * pool, int32 overflow in the threshold products.
 */
#ifndef MSX_ORACLE_H
#define MSX_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* per-record aux presence bits (same encoding as include/msamtools_amd.h) */
#define ORC_HAS_MD 1u
#define ORC_HAS_NM 2u
#define ORC_HAS_AS 4u

/* mAlignmentSummary, mBamVector.h:38-47 (gapopen/gapextend are never set on
 * the filter path and are omitted). */
typedef struct {
	int32_t match;
	int32_t mismatch;
	int32_t length;
	int32_t query_length;
	int32_t query_clip;
	int32_t edit;
} orc_summary;

/* A stream of alignment records in input order, structure-of-arrays.
 * QNAMEs are either real strings (qname_off/qname) or, for synthetic data
 * with no strings, integer ids (name_id): two records have "the same QNAME"
 * iff their strings (or ids) are equal. */
typedef struct {
	int64_t         n;
	const uint32_t *qname_off;  /* [n+1] byte offsets into qname, or NULL    */
	const char     *qname;      /* concatenated names, no terminators        */
	const int32_t  *name_id;    /* [n] used when qname_off == NULL           */
	const uint16_t *flag;       /* [n] BAM FLAG                              */
	const uint8_t  *rflags;     /* [n] ORC_HAS_* bits                        */
	const int32_t  *tid;        /* [n]                                       */
	const int32_t  *pos;        /* [n] 0-based leftmost position             */
	const uint32_t *cigar_off;  /* [n+1] index into cigar                    */
	const uint32_t *cigar;      /* packed BAM CIGAR: len<<4 | op             */
	const uint32_t *md_off;     /* [n+1] byte offsets into md                */
	const uint8_t  *md;         /* MD:Z payloads, no terminators             */
	const int32_t  *nm;         /* [n] (int32_t) bam_aux2i(NM), 0 if absent  */
	const int32_t  *as;         /* [n] (int32_t) bam_aux2i(AS), 0 if absent  */
} orc_records;

/* msam_filter.c:420-457 (thresholds) and :497 (switches) */
typedef struct {
	int32_t min_length;   /* global->MIN_LENGTH = -l                          */
	int32_t ppt;          /* global->PPT = 10*-p or --ppt                     */
	int32_t max_clip;     /* global->MAX_CLIP = 100 - (-z), 100 when no -z    */
	int32_t rescore;      /* --rescore                                        */
	int32_t invert;       /* -v                                               */
	int32_t keep_unmapped;/* -k                                               */
	int32_t besthit;      /* --besthit                                        */
	int32_t uniqhit;      /* --uniqhit                                        */
} orc_filter_params;

#define ORC_OK              0
#define ORC_ERR_NO_MD_NM    1  /* msam_filter.c:150-152 */
#define ORC_ERR_NO_AS       2  /* msam_filter.c:219-221 */
#define ORC_ERR_NO_FILTER   3  /* msam_filter.c:82-84   */

#define ORC_MULTI_ADD_ALL            1  /* msam_profile.c:5-8 */
#define ORC_MULTI_SHARE_EQUAL        2
#define ORC_MULTI_SHARE_PROPORTIONAL 3
#define ORC_MULTI_IGNORE             4

/* --- per-record kernels -------------------------------------------------- */

/* mBamVector.c:23-38 */
void orc_cigar2details(const uint32_t *cigar, uint32_t n_cigar,
                       int32_t *alen, int32_t *qlen, int32_t *qclip);
/* mBamVector.c:40-133; md == NULL means no MD tag */
void orc_get_summary(const uint32_t *cigar, uint32_t n_cigar, const char *md,
                     orc_summary *summary);
/* msam_filter.c:31-63,79-85: returns 1 when the record FAILS the active set */
int  orc_filter_fails(const orc_summary *a, const orc_filter_params *p);
/* msam_filter.c:145-157 for every record: fills length/qlen/qclip/edit arrays
 * (any may be NULL).  status[i]: 0 ok, 1 neither MD nor NM. */
void orc_aln_stats(const orc_records *r, int32_t *length, int32_t *qlen,
                   int32_t *qclip, int32_t *edit, uint8_t *status);

/* --- filter stream (mFilterFileWrapper + mFilterFile + writers) ---------- */

/* msam_filter.c:65-263.  emit_idx[cap >= n] receives the indices of the
 * records written, in output order; as_out[n] (may be NULL) receives the AS
 * value each record carries after --rescore (input AS otherwise).
 * Returns ORC_OK or an ORC_ERR_*; *err_record = offending record index. */
int orc_filter(const orc_records *r, const orc_filter_params *p,
               int32_t *emit_idx, int64_t *n_emit, int32_t *as_out,
               int64_t *err_record);

/* --- profile ------------------------------------------------------------- */

typedef struct {
	uint32_t insert_count;        /* return value of msam_profile.c:204     */
	uint32_t uniq_mapper_count;   /* msam.h:36-38                           */
	uint32_t multi_mapper_count;
	uint32_t purged_insert_count;
	int32_t  iterations;          /* last k printed at msam_profile.c:381   */
	int32_t  converged;
	double   last_delta;
} orc_profile_stats;

/* msam_profile.c:23-243 (count) + :248-425 (abundance), one sample.
 * sel: optional list of n_sel record indices (the filter output stream, in
 * output order); NULL = every record of r in order.
 * fmap: tid -> feature (NULL = identity).  abundance[n_features] receives
 * m->elem[row][1..], i.e. without the Unknown column.
 * ui_out (optional, [n_features]) receives ui_insert_count before reset. */
int orc_profile(const orc_records *r, const int32_t *sel, int64_t n_sel,
                const int32_t *fmap, int32_t n_features, int32_t share_type,
                double *abundance, uint32_t *ui_out, orc_profile_stats *stats);

/* msam_profile.c:858-975 post-processing on [Unknown, features...].
 * values[n_features+1] in: values[0] ignored, values[1..] = abundance.
 * unit_type: 1 rel, 2 fpkm, 3 tpm, 4 ab (msam_profile.c:10-13).
 * total_inserts <= 0 means --total not given.  mincount < 0 means not given. */
void orc_profile_finish(double *values, int32_t n_features,
                        const uint32_t *feature_len, int32_t unit_type,
                        int32_t length_normalize, int32_t total_inserts,
                        int32_t mincount, int32_t share_type,
                        const orc_profile_stats *stats,
                        double *purged_inserts, double *effective_inserts);

/* --- coverage (msam_coverage.c:33-87, 106-139) --------------------------- */

/* cov_off[n_targets+1] gives each target's offset in cov (= prefix sum of
 * target_len); cov must be zeroed by the caller. */
void orc_coverage(const orc_records *r, const int64_t *cov_off,
                  int32_t n_targets, int32_t *cov);

/* --- feature order of `profile --genome` (msam_profile.c:779-805, zoeTools.c:202-363) --- */

/* names[n] are inserted one by one into the reference's string hash table
 * (zoeSetHash); order[] receives, for every position of zoeKeysOfHash, the
 * index into names[] of the first occurrence of that key; returns the number
 * of distinct keys.  Pinned by tests/golden/genome_order_vectors.json, which
 * was produced by the reference's own zoeTools.c. */
int32_t orc_key_order(const char *const *names, int32_t n, int32_t *order);

#ifdef __cplusplus
}
#endif

/* --- `summary` (msam_summary.c, mBamVector.c:135-236) ------------------------ */

/* bam_get_extended_summary (mBamVector.c:135-236): the summary subcommand's own record walk.  MD only -- NM is never
 * looked at -- and the MD count is the inner loop's (one per mismatching base). */
typedef struct {
	int32_t match, mismatch, gapopen, gapextend, query_length, query_clip, length, edit;
} orc_ext_summary;
void orc_ext_summary_record(const uint32_t *cigar, uint32_t n_cigar, const char *md, int has_md, orc_ext_summary *out);

/* mSummarizeAlignments (msam_summary.c:42-74): for every record the table prints -- mapped, not secondary, not within
 * `edge` bases of either end of its target (bam_endpos: pos + reference length of the CIGAR, + 1 when that is 0) --
 * its number in sel[] and {query_length, glocal_len, match, edit} in vals[4 * k ..]; returns how many.
 * target_len[tid] as the header gives it. */
int64_t orc_summary_table(const orc_records *r, const uint32_t *target_len, uint32_t edge, int64_t *sel, int32_t *vals);

/* mSummarizeAlignmentsStats (msam_summary.c:76-135): dist[0 .. 4096] of the chosen statistic
 * (0 mapped, 1 unmapped, 2 edit, 3 score; values above 4096 counted at 4096, "negative" ones at 0). */
void orc_summary_stats(const orc_records *r, const uint32_t *target_len, uint32_t edge, int stats_type, int64_t *dist);

/* mCountInserts (msam_summary.c:19-40): mapped records whose QNAME differs from the previous mapped record's */
int64_t orc_summary_count(const orc_records *r);

#endif
