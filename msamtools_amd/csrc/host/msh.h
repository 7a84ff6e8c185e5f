/*
 * msh.h -- host side (C) of the MI355X-native msamtools filter/profile path:
 * BGZF/BAM and SAM-text input and output, QNAME pools, SoA batch packing, and
 * the `msamtools filter` / `msamtools profile` command lines.  Counterpart of
 * the reference's msam_helper.c + htslib usage (msam_helper.c:196-293) and of
 * the drivers msam_filter.c:267-507 / msam_profile.c:503-1015; all compute
 * goes through the C ABI of include/msamtools_amd.h.
 */
#ifndef MSH_H
#define MSH_H

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../../include/msamtools_amd.h"

#define PROGRAM "msamtools"
#ifndef MSH_VERSION
#define MSH_VERSION "1.1.3-amd"
#endif
#ifndef MSH_GIT_COMMIT
#define MSH_GIT_COMMIT "unknown"
#endif

/* ---- errors (mCommon.c:3-31) ---------------------------------------------- */
void mDie(const char *fmt, ...);    /* "Fatal Error: ..." on stderr, exit 1 */
void mQuit(const char *fmt, ...);   /* text on stderr, exit 1               */
#include <pthread.h>
extern pthread_t msh_main_thread;   /* set by main(): mDie on any other thread leaves with _exit */
extern int msh_main_thread_set;

/* ---- growable byte string -------------------------------------------------- */
typedef struct {
	char *s;
	size_t l, m;
} kstr;
void ks_reserve(kstr *k, size_t extra);
void msh_huge_hint(void *p, size_t n);       /* a heap block of 8 MB and more: its 2 MB-aligned part advised to huge pages */
void ks_put(kstr *k, const void *p, size_t n);
void ks_puts(kstr *k, const char *s);
void ks_putc(kstr *k, int c);
void ks_printf(kstr *k, const char *fmt, ...);

/* ---- header ----------------------------------------------------------------- */
typedef struct {
	kstr text;              /* SAM header text ('\n'-terminated lines)  */
	int32_t n_targets;
	char **target_name;
	uint32_t *target_len;
} msh_hdr;
/* value of @HD SO: or NULL (caller frees) -- sam_hdr_find_tag_hd(h,"SO") */
char *msh_hdr_sort_order(const msh_hdr *h);
/* sam_hdr_add_pg(): unique ID, PP chained to every existing @PG chain end */
void msh_hdr_add_pg(kstr *text, const char *name, const char *vn, const char *cl, const char *ds);
int32_t msh_hdr_name2tid(const msh_hdr *h, const char *name);

/* ---- BAM record accessors (SAMv1 4.2; record bytes WITHOUT block_size) ------ */
static inline int32_t le32(const uint8_t *p) { return (int32_t)((uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24); }
static inline uint32_t le16(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8; }
#define REC_TID(r) le32((r) + 0)
#define REC_POS(r) le32((r) + 4)
#define REC_LQNAME(r) ((r)[8])
#define REC_MAPQ(r) ((r)[9])
#define REC_NCIGAR(r) le16((r) + 12)
#define REC_FLAG(r) le16((r) + 14)
#define REC_LSEQ(r) le32((r) + 16)
#define REC_QNAME(r) ((const char *)(r) + 32)
#define REC_CIGAR(r) ((r) + 32 + REC_LQNAME(r))
#define REC_AUX(r) (REC_CIGAR(r) + 4 * REC_NCIGAR(r) + (REC_LSEQ(r) + 1) / 2 + REC_LSEQ(r))
/* bam_aux_get: first tag match, pointer to the type byte or NULL */
const uint8_t *msh_aux_get(const uint8_t *rec, size_t len, const char tag[2]);
/* bam_aux2i: c C s S i I, 0 for anything else */
int64_t msh_aux2i(const uint8_t *type_ptr);
const uint8_t *msh_real_cigar(const uint8_t *r, size_t len, uint32_t *n_out, const uint8_t **cg_tag);   /* the CIGAR to compute from: the record's own, or its CG:B:I tag's (htslib: bam_tag2cigar) */
/* size in bytes of the aux field starting at its type byte (type + payload) */
size_t msh_aux_size(const uint8_t *type_ptr, const uint8_t *end);
void msh_rec_check(const uint8_t *r, size_t len);       /* dies unless the fields the record announces fit its length */

/* ---- input: BAM (BGZF, multi-threaded inflate) or SAM text, file or "-" ---- */
typedef struct msh_in msh_in;
msh_in *msh_open(const char *path);             /* format is auto-detected, as htslib does */
const msh_hdr *msh_header(msh_in *in);
int msh_read(msh_in *in, kstr *rec);            /* 0 = record in rec (l = length), -1 = EOF */
void msh_close(msh_in *in);

/* bulk access for BAM input: a contiguous span of inflated bytes holding whole records
 * (each with its 4-byte block_size prefix) and possibly one trailing partial record */
int msh_is_bam(const msh_in *in);
int64_t msh_in_bytes(const msh_in *in);    /* size of a regular input file; -1 otherwise */
int msh_span_fill(msh_in *in);                        /* inflate the next batch of blocks; 0 at EOF */
const uint8_t *msh_span(msh_in *in, size_t *len);     /* unconsumed bytes; invalidated by msh_span_fill */
void msh_span_consume(msh_in *in, size_t n);
/* pipelined reader: inflate the next batch of blocks onto the end of a caller-owned buffer; 0 at EOF */
size_t msh_inflate_append(msh_in *in, uint8_t **buf, size_t *len, size_t *cap);
void msh_release_input(msh_in *in);   /* at the end of the input: the mapping's pages leave the page tables */
int msh_raw_append(msh_in *in, uint8_t *buf, size_t cap, size_t *len, msx_bgzf_block *blk, int *n, int max_blocks, size_t *out_total);
void msh_inflate_table(const uint8_t *comp, const msx_bgzf_block *blk, int n, uint8_t *out);
void msh_inflate_limit(int blocks);      /* at most this many BGZF blocks per msh_inflate_append call (0: the default batch) */
/* the same for SAM text: the next chunk of lines parsed on all threads into BAM records; at most MSH_SAM_APPEND_MAX bytes */
size_t msh_sam_append(msh_in *in, uint8_t **buf, size_t *len, size_t *cap);
extern void (*msh_exit_hook)(int rc);     /* set by msh_main.c under MSX_DETACH=1; called before every _exit */
int msh_idle_ms(void);                       /* MSX_IDLE_MS: a batch ends when the producer has been quiet this long (default 50; 0: never) */
int msh_input_ready(msh_in *in, int ms);     /* the next append can make progress without waiting longer than ms for the producer */
#define MSH_SAM_APPEND_MAX ((size_t)48 << 20)

/* ---- threads ------------------------------------------------------------------ */
int msh_threads(void);                                 /* MSX_THREADS or the online CPU count, <= 64 */
typedef void (*msh_pf)(void *arg, int tid, int nth);
void msh_parallel(int nth, msh_pf fn, void *arg);      /* fn(arg, tid, nth) on nth threads */
/* raw DEFLATE of one BGZF block, whole input to whole output (msh_inflate.c); 0: not vouched for, give it to zlib */
int msh_fast_inflate(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len);
uint32_t msh_crc32(const void *p, size_t n);           /* CRC-32 of a BGZF payload (carry-less multiplication where the CPU has it) */

/* ---- output ------------------------------------------------------------------ */
enum { MSH_OUT_SAM = 0, MSH_OUT_SAM_HDR = 1, MSH_OUT_BAM = 2, MSH_OUT_UBAM = 3 };
typedef struct msh_out msh_out;
msh_out *msh_out_open(FILE *fp, int mode, const msh_hdr *h, const char *hdr_text);
void msh_write(msh_out *o, const uint8_t *rec, size_t len);
/* records base + rec_off[idx[k]] (+4 = past the block_size prefix), k = 0..n-1; multi-threaded */
void msh_write_many(msh_out *o, const uint8_t *base, const size_t *rec_off, const int32_t *idx, size_t n);
/* a ready-made record stream ([block_size | record] back to back, as msx_unpack_emit returns it) */
void msh_write_stream(msh_out *o, const uint8_t *bytes, size_t n);
/* finished BGZF blocks, back to back (framed on the device): written as they are */
void msh_write_framed(msh_out *o, const uint8_t *blocks, size_t n);
void msh_out_drain(msh_out *o);
void msh_out_flush(msh_out *o);          /* the open block and stdio's buffer go out now (the writer is about to wait for input) */     /* before dying: what was handed to the writer is written out */
void msh_out_close(msh_out *o);

/* SAM text <-> BAM record */
void msh_sam_format(const msh_hdr *h, const uint8_t *rec, size_t len, kstr *line);   /* no trailing '\n' */
void msh_sam_parse(const msh_hdr *h, char *line, kstr *rec);                        /* line is modified */

/* ---- ordered string set with the key order of the reference's hash table (msh_genome.c) ---- */
typedef struct msh_keyset msh_keyset;
msh_keyset *msh_keyset_new(void);
int32_t msh_keyset_put(msh_keyset *k, const char *key, int32_t val);   /* id (order of first insertion); sets the value */
int32_t msh_keyset_find(const msh_keyset *k, const char *key);          /* id or -1 */
int32_t msh_keyset_size(const msh_keyset *k);
int32_t msh_keyset_walk(const msh_keyset *k, int32_t pos);              /* id at position pos of the key walk */
const char *msh_keyset_key(const msh_keyset *k, int32_t id);
int32_t msh_keyset_value(const msh_keyset *k, int32_t id);
void msh_keyset_free(msh_keyset *k);
/* profile --genome=<file> (msam_profile.c:760-852): fmap[n_targets], feature names and summed lengths */
int32_t *msh_genome_map(const char *path, const msh_hdr *h, int32_t *n_features, char ***names, uint32_t **lens);

/* ---- subcommands ------------------------------------------------------------- */
int msam_filter_main(int argc, char *argv[]);
int msam_profile_main(int argc, char *argv[]);
int msam_coverage_main(int argc, char *argv[]);
int msam_summary_main(int argc, char *argv[]);

#endif
