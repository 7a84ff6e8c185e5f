/* msh_cli.h -- internals shared by the files of the command line (msh_common.c, msh_pipeline.c, msh_filter.c,
 * msh_profile.c, msh_coverage.c, msh_main.c, msh_dev.c). */
#ifndef MSH_CLI_H
#define MSH_CLI_H
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

#define COORD_ORDER_CHECK_RECORDS 100000
#define TIC double t0_ = now_s()
#define TOC(acc) do { double t1_ = now_s(); (acc) += t1_ - t0_; t0_ = t1_; } while (0)

#define MSX(call) do { if ((call) != MSX_OK) mDie("%s", msx_last_error(g_ctx)); } while (0)

/* One process, several GPUs: MSX_DEVICES="0,1,2,3" (device ids, one context and one device thread each; the decode
 * stage deals the batches of the one input to them and the writer puts filter's output back in input order --
 * SURVEY.md 8e).  Unset: one device (MSX_DEVICE, default 0). */
#define MSH_MAX_DEVICES 16

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

#define RB_REC(b, i) ((b)->base + (b)->rec_off[i] + 4)
#define RB_LEN(b, i) ((b)->rec_off[(i) + 1] - (b)->rec_off[i] - 4)

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
#define MSH_POOL_MAX 128
#define PQ_END (-1)                     /* queue item: end of the stream (one per consumer) */

typedef struct {
	pthread_mutex_t mu;
	pthread_cond_t cv;
	int item[2 * PIPE_SLOTS_MAX + 4], n;
} pq;

#define PQ_NONE (-2)                    /* pq_try_pop: the queue is empty */

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
	size_t pin_rec, pin_cig, pin_md, pin_grp;   /* what of the SoA arrays is page-locked (records, CIGAR words, MD bytes, pools) */
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
	msx_event *ev_out;         /* ... are there once this has been waited for (= ev_out_by[the context that worked on the batch]) */
	msx_event *ev_out_by[MSH_MAX_DEVICES];
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
	int comp_ramp;                 /* ... batches from this one on take twice comp_blocks, in larger buffers (0: none do) */
	size_t big_rcap;
	int big_from;                  /* the pin thread makes the larger buffers when this many batches have been filled */
	uint8_t *big_rbuf[PIPE_SLOTS_MAX];
	int comp_mode, comp_blocks;    /* ... and inflated there as well: the decode stage only copies the blocks' payloads */
	size_t ocap_cfg;
	/* output buffers are a pool of their own: a slot is the decode stage's unit (a buffer of compressed blocks), and the
	 * decoder must not run out of slots because the writer still holds the outputs of earlier batches */
	uint8_t *ob[PIPE_OBUFS];
	size_t ob_cap[PIPE_OBUFS];
	/* batch seq takes buffer seq % PIPE_OBUFS, and only once it is among the PIPE_OBUFS batches the writer takes next: handed
	 * out first come, first served, several device threads could take every buffer for batches BEHIND the one the writer is
	 * waiting for, whose thread then finds none (a hang of the three-context test) */
	pthread_mutex_t ob_mu;
	pthread_cond_t ob_cv;
	int ob_state[PIPE_OBUFS];      /* 0: not page-locked yet, 1: free, 2: holds a batch's output */
	size_t ob_written;             /* batches the writer is done with */
	int raw_started, raw_done;
	/* several contexts: the stream's carry behind raw batch seq - 1, handed from the context that finished it to the one
	 * that walks batch `baton_seq` (msx_unpack_carry -> msx_unpack_seed) */
	pthread_mutex_t baton_mu;
	pthread_cond_t baton_cv;
	size_t baton_seq;             /* the raw batch whose predecessor's carry is in baton */
	int baton_fresh;              /* ... and has not been taken: the first raw batch is seeded by the decode stage instead */
	kstr baton;
	char baton_name[256];
	int baton_has_name;
	int first_state, first_slot;   /* 0: batch 0 not decoded yet; 1: it is, in slot first_slot; 2: the input holds no record (first_mu) */
	int out_opened;                /* the preflight has passed and the output is open (first_mu) */
	pthread_mutex_t first_mu;
	pthread_cond_t first_cv;
	int with_obuf;             /* filter: there are output buffers as well */
	int pin_started, pin_quit, pin_joined;       /* pin_joined: 1 while one thread joins them, 2 once joined */
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
/* The batches behind this one, as far as the decode stage has them ready (two at most): their blocks start their way up
 * and are inflated beside the work on this batch.  `ahead` is the caller's queue of slots taken off q_dev ahead of their
 * turn (in order; an end-of-stream token or a slot that cannot be sent ahead closes it). */
typedef struct { int item[2], n, closed; } ahead_q;



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

/* msh_common.c */
extern __thread msx_ctx *g_ctx;
extern double t_decode, t_upload, t_gpu, t_fetch, t_write;
extern double g_t_main;
extern int g_dist;
/* msh_common.c */
double now_s(void);
void fast_exit(void);
int dist_world(void);
int dist_rank(void);
void ctx_open_dev(int id);
void ctx_open(void);
int device_list(int *ids);
void runtime_warmup_start(void);    /* the HIP runtime's start-up on a thread of its own, beside the opening of the input */
void runtime_warmup_join(void);     /* before main() returns (not needed before _exit): see msh_common.c */
size_t batch_target(void);
char *command_line(int argc, char *argv[]);
void rb_reserve(rbatch *b);
void rb_clear(rbatch *b);
void rb_mark_group(rbatch *b);
void rb_append(rbatch *b, const uint8_t *r, size_t len, int want_stats);
void rb_host_view(rbatch *b, msx_batch *h, int with_groups);
void qn_format(const qn_result *r, char *buf, size_t n);
qn_result qn_check(const msh_hdr *hdr, const rbatch *first);
/* msh_pipeline.c */
void fill_filter_batch(reader *rd, rbatch *b, size_t target, int pools, int want_stats);
void fill_batch_bulk(reader *rd, rbatch *b, size_t target, int mode, int want_stats);
void pq_push(pq *q, int v);
int pq_pop(pq *q);
int pq_try_pop(pq *q);
void *xmalloc(size_t n);
void mem_report(const char *tag);   /* MSX_TIMING=2 */
void pipe_init(pipe_t *P, msh_in *in, int mode, int want_stats, int n_consumers);
uint8_t *io_alloc(size_t bytes);
void io_populate(uint8_t *p, size_t bytes);
void pin_start(pipe_t *P, int with_obuf);
void pin_join(pipe_t *P);
void pin_wait(pipe_t *P, pslot *s);
int msh_trace_on(void);
#define MSH_TRACE(...) do { if (msh_trace_on()) { fprintf(stderr, "# trace %.3f: ", now_s()); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); } } while (0)
int ob_acquire(pipe_t *P, size_t seq);
void ob_release(pipe_t *P, int i);
void ob_written(pipe_t *P, size_t seq);
void unpack_slot_enqueue(pipe_t *P, pslot *s, msx_unpack *unpack, const msx_unpack_params *up);
void unpack_slots_ahead(pipe_t *P, msx_unpack *unpack, ahead_q *A);
int ahead_pop(ahead_q *A);
void unpack_slot_finish(pipe_t *P, pslot *s, msx_unpack *unpack, const msx_unpack_params *up, msx_unpack_result *ur, msx_batch *db);
void pipe_enable_raw(pipe_t *P, int with_obuf);
void *pipe_decode_thread(void *arg);
void pipe_pin_slot(pipe_t *P, pslot *s);
/* msh_profile.c */
void prof_opts_derive(prof_opts *o);
void prof_features(const prof_opts *o, const msh_hdr *hdr, prof_feat *F);
void gz_member(const kstr *in, kstr *out);
void fd_write_all(int fd, const void *p, size_t n);
void profile_report(const prof_opts *o, const prof_feat *F, const msx_profile_stats *st, double *row,
                           const qn_result *qn, const char *cl);
void profile_combine_and_finalize(msx_ctx **ctx, msx_profile **prof, int n_dev, int share_type, double *row,
                                         msx_profile_stats *st);
/* msh_filter.c */
void *filter_dev_thread(void *arg);
int msam_filter_main(int argc, char *argv[]);
/* msh_profile.c */
int msam_profile_main(int argc, char *argv[]);
/* msh_coverage.c */
int msam_coverage_main(int argc, char *argv[]);
/* msh_main.c */
int usage(FILE *out);
/* msh_dev.c */
int recode_main(int argc, char *argv[]);
int synth_main(int argc, char *argv[]);
int pipetest_main(int argc, char *argv[]);
int digest_main(int argc, char *argv[]);
int restream_main(int argc, char *argv[]);
int rawtest_main(int argc, char *argv[]);
#endif
