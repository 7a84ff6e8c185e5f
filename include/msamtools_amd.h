/*
 * msamtools_amd.h -- C ABI of the MI355X-native msamtools filter -> profile
 * hot path (libmsamtools_amd.so).
 *
 * Plain C: pointers, sizes and PODs only; no HIP, torch or C++ types cross
 * this boundary.  Every entry point names the reference seam it replaces
 * (file:line under arumugamlab/msamtools v1.1.3).  The reference has no FFI
 * or plugin interface; its seams for this path are the function pointers and
 * per-pool calls inside msam_filter.c / msam_profile.c, so the
 * binding a maintainer would add is a direct C call -- see INTEGRATION.md.
 *
 * Division of labour
 *   host (C):  BGZF/BAM or SAM decode, the QNAME string compares that define
 *              pools (group_off), SoA packing, record emission, profile text.
 *   device:    everything per record / per pool / per feature:
 *              CIGAR+MD walk, -l/-p/-z predicates, --rescore, pool membership
 *              incl. the unmapped/-k/-v rules, --besthit/--uniqhit selection
 *              and output order, insert counting, multi-mapper lists,
 *              proportional sharing iterations, per-base coverage pile-up.
 *
 * Error convention: every int-returning function returns MSX_OK (0) or an
 * MSX_ERR_* code; msx_last_error(ctx) holds the text.  Data errors carry the
 * reference's own message (the caller prints "Fatal Error: <text>" to stderr
 * and exits 1, as mDie does: mCommon.c:22-31).
 */
#ifndef MSAMTOOLS_AMD_H
#define MSAMTOOLS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSX_ABI_VERSION 7

/* ---- status codes ------------------------------------------------------- */
#define MSX_OK              0
#define MSX_ERR_NO_MD_NM    1   /* msam_filter.c:150-152 (mDie)                 */
#define MSX_ERR_NO_AS       2   /* msam_filter.c:219-221 (mDie)                 */
#define MSX_ERR_NO_FILTER   3   /* msam_filter.c:82-84   (mDie)                 */
#define MSX_ERR_SHARE_TYPE  4   /* msam_profile.c:122-124, :188-190 (mDie)      */
#define MSX_ERR_HIP       (-10) /* HIP runtime failure, text has the HIP error  */
#define MSX_ERR_ARG       (-11) /* bad argument                                  */
#define MSX_ERR_NOMEM     (-12) /* device or host allocation failed              */
#define MSX_ERR_NO_DEVICE (-13) /* no usable gfx950 device: there is NO CPU fallback */
#define MSX_ERR_DIST      (-14) /* RCCL / rendezvous failure                      */
#define MSX_ERR_INFLATE   (-15) /* a BGZF block the device inflater refused: inflate the batch on the host */

/* ---- per-record aux presence bits (msx_batch.rflags) --------------------- */
#define MSX_HAS_MD 1u   /* bam_aux_get(b,"MD") != NULL (msam_filter.c:146)      */
#define MSX_HAS_NM 2u   /* bam_aux_get(b,"NM") != NULL (msam_filter.c:149)      */
#define MSX_HAS_AS 4u   /* bam_aux_get(b,"AS") != NULL (msam_filter.c:219)      */

/* ---- multi-mapper share types (msam_profile.c:5-8) ----------------------- */
#define MSX_MULTI_ADD_ALL            1
#define MSX_MULTI_SHARE_EQUAL        2
#define MSX_MULTI_SHARE_PROPORTIONAL 3
#define MSX_MULTI_IGNORE             4

typedef struct msx_ctx msx_ctx;
typedef struct msx_profile msx_profile;

/*
 * A batch of alignment records in input order, structure-of-arrays.
 * Replaces the stream of bam1_t that mSamRead() hands to mFilterFile /
 * mEstimateInsertCountOnFile (msam_helper.c:246-268, msam_filter.c:119,
 * msam_profile.c:222).  All pointers of one batch live in the same memory
 * space: host (for msx_batch_upload) or device (for the compute entry points).
 *
 * group_off: the pools.  group_off[g] is the index of the first record of pool
 * g; group_off[n_groups] == n_records.  The host computes it with the exact
 * string rule of the loop it feeds:
 *   filter  (msam_filter.c:120-125,170): a pool closes when a record's QNAME
 *           differs from the QNAME of the last MAPPED record seen;
 *   profile (msam_profile.c:223-232): records with tid == -1 are transparent;
 *           a pool closes when a QNAME differs from the previous such record.
 * For a QNAME-grouped file both rules give "one pool per QNAME run".
 * group_off may be NULL for plain -l/-p/-z filtering (output does not depend
 * on pools: mWriteBamPool, mBamVector.c:343-348).
 */
typedef struct msx_batch {
	int64_t         n_records;
	int64_t         n_groups;
	const uint16_t *flag;       /* [n] bam1_core_t.flag                          */
	const uint8_t  *rflags;     /* [n] MSX_HAS_* bits                            */
	const int32_t  *tid;        /* [n] bam1_core_t.tid                           */
	const int32_t  *pos;        /* [n] bam1_core_t.pos (coverage only; may be NULL) */
	const uint32_t *cigar_off;  /* [n+1] offsets into cigar                      */
	const uint32_t *cigar;      /* bam_get_cigar(): len<<4|op                    */
	const uint32_t *md_off;     /* [n+1] byte offsets into md                    */
	const uint8_t  *md;         /* MD:Z payloads back to back, no terminators    */
	const int32_t  *nm;         /* [n] (int32_t) bam_aux2i(NM); 0 when absent    */
	const int32_t  *as;         /* [n] (int32_t) bam_aux2i(AS); 0 when absent    */
	const uint32_t *group_off;  /* [n_groups+1], see above                       */
	const uint64_t *qname_hash; /* [n] optional, not read by the kernels (shard routing) */
	int32_t         pool_rule;  /* MSX_POOLS_PROFILE or MSX_POOLS_FILTER: which loop's rule group_off follows.
	                               Read by the profile entry points only, see msx_profile_accumulate. */
	int32_t         reserved_;
} msx_batch;
#define MSX_POOLS_PROFILE 0   /* msam_profile.c:223-232 (or any input where one pool = one QNAME run)      */
#define MSX_POOLS_FILTER  1   /* msam_filter.c:120-125,170: a record of another name closes the pool, but only
                                 a MAPPED record renames the read being collected                            */

/* Thresholds and switches of `msamtools filter`, already validated/derived as
 * msam_filter.c:420-457 does (PPT = 10*-p or --ppt; MAX_CLIP = 100 - -z). */
typedef struct msx_filter_params {
	int32_t min_length;     /* global->MIN_LENGTH                              */
	int32_t ppt;            /* global->PPT                                     */
	int32_t max_clip;       /* global->MAX_CLIP (100 when -z absent)           */
	int32_t rescore;        /* --rescore                                       */
	int32_t invert;         /* -v                                              */
	int32_t keep_unmapped;  /* -k                                              */
	int32_t besthit;        /* --besthit                                       */
	int32_t uniqhit;        /* --uniqhit                                       */
	int32_t fatal_pool_partial;
	/* 0 in normal use.  1: a best-hit pool that holds a participating record without AS keeps what the reference had
	 * written of it when it died there (msam_filter.c:219-221 inside :247-263): the READ1 winners if only the READ2 pass
	 * meets such a record, nothing otherwise.  The call still fails with MSX_ERR_NO_AS; msx_filter_status.n_emit and
	 * emit_idx are valid all the same -- the caller that has cut the batch behind the offending pool writes them and
	 * then dies with the reference's message. */
} msx_filter_params;

/* Device output buffers of one filter call, caller-allocated (device memory).
 * keep[i]: 0 = not written; 1 = written in the READ1/unpaired pass;
 *          2 = written in the READ2 pass (msam_filter.c:247-263).
 * emit_idx: indices of the records written, in the reference's output order
 *          (pool by pool; within a pool all pass-1 records, then pass-2).
 * as_out:  AS each record carries on output (only written with --rescore,
 *          msam_filter.c:160-168; may be NULL otherwise). */
typedef struct msx_filter_out {
	uint8_t *keep;       /* [n_records]                                        */
	int32_t *emit_idx;   /* [n_records] capacity; may be NULL if not wanted    */
	int32_t *as_out;     /* [n_records] or NULL                                */
} msx_filter_out;

/* Host-side result of msx_filter_finish(). */
typedef struct msx_filter_status {
	int64_t n_emit;      /* number of valid entries in emit_idx / kept records */
	int64_t err_record;  /* first offending record for MSX_ERR_NO_MD_NM/NO_AS  */
} msx_filter_status;

/* Counters of one profile (msam.h:36-38 + return value of msam_profile.c:204) */
typedef struct msx_profile_stats {
	uint32_t insert_count;        /* "Mapped inserts"                          */
	uint32_t uniq_mapper_count;
	uint32_t multi_mapper_count;
	uint32_t purged_insert_count; /* msam_profile.c:394-405                    */
	int32_t  iterations;          /* last k reached by the loop at :331        */
	int32_t  converged;           /* 1 if DELTA^2 < 1e-10 was reached (:383)   */
	double   delta[20];           /* delta[k] = DELTA^2 printed at :381, k>=1  */
} msx_profile_stats;

/* ---- context ------------------------------------------------------------- */

/* The HIP runtime's own start-up (driver open, device enumeration, the device's primary context: 60-90 ms) without a
 * msx_ctx: a caller that has host-side work to do before it needs the device -- the command line parses a header of a
 * million @SQ lines first (msam_helper.c:153-164 reads it before anything else) -- starts this on a helper thread at
 * once; the msx_ctx_create that follows finds the runtime up.  Since round 5 it also has the library's code objects loaded
 * onto the device (the runtime loads a module when one of its kernels is first asked for: a few milliseconds each, otherwise paid
 * by a command's first batches; MSX_NO_MODULE_WARMUP=1: as before).  Returns MSX_OK or what msx_ctx_create would fail with. */
int  msx_runtime_warmup(int device_id);

/* Binds one GPU (one ctx per GPU / per rank).  Fails with MSX_ERR_NO_DEVICE if
 * there is no gfx950 device: the library has no CPU path. */
int  msx_ctx_create(msx_ctx **ctx, int device_id);
/* Side lanes: two more streams of the context on which a step's independent scans and compactions overlap (made when a step
 * first uses them).  They pay for steps over ~10^7 records and more; a caller that feeds batches of a million records -- the
 * command line -- turns them off (on = 0) and saves their creation and a fork / join per batch.  Results do not depend on it. */
int  msx_ctx_set_lanes(msx_ctx *ctx, int on);
void msx_ctx_destroy(msx_ctx *ctx);
const char *msx_last_error(const msx_ctx *ctx);  /* ctx may be NULL: last create error */
int  msx_abi_version(void);
/* Test aid (no counterpart in the reference).  With MSX_GUARD=1 in the environment every device allocation the library makes
 * carries 512 guard bytes in front and behind; an allocation whose guards a kernel has written into aborts the process when it
 * is freed, and this call looks at all live ones now: the number of damaged guard bytes (each reported on stderr), 0 if all
 * are intact, -1 if the guard is off. */
int64_t msx_debug_guard_check(void);
/* ... and the guard checking itself: a kernel writes one byte behind (front = 0) or in front of (front = 1) a small allocation;
 * returns the damaged guard bytes found (1), -1 if the guard is off. */
int64_t msx_debug_guard_selftest(int front);
/* The HIP stream all work of this ctx is enqueued on (a hipStream_t), so the
 * caller can record events on it or make other streams wait for it.  One
 * exception to "all work": msx_filter_profile_enqueue leaves two chains running
 * on internal side streams (see there); every entry point of this library that
 * takes the ctx joins them into this stream before it does anything else, so
 * the exception is only visible to a caller that enqueues its own work on this
 * stream right after msx_filter_profile_enqueue -- call msx_ctx_sync() or
 * msx_filter_finish() first in that case. */
void *msx_ctx_stream(msx_ctx *ctx);
int  msx_ctx_sync(msx_ctx *ctx);

/* ---- batches ------------------------------------------------------------- */

/* Copies a host batch into freshly allocated device memory; *dev receives the
 * device-pointer view.  Synchronous (pageable source, one allocation per array):
 * the simple form, used by the tests.  Free with msx_batch_free. */
int  msx_batch_upload(msx_ctx *ctx, const msx_batch *host, msx_batch *dev);
void msx_batch_free(msx_ctx *ctx, msx_batch *dev);

/* The streaming form the command line uses (counterpart of mSamRead handing
 * record after record to the loops of msam_filter.c:119 / msam_profile.c:222,
 * msam_helper.c:246-268): a *stage* is a set of device buffers that is reused
 * batch after batch -- no allocation, no synchronisation per batch.
 * msx_stage_upload enqueues hipMemcpyAsync copies on the ctx stream; they are
 * true DMA transfers that overlap the host's work on the next batch when the
 * host arrays are page-locked: allocate them with msx_host_alloc, or pin
 * existing memory with msx_host_register.  The host arrays must stay untouched
 * until the stream has passed the copies (msx_filter_finish / msx_ctx_sync).
 * msx_stage_outputs hands out the stage's keep / emit_idx / as_out buffers. */
typedef struct msx_stage msx_stage;
int  msx_stage_create(msx_ctx *ctx, msx_stage **stage);
void msx_stage_destroy(msx_ctx *ctx, msx_stage *stage);
int  msx_stage_upload(msx_ctx *ctx, msx_stage *stage, const msx_batch *host, msx_batch *dev);
int  msx_stage_outputs(msx_ctx *ctx, msx_stage *stage, int64_t n_records, int want_as_out, msx_filter_out *out);
int  msx_host_alloc(msx_ctx *ctx, void **ptr, size_t bytes);     /* hipHostMalloc */
void msx_host_free(msx_ctx *ctx, void *ptr);
int  msx_host_register(msx_ctx *ctx, void *ptr, size_t bytes);   /* hipHostRegister: pins memory the caller owns */
int  msx_host_unregister(msx_ctx *ctx, void *ptr);
/* asynchronous device -> host copy on the ctx stream (host memory should be page-locked) */
int  msx_dev_to_host_async(msx_ctx *ctx, void *host, const void *dev, size_t bytes);
/* A marker on the ctx stream: record it behind the work that reads a host buffer (msx_stage_upload) and wait
 * for it before the buffer is reused -- one marker per batch slot instead of a stream synchronisation per batch
 * (the loops of msam_filter.c:119-186 / msam_profile.c:222-234 hand over record after record; here a slot's
 * page-locked arrays go back to the decoder as soon as their copies have left). */
typedef struct msx_event msx_event;
int  msx_event_create(msx_ctx *ctx, msx_event **ev);
int  msx_event_record(msx_ctx *ctx, msx_event *ev);
int  msx_event_wait(msx_ctx *ctx, msx_event *ev);      /* host waits; returns at once if never recorded */
void msx_event_destroy(msx_ctx *ctx, msx_event *ev);

/* ---- the record walk on the device -------------------------------------------------
 *
 * mSamRead (msam_helper.c:246-268) hands the loops of msam_filter.c:119-186 and
 * msam_profile.c:222-234 one record after another.  With msx_unpack the host does
 * only what must be done there (read, BGZF inflate): the inflated BAM byte stream of a
 * batch -- records with their 4-byte block_size prefixes, cut anywhere -- is uploaded
 * as it is, and the device finds the record boundaries, reads the core fields, scans
 * every aux block for MD / NM / AS (bam_aux_get / bam_aux2i), packs CIGAR and MD,
 * compares QNAMEs for the pools of the chosen loop and ends the batch at its last pool
 * boundary; the open pool and the cut record behind it stay on the device and head the
 * next batch.  msx_unpack_finish hands out the msx_batch view (device pointers owned by
 * the unpacker, valid until the next msx_unpack_enqueue) for the compute entry points;
 * msx_unpack_emit returns filter's output records as one byte string (block_size
 * prefixes included, output order), ready for BGZF framing. */
typedef struct msx_unpack msx_unpack;
typedef struct msx_unpack_params {
	int32_t pool_mode;         /* 0: no pools; 1: msam_filter.c:120-125,170; 2: msam_profile.c:223-232;
	                              3: profile's rule over the records filter can write (no best hit)      */
	int32_t unmapped_visible;  /* mode 3: filter -k -v writes unmapped records (msam_filter.c:132-138)  */
	int32_t want_aux;          /* MD / NM / AS wanted (rflags, nm, as)                                  */
	int32_t want_stats;        /* CIGAR and MD arrays wanted (cigar_off, cigar, md_off, md)             */
	int32_t n_targets;         /* header's reference count (plausibility of guessed record starts)     */
	int32_t last;              /* no more bytes follow: every record belongs to the batch, a cut one is an error */
	int32_t cut_mapped;        /* prefer to end the batch in front of a pool that begins with a mapped record */
	int32_t reserved_;
} msx_unpack_params;
typedef struct msx_unpack_result {
	int64_t n_records, n_groups;   /* of the batch                                                   */
	int64_t bytes_consumed;        /* of (carried bytes + new bytes)                                  */
	int64_t carry_bytes;           /* kept on the device for the next batch                           */
	int64_t bad_guesses;           /* record-start guesses the chase had to repair                    */
} msx_unpack_result;
int  msx_unpack_create(msx_ctx *ctx, msx_unpack **u);
void msx_unpack_destroy(msx_ctx *ctx, msx_unpack *u);
/* bytes that precede the first msx_unpack_enqueue's (records a host-side reader has left over) and the QNAME of
 * the last record that named a pool before them (NULL: none) */
int  msx_unpack_seed(msx_ctx *ctx, msx_unpack *u, const uint8_t *carry, size_t n, const char *prev_name);
/* the other direction: what the last msx_unpack_finish left for the next batch (the open pool, the cut record, the last naming
 * QNAME), to the host -- for msx_unpack_seed of another unpacker: one input (the loop of msam_filter.c:119-186 reads ONE
 * stream) dealt batch by batch to several GPUs keeps inflate and walk on the devices, this hand-over alone is serial.
 * host == NULL or cap < *n: only the size is told. */
int  msx_unpack_carry(msx_ctx *ctx, msx_unpack *u, uint8_t *host, size_t cap, size_t *n, char name[256], int *has_name);
int  msx_unpack_enqueue(msx_ctx *ctx, msx_unpack *u, const uint8_t *host_bytes, size_t n, const msx_unpack_params *prm);
int  msx_unpack_finish(msx_ctx *ctx, msx_unpack *u, msx_unpack_result *res, msx_batch *dev_view);
/* optional: after msx_unpack_finish, send the bytes of the NEXT msx_unpack_enqueue ahead (a copy stream of their own),
 * so that they travel while the current batch is filtered and its output fetched; the next enqueue names the same bytes */
int  msx_unpack_prefetch(msx_ctx *ctx, msx_unpack *u, const uint8_t *host_bytes, size_t n);
int  msx_unpack_emit(msx_ctx *ctx, msx_unpack *u, const int32_t *emit_idx_dev, int64_t n_emit, uint8_t *host_out,
                     size_t host_cap, int64_t *n_bytes);
/* msx_unpack_emit in two steps: gather on the device (*n_bytes: how long the byte string is -- the caller's buffer can be
 * sized now), then fetch; with `done` the bytes travel on a copy stream of their own while the next batch is enqueued:
 * msx_event_wait(done) before host_out is read. */
int  msx_unpack_emit_gather(msx_ctx *ctx, msx_unpack *u, const int32_t *emit_idx_dev, int64_t n_emit, int64_t *n_bytes);
int  msx_unpack_emit_fetch(msx_ctx *ctx, msx_unpack *u, uint8_t *host_out, size_t host_cap, msx_event *done);
/* msx_unpack_emit_gather with the BGZF layer of the writer on the device as well (mSamWrite -> sam_write1 -> bgzf_write:
 * msam_helper.c:270-272; "wbu" / "wb": msam_filter.c:464-470): the gathered byte string is cut into payloads of 0xff00
 * bytes and every payload becomes a finished BGZF block -- header with BSIZE, DEFLATE stream, CRC-32, ISIZE -- so that what
 * msx_unpack_emit_fetch brings down is what goes into the file, byte for byte.  level 0: stored blocks (-bu); level >= 1:
 * DEFLATE with dynamic Huffman codes (-b).  One encoder; what the level dials is its window and table sizes -- the LDS a wave
 * takes, hence the waves a compute unit keeps resident: 7-9 window 8 KB, tables of 2048 + 2048 entries (22 GB/s, 1.08 of zlib
 * -6's size on BAM records); 4-6 window 2.5 KB, 1024 + 2048 (40 GB/s, 1.085; -b asks for 6, htslib's default); 1-3 window
 * 2.5 KB, 1024 + 1024 (44 GB/s, 1.09).
 * *n_bytes: bytes of finished blocks; *n_blocks: how many (payloads all full but the last). */
int  msx_unpack_emit_gather_bgzf(msx_ctx *ctx, msx_unpack *u, const int32_t *emit_idx_dev, int64_t n_emit, int level,
                                 int64_t *n_bytes, int64_t *n_blocks);
/* The same in two steps (level >= 1), so that the encoder runs BESIDE the next batch instead of in front of it: _enqueue gathers
 * the batch's records and hands them to the encoder on a stream of its own, then returns; _complete -- once per _enqueue, in
 * the same order, any time later -- waits for that batch's blocks, tells their size and makes them what the next
 * msx_unpack_emit_fetch brings down.  At most two batches between _enqueue and the end of their fetch. */
int  msx_unpack_emit_bgzf_enqueue(msx_ctx *ctx, msx_unpack *u, const int32_t *emit_idx_dev, int64_t n_emit, int level);
int  msx_unpack_emit_bgzf_complete(msx_ctx *ctx, msx_unpack *u, int64_t *n_bytes, int64_t *n_blocks);
/* record offsets of the last batch, u32[n + 1] relative to its first byte (tests, SAM-text writers) */
int  msx_unpack_offsets(msx_ctx *ctx, msx_unpack *u, uint32_t *host, int64_t n);

/* ---- BGZF inflate on the device ------------------------------------------------------
 *
 * mSamRead (msam_helper.c:246-268) reads through htslib's BGZF layer (sam_read1 -> bgzf_read ->
 * inflate of one <= 64 KB raw DEFLATE stream per block, CRC-32 and ISIZE in the trailer).  Blocks are
 * independent: the host only walks the block headers (BSIZE chain) and hands the compressed payloads
 * over as they are; one wave per block inflates them and a second kernel checks every CRC-32.
 * A block the device does not vouch for (status != 0: a code without a table entry, a distance before
 * the start of the block, lengths that do not add up, CRC mismatch) is the caller's to inflate again
 * on the host, whose decoder produces the diagnostics. */
typedef struct msx_bgzf_block {
	uint64_t in_off;           /* first byte of the block's DEFLATE stream in the compressed buffer      */
	uint64_t out_off;          /* where its bytes go in the output buffer (running sum of out_len)        */
	uint32_t in_len;           /* BSIZE + 1 - 12 - XLEN - 8                                               */
	uint32_t out_len;          /* ISIZE (<= 65536)                                                        */
	uint32_t crc32;            /* of the inflated bytes (trailer)                                         */
	uint32_t reserved_;
} msx_bgzf_block;
/* device pointers throughout; d_status: u32[n_blocks]; waits for the result and reports how many blocks were refused */
int  msx_bgzf_inflate(msx_ctx *ctx, const void *d_comp, size_t comp_len, const msx_bgzf_block *d_blocks,
                      int64_t n_blocks, void *d_out, uint32_t *d_status, int64_t *n_refused);
/* msx_unpack_enqueue with the batch's new bytes still compressed (host pointers: the payloads, back to back or not,
 * and their table; out_off counts from the batch's first NEW byte).  msx_unpack_finish returns MSX_ERR_INFLATE if a
 * block was refused: nothing has been consumed then, and the same batch can be handed to msx_unpack_enqueue inflated. */
int  msx_unpack_enqueue_bgzf(msx_ctx *ctx, msx_unpack *u, const uint8_t *host_comp, size_t comp_len,
                             const msx_bgzf_block *host_blocks, int64_t n_blocks, const msx_unpack_params *prm);
/* optional, at any time: upload and inflate the blocks of a coming msx_unpack_enqueue_bgzf on streams of their own, while
 * the current batch is walked, filtered and fetched.  Up to two batches may be on their way; they must be enqueued in
 * the order they were sent (each enqueue names the same buffer, length and block count as its prefetch).  A batch sent
 * ahead is walked in the buffer it was inflated into, and that buffer is filled again once the NEXT batch has been enqueued:
 * whatever reads a batch's bytes (msx_unpack_emit*) must be enqueued before the batch after it is. */
int  msx_unpack_prefetch_bgzf(msx_ctx *ctx, msx_unpack *u, const uint8_t *host_comp, size_t comp_len,
                              const msx_bgzf_block *host_blocks, int64_t n_blocks);

/* ---- BGZF blocks written on the device ------------------------------------------------
 *
 * The other direction (bgzf_write under sam_write1, msam_helper.c:270-272): n_bytes of device memory cut into payloads of
 * 0xff00 bytes, each framed as one BGZF block in d_out, back to back.  level 0: stored blocks (what htslib writes for
 * "wbu", msam_filter.c:464-470); level >= 1: raw DEFLATE ("wb"), LZ77 + dynamic Huffman codes per block (levels 1-3 / 4-6 /
 * 7-9: three sizes of window and tables, see msx_unpack_emit_gather_bgzf) -- the bytes differ
 * from zlib's, the records do not (the reference's tests compare records: tests/functions.sh:160-163).
 * msx_bgzf_bound: bytes d_out must hold.  Waits for the result.  One call takes at most what keeps msx_bgzf_bound below
 * 2^32 for level >= 1 (block offsets are 32-bit words on the device; MSX_ERR_ARG beyond), 0xfff00000 bytes for level 0.
 * The encoder has ONE set of scratch per context: its launches -- this call on the context's stream, the unpacker's
 * emit calls on the stream they name -- are ordered one behind the other whatever streams they run on. */
int64_t msx_bgzf_bound(int64_t n_bytes, int level);
int  msx_bgzf_deflate(msx_ctx *ctx, const void *d_in, size_t n_bytes, int level, void *d_out, size_t out_cap,
                      int64_t *n_out, int64_t *n_blocks);

/* ---- filter: replaces mFilterFileWrapper/mFilterFile + writers ----------- */

/* msam_filter.c:65-263 on one device batch.  Enqueues on the ctx stream and
 * returns; out->* are valid after msx_filter_finish().  Kernels:
 *   aln_stats_filter  bam_get_summary / bam_cigar2details + _FILTER_* + rescore
 *                     (mBamVector.c:23-133, msam_filter.c:31-63,132-183)
 *   besthit_select    mWriteBestHitBamPool{,ByMate} / Unique (msam_filter.c:192-263)
 *   emit_order        the order of mSamWrite calls (msam_filter.c:235-244) */
int  msx_filter_enqueue(msx_ctx *ctx, const msx_batch *dev,
                        const msx_filter_params *params, const msx_filter_out *out);
/* The pipe `msamtools filter ... | msamtools profile -` on one device batch, in
 * one call: exactly msx_filter_enqueue(ctx, dev, params, out) followed by
 * msx_profile_accumulate(ctx, p, dev, out->keep) -- same outputs, same
 * accumulators -- but with --besthit / --uniqhit the per-pool insert accounting
 * (msam_profile.c:65-243) runs inside the best-hit kernel, on the winners it
 * has just selected (msam_filter.c:192-263), instead of in a second pass over
 * the pools.  Declared after msx_profile below. */

/* Waits for the stream; returns MSX_OK or the data error the reference would
 * have died with (MSX_ERR_NO_MD_NM / MSX_ERR_NO_AS), status filled either way. */
int  msx_filter_finish(msx_ctx *ctx, msx_filter_status *status);

/* Per-record alignment statistics only (the mAlignmentSummary fields filter
 * reads: mBamVector.h:38-47), for inspection and tests.  Any output may be
 * NULL.  status[i] = 1 when the record has neither MD nor NM. */
int  msx_aln_stats(msx_ctx *ctx, const msx_batch *dev, int32_t *length,
                   int32_t *query_length, int32_t *query_clip, int32_t *edit,
                   uint8_t *status);

/* ---- profile: replaces mInitInsertCounts / mEstimateInsertCountOnPool /
 *      mInsertCountToAbundanceMatrix (msam_profile.c:23-425) --------------- */

/* fmap: host array [n_targets] tid -> feature (global->fmap, msam_profile.c:
 * 757-852) or NULL for identity (n_features == n_targets). */
int  msx_profile_create(msx_ctx *ctx, msx_profile **p, int32_t n_features,
                        int32_t share_type, const int32_t *fmap, int32_t n_targets);
void msx_profile_destroy(msx_ctx *ctx, msx_profile *p);
int  msx_profile_reset(msx_ctx *ctx, msx_profile *p);

/* mEstimateInsertCountOnFile + OnPool over one device batch (msam_profile.c:
 * 65-243).  keep == NULL: every record of the batch is a record of the input
 * stream (the `profile` subcommand).  keep != NULL (device, [n_records], from
 * msx_filter): the stream is filter's output for this batch, in its output
 * order -- the fused `filter ... | profile -` pipe; dev->group_off are then
 * filter's pools.  profile re-pools what filter writes by QNAME
 * (msam_profile.c:223-232); with dev->pool_rule == MSX_POOLS_FILTER the device
 * does the same: the filter loop opens a new pool at every record whose QNAME
 * differs from the last MAPPED record's (msam_filter.c:120-125,170), so a pool
 * that begins with an unmapped record holds -- behind it -- more alignments of
 * the read the pool before it held (A mapped, B unmapped, A mapped: two filter
 * pools, one insert); such a pool is counted together with its predecessor.
 * dev->flag is needed for that.  With MSX_POOLS_PROFILE every pool is one
 * insert.  Enqueues and returns. */
int  msx_profile_accumulate(msx_ctx *ctx, msx_profile *p, const msx_batch *dev,
                            const uint8_t *keep);

int  msx_filter_profile_enqueue(msx_ctx *ctx, const msx_batch *dev,
                                const msx_filter_params *params, const msx_filter_out *out,
                                msx_profile *p);
/* (Returns with filter's output order -- out->emit_idx -- and the unique-insert
 * counts still being computed on two side streams; they are joined by the next
 * call on this ctx, whichever it is.  msx_profile_finalize_enqueue makes use of
 * that: it builds the sharing store, which needs neither, before it joins.) */

/* One process driving several contexts (one per GPU, batches of one sample dealt round-robin; the reference is one
 * loop over one file, msam_profile.c:222-234): adds what `src` has counted -- ui, d, {inserts, uniq, multi} and its
 * multi-mapper lists (global->multi_mappers) -- to `dst`, which may live on another device (peer copy).  `src` is
 * left as it is.  Afterwards msx_profile_finalize(dst) is mInsertCountToAbundanceMatrix over all the inserts.
 * Batches must have been cut at insert boundaries (as for ranks).  Synchronises both streams. */
int  msx_profile_merge(msx_ctx *dst_ctx, msx_profile *dst, msx_ctx *src_ctx, msx_profile *src);

/* Device pointers to the accumulators, for a cross-GPU all-reduce(sum) by the
 * caller (RCCL): ui_insert_count u32[n_features], d_insert_count f64
 * [n_features] (NULL unless share_type is EQUAL), counters u32[4] =
 * {insert_count, uniq_mapper_count, multi_mapper_count, purged_insert_count}.
 * (EQUAL: the shares counted so far are complete in d[] when this returns -- they are gathered as exact integers and
 * folded in here; accumulate again and ask again before reading d[] again.) */
int  msx_profile_accumulators(msx_ctx *ctx, msx_profile *p, uint32_t **ui,
                              double **d, uint32_t **counters);

/* Proportional sharing, split so a collective can sit between the halves
 * (msam_profile.c:317-410):
 *   begin  : a(i,0) = U(i) = ui/2 (+ d for EQUAL)              :284-289,:326
 *   local  : increment += a(e)/sum over THIS rank's multi-mappers :341-365
 *            (*inc = device f64[n_features]; all-reduce it across ranks)
 *   apply  : a = U + increment, clamp < 1e-20, DELTA^2           :368-380
 *            *delta receives DELTA^2 (host); identical on every rank
 *   purged : this rank's multi-mappers whose features sum to 0   :394-404
 */
/* Fully asynchronous form (no host round trip per iteration; used by bench.py
 * with RCCL running on the ctx stream): call local/apply_enqueue 19 times
 * unconditionally -- a device-side flag turns the launches after convergence
 * into no-ops, and every rank sees the same flag because it sees the same
 * all-reduced `share`.  purged_enqueue leaves this rank's count in a device
 * word (*purged_dev, u32) for a final all-reduce; msx_profile_fetch() then
 * returns iterations / converged / delta[]. */
int  msx_profile_prop_apply_enqueue(msx_ctx *ctx, msx_profile *p);
/* device pointer of the per-iteration all-reduce vector (f64[n_features]); fixed for the profile's lifetime */
int  msx_profile_share_dev(msx_ctx *ctx, msx_profile *p, double **share);
int  msx_profile_prop_purged_enqueue(msx_ctx *ctx, msx_profile *p, uint32_t **purged_dev);
int  msx_profile_prop_begin(msx_ctx *ctx, msx_profile *p);
int  msx_profile_prop_local(msx_ctx *ctx, msx_profile *p, double **inc);
int  msx_profile_prop_apply(msx_ctx *ctx, msx_profile *p, double *delta);
/* The local half in SLICES of the feature range (what MSX_DIST_SLICES has msx_profile_finalize_dist_enqueue do with RCCL): call
 * with slice = 0 .. n_slices - 1 (n_slices <= 4) in order; after call i, share[*first, *first + *count) -- *inc is the vector --
 * is this rank's complete part for those features and may be all-reduced while the next slice is computed.  The slices
 * are the n_slices equal parts of [0, n_features): the same ranges on every rank, whatever its shard holds. */
int  msx_profile_prop_local_slice(msx_ctx *ctx, msx_profile *p, int slice, int n_slices, double **inc, int32_t *first, int32_t *count);
int  msx_profile_prop_purged(msx_ctx *ctx, msx_profile *p, uint32_t *purged_local);

/* Single-GPU convenience = mInsertCountToAbundanceMatrix for one sample:
 * runs begin / (local, apply) x <=19 / purged on the device with no host
 * round trip per iteration, then copies the abundance row (without the
 * Unknown column) to abundance_host[n_features] and fills stats. */
int  msx_profile_finalize(msx_ctx *ctx, msx_profile *p, double *abundance_host,
                          msx_profile_stats *stats);
/* Same device work, enqueued only (bench timing); results stay on the device.
 * msx_profile_fetch() then syncs and copies them out. */
int  msx_profile_finalize_enqueue(msx_ctx *ctx, msx_profile *p);
int  msx_profile_fetch(msx_ctx *ctx, msx_profile *p, double *abundance_host,
                       msx_profile_stats *stats);
/* Device pointer of the current abundance vector a(i,k), f64[n_features]. */
int  msx_profile_abundance_dev(msx_ctx *ctx, msx_profile *p, double **a);
/* Size of this rank's multi-mapper store (global->multi_mappers, msam_profile.c:
 * 36-39): number of multi-mapped inserts kept for sharing and the total number
 * of (insert, feature) entries.  Synchronises the stream. */
int  msx_profile_multi_size(msx_ctx *ctx, msx_profile *p, int64_t *n_lists, int64_t *n_entries);
/* Size of the derived store the sharing iterations run on (built by prop_begin /
 * finalize: lists renumbered for locality, identical feature sets merged into one
 * weighted list).  Synchronises the stream. */
int  msx_profile_shared_size(msx_ctx *ctx, msx_profile *p, int64_t *n_lists, int64_t *n_entries);

/* ---- several GPUs: one process (rank) per GPU, RCCL over xGMI ------------------
 *
 * The reference is one process over one file.  Here the record stream is cut at
 * pool (QNAME) boundaries into one shard per rank; msam_filter.c:98-263 and the
 * per-pool part of msam_profile.c:65-243 need no communication.  What the
 * reference accumulates over the WHOLE file is exchanged: the per-reference
 * counts and counters once (msam_profile.c:75-186, integer, bit-exact under any
 * order), and inside the loop of msam_profile.c:331-389 the per-iteration
 * increment, as the vector `share` (f64[n_features]).  Every rank then applies
 * the same update and takes the same convergence decision (:383), so no rank
 * waits for its host.  librccl is loaded at run time by msx_dist_init only.
 *
 * A rank creates its context on its own GPU (msx_ctx_create(LOCAL_RANK)), then
 * either msx_dist_init_env (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as set
 * by torchrun, mpirun wrappers, ...; the 128-byte communicator id travels over a
 * TCP rendezvous on MSX_DIST_PORT, default MASTER_PORT + 17) or msx_dist_init with
 * an id the caller has distributed itself (msx_dist_unique_id on rank 0). */
#define MSX_DIST_ID_BYTES 128
int  msx_dist_unique_id(uint8_t id[MSX_DIST_ID_BYTES]);
int  msx_dist_init(msx_ctx *ctx, const uint8_t id[MSX_DIST_ID_BYTES], int rank, int world);
int  msx_dist_init_env(msx_ctx *ctx);
void msx_dist_finalize(msx_ctx *ctx);
int  msx_dist_rank(const msx_ctx *ctx);    /* 0 without a communicator */
int  msx_dist_world(const msx_ctx *ctx);   /* 1 without a communicator */
/* rank 0 sends `bytes` of payload to every other rank (plain TCP; used by msx_dist_init_env) */
int  msx_dist_rendezvous(const char *addr, int port, int rank, int world, void *payload, size_t bytes,
                         int timeout_s);
/* stream-ordered reductions of one host value over all ranks (harness use: barrier, the slowest
 * rank's elapsed time, record totals); identity with one rank */
int  msx_dist_barrier(msx_ctx *ctx);
int  msx_dist_max_f64(msx_ctx *ctx, double *value);
int  msx_dist_sum_i64(msx_ctx *ctx, int64_t *value);
/* all-reduce(sum) of ui / d / {inserts, uniq, multi} of this rank's profile, enqueued on the ctx stream */
int  msx_profile_allreduce_counts(msx_ctx *ctx, msx_profile *p);
/* mInsertCountToAbundanceMatrix over all ranks' shards: msx_profile_allreduce_counts, then
 * msx_profile_finalize_enqueue's device work with `share` all-reduced inside every iteration and the
 * purged count summed at the end.  Identical results on every rank (fetch them with msx_profile_fetch).  Without a
 * communicator it IS msx_profile_finalize_enqueue.  With one, the loop looks at the convergence flag (msam_profile.c:383)
 * every MSX_DIST_POLL iterations (default 8) -- a 4-byte copy and a wait for the stream -- and stops enqueueing
 * collectives once it is set; MSX_DIST_POLL=0: all 19 iterations are enqueued and nothing waits for the host. */
int  msx_profile_finalize_dist_enqueue(msx_ctx *ctx, msx_profile *p);

/* ---- coverage: replaces mUpdateCoverageForAlignment (msam_coverage.c:33-87) */

/* cov: device int32[cov_off[n_targets] + 1], zeroed by the caller once per
 * file; cov_off: device int64[n_targets+1] prefix sum of target_len.
 * accumulate records every M/=/X run as +1/-1 differences (D and N advance;
 * I, S, H, P do not; tid < 0 skipped); after the last batch
 * msx_coverage_finish() converts the buffer in place into per-base depths
 * (what mUpdateCoverageForAlignment builds by adding 1 per base).  The
 * reference has no bounds check (a run past its target's end writes past that
 * target's array); here such a run is cut at the target's ends. */
int  msx_coverage_accumulate(msx_ctx *ctx, const msx_batch *dev,
                             const int64_t *cov_off, int32_t n_targets,
                             int64_t total_len /* = cov_off[n_targets], as the caller summed it */, int32_t *cov,
                             uint8_t *covered /* device u8[n_targets] or NULL: global->covered[tid],
                                                 set for every target that has an alignment (msam_coverage.c:45-49) */);
int  msx_coverage_finish(msx_ctx *ctx, int32_t *cov, int64_t total_len);
/* A whole sample in one batch (the loop of msam_coverage.c:293-301 over one file that fits the device): per-base depths
 * straight into cov[0 .. total_len] -- the caller need not zero it, no msx_coverage_finish follows.  Same result as
 * zero + msx_coverage_accumulate + msx_coverage_finish; the depth array is written once instead of touched three times
 * (the runs, cut at tile boundaries, sorted by tile; every tile's depths finished where its pieces are gathered).  Runs
 * are expected to lie inside their targets (beyond a target's end the reference writes outside its array,
 * msam_coverage.c:66-70; here such a run is cut at the end).  Batches under 65 536 records, and batches whose overflow
 * lists run full (most records with several long runs), take the streamed path inside the call.  Waits for the result. */
int  msx_coverage_depths(msx_ctx *ctx, const msx_batch *dev, const int64_t *cov_off, int32_t n_targets, int64_t total_len,
                         int32_t *cov, uint8_t *covered);
/* A whole sample, batch after batch (the loop of msam_coverage.c:293-301 as the command line runs it): msx_coverage_collect
 * in the place of msx_coverage_accumulate, msx_coverage_collect_finish in the place of msx_coverage_finish.  The batches'
 * pieces stay on the device (5 bytes per record) and the finish sorts and sums them once, writing cov[0 .. total_len] once:
 * msx_coverage_depths without the batch having to be the sample.  cov need NOT be zeroed and holds nothing until the finish.
 * What the one-word-per-piece form does not take goes the streamed way inside the same calls and the results are added:
 * samples of more than 255 x 2^20 cells (every batch), a batch whose overflow lists ran full, batches beyond 2^31 items.
 * One sample at a time per context; every call names the same cov / total_len.  *n_batches_streamed (or NULL): how many
 * batches took the streamed way.  msx_coverage_collect waits for the batch's pieces (the batch's arrays may be given back
 * when it returns); the finish enqueues and returns -- the depths are on the context's stream. */
int  msx_coverage_collect(msx_ctx *ctx, const msx_batch *dev, const int64_t *cov_off, int32_t n_targets, int64_t total_len,
                          int32_t *cov, uint8_t *covered);
int  msx_coverage_collect_finish(msx_ctx *ctx, int32_t *cov, int64_t total_len, int64_t *n_batches_streamed);
/* after msx_coverage_finish: per target, the number of positions with a depth other than 0 and the sum of the depths --
 * what mWriteCoverageSummaryToStream (msam_coverage.c:188-219) divides by the target's length; host arrays of n_targets */
int  msx_coverage_summary(msx_ctx *ctx, const int32_t *cov, const int64_t *cov_off, int32_t n_targets,
                          int64_t *touched_host, int64_t *sum_host);

/* ---- synthetic workloads (bench.py / tests) -------- */

typedef struct msx_synth_params {
	uint64_t seed;
	int64_t  n_groups;       /* QNAME groups (reads/inserts)                    */
	int32_t  n_refs;         /* references (= features)                         */
	int32_t  mean_extra_hits;/* hits per read = 1 + Poisson(mean_extra_hits), <= 4 supported */
	int64_t  first_group;    /* generate groups [first_group, first_group+n_groups) of the
	                            infinite deterministic stream (sharding / prefixes) */
} msx_synth_params;

/* Sizes needed for the arrays of a synthetic batch (computed on the device). */
typedef struct msx_synth_sizes {
	int64_t n_records, n_cigar, n_md;
} msx_synth_sizes;

/* Generates a batch directly in device memory owned by the ctx (free with
 * msx_batch_free).  Deterministic in (seed, group index): the same groups come
 * out of msx_synth_host(). */
int  msx_synth_device(msx_ctx *ctx, const msx_synth_params *sp, msx_batch *dev,
                      msx_synth_sizes *sizes);
/* Host twin (plain C loops, no GPU needed): allocates with malloc; free with
 * msx_synth_host_free.  Used to feed the CPU oracle the same data. */
int  msx_synth_host(const msx_synth_params *sp, msx_batch *host, msx_synth_sizes *sizes);
void msx_synth_host_free(msx_batch *host);

/* ---- raw device memory helpers for C callers (thin hipMalloc wrappers) ---- */
int  msx_dev_alloc(msx_ctx *ctx, void **ptr, size_t bytes);
void msx_dev_free(msx_ctx *ctx, void *ptr);
int  msx_dev_zero(msx_ctx *ctx, void *ptr, size_t bytes);       /* async on ctx stream */
int  msx_dev_to_host(msx_ctx *ctx, void *host, const void *dev, size_t bytes); /* sync */
int  msx_host_to_dev(msx_ctx *ctx, void *dev, const void *host, size_t bytes); /* sync */

/* ---- timing of the dominant kernel (bench.py roofline) -------------------- */

/* Per-kernel device time measured with HIP events on the ctx stream while
 * enabled (one event pair around every launch).  names: "k_aln_stats_flat",
 * "k_besthit_select", "k_emit_order", "k_insert_count", "k_multi_compact",
 * "k_general_recip", "k_share_reduce", "k_partial_reduce", "k_prop_apply", "k_list_order",
 * "k_rs_hist", "k_rs_scatter",
 * "k_coverage_pileup", "scan", "synth".  Returns total ms and the number of
 * timed launches since the last reset. */
int  msx_timing_enable(msx_ctx *ctx, int on);
int  msx_timing_reset(msx_ctx *ctx);
int  msx_timing_get(msx_ctx *ctx, const char *name, double *ms_total, int64_t *launches);
/* Algorithmic bytes of the timed launches of `name`, for the kernels whose element counts exist on the
 * device only ("scan", "k_rs_hist", "k_rs_scatter": every launch priced by its own length); 0 for the
 * others, which the caller prices from the batch sizes (bench.py). */
int  msx_timing_get_bytes(msx_ctx *ctx, const char *name, int64_t *bytes_total);

#ifdef __cplusplus
}
#endif
#endif /* MSAMTOOLS_AMD_H */
