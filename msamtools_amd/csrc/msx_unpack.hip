// msx_unpack.hip -- the record walk of the reader loops on the device.
//
// mSamRead (msam_helper.c:246-268) hands the loops of msam_filter.c:119-186 / msam_profile.c:222-234 one bam1_t
// after another; the host pipeline of this repository (csrc/host) turned that into "inflate, find the record
// boundaries, scan every record's aux block, fill the SoA arrays, compare QNAMEs" on the host cores -- half of the
// decode stage's time once inflate had been made fast.  Here the inflated BAM bytes of a batch are uploaded as they
// are and everything after inflate happens on the device:
//   k_chase_walk      record boundaries.  The block_size chain is serial; every lane takes a 4 KB segment, guesses a
//                     record start in it (a header that looks like one, followed by two more that do) and walks its
//                     segment; k_chase_check counts the segments that do not begin where the segment before them
//                     ended, k_chase_fix repairs those -- serially, rarely: from a true start the chain is the true
//                     chain, so guesses only decide how parallel the walk was.
//   k_rec_fields      one lane per record: core fields, one pass over the aux block for MD / NM / AS
//                     (bam_aux_get: first occurrence; bam_aux2i: c C s S i I), CIGAR and MD lengths
//   k_rec_payload     CIGAR words and MD bytes packed back to back (msx_batch.cigar / md)
//   k_name_*          the pool rule: a record's QNAME against the QNAME of the nearest earlier record that names the
//                     read being collected (msam_filter.c:120-125,170 / msam_profile.c:223-232) -- a prefix maximum
//                     of record indices, then a string compare per record
//   k_pool_*          group_off from the boundary bits, the batch's last pool boundary (the open pool and the cut
//                     record behind it stay on the device as the next batch's first bytes), the carried name
//   k_emit_*          filter's output records, block_size prefixes included, concatenated in output order
// Integer / byte work, HBM-bound; no MFMA.
#include "msx_internal.h"

#include <cstdlib>
#include <cstring>

#define UP_SEG 4096u               // bytes per chase segment (one lane each: ~55 records of 71 bytes)
#define UP_NONE 0xffffffffu
#define UP_TILE 2048               // records per workgroup in the name kernels (8 per thread)

// device-side state of one batch
struct up_state {
	uint32_t n_total;              // complete records found
	uint32_t tail_off;             // first byte not covered by them
	uint32_t bad_segments;         // guesses the join had to repair
	uint32_t status;               // MSX_UP_* below
	uint32_t cut_any, cut_mapped;  // last pool boundary (record index) / last one whose record is mapped, in the batch's second half
	uint32_t n_batch, n_groups, cut_off;
	uint32_t inflate_bad;          // msx_unpack_enqueue_bgzf: blocks the device inflater refused
	uint32_t inflate_ticket;       // ... and the counter its workgroups draw their blocks from (msx_bgzf_inflate_launch: d_n_bad[1])
	uint32_t inflate_back;         // blocks the lane-parallel kernel handed back to the serial one (d_n_bad[2]) ...
	uint32_t inflate_ticket2;      // ... and that launch's ticket (d_n_bad[3])
	uint32_t has_prev;             // prev_name holds the QNAME of the last naming record of earlier batches
	uint32_t emit_bytes;           // what msx_unpack_emit_fetch brings down (the record stream, or the BGZF blocks made of it)
	uint32_t emit_raw;             // bytes of the record stream
};
#define MSX_UP_CORRUPT 1u          // block_size < 32 on the true chain, or a record whose fields do not fit its length

struct msx_unpack {
	msx_buf raw[2];                // ping-pong: [carry][new bytes] of the current batch, the next batch's carry lands in the other
	int cur = 0;
	size_t carry_len = 0, n_bytes = 0;
	msx_buf seg_first, seg_end, seg_cnt, seg_base;
	msx_buf comp, blk, blk_status;  // msx_unpack_enqueue_bgzf: the compressed payloads, their table, the inflater's verdicts
	bool bgzf = false;              // the current batch came in compressed
	// msx_unpack_prefetch_bgzf: the NEXT batches (up to two) uploaded and inflated on streams of their own while the current
	// one is walked, filtered and fetched -- into staging buffers (the carry a batch will follow is not known yet);
	// msx_unpack_enqueue_bgzf then copies the oldest behind the carry.  Two sets: the upload of batch k+2 travels while batch
	// k+1 is being inflated, and the inflater never waits for the host.
	struct pre_set {
		msx_buf comp, blk, status, out, cnt;
		hipEvent_t h2d_done = nullptr, inf_done = nullptr, freed = nullptr;
		const uint8_t *key = nullptr;
		size_t comp_len = 0, total = 0;
		int64_t nblk = 0;
		bool used = false;              // `freed` has been recorded at least once
	} pre[3];
	int ahead_head = 0, ahead_n = 0;   // the oldest pending set, how many are pending
	// a stream per batch on its way (two at most), made when first needed: the blocks go up and are inflated on it.  (Until the
	// end of round 6 the first call made four streams -- one per set and one for the uploads: 30 ms on the device thread between
	// batch 0 and batch 1, 7.5 ms a stream: its hardware queue and the queue's 173 MB of host memory -- of which the command
	// line's one batch ahead ever used two.)
	hipStream_t inf_stream[2] = {nullptr, nullptr};
	// Round 6: a batch sent ahead is WALKED WHERE IT WAS INFLATED -- its set's `out`, which leaves UP_HEAD bytes free in front of
	// the blocks' bytes: the carry (the open pool and the cut record of the batch before: kilobytes) is copied in front of them,
	// instead of the batch (a quarter of a gigabyte) behind the carry.  bytes[parity]: where the batch of that parity begins --
	// raw[parity].p, or inside a set; in_place[parity]: that set (until the batch after it is enqueued: its emits are through by then).
	// Three sets: the batch at hand and two sent ahead.
	uint8_t *bytes[2] = {nullptr, nullptr};
	int in_place[2] = {-1, -1};
	msx_buf rec_off, flag, rflags, tid, pos, nm, as, cig_cnt, cig_src, cigar_off, md_len, md_off, md_src, bd, pidx, gflag, gpos,
	    group_off, tile_last, cigar, md, out_len, out_off, out, framed;
	char *prev_name = nullptr;     // device, 256 bytes
	up_state *d_state = nullptr, *h_state = nullptr;
	msx_unpack_params prm = {};
	int64_t n_total = 0, n_batch = 0, n_groups = 0;
	bool enqueued = false;
	// msx_unpack_prefetch: the next batch's bytes on their way up (a stream of its own) while the current batch is filtered
	hipStream_t copy_stream = nullptr;
	hipEvent_t copy_done = nullptr;
	hipEvent_t out_copied = nullptr;   // msx_unpack_emit_fetch: the gather buffer has been read
	bool out_busy = false;
	size_t gathered = 0;               // msx_unpack_emit_gather: bytes in `out` (msx_unpack_emit_gather_bgzf: in `framed`)
	const void *fetch_src = nullptr;   // what msx_unpack_emit_fetch brings down
	// msx_unpack_emit_bgzf_enqueue / _complete: the encoder on a stream of its own, two batches in flight.  Batch k's
	// gathered records (eo[k & 1]) are deflated into ef[k & 1] while batch k + 1 is walked and filtered on the context's
	// stream; sizes come back through a pinned pair per parity.
	msx_buf eo[2], ef[2];
	hipStream_t df_stream = nullptr;
	hipEvent_t ev_gathered = nullptr, ev_deflated[2] = {nullptr, nullptr}, ev_fetched[2] = {nullptr, nullptr};
	bool deflated_used[2] = {false, false}, fetched_used[2] = {false, false};
	uint32_t *d_emit = nullptr, *h_emit = nullptr;      // [4]: {bytes of blocks, bytes of records} per parity
	uint64_t eseq_in = 0, eseq_out = 0;                 // batches enqueued / completed
	int64_t e_nemit[2] = {0, 0};
	int fetch_par = -1;                                 // the parity msx_unpack_emit_fetch is bringing down (-1: the plain path)
	const uint8_t *pre_host = nullptr;
	size_t pre_n = 0;
};

// (global memory takes unaligned dword / qword loads on this target: one instruction instead of four byte loads)
typedef uint32_t __attribute__((aligned(1))) u32_u;
typedef uint16_t __attribute__((aligned(1))) u16_u;
typedef unsigned long long __attribute__((aligned(1))) u64_u;
__device__ __forceinline__ uint32_t ld32(const uint8_t *p) { return *reinterpret_cast<const u32_u *>(p); }
__device__ __forceinline__ uint32_t ld16(const uint8_t *p) { return *reinterpret_cast<const u16_u *>(p); }
__device__ __forceinline__ unsigned long long ld64(const uint8_t *p) { return *reinterpret_cast<const u64_u *>(p); }

// ---------------------------------------------------------------------------
// record boundaries
// ---------------------------------------------------------------------------
// does a record plausibly start at `off`?  (only guesses: a wrong one is repaired by k_chase_fix)
__device__ __forceinline__ bool up_plausible(const uint8_t *u, uint32_t off, uint32_t n, int32_t nt) {
	if ((uint64_t)off + 36u > n) return false;
	const uint32_t bs = ld32(u + off);
	if (bs < 32u || bs > (64u << 20)) return false;
	const uint8_t *r = u + off + 4;
	const int32_t tid = (int32_t)ld32(r), pos = (int32_t)ld32(r + 4), mtid = (int32_t)ld32(r + 20), mpos = (int32_t)ld32(r + 24);
	const int32_t ls = (int32_t)ld32(r + 16);
	if (tid < -1 || tid >= nt || mtid < -1 || mtid >= nt || pos < -1 || mpos < -1) return false;
	const uint32_t lq = r[8], nc = ld16(r + 12);
	if (lq < 1u || ls < 0) return false;
	if (32ull + lq + 4ull * nc + ((uint64_t)ls + 1) / 2 + (uint64_t)ls > (uint64_t)bs) return false;
	if ((uint64_t)off + 4u + 32u + lq > n) return true;
	if (r[32 + lq - 1] != 0) return false;
	for (uint32_t k = 0; k + 1 < lq; k++)
		if (r[32 + k] < 33 || r[32 + k] > 126) return false;
	return true;
}

// the records that START in [off, hi): their number, and where the chain stands afterwards (>= hi, or the offset of a
// record cut by the buffer's end, or of a block_size < 32 -- then *invalid)
template <bool WRITE>
__device__ __forceinline__ uint32_t up_walk(const uint8_t *u, uint32_t n, uint32_t off, uint32_t hi, uint32_t *end, bool *invalid,
                                            uint32_t *out) {
	uint32_t cnt = 0;
	*invalid = false;
	while (off < hi && (uint64_t)off + 4u <= n) {
		const uint32_t bs = ld32(u + off);
		if (bs < 32u) { *invalid = true; break; }
		if ((uint64_t)off + 4u + bs > n) break;               // cut by the buffer's end
		if (WRITE) out[cnt] = off;
		cnt++;
		off += 4u + bs;
	}
	*end = off;
	return cnt;
}

__global__ __launch_bounds__(MSX_BLOCK) void k_chase_walk(const uint8_t *__restrict__ u, uint32_t n, uint32_t nseg, int32_t nt,
                                                          uint32_t *__restrict__ seg_first, uint32_t *__restrict__ seg_end,
                                                          uint32_t *__restrict__ seg_cnt) {
	const uint32_t k = blockIdx.x * MSX_BLOCK + threadIdx.x;
	if (k >= nseg) return;
	const uint32_t lo = k * UP_SEG, hi = (lo + UP_SEG < n) ? lo + UP_SEG : n;
	uint32_t first = lo;
	if (k > 0) {
		first = UP_NONE;
		for (uint32_t off = lo; off < hi; off++) {
			if (!up_plausible(u, off, n, nt)) continue;
			const uint32_t o2 = off + 4u + ld32(u + off);
			if ((uint64_t)o2 + 36u <= n) {
				if (!up_plausible(u, o2, n, nt)) continue;
				const uint32_t o3 = o2 + 4u + ld32(u + o2);
				if ((uint64_t)o3 + 36u <= n && !up_plausible(u, o3, n, nt)) continue;
			}
			first = off;
			break;
		}
	}
	uint32_t end = first, cnt = 0;
	bool inv = false;
	if (first != UP_NONE) cnt = up_walk<false>(u, n, first, hi, &end, &inv, nullptr);
	seg_first[k] = first;
	seg_end[k] = end;
	seg_cnt[k] = (first != UP_NONE && !inv) ? cnt : UP_NONE;      // (a walk that ran into a block_size < 32 started from a wrong guess)
}

// Segment k is right when it starts where segment k - 1 ended.  k_chase_check counts the segments that are not
// (wrong guesses, segments a long record jumps over); k_chase_fix -- one lane, and nothing to do in the usual case --
// walks those from where the chain really stands; the counts are then scanned into positions.
__global__ __launch_bounds__(MSX_BLOCK) void k_chase_check(uint32_t n, uint32_t nseg, const uint32_t *__restrict__ seg_first,
                                                           const uint32_t *__restrict__ seg_end, uint32_t *__restrict__ seg_cnt,
                                                           up_state *st) {
	const uint32_t k = blockIdx.x * MSX_BLOCK + threadIdx.x;
	bool bad = false;
	if (k < nseg) {
		if (k == 0) bad = seg_cnt[0] == UP_NONE;
		else {
			const uint32_t pe = seg_end[k - 1], lo = k * UP_SEG, hi = (lo + UP_SEG < n) ? lo + UP_SEG : n;
			bad = !(pe < hi && seg_first[k] == pe && seg_cnt[k] != UP_NONE);
		}
	}
	const unsigned long long m = __ballot(bad);
	if (m && (threadIdx.x & 63) == 0) atomicAdd(&st->bad_segments, (uint32_t)__popcll(m));
}

__global__ void k_chase_fix(const uint8_t *__restrict__ u, uint32_t n, uint32_t nseg, uint32_t *__restrict__ seg_first,
                            uint32_t *__restrict__ seg_end, uint32_t *__restrict__ seg_cnt, up_state *st) {
	if (st->bad_segments == 0) return;
	uint32_t pos = 0, status = 0, fixed = 0;
	for (uint32_t k = 0; k < nseg; k++) {
		const uint32_t lo = k * UP_SEG, hi = (lo + UP_SEG < n) ? lo + UP_SEG : n;
		if (k > 0 && seg_first[k] == pos && seg_cnt[k] != UP_NONE && pos < hi) { pos = seg_end[k]; continue; }
		if (k == 0 && seg_cnt[0] != UP_NONE) { pos = seg_end[0]; continue; }
		uint32_t end = pos, cnt = 0;
		bool inv = false;
		if (pos < hi) cnt = up_walk<false>(u, n, pos, hi, &end, &inv, nullptr);
		if (inv) status = MSX_UP_CORRUPT;                 // on the true chain: the data is corrupt
		seg_first[k] = pos;
		seg_cnt[k] = cnt;
		seg_end[k] = end;
		pos = end;
		fixed++;
		if (inv) {                                        // nothing behind it can be trusted
			for (uint32_t q = k + 1; q < nseg; q++) { seg_first[q] = end; seg_cnt[q] = 0; seg_end[q] = end; }
			break;
		}
	}
	st->bad_segments = fixed;
	if (status) st->status = status;
}

__global__ void k_chase_total(uint32_t nseg, const uint32_t *__restrict__ seg_base, const uint32_t *__restrict__ seg_end, up_state *st) {
	st->n_total = seg_base[nseg];
	st->tail_off = nseg ? seg_end[nseg - 1] : 0u;
}

__global__ __launch_bounds__(MSX_BLOCK) void k_chase_write(const uint8_t *__restrict__ u, uint32_t n, uint32_t nseg,
                                                           const uint32_t *__restrict__ seg_first, const uint32_t *__restrict__ seg_cnt,
                                                           const uint32_t *__restrict__ seg_base, uint32_t *__restrict__ rec_off,
                                                           const up_state *st) {
	const uint32_t k = blockIdx.x * MSX_BLOCK + threadIdx.x;
	if (k == 0) rec_off[st->n_total] = st->tail_off;
	if (k >= nseg || seg_cnt[k] == 0u) return;
	const uint32_t lo = k * UP_SEG, hi = (lo + UP_SEG < n) ? lo + UP_SEG : n;
	uint32_t end;
	bool inv;
	(void)up_walk<true>(u, n, seg_first[k], hi, &end, &inv, rec_off + seg_base[k]);
}

// ---------------------------------------------------------------------------
// fields
// ---------------------------------------------------------------------------
// *bad: the field does not fit the record (a fixed-size value cut by its end, a string without its NUL, an array longer
// than what is left, an unknown type) -- msh_aux_size and htslib refuse such a record
__device__ __forceinline__ uint32_t aux_size(const uint8_t *t, const uint8_t *end, bool *bad) {      // type byte + payload, as msh_aux_size
	uint32_t fs = 0;
	switch (*t) {
	case 'A': case 'c': case 'C': fs = 2; break;
	case 's': case 'S': fs = 3; break;
	case 'i': case 'I': case 'f': fs = 5; break;
	case 'Z': case 'H': {
		const uint8_t *p = t + 1;
		while (p < end && *p) p++;
		if (p >= end) *bad = true;
		return (uint32_t)(p - t) + 1u;
	}
	case 'B': {
		if (t + 6 > end) { *bad = true; return (uint32_t)(end - t); }
		const uint32_t cnt = ld32(t + 2);
		uint32_t es = 0;
		switch (t[1]) { case 'c': case 'C': es = 1; break; case 's': case 'S': es = 2; break; case 'i': case 'I': case 'f': es = 4; break; default: break; }
		const uint64_t sz = 6ull + (uint64_t)cnt * es;
		if (es == 0 || sz > (uint64_t)(end - t)) { *bad = true; return (uint32_t)(end - t); }
		return (uint32_t)sz;
	}
	default: *bad = true; return (uint32_t)(end - t);      // unknown type: nothing behind it can be parsed
	}
	if (t + fs > end) { *bad = true; return (uint32_t)(end - t); }
	return fs;
}

__device__ __forceinline__ int32_t aux2i(const uint8_t *t) {       // bam_aux2i, truncated to int32 as the reference does
	switch (*t) {
	case 'c': return (int8_t)t[1];
	case 'C': return t[1];
	case 's': return (int16_t)ld16(t + 1);
	case 'S': return (int32_t)ld16(t + 1);
	case 'i': case 'I': return (int32_t)ld32(t + 1);
	default: return 0;
	}
}

__global__ __launch_bounds__(MSX_BLOCK) void k_rec_fields(const uint8_t *__restrict__ u, uint32_t n_rec,
                                                          const uint32_t *__restrict__ rec_off, int want_aux, int want_stats,
                                                          uint16_t *__restrict__ flag, uint8_t *__restrict__ rflags,
                                                          int32_t *__restrict__ tid, int32_t *__restrict__ pos,
                                                          int32_t *__restrict__ nm, int32_t *__restrict__ as,
                                                          uint32_t *__restrict__ cig_cnt, uint32_t *__restrict__ cig_src,
                                                          uint32_t *__restrict__ md_len, uint32_t *__restrict__ md_src, up_state *st) {
	const uint32_t i = blockIdx.x * MSX_BLOCK + threadIdx.x;
	if (i >= n_rec) return;
	const uint32_t o = rec_off[i], len = rec_off[i + 1] - o - 4u;
	const uint8_t *r = u + o + 4;
	const uint32_t lq = r[8], nc = ld16(r + 12), fl = ld16(r + 14);
	const int32_t ls = (int32_t)ld32(r + 16);
	const uint64_t aux0 = 32ull + lq + 4ull * nc + (ls >= 0 ? ((uint64_t)ls + 1) / 2 + (uint64_t)ls : 0ull);
	// msh_rec_check: the fields the record announces fit its length, the name is terminated
	if (ls < 0 || lq < 1u || aux0 > len || r[32 + lq - 1] != 0) {
		st->status = MSX_UP_CORRUPT;
		flag[i] = 4; rflags[i] = 0; tid[i] = -1; pos[i] = 0; nm[i] = 0; as[i] = 0; cig_cnt[i] = 0; cig_src[i] = 0; md_len[i] = 0; md_src[i] = 0;
		return;
	}
	flag[i] = (uint16_t)fl;
	tid[i] = (int32_t)ld32(r);
	pos[i] = (int32_t)ld32(r + 4);
	uint32_t rf = 0, ml = 0, msrc = 0;
	int32_t vnm = 0, vas = 0;
	// A CIGAR of more than 65535 operations is stored as the placeholder <l_seq>S<reference length>N with the real one in a
	// CG:B:I tag (SAMv1 4.2.2); htslib's sam_read1 (under mSamRead, msam_helper.c:246-268: bam_tag2cigar) swaps it in, so
	// the reference's statistics (mBamVector.c:23-133) and pile-up (msam_coverage.c:33-87) are computed from the real one.
	// The same rule here: mapped, first operation S of l_seq bases, the first CG tag of type B with I / i elements and at
	// least n_cigar of them (fewer than 2^29).  The record's bytes pass through as they are.
	uint32_t csrc = (uint32_t)(r + 32 + lq - u), cn = nc;
	const bool placeholder = nc >= 1u && (int32_t)ld32(r) >= 0 && (int32_t)ld32(r + 4) >= 0 &&
	                         (ld32(r + 32 + lq) & 15u) == (uint32_t)MSX_OP_SOFT_CLIP && (ld32(r + 32 + lq) >> 4) == (uint32_t)ls;
	if (want_aux || (placeholder && want_stats)) {
		const uint8_t *p = r + aux0, *end = r + len;
		bool md = false, hnm = false, has = false, hcg = false;
		bool bad = false;
		while (p + 3 <= end && !bad) {
			const uint32_t sz = aux_size(p + 2, end, &bad);
			if (bad) break;                              // (nothing of a field that does not fit is used)
			if (p[0] == 'M' && p[1] == 'D' && !md) {
				md = true;
				if (p[2] == 'Z') { ml = sz - 2u; msrc = (uint32_t)(p + 3 - u); }      // type byte and NUL are not part of the string
			} else if (p[0] == 'N' && p[1] == 'M' && !hnm) { hnm = true; vnm = aux2i(p + 2); }
			else if (p[0] == 'A' && p[1] == 'S' && !has) { has = true; vas = aux2i(p + 2); }
			else if (p[0] == 'C' && p[1] == 'G' && !hcg) {
				hcg = true;
				if (placeholder && p[2] == 'B' && (p[3] == 'I' || p[3] == 'i')) {
					const uint32_t k = ld32(p + 4);
					if (k >= nc && k < (1u << 29)) { cn = k; csrc = (uint32_t)(p + 8 - u); }
				}
			}
			p += 2 + sz;
		}
		if (bad) st->status = MSX_UP_CORRUPT;
		if (want_aux) rf = (md ? MSX_HAS_MD : 0u) | (hnm ? MSX_HAS_NM : 0u) | (has ? MSX_HAS_AS : 0u);
		else { ml = 0; msrc = 0; vnm = 0; vas = 0; }
	}
	rflags[i] = (uint8_t)rf;
	nm[i] = vnm;
	as[i] = vas;
	cig_cnt[i] = want_stats ? cn : 0u;
	cig_src[i] = csrc;
	md_len[i] = want_stats ? ml : 0u;
	md_src[i] = msrc;
}

__global__ __launch_bounds__(MSX_BLOCK) void k_rec_payload(const uint8_t *__restrict__ u, uint32_t n_rec,
                                                           const uint32_t *__restrict__ rec_off,
                                                           const uint32_t *__restrict__ cigar_off, const uint32_t *__restrict__ md_off,
                                                           const uint32_t *__restrict__ md_src, const uint32_t *__restrict__ cig_src,
                                                           uint32_t *__restrict__ cigar, uint8_t *__restrict__ md) {
	const uint32_t i = blockIdx.x * MSX_BLOCK + threadIdx.x;
	if (i >= n_rec) return;
	const uint32_t c0 = cigar_off[i], nc = cigar_off[i + 1] - c0;
	if (nc) {
		const uint8_t *cg = u + cig_src[i];            // the record's own CIGAR, or its CG:B:I tag's array (k_rec_fields)
		for (uint32_t k = 0; k < nc; k++) cigar[c0 + k] = ld32(cg + 4 * k);
	}
	const uint32_t m0 = md_off[i], ml = md_off[i + 1] - m0;
	if (ml) {
		const uint8_t *src = u + md_src[i];
		for (uint32_t k = 0; k < ml; k++) md[m0 + k] = src[k];
	}
}

// ---------------------------------------------------------------------------
// pools
// ---------------------------------------------------------------------------
// mode 1  msam_filter.c:120-125,170: every record is compared, a MAPPED one names the read
// mode 2  msam_profile.c:223-232: records with tid == -1 are skipped entirely
// mode 3  profile's rule over the records filter can write when pools do not shape its output
__device__ __forceinline__ bool names_pool(int mode, int uv, uint32_t fl, int32_t t) {
	if (mode == 1) return !(fl & 4u);
	if (mode == 2) return t != -1;
	return t != -1 && (uv || !(fl & 4u));
}
__device__ __forceinline__ bool rule_sees(int mode, int uv, uint32_t fl, int32_t t) { return mode == 1 ? true : names_pool(mode, uv, fl, t); }

__global__ __launch_bounds__(MSX_BLOCK) void k_name_tile_last(uint32_t n_rec, int mode, int uv, const uint16_t *__restrict__ flag,
                                                              const int32_t *__restrict__ tid, int32_t *__restrict__ tile_last) {
	__shared__ int32_t s_m;
	if (threadIdx.x == 0) s_m = -1;
	__syncthreads();
	const uint32_t base = blockIdx.x * UP_TILE;
	int32_t m = -1;
	for (uint32_t q = threadIdx.x; q < UP_TILE; q += MSX_BLOCK) {
		const uint32_t i = base + q;
		if (i < n_rec && names_pool(mode, uv, flag[i], tid[i])) m = (int32_t)i > m ? (int32_t)i : m;
	}
	for (int d = 32; d > 0; d >>= 1) { const int32_t o = __shfl_down(m, d, 64); m = o > m ? o : m; }
	if ((threadIdx.x & 63) == 0 && m >= 0) atomicMax(&s_m, m);
	__syncthreads();
	if (threadIdx.x == 0) tile_last[blockIdx.x] = s_m;
}

// exclusive prefix maximum over the tiles (one workgroup; a batch has a few thousand tiles)
__global__ __launch_bounds__(MSX_BLOCK) void k_name_tile_scan(uint32_t n_tiles, int32_t *__restrict__ tile_last) {
	__shared__ int32_t s_w[MSX_BLOCK / 64], s_run;
	if (threadIdx.x == 0) s_run = -1;
	__syncthreads();
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	for (uint32_t base = 0; base < n_tiles; base += MSX_BLOCK) {
		const uint32_t k = base + threadIdx.x;
		const int32_t v = k < n_tiles ? tile_last[k] : -1;
		int32_t inc = v;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const int32_t o = __shfl_up(inc, d, 64);
			if (lane >= d) inc = o > inc ? o : inc;
		}
		if (lane == 63) s_w[w] = inc;
		__syncthreads();
		int32_t pre = s_run;
		for (int q = 0; q < w; q++) pre = s_w[q] > pre ? s_w[q] : pre;
		int32_t ex = __shfl_up(inc, 1, 64);
		if (lane == 0) ex = -1;
		ex = ex > pre ? ex : pre;
		int32_t tot = s_run;
		for (int q = 0; q < MSX_BLOCK / 64; q++) tot = s_w[q] > tot ? s_w[q] : tot;
		__syncthreads();
		if (k < n_tiles) tile_last[k] = ex;
		if (threadIdx.x == 0) s_run = tot;
		__syncthreads();
	}
}

// names differ?  (la, lb include the terminating NUL; eight bytes at a time -- the reads may run past a name's end
// into the record's own CIGAR / the next record: the buffer is padded by 64 bytes)
__device__ __forceinline__ bool name_differs8(const uint8_t *a, uint32_t la, const uint8_t *b, uint32_t lb) {
	if (la != lb) return true;
	uint32_t k = 0;
	for (; k + 8u <= la; k += 8u)
		if (ld64(a + k) != ld64(b + k)) return true;
	if (k < la) {
		const unsigned long long m = ~0ull >> (8u * (8u - (la - k)));
		if ((ld64(a + k) ^ ld64(b + k)) & m) return true;
	}
	return false;
}

// bd[i]: record i opens a pool; pidx[i]: the nearest earlier naming record (-1: none in this batch), for i = 0 .. n_rec.
// Also the batch's last boundary, and the last one in its second half whose record is mapped (cut_mapped).
// One record per lane, a tile in UP_TILE / MSX_BLOCK rows; the running maximum goes from row to row.
__global__ __launch_bounds__(MSX_BLOCK) void k_name_bounds(const uint8_t *__restrict__ u, uint32_t n_rec, int mode, int uv,
                                                           const uint32_t *__restrict__ rec_off, const uint16_t *__restrict__ flag,
                                                           const int32_t *__restrict__ tid, const int32_t *__restrict__ tile_carry,
                                                           const char *__restrict__ prev_name, uint8_t *__restrict__ bd,
                                                           int32_t *__restrict__ pidx, uint32_t *__restrict__ gflag, up_state *st) {
	__shared__ int32_t s_w[MSX_BLOCK / 64];
	__shared__ uint32_t s_cut[2];
	__shared__ uint8_t s_prev[256];
	__shared__ uint32_t s_plen;
	if (threadIdx.x < 2) s_cut[threadIdx.x] = 0;
	const bool has_prev = st->has_prev != 0;
	s_prev[threadIdx.x] = has_prev ? (uint8_t)prev_name[threadIdx.x] : 0;
	__syncthreads();
	if (threadIdx.x == 0) { uint32_t lp = 0; while (lp < 255u && s_prev[lp]) lp++; s_plen = lp + 1u; }
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	int32_t carry = tile_carry[blockIdx.x];
	uint32_t cut_any = 0, cut_map = 0;
	for (uint32_t row = 0; row < UP_TILE / MSX_BLOCK; row++) {
		const uint32_t i = blockIdx.x * UP_TILE + row * MSX_BLOCK + threadIdx.x;
		uint32_t fl = 4u;
		int32_t t = -1, own = -1;
		if (i < n_rec) {
			fl = flag[i]; t = tid[i];
			if (names_pool(mode, uv, fl, t)) own = (int32_t)i;
		}
		int32_t inc = own;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const int32_t o = __shfl_up(inc, d, 64);
			if (lane >= d) inc = o > inc ? o : inc;
		}
		__syncthreads();                                 // (s_w of the previous row has been read)
		if (lane == 63) s_w[w] = inc;
		__syncthreads();
		int32_t run = carry, tot = carry;
		for (int q = 0; q < MSX_BLOCK / 64; q++) { if (q < w) run = s_w[q] > run ? s_w[q] : run; tot = s_w[q] > tot ? s_w[q] : tot; }
		int32_t ex = __shfl_up(inc, 1, 64);
		if (lane == 0) ex = -1;
		run = ex > run ? ex : run;                        // nearest naming record before record i
		carry = tot;
		if (i <= n_rec) pidx[i] = run;
		if (i < n_rec) {
			bool b = false;
			if (mode != 0 && rule_sees(mode, uv, fl, t)) {
				const uint8_t *r = u + rec_off[i] + 4;
				if (run >= 0) {
					const uint8_t *p = u + rec_off[run] + 4;
					b = name_differs8(r + 32, r[8], p + 32, p[8]);
				} else if (has_prev) {
					const uint32_t la = r[8];
					b = la != s_plen;
					for (uint32_t k = 0; !b && k < la; k++) b = r[32 + k] != s_prev[k];
				}
			}
			bd[i] = b ? 1 : 0;
			gflag[i] = (b || i == 0u) ? 1u : 0u;
			if (b && i > 0u) {
				cut_any = i;
				if (!(fl & 4u) && i > n_rec / 2u) cut_map = i;
			}
		}
	}
	if (cut_any) atomicMax(&s_cut[0], cut_any);
	if (cut_map) atomicMax(&s_cut[1], cut_map);
	__syncthreads();
	if (threadIdx.x == 0) {
		if (s_cut[0]) atomicMax(&st->cut_any, s_cut[0]);
		if (s_cut[1]) atomicMax(&st->cut_mapped, s_cut[1]);
	}
}

// where the batch ends, its pools, and the name carried into the next one
__global__ void k_pool_cut(uint32_t n_rec, int mode, int last, int cut_mapped, const uint32_t *__restrict__ rec_off,
                           const uint32_t *__restrict__ gpos, const int32_t *__restrict__ pidx, const uint8_t *__restrict__ u,
                           char *__restrict__ prev_name, up_state *st) {
	uint32_t nb = n_rec;
	if (!last && mode != 0) nb = (cut_mapped && st->cut_mapped) ? st->cut_mapped : st->cut_any;   // 0: one pool fills the batch, nothing is final yet
	if (!last && mode == 0) nb = n_rec;
	st->n_batch = nb;
	st->n_groups = mode != 0 ? gpos[nb] : 0u;
	st->cut_off = rec_off[nb];
	const int32_t p = pidx[nb];
	if (p >= 0) {
		const uint8_t *r = u + rec_off[p] + 4;
		const uint32_t lq = r[8];
		for (uint32_t k = 0; k < lq && k < 255u; k++) prev_name[k] = (char)r[32 + k];
		prev_name[lq < 255u ? lq : 255u] = 0;
		st->has_prev = 1;
	}
}

__global__ __launch_bounds__(MSX_BLOCK) void k_pool_offsets(uint32_t n_rec, const uint32_t *__restrict__ gflag,
                                                            const uint32_t *__restrict__ gpos, uint32_t *__restrict__ group_off,
                                                            const up_state *st) {
	const uint32_t i = blockIdx.x * MSX_BLOCK + threadIdx.x;
	const uint32_t nb = st->n_batch;
	if (i == 0) group_off[st->n_groups] = nb;
	if (i < nb && gflag[i]) group_off[gpos[i]] = i;
}

// ---------------------------------------------------------------------------
// filter's output: the emitted records, block_size prefixes included, back to back
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(MSX_BLOCK) void k_emit_len(uint32_t n_emit, const int32_t *__restrict__ emit,
                                                        const uint32_t *__restrict__ rec_off, uint32_t *__restrict__ out_len) {
	const uint32_t k = blockIdx.x * MSX_BLOCK + threadIdx.x;
	if (k < n_emit) { const uint32_t i = (uint32_t)emit[k]; out_len[k] = rec_off[i + 1] - rec_off[i]; }
}

// one wave per 16 consecutive output records; the lanes copy a record's bytes side by side
#define EM_PER_WAVE 16
__global__ __launch_bounds__(MSX_BLOCK) void k_emit_copy(const uint8_t *__restrict__ u, uint32_t n_emit, const int32_t *__restrict__ emit,
                                                         const uint32_t *__restrict__ rec_off, const uint32_t *__restrict__ out_off,
                                                         uint8_t *__restrict__ out, up_state *st, uint32_t *__restrict__ tot_out = nullptr) {
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = (blockIdx.x * MSX_BLOCK + threadIdx.x) >> 6;
	if (wave == 0 && lane == 0) {
		if (tot_out) *tot_out = out_off[n_emit];          // (the overlapped path: the state is the next batch's by the time it is read)
		else st->emit_bytes = st->emit_raw = out_off[n_emit];
	}
	const uint32_t k0 = wave * EM_PER_WAVE;
	for (uint32_t q = 0; q < EM_PER_WAVE; q++) {
		const uint32_t k = k0 + q;
		if (k >= n_emit) return;
		const uint32_t i = (uint32_t)emit[k], s = rec_off[i], len = rec_off[i + 1] - s, d = out_off[k];
		for (uint32_t b = lane; b < len; b += 64u) out[d + b] = u[s + b];
	}
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
extern "C" int msx_unpack_create(msx_ctx *ctx, msx_unpack **out) {
	if (!ctx || !out) return MSX_ERR_ARG;
	*out = nullptr;
	msx_join(ctx);
	msx_unpack *u = new msx_unpack();
	if (hipMalloc((void **)&u->prev_name, 256) != hipSuccess || hipMalloc((void **)&u->d_state, sizeof(up_state)) != hipSuccess ||
	    hipHostMalloc((void **)&u->h_state, sizeof(up_state), hipHostMallocDefault) != hipSuccess) {
		msx_unpack_destroy(ctx, u);
		return msx_fail(ctx, MSX_ERR_NOMEM, "msx_unpack_create: allocation failed");
	}
	(void)hipMemsetAsync(u->prev_name, 0, 256, ctx->stream);
	(void)hipMemsetAsync(u->d_state, 0, sizeof(up_state), ctx->stream);
	if (hipStreamCreateWithFlags(&u->copy_stream, hipStreamNonBlocking) != hipSuccess ||
	    hipEventCreateWithFlags(&u->copy_done, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&u->out_copied, hipEventDisableTiming) != hipSuccess) {
		msx_unpack_destroy(ctx, u);
		return msx_fail(ctx, MSX_ERR_HIP, "msx_unpack_create: stream setup failed");
	}
	*out = u;
	return MSX_OK;
}

extern "C" void msx_unpack_destroy(msx_ctx *ctx, msx_unpack *u) {
	if (!u) return;
	if (ctx && ctx->stream) { msx_join(ctx); (void)hipStreamSynchronize(ctx->stream); }
	for (auto st : u->inf_stream) if (st) (void)hipStreamSynchronize(st);
	if (u->df_stream) (void)hipStreamSynchronize(u->df_stream);
	if (u->copy_stream) (void)hipStreamSynchronize(u->copy_stream);
	// what the context remembers of this unpacker's streams must not outlive them: the encoder's scratch was last used on
	// df_stream (drained just above: nothing of it is in flight), the inflater keeps a scratch set per launching stream
	if (ctx) {
		if (u->df_stream && ctx->df_last == u->df_stream) { ctx->df_last = nullptr; ctx->df_used = false; }
		for (auto &c : ctx->inf)
			if (c.used && c.stream && (c.stream == u->inf_stream[0] || c.stream == u->inf_stream[1])) { c.used = false; c.stream = nullptr; }   // (the buffers stay for the next stream)
	}
	msx_buf *bufs[] = {&u->raw[0], &u->raw[1], &u->seg_first, &u->seg_end, &u->seg_cnt, &u->seg_base, &u->rec_off, &u->flag,
	                   &u->rflags, &u->tid, &u->pos, &u->nm, &u->as, &u->cig_cnt, &u->cig_src, &u->cigar_off, &u->md_len, &u->md_off, &u->md_src,
	                   &u->bd, &u->pidx, &u->gflag, &u->gpos, &u->group_off, &u->tile_last, &u->cigar, &u->md, &u->out_len,
	                   &u->out_off, &u->out, &u->framed, &u->eo[0], &u->eo[1], &u->ef[0], &u->ef[1], &u->comp, &u->blk, &u->blk_status, &u->pre[0].comp, &u->pre[0].blk,
	                   &u->pre[0].status, &u->pre[0].out, &u->pre[0].cnt, &u->pre[1].comp, &u->pre[1].blk, &u->pre[1].status,
	                   &u->pre[1].out, &u->pre[1].cnt, &u->pre[2].comp, &u->pre[2].blk, &u->pre[2].status, &u->pre[2].out, &u->pre[2].cnt};
	for (auto *b : bufs) { if (b->p) (void)hipFree(b->p); b->p = nullptr; b->cap = 0; }
	if (u->copy_stream) { (void)hipStreamSynchronize(u->copy_stream); (void)hipStreamDestroy(u->copy_stream); }
	for (auto st : u->inf_stream) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
	for (auto &ps : u->pre) {
		if (ps.h2d_done) (void)hipEventDestroy(ps.h2d_done);
		if (ps.inf_done) (void)hipEventDestroy(ps.inf_done);
		if (ps.freed) (void)hipEventDestroy(ps.freed);
	}
	if (u->copy_done) (void)hipEventDestroy(u->copy_done);
	if (u->out_copied) (void)hipEventDestroy(u->out_copied);
	if (u->df_stream) (void)hipStreamDestroy(u->df_stream);
	if (u->ev_gathered) (void)hipEventDestroy(u->ev_gathered);
	for (int q = 0; q < 2; q++) {
		if (u->ev_deflated[q]) (void)hipEventDestroy(u->ev_deflated[q]);
		if (u->ev_fetched[q]) (void)hipEventDestroy(u->ev_fetched[q]);
	}
	if (u->d_emit) (void)hipFree(u->d_emit);
	if (u->h_emit) (void)hipHostFree(u->h_emit);
	if (u->prev_name) (void)hipFree(u->prev_name);
	if (u->d_state) (void)hipFree(u->d_state);
	if (u->h_state) (void)hipHostFree(u->h_state);
	delete u;
}

// grow a buffer keeping its first `keep` bytes
static int grow_keep_n(msx_ctx *ctx, msx_buf *b, size_t bytes, size_t keep) {
	if (bytes <= b->cap && b->p) return MSX_OK;
	const size_t want = bytes + bytes / 4 + 4096;
	void *np = nullptr;
	if (hipMalloc(&np, want) != hipSuccess) return msx_fail(ctx, MSX_ERR_NOMEM, "hipMalloc(%zu) failed", want);
	if (b->p) {
		if (keep) (void)hipMemcpyAsync(np, b->p, keep, hipMemcpyDeviceToDevice, ctx->stream);
		(void)hipStreamSynchronize(ctx->stream);
		(void)hipFree(b->p);
	}
	b->p = np;
	b->cap = want;
	return MSX_OK;
}

extern "C" int msx_unpack_seed(msx_ctx *ctx, msx_unpack *u, const uint8_t *carry, size_t n, const char *prev_name) {
	if (!ctx || !u) return MSX_ERR_ARG;
	msx_join(ctx);
	int rc = grow_keep_n(ctx, &u->raw[u->cur], n + 64, 0);
	if (rc) return rc;
	if (n) MSX_HIP(ctx, hipMemcpyAsync(u->raw[u->cur].p, carry, n, hipMemcpyHostToDevice, ctx->stream));
	u->carry_len = n;
	up_state z;
	memset(&z, 0, sizeof z);
	char nm[256];
	memset(nm, 0, sizeof nm);
	if (prev_name) { strncpy(nm, prev_name, 255); z.has_prev = 1; }
	MSX_HIP(ctx, hipMemcpyAsync(u->prev_name, nm, 256, hipMemcpyHostToDevice, ctx->stream));
	MSX_HIP(ctx, hipMemcpyAsync(u->d_state, &z, sizeof z, hipMemcpyHostToDevice, ctx->stream));
	MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));       // (nm and z live on this stack)
	return MSX_OK;
}

// What msx_unpack_finish left for the next batch -- the bytes of the open pool and the cut record, the QNAME of the last
// naming record -- brought to the host, for msx_unpack_seed of ANOTHER unpacker (another context, another GPU): one input
// dealt batch by batch to several devices keeps inflate and walk on the devices, only this hand-over is serial.
// host == NULL or cap too small: *n tells the size, nothing is copied.
extern "C" int msx_unpack_carry(msx_ctx *ctx, msx_unpack *u, uint8_t *host, size_t cap, size_t *n, char name[256], int *has_name) {
	if (!ctx || !u || !n) return MSX_ERR_ARG;
	if (u->enqueued) return msx_fail(ctx, MSX_ERR_ARG, "msx_unpack_carry between msx_unpack_enqueue and msx_unpack_finish");
	msx_join(ctx);
	*n = u->carry_len;
	if (has_name) *has_name = 0;
	if (!host || cap < u->carry_len) return MSX_OK;
	if (u->carry_len) MSX_HIP(ctx, hipMemcpyAsync(host, u->raw[u->cur].p, u->carry_len, hipMemcpyDeviceToHost, ctx->stream));
	up_state z;
	MSX_HIP(ctx, hipMemcpyAsync(&z, u->d_state, sizeof z, hipMemcpyDeviceToHost, ctx->stream));
	if (name) MSX_HIP(ctx, hipMemcpyAsync(name, u->prev_name, 256, hipMemcpyDeviceToHost, ctx->stream));
	MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (has_name) *has_name = z.has_prev != 0;
	return MSX_OK;
}

#define UP_RES(field, bytes) if ((rc = msx_reserve(ctx, &u->field, (bytes)))) return rc
// room in front of a set's inflated bytes for the carry (a larger carry: the batch is copied, as before)
#ifdef MSX_DEBUG_SWITCHES
// (MSX_UP_HEAD=<bytes>, libmsamtools_amd_dbg only: a few hundred bytes make most batches take the copy -- tests)
static size_t up_head() {
	static const size_t v = [] { const char *e = getenv("MSX_UP_HEAD"); const long long n = e ? atoll(e) : 0; return n > 0 ? ((size_t)n + 255) & ~(size_t)255 : (size_t)4 << 20; }();
	return v;
}
#define UP_HEAD up_head()
#else
#define UP_HEAD ((size_t)4 << 20)
#endif

// A new batch is being enqueued: the batches before it are through with their bytes (their walks have finished, their emits --
// which must be enqueued before the next batch is -- lie in front of this point of the context's stream): the sets they were
// walked in may be filled again.
static int up_release_in_place(msx_ctx *ctx, msx_unpack *u) {
	for (int par = 0; par < 2; par++) {
		if (u->in_place[par] < 0) continue;
		msx_unpack::pre_set &ps = u->pre[u->in_place[par]];
		MSX_HIP(ctx, hipEventRecord(ps.freed, ctx->stream));
		ps.used = true;
		u->in_place[par] = -1;
	}
	return MSX_OK;
}

// The bytes of the NEXT msx_unpack_enqueue, sent ahead on a stream of their own: called after msx_unpack_finish of the
// current batch (the carry's length is known then), they travel while the current batch is filtered and its output is
// gathered and downloaded.  The following msx_unpack_enqueue must name the same bytes.
extern "C" int msx_unpack_prefetch(msx_ctx *ctx, msx_unpack *u, const uint8_t *host_bytes, size_t n_new) {
	if (!ctx || !u || (!host_bytes && n_new)) return MSX_ERR_ARG;
	if (u->enqueued) return msx_fail(ctx, MSX_ERR_ARG, "msx_unpack_prefetch between msx_unpack_enqueue and msx_unpack_finish");
	msx_join(ctx);
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	const size_t n = u->carry_len + n_new;
	if (n > 0xfffffff0ull - 64 || n_new == 0) return MSX_OK;             // (msx_unpack_enqueue will say so / nothing to send)
	int rc;
	if ((rc = grow_keep_n(ctx, &u->raw[u->cur], n + 1024, u->carry_len))) return rc;
	MSX_HIP(ctx, hipMemcpyAsync((uint8_t *)u->raw[u->cur].p + u->carry_len, host_bytes, n_new, hipMemcpyHostToDevice, u->copy_stream));
	MSX_HIP(ctx, hipEventRecord(u->copy_done, u->copy_stream));
	u->pre_host = host_bytes;
	u->pre_n = n_new;
	return MSX_OK;
}

static int up_enqueue_walk(msx_ctx *ctx, msx_unpack *u, size_t n, const msx_unpack_params *prm);

extern "C" int msx_unpack_enqueue(msx_ctx *ctx, msx_unpack *u, const uint8_t *host_bytes, size_t n_new, const msx_unpack_params *prm) {
	if (!ctx || !u || !prm || (!host_bytes && n_new)) return MSX_ERR_ARG;
	msx_join(ctx);
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	const size_t n = u->carry_len + n_new;
	if (n > 0xfffffff0ull - 64) return msx_fail(ctx, MSX_ERR_ARG, "msx_unpack_enqueue: more than 4 GiB in one batch");
	int rc;
	const bool sent = u->pre_n != 0 && u->pre_host == host_bytes && u->pre_n == n_new;
	if (u->pre_n != 0 && !sent) {
		MSX_HIP(ctx, hipStreamSynchronize(u->copy_stream));
		u->pre_n = 0;
		return msx_fail(ctx, MSX_ERR_ARG, "msx_unpack_enqueue: other bytes than msx_unpack_prefetch sent ahead");
	}
	u->pre_n = 0;
	up_release_in_place(ctx, u);
	if ((rc = grow_keep_n(ctx, &u->raw[u->cur], n + 1024, u->carry_len))) return rc;     // (a corrupt record's name length may point 255 bytes past the data)
	uint8_t *raw = (uint8_t *)u->raw[u->cur].p;
	u->bytes[u->cur] = raw;
	// the state of this batch (has_prev and prev_name carry over)
	MSX_HIP(ctx, hipMemsetAsync(u->d_state, 0, offsetof(up_state, has_prev), ctx->stream));
	MSX_HIP(ctx, hipMemsetAsync(&u->d_state->emit_bytes, 0, 4, ctx->stream));
	if (sent) MSX_HIP(ctx, hipStreamWaitEvent(ctx->stream, u->copy_done, 0));
	else if (n_new) MSX_HIP(ctx, hipMemcpyAsync(raw + u->carry_len, host_bytes, n_new, hipMemcpyHostToDevice, ctx->stream));
	u->bgzf = false;
	return up_enqueue_walk(ctx, u, n, prm);
}

// the batch's bytes are (or will be, in stream order) in raw[cur][0, n): find the records
static int up_enqueue_walk(msx_ctx *ctx, msx_unpack *u, size_t n, const msx_unpack_params *prm) {
	int rc;
	uint8_t *raw = u->bytes[u->cur];
	MSX_HIP(ctx, hipMemsetAsync(raw + n, 0, 64, ctx->stream));
	u->n_bytes = n;
	u->prm = *prm;
	u->enqueued = true;
	const uint32_t nseg = (uint32_t)((n + UP_SEG - 1) / UP_SEG);
	const size_t cap = n / 36 + 16;               // records: a record is at least 37 bytes with its block_size
	UP_RES(seg_first, (size_t)(nseg + 2) * 4); UP_RES(seg_end, (size_t)(nseg + 2) * 4);
	UP_RES(seg_cnt, (size_t)(nseg + 2) * 4); UP_RES(seg_base, (size_t)(nseg + 2) * 4);
	UP_RES(rec_off, (cap + 2) * 4);
	if (nseg == 0) return MSX_OK;
	hipLaunchKernelGGL(k_chase_walk, dim3((nseg + MSX_BLOCK - 1) / MSX_BLOCK), dim3(MSX_BLOCK), 0, ctx->stream, raw, (uint32_t)n, nseg,
	                   prm->n_targets, (uint32_t *)u->seg_first.p, (uint32_t *)u->seg_end.p, (uint32_t *)u->seg_cnt.p);
	hipLaunchKernelGGL(k_chase_check, dim3((nseg + MSX_BLOCK - 1) / MSX_BLOCK), dim3(MSX_BLOCK), 0, ctx->stream, (uint32_t)n, nseg,
	                   (const uint32_t *)u->seg_first.p, (const uint32_t *)u->seg_end.p, (uint32_t *)u->seg_cnt.p, u->d_state);
	hipLaunchKernelGGL(k_chase_fix, dim3(1), dim3(1), 0, ctx->stream, raw, (uint32_t)n, nseg, (uint32_t *)u->seg_first.p,
	                   (uint32_t *)u->seg_end.p, (uint32_t *)u->seg_cnt.p, u->d_state);
	if ((rc = msx_scan_u32(ctx, (const uint32_t *)u->seg_cnt.p, (uint32_t *)u->seg_base.p, nseg))) return rc;
	hipLaunchKernelGGL(k_chase_total, dim3(1), dim3(1), 0, ctx->stream, nseg, (const uint32_t *)u->seg_base.p,
	                   (const uint32_t *)u->seg_end.p, u->d_state);
	hipLaunchKernelGGL(k_chase_write, dim3((nseg + MSX_BLOCK - 1) / MSX_BLOCK), dim3(MSX_BLOCK), 0, ctx->stream, raw, (uint32_t)n, nseg,
	                   (const uint32_t *)u->seg_first.p, (const uint32_t *)u->seg_cnt.p, (const uint32_t *)u->seg_base.p,
	                   (uint32_t *)u->rec_off.p, (const up_state *)u->d_state);
	MSX_HIP(ctx, hipGetLastError());
	return MSX_OK;
}

static int up_check_table(msx_ctx *ctx, const msx_bgzf_block *host_blocks, int64_t n_blocks, size_t comp_len, size_t *n_new_out) {
	size_t n_new = 0;
	for (int64_t i = 0; i < n_blocks; i++) {
		const msx_bgzf_block &b = host_blocks[i];
		if (b.out_off != n_new || b.out_len > 65536u || b.in_off > comp_len || b.in_len > comp_len - b.in_off)
			return msx_fail(ctx, MSX_ERR_ARG, "BGZF block %lld of the table is inconsistent", (long long)i);
		n_new += b.out_len;
	}
	*n_new_out = n_new;
	return MSX_OK;
}

// The bytes of a coming msx_unpack_enqueue_bgzf, sent ahead and inflated on streams of their own.  May be called at any
// time (also between msx_unpack_enqueue* and msx_unpack_finish of the current batch): nothing here depends on the carry.
// Up to two batches may be on their way; they are consumed in the order they were sent.
extern "C" int msx_unpack_prefetch_bgzf(msx_ctx *ctx, msx_unpack *u, const uint8_t *host_comp, size_t comp_len,
                                        const msx_bgzf_block *host_blocks, int64_t n_blocks) {
	if (!ctx || !u || n_blocks <= 0 || !host_comp || !host_blocks) return MSX_ERR_ARG;
	if (u->ahead_n >= 2) return msx_fail(ctx, MSX_ERR_ARG, "msx_unpack_prefetch_bgzf: two batches are on their way already");
	if (n_blocks > (1 << 24)) return msx_fail(ctx, MSX_ERR_ARG, "msx_unpack_prefetch_bgzf: too many blocks");
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	int rc;
	size_t n_new = 0;
	if ((rc = up_check_table(ctx, host_blocks, n_blocks, comp_len, &n_new))) return rc;
	if (!u->pre[0].inf_done)
		for (auto &ps : u->pre) {
			MSX_HIP(ctx, hipEventCreateWithFlags(&ps.h2d_done, hipEventDisableTiming));
			MSX_HIP(ctx, hipEventCreateWithFlags(&ps.inf_done, hipEventDisableTiming));
			MSX_HIP(ctx, hipEventCreateWithFlags(&ps.freed, hipEventDisableTiming));
		}
	// the stream: by how many batches are on their way already (0 or 1) -- the second one only if a caller ever sends two ahead
	if (!u->inf_stream[u->ahead_n]) {
		int lo = 0, hi = 0;
		MSX_HIP(ctx, hipDeviceGetStreamPriorityRange(&lo, &hi));           // (lo: the least urgent)
		MSX_HIP(ctx, hipStreamCreateWithPriority(&u->inf_stream[u->ahead_n], hipStreamNonBlocking, lo));
	}
	hipStream_t inf = u->inf_stream[u->ahead_n];
	const int set = (u->ahead_head + u->ahead_n) % 3;
	msx_unpack::pre_set &ps = u->pre[set];
	// (the set may still be read by the batch that was walked in it: its upload waits for that batch's release; buffers are
	// grown only when nothing of the set is in flight any more)
	if (ps.used) MSX_HIP(ctx, hipStreamWaitEvent(inf, ps.freed, 0));
	if (ps.comp.cap < comp_len + 64 || ps.blk.cap < (size_t)n_blocks * sizeof(msx_bgzf_block) || ps.status.cap < (size_t)n_blocks * 4 ||
	    ps.out.cap < UP_HEAD + n_new + 2048 || ps.cnt.cap < 64) {
		if (ps.used) MSX_HIP(ctx, hipEventSynchronize(ps.freed));
		for (auto st : u->inf_stream) if (st) MSX_HIP(ctx, hipStreamSynchronize(st));
	}
	if ((rc = msx_reserve(ctx, &ps.comp, comp_len + 64))) return rc;
	if ((rc = msx_reserve(ctx, &ps.blk, (size_t)n_blocks * sizeof(msx_bgzf_block)))) return rc;
	if ((rc = msx_reserve(ctx, &ps.status, (size_t)n_blocks * 4))) return rc;
	if ((rc = msx_reserve(ctx, &ps.out, UP_HEAD + n_new + 2048))) return rc;       // (+ what the walk may read and clear behind the bytes)
	if ((rc = msx_reserve(ctx, &ps.cnt, 64))) return rc;
	// the upload, and the inflater behind it on the same stream (a second batch sent ahead travels on the other stream while
	// this one is being inflated)
	MSX_HIP(ctx, hipMemcpyAsync(ps.comp.p, host_comp, comp_len, hipMemcpyHostToDevice, inf));
	MSX_HIP(ctx, hipMemcpyAsync(ps.blk.p, host_blocks, (size_t)n_blocks * sizeof(msx_bgzf_block), hipMemcpyHostToDevice, inf));
	MSX_HIP(ctx, hipEventRecord(ps.h2d_done, inf));
	MSX_HIP(ctx, hipMemsetAsync(ps.cnt.p, 0, 8, inf));
	if ((rc = msx_bgzf_inflate_launch(ctx, inf, 8, (const uint8_t *)ps.comp.p, comp_len, (const msx_bgzf_block *)ps.blk.p,
	                                  n_blocks, (uint8_t *)ps.out.p + UP_HEAD, (uint32_t *)ps.status.p, (uint32_t *)ps.cnt.p)))
		return rc;
	MSX_HIP(ctx, hipEventRecord(ps.inf_done, inf));
	ps.key = host_comp;
	ps.comp_len = comp_len;
	ps.nblk = n_blocks;
	ps.total = n_new;
	u->ahead_n++;
	return MSX_OK;
}

// The batch's new bytes arrive compressed: BGZF payloads and their table (msx_bgzf_block, out_off from the batch's first
// new byte).  They are inflated behind the carry, where msx_unpack_enqueue would have copied them.
extern "C" int msx_unpack_enqueue_bgzf(msx_ctx *ctx, msx_unpack *u, const uint8_t *host_comp, size_t comp_len,
                                       const msx_bgzf_block *host_blocks, int64_t n_blocks, const msx_unpack_params *prm) {
	if (!ctx || !u || !prm || n_blocks < 0 || (n_blocks > 0 && (!host_comp || !host_blocks))) return MSX_ERR_ARG;
	if (u->pre_n != 0) return msx_fail(ctx, MSX_ERR_ARG, "msx_unpack_enqueue_bgzf after msx_unpack_prefetch");
	if (u->enqueued) return msx_fail(ctx, MSX_ERR_ARG, "msx_unpack_enqueue_bgzf between msx_unpack_enqueue and msx_unpack_finish");
	if (n_blocks > (1 << 24)) return msx_fail(ctx, MSX_ERR_ARG, "msx_unpack_enqueue_bgzf: too many blocks");
	msx_join(ctx);
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	int rc;
	size_t n_new = 0;
	msx_unpack::pre_set &ps = u->pre[u->ahead_head];
	const bool sent = u->ahead_n > 0 && ps.key == host_comp && ps.comp_len == comp_len && ps.nblk == n_blocks;
	if (u->ahead_n > 0 && !sent) {
		for (auto st : u->inf_stream) if (st) MSX_HIP(ctx, hipStreamSynchronize(st));
		u->ahead_n = 0;
		return msx_fail(ctx, MSX_ERR_ARG, "msx_unpack_enqueue_bgzf: other blocks than msx_unpack_prefetch_bgzf sent ahead");
	}
	if (sent) n_new = ps.total;
	else if ((rc = up_check_table(ctx, host_blocks, n_blocks, comp_len, &n_new))) return rc;
	const size_t n = u->carry_len + n_new;
	if (n > 0xfffffff0ull - 64) return msx_fail(ctx, MSX_ERR_ARG, "msx_unpack_enqueue_bgzf: more than 4 GiB in one batch");
	up_release_in_place(ctx, u);
	const bool in_place = sent && n_new > 0 && u->carry_len <= UP_HEAD;
	if (!in_place && (rc = grow_keep_n(ctx, &u->raw[u->cur], n + 1024, u->carry_len))) return rc;
	uint8_t *raw = (uint8_t *)u->raw[u->cur].p;
	MSX_HIP(ctx, hipMemsetAsync(u->d_state, 0, offsetof(up_state, has_prev), ctx->stream));
	MSX_HIP(ctx, hipMemsetAsync(&u->d_state->emit_bytes, 0, 4, ctx->stream));
	if (sent) {
		MSX_HIP(ctx, hipStreamWaitEvent(ctx->stream, ps.inf_done, 0));
		if (in_place) {
			// inflated already, UP_HEAD bytes into the set's buffer: the carry in front of it, the batch walked where it lies
			raw = (uint8_t *)ps.out.p + UP_HEAD - u->carry_len;
			if (u->carry_len) MSX_HIP(ctx, hipMemcpyAsync(raw, u->raw[u->cur].p, u->carry_len, hipMemcpyDeviceToDevice, ctx->stream));
			u->in_place[u->cur] = u->ahead_head;
		} else {
			// (a carry larger than the room in front: the batch behind the carry, as until round 6)
			if (n_new) MSX_HIP(ctx, hipMemcpyAsync(raw + u->carry_len, (const uint8_t *)ps.out.p + UP_HEAD, n_new, hipMemcpyDeviceToDevice, ctx->stream));
			MSX_HIP(ctx, hipEventRecord(ps.freed, ctx->stream));
			ps.used = true;
		}
		MSX_HIP(ctx, hipMemcpyAsync(&u->d_state->inflate_bad, ps.cnt.p, 4, hipMemcpyDeviceToDevice, ctx->stream));
		u->ahead_head = (u->ahead_head + 1) % 3;
		u->ahead_n--;
	} else if (n_blocks > 0) {
		UP_RES(comp, comp_len + 64);
		UP_RES(blk, (size_t)n_blocks * sizeof(msx_bgzf_block));
		UP_RES(blk_status, (size_t)n_blocks * 4);
		MSX_HIP(ctx, hipMemcpyAsync(u->comp.p, host_comp, comp_len, hipMemcpyHostToDevice, ctx->stream));
		MSX_HIP(ctx, hipMemcpyAsync(u->blk.p, host_blocks, (size_t)n_blocks * sizeof(msx_bgzf_block), hipMemcpyHostToDevice, ctx->stream));
		if ((rc = msx_bgzf_inflate_launch(ctx, ctx->stream, 0, (const uint8_t *)u->comp.p, comp_len, (const msx_bgzf_block *)u->blk.p, n_blocks,
		                                  raw + u->carry_len, (uint32_t *)u->blk_status.p, &u->d_state->inflate_bad)))
			return rc;
	}
	u->bgzf = true;
	u->bytes[u->cur] = raw;
	return up_enqueue_walk(ctx, u, n, prm);
}

static int up_fetch_state(msx_ctx *ctx, msx_unpack *u) {
	MSX_HIP(ctx, hipMemcpyAsync(u->h_state, u->d_state, sizeof(up_state), hipMemcpyDeviceToHost, ctx->stream));
	MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return MSX_OK;
}

extern "C" int msx_unpack_finish(msx_ctx *ctx, msx_unpack *u, msx_unpack_result *res, msx_batch *dev) {
	if (!ctx || !u || !res || !dev) return MSX_ERR_ARG;
	if (!u->enqueued) return msx_fail(ctx, MSX_ERR_ARG, "msx_unpack_finish without msx_unpack_enqueue");
	u->enqueued = false;
	msx_join(ctx);
	memset(res, 0, sizeof *res);
	memset(dev, 0, sizeof *dev);
	const msx_unpack_params &P = u->prm;
	const size_t n = u->n_bytes;
	uint8_t *raw = u->bytes[u->cur];
	int rc;
	if (n > 0) {
		if ((rc = up_fetch_state(ctx, u))) return rc;               // sync 1: how many records
		if (u->bgzf && u->h_state->inflate_bad)                      // (nothing consumed: the carry stands where it stood)
			return msx_fail(ctx, MSX_ERR_INFLATE, "%u BGZF block(s) of the batch refused by the device inflater", u->h_state->inflate_bad);
		if (u->h_state->status == MSX_UP_CORRUPT) return msx_fail(ctx, MSX_ERR_ARG, "Corrupt BAM record");
	} else {
		memset(u->h_state, 0, sizeof(up_state));
	}
	const uint32_t nr = n ? u->h_state->n_total : 0u;
	u->n_total = nr;
	res->bad_guesses = n ? u->h_state->bad_segments : 0;
	if (P.last && n && u->h_state->tail_off != (uint32_t)n) return msx_fail(ctx, MSX_ERR_ARG, "Truncated BAM record");
	if (nr == 0) {
		// nothing complete yet (or nothing at all): everything stays as the carry -- which lives at the head of raw[cur]
		if (n && raw != (uint8_t *)u->raw[u->cur].p) {              // (the batch was walked in its set: over it goes)
			if ((rc = grow_keep_n(ctx, &u->raw[u->cur], n + 1024, 0))) return rc;
			MSX_HIP(ctx, hipMemcpyAsync(u->raw[u->cur].p, raw, n, hipMemcpyDeviceToDevice, ctx->stream));
			u->bytes[u->cur] = (uint8_t *)u->raw[u->cur].p;
		}
		res->carry_bytes = (int64_t)n;
		u->carry_len = n;
		return MSX_OK;
	}
	const size_t c = (size_t)nr + 8;
	UP_RES(flag, c * 2); UP_RES(rflags, c); UP_RES(tid, c * 4); UP_RES(pos, c * 4); UP_RES(nm, c * 4); UP_RES(as, c * 4);
	UP_RES(cig_cnt, c * 4); UP_RES(cig_src, c * 4); UP_RES(cigar_off, (c + 1) * 4); UP_RES(md_len, c * 4); UP_RES(md_off, (c + 1) * 4); UP_RES(md_src, c * 4);
	UP_RES(bd, c); UP_RES(pidx, (c + 1) * 4); UP_RES(gflag, c * 4); UP_RES(gpos, (c + 1) * 4); UP_RES(group_off, (c + 1) * 4);
	const uint32_t n_tiles = (nr + 1 + UP_TILE - 1) / UP_TILE;
	UP_RES(tile_last, (size_t)(n_tiles + 2) * 4);
	const unsigned gr = (nr + MSX_BLOCK - 1) / MSX_BLOCK;
	const uint32_t *rec_off = (const uint32_t *)u->rec_off.p;
	hipLaunchKernelGGL(k_rec_fields, dim3(gr), dim3(MSX_BLOCK), 0, ctx->stream, raw, nr, rec_off, P.want_aux, P.want_stats,
	                   (uint16_t *)u->flag.p, (uint8_t *)u->rflags.p, (int32_t *)u->tid.p, (int32_t *)u->pos.p, (int32_t *)u->nm.p,
	                   (int32_t *)u->as.p, (uint32_t *)u->cig_cnt.p, (uint32_t *)u->cig_src.p, (uint32_t *)u->md_len.p, (uint32_t *)u->md_src.p, u->d_state);
	if (P.want_stats) {
		if ((rc = msx_scan_u32(ctx, (const uint32_t *)u->cig_cnt.p, (uint32_t *)u->cigar_off.p, nr))) return rc;
		if ((rc = msx_scan_u32(ctx, (const uint32_t *)u->md_len.p, (uint32_t *)u->md_off.p, nr))) return rc;
		// (the payload arrays are bounded by the bytes they come from)
		UP_RES(cigar, n + 64);
		UP_RES(md, n + 64);
		hipLaunchKernelGGL(k_rec_payload, dim3(gr), dim3(MSX_BLOCK), 0, ctx->stream, raw, nr, rec_off, (const uint32_t *)u->cigar_off.p,
		                   (const uint32_t *)u->md_off.p, (const uint32_t *)u->md_src.p, (const uint32_t *)u->cig_src.p, (uint32_t *)u->cigar.p, (uint8_t *)u->md.p);
	}
	if (P.pool_mode != 0) {
		hipLaunchKernelGGL(k_name_tile_last, dim3(n_tiles), dim3(MSX_BLOCK), 0, ctx->stream, nr, P.pool_mode, P.unmapped_visible,
		                   (const uint16_t *)u->flag.p, (const int32_t *)u->tid.p, (int32_t *)u->tile_last.p);
		hipLaunchKernelGGL(k_name_tile_scan, dim3(1), dim3(MSX_BLOCK), 0, ctx->stream, n_tiles, (int32_t *)u->tile_last.p);
		hipLaunchKernelGGL(k_name_bounds, dim3(n_tiles), dim3(MSX_BLOCK), 0, ctx->stream, raw, nr, P.pool_mode, P.unmapped_visible, rec_off,
		                   (const uint16_t *)u->flag.p, (const int32_t *)u->tid.p, (const int32_t *)u->tile_last.p,
		                   (const char *)u->prev_name, (uint8_t *)u->bd.p, (int32_t *)u->pidx.p, (uint32_t *)u->gflag.p, u->d_state);
		if ((rc = msx_scan_u32(ctx, (const uint32_t *)u->gflag.p, (uint32_t *)u->gpos.p, nr))) return rc;
	} else {
		MSX_HIP(ctx, hipMemsetAsync(u->pidx.p, 0xff, (c + 1) * 4, ctx->stream));
	}
	hipLaunchKernelGGL(k_pool_cut, dim3(1), dim3(1), 0, ctx->stream, nr, P.pool_mode, P.last, P.cut_mapped, rec_off,
	                   (const uint32_t *)u->gpos.p, (const int32_t *)u->pidx.p, raw, u->prev_name, u->d_state);
	if (P.pool_mode != 0)
		hipLaunchKernelGGL(k_pool_offsets, dim3(gr), dim3(MSX_BLOCK), 0, ctx->stream, nr, (const uint32_t *)u->gflag.p,
		                   (const uint32_t *)u->gpos.p, (uint32_t *)u->group_off.p, (const up_state *)u->d_state);
	MSX_HIP(ctx, hipGetLastError());
	if ((rc = up_fetch_state(ctx, u))) return rc;                   // sync 2: where the batch ends
	if (u->h_state->status == MSX_UP_CORRUPT) return msx_fail(ctx, MSX_ERR_ARG, "Corrupt BAM record");
	const up_state &S = *u->h_state;
	u->n_batch = S.n_batch;
	u->n_groups = S.n_groups;
	res->n_records = S.n_batch;
	res->n_groups = S.n_groups;
	res->bytes_consumed = S.cut_off;
	res->carry_bytes = (int64_t)n - (int64_t)S.cut_off;
	// the bytes behind the batch become the head of the next one (in the other buffer: this one stays valid for
	// msx_unpack_emit_enqueue and for the views handed out below)
	{
		const int nxt = u->cur ^ 1;
		const size_t cl = n - S.cut_off;
		if ((rc = grow_keep_n(ctx, &u->raw[nxt], cl + 1024, 0))) return rc;
		if (cl) MSX_HIP(ctx, hipMemcpyAsync(u->raw[nxt].p, raw + S.cut_off, cl, hipMemcpyDeviceToDevice, ctx->stream));
		u->carry_len = cl;
		u->cur = nxt;
	}
	dev->n_records = S.n_batch;
	dev->n_groups = P.pool_mode != 0 ? (int64_t)S.n_groups : 0;
	dev->flag = (const uint16_t *)u->flag.p; dev->rflags = (const uint8_t *)u->rflags.p;
	dev->tid = (const int32_t *)u->tid.p; dev->pos = (const int32_t *)u->pos.p;
	dev->nm = (const int32_t *)u->nm.p; dev->as = (const int32_t *)u->as.p;
	if (P.want_stats) {
		dev->cigar_off = (const uint32_t *)u->cigar_off.p; dev->cigar = (const uint32_t *)u->cigar.p;
		dev->md_off = (const uint32_t *)u->md_off.p; dev->md = (const uint8_t *)u->md.p;
	}
	if (P.pool_mode != 0) dev->group_off = (const uint32_t *)u->group_off.p;
	dev->pool_rule = P.pool_mode == 1 ? MSX_POOLS_FILTER : MSX_POOLS_PROFILE;
	return MSX_OK;
}

// Filter's output records gathered into one byte string on the device (block_size prefixes included, output order);
// *n_bytes tells how long it is.  msx_unpack_emit_fetch brings it down.
static int up_emit_gather(msx_ctx *ctx, msx_unpack *u, const int32_t *emit_idx_dev, int64_t n_emit, int level) {
	msx_join(ctx);
	u->gathered = 0;
	u->fetch_src = nullptr;
	u->fetch_par = -1;
	if (n_emit <= 0) return MSX_OK;
	// (the gather buffer may still be on its way down for the batch before)
	if (u->out_busy) { MSX_HIP(ctx, hipStreamWaitEvent(ctx->stream, u->out_copied, 0)); u->out_busy = false; }
	if (n_emit > u->n_batch) return msx_fail(ctx, MSX_ERR_ARG, "msx_unpack_emit: more records than the batch holds");
	int rc;
	const uint32_t ne = (uint32_t)n_emit;
	// the batch's bytes are in the buffer the carry was NOT copied into
	const uint8_t *raw = u->bytes[u->cur ^ 1];
	UP_RES(out_len, ((size_t)ne + 8) * 4);
	UP_RES(out_off, ((size_t)ne + 8) * 4);
	UP_RES(out, u->n_bytes + 64);
	hipLaunchKernelGGL(k_emit_len, dim3((ne + MSX_BLOCK - 1) / MSX_BLOCK), dim3(MSX_BLOCK), 0, ctx->stream, ne, emit_idx_dev,
	                   (const uint32_t *)u->rec_off.p, (uint32_t *)u->out_len.p);
	if ((rc = msx_scan_u32(ctx, (const uint32_t *)u->out_len.p, (uint32_t *)u->out_off.p, ne))) return rc;
	const uint32_t n_waves = (ne + EM_PER_WAVE - 1) / EM_PER_WAVE;
	hipLaunchKernelGGL(k_emit_copy, dim3((n_waves + 3) / 4), dim3(MSX_BLOCK), 0, ctx->stream, raw, ne, emit_idx_dev,
	                   (const uint32_t *)u->rec_off.p, (const uint32_t *)u->out_off.p, (uint8_t *)u->out.p, u->d_state);
	MSX_HIP(ctx, hipGetLastError());
	if (level == 0) {
		// -u: framed on the device as stored blocks; the launch is sized by the batch's bytes, the kernel reads how many
		// were emitted (out_off[n_emit]) where the scan has left it
		UP_RES(framed, (size_t)msx_bgzf_bound((int64_t)u->n_bytes + 64, 0) + 64);
		if ((rc = msx_bgzf_store_launch(ctx, ctx->stream, (const uint8_t *)u->out.p, (const uint32_t *)u->out_off.p + ne, u->n_bytes,
		                                (uint8_t *)u->framed.p)))
			return rc;
	} else if (level > 0) {
		// -b: deflated on the device, the blocks moved back to back; the stream's length lands in the state
		UP_RES(framed, (size_t)msx_bgzf_bound((int64_t)u->n_bytes + 64, level) + 64);
		if ((rc = msx_bgzf_deflate_launch(ctx, ctx->stream, (const uint8_t *)u->out.p, (const uint32_t *)u->out_off.p + ne, u->n_bytes,
		                                  (uint8_t *)u->framed.p, &u->d_state->emit_bytes, level)))
			return rc;
	}
	if ((rc = up_fetch_state(ctx, u))) return rc;                   // sync: how many bytes
	return MSX_OK;
}

extern "C" int msx_unpack_emit_gather(msx_ctx *ctx, msx_unpack *u, const int32_t *emit_idx_dev, int64_t n_emit, int64_t *n_bytes) {
	if (!ctx || !u || !n_bytes || (n_emit > 0 && !emit_idx_dev)) return MSX_ERR_ARG;
	*n_bytes = 0;
	int rc = up_emit_gather(ctx, u, emit_idx_dev, n_emit, -1);
	if (rc || n_emit <= 0) return rc;
	u->gathered = u->h_state->emit_bytes;
	u->fetch_src = u->out.p;
	*n_bytes = (int64_t)u->gathered;
	return MSX_OK;
}

// The same, and the byte string cut into BGZF blocks on the device (msx_deflate.hip): what comes down is what goes into
// the file.  level 0: stored blocks (filter -bu); level >= 1: DEFLATE (filter -b).  *n_bytes: bytes of finished blocks, *n_blocks: how many; the blocks of
// one call are full except the last.
extern "C" int msx_unpack_emit_gather_bgzf(msx_ctx *ctx, msx_unpack *u, const int32_t *emit_idx_dev, int64_t n_emit, int level,
                                           int64_t *n_bytes, int64_t *n_blocks) {
	if (!ctx || !u || !n_bytes || (n_emit > 0 && !emit_idx_dev)) return MSX_ERR_ARG;
	if (level < 0 || level > 9) return msx_fail(ctx, MSX_ERR_ARG, "msx_unpack_emit_gather_bgzf: level %d", level);
	*n_bytes = 0;
	if (n_blocks) *n_blocks = 0;
	int rc = up_emit_gather(ctx, u, emit_idx_dev, n_emit, level);
	if (rc || n_emit <= 0) return rc;
	const int64_t raw_bytes = (int64_t)u->h_state->emit_raw;
	u->gathered = level == 0 ? (size_t)msx_bgzf_bound(raw_bytes, 0) : (size_t)u->h_state->emit_bytes;
	u->fetch_src = u->framed.p;
	*n_bytes = (int64_t)u->gathered;
	if (n_blocks) *n_blocks = (raw_bytes + 0xff00 - 1) / 0xff00;
	return MSX_OK;
}

// The same in two steps that need not follow each other at once (level >= 1: filter -b).  _enqueue gathers batch k's records
// on the context's stream and hands them to the encoder on a stream of its own, then returns; the caller goes on with batch
// k + 1 (walk, filter) while batch k is being deflated.  _complete -- called once per _enqueue, in the same order -- waits for the
// oldest batch's blocks, tells their size and makes them what the next msx_unpack_emit_fetch brings down.  Two batches may be
// in flight (one being deflated, one whose blocks are travelling to the host).
extern "C" int msx_unpack_emit_bgzf_enqueue(msx_ctx *ctx, msx_unpack *u, const int32_t *emit_idx_dev, int64_t n_emit, int level) {
	if (!ctx || !u || (n_emit > 0 && !emit_idx_dev)) return MSX_ERR_ARG;
	if (level < 1 || level > 9) return msx_fail(ctx, MSX_ERR_ARG, "msx_unpack_emit_bgzf_enqueue: level %d", level);
	if (u->eseq_in - u->eseq_out >= 2) return msx_fail(ctx, MSX_ERR_ARG, "msx_unpack_emit_bgzf_enqueue: two batches are in flight already");
	msx_join(ctx);
	int rc;
	if (!u->df_stream) {
		if (hipStreamCreateWithFlags(&u->df_stream, hipStreamNonBlocking) != hipSuccess ||
		    hipEventCreateWithFlags(&u->ev_gathered, hipEventDisableTiming) != hipSuccess ||
		    hipEventCreateWithFlags(&u->ev_deflated[0], hipEventDisableTiming) != hipSuccess ||
		    hipEventCreateWithFlags(&u->ev_deflated[1], hipEventDisableTiming) != hipSuccess ||
		    hipEventCreateWithFlags(&u->ev_fetched[0], hipEventDisableTiming) != hipSuccess ||
		    hipEventCreateWithFlags(&u->ev_fetched[1], hipEventDisableTiming) != hipSuccess ||
		    hipMalloc((void **)&u->d_emit, 16) != hipSuccess || hipHostMalloc((void **)&u->h_emit, 16, hipHostMallocDefault) != hipSuccess)
			return msx_fail(ctx, MSX_ERR_HIP, "msx_unpack_emit_bgzf_enqueue: stream setup failed");
	}
	const int par = (int)(u->eseq_in & 1u);
	u->eseq_in++;
	u->e_nemit[par] = n_emit > 0 ? n_emit : 0;
	if (n_emit <= 0) return MSX_OK;
	if (n_emit > u->n_batch) return msx_fail(ctx, MSX_ERR_ARG, "msx_unpack_emit: more records than the batch holds");
	const uint32_t ne = (uint32_t)n_emit;
	const uint8_t *raw = u->bytes[u->cur ^ 1];
	// (this parity's record buffer was read by the encoder two batches ago)
	if (u->deflated_used[par]) MSX_HIP(ctx, hipStreamWaitEvent(ctx->stream, u->ev_deflated[par], 0));
	UP_RES(out_len, ((size_t)ne + 8) * 4);
	UP_RES(out_off, ((size_t)ne + 8) * 4);
	UP_RES(eo[par], u->n_bytes + 64);
	hipLaunchKernelGGL(k_emit_len, dim3((ne + MSX_BLOCK - 1) / MSX_BLOCK), dim3(MSX_BLOCK), 0, ctx->stream, ne, emit_idx_dev,
	                   (const uint32_t *)u->rec_off.p, (uint32_t *)u->out_len.p);
	if ((rc = msx_scan_u32(ctx, (const uint32_t *)u->out_len.p, (uint32_t *)u->out_off.p, ne))) return rc;
	const uint32_t n_waves = (ne + EM_PER_WAVE - 1) / EM_PER_WAVE;
	hipLaunchKernelGGL(k_emit_copy, dim3((n_waves + 3) / 4), dim3(MSX_BLOCK), 0, ctx->stream, raw, ne, emit_idx_dev,
	                   (const uint32_t *)u->rec_off.p, (const uint32_t *)u->out_off.p, (uint8_t *)u->eo[par].p, u->d_state, u->d_emit + 2 * par + 1);
	MSX_HIP(ctx, hipGetLastError());
	MSX_HIP(ctx, hipEventRecord(u->ev_gathered, ctx->stream));
	// the encoder's stream: behind the gather, and behind the journey of the blocks this parity held before
	MSX_HIP(ctx, hipStreamWaitEvent(u->df_stream, u->ev_gathered, 0));
	if (u->fetched_used[par]) MSX_HIP(ctx, hipStreamWaitEvent(u->df_stream, u->ev_fetched[par], 0));
	UP_RES(ef[par], (size_t)msx_bgzf_bound((int64_t)u->n_bytes + 64, level) + 64);
	if ((rc = msx_bgzf_deflate_launch(ctx, u->df_stream, (const uint8_t *)u->eo[par].p, u->d_emit + 2 * par + 1, u->n_bytes, (uint8_t *)u->ef[par].p,
	                                  u->d_emit + 2 * par, level)))
		return rc;
	MSX_HIP(ctx, hipMemcpyAsync(u->h_emit + 2 * par, u->d_emit + 2 * par, 8, hipMemcpyDeviceToHost, u->df_stream));
	MSX_HIP(ctx, hipEventRecord(u->ev_deflated[par], u->df_stream));
	u->deflated_used[par] = true;
	return MSX_OK;
}

extern "C" int msx_unpack_emit_bgzf_complete(msx_ctx *ctx, msx_unpack *u, int64_t *n_bytes, int64_t *n_blocks) {
	if (!ctx || !u || !n_bytes) return MSX_ERR_ARG;
	if (u->eseq_out >= u->eseq_in) return msx_fail(ctx, MSX_ERR_ARG, "msx_unpack_emit_bgzf_complete without msx_unpack_emit_bgzf_enqueue");
	const int par = (int)(u->eseq_out & 1u);
	u->eseq_out++;
	*n_bytes = 0;
	if (n_blocks) *n_blocks = 0;
	u->gathered = 0;
	u->fetch_src = nullptr;
	u->fetch_par = par;
	if (u->e_nemit[par] <= 0) return MSX_OK;
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	MSX_HIP(ctx, hipEventSynchronize(u->ev_deflated[par]));
	u->gathered = u->h_emit[2 * par];
	u->fetch_src = u->ef[par].p;
	*n_bytes = (int64_t)u->gathered;
	if (n_blocks) *n_blocks = ((int64_t)u->h_emit[2 * par + 1] + 0xff00 - 1) / 0xff00;
	return MSX_OK;
}

// The gathered bytes to the host.  done == NULL: waits for them.  Otherwise they travel on a copy stream of their own while
// the caller goes on with the next batch; msx_event_wait(done) before host_out is read.
extern "C" int msx_unpack_emit_fetch(msx_ctx *ctx, msx_unpack *u, uint8_t *host_out, size_t host_cap, msx_event *done) {
	if (!ctx || !u || (u->gathered && !host_out)) return MSX_ERR_ARG;
	if (done) done->recorded = false;
	const size_t tb = u->gathered;
	if (tb == 0) return MSX_OK;
	if (tb > host_cap) return msx_fail(ctx, MSX_ERR_ARG, "msx_unpack_emit: output buffer too small (%zu > %zu)", tb, host_cap);
	if (!done) {
		MSX_HIP(ctx, hipMemcpyAsync(host_out, u->fetch_src, tb, hipMemcpyDeviceToHost, ctx->stream));
		MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		return MSX_OK;
	}
	// (the kernels that wrote u->out have been waited for by msx_unpack_emit_gather)
	MSX_HIP(ctx, hipMemcpyAsync(host_out, u->fetch_src, tb, hipMemcpyDeviceToHost, u->copy_stream));
	if (u->fetch_par >= 0) {                  // the overlapped path: the encoder's next use of this parity's buffer waits for this
		MSX_HIP(ctx, hipEventRecord(u->ev_fetched[u->fetch_par], u->copy_stream));
		u->fetched_used[u->fetch_par] = true;
	} else {
		MSX_HIP(ctx, hipEventRecord(u->out_copied, u->copy_stream));
		u->out_busy = true;
	}
	MSX_HIP(ctx, hipEventRecord(done->ev, u->copy_stream));
	done->recorded = true;
	return MSX_OK;
}

extern "C" int msx_unpack_emit(msx_ctx *ctx, msx_unpack *u, const int32_t *emit_idx_dev, int64_t n_emit, uint8_t *host_out,
                               size_t host_cap, int64_t *n_bytes) {
	if (n_emit > 0 && !host_out) return MSX_ERR_ARG;
	int rc = msx_unpack_emit_gather(ctx, u, emit_idx_dev, n_emit, n_bytes);
	if (rc) return rc;
	return msx_unpack_emit_fetch(ctx, u, host_out, host_cap, nullptr);
}

// for tests: the record offsets of the batch (u32 [n_records + 1], relative to the batch's first byte)
extern "C" int msx_unpack_offsets(msx_ctx *ctx, msx_unpack *u, uint32_t *host, int64_t n) {
	if (!ctx || !u || !host) return MSX_ERR_ARG;
	msx_join(ctx);
	if (n > u->n_total + 1) n = u->n_total + 1;
	if (n > 0) {
		MSX_HIP(ctx, hipMemcpyAsync(host, u->rec_off.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
		MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	}
	return MSX_OK;
}

// msx_runtime_warmup: this translation unit's code object loaded onto the device ahead of its first launch (the runtime loads a
// module when one of its kernels is first asked for: 2-10 ms each, otherwise paid by the first batches of a command)
void msx_touch_unpack(void) {
	hipFuncAttributes attr;
	(void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_chase_walk));
}
