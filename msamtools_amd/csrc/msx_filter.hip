// msx_filter.hip -- device side of `msamtools filter`:
//   k_aln_stats_filter  per-record CIGAR/MD walk, -l/-p/-z predicates, --rescore,
//                       pool membership          (mBamVector.c:23-133, msam_filter.c:31-63,132-183)
//   k_besthit_select    per-pool best-hit / unique-best-hit selection, mate aware
//                                                 (msam_filter.c:192-263)
//   k_emit_*            the order in which the reference calls mSamWrite
//                                                 (msam_filter.c:235-244, mBamVector.c:343-348)
// All integer work, HBM-bound; no MFMA.  One wave64 lane per record (stats) or
// per pool (selection); variable-length CIGAR/MD payloads of a 256-record tile
// are staged into LDS with coalesced dword loads.
#include "msx_internal.h"
#include "msx_count.h"
#include "msx_md.h"
#include "msx_stats.h"

#include <climits>
#include <cstdlib>

// LDS staging capacities per 256-record tile.  Typical tile: ~300 CIGAR words,
// ~1.5 KB of MD.  Payload beyond the capacity is read straight from global.
#define CAP_CIG 1024          // dwords
#define CAP_MDW 1536          // dwords (6 KB)

// Software-pipelined over the tiles of a workgroup (tile k of block b is tile
// b + k*gridDim.x): while tile i is being walked out of LDS, the payload of tile
// i+1 (CIGAR words, MD dwords, FLAG, aux bits) and the offsets of tile i+2 are
// already in flight into registers; they are written to LDS after the barrier
// that ends tile i.  No global-memory latency sits between two tiles.
#define CIG_REGS (CAP_CIG / MSX_BLOCK)   // 4
#define MD_REGS (CAP_MDW / MSX_BLOCK)    // 6

struct TileOff {          // offsets of one tile, one entry per thread (+1 extra held by thread 0)
	uint32_t co, mo, co_last, mo_last;
};

__device__ __forceinline__ void load_offsets(const FilterArgs &A, int64_t tile, int tid, TileOff &o) {
	const int64_t t0 = tile * MSX_BLOCK;
	const int64_t left = A.n - t0;
	const int nt = left < MSX_BLOCK ? (int)left : MSX_BLOCK;
	o.co = o.mo = o.co_last = o.mo_last = 0;
	if (tid <= nt && t0 + tid <= A.n) {
		o.co = A.cigar_off[t0 + tid];
		o.mo = A.md_off[t0 + tid];
	}
	if (tid == 0 && nt == MSX_BLOCK) {
		o.co_last = A.cigar_off[t0 + MSX_BLOCK];
		o.mo_last = A.md_off[t0 + MSX_BLOCK];
	}
}

// (waves per SIMD pinned: the kernel is instruction-bound and loses 10 % when a few more registers drop it to 5)
__global__ __launch_bounds__(MSX_BLOCK) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_aln_stats_filter(FilterArgs A) {
	__shared__ uint32_t s_coff[2][MSX_BLOCK + 1];
	__shared__ uint32_t s_moff[2][MSX_BLOCK + 1];
	__shared__ uint32_t s_cig[CAP_CIG];
	__shared__ uint32_t s_md[CAP_MDW + 2];    // +2: the realigned walk reads one word past a string's last

	const int tid = threadIdx.x;
	const int64_t n_tiles = (A.n + MSX_BLOCK - 1) / MSX_BLOCK;
	const int64_t first = blockIdx.x, step = gridDim.x;
	if (first >= n_tiles) return;
	const uint32_t *md4 = reinterpret_cast<const uint32_t *>(A.md);

	uint32_t cr[CIG_REGS], mr[MD_REGS];
	uint32_t fl_n = 0, rf_n = 0;
	TileOff on;

	// helpers as lambdas over the LDS arrays
	auto store_offsets = [&](int buf, int64_t tile, const TileOff &o) {
		const int64_t left = A.n - tile * MSX_BLOCK;
		const int nt = left < MSX_BLOCK ? (int)left : MSX_BLOCK;
		if (tid <= nt) { s_coff[buf][tid] = o.co; s_moff[buf][tid] = o.mo; }
		if (tid == 0 && nt == MSX_BLOCK) { s_coff[buf][MSX_BLOCK] = o.co_last; s_moff[buf][MSX_BLOCK] = o.mo_last; }
	};
	auto issue_payload = [&](int buf, int64_t tile) {
		const int64_t t0 = tile * MSX_BLOCK;
		const int64_t left = A.n - t0;
		const int nt = left < MSX_BLOCK ? (int)left : MSX_BLOCK;
		// tile geometry is workgroup-uniform: keep it in scalar registers so that whole 256-word
		// slices beyond the tile's payload are skipped by scalar branches, not predicated off
		const uint32_t c0 = __builtin_amdgcn_readfirstlane(s_coff[buf][0]);
		uint32_t clen = __builtin_amdgcn_readfirstlane(s_coff[buf][nt]) - c0;
		if (clen > CAP_CIG) clen = CAP_CIG;
		const uint32_t *cbase = A.cigar + c0;
#pragma unroll
		for (int q = 0; q < CIG_REGS; q++) {
			cr[q] = 0u;
			if ((uint32_t)(q * MSX_BLOCK) < clen) {
				const uint32_t w = tid + q * MSX_BLOCK;
				if (w < clen) cr[q] = cbase[w];
			}
		}
		if (A.md_aligned) {
			const uint32_t m0a = __builtin_amdgcn_readfirstlane(s_moff[buf][0]) & ~3u;
			uint32_t mw = (__builtin_amdgcn_readfirstlane(s_moff[buf][nt]) - m0a + 3u) >> 2;
			if (mw > CAP_MDW) mw = CAP_MDW;
			const uint32_t *mbase = md4 + (m0a >> 2);
#pragma unroll
			for (int q = 0; q < MD_REGS; q++) {
				mr[q] = 0u;
				if ((uint32_t)(q * MSX_BLOCK) < mw) {
					const uint32_t w = tid + q * MSX_BLOCK;
					if (w < mw) mr[q] = mbase[w];
				}
			}
		}
		if (tid < nt) { fl_n = A.flag[t0 + tid]; rf_n = A.rflags[t0 + tid]; }
	};
	auto store_payload = [&]() {
#pragma unroll
		for (int q = 0; q < CIG_REGS; q++) s_cig[tid + q * MSX_BLOCK] = cr[q];
#pragma unroll
		for (int q = 0; q < MD_REGS; q++) s_md[tid + q * MSX_BLOCK] = mr[q];
	};

	// prologue: offsets(0) -> LDS, payload(0) and offsets(1) -> registers -> LDS
	load_offsets(A, first, tid, on);
	store_offsets(0, first, on);
	__syncthreads();
	issue_payload(0, first);
	if (first + step < n_tiles) load_offsets(A, first + step, tid, on);
	store_payload();
	if (first + step < n_tiles) store_offsets(1, first + step, on);
	__syncthreads();

	int buf = 0;
	for (int64_t tile = first; tile < n_tiles; tile += step, buf ^= 1) {
		const int64_t t0 = tile * MSX_BLOCK;
		const int nt = (int)((A.n - t0 < MSX_BLOCK) ? (A.n - t0) : MSX_BLOCK);
		const int64_t t = t0 + tid;
		const bool live = tid < nt;
		const uint32_t flag = fl_n, rf = rf_n;          // this tile's, loaded one iteration ago
		const bool has_next = tile + step < n_tiles, has_next2 = tile + 2 * step < n_tiles;

		// geometry of the staged payload of this tile
		const uint32_t c0 = __builtin_amdgcn_readfirstlane(s_coff[buf][0]);
		const uint32_t c_end = __builtin_amdgcn_readfirstlane(s_coff[buf][nt]);
		uint32_t clen = c_end - c0;
		if (clen > CAP_CIG) clen = CAP_CIG;
		uint32_t m0a = 0, mbytes = 0, m_end = 0;
		if (A.md_aligned) {
			m0a = __builtin_amdgcn_readfirstlane(s_moff[buf][0]) & ~3u;
			m_end = __builtin_amdgcn_readfirstlane(s_moff[buf][nt]);
			uint32_t mw = (m_end - m0a + 3u) >> 2;
			if (mw > CAP_MDW) mw = CAP_MDW;
			mbytes = mw << 2;
		}

		// tile-uniform: every CIGAR word and MD byte of this tile was staged
		const bool all_staged = A.md_aligned && (c_end - c0) <= CAP_CIG && (m_end - m0a) <= mbytes;

		// in flight while this tile is computed
		if (has_next) issue_payload(buf ^ 1, tile + step);
		if (has_next2) load_offsets(A, tile + 2 * step, tid, on);

		if (live) {
			uint32_t pooled = 0;
			if (A.as_out) A.as_out[t] = A.as[t];   // replaced below when the record is rescored
			if ((flag & MSX_F_UNMAP) && !A.o_len) {
				// msam_filter.c:132-138 (msx_aln_stats reports statistics for every record)
				pooled = (A.choice != 0 && A.keep_unmapped && A.ppt >= 0 && A.invert == 1) ? 1u : 0u;
			} else {
				uint32_t alen = 0, qlen = 0, qclip = 0, edit = 0;   // wrap like int32
				bool bad = false;
				const uint32_t cs = s_coff[buf][tid], ce = s_coff[buf][tid + 1];
				if (all_staged) {
					// fast path: the whole tile's payload is in LDS.  CIGAR by per-op bit tables
					// (bit i = op i contributes): MD path mBamVector.c:60-97, NM path :23-38.
					const bool mdp = (rf & MSX_HAS_MD) != 0;
					bad = !mdp && !(rf & MSX_HAS_NM);
					const uint32_t T_ALEN = mdp ? 0x187u : 0xff87u;    // M I D = X (NM path: all but N P S H)
					const uint32_t T_EDIT = mdp ? 0x006u : 0u;         // I D
					for (uint32_t k = cs; k < ce; ++k) {
						const uint32_t c = s_cig[k - c0];
						const uint32_t op = c & 0xf, w = c >> 4;
						alen += w & (uint32_t)__builtin_amdgcn_sbfe((int)T_ALEN, op, 1u);    // 0 or ~0: bit `op` of the table
						qlen += w & (uint32_t)__builtin_amdgcn_sbfe(0x1b3, op, 1u);          // M I S H = X
						edit += w & (uint32_t)__builtin_amdgcn_sbfe((int)T_EDIT, op, 1u);
						qclip += w & (uint32_t)__builtin_amdgcn_sbfe(0x030, op, 1u);         // S H
					}
					if (mdp) {
						// words realigned to the start of the string: full words, then one masked tail
						MdBits s = {0u, 0u, 0u, 0u};
						const uint32_t ms = s_moff[buf][tid], bs = ms - m0a, len = s_moff[buf][tid + 1] - ms;
						const uint32_t sh = bs & 3u, nfull = len >> 2, rem = len & 3u;
						uint32_t wi = bs >> 2;
						uint32_t lo = s_md[wi];
						for (uint32_t q = 0; q < nfull; ++q) {
							const uint32_t hi = s_md[++wi];
							md_word_aligned(s, MSX_ALIGNBYTE(hi, lo, sh), 0x80808080u);
							lo = hi;
						}
						if (rem) {
							const uint32_t hi = s_md[wi + 1];
							md_word_aligned(s, MSX_ALIGNBYTE(hi, lo, sh), 0x80808080u >> (8u * (4u - rem)));
						}
						edit += s.edit;
					} else if (!bad) {
						edit = (uint32_t)A.nm[t];                      // msam_filter.c:155
					} else {
						atomicMin(&A.st->first_no_mdnm, (unsigned long long)t);   // msam_filter.c:150-152
					}
				} else if (rf & MSX_HAS_MD) {
					// bam_get_summary, mBamVector.c:60-97
					for (uint32_t k = cs; k < ce; ++k) {
						uint32_t c = (k - c0 < clen) ? s_cig[k - c0] : A.cigar[k];
						uint32_t op = c & 0xf, w = c >> 4;
						if (op == MSX_OP_MATCH || op == MSX_OP_EQUAL || op == MSX_OP_DIFF) {
							qlen += w; alen += w;
						} else if (op == MSX_OP_INS) {
							qlen += w; edit += w; alen += w;
						} else if (op == MSX_OP_DEL) {
							edit += w; alen += w;
						} else if (op == MSX_OP_SOFT_CLIP || op == MSX_OP_HARD_CLIP) {
							qclip += w; qlen += w;
						}
					}
					// MD walk, mBamVector.c:101-118
					MdState s = {0u, 0u, 0u, 0};
					const uint32_t ms = s_moff[buf][tid], me = s_moff[buf][tid + 1];
					if (ms < me) {
						if (A.md_aligned && (me - m0a) <= mbytes) {
							const uint32_t bs = ms - m0a, be = me - m0a;
							for (uint32_t w = bs >> 2; (w << 2) < be; ++w) {
								const uint32_t word = s_md[w];
								const uint32_t p = w << 2;
								md_byte(s, word & 0xffu, p >= bs && p < be);
								md_byte(s, (word >> 8) & 0xffu, p + 1 >= bs && p + 1 < be);
								md_byte(s, (word >> 16) & 0xffu, p + 2 >= bs && p + 2 < be);
								md_byte(s, word >> 24, p + 3 >= bs && p + 3 < be);
							}
						} else {
							for (uint32_t j = ms; j < me; ++j) md_byte(s, A.md[j], true);
						}
					}
					edit += (uint32_t)s.edit;
				} else if (rf & MSX_HAS_NM) {
					// bam_cigar2details, mBamVector.c:23-38
					for (uint32_t k = cs; k < ce; ++k) {
						uint32_t c = (k - c0 < clen) ? s_cig[k - c0] : A.cigar[k];
						uint32_t op = c & 0xf, w = c >> 4;
						if (op == MSX_OP_HARD_CLIP || op == MSX_OP_SOFT_CLIP) {
							qclip += w; qlen += w;
						} else if (!(op == MSX_OP_REF_SKIP || op == MSX_OP_PAD)) {
							alen += w;
							if (op == MSX_OP_MATCH || op == MSX_OP_EQUAL || op == MSX_OP_DIFF || op == MSX_OP_INS)
								qlen += w;
						}
					}
					edit = (uint32_t)A.nm[t];   // msam_filter.c:155
				} else {
					bad = true;                 // msam_filter.c:150-152
					atomicMin(&A.st->first_no_mdnm, (unsigned long long)t);
				}
				if (A.o_status) A.o_status[t] = bad ? 1 : 0;
				if (A.o_len) {
					A.o_len[t] = (int32_t)alen; A.o_qlen[t] = (int32_t)qlen;
					A.o_qclip[t] = (int32_t)qclip; A.o_edit[t] = (int32_t)edit;
				}
				if (!bad) {
					if (A.rescore)          // msam_filter.c:160-168: hit=+1, miss=-1
						A.as_out[t] = (int32_t)((alen - edit) - edit);
					// msam_filter.c:31-35 in wrapping int32 arithmetic
					const int32_t L = (int32_t)alen;
					bool fl = L < A.min_length;
					bool fz, fp;
					// 32-bit integer multiplies run at quarter rate; when every operand of this wave fits 24 bits
					// (always, for real reads) v_mul_u32_u24 gives the same low 32 bits at full rate
					const uint32_t ident = alen - edit;
					const bool small = A.ppt >= 0 && (uint32_t)A.ppt < (1u << 24) && (uint32_t)A.max_clip < (1u << 24) &&
					                   __ballot(((alen | qlen | qclip | ident) >> 24) != 0u) == 0ull;
					if (small) {
						fz = (int32_t)__umul24(100u, qclip) >
						     (int32_t)__umul24((uint32_t)A.max_clip, qlen);
						fp = (int32_t)__umul24(1000u, ident) <
						     (int32_t)__umul24(alen, (uint32_t)A.ppt);
					} else {
						fz = (int32_t)(100u * qclip) > (int32_t)((uint32_t)A.max_clip * qlen);
						fp = (A.ppt < 0)
						         ? ((int32_t)(1000u * (edit - alen)) < (int32_t)(alen * (uint32_t)A.ppt))
						         : ((int32_t)(1000u * (alen - edit)) < (int32_t)(alen * (uint32_t)A.ppt));
					}
					bool fails = ((A.choice & 1) && fl) || ((A.choice & 2) && fp) || ((A.choice & 4) && fz);
					pooled = (A.choice == 0 || (int)fails == A.invert) ? 1u : 0u;   // msam_filter.c:181
				}
			}
			if (A.pool) {
				if (A.pool_as_code && pooled) {
					// msam_filter.c:223: AS is read from the record; after --rescore every mapped record has one (:167)
					const bool has = (rf & MSX_HAS_AS) || (A.rescore && !(flag & MSX_F_UNMAP));
					pooled = MSX_PC_IN | (has ? MSX_PC_HAS_AS : 0u) | (flag & MSX_F_MATES);
				}
				A.pool[t] = (uint8_t)pooled;
			}
		}
		__syncthreads();                 // every lane is done with this tile's LDS image
		if (has_next) store_payload();
		if (has_next2) store_offsets(buf, tile + 2 * step, on);
		__syncthreads();
	}
}

// ---------------------------------------------------------------------------
// best-hit selection: one lane per pool.
// ---------------------------------------------------------------------------
struct SelectArgs {
	int64_t n, n_groups;
	const uint32_t *group_off;
	const uint16_t *flag;
	const uint8_t *rflags;
	const uint8_t *pool;      // null: pooled = mapped (plain --besthit, msam_filter.c:104)
	int32_t pool_is_code;     // pool bytes are MSX_PC_* codes: FLAG and the aux bits need not be fetched
	const int32_t *as;        // AS to compare (as_out after --rescore)
	int32_t rescored;         // every mapped pooled record has AS (msam_filter.c:167)
	int32_t unique_only;      // --uniqhit
	uint8_t *keep;            // [n]
	uint32_t *gcount;         // [n_groups] records written per pool
	msx_dev_status *st;
};

// One lane per pool.  Pools are short (1 + Poisson(4) hits per read), so a lane
// fetches its pool's FLAGs, pool bytes, aux bits and AS values (<= 12 records) with
// ~27 independent loads issued back to back -- one memory round trip per pool
// instead of one per record -- realigns the sub-dword windows with alignbit /
// alignbyte so that record r sits at a compile-time position, and runs both
// passes (max + tie count per mate class; keep codes) out of registers.  Longer
// pools take the per-record loop.
#define BH_WIN 12

struct BhAcc {
	int32_t b0, b1, b2;          // best AS per mate class: neither bit, READ1, READ2
	uint32_t n0, n1, n2;         // ties
	uint32_t paired;
	uint32_t noas0, noas1, noas2;   // first participating record without AS
};

__device__ __forceinline__ void bh_count(BhAcc &c, uint32_t i, uint32_t fl, bool pooled, bool has, int32_t sc) {
	if (!pooled) return;
	const uint32_t cls = fl & MSX_F_MATES;
	c.paired |= cls;                                                 // mBamPoolIsPaired :196-204
	if (cls == 0) {
		if (!has) { if (c.noas0 == 0xffffffffu) c.noas0 = i; }
		else if (sc > c.b0) { c.b0 = sc; c.n0 = 1; } else if (sc == c.b0) c.n0++;
	} else if (cls == 0x40u) {
		if (!has) { if (c.noas1 == 0xffffffffu) c.noas1 = i; }
		else if (sc > c.b1) { c.b1 = sc; c.n1 = 1; } else if (sc == c.b1) c.n1++;
	} else if (cls == 0x80u) {
		if (!has) { if (c.noas2 == 0xffffffffu) c.noas2 = i; }
		else if (sc > c.b2) { c.b2 = sc; c.n2 = 1; } else if (sc == c.b2) c.n2++;
	}
}

__device__ __forceinline__ uint8_t bh_keep(const BhAcc &c, bool w0, bool w1, bool w2, uint32_t fl, bool pooled,
                                           bool has, int32_t sc) {
	if (!pooled || !has) return 0;
	const uint32_t cls = fl & MSX_F_MATES;
	if (cls == 0 && w0 && sc == c.b0) return 1;
	if (cls == 0x40u && w1 && sc == c.b1) return 1;
	if (cls == 0x80u && w2 && sc == c.b2) return 2;
	return 0;
}

// COUNT: the fused `filter | profile` form.  The pool's winners are in registers (k1/k2
// masks) when the selection is done, so the insert accounting of msx_count.h runs right
// here instead of in a second kernel that would fetch group_off and the keep codes again.
template <bool COUNT>
__global__ __launch_bounds__(MSX_BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_besthit_select(SelectArgs A, CountArgs P) {
	__shared__ uint32_t s_c[3][MSX_BLOCK / 64];
	__shared__ int32_t s_key[COUNT ? UI_TBL : 1];
	__shared__ uint32_t s_val[COUNT ? UI_TBL : 1];
	BlockCounts bc = {0u, 0u, 0u};
	if (COUNT) count_block_begin(P, s_key, s_val);
	const int64_t stride = (int64_t)gridDim.x * MSX_BLOCK;
	int64_t g = (int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x;
	// the next pool's bounds are fetched while this one is processed (one round trip less per pool)
	uint32_t s_nx = 0, e_nx = 0;
	if (g < A.n_groups) { s_nx = A.group_off[g]; e_nx = A.group_off[g + 1]; }
	for (; g < A.n_groups; g += stride) {
		const uint32_t s = s_nx, e = e_nx;
		if (g + stride < A.n_groups) { s_nx = A.group_off[g + stride]; e_nx = A.group_off[g + stride + 1]; }
		const uint32_t len = e - s;
		BhAcc c = {INT_MIN, INT_MIN, INT_MIN, 0u, 0u, 0u, 0u, 0xffffffffu, 0xffffffffu, 0xffffffffu};
		uint32_t cnt = 0;
		if (len <= BH_WIN && (uint64_t)(s & ~3u) + 16u <= (uint64_t)A.n && (uint64_t)s + BH_WIN <= (uint64_t)A.n) {
			// ---- bulk loads (independent of each other) ----
			const bool coded = A.pool_is_code != 0;      // (kernel argument: a scalar branch)
			const uint32_t *f32p = reinterpret_cast<const uint32_t *>(A.flag) + (s >> 1);
			uint32_t fw[7] = {0u, 0u, 0u, 0u, 0u, 0u, 0u};
			const uint32_t *r32p = reinterpret_cast<const uint32_t *>(A.rflags) + (s >> 2);
			uint32_t rw[4] = {0u, 0u, 0u, 0u}, pw[4] = {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u};
			if (!coded) {
#pragma unroll
				for (int q = 0; q < 7; q++)        // only the dwords the pool reaches into
					fw[q] = (2u * ((s >> 1) + (uint32_t)q) < e) ? f32p[q] : 0u;
#pragma unroll
				for (int q = 0; q < 4; q++) rw[q] = (4u * ((s >> 2) + (uint32_t)q) < e) ? r32p[q] : 0u;
			}
			if (A.pool) {
				const uint32_t *p32p = reinterpret_cast<const uint32_t *>(A.pool) + (s >> 2);
#pragma unroll
				for (int q = 0; q < 4; q++) pw[q] = (4u * ((s >> 2) + (uint32_t)q) < e) ? p32p[q] : 0u;
			}
			int32_t sc[BH_WIN];
#pragma unroll
			for (int r = 0; r < BH_WIN; r++) sc[r] = ((uint32_t)r < len) ? A.as[s + r] : INT_MIN;
			// ---- realign so that record r is at a fixed position ----
			const uint32_t fsh = 16u * (s & 1u), bsh = s & 3u;
			uint32_t fa[6], ra[3], pa[3];
#pragma unroll
			for (int q = 0; q < 6; q++) fa[q] = __builtin_amdgcn_alignbit(fw[q + 1], fw[q], fsh);
#pragma unroll
			for (int q = 0; q < 3; q++) {
				ra[q] = __builtin_amdgcn_alignbyte(rw[q + 1], rw[q], bsh);
				pa[q] = __builtin_amdgcn_alignbyte(pw[q + 1], pw[q], bsh);
			}
			// ---- who takes part: code = mate bits | 0x100 (participates) | 0x200 (has AS) ----
			uint32_t code[BH_WIN];
			uint32_t paired = 0;
#pragma unroll
			for (int r = 0; r < BH_WIN; r++) {
				const uint32_t pb = (pa[r >> 2] >> (8 * (r & 3))) & 0xffu;
				if (coded) {
					// the stats kernel has already decided participation and looked at FLAG and the aux bits
					code[r] = ((uint32_t)r < len && (pb & MSX_PC_IN))
					              ? ((pb & MSX_F_MATES) | 0x100u | ((pb & MSX_PC_HAS_AS) ? 0x200u : 0u)) : 0u;
				} else {
					const uint32_t fl = (fa[r >> 1] >> (16 * (r & 1))) & 0xffffu;
					const uint32_t rf = (ra[r >> 2] >> (8 * (r & 3))) & 0xffu;
					const bool part = ((uint32_t)r < len) && (A.pool ? (pb != 0) : !(fl & MSX_F_UNMAP));
					const bool has = (rf & MSX_HAS_AS) || (A.rescored && !(fl & MSX_F_UNMAP));
					code[r] = part ? ((fl & MSX_F_MATES) | 0x100u | (has ? 0x200u : 0u)) : 0u;
				}
				paired |= code[r] & MSX_F_MATES;                                 // mBamPoolIsPaired :196-204
			}
			// a paired pool is judged per mate (READ1 -> A, READ2 -> B), an unpaired one as a whole (A)
			const uint32_t selA = paired ? 0x140u : 0x100u, selB = paired ? 0x180u : 0xffffu;
			int32_t bA = INT_MIN, bB = INT_MIN;
			uint32_t hasA = 0, hasB = 0, noas = 0;                                // bit r = record r
#pragma unroll
			for (int r = 0; r < BH_WIN; r++) {
				const uint32_t cl = code[r] & 0x1c0u;
				const bool inA = (cl == selA), inB = (cl == selB), has = (code[r] & 0x200u) != 0;
				hasA |= (inA && has) ? (1u << r) : 0u;
				hasB |= (inB && has) ? (1u << r) : 0u;
				noas |= ((inA || inB) && !has) ? (1u << r) : 0u;
				const int32_t sa = (inA && has) ? sc[r] : INT_MIN, sb = (inB && has) ? sc[r] : INT_MIN;
				bA = sa > bA ? sa : bA;
				bB = sb > bB ? sb : bB;
			}
			uint32_t eqA = 0, eqB = 0;
#pragma unroll
			for (int r = 0; r < BH_WIN; r++) {
				eqA |= (sc[r] == bA) ? (1u << r) : 0u;
				eqB |= (sc[r] == bB) ? (1u << r) : 0u;
			}
			eqA &= hasA;                                                          // the records holding the best score
			eqB &= hasB;
			const uint32_t nA = (uint32_t)__popc(eqA), nB = (uint32_t)__popc(eqB);
			const bool wA = nA > 0 && (!A.unique_only || nA == 1);                // :232-233
			const bool wB = nB > 0 && (!A.unique_only || nB == 1);
			// keep codes: 1 = written in the first pass (unpaired winners, READ1 winners), 2 = READ2 winners
			const uint32_t k1 = wA ? eqA : 0u, k2 = wB ? eqB : 0u;
#pragma unroll
			for (int r = 0; r < BH_WIN; r++)
				if ((uint32_t)r < len) A.keep[s + r] = (uint8_t)(((k1 >> r) & 1u) | (((k2 >> r) & 1u) << 1));
			cnt = (uint32_t)__popc(k1 | k2);
			if (COUNT) {
				PoolAcc v;
				pool_begin(P, v, s);
				pool_visit_masks(P, v, s, k1, k2);
				pool_finish(P, g, v, s_key, s_val, bc);
			}
			if (noas) {
				const uint32_t first = s + (uint32_t)__ffs((int)noas) - 1u;
				if (paired) c.noas1 = first; else c.noas0 = first;
			}
			c.paired = paired;
		} else {
			for (uint32_t i = s; i < e; ++i) {
				const uint32_t fl = A.flag[i];
				const bool pooled = A.pool ? (A.pool[i] != 0) : !(fl & MSX_F_UNMAP);
				const bool has = A.pool_is_code ? ((A.pool[i] & MSX_PC_HAS_AS) != 0)
				                                : ((A.rflags[i] & MSX_HAS_AS) || (A.rescored && !(fl & MSX_F_UNMAP)));
				bh_count(c, i, fl, pooled, has, A.as[i]);
			}
			const bool w0 = !c.paired && c.n0 > 0 && (!A.unique_only || c.n0 == 1);
			const bool w1 = c.paired && c.n1 > 0 && (!A.unique_only || c.n1 == 1);
			const bool w2 = c.paired && c.n2 > 0 && (!A.unique_only || c.n2 == 1);
			for (uint32_t i = s; i < e; ++i) {
				const uint32_t fl = A.flag[i];
				const bool pooled = A.pool ? (A.pool[i] != 0) : !(fl & MSX_F_UNMAP);
				const bool has = A.pool_is_code ? ((A.pool[i] & MSX_PC_HAS_AS) != 0)
				                                : ((A.rflags[i] & MSX_HAS_AS) || (A.rescored && !(fl & MSX_F_UNMAP)));
				const uint8_t k = bh_keep(c, w0, w1, w2, fl, pooled, has, A.as[i]);
				A.keep[i] = k;
				cnt += (k != 0);
			}
			if (COUNT) {
				// long pool: the keep codes this lane just wrote, in output order
				PoolAcc v;
				pool_begin(P, v, s);
				for (uint32_t pass = 1; pass <= 2; ++pass)
					for (uint32_t i = s; i < e; ++i)
						if (A.keep[i] == pass) pool_visit(P, v, P.tid[i]);
				pool_finish(P, g, v, s_key, s_val, bc);
			}
		}
		// msam_filter.c:219-221: a participating record without AS is fatal
		const uint32_t bad = c.paired ? (c.noas1 < c.noas2 ? c.noas1 : c.noas2) : c.noas0;
		if (bad != 0xffffffffu) atomicMin(&A.st->first_no_as, (unsigned long long)bad);
		A.gcount[g] = cnt;
	}
	if (COUNT) count_block_end(P, s_key, s_val, s_c, bc);
}

// emit order for pools: all pass-1 records of a pool, then its pass-2 records.
// The keep codes of a pool (<= 20 records) arrive as six independent aligned dword
// loads -- one memory round trip per pool instead of one per record.
__global__ __launch_bounds__(MSX_BLOCK) void k_emit_groups(int64_t n_records, int64_t n_groups,
                                                           const uint32_t *__restrict__ group_off,
                                                           const uint8_t *__restrict__ keep,
                                                           const uint32_t *__restrict__ gbase,
                                                           int32_t *__restrict__ emit_idx, msx_dev_status *st) {
	const int64_t stride = (int64_t)gridDim.x * MSX_BLOCK;
	for (int64_t g = (int64_t)blockIdx.x * MSX_BLOCK + threadIdx.x; g < n_groups; g += stride) {
		const uint32_t s = group_off[g], e = group_off[g + 1];
		uint32_t o = gbase[g];
		const uint32_t cnt = gbase[g + 1] - o;
		if (!cnt || !emit_idx) continue;
		if (e - s <= 20u && (uint64_t)(s & ~3u) + 24u <= (uint64_t)n_records) {
			const uint32_t base = s & ~3u, sh = s - base, len = e - s;
			const uint32_t *kw = reinterpret_cast<const uint32_t *>(keep) + (base >> 2);
			uint32_t w[6];
#pragma unroll
			for (int q = 0; q < 6; q++) w[q] = (base + 4u * (uint32_t)q < e) ? kw[q] : 0u;   // only the dwords the pool reaches into
			uint32_t m1 = 0, m2 = 0;
#pragma unroll
			for (int q = 0; q < 6; q++) {
#pragma unroll
				for (int bq = 0; bq < 4; bq++) {
					const uint32_t r = (uint32_t)(q * 4 + bq) - sh;
					const uint32_t kc = (w[q] >> (8 * bq)) & 0xffu;
					if ((uint32_t)(q * 4 + bq) >= sh && r < len) {
						m1 |= (kc == 1u ? 1u : 0u) << r;
						m2 |= (kc == 2u ? 1u : 0u) << r;
					}
				}
			}
			while (m1) { const uint32_t b = (uint32_t)__ffs((int)m1) - 1u; m1 &= m1 - 1u; emit_idx[o++] = (int32_t)(s + b); }
			while (m2) { const uint32_t b = (uint32_t)__ffs((int)m2) - 1u; m2 &= m2 - 1u; emit_idx[o++] = (int32_t)(s + b); }
		} else {
			uint32_t n2 = 0;
			for (uint32_t i = s; i < e; ++i) {
				const uint8_t k = keep[i];
				if (k == 1) emit_idx[o++] = (int32_t)i;
				n2 += (k == 2);
			}
			if (n2)
				for (uint32_t i = s; i < e; ++i)
					if (keep[i] == 2) emit_idx[o++] = (int32_t)i;
		}
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) st->n_emit = gbase[n_groups];
}

// plain filter (no best-hit): output = pooled records in input order.
// count per 2048-record chunk -> scan -> ordered fill.
#define EMIT_CHUNK 2048
__global__ __launch_bounds__(MSX_BLOCK) void k_emit_count(int64_t n, const uint8_t *__restrict__ keep,
                                                          uint32_t *__restrict__ ccount) {
	__shared__ uint32_t s_w[4];
	const int64_t base = (int64_t)blockIdx.x * EMIT_CHUNK;
	uint32_t c = 0;
	for (int k = 0; k < EMIT_CHUNK / MSX_BLOCK; k++) {
		int64_t i = base + k * MSX_BLOCK + threadIdx.x;
		if (i < n) c += keep[i] != 0;
	}
	for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
	if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
	__syncthreads();
	if (threadIdx.x == 0) ccount[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ __launch_bounds__(MSX_BLOCK) void k_emit_fill(int64_t n, int64_t n_chunks, const uint8_t *__restrict__ keep,
                                                         const uint32_t *__restrict__ cbase,
                                                         int32_t *__restrict__ emit_idx, msx_dev_status *st) {
	__shared__ uint32_t s_w[4];
	const int64_t base = (int64_t)blockIdx.x * EMIT_CHUNK;
	uint32_t run = cbase[blockIdx.x];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	for (int k = 0; k < EMIT_CHUNK / MSX_BLOCK; k++) {
		int64_t i = base + k * MSX_BLOCK + threadIdx.x;
		const bool kp = (i < n) && keep[i] != 0;
		const unsigned long long bal = __ballot(kp);
		if (lane == 0) s_w[w] = (uint32_t)__popcll(bal);
		__syncthreads();
		uint32_t before = 0, tot = 0;
		for (int q = 0; q < 4; q++) {
			if (q < w) before += s_w[q];
			tot += s_w[q];
		}
		if (kp && emit_idx)
			emit_idx[run + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = (int32_t)i;
		run += tot;
		__syncthreads();
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) st->n_emit = cbase[n_chunks];
}

__global__ void k_status_init(msx_dev_status *st) {
	st->first_no_mdnm = ~0ull;
	st->first_no_as = ~0ull;
	st->n_emit = 0;
}

// ---------------------------------------------------------------------------
// host entry points
// ---------------------------------------------------------------------------
static int filter_choice(const msx_filter_params *p) {
	int c = 0;
	if (p->min_length > 0) c |= 1;   // msam_filter.c:79-81
	if (p->ppt != 0) c |= 2;
	if (p->max_clip < 100) c |= 4;
	return c;
}

// The statistics kernel: k_aln_stats_flat (msx_stats.hip); MSX_STATS_V1=1 keeps the lane-per-record
// LDS-staged kernel of round 1 above for comparison runs.
static void launch_stats(msx_ctx *ctx, FilterArgs &A) {
	static const bool v1 = [] { const char *e = getenv("MSX_STATS_V1"); return e && atoi(e) != 0; }();
	if (v1) {
		hipLaunchKernelGGL(k_aln_stats_filter, dim3(msx_grid_x(ctx, A.n, MSX_BLOCK, 4)), dim3(MSX_BLOCK), 0,
		                   ctx->stream, A);
		return;
	}
	const uintptr_t al8 = (uintptr_t)A.cigar_off | (uintptr_t)A.md_off;
	A.wide_ok = ((al8 & 7u) == 0 && ((uintptr_t)A.flag & 3u) == 0 && ((uintptr_t)A.rflags & 1u) == 0 &&
	             ((uintptr_t)A.pool & 1u) == 0) ? 1 : 0;
	msx_launch_aln_stats_flat(ctx, A, msx_grid_x(ctx, A.n, 128 * (MSX_BLOCK / 64), 4));
}

static void fill_args(FilterArgs &A, const msx_batch *b) {
	A.n = b->n_records;
	A.flag = b->flag;
	A.rflags = b->rflags;
	A.cigar_off = b->cigar_off;
	A.cigar = b->cigar;
	A.md_off = b->md_off;
	A.md = b->md;
	A.nm = b->nm;
	A.as = b->as;
	A.md_aligned = (((uintptr_t)b->md) & 3u) == 0 ? 1 : 0;
}

static int filter_enqueue_impl(msx_ctx *ctx, const msx_batch *b, const msx_filter_params *p, const msx_filter_out *out,
                               msx_profile *prof) {
	if (!ctx || !b || !p || !out || !out->keep) return msx_fail(ctx, MSX_ERR_ARG, "msx_filter_enqueue: null argument");
	if (prof && (!b->group_off || !b->tid))
		return msx_fail(ctx, MSX_ERR_ARG, "msx_filter_profile_enqueue needs tid and group_off");
	const int choice = filter_choice(p);
	const bool best = p->besthit || p->uniqhit;
	if (choice == 0 && !best)
		return msx_fail(ctx, MSX_ERR_NO_FILTER,
		                "'filter' command requires atleast one of --ppt, -l, -p, -z, --besthit or --uniqhit");
	if (best && !b->group_off)
		return msx_fail(ctx, MSX_ERR_ARG, "--besthit/--uniqhit need msx_batch.group_off (QNAME pools)");
	if (p->rescore && !out->as_out)
		return msx_fail(ctx, MSX_ERR_ARG, "--rescore needs msx_filter_out.as_out");
	if (b->n_records > 0x7fffffffLL) return msx_fail(ctx, MSX_ERR_ARG, "batch too large");
	if (!b->flag || !b->rflags) return msx_fail(ctx, MSX_ERR_ARG, "msx_filter_enqueue: flag/rflags missing");
	if ((choice != 0 || p->rescore) && (!b->cigar_off || !b->cigar || !b->md_off || !b->md || !b->nm))
		return msx_fail(ctx, MSX_ERR_ARG, "msx_filter_enqueue: -l/-p/-z/--rescore need cigar, md and nm arrays");
	if (best && !b->as) return msx_fail(ctx, MSX_ERR_ARG, "msx_filter_enqueue: --besthit/--uniqhit need the as array");
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	const int64_t n = b->n_records;
	hipLaunchKernelGGL(k_status_init, dim3(1), dim3(1), 0, ctx->stream, ctx->d_status);
	ctx->filter_pending = true;
	if (n == 0) return MSX_OK;

	const bool need_stats = choice != 0 || p->rescore;
	uint8_t *pool = nullptr;
	int rc;
	if (need_stats) {
		FilterArgs A = {};
		fill_args(A, b);
		A.min_length = p->min_length;
		A.ppt = p->ppt;
		A.max_clip = p->max_clip;
		A.choice = choice;
		A.rescore = p->rescore;
		A.invert = p->invert;
		A.keep_unmapped = p->keep_unmapped;
		A.as_out = p->rescore ? out->as_out : nullptr;
		A.st = ctx->d_status;
		if (best) {
			if ((rc = msx_reserve(ctx, &ctx->pool_code, (size_t)n))) return rc;
			pool = (uint8_t *)ctx->pool_code.p;
			A.pool_as_code = 1;
		} else {
			pool = out->keep;   // mWriteBamPool: keep == pooled
		}
		A.pool = pool;
		msx_time_begin(ctx, MSX_K_ALN_STATS);
		launch_stats(ctx, A);
		msx_time_end(ctx);
	}
	if (best) {
		const int64_t ng = b->n_groups;
		if ((rc = msx_reserve(ctx, &ctx->gcount, (size_t)(ng + 8) * 4))) return rc;
		if ((rc = msx_reserve(ctx, &ctx->gbase, (size_t)(ng + 8) * 4))) return rc;
		SelectArgs S = {};
		S.n = n;
		S.n_groups = ng;
		S.group_off = b->group_off;
		S.flag = b->flag;
		S.rflags = b->rflags;
		S.pool = pool;
		S.pool_is_code = (pool && need_stats) ? 1 : 0;
		S.as = p->rescore ? out->as_out : b->as;
		S.rescored = p->rescore;
		S.unique_only = p->uniqhit ? 1 : 0;   // msam_filter.c:88-91: --uniqhit wins
		S.keep = out->keep;
		S.gcount = (uint32_t *)ctx->gcount.p;
		S.st = ctx->d_status;
		CountArgs P = {};
		bool by_part = false;
		if (prof && ng > 0 && (rc = msx_profile_count_prepare(ctx, prof, b, out->keep, &P, &by_part))) return rc;
		msx_time_begin(ctx, MSX_K_BESTHIT);
		if (prof && ng > 0)
			hipLaunchKernelGGL(k_besthit_select<true>, dim3(msx_grid_x(ctx, ng, MSX_BLOCK, 4)), dim3(MSX_BLOCK), 0,
			                   ctx->stream, S, P);
		else
			hipLaunchKernelGGL(k_besthit_select<false>, dim3(msx_grid_x(ctx, ng, MSX_BLOCK, 4)), dim3(MSX_BLOCK), 0,
			                   ctx->stream, S, P);
		msx_time_end(ctx);
		// three independent chains from here: counting the unique-insert keys (side lane 0, inside
		// msx_profile_count_finish), appending the multi-mapper lists (main stream), and filter's own
		// output order (side lane 1); all three are latency-bound and overlap well
		const bool forked = prof && ng > 0 && msx_fork(ctx);
		if (prof && ng > 0 && (rc = msx_profile_count_finish(ctx, prof, b, by_part))) { msx_join(ctx); return rc; }
		if (forked) msx_lane_enter(ctx, 1);
		rc = msx_scan_u32(ctx, S.gcount, (uint32_t *)ctx->gbase.p, ng);
		if (!rc) {
			msx_time_begin(ctx, MSX_K_EMIT);
			hipLaunchKernelGGL(k_emit_groups, dim3(msx_grid(ctx, ng, MSX_BLOCK)), dim3(MSX_BLOCK), 0, ctx->stream,
			                   n, ng, b->group_off, (const uint8_t *)out->keep, (const uint32_t *)ctx->gbase.p,
			                   out->emit_idx, ctx->d_status);
			msx_time_end(ctx);
		}
		msx_join(ctx);
		if (rc) return rc;
	} else {
		const int64_t nc = (n + EMIT_CHUNK - 1) / EMIT_CHUNK;
		if ((rc = msx_reserve(ctx, &ctx->gcount, (size_t)(nc + 8) * 4))) return rc;
		if ((rc = msx_reserve(ctx, &ctx->gbase, (size_t)(nc + 8) * 4))) return rc;
		msx_time_begin(ctx, MSX_K_EMIT);
		hipLaunchKernelGGL(k_emit_count, dim3((unsigned)nc), dim3(MSX_BLOCK), 0, ctx->stream, n,
		                   (const uint8_t *)out->keep, (uint32_t *)ctx->gcount.p);
		msx_time_end(ctx);
		if ((rc = msx_scan_u32(ctx, (const uint32_t *)ctx->gcount.p, (uint32_t *)ctx->gbase.p, nc))) return rc;
		msx_time_begin(ctx, MSX_K_EMIT);
		hipLaunchKernelGGL(k_emit_fill, dim3((unsigned)nc), dim3(MSX_BLOCK), 0, ctx->stream, n, nc,
		                   (const uint8_t *)out->keep, (const uint32_t *)ctx->gbase.p, out->emit_idx,
		                   ctx->d_status);
		msx_time_end(ctx);
		// without best-hit selection there is no per-pool kernel to fuse into: plain sequence
		if (prof && (rc = msx_profile_accumulate(ctx, prof, b, out->keep))) return rc;
	}
	MSX_HIP(ctx, hipGetLastError());
	return MSX_OK;
}

extern "C" int msx_filter_enqueue(msx_ctx *ctx, const msx_batch *b, const msx_filter_params *p,
                                  const msx_filter_out *out) {
	return filter_enqueue_impl(ctx, b, p, out, nullptr);
}

extern "C" int msx_filter_profile_enqueue(msx_ctx *ctx, const msx_batch *b, const msx_filter_params *p,
                                          const msx_filter_out *out, msx_profile *prof) {
	if (!prof) return msx_fail(ctx, MSX_ERR_ARG, "msx_filter_profile_enqueue: null profile");
	return filter_enqueue_impl(ctx, b, p, out, prof);
}

extern "C" int msx_filter_finish(msx_ctx *ctx, msx_filter_status *status) {
	if (!ctx) return MSX_ERR_ARG;
	if (status) { status->n_emit = 0; status->err_record = -1; }
	if (!ctx->filter_pending) return msx_fail(ctx, MSX_ERR_ARG, "msx_filter_finish without msx_filter_enqueue");
	ctx->filter_pending = false;
	MSX_HIP(ctx, hipMemcpyAsync(ctx->h_status, ctx->d_status, sizeof(msx_dev_status), hipMemcpyDeviceToHost,
	                            ctx->stream));
	MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	const msx_dev_status s = *ctx->h_status;
	if (status) status->n_emit = (int64_t)s.n_emit;
	// the reference dies at the first offending record in stream order; stats
	// errors are raised while reading (before the pool is written)
	if (s.first_no_mdnm != ~0ull && (s.first_no_as == ~0ull || s.first_no_mdnm <= s.first_no_as)) {
		if (status) status->err_record = (int64_t)s.first_no_mdnm;
		return msx_fail(ctx, MSX_ERR_NO_MD_NM,
		                "Either NM or MD must be present in SAM/BAM input for 'filter' command. "
		                "Type 'msamtools filter -h' for details.");
	}
	if (s.first_no_as != ~0ull) {
		if (status) status->err_record = (int64_t)s.first_no_as;
		return msx_fail(ctx, MSX_ERR_NO_AS,
		                "Required field AS not found in SAM/BAM input. Type 'msamtools -h' for details.");
	}
	return MSX_OK;
}

extern "C" int msx_aln_stats(msx_ctx *ctx, const msx_batch *b, int32_t *length, int32_t *qlen, int32_t *qclip,
                             int32_t *edit, uint8_t *status) {
	if (!ctx || !b) return MSX_ERR_ARG;
	if (!length || !qlen || !qclip || !edit)
		return msx_fail(ctx, MSX_ERR_ARG, "msx_aln_stats: length/query_length/query_clip/edit are all required");
	if (!b->flag || !b->rflags || !b->cigar_off || !b->cigar || !b->md_off || !b->md || !b->nm)
		return msx_fail(ctx, MSX_ERR_ARG, "msx_aln_stats: flag, rflags, cigar, md and nm arrays are required");
	MSX_HIP(ctx, hipSetDevice(ctx->device));
	hipLaunchKernelGGL(k_status_init, dim3(1), dim3(1), 0, ctx->stream, ctx->d_status);
	if (b->n_records == 0) return MSX_OK;
	FilterArgs A = {};
	fill_args(A, b);
	A.max_clip = 100;
	A.o_len = length;
	A.o_qlen = qlen;
	A.o_qclip = qclip;
	A.o_edit = edit;
	A.o_status = status;
	A.st = ctx->d_status;
	msx_time_begin(ctx, MSX_K_ALN_STATS);
	launch_stats(ctx, A);
	msx_time_end(ctx);
	MSX_HIP(ctx, hipGetLastError());
	MSX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return MSX_OK;
}
