// msx_stats.hip -- k_aln_stats_flat: per-record CIGAR/MD statistics, the -l/-p/-z
// predicates, --rescore and pool membership (mBamVector.c:23-133, msam_filter.c:31-63,
// 132-183), wave-autonomous, no workgroup barriers.
//
// One wave takes a tile of 128 consecutive records, two per lane.  The lane-per-record
// MD walk costs as many steps as the LONGEST string of the wave (a 23-byte MD of a
// read with seven mismatches makes all 64 lanes walk six words); here the tile's MD
// bytes -- contiguous in the md buffer -- are walked flat instead, 16 bytes per lane,
// work proportional to the bytes (msx_md.h, "Flat walk"):
//   1. every record marks the first byte of its string in an LDS bitmap (ds_or);
//   2. every lane classifies its 16 bytes and adds (members + carets/starts): the carry
//      of that addition is exactly the "do not count" state of the token rule; carries
//      between lanes come from two ballots and one 64-bit scalar addition;
//   3. per-word masks of counted bytes and their running count go to LDS;
//   4. a record's MD edit count is F(end) - F(start), F(b) = count[b >> 2] + popcount of
//      the word's counted bytes below b.
// Strings longer than the 1 KiB a pass covers simply take more passes (the carry and
// the partial differences continue), so any length stays exact.
// CIGAR words are gathered straight from global memory (consecutive records hold
// consecutive words: the gather is dense), first operation unconditionally, further
// ones while any record of the wave has more.
#include "msx_internal.h"
#include "msx_md.h"
#include "msx_stats.h"

#define SF_RECS 128                 // records per wave tile
#define SF_WORDS 256                // MD words per pass: 16 bytes per lane
#define SF_BYTES (4 * SF_WORDS)

struct __attribute__((aligned(16))) SfLds {
	uint32_t start[SF_WORDS + 4];   // 0x01 in every byte that begins a string
	uint32_t c80[SF_WORDS + 4];     // bit 7 of every counted byte; [SF_WORDS] = 0
	uint32_t pre[SF_WORDS + 4];     // counted bytes in the words before; [SF_WORDS] = the pass total
};

// LDS hand-over between the lanes of ONE wave: the hardware executes a wave's LDS operations in
// order; this only keeps the compiler from moving them across
__device__ __forceinline__ void wave_sync() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// inclusive sum over the 64 lanes (row_shr 1/2/4/8 inside rows of 16, then row_bcast 15 and 31)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
	return x;
}

// F(pb): counted bytes of the pass before byte position pb (0..SF_BYTES)
__device__ __forceinline__ uint32_t sf_count_before(const SfLds &L, uint32_t pb) {
	const uint32_t w = pb >> 2;
	const uint32_t below = (1u << (8u * (pb & 3u))) - 1u;
	return L.pre[w] + (uint32_t)__popc(L.c80[w] & below);
}

struct SfAcc {
	uint32_t alen, qlen, qclip, edit;     // wrap like int32
};

// one CIGAR operation by per-op bit tables (bit i = op i contributes).
// MD path mBamVector.c:60-97, NM path :23-38.
__device__ __forceinline__ void sf_cigar_op(SfAcc &a, uint32_t c, uint32_t t_alen, uint32_t t_edit) {
	const uint32_t op = c & 0xfu, w = c >> 4;
	a.alen += w & (uint32_t)__builtin_amdgcn_sbfe((int)t_alen, op, 1u);
	a.qlen += w & (uint32_t)__builtin_amdgcn_sbfe(0x1b3, op, 1u);            // M I S H = X
	a.edit += w & (uint32_t)__builtin_amdgcn_sbfe((int)t_edit, op, 1u);
	a.qclip += w & (uint32_t)__builtin_amdgcn_sbfe(0x030, op, 1u);           // S H
}

struct SfTile {              // what a lane holds of a tile before any dependent load: offsets and flags of its two records
	uint32_t co0, co1, co2;  // cigar_off[tA], [tA+1], [tA+2]
	uint32_t mo0, mo1, mo2;  // md_off likewise
	uint32_t fl;             // FLAG of record A | FLAG of record B << 16
	uint32_t rf;             // aux bits of A | of B << 8
};

// (record indices fit 31 bits: msx_filter_enqueue rejects larger batches; the tile number is wave-uniform)
__device__ __forceinline__ void sf_load_tile(const FilterArgs &A, uint32_t n, uint32_t tile, int lane, SfTile &o) {
	const uint32_t t0 = tile * SF_RECS, tA = t0 + 2u * (uint32_t)lane;
	if (t0 + SF_RECS <= n && A.wide_ok) {
		// whole tile: the three offsets as consecutive dwords, 4 + 2 bytes for the flags
		o.co0 = A.cigar_off[tA]; o.co1 = A.cigar_off[tA + 1]; o.co2 = A.cigar_off[tA + 2];
		o.mo0 = A.md_off[tA]; o.mo1 = A.md_off[tA + 1]; o.mo2 = A.md_off[tA + 2];
		o.fl = *reinterpret_cast<const uint32_t *>(A.flag + tA);
		o.rf = *reinterpret_cast<const uint16_t *>(A.rflags + tA);
	} else {
		// the last tile (or unaligned arrays): indices clamped to n, records beyond n are empty
		const uint32_t i0 = tA < n ? tA : n, i1 = tA + 1 < n ? tA + 1 : n, i2 = tA + 2 < n ? tA + 2 : n;
		o.co0 = A.cigar_off[i0]; o.co1 = A.cigar_off[i1]; o.co2 = A.cigar_off[i2];
		o.mo0 = A.md_off[i0]; o.mo1 = A.md_off[i1]; o.mo2 = A.md_off[i2];
		o.fl = (tA < n ? (uint32_t)A.flag[tA] : 0u) | (tA + 1 < n ? (uint32_t)A.flag[tA + 1] << 16 : 0u);
		o.rf = (tA < n ? (uint32_t)A.rflags[tA] : 0u) | (tA + 1 < n ? (uint32_t)A.rflags[tA + 1] << 8 : 0u);
	}
}

// One pass of the flat MD walk: bytes [lo, lo + SF_BYTES) of the tile's span (positions relative to the
// 16-byte aligned `base`).  p0/p1/p2 = the lane's three string boundaries in that space.  MULTI: the tile
// needs more than one pass, so boundaries are clamped to this pass.
template <bool MULTI>
__device__ __forceinline__ void sf_md_pass(SfLds &L, int lane, const uint8_t *base, uint32_t span, uint32_t lo,
                                           uint32_t p0, uint32_t p1, uint32_t p2, bool realA, bool realB,
                                           uint32_t &cin_pass, uint32_t &mdeA, uint32_t &mdeB) {
	*reinterpret_cast<uint4 *>(&L.start[4 * lane]) = make_uint4(0u, 0u, 0u, 0u);
	const uint32_t cpos = lo + 16u * (uint32_t)lane;
	uint4 xv = make_uint4(0u, 0u, 0u, 0u);
	if (cpos < span) xv = *reinterpret_cast<const uint4 *>(base + cpos);   // aligned block holding >= 1 valid byte
	wave_sync();
	{
		const uint32_t pa = p0 - lo, pb = p1 - lo;     // wraps when the string begins before this pass
		if (realA && (!MULTI || pa < SF_BYTES)) atomicOr(&L.start[pa >> 2], 1u << (8u * (pa & 3u)));
		if (realB && (!MULTI || pb < SF_BYTES)) atomicOr(&L.start[pb >> 2], 1u << (8u * (pb & 3u)));
	}
	wave_sync();
	const uint4 sv = *reinterpret_cast<const uint4 *>(&L.start[4 * lane]);
	const uint32_t x[4] = {xv.x, xv.y, xv.z, xv.w}, st[4] = {sv.x, sv.y, sv.z, sv.w};
	uint32_t nd[4], M[4], S[4], u[4];
	md_chunk_prepare(x, st, nd, M, S);
	const uint32_t g = md_chunk_chain(M, S, 0u, u);
	const bool prop = (u[0] & u[1] & u[2] & u[3]) == 0xffffffffu;
	// carries between lanes: lane i generates (G) or passes on (P); into lane i = bit i of (a + b + cin) ^ P
	const unsigned long long G = __ballot(g != 0u), P = __ballot(prop);
	const unsigned long long a = G | P;
	const unsigned long long s = a + G + cin_pass;
	cin_pass = (s < a || (cin_pass && s == a)) ? 1u : 0u;
	const uint32_t cin = __builtin_amdgcn_inverse_ballot_w64(s ^ P) ? 1u : 0u;
	// (a lane without carry-in keeps its sums; with one, the carry ripples through its leading 0xFF.. words)
	md_chunk_chain(M, S, cin, u);
	const uint32_t c0 = u[0] & nd[0], c1 = u[1] & nd[1], c2 = u[2] & nd[2], c3 = u[3] & nd[3];
	const uint32_t i0 = (uint32_t)__popc(c0), i1 = i0 + (uint32_t)__popc(c1), i2 = i1 + (uint32_t)__popc(c2),
	               i3 = i2 + (uint32_t)__popc(c3);
	const uint32_t incl = wave_incl_scan(i3), E = incl - i3;
	*reinterpret_cast<uint4 *>(&L.c80[4 * lane]) = make_uint4(c0, c1, c2, c3);
	*reinterpret_cast<uint4 *>(&L.pre[4 * lane]) = make_uint4(E, E + i0, E + i1, E + i2);
	if (lane == 63) { L.c80[SF_WORDS] = 0u; L.pre[SF_WORDS] = incl; }
	wave_sync();
	int32_t b0 = (int32_t)(p0 - lo), b1 = (int32_t)(p1 - lo), b2 = (int32_t)(p2 - lo);
	if (MULTI) {
		b0 = b0 < 0 ? 0 : (b0 > SF_BYTES ? SF_BYTES : b0);
		b1 = b1 < 0 ? 0 : (b1 > SF_BYTES ? SF_BYTES : b1);
		b2 = b2 < 0 ? 0 : (b2 > SF_BYTES ? SF_BYTES : b2);
	}
	const uint32_t f0 = sf_count_before(L, (uint32_t)b0), f1 = sf_count_before(L, (uint32_t)b1),
	               f2 = sf_count_before(L, (uint32_t)b2);
	mdeA += f1 - f0;
	mdeB += f2 - f1;
	wave_sync();               // the next pass (or tile) rewrites the LDS image
}

// the part of a record after its statistics: --rescore, predicates, pool byte (msam_filter.c:31-63,132-183)
// EXTRA = false: the plain filter call (no per-record statistics out, no --rescore), the variant the
// bench runs; its dead arguments cost no scalar registers.
template <bool EXTRA>
__device__ __forceinline__ uint32_t sf_finish_record(const FilterArgs &A, uint32_t t, const SfAcc &s, bool bad, bool small) {
	uint32_t pooled = 0;
	if (EXTRA && A.o_status) A.o_status[t] = bad ? 1 : 0;
	if (EXTRA && A.o_len) {
		A.o_len[t] = (int32_t)s.alen; A.o_qlen[t] = (int32_t)s.qlen;
		A.o_qclip[t] = (int32_t)s.qclip; A.o_edit[t] = (int32_t)s.edit;
	}
	if (!bad) {
		if (EXTRA && A.rescore)          // msam_filter.c:160-168: hit=+1, miss=-1
			A.as_out[t] = (int32_t)((s.alen - s.edit) - s.edit);
		// msam_filter.c:31-35 in wrapping int32 arithmetic
		const bool fl = (int32_t)s.alen < A.min_length;
		bool fz, fp;
		const uint32_t ident = s.alen - s.edit;
		if (small) {
			// every operand of this wave fits 24 bits: v_mul_u32_u24 gives the same low 32 bits at full rate
			fz = (int32_t)__umul24(100u, s.qclip) > (int32_t)__umul24((uint32_t)A.max_clip, s.qlen);
			fp = (int32_t)__umul24(1000u, ident) < (int32_t)__umul24(s.alen, (uint32_t)A.ppt);
		} else {
			fz = (int32_t)(100u * s.qclip) > (int32_t)((uint32_t)A.max_clip * s.qlen);
			fp = (A.ppt < 0) ? ((int32_t)(1000u * (s.edit - s.alen)) < (int32_t)(s.alen * (uint32_t)A.ppt))
			                 : ((int32_t)(1000u * ident) < (int32_t)(s.alen * (uint32_t)A.ppt));
		}
		const bool fails = ((A.choice & 1) && fl) || ((A.choice & 2) && fp) || ((A.choice & 4) && fz);
		pooled = (A.choice == 0 || (int)fails == A.invert) ? 1u : 0u;   // msam_filter.c:181
	}
	return pooled;
}

// the byte handed to k_besthit_select (FilterArgs.pool_as_code) or written as keep
template <bool EXTRA>
__device__ __forceinline__ uint32_t sf_pool_byte(const FilterArgs &A, uint32_t pooled, uint32_t flag, uint32_t rf) {
	if (A.pool_as_code) {
		// msam_filter.c:223: AS is read from the record; after --rescore every mapped record has one (:167)
		const bool has = (rf & MSX_HAS_AS) || (EXTRA && A.rescore && !(flag & MSX_F_UNMAP));
		pooled = pooled ? (MSX_PC_IN | (has ? MSX_PC_HAS_AS : 0u) | (flag & MSX_F_MATES)) : 0u;
		pooled |= (flag & MSX_F_UNMAP) ? MSX_PC_UNMAP : 0u;
	}
	return pooled;
}

template <bool EXTRA>
__global__ __launch_bounds__(MSX_BLOCK) void k_aln_stats_flat(FilterArgs A) {
	__shared__ SfLds s_lds[MSX_BLOCK / 64];
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	SfLds &L = s_lds[wave];
	const int lane = threadIdx.x & 63;
	const uint32_t n = (uint32_t)A.n;
	const uint32_t n_tiles = (n + SF_RECS - 1) / SF_RECS;
	const uint32_t first = blockIdx.x * (MSX_BLOCK / 64) + wave;
	const uint32_t step = gridDim.x * (MSX_BLOCK / 64);
	if (first >= n_tiles) return;
	const uint32_t d = (uint32_t)(reinterpret_cast<uintptr_t>(A.md) & 15u);
	const bool stats_all = EXTRA && A.o_len != nullptr;      // msx_aln_stats: statistics for every record, mapped or not

	// (also tried: the next tile's first CIGAR words and MD bytes in flight as well, three tiles per wave --
	// 22 more registers, 5 waves per SIMD instead of 8, 0.65 ms against 0.59)
	SfTile cur, nxt;
	sf_load_tile(A, n, first, lane, cur);
	for (uint32_t tile = first; tile < n_tiles; tile += step) {
		const bool has_next = tile + step < n_tiles;       // (n_tiles + step < 2^32: n < 2^31)
		if (has_next) sf_load_tile(A, n, tile + step, lane, nxt);      // in flight while this tile is computed

		const uint32_t tA = tile * SF_RECS + 2u * (uint32_t)lane, tB = tA + 1;
		const bool realA = tA < n, realB = tB < n;
		const uint32_t flagA = cur.fl & 0xffffu, flagB = cur.fl >> 16;
		const uint32_t rfA = cur.rf & 0xffu, rfB = (cur.rf >> 8) & 0xffu;
		// msam_filter.c:132-138: unmapped records never reach the statistics
		const bool walkA = realA && (stats_all || !(flagA & MSX_F_UNMAP));
		const bool walkB = realB && (stats_all || !(flagB & MSX_F_UNMAP));
		const bool mdA = (rfA & MSX_HAS_MD) != 0, mdB = (rfB & MSX_HAS_MD) != 0;

		// ---- first CIGAR words: issued before the MD pass so that they arrive underneath it ----
		const uint32_t nA = cur.co1 - cur.co0, nB = cur.co2 - cur.co1;
		uint32_t cA = 0, cB = 0;
		if (walkA && nA) cA = A.cigar[cur.co0];
		if (walkB && nB) cB = A.cigar[cur.co1];

		// ---- MD: flat walk over the tile's bytes ----
		uint32_t mdeA = 0, mdeB = 0;
		{
			const uint32_t m_begin = __builtin_amdgcn_readfirstlane(cur.mo0);
			const uint32_t m_end = __builtin_amdgcn_readlane(cur.mo2, 63);
			// (a tile without MD bytes is skipped: every 16-byte block read then holds at least one byte
			// of the md array, so no read can leave the pages the array lies in)
			if (m_end > m_begin) {
				const uint32_t B0 = (m_begin + d) & ~15u;         // position 0, in (offset + d) space
				const uint32_t span = m_end + d - B0;
				const uint8_t *base = A.md + ((int64_t)B0 - (int64_t)d);    // 16-byte aligned address
				const uint32_t p0 = cur.mo0 + d - B0, p1 = cur.mo1 + d - B0, p2 = cur.mo2 + d - B0;
				uint32_t cin_pass = 0;
				if (span <= SF_BYTES) {
					sf_md_pass<false>(L, lane, base, span, 0u, p0, p1, p2, realA, realB, cin_pass, mdeA, mdeB);
				} else {
					for (uint32_t lo = 0; lo < span; lo += SF_BYTES)
						sf_md_pass<true>(L, lane, base, span, lo, p0, p1, p2, realA, realB, cin_pass, mdeA, mdeB);
				}
			}
		}

		// ---- CIGAR ----
		SfAcc sA = {0u, 0u, 0u, 0u}, sB = {0u, 0u, 0u, 0u};
		const uint32_t taA = mdA ? 0x187u : 0xff87u, teA = mdA ? 0x006u : 0u;   // alen: M I D = X (NM path: all but N P S H); edit: I D
		const uint32_t taB = mdB ? 0x187u : 0xff87u, teB = mdB ? 0x006u : 0u;
		sf_cigar_op(sA, cA, taA, teA);          // (an absent word is 0: zero M bases)
		sf_cigar_op(sB, cB, taB, teB);
		for (uint32_t k = 1; __ballot((walkA && k < nA) || (walkB && k < nB)) != 0ull; ++k) {
			uint32_t wa = 0, wb = 0;
			if (walkA && k < nA) wa = A.cigar[cur.co0 + k];
			if (walkB && k < nB) wb = A.cigar[cur.co1 + k];
			sf_cigar_op(sA, wa, taA, teA);
			sf_cigar_op(sB, wb, taB, teB);
		}
		// edit: MD path adds the MD count (mBamVector.c:101-118); NM path takes NM (msam_filter.c:155)
		bool badA = false, badB = false;
		if (walkA) {
			if (mdA) sA.edit += mdeA;
			else if (rfA & MSX_HAS_NM) sA.edit = (uint32_t)A.nm[tA];
			else badA = true;
		}
		if (walkB) {
			if (mdB) sB.edit += mdeB;
			else if (rfB & MSX_HAS_NM) sB.edit = (uint32_t)A.nm[tB];
			else badB = true;
		}
		if (__ballot(badA || badB) != 0ull) {      // msam_filter.c:150-152
			if (badA) atomicMin(&A.st->first_no_mdnm, (unsigned long long)tA);
			if (badB) atomicMin(&A.st->first_no_mdnm, (unsigned long long)tB);
		}

		// ---- predicates and the pool byte ----
		const bool small = A.ppt >= 0 && (uint32_t)A.ppt < (1u << 24) && (uint32_t)A.max_clip < (1u << 24) &&
		                   __ballot(((sA.alen | sA.qlen | sA.qclip | (sA.alen - sA.edit) | sB.alen | sB.qlen | sB.qclip |
		                              (sB.alen - sB.edit)) >> 24) != 0u) == 0ull;
		uint32_t pA = 0, pB = 0;
		if (EXTRA && A.as_out) {                   // replaced below when the record is rescored
			if (realA) A.as_out[tA] = A.as[tA];
			if (realB) A.as_out[tB] = A.as[tB];
		}
		// msam_filter.c:132-138: an unmapped record is pooled only for -k -v with a filter and PPT >= 0
		const uint32_t unmapped_pooled = (A.choice != 0 && A.keep_unmapped && A.ppt >= 0 && A.invert == 1) ? 1u : 0u;
		if (walkA) pA = sf_finish_record<EXTRA>(A, tA, sA, badA, small);
		else if (realA) pA = unmapped_pooled;
		if (walkB) pB = sf_finish_record<EXTRA>(A, tB, sB, badB, small);
		else if (realB) pB = unmapped_pooled;
		if (A.pool) {
			pA = sf_pool_byte<EXTRA>(A, pA, flagA, rfA);
			pB = sf_pool_byte<EXTRA>(A, pB, flagB, rfB);
			if (tile * SF_RECS + SF_RECS <= n && A.wide_ok) {
				*reinterpret_cast<uint16_t *>(A.pool + tA) = (uint16_t)(pA | (pB << 8));
			} else {
				if (realA) A.pool[tA] = (uint8_t)pA;
				if (realB) A.pool[tB] = (uint8_t)pB;
			}
		}
		cur = nxt;
	}
}

void msx_launch_aln_stats_flat(msx_ctx *ctx, const FilterArgs &A, int grid) {
	if (A.o_len || A.o_status || A.as_out || A.rescore)
		hipLaunchKernelGGL(k_aln_stats_flat<true>, dim3(grid), dim3(MSX_BLOCK), 0, ctx->stream, A);
	else
		hipLaunchKernelGGL(k_aln_stats_flat<false>, dim3(grid), dim3(MSX_BLOCK), 0, ctx->stream, A);
}

// msx_runtime_warmup: this translation unit's code object loaded onto the device ahead of its first launch (the runtime loads a
// module when one of its kernels is first asked for: 2-10 ms each, otherwise paid by the first batches of a command)
void msx_touch_stats(void) {
	hipFuncAttributes attr;
	(void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_aln_stats_flat<false>));
}
