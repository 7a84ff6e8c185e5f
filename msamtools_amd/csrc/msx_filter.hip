// msx_filter.hip -- device side of `msamtools filter`:
//   (k_aln_stats_flat, the per-record CIGAR/MD walk with the -l/-p/-z predicates, is in msx_stats.hip)
//   k_besthit_select    per-pool best-hit / unique-best-hit selection, mate aware
//                                                 (msam_filter.c:192-263)
//   k_emit_*            the order in which the reference calls mSamWrite
//                                                 (msam_filter.c:235-244, mBamVector.c:343-348)
// All integer work, HBM-bound; no MFMA.
#include "msx_internal.h"
#include "msx_count.h"
#include "msx_md.h"
#include "msx_stats.h"

#include <climits>
#include <cstdlib>

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
	int32_t partial;          // msx_filter_params.fatal_pool_partial
	uint8_t *keep;            // [n]
	uint32_t *gcount;         // [n_groups] records written per pool
	msx_dev_status *st;
};

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

// The pool walked by one lane, record after record: any pool length (a wave whose 64 pools do not fit the
// flat path below takes it).
__device__ __forceinline__ uint32_t bh_rec_code(const SelectArgs &A, uint32_t i) {
	// MSX_PC_IN | MSX_PC_HAS_AS | mate bits of a record that takes part, 0 otherwise
	// (+ MSX_PC_UNMAP for an unmapped record, pooled or not)
	if (A.pool_is_code) return A.pool[i];                // k_aln_stats_flat has looked at FLAG and the aux bits already
	const uint32_t fl = A.flag[i];
	const bool pooled = A.pool ? (A.pool[i] != 0) : !(fl & MSX_F_UNMAP);
	const bool has = (A.rflags[i] & MSX_HAS_AS) || (A.rescored && !(fl & MSX_F_UNMAP));
	return (pooled ? (MSX_PC_IN | (has ? MSX_PC_HAS_AS : 0u) | (fl & MSX_F_MATES)) : 0u) |
	       ((fl & MSX_F_UNMAP) ? MSX_PC_UNMAP : 0u);
}

template <bool COUNT>
__device__ __forceinline__ void bh_pool_serial(const SelectArgs &A, const CountArgs &P, int64_t g, uint32_t s, uint32_t e,
                                               int32_t *s_key, uint32_t *s_val, BlockCounts &bc) {
	BhAcc c = {INT_MIN, INT_MIN, INT_MIN, 0u, 0u, 0u, 0u, 0xffffffffu, 0xffffffffu, 0xffffffffu};
	uint32_t cnt = 0;
	for (uint32_t i = s; i < e; ++i) {
		const uint32_t pc = bh_rec_code(A, i);
		bh_count(c, i, pc & MSX_F_MATES, (pc & MSX_PC_IN) != 0, (pc & MSX_PC_HAS_AS) != 0, A.as[i]);
	}
	bool w0 = !c.paired && c.n0 > 0 && (!A.unique_only || c.n0 == 1);
	bool w1 = c.paired && c.n1 > 0 && (!A.unique_only || c.n1 == 1);
	bool w2 = c.paired && c.n2 > 0 && (!A.unique_only || c.n2 == 1);
	if (A.partial) {            // what the reference had written when it died in this pool: a pass that meets no record without AS
		w0 = w0 && c.noas0 == 0xffffffffu;
		w1 = w1 && c.noas1 == 0xffffffffu;
		w2 = w2 && c.noas1 == 0xffffffffu && c.noas2 == 0xffffffffu;
	}
	for (uint32_t i = s; i < e; ++i) {
		const uint32_t pc = bh_rec_code(A, i);
		const uint8_t k = bh_keep(c, w0, w1, w2, pc & MSX_F_MATES, (pc & MSX_PC_IN) != 0, (pc & MSX_PC_HAS_AS) != 0, A.as[i]);
		A.keep[i] = k;
		cnt += (k != 0);
	}
	if (COUNT) {
		if (pool_follows(P, g) || pool_follows(P, g + 1)) {
			if (P.pinfo) P.pinfo[g] = MSX_PINFO_NONE;        // member of a chain of pools: k_insert_chains counts it
		} else {
			// the keep codes this lane just wrote, in output order
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

// ---------------------------------------------------------------------------
// k_besthit_select: one wave per 64 consecutive pools, one lane per RECORD for the selection.
// The records of 64 pools are one contiguous range [S, S + R) of the batch (R ~ 320 at 5 hits per read), so the
// pool bytes, scores and keep codes move as whole rows of 64 consecutive records -- coalesced, where one lane
// per pool fetched ~27 dwords per pool through sub-dword windows and spent 966 lane-instructions on it:
//   1  lane p puts its pool's start into a bitmap of the range (LDS); a record's pool is the number of
//      starts at or before it (word prefix + popcount)
//   2  every record adds its score to its pool's maximum per mate class, and its mate bits to the pool's
//      "paired" word (LDS atomics; msam_filter.c:196-230)
//   3  every record compares itself with its pool's maximum (--uniqhit: ties counted in between, :232-233),
//      writes its keep code and sets its bit in the pool's first-pass / second-pass winner mask
//   4  lane p again: records written, the fatal-record check and -- COUNT, the fused `filter | profile`
//      form -- the insert accounting of msx_count.h straight from the winner masks.
// A wave whose range exceeds BF_RMAX records, or with a pool of more than 32 records (the masks) or none,
// walks its pools one lane each (bh_pool_serial).
// ---------------------------------------------------------------------------
#define BF_ROWS 12
#define BF_RMAX (64 * BF_ROWS)
#define BF_WORDS (BF_RMAX / 32)

struct BfWave {
	uint32_t bits[BF_WORDS], wpre[BF_WORDS];
	int32_t best[3][64];          // per mate class: neither bit, READ1, READ2
	uint32_t ties[3][64], noas[3][64];
	uint32_t pair[64], k1[64], k2[64], start[64];
	uint32_t fum[64];             // the pool's first record is unmapped (chains of pools, msx_count.h)
};

__device__ __forceinline__ void bf_wave_sync() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool COUNT>
__global__ __launch_bounds__(MSX_BLOCK) void k_besthit_select(SelectArgs A, CountArgs P) {
	__shared__ BfWave s_w[MSX_BLOCK / 64];
	__shared__ uint32_t s_c[3][MSX_BLOCK / 64];
	__shared__ int32_t s_key[COUNT ? UI_TBL : 1];
	__shared__ uint32_t s_val[COUNT ? UI_TBL : 1];
	BlockCounts bc = {0u, 0u, 0u};
	if (COUNT) count_block_begin(P, s_key, s_val);
	const uint32_t lane = threadIdx.x & 63u;
	BfWave &L = s_w[threadIdx.x >> 6];
	const int64_t n_tiles = (A.n_groups + 63) >> 6;
	const int64_t wstride = (int64_t)gridDim.x * (MSX_BLOCK / 64);
	int64_t tile = (int64_t)blockIdx.x * (MSX_BLOCK / 64) + (threadIdx.x >> 6);
	// the next tile's bounds are fetched while this one is processed
	uint32_t s_nx = 0, e_nx = 0;
	if (tile * 64 + lane < A.n_groups) { s_nx = A.group_off[tile * 64 + lane]; e_nx = A.group_off[tile * 64 + lane + 1]; }
	for (; tile < n_tiles; tile += wstride) {
		const int64_t g = tile * 64 + lane;
		const bool gv = g < A.n_groups;
		const uint32_t s = s_nx, e = e_nx;
		{
			const int64_t gn = g + wstride * 64;
			if (gn < A.n_groups) { s_nx = A.group_off[gn]; e_nx = A.group_off[gn + 1]; }
		}
		const unsigned long long vm = __ballot(gv);
		const uint32_t S = (uint32_t)__builtin_amdgcn_readfirstlane((int)s);
		const uint32_t E = (uint32_t)__builtin_amdgcn_readlane((int)e, 63 - __clzll((long long)vm));
		const uint32_t R = E - S, len = e - s;
		if (R > BF_RMAX || __ballot(gv && (len == 0u || len > 32u)) != 0ull) {
			if (gv) bh_pool_serial<COUNT>(A, P, g, s, e, s_key, s_val, bc);
			continue;
		}
		// ---- 1: pool slots, start bitmap ----
		if (lane < BF_WORDS) L.bits[lane] = 0u;
		L.best[0][lane] = INT_MIN; L.best[1][lane] = INT_MIN; L.best[2][lane] = INT_MIN;
		L.noas[0][lane] = 0xffffffffu; L.noas[1][lane] = 0xffffffffu; L.noas[2][lane] = 0xffffffffu;
		L.pair[lane] = 0u; L.k1[lane] = 0u; L.k2[lane] = 0u;
		L.start[lane] = s - S;
		if (COUNT && P.chain_flag) L.fum[lane] = 0u;
		if (A.unique_only) { L.ties[0][lane] = 0u; L.ties[1][lane] = 0u; L.ties[2][lane] = 0u; }
		// all of the range's records in flight before anything is looked at
		uint32_t pc[BF_ROWS];
		int32_t sc[BF_ROWS];
#pragma unroll
		for (int r = 0; r < BF_ROWS; r++) {
			const uint32_t off = 64u * (uint32_t)r + lane;
			pc[r] = 0u; sc[r] = 0;
			if (off < R) { pc[r] = bh_rec_code(A, S + off); sc[r] = A.as[S + off]; }
		}
		bf_wave_sync();
		if (gv) atomicOr(&L.bits[(s - S) >> 5], 1u << ((s - S) & 31u));
		bf_wave_sync();
		{
			const uint32_t wc = lane < BF_WORDS ? (uint32_t)__popc(L.bits[lane]) : 0u;
			uint32_t incl = wc;
#pragma unroll
			for (int d = 1; d < 32; d <<= 1) {
				const uint32_t o = __shfl_up(incl, d, 64);
				if (lane >= (uint32_t)d) incl += o;
			}
			if (lane < BF_WORDS) L.wpre[lane] = incl - wc;
		}
		bf_wave_sync();
		// ---- 2: maxima per pool and mate class ----
#pragma unroll
		for (int r = 0; r < BF_ROWS; r++) {
			if (64u * (uint32_t)r >= R) break;                                   // (wave-uniform)
			const uint32_t off = 64u * (uint32_t)r + lane;
			if (off < R) {
				const uint32_t wd = off >> 5;
				const uint32_t pid = L.wpre[wd] + (uint32_t)__popc(L.bits[wd] & (0xffffffffu >> (31u - (off & 31u)))) - 1u;
				const uint32_t c = pc[r], cls = c & MSX_F_MATES, ci = cls >> 6;
				if (c & MSX_PC_IN) {
					if (cls) atomicOr(&L.pair[pid], cls);                         // mBamPoolIsPaired :196-204
					if (ci < 3u) {
						if (c & MSX_PC_HAS_AS) atomicMax(&L.best[ci][pid], sc[r]);
						else atomicMin(&L.noas[ci][pid], S + off);
					}
				}
				if (COUNT && P.chain_flag && (c & MSX_PC_UNMAP) && off == L.start[pid]) L.fum[pid] = 1u;
				pc[r] = c | (pid << 8);
			}
		}
		bf_wave_sync();
		// ---- 3: winners ----
#pragma unroll
		for (int r = 0; r < BF_ROWS; r++) {
			if (64u * (uint32_t)r >= R) break;
			const uint32_t off = 64u * (uint32_t)r + lane;
			const uint32_t c = pc[r], ci = (c & MSX_F_MATES) >> 6, pid = c >> 8;
			bool win = false;
			if (off < R && (c & MSX_PC_IN) && (c & MSX_PC_HAS_AS) && ci < 3u) {
				// a paired pool is judged per mate (READ1, READ2), an unpaired one as a whole
				const bool judged = L.pair[pid] ? (ci != 0u) : (ci == 0u);
				win = judged && sc[r] == L.best[ci][pid];
				// (wave-uniform switch) the pool the reference died in: a pass that met a record without AS wrote nothing,
				// and the READ2 pass comes after the READ1 pass
				if (A.partial && (L.noas[ci][pid] != 0xffffffffu || (ci == 2u && L.noas[1][pid] != 0xffffffffu))) win = false;
			}
			if (A.unique_only && win) atomicAdd(&L.ties[ci][pid], 1u);
			pc[r] = (c & ~(uint32_t)MSX_PC_IN) | (win ? MSX_PC_IN : 0u);          // bit 0 from here on: winner
		}
		if (A.unique_only) bf_wave_sync();
#pragma unroll
		for (int r = 0; r < BF_ROWS; r++) {
			if (64u * (uint32_t)r >= R) break;
			const uint32_t off = 64u * (uint32_t)r + lane;
			if (off < R) {
				const uint32_t c = pc[r], ci = (c & MSX_F_MATES) >> 6, pid = c >> 8;
				bool win = (c & MSX_PC_IN) != 0u;
				if (A.unique_only && win && L.ties[ci][pid] != 1u) win = false;   // :232-233
				// keep codes: 1 = written in the first pass (unpaired winners, READ1 winners), 2 = READ2 winners
				A.keep[S + off] = (uint8_t)(win ? (ci == 2u ? 2u : 1u) : 0u);
				if (win) atomicOr(ci == 2u ? &L.k2[pid] : &L.k1[pid], 1u << (off - L.start[pid]));
			}
		}
		bf_wave_sync();
		// ---- 4: per pool ----
		if (gv) {
			const uint32_t k1 = L.k1[lane], k2 = L.k2[lane];
			A.gcount[g] = (uint32_t)__popc(k1 | k2);
			// msam_filter.c:219-221: a participating record without AS is fatal
			const uint32_t n1 = L.noas[1][lane], n2 = L.noas[2][lane];
			const uint32_t bad = L.pair[lane] ? (n1 < n2 ? n1 : n2) : L.noas[0][lane];
			if (bad != 0xffffffffu) atomicMin(&A.st->first_no_as, (unsigned long long)bad);
			if (COUNT) {
				// a pool that begins with an unmapped record continues the insert of the pool before it: the members
				// of such a chain are counted together by k_insert_chains (msx_count.h)
				bool member = false;
				if (P.chain_flag) {
					const bool last = ((vm >> lane) >> 1) == 0ull;                     // the tile's last pool: its successor is another wave's
					member = (g > 0 && L.fum[lane] != 0u) || (last ? pool_follows(P, g + 1) : L.fum[(lane + 1u) & 63u] != 0u);
				}
				if (member) {
					if (P.pinfo) P.pinfo[g] = MSX_PINFO_NONE;
				} else {
					PoolAcc v;
					pool_begin(P, v, s);
					pool_visit_masks(P, v, s, k1, k2);
					pool_finish(P, g, v, s_key, s_val, bc);
				}
			}
		}
		bf_wave_sync();                                                            // the slots are reused by the next tile
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

// The statistics kernel: k_aln_stats_flat (msx_stats.hip), one wave per 128 records
static void launch_stats(msx_ctx *ctx, FilterArgs &A) {
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
	msx_join(ctx);
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
		S.partial = p->fatal_pool_partial ? 1 : 0;
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
		if (prof && ng > 0 && P.chain_flag) msx_profile_count_chains(ctx, P);
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
		// The side lanes are left running: whatever follows on this context joins them first (msx_join at the
		// head of every entry point) -- except the build of the sharing store in msx_profile_prop_begin,
		// which does not depend on them and overlaps their tails.
		msx_lane_leave(ctx);
		if (rc) { msx_join(ctx); return rc; }
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
	// The chain rule that joins filter's pools into profile's inserts (k_insert_chains) rests on unmapped records never
	// being written.  -k -v with thresholds writes them (msam_filter.c:132-138), and one with a name and an RNAME of its
	// own would split a read's alignments into two inserts for the pipe.  The reference's command line never gets there
	// (--invert cannot be combined with --besthit or --uniqhit, msam_filter.c:398-402); neither does this entry point.
	if (p && p->invert && (p->besthit || p->uniqhit) && b && b->pool_rule == MSX_POOLS_FILTER)
		return msx_fail(ctx, MSX_ERR_ARG, "--invert cannot be combined with --besthit or --uniqhit");
	return filter_enqueue_impl(ctx, b, p, out, prof);
}

extern "C" int msx_filter_finish(msx_ctx *ctx, msx_filter_status *status) {
	if (!ctx) return MSX_ERR_ARG;
	msx_join(ctx);
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
	msx_join(ctx);
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

// msx_runtime_warmup: this translation unit's code object loaded onto the device ahead of its first launch (the runtime loads a
// module when one of its kernels is first asked for: 2-10 ms each, otherwise paid by the first batches of a command)
void msx_touch_filter(void) {
	hipFuncAttributes attr;
	(void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_besthit_select<true>));
}
