// msx_synth.h -- deterministic synthetic alignment stream, shared verbatim by
// the device generator kernels and the host twin (integer arithmetic only, so
// both produce identical bytes for the same (seed, group index)).
//
// Model (BASELINE.md section 2 / SURVEY.md 8d, after the reference's own
// validation generator, validation/generate_synthetic_alignments.py:165,
// :880-904, :1034, :1057-1059): paired 2x100 bp reads, QNAME-grouped; records
// per read = 1 + Poisson(mean_extra_hits); FLAGs by the build_flag rule;
// per-mate mismatches ~ {0:.50, 1:.30, 2:.10, 3:.10}; alternative hits carry
// extra mismatches; CIGAR mix 90% 100M, 5% soft-clipped, 5% single 1-3 bp
// indel; tags NM:i MD:Z AS:i with AS = aligned query bases - 2*NM; reference
// popularity log-uniform (Zipf s~1) with homologous neighbours so that
// multi-mapper lists repeat.
#ifndef MSX_SYNTH_H
#define MSX_SYNTH_H

#include <stdint.h>

#ifdef __HIPCC__
#define MSX_HD __host__ __device__ __forceinline__
#else
#define MSX_HD static inline
#endif

#define MSX_SYNTH_MAX_HITS 16
#define MSX_SYNTH_MD_CAP 56
#define MSX_SYNTH_READ_LEN 100

typedef struct {
	uint64_t seed;
	int32_t n_refs;
	int32_t lambda;      // mean extra hits, 0..4
	uint32_t log2n_q16;  // floor(log2(n_refs) * 65536)
} msx_synth_model;

typedef struct {
	uint16_t flag;
	int32_t tid;
	int32_t pos;
	int32_t nm;
	int32_t as;
	uint32_t n_cigar;
	uint32_t cigar[3];
	uint32_t md_len;
	uint8_t md[MSX_SYNTH_MD_CAP];
} msx_synth_rec;

MSX_HD uint64_t msx_mix64(uint64_t x) {
	x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
	x ^= x >> 27; x *= 0x94d049bb133111ebull;
	x ^= x >> 31;
	return x;
}

// independent 64-bit draw for (group, record, purpose)
MSX_HD uint64_t msx_draw(uint64_t seed, uint64_t g, uint32_t k, uint32_t purpose) {
	return msx_mix64(seed + 0x9e3779b97f4a7c15ull * (g + 1) + ((uint64_t)(k * 64u + purpose + 1u) << 40) * 0x632be59bd9b4e019ull
	                 + (uint64_t)(k * 64u + purpose + 1u));
}

// floor(log2(n) * 65536) with integer arithmetic (n >= 1)
MSX_HD uint32_t msx_log2_q16(uint32_t n) {
	uint32_t ip = 0;
	while ((n >> (ip + 1)) != 0) ip++;
	// mantissa in Q31: x in [1,2)
	uint64_t x = ((uint64_t)n << 31) >> ip;
	uint32_t frac = 0;
	for (int i = 0; i < 16; i++) {
		x = (x * x) >> 31;
		frac <<= 1;
		if (x >= (2ull << 31)) { x >>= 1; frac |= 1; }
	}
	return (ip << 16) | frac;
}

MSX_HD int msx_synth_hits(const msx_synth_model *m, uint64_t g) {
	// 1 + Poisson(lambda) by inverse CDF on a 32-bit uniform (tables: cdf * 2^32)
	const uint32_t T1[16] = {0x5e2d58d8u, 0xbc5ab1b1u, 0xeb715e1du, 0xfb239797u, 0xff1025f5u, 0xffd90f3bu, 0xfffa8b71u, 0xffff540cu, 0xffffed1fu, 0xfffffe21u, 0xffffffd4u, 0xfffffffcu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
	const uint32_t T2[16] = {0x22a55547u, 0x67efffd6u, 0xad3aaa65u, 0xdb6c716fu, 0xf28554f4u, 0xfbc27cc3u, 0xfed6df5du, 0xffb8201bu, 0xfff0704bu, 0xfffcf3e4u, 0xffff749cu, 0xffffe91bu, 0xfffffc85u, 0xffffff82u, 0xffffffefu, 0xfffffffdu};
	const uint32_t T3[16] = {0x0cbed866u, 0x32fb6199u, 0x6c562f66u, 0xa5b0fd33u, 0xd0b5178cu, 0xea845a8fu, 0xf76bfc10u, 0xfcf3d391u, 0xff06c461u, 0xffb7bf51u, 0xffecd766u, 0xfffb5254u, 0xfffef110u, 0xffffc6ecu, 0xfffff4c0u, 0xfffffdebu};
	const uint32_t T4[16] = {0x04b0556eu, 0x1771ab26u, 0x3cf45696u, 0x6ef7e5d6u, 0xa0fb7517u, 0xc8fe4e17u, 0xe3aadec2u, 0xf2e8e848u, 0xfa87ed0bu, 0xfdeb0b9bu, 0xff45e4a1u, 0xffc40500u, 0xffee0fcbu, 0xfffaff6bu, 0xfffeb199u, 0xffffadeau};
	if (m->lambda <= 0) return 1;
	const uint32_t u = (uint32_t)(msx_draw(m->seed, g, 0, 0) >> 32);
	int k = 0;
	for (; k < 15; k++) {
		uint32_t c = m->lambda == 1 ? T1[k] : m->lambda == 2 ? T2[k] : m->lambda == 3 ? T3[k] : T4[k];
		if (u < c) break;
	}
	return 1 + k;    // <= MSX_SYNTH_MAX_HITS
}

// log-uniform rank in [0, n_refs): P(rank < r) ~ log(r+1)/log(n)
MSX_HD int32_t msx_synth_zipf(const msx_synth_model *m, uint32_t u) {
	const uint64_t e = ((uint64_t)u * (uint64_t)m->log2n_q16) >> 32;   // Q16 exponent
	const uint32_t ip = (uint32_t)(e >> 16), fp = (uint32_t)(e & 0xffffu);
	uint64_t v = (1ull << ip) + (((1ull << ip) * fp) >> 16);           // 2^ip * (1 + frac)
	int64_t r = (int64_t)v - 1;
	if (r >= m->n_refs) r = m->n_refs - 1;
	if (r < 0) r = 0;
	return (int32_t)r;
}

MSX_HD uint32_t msx_synth_ref_len(uint64_t seed, int32_t tid) {
	return 400u + (uint32_t)(msx_mix64(seed ^ (0xabcdef12345ull + (uint64_t)tid * 0x9e3779b97f4a7c15ull)) & 0xfffu);
}

// mate (1 or 2) of record k of group g
MSX_HD uint32_t msx_synth_mate(const msx_synth_model *m, uint64_t g, uint32_t k) {
	if (k == 0) return 1;
	return 1u + (uint32_t)(msx_draw(m->seed, g, k, 1) & 1u);
}

MSX_HD uint32_t msx_put_num(uint8_t *dst, uint32_t v) {
	if (v >= 100) { dst[0] = (uint8_t)(48 + v / 100); dst[1] = (uint8_t)(48 + (v / 10) % 10); dst[2] = (uint8_t)(48 + v % 10); return 3; }
	if (v >= 10) { dst[0] = (uint8_t)(48 + v / 10); dst[1] = (uint8_t)(48 + v % 10); return 2; }
	dst[0] = (uint8_t)(48 + v);
	return 1;
}

// Record k (of h) of group g.  mates = bitmask, bit j set when record j is mate 2.
MSX_HD void msx_synth_record(const msx_synth_model *m, uint64_t g, uint32_t k, uint32_t h, uint32_t mates,
                             msx_synth_rec *r) {
	const char BASES[4] = {'A', 'C', 'G', 'T'};
	const uint32_t mate2 = (mates >> k) & 1u;
	// secondary: an earlier record of the same mate exists; mate_present: the other mate has a record
	const uint32_t lower = mates & ((1u << k) - 1u);
	const uint32_t same_before = mate2 ? lower : (~mates & ((1u << k) - 1u));
	const uint32_t all = (h >= 32) ? 0xffffffffu : ((1u << h) - 1u);
	const uint32_t other_any = mate2 ? (~mates & all) : (mates & all);
	const uint64_t d_flag = msx_draw(m->seed, g, k, 2);
	uint32_t flag = 0x1u | (mate2 ? 0x80u : 0x40u);             // build_flag, :880-904
	if (d_flag & 1u) flag |= 0x10u;
	if (other_any) { flag |= 0x2u; if (d_flag & 2u) flag |= 0x20u; }
	else flag |= 0x8u;
	if (same_before) flag |= 0x100u;
	r->flag = (uint16_t)flag;

	// reference: r0 for the primary of each mate; alternatives are the same
	// reference (repeat), a homologous neighbour, or an unrelated one
	const int32_t r0 = msx_synth_zipf(m, (uint32_t)(msx_draw(m->seed, g, 0, 3) >> 32));
	int32_t tid = r0;
	if (same_before) {
		const uint64_t d = msx_draw(m->seed, g, k, 4);
		const uint32_t sel = (uint32_t)(d & 7u);
		if (sel < 2) tid = r0;
		else if (sel < 7) tid = (int32_t)(((int64_t)r0 + 1 + (int64_t)((d >> 8) & 3u)) % m->n_refs);
		else tid = msx_synth_zipf(m, (uint32_t)(d >> 32));
	}
	r->tid = tid;
	const uint32_t rl = msx_synth_ref_len(m->seed, tid);
	r->pos = (int32_t)((msx_draw(m->seed, g, k, 5) >> 16) % (uint64_t)(rl - 150u));

	// mismatches: {0:.50,1:.30,2:.10,3:.10}; alternatives add {0:.4,1:.3,2:.2,4:.1}
	const uint64_t d_nm = msx_draw(m->seed, g, k, 6);
	const uint32_t u10 = (uint32_t)((d_nm & 0xffffu) * 10u >> 16);
	uint32_t nmis = u10 < 5 ? 0u : u10 < 8 ? 1u : u10 < 9 ? 2u : 3u;
	if (same_before) {
		const uint32_t v10 = (uint32_t)(((d_nm >> 16) & 0xffffu) * 10u >> 16);
		nmis += v10 < 4 ? 0u : v10 < 7 ? 1u : v10 < 9 ? 2u : 4u;
	}

	// CIGAR class
	const uint64_t d_c = msx_draw(m->seed, g, k, 7);
	const uint32_t cls = (uint32_t)((d_c & 0xffffu) * 20u >> 16);   // 0..19
	uint32_t Lm = MSX_SYNTH_READ_LEN;   // M bases (MD covers these)
	uint32_t del_at = 0, del_len = 0, ins_len = 0;
	if (cls == 0) {                                                  // soft clip, 5%
		const uint32_t kc = 1u + (uint32_t)(((d_c >> 16) & 0xffffu) * 40u >> 16);
		Lm = MSX_SYNTH_READ_LEN - kc;
		r->n_cigar = 2;
		if ((d_c >> 32) & 1u) { r->cigar[0] = (kc << 4) | 4u; r->cigar[1] = (Lm << 4) | 0u; }
		else { r->cigar[0] = (Lm << 4) | 0u; r->cigar[1] = (kc << 4) | 4u; }
		r->cigar[2] = 0;
	} else if (cls == 1) {                                           // single indel, 5%
		const uint32_t il = 1u + (uint32_t)(((d_c >> 16) & 0xffffu) * 3u >> 16);
		const uint32_t at = 10u + (uint32_t)(((d_c >> 32) & 0xffffu) * 80u >> 16);
		r->n_cigar = 3;
		if ((d_c >> 48) & 1u) {                                      // insertion
			ins_len = il;
			Lm = MSX_SYNTH_READ_LEN - il;
			const uint32_t a = at < Lm ? at : Lm - 1u;
			r->cigar[0] = (a << 4) | 0u; r->cigar[1] = (il << 4) | 1u; r->cigar[2] = ((Lm - a) << 4) | 0u;
		} else {                                                     // deletion
			del_len = il;
			del_at = at;
			r->cigar[0] = (at << 4) | 0u; r->cigar[1] = (il << 4) | 2u;
			r->cigar[2] = ((MSX_SYNTH_READ_LEN - at) << 4) | 0u;
		}
	} else {
		r->n_cigar = 1;
		r->cigar[0] = (MSX_SYNTH_READ_LEN << 4) | 0u;
		r->cigar[1] = r->cigar[2] = 0;
	}

	// MD: stratified mismatch positions over the Lm aligned bases, deletion at del_at
	uint32_t n = 0, run = 0, cur = 0;   // cur = next M offset to account for
	const uint32_t stratum = nmis ? Lm / nmis : 0;
	bool del_done = del_len == 0;
	for (uint32_t j = 0; j <= nmis; j++) {
		uint32_t p = Lm;   // sentinel: end of the alignment
		uint8_t base = 0;
		if (j < nmis) {
			const uint64_t d = msx_draw(m->seed, g, k, 8 + j);
			p = j * stratum + (uint32_t)((d & 0xffffu) * stratum >> 16);
			base = (uint8_t)BASES[(d >> 16) & 3u];
		}
		if (!del_done && del_at <= p) {
			run += del_at - cur;
			cur = del_at;
			n += msx_put_num(r->md + n, run);
			r->md[n++] = '^';
			const uint64_t dd = msx_draw(m->seed, g, k, 20);
			for (uint32_t q = 0; q < del_len; q++) r->md[n++] = (uint8_t)BASES[(dd >> (2 * q)) & 3u];
			run = 0;
			del_done = true;
		}
		run += p - cur;
		n += msx_put_num(r->md + n, run);
		if (j < nmis) {
			r->md[n++] = base;
			cur = p + 1;
			run = 0;
		}
	}
	r->md_len = n;
	r->nm = (int32_t)(nmis + del_len + ins_len);
	r->as = (int32_t)(Lm + ins_len) - 2 * r->nm;
}

#endif
