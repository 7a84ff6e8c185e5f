// msx_md.h -- the MD:Z token rule of bam_get_summary (mBamVector.c:112-118): byte at a
// time (md_byte, the rule as the reference applies it) and as the carry-chain arithmetic
// of the statistics kernel's flat walk.  Shared by msx_stats.hip and a host test harness
// (tests/c/md_flat_test.c).
#ifndef MSX_MD_H
#define MSX_MD_H

#include <stdint.h>

#ifdef __HIPCC__
#define MSX_MD_FN __device__ __forceinline__
#define MSX_POPC(x) __popc(x)
#else
#define MSX_MD_FN static inline
#define MSX_POPC(x) __builtin_popcount(x)
#endif

// One MD byte through the token rule of mBamVector.c:112-118: count the bytes
// of every maximal run of non-[^0-9] characters whose predecessor is a digit
// (a run at the start of the string or right after '^' is not counted).
struct MdState {
	uint32_t prevL, prevD, counting;
	int32_t edit;
};

// ---------------------------------------------------------------------------
// Flat walk: the same rule over a whole span of MD strings stored back to back,
// sixteen bytes per lane, work proportional to the bytes instead of to the
// longest string of the wave (k_aln_stats_flat).
//
// Restated rule: a byte that is neither a digit nor '^' (a "letter") counts
// unless the maximal run of letters it belongs to begins at the first byte of
// its string or right after a '^'.  Let a *member* be any non-digit byte
// (letters and carets).  Give every member byte the value 0xFF and every digit
// 0x7F, and add 1 at every caret and at every first byte of a string: the carry
// of that addition walks exactly through the letters that must NOT be counted
// (it starts at a caret or at a string start, runs through 0xFF bytes and is
// absorbed by the first digit), turning them into 0x00/0x01, while a letter
// that no carry reaches keeps its 0xFF.  A caret always receives its own +1,
// so it never keeps bit 7.  Hence
//     counted = sum & nondigit80          (bit 7 of every counted byte)
// and the carry between words / lanes / passes is an ordinary add-with-carry.
// A carry arriving at a string's first byte from the string before it changes
// nothing there (that byte is already forced), so strings need no separator.
// ---------------------------------------------------------------------------
// bit 7 of every byte that is not '0'..'9'
MSX_MD_FN uint32_t md_nondigit80(uint32_t x) {
	const uint32_t z = x ^ 0x30303030u;                      // digits -> 0x00..0x09
	return (((z & 0x7f7f7f7fu) + 0x76767676u) | z) & 0x80808080u;
}
// bit 7 of every byte that is '^'
MSX_MD_FN uint32_t md_caret80(uint32_t x) {
	const uint32_t y = x ^ 0x5e5e5e5eu;                      // '^' -> zero byte
	return ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y) & 0x80808080u;
}
// One lane's 16 bytes: x = data words, st = 0x01 in every byte that starts a string.
// nd = non-digit flags, M/S = the two addends described above.
MSX_MD_FN void md_chunk_prepare(const uint32_t x[4], const uint32_t st[4], uint32_t nd[4], uint32_t M[4], uint32_t S[4]) {
	for (int q = 0; q < 4; q++) {
		nd[q] = md_nondigit80(x[q]);
		M[q] = nd[q] | 0x7f7f7f7fu;
		S[q] = (md_caret80(x[q]) >> 7) | st[q];
	}
}
// u = M + S + cin over the lane's 128 bits; returns the carry out
MSX_MD_FN uint32_t md_chunk_chain(const uint32_t M[4], const uint32_t S[4], uint32_t cin, uint32_t u[4]) {
#ifdef __HIPCC__
	// v_addc_co_u32 chain
	unsigned int c = cin;
	u[0] = __builtin_addc(M[0], S[0], c, &c);
	u[1] = __builtin_addc(M[1], S[1], c, &c);
	u[2] = __builtin_addc(M[2], S[2], c, &c);
	u[3] = __builtin_addc(M[3], S[3], c, &c);
	return c;
#else
	unsigned long long t = (unsigned long long)M[0] + S[0] + cin;
	u[0] = (uint32_t)t;
	t = (unsigned long long)M[1] + S[1] + (t >> 32);
	u[1] = (uint32_t)t;
	t = (unsigned long long)M[2] + S[2] + (t >> 32);
	u[2] = (uint32_t)t;
	t = (unsigned long long)M[3] + S[3] + (t >> 32);
	u[3] = (uint32_t)t;
	return (uint32_t)(t >> 32);
#endif
}

MSX_MD_FN void md_byte(MdState &s, uint32_t c, bool valid) {
	uint32_t isD = (c - 48u) < 10u;
	uint32_t isL = (!isD && c != 94u) ? 1u : 0u;
	if (valid) {
		if (isL & (s.prevL ^ 1u)) s.counting = s.prevD;
		s.edit += (int32_t)(isL & s.counting);
		s.prevL = isL;
		s.prevD = isD;
	}
}

#endif
