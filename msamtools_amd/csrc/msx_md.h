// msx_md.h -- the MD:Z token rule of bam_get_summary (mBamVector.c:112-118) as
// byte-at-a-time and dword-at-a-time (SWAR) state machines.  Shared by the
// stats kernel and a host test harness (tests/c/md_swar_test.c).
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

// Four MD bytes at once (SWAR on one dword, byte 0 = first character).  Same
// rule as md_byte: L = bytes that are neither digits nor '^'; a run of L bytes
// is counted when the byte before its first character is a digit.  lo/hi
// delimit the valid bytes of the dword ([lo, hi), hi > lo).
MSX_MD_FN void md_word(MdState &s, uint32_t x, uint32_t lo, uint32_t hi) {
	uint32_t vm = 0x80808080u;
	if (lo) vm &= 0xffffffffu << (8u * lo);
	if (hi < 4u) vm &= 0xffffffffu >> (8u * (4u - hi));
	const uint32_t x7 = x & 0x7f7f7f7fu;
	uint32_t D = (x7 + 0x50505050u) & ~(x7 + 0x46464646u) & ~x & vm;      // '0'..'9'
	const uint32_t y = x ^ 0x5e5e5e5eu;                                     // '^' -> zero byte
	const uint32_t C = ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y) & 0x80808080u;
	const uint32_t L = vm & ~D & ~C;
	const uint32_t Lprev = (L << 8) | (s.prevL ? 0x80u : 0u);
	const uint32_t Dprev = (D << 8) | (s.prevD ? 0x80u : 0u);
	const uint32_t cont = L & Lprev;                                        // letters continuing a run
	uint32_t Cn = (L & ~Lprev & Dprev) | ((cont & 0x80u) & (s.counting ? 0x80u : 0u));
	Cn |= (Cn << 8) & cont;
	Cn |= (Cn << 8) & cont;
	Cn |= (Cn << 8) & cont;
	s.edit += (int32_t)MSX_POPC(Cn);
	const uint32_t last = 0x80u << (8u * (hi - 1u));
	s.prevL = (L & last) ? 1u : 0u;
	s.prevD = (D & last) ? 1u : 0u;
	s.counting = (Cn & last) ? 1u : 0u;
}

// The same rule on words that were first realigned to the start of the string
// (v_alignbyte of two consecutive staged dwords), so byte 0 of word i is
// character 4i: no leading mask, and only the last word of a string needs a
// trailing one.  The state between words is kept as bit-7 masks taken straight
// from bit 31 of the previous word's masks.
struct MdBits {
	uint32_t L, D, Cn;     // 0x80 or 0: last byte was a letter / a digit / a counted letter
	uint32_t edit;
};

#ifdef __HIPCC__
#define MSX_ALIGNBYTE(hi, lo, sh) __builtin_amdgcn_alignbyte((hi), (lo), (sh))
#else
MSX_MD_FN uint32_t MSX_ALIGNBYTE(uint32_t hi, uint32_t lo, uint32_t sh) {
	return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8u * (sh & 3u)));
}
#endif

// vm: 0x80 in every valid byte (0x80808080 for a full word)
MSX_MD_FN void md_word_aligned(MdBits &s, uint32_t x, uint32_t vm) {
	const uint32_t x7 = x & 0x7f7f7f7fu;
	const uint32_t D = (x7 + 0x50505050u) & ~(x7 + 0x46464646u) & ~x & vm;    // '0'..'9'
	const uint32_t y = x ^ 0x5e5e5e5eu;                                       // '^' -> zero byte
	const uint32_t C = ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y);
	const uint32_t L = vm & ~D & ~C;
	const uint32_t Lprev = (L << 8) | s.L;
	const uint32_t Dprev = (D << 8) | s.D;
	const uint32_t cont = L & Lprev;                                          // letters continuing a run
	uint32_t Cn = (L & ~Lprev & Dprev) | (cont & s.Cn);
	Cn |= (Cn << 8) & cont;
	Cn |= (Cn << 8) & cont;
	Cn |= (Cn << 8) & cont;
	s.edit += (uint32_t)MSX_POPC(Cn);
	s.L = L >> 24;
	s.D = D >> 24;
	s.Cn = Cn >> 24;
}

// ---------------------------------------------------------------------------
// Flat walk: the same rule over a whole span of MD strings stored back to back,
// sixteen bytes per lane, work proportional to the bytes instead of to the
// longest string of the wave (k_aln_stats_filter).
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
