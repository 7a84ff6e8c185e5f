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
