// msx_crc.h -- CRC-32 (the gzip polynomial, reflected) of a byte string by one wave: BGZF trailers, read
// (msx_inflate.hip: k_bgzf_crc checks what htslib's bgzf_read checks under msam_helper.c:246-268) and written
// (msx_deflate.hip: the blocks sam_write1 / bgzf_write produce under msam_helper.c:270-272).
#ifndef MSX_CRC_H
#define MSX_CRC_H
#include <cstdint>

#define CRC_POLY 0xedb88320u
// a * b mod P; bit 31 is the coefficient of x^0
__host__ __device__ constexpr uint32_t crc_mul(uint32_t a, uint32_t b) {
	uint32_t r = 0u;
	for (int k = 0; k < 32; k++) {
		if (a & (0x80000000u >> k)) r ^= b;
		b = (b >> 1) ^ ((b & 1u) ? CRC_POLY : 0u);
	}
	return r;
}
// x^(8 n) mod P
__host__ __device__ constexpr uint32_t crc_xpow8(uint32_t n) {
	uint32_t r = 0x80000000u, p = 0x00800000u;       // 1, x^8
	while (n) {
		if (n & 1u) r = crc_mul(r, p);
		p = crc_mul(p, p);
		n >>= 1;
	}
	return r;
}
#define CRC_SLICE 64u                                 // bytes per lane and pass
#define CRC_PASS (64u * CRC_SLICE)                    // bytes per pass of a wave

// the byte table, 256 words of LDS, filled by the 64 lanes of a wave (barrier or wave-level wait by the caller)
__device__ __forceinline__ void crc_table_fill(uint32_t *tab, uint32_t lane) {
	for (uint32_t k = lane; k < 256u; k += 64u) {
		uint32_t c = k;
		for (int j = 0; j < 8; j++) c = (c >> 1) ^ ((c & 1u) ? CRC_POLY : 0u);
		tab[k] = c;
	}
}

// CRC-32 of p[0..n), n > 0, by one wave; the result in every lane.  Passes of 4 KB: every lane takes 64 consecutive
// bytes (a pass reads 4 KB of consecutive memory: every fetched line is used whole while it is in the L1) and runs them
// through the table.  The register is linear in what has been fed: state(A || B) = state(A) * x^(8 |B|) + state(B).
// Slices are aligned to the END of the string, so that everything to the right of a slice is full slices: a lane chains
// its own slices of successive passes with one constant (x^(8 * 4096), Horner), moves its sum over the (63 - lane)
// slices to its right at the end, and the lanes' results are added up.  What is short or empty is the front of the
// first pass, and a state of 0 contributes nothing.  The lane that holds byte 0 starts from the register's initial
// value (all ones).
__device__ __forceinline__ uint32_t crc_wave(const uint8_t *__restrict__ p, uint32_t n, const uint32_t *tab, uint32_t lane) {
	constexpr uint32_t X_PASS = crc_xpow8(CRC_PASS);
	const uint32_t n_pass = (n + CRC_PASS - 1u) / CRC_PASS;
	uint32_t acc = 0u;                               // this lane's slices of all passes: Horner over the passes
	for (uint32_t q = 0; q < n_pass; q++) {
		// this pass ends (n_pass - 1 - q) passes in front of the string's end
		const int64_t pass_end = (int64_t)n - (int64_t)(n_pass - 1u - q) * CRC_PASS;
		const int64_t lo_s = pass_end - (int64_t)(64u - lane) * CRC_SLICE;
		const uint32_t hi = (uint32_t)(lo_s + CRC_SLICE > 0 ? lo_s + CRC_SLICE : 0);
		const uint32_t lo = lo_s > 0 ? (uint32_t)lo_s : 0u;
		uint32_t s = (hi > 0u && lo == 0u) ? 0xffffffffu : 0u;
		uint32_t i = lo;
		for (; i + 4u <= hi; i += 4u) {
			const uint32_t w = *reinterpret_cast<const uint32_t __attribute__((aligned(1))) *>(p + i);
			s = tab[(s ^ w) & 0xffu] ^ (s >> 8);
			s = tab[(s ^ (w >> 8)) & 0xffu] ^ (s >> 8);
			s = tab[(s ^ (w >> 16)) & 0xffu] ^ (s >> 8);
			s = tab[(s ^ (w >> 24)) & 0xffu] ^ (s >> 8);
		}
		for (; i < hi; i++) s = tab[(s ^ p[i]) & 0xffu] ^ (s >> 8);
		acc = crc_mul(acc, X_PASS) ^ s;
	}
	// the lanes' sums, each moved over the 64-byte slices to its right, added up
	uint32_t t = crc_mul(acc, crc_xpow8(CRC_SLICE * (63u - lane)));
	for (uint32_t step = 32u; step >= 1u; step >>= 1) t ^= (uint32_t)__shfl_xor((int)t, step);
	return ~t;
}
#endif
