// msx_listkey.h -- sort key and signature of a multi-mapper list (used where the list is written:
// k_multi_compact, msx_profile.hip; consumed by the merged-list store, msx_prop.hip).
#ifndef MSX_LISTKEY_H
#define MSX_LISTKEY_H

#include "msx_internal.h"

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
	x ^= x >> 16; x *= 0x7feb352du;
	x ^= x >> 15; x *= 0x846ca68bu;
	x ^= x >> 16;
	return x;
}

// Signature of a list: sets of <= 3 features (the common case) are packed exactly --
// three 21-bit fields in ascending order, unused fields = SIG_PAD -- so that equal
// signatures mean equal sets and the set itself can be rebuilt from the signature;
// anything else gets bit 63, a 31-bit hash of the set and, in the low half, the number of the
// list (so that it can be found again after the sort), and is compared entry by entry.
#define SIG_PAD 0x1fffffu
#define SIG_HASHED (1ull << 63)

__device__ __forceinline__ uint32_t sig_len(unsigned long long sg) {
	return 1u + (((sg >> 21) & SIG_PAD) != SIG_PAD) + (((sg >> 42) & SIG_PAD) != SIG_PAD);
}

// bits of the sort key below the smallest feature: a hash of the set, so that equal sets end up adjacent
__host__ __device__ __forceinline__ int list_hash_bits(int feature_bits) {
	int hb = 32 - feature_bits;
	return hb > 12 ? 12 : (hb < 0 ? 0 : hb);
}

// key = (smallest feature << hash_bits) | set hash, signature as above, of the list f[0..n) that is list number j
__device__ __forceinline__ void list_key_sig(const int32_t *__restrict__ f, uint32_t n, uint32_t j, int hash_bits,
                                             uint32_t *key, unsigned long long *sig) {
	uint32_t mn = 0xffffffffu, h = n * 0x9e3779b9u, h2 = n * 0x85ebca6bu;
	unsigned long long sg;
	if (n <= 3u) {
		// independent loads, then a 3-element sorting network
		uint32_t f0 = SIG_PAD, f1 = SIG_PAD, f2 = SIG_PAD;
		if (n > 0u) f0 = (uint32_t)f[0];
		if (n > 1u) f1 = (uint32_t)f[1];
		if (n > 2u) f2 = (uint32_t)f[2];
		const bool fits = (n > 0u) && f0 < SIG_PAD && (n < 2u || f1 < SIG_PAD) && (n < 3u || f2 < SIG_PAD);
		if (n > 0u) { h += mix32(f0); h2 += mix32(f0 ^ 0x5bd1e995u); }
		if (n > 1u) { h += mix32(f1); h2 += mix32(f1 ^ 0x5bd1e995u); }
		if (n > 2u) { h += mix32(f2); h2 += mix32(f2 ^ 0x5bd1e995u); }
		uint32_t a = f0, b = f1, c = f2, t;
		if (a > b) { t = a; a = b; b = t; }
		if (b > c) { t = b; b = c; c = t; }
		if (a > b) { t = a; a = b; b = t; }
		mn = a;
		sg = fits ? ((unsigned long long)a | ((unsigned long long)b << 21) | ((unsigned long long)c << 42))
		          : (SIG_HASHED | ((unsigned long long)(h2 >> 1) << 32) | (unsigned long long)j);
		if (n == 0u) mn = 0xffffffffu;
	} else {
		for (uint32_t k = 0; k < n; ++k) {
			const uint32_t x = (uint32_t)f[k];
			mn = x < mn ? x : mn;
			h += mix32(x);                                   // commutative: a hash of the set
			h2 += mix32(x ^ 0x5bd1e995u);
		}
		sg = SIG_HASHED | ((unsigned long long)(h2 >> 1) << 32) | (unsigned long long)j;
	}
	const uint32_t hb = hash_bits > 0 ? (h & ((1u << hash_bits) - 1u)) : 0u;
	*key = hash_bits > 0 ? ((mn << hash_bits) | hb) : mn;
	*sig = sg;                                               // travels through the list sort as the value
}

#endif
