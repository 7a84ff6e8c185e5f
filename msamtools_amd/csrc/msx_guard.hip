// msx_guard.hip -- MSX_GUARD=1: guard bytes around every device allocation of the library (msx_internal.h); the one translation
// unit that sees the runtime's own hipMalloc / hipFree.
#define MSX_GUARD_IMPL
#include "msx_internal.h"

#include <cstdlib>
#include <mutex>
#include <unordered_map>

#define MSX_GUARD_BYTES 512
#define MSX_GUARD_FILL 0xC3
namespace {
struct guard_rec { char *base; size_t n; };
std::mutex g_guard_mu;
std::unordered_map<void *, guard_rec> g_guard;
// the damaged bytes of one allocation's guards (0: intact); prints the first
size_t guard_damage(void *user, const guard_rec &r, const char *when) {
	unsigned char h[2 * MSX_GUARD_BYTES];
	if (hipMemcpy(h, r.base, MSX_GUARD_BYTES, hipMemcpyDeviceToHost) != hipSuccess ||
	    hipMemcpy(h + MSX_GUARD_BYTES, (char *)user + r.n, MSX_GUARD_BYTES, hipMemcpyDeviceToHost) != hipSuccess) {
		fprintf(stderr, "MSX_GUARD: cannot read the guards of %p, a %zu-byte allocation (%s)\n", user, r.n, hipGetErrorString(hipGetLastError()));
		return 1;
	}
	size_t bad = 0;
	long first = 0;
	for (int i = 0; i < 2 * MSX_GUARD_BYTES; i++)
		if (h[i] != MSX_GUARD_FILL) {
			if (!bad) first = i < MSX_GUARD_BYTES ? (long)i - MSX_GUARD_BYTES : (long)r.n + (i - MSX_GUARD_BYTES);
			bad++;
		}
	if (bad)
		fprintf(stderr, "MSX_GUARD: %zu guard byte(s) of a %zu-byte device allocation overwritten (%s); the first at offset %ld "
		        "(the allocation is [0, %zu))\n", bad, r.n, when, first, r.n);
	return bad;
}
}   // namespace

bool msx_guard_on() {
	static const bool on = getenv("MSX_GUARD") && atoi(getenv("MSX_GUARD")) > 0;
	return on;
}

hipError_t msx_guard_malloc(void **p, size_t n) {
	if (!msx_guard_on()) return hipMalloc(p, n);
	char *base = nullptr;
	hipError_t e = hipMalloc((void **)&base, n + 2 * MSX_GUARD_BYTES);
	if (e != hipSuccess) return e;
	if ((e = hipMemset(base, MSX_GUARD_FILL, MSX_GUARD_BYTES)) != hipSuccess ||
	    (e = hipMemset(base + MSX_GUARD_BYTES + n, MSX_GUARD_FILL, MSX_GUARD_BYTES)) != hipSuccess ||
	    (e = hipDeviceSynchronize()) != hipSuccess) {
		(void)hipFree(base);
		return e;
	}
	*p = base + MSX_GUARD_BYTES;
	std::lock_guard<std::mutex> lk(g_guard_mu);
	g_guard[*p] = guard_rec{base, n};
	return hipSuccess;
}

hipError_t msx_guard_free(void *p) {
	if (!msx_guard_on() || !p) return hipFree(p);
	guard_rec r;
	{
		std::lock_guard<std::mutex> lk(g_guard_mu);
		auto it = g_guard.find(p);
		if (it == g_guard.end()) {                           // (not one of ours: said, and left to the runtime)
			fprintf(stderr, "MSX_GUARD: hipFree of %p, which no guarded allocation returned\n", p);
			return hipFree(p);
		}
		r = it->second;
		g_guard.erase(it);
	}
	(void)hipDeviceSynchronize();                            // whatever still writes into it has finished
	if (guard_damage(p, r, "seen when it was freed")) abort();
	return hipFree(r.base);
}

extern "C" int64_t msx_debug_guard_check(void) {
	if (!msx_guard_on()) return -1;
	(void)hipDeviceSynchronize();
	std::lock_guard<std::mutex> lk(g_guard_mu);
	int64_t bad = 0;
	for (auto &kv : g_guard) bad += (int64_t)guard_damage(kv.first, kv.second, "msx_debug_guard_check");
	return bad;
}


// the guard looking at itself: a kernel writes one byte behind (front = 0) or in front of (front = 1) a 100-byte allocation;
// returns the damaged guard bytes seen (1 expected), -1 if the guard is off.  The allocation is released without the check
// that would abort.
__global__ void k_guard_selftest(unsigned char *p, long at) { p[at] = 0x11; }
extern "C" int64_t msx_debug_guard_selftest(int front) {
	if (!msx_guard_on()) return -1;
	void *p = nullptr;
	if (msx_guard_malloc(&p, 100) != hipSuccess) return -2;
	hipLaunchKernelGGL(k_guard_selftest, dim3(1), dim3(1), 0, 0, (unsigned char *)p, front ? -1L : 100L);
	(void)hipDeviceSynchronize();
	guard_rec r;
	{
		std::lock_guard<std::mutex> lk(g_guard_mu);
		r = g_guard[p];
		g_guard.erase(p);
	}
	const int64_t bad = (int64_t)guard_damage(p, r, "msx_debug_guard_selftest: expected");
	(void)hipFree(r.base);
	return bad;
}
