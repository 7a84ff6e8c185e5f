// H2D / D2H rates for page-locked memory obtained in two ways: hipHostMalloc, and malloc + hipHostRegister.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
	const size_t n = (size_t)160 << 20;
	void *d = nullptr, *hm = nullptr;
	hipMalloc(&d, n);
	hipHostMalloc(&hm, n, hipHostMallocDefault);
	void *hr = aligned_alloc(4096, n);
	memset(hr, 1, n); memset(hm, 1, n);
	double t0 = now();
	hipHostRegister(hr, n, hipHostRegisterPortable);
	printf("hipHostRegister of %zu MB: %.1f ms\n", n >> 20, (now() - t0) * 1e3);
	hipStream_t s; hipStreamCreate(&s);
	for (int which = 0; which < 2; which++) {
		void *h = which ? hr : hm;
		for (int dir = 0; dir < 2; dir++) {
			for (int rep = 0; rep < 4; rep++) {
				hipStreamSynchronize(s);
				double t = now();
				if (dir == 0) hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s); else hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s);
				hipStreamSynchronize(s);
				double dt = now() - t;
				if (rep == 3) printf("%s %s: %.1f GB/s\n", which ? "registered" : "hipHostMalloc", dir ? "D2H" : "H2D", n / dt / 1e9);
			}
		}
	}
	// both directions at once on two streams
	hipStream_t s2; hipStreamCreate(&s2);
	void *d2; hipMalloc(&d2, n);
	for (int rep = 0; rep < 4; rep++) {
		hipStreamSynchronize(s); hipStreamSynchronize(s2);
		double t = now();
		hipMemcpyAsync(d, hm, n, hipMemcpyHostToDevice, s);
		hipMemcpyAsync(hr, d2, n, hipMemcpyDeviceToHost, s2);
		hipStreamSynchronize(s); hipStreamSynchronize(s2);
		if (rep == 3) printf("both directions at once: %.1f GB/s each way\n", n / (now() - t) / 1e9);
	}
	return 0;
}
