// how long page-locking takes: hipHostRegister on malloc'ed / THP-advised / pre-touched memory, hipHostMalloc
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
	hipFree(nullptr);
	const size_t n = 80u << 20;
	for (int rep = 0; rep < 2; rep++) {
		{ void *p = nullptr; double t = now(); posix_memalign(&p, 4096, n); double t1 = now(); hipHostRegister(p, n, hipHostRegisterDefault); double t2 = now();
		  printf("untouched 4K pages: register %.1f ms\n", (t2 - t1) * 1e3); hipHostUnregister(p); free(p); (void)t; }
		{ void *p = nullptr; posix_memalign(&p, 4096, n); double t0 = now(); memset(p, 1, n); double t1 = now(); hipHostRegister(p, n, hipHostRegisterDefault); double t2 = now();
		  printf("touched 4K pages: touch %.1f ms, register %.1f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3); hipHostUnregister(p); free(p); }
		{ void *p = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); madvise(p, n, MADV_HUGEPAGE); double t0 = now(); memset(p, 1, n); double t1 = now();
		  hipHostRegister(p, n, hipHostRegisterDefault); double t2 = now();
		  printf("THP-advised, touched: touch %.1f ms, register %.1f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3); hipHostUnregister(p); munmap(p, n); }
		{ void *p = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); madvise(p, n, MADV_HUGEPAGE); double t1 = now();
		  hipHostRegister(p, n, hipHostRegisterDefault); double t2 = now();
		  printf("THP-advised, untouched: register %.1f ms\n", (t2 - t1) * 1e3); hipHostUnregister(p); munmap(p, n); }
		{ void *p = nullptr; double t1 = now(); hipHostMalloc(&p, n, hipHostMallocDefault); double t2 = now(); printf("hipHostMalloc %.1f ms\n", (t2 - t1) * 1e3); hipHostFree(p); }
	}
	FILE *f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r"); char b[128] = {0}; if (f) { fgets(b, 127, f); fclose(f); } printf("THP: %s", b);
	return 0;
}
