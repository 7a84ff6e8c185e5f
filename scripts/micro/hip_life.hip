// How long does the kernel take to let go of a process, as a function of what the process asked the HIP runtime for and of how long it
// lived?  hipcc --offload-arch=gfx950 -o /tmp/hip_life scripts/micro/hip_life.hip ; /tmp/hip_life <ms to live> <what>
//   what: 0 no HIP at all, 1 hipFree(0), 2 + a kernel on the null stream, 3 + four streams with a kernel each, 4 + hipHostMalloc 64 MB,
//         5 + hipMalloc 1 GB, 6 + hipHostRegister of 64 MB, 7 + events recorded and waited for across streams
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <unistd.h>
static double now() { timespec t; clock_gettime(CLOCK_REALTIME, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
__global__ void k_touch(int *p) { if (p) p[threadIdx.x] = 1; }
int main(int argc, char **argv) {
	const double t0 = now();
	const int ms = argc > 1 ? atoi(argv[1]) : 0, what = argc > 2 ? atoi(argv[2]) : 1;
	int *d = nullptr;
	if (what >= 1 && (hipSetDevice(0) != hipSuccess || hipFree(nullptr) != hipSuccess)) { fprintf(stderr, "no device\n"); return 1; }
	if (what >= 2) { (void)hipMalloc(&d, 4096); hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, 0, d); (void)hipDeviceSynchronize(); }
	hipStream_t st[4] = {nullptr, nullptr, nullptr, nullptr};
	if (what >= 3) for (int i = 0; i < 4; i++) { (void)hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking); hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, st[i], d); }
	if (what >= 3) (void)hipDeviceSynchronize();
	if (what >= 4) { void *h = nullptr; (void)hipHostMalloc(&h, 64 << 20, hipHostMallocDefault); }
	if (what >= 5) { void *g = nullptr; (void)hipMalloc(&g, (size_t)1 << 30); (void)hipMemset(g, 0, (size_t)1 << 30); }
	if (what >= 6) { void *m = aligned_alloc(1 << 21, 64 << 20); for (size_t i = 0; i < (64u << 20); i += 4096) ((char *)m)[i] = 1; (void)hipHostRegister(m, 64 << 20, hipHostRegisterPortable); }
	if (what >= 7) { hipEvent_t e; (void)hipEventCreateWithFlags(&e, hipEventDisableTiming); (void)hipEventRecord(e, st[0]); (void)hipStreamWaitEvent(st[1], e, 0); (void)hipDeviceSynchronize(); }
	const double t1 = now();
	usleep(1000u * (unsigned)ms);
	printf("%.6f set_up_after %.3f lived %.3f\n", now(), t1 - t0, now() - t0);      // (absolute time of the last line: the caller subtracts)
	fflush(stdout);
	_exit(0);
}
