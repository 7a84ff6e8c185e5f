// Microbenchmark: rate of scattered u32 atomic adds into a 4 MB counter array,
// (a) one array shared by every XCD, (b) one private copy per XCD chosen by the
// hardware XCC_ID, (c) plain scattered 4-byte stores for comparison.
// Build: hipcc -O3 --offload-arch=gfx950 -o atomic_rate atomic_rate.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ __launch_bounds__(256) void k_shared(uint32_t *ui, uint32_t nf, int64_t n) {
	const int64_t stride = (int64_t)gridDim.x * 256;
	for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) atomicAdd(&ui[mix((uint32_t)i) % nf], 2u);
}
__global__ __launch_bounds__(256) void k_private(uint32_t *ui, uint32_t nf, int64_t n) {
	const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
	uint32_t *mine = ui + (size_t)xcc * nf;
	const int64_t stride = (int64_t)gridDim.x * 256;
	for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) atomicAdd(&mine[mix((uint32_t)i) % nf], 2u);
}
__global__ __launch_bounds__(256) void k_store(uint32_t *ui, uint32_t nf, int64_t n) {
	const int64_t stride = (int64_t)gridDim.x * 256;
	for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) ui[mix((uint32_t)i) % nf] = (uint32_t)i;
}
__global__ void k_xcc(uint32_t *out) { if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u; }
int main() {
	const uint32_t nf = 1u << 20;
	const int64_t n = 20000000;
	uint32_t *ui; CK(hipMalloc(&ui, (size_t)8 * nf * 4));
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	uint32_t *h = (uint32_t *)malloc((size_t)8 * nf * 4);
	for (int grid : {2048, 8192}) {
		for (int which = 0; which < 3; which++) {
			float best = 1e9;
			for (int rep = 0; rep < 4; rep++) {
				CK(hipMemset(ui, 0, (size_t)8 * nf * 4));
				CK(hipEventRecord(a));
				if (which == 0) hipLaunchKernelGGL(k_shared, dim3(grid), dim3(256), 0, 0, ui, nf, n);
				if (which == 1) hipLaunchKernelGGL(k_private, dim3(grid), dim3(256), 0, 0, ui, nf, n);
				if (which == 2) hipLaunchKernelGGL(k_store, dim3(grid), dim3(256), 0, 0, ui, nf, n);
				CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
				float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
			}
			CK(hipMemcpy(h, ui, (size_t)8 * nf * 4, hipMemcpyDeviceToHost));
			uint64_t tot = 0; for (size_t i = 0; i < (size_t)8 * nf; i++) tot += h[i];
			printf("grid %d %s: %.3f ms  (%.1f G adds/s)  sum=%llu expect=%lld\n", grid,
			       which == 0 ? "shared " : which == 1 ? "private" : "store  ", best, n / best / 1e6,
			       (unsigned long long)tot, (long long)(which == 2 ? 0 : 2 * n));
		}
	}
	uint32_t *x; CK(hipMalloc(&x, 64 * 4));
	hipLaunchKernelGGL(k_xcc, dim3(32), dim3(64), 0, 0, x);
	uint32_t hx[32]; CK(hipMemcpy(hx, x, 32 * 4, hipMemcpyDeviceToHost));
	printf("xcc of blocks 0..31:"); for (int i = 0; i < 32; i++) printf(" %u", hx[i]); printf("\n");
	return 0;
}
