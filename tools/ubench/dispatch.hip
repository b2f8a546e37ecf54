// micro-benchmark: cost of dispatching many tiny 64-thread workgroups on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ __launch_bounds__(64) void k_empty(uint32_t *out) { if (threadIdx.x == 0 && blockIdx.x == 0xFFFFFFFFu) out[0] = 1; }
__global__ __launch_bounds__(64) void k_oneload(const uint64_t *off, uint32_t *out) { uint64_t a = off[blockIdx.x], b = off[blockIdx.x + 1]; if (threadIdx.x == 0) out[blockIdx.x] = (uint32_t)(b - a); }
__global__ __launch_bounds__(64) void k_twoload(const uint64_t *off, const uint8_t *seq, uint32_t *out) { uint64_t a = off[blockIdx.x]; uint8_t c = seq[a + threadIdx.x]; __shared__ uint8_t sb[64]; sb[threadIdx.x] = c; __syncthreads(); if (threadIdx.x == 0) out[blockIdx.x] = sb[63]; }
template <int LDSB> __global__ __launch_bounds__(64) void k_lds(const uint64_t *off, uint32_t *out) { __shared__ uint8_t sb[LDSB]; sb[threadIdx.x] = (uint8_t)off[blockIdx.x]; __syncthreads(); if (threadIdx.x == 0) out[blockIdx.x] = sb[63]; }
int main() {
	const size_t n = 8000000;
	uint64_t *off; uint8_t *seq; uint32_t *out;
	hipMalloc(&off, (n + 1) * 8); hipMalloc(&seq, n * 144 + 64); hipMalloc(&out, n * 4);
	hipMemset(seq, 65, n * 144 + 64);
	uint64_t *h = (uint64_t*)malloc((n + 1) * 8); for (size_t i = 0; i <= n; ++i) h[i] = i * 144; hipMemcpy(off, h, (n + 1) * 8, hipMemcpyHostToDevice);
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); float ms;
#define T(name, launch) for (int r = 0; r < 3; ++r) { hipEventRecord(a); launch; hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); } printf("%-28s %8.3f ms\n", name, ms);
	T("empty 8M x64", hipLaunchKernelGGL(k_empty, dim3(n), dim3(64), 0, 0, out));
	T("one dependent load", hipLaunchKernelGGL(k_oneload, dim3(n), dim3(64), 0, 0, off, out));
	T("two dependent loads + sync", hipLaunchKernelGGL(k_twoload, dim3(n), dim3(64), 0, 0, off, seq, out));
	T("lds 4KB", hipLaunchKernelGGL((k_lds<4096>), dim3(n), dim3(64), 0, 0, off, out));
	T("lds 11KB", hipLaunchKernelGGL((k_lds<11264>), dim3(n), dim3(64), 0, 0, off, out));
	T("empty 2M x256", hipLaunchKernelGGL(k_empty, dim3(n / 4), dim3(256), 0, 0, out));
	return 0;
}
