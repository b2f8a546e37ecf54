// What one "small result back to the host" costs: kernel + 8-byte D2H + stream sync, pageable against pinned destination.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_set(unsigned long long *p, unsigned long long v) { *p = v; }
__global__ void k_copy8(unsigned long long *dst, const unsigned long long *src, int n) { if ((int)threadIdx.x < n) dst[threadIdx.x] = src[threadIdx.x]; }
int main()
{
	unsigned long long *d = nullptr, *pin = nullptr, pageable = 0;
	hipMalloc(&d, 64); hipHostMalloc(&pin, 64, hipHostMallocDefault);
	hipStream_t st; hipStreamCreate(&st);
	for (int mode = 0; mode < 5; ++mode) {
		for (int warm = 0; warm < 2; ++warm) {
			auto t0 = std::chrono::steady_clock::now();
			const int N = 2000;
			for (int i = 0; i < N; ++i) {
				hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, st, d, (unsigned long long)i);
				if (mode == 0) hipMemcpyAsync(&pageable, d, 8, hipMemcpyDeviceToHost, st);
				else if (mode == 1) hipMemcpyAsync(pin, d, 8, hipMemcpyDeviceToHost, st);
				else if (mode == 3) hipLaunchKernelGGL(k_copy8, dim3(1), dim3(64), 0, st, pin, (const unsigned long long*)d, 1);   // a kernel stores into the pinned page
				else if (mode == 4) hipMemsetAsync(d + 1, 0, 8, st);
				hipStreamSynchronize(st);
			}
			auto t1 = std::chrono::steady_clock::now();
			if (warm) printf("%s: %.2f us per round trip\n", mode == 0 ? "pageable" : mode == 1 ? "pinned" : mode == 2 ? "sync only" : mode == 3 ? "copy kernel into the pinned page" : "kernel + hipMemsetAsync", std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
		}
	}
	return 0;
}
