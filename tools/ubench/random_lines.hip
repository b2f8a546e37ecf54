// What the card delivers for the Stage-2 lookups' access pattern: every lane reads ONE 64-byte line at a random place of a table
// (eight 8-byte loads, as k_realign_reads takes its index line), nothing depends on anything, 256 threads per workgroup.  Table sizes
// from one that sits in the Infinity Cache to the 16.8 GB of the benchmark's contig index.
//   hipcc -O2 --offload-arch=gfx950 random_lines.hip -o _bin/random_lines && _bin/random_lines
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void k_lines(const unsigned long long *__restrict__ table, uint64_t n_lines, uint64_t n_lookups, unsigned long long *__restrict__ sink)
{
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n_lookups) return;
	uint64_t z = (t + 1) * 0x9E3779B97F4A7C15ull;                           // splitmix64: the line of lookup t
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
	const unsigned long long *l = table + (z % n_lines) * 8;
	unsigned long long a = 0;
#pragma unroll
	for (int s = 0; s < 8; ++s) a ^= l[s];
	if (a == 0x123456789ull) sink[0] = a;                                     // (never: keeps the loads)
}

int main()
{
	const uint64_t n_lookups = 200000000ull;
	const double sizes_gb[] = {0.032, 0.25, 2.0, 8.0, 16.8};
	unsigned long long *sink = nullptr; hipMalloc(&sink, 8);
	for (double gb : sizes_gb) {
		const uint64_t n_lines = (uint64_t)(gb * 1e9 / 64);
		unsigned long long *table = nullptr;
		if (hipMalloc(&table, n_lines * 64) != hipSuccess) { printf("%.3f GB: no memory\n", gb); continue; }
		hipMemset(table, 1, n_lines * 64);
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		const unsigned blocks = (unsigned)((n_lookups + 255) / 256);
		hipLaunchKernelGGL(k_lines, dim3(blocks), dim3(256), 0, 0, table, n_lines, n_lookups, sink);   // warm-up
		hipEventRecord(e0, 0);
		hipLaunchKernelGGL(k_lines, dim3(blocks), dim3(256), 0, 0, table, n_lines, n_lookups, sink);
		hipEventRecord(e1, 0); hipEventSynchronize(e1);
		float ms = 0; hipEventElapsedTime(&ms, e0, e1);
		printf("table %7.3f GB: %6.2f ms for %llu M random 64-byte lines = %5.1f G lines/s = %5.2f TB/s\n", gb, ms, (unsigned long long)(n_lookups / 1000000), n_lookups / ms / 1e6, n_lookups * 64.0 / ms / 1e9);
		hipFree(table);
	}
	return 0;
}
