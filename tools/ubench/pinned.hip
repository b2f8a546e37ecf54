// CPU read speed of hipHostMalloc memory against malloc memory: sequential sum and random reads
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void bench(const char *name, uint32_t *p, size_t n)
{
	for (size_t i = 0; i < n; ++i) p[i] = (uint32_t)(i * 2654435761u);
	double t0 = now(); uint64_t s = 0; for (size_t i = 0; i < n; ++i) s += p[i]; double t1 = now();
	uint64_t x = 88172645463325252ull, r = 0;
	for (int i = 0; i < 1000000; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; r += p[x % n]; }
	double t2 = now();
	printf("%-28s sequential %.1f ms (%.1f GB/s)  1M random reads %.1f ms (%.0f ns each)  [%llu]\n", name, t1 - t0, n * 4 / (t1 - t0) / 1e6, t2 - t1, (t2 - t1) * 1e6 / 1e6,
	       (unsigned long long)(s + r));
}
int main()
{
	const size_t n = 16u << 20;
	uint32_t *a = (uint32_t*)malloc(n * 4);
	bench("malloc", a, n);
	uint32_t *b = nullptr;
	(void)hipHostMalloc((void**)&b, n * 4, hipHostMallocDefault);
	bench("hipHostMallocDefault", b, n);
	uint32_t *c = nullptr;
	(void)hipHostMalloc((void**)&c, n * 4, hipHostMallocNonCoherent);
	bench("hipHostMallocNonCoherent", c, n);
	uint32_t *d = (uint32_t*)malloc(n * 4);
	(void)hipHostRegister(d, n * 4, hipHostRegisterDefault);
	bench("malloc + hipHostRegister", d, n);
	return 0;
}
