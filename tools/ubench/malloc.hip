// times hipMalloc / hipFree / hipMemset of large blocks
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
	for (size_t gb : {1, 4, 16, 34}) {
		void *p = nullptr;
		for (int it = 0; it < 3; ++it) {
			double t0 = now(); hipMalloc(&p, gb << 30); double t1 = now();
			hipMemset(p, 0xFF, gb << 30); hipDeviceSynchronize(); double t2 = now();
			hipFree(p); double t3 = now();
			printf("%zu GB: malloc %.2f ms, memset %.2f ms, free %.2f ms\n", gb, t1 - t0, t2 - t1, t3 - t2);
		}
	}
	return 0;
}
