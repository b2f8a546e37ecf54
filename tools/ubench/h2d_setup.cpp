// What the ingest path pays the runtime for before a byte moves: page-locking, stream creation, and small host-to-device copies issued
// by several threads.   hipcc -O2 h2d_setup.cpp -o _bin/h2d_setup -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	hipSetDevice(0); hipFree(0);
	for (size_t mb : {1, 4, 16, 64, 256}) {
		void *p = nullptr; double t = now(); hipHostMalloc(&p, mb << 20, hipHostMallocDefault); double a = now() - t; t = now(); hipHostFree(p);
		printf("hipHostMalloc %4zu MB: %7.2f ms, free %6.2f ms\n", mb, a, now() - t);
	}
	{ double t = now(); hipStream_t s[8]; for (auto &x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking); printf("8 streams: %.2f ms\n", now() - t); for (auto &x : s) hipStreamDestroy(x); }
	char *dev = nullptr; hipMalloc(&dev, (size_t)2 << 30);
	char *pin = nullptr; hipHostMalloc(&pin, (size_t)256 << 20, hipHostMallocDefault);
	for (int nthreads : {1, 4, 16, 64}) for (size_t kb : {256, 1024, 4096}) {
		const size_t total = (size_t)1280 << 20, per = total / nthreads, blk = kb << 10;
		double t = now();
		std::vector<std::thread> th;
		for (int i = 0; i < nthreads; ++i) th.emplace_back([&, i]() {
			hipSetDevice(0);
			hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
			for (size_t o = 0; o < per; o += blk) { hipMemcpyAsync(dev + (size_t)i * per + o, pin + ((size_t)i * (4 << 20)) % ((size_t)252 << 20), blk, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); }
			hipStreamDestroy(s);
		});
		for (auto &x : th) x.join();
		const double ms = now() - t;
		printf("%2d threads x blocks of %4zu KB, 1.28 GB: %7.2f ms = %5.1f GB/s\n", nthreads, kb, ms, 1.342 / ms * 1e3);
	}
	return 0;
}
