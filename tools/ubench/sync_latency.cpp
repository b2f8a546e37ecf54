// What waiting for the stream costs per round trip: a kernel that stores its result into pinned memory, then
//   (a) hipStreamSynchronize                      (what mcom_stream_sync does)
//   (b) the host spins on the word the kernel itself stored (a sequence number: no runtime call at all)
//   (c) hipStreamWriteValue32 of a sequence number behind the kernel, the host spins on that word
//   (d) an event recorded behind the kernel, hipEventQuery spun
// and the same with a second, dependent kernel launched right after the wait (the gap the GPU sees between the two is what counts).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
__global__ void k_set(volatile unsigned long long *p, unsigned long long v, unsigned long long *d) { *d = v; __threadfence_system(); *p = v; }
int main()
{
	unsigned long long *d = nullptr; volatile unsigned long long *pin = nullptr; unsigned int *flag = nullptr;
	hipMalloc(&d, 64); hipHostMalloc((void**)&pin, 64, hipHostMallocDefault); hipHostMalloc((void**)&flag, 64, hipHostMallocDefault);
	pin[0] = 0; flag[0] = 0;
	hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
	hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
	const int N = 3000;
	for (int mode = 0; mode < 4; ++mode) {
		for (int warm = 0; warm < 2; ++warm) {
			unsigned long long seq = (unsigned long long)(mode * 2 + warm) << 32;
			auto t0 = std::chrono::steady_clock::now();
			bool ok = true;
			for (int i = 0; i < N && ok; ++i) {
				++seq;
				hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, st, pin, seq, d);
				if (mode == 0) hipStreamSynchronize(st);
				else if (mode == 1) { while (pin[0] != seq) {} }
				else if (mode == 2) {
					if (hipStreamWriteValue32(st, flag, (unsigned int)seq, 0) != hipSuccess) { printf("hipStreamWriteValue32 failed\n"); ok = false; break; }
					while (*(volatile unsigned int*)flag != (unsigned int)seq) {}
				} else { hipEventRecord(ev, st); while (hipEventQuery(ev) == hipErrorNotReady) {} }
			}
			auto t1 = std::chrono::steady_clock::now();
			if (warm && ok) printf("%s: %.2f us per round trip\n", mode == 0 ? "hipStreamSynchronize" : mode == 1 ? "spin on the kernel's own store" : mode == 2 ? "hipStreamWriteValue32 + spin" : "hipEventQuery spin",
			                       std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
			hipStreamSynchronize(st);
		}
	}
	return 0;
}
