// Issue cost of the VALU instructions the hash loops are made of (gfx950): cycles per wave-instruction on one SIMD, measured with
// enough waves to fill every SIMD (8 per SIMD) and 4 independent chains per lane, so that latency is hidden and only issue remains.
//   hipcc -O2 --offload-arch=gfx950 valu_rates.hip -o _bin/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP 256
#define CHAINS 4
#define BODY(ASM)                                                                                                  \
	for (int it = 0; it < iters; ++it) {                                                                           \
		_Pragma("unroll") for (int r = 0; r < REP / CHAINS; ++r) {                                                 \
			asm volatile(ASM : "+v"(a0), "+v"(x0) : "v"(c), "v"(y), "s"(sc) : "vcc", "s10", "s11");                \
			asm volatile(ASM : "+v"(a1), "+v"(x1) : "v"(c), "v"(y), "s"(sc) : "vcc", "s10", "s11");                \
			asm volatile(ASM : "+v"(a2), "+v"(x2) : "v"(c), "v"(y), "s"(sc) : "vcc", "s10", "s11");                \
			asm volatile(ASM : "+v"(a3), "+v"(x3) : "v"(c), "v"(y), "s"(sc) : "vcc", "s10", "s11");                \
		}                                                                                                          \
	}

template <int WHICH> __global__ __launch_bounds__(256) void k(uint64_t *out, int iters, uint32_t sc)
{
	uint64_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, c = blockIdx.x + 3;
	uint32_t x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, y = blockIdx.x + 7;
	if (WHICH == 0) { BODY("v_xor_b32 %1, %1, %3") }
	if (WHICH == 1) { BODY("v_lshl_add_u64 %0, %0, 3, %2") }
	if (WHICH == 2) { BODY("v_lshlrev_b64 %0, 3, %0") }
	if (WHICH == 3) { BODY("v_mad_u64_u32 %0, s[10:11], %1, %4, %0") }
	if (WHICH == 4) { BODY("v_alignbit_b32 %1, %1, %3, 30") }
	if (WHICH == 5) { BODY("v_cmp_lt_u64 vcc, %0, %2") }
	if (WHICH == 6) { BODY("v_mul_lo_u32 %1, %1, %3") }
	if (WHICH == 7) { BODY("v_and_or_b32 %1, %1, 3, %3") }
	if (WHICH == 8) { BODY("v_lshrrev_b64 %0, 2, %0") }
	if (WHICH == 9) { BODY("v_cndmask_b32 %1, %1, %3, vcc") }
	if (WHICH == 10) { BODY("v_bfe_u32 %1, %1, 4, 2") }
	if (WHICH == 11) { BODY("v_lshl_or_b32 %1, %1, 2, %3") }
	if (WHICH == 12) { BODY("v_mul_hi_u32 %1, %1, %3") }
	if (WHICH == 13) { BODY("v_add_co_u32 %1, vcc, %1, %3") }
	if (WHICH == 14) { BODY("v_pk_add_u16 %1, %1, %3") }
	if (WHICH == 15) { BODY("v_add3_u32 %1, %1, %3, %3") }
	out[(size_t)blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + x0 + x1 + x2 + x3;
}

template <int WHICH> static void run(const char *name, uint64_t *d, int cus, double mhz)
{
	const int iters = 200, blocks = cus * 8;                                      // 8 workgroups of 4 waves per CU: 8 waves per SIMD
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k<WHICH>, dim3(blocks), dim3(256), 0, 0, d, 2, 12345u);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<WHICH>, dim3(blocks), dim3(256), 0, 0, d, iters, 12345u);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms = 0; hipEventElapsedTime(&ms, e0, e1);
	const double per_simd = (double)iters * REP * 8;                               // wave-instructions issued on one SIMD
	printf("%-18s %7.3f ms  %5.2f cycles per wave-instruction (at %.0f MHz)\n", name, ms, ms * 1e-3 * mhz * 1e6 / per_simd, mhz);
}

int main()
{
	hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
	const int cus = pr.multiProcessorCount; const double mhz = pr.clockRate / 1e3;
	uint64_t *d; hipMalloc(&d, (size_t)cus * 8 * 256 * 8);
	printf("%d CUs\n", cus);
	run<0>("v_xor_b32", d, cus, mhz); run<1>("v_lshl_add_u64", d, cus, mhz); run<2>("v_lshlrev_b64", d, cus, mhz); run<3>("v_mad_u64_u32", d, cus, mhz);
	run<4>("v_alignbit_b32", d, cus, mhz); run<5>("v_cmp_lt_u64", d, cus, mhz); run<6>("v_mul_lo_u32", d, cus, mhz); run<7>("v_and_or_b32", d, cus, mhz);
	run<8>("v_lshrrev_b64", d, cus, mhz); run<9>("v_cndmask_b32", d, cus, mhz); run<10>("v_bfe_u32", d, cus, mhz); run<11>("v_lshl_or_b32", d, cus, mhz);
	run<12>("v_mul_hi_u32", d, cus, mhz); run<13>("v_add_co_u32", d, cus, mhz); run<14>("v_pk_add_u16", d, cus, mhz); run<15>("v_add3_u32", d, cus, mhz);
	return 0;
}
