// minicom_amd/csrc/reads.hip -- read ingest kernels for gfx950 (MI355X).
//
//   k_classify_pack : ASCII reads -> class, N count, N mask, 2-bit packed rows (N substituted)
//                     restates process_reads up to the sketch call (reference kthread_reads.c:55-205)
//   k_sketch_reads  : one minimizer per read from packed rows = mm_sketch_two (reference sketch.c:238-289)
//   k_synth_reads   : counter-based synthetic reads (minicom_amd/synth.py)
//
// Both hot kernels are integer/byte work: no MFMA.  k_classify_pack is HBM-bound (L + 8W + 4 bytes per
// read), k_sketch_reads is VALU-bound (one 2k-bit invertible hash per base), see DESIGN.md.
#include "mcom_dev.hpp"

// ------------------------------------------------------------------------------------------------
// k_classify_pack: G lanes cooperate on one read, 8 bases per lane (G = 16 for L <= 128, else 32), so a
// 64-wide wave holds 64/G reads and every global access of a wave is one contiguous span.
// ------------------------------------------------------------------------------------------------
// Cross-lane exchanges inside a row of 16 lanes as DPP modifiers of ordinary VALU moves (no trip through the LDS crossbar, which is
// what __shfl_xor compiles to: ds_bpermute_b32, ~100 cycles each, 18 of them per iteration of the classify kernel).  A sum or an
// OR over the row only needs every step to pair disjoint halves: lane ^ 1, lane ^ 2 (quad permutes), then the mirror of the half
// row and of the row (all lanes of a quad / of a half row already agree).
template <int CTRL> __device__ __forceinline__ uint32_t dpp_u32(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
template <int CTRL> __device__ __forceinline__ uint64_t dpp_u64(uint64_t v)
{
	return (uint64_t)dpp_u32<CTRL>((uint32_t)v) | ((uint64_t)dpp_u32<CTRL>((uint32_t)(v >> 32)) << 32);
}
#define DPP_XOR1 0xB1        /* quad_perm [1,0,3,2] */
#define DPP_XOR2 0x4E        /* quad_perm [2,3,0,1] */
#define DPP_HALF_MIRROR 0x141
#define DPP_MIRROR 0x140
__device__ __forceinline__ uint32_t row16_sum(uint32_t v)
{
	v += dpp_u32<DPP_XOR1>(v); v += dpp_u32<DPP_XOR2>(v); v += dpp_u32<DPP_HALF_MIRROR>(v); v += dpp_u32<DPP_MIRROR>(v);
	return v;
}

__device__ __forceinline__ uint32_t load_u32_any(const uint8_t *p)
{
	uint32_t v;
	__builtin_memcpy(&v, p, 4);
	return v;
}

// 4 ASCII bases in a dword -> 4 x 2-bit codes in the low byte (A0 C1 G2 T3; N and anything else -> junk)
__device__ __forceinline__ uint32_t ascii4_to_codes(uint32_t d)
{
	uint32_t t = ((d >> 1) ^ (d >> 2)) & 0x03030303u;
	return (t | (t >> 6) | (t >> 12) | (t >> 18)) & 0xFFu;
}
// bit j set when byte j of d is 'N'/'n' (bit 3 of the byte distinguishes N from A,C,G,T)
__device__ __forceinline__ uint32_t ascii4_nbits(uint32_t d)
{
	uint32_t t = (d >> 3) & 0x01010101u;
	return (t | (t >> 7) | (t >> 14) | (t >> 21)) & 0xFu;
}

template <int G>
__global__ __launch_bounds__(256) void k_classify_pack(const uint8_t *__restrict__ ascii, size_t pitch, size_t n,
                                                       int L, int e, uint64_t *__restrict__ packed, int W,
                                                       uint8_t *__restrict__ cls, uint16_t *__restrict__ ncnt,
                                                       uint64_t *__restrict__ nmask, int NW)
{
	constexpr int RPW = 64 / G;                       // reads per wave
	const int lane = threadIdx.x & 63;
	const int sub = lane / G, j = lane % G;
	const size_t wave0 = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	const size_t nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
	const size_t total_bytes = n ? (n - 1) * pitch + (size_t)L : 0;

	for (size_t base = wave0 * RPW; base < n; base += nwaves * RPW) {
		const size_t r = base + sub;
		const bool live = r < n;
		const int c0 = 8 * j;                         // first base of this lane
		int nv = L - c0; nv = nv < 0 ? 0 : (nv > 8 ? 8 : nv);
		if (!live) nv = 0;
		uint32_t d0 = 0, d1 = 0;
		if (nv > 0) {
			const size_t off = r * pitch + (size_t)c0;
			if (off + 8 <= total_bytes) { d0 = load_u32_any(ascii + off); d1 = load_u32_any(ascii + off + 4); }
			else {
				for (int q = 0; q < nv; ++q) {
					uint32_t b = ascii[off + q];
					if (q < 4) d0 |= b << (8 * q); else d1 |= b << (8 * (q - 4));
				}
			}
		}
		const uint32_t vmask = nv >= 8 ? 0xFFu : ((1u << nv) - 1u);          // valid bases of this lane
		uint32_t codes = (ascii4_to_codes(d0) | (ascii4_to_codes(d1) << 8)); // 8 x 2 bits
		uint32_t nb = (ascii4_nbits(d0) | (ascii4_nbits(d1) << 4)) & vmask;  // 8 N flags
		// per-lane base counts among valid non-N bases: 2-bit fields -> one-hot tests
		uint32_t lo = codes & 0x5555u, hi = (codes >> 1) & 0x5555u;          // low/high bit of each code, at even bits
		uint32_t ok = 0;                                                     // valid & not N, spread to even bits
		{
			uint32_t m = vmask & ~nb;                                        // 8 bits
			m = (m | (m << 4)) & 0x0F0Fu; m = (m | (m << 2)) & 0x3333u; m = (m | (m << 1)) & 0x5555u;
			ok = m;
		}
		uint32_t cA = __popc(~lo & ~hi & ok), cC = __popc(lo & ~hi & ok), cG = __popc(~lo & hi & ok), cT = __popc(lo & hi & ok);
		uint32_t acc0 = cA | (cT << 16), acc1 = cG | (cC << 16), accN = __popc(nb);
		if (G == 16) { acc0 = row16_sum(acc0); acc1 = row16_sum(acc1); accN = row16_sum(accN); }   // one DPP row
		else {
#pragma unroll
			for (int s = 1; s < G; s <<= 1) { acc0 += __shfl_xor(acc0, s, 64); acc1 += __shfl_xor(acc1, s, 64); accN += __shfl_xor(accN, s, 64); }
		}
		const int nA = acc0 & 0xFFFF, nT = acc0 >> 16, nG = acc1 & 0xFFFF, nC = acc1 >> 16, nN = (int)accN;
		int c;                                                               // kthread_reads.c:84-224
		if (nA == L) c = MCOM_CLS_ALLA;
		else if (nT == L) c = MCOM_CLS_ALLT;
		else if (nN == L) c = MCOM_CLS_ALLN;
		else if (nT + nG + nC + nN <= e) c = MCOM_CLS_NEARA;
		else if (nA + nG + nC + nN <= e) c = MCOM_CLS_NEART;
		else if (nA + nT + nG + nC <= e) c = MCOM_CLS_NEARN;
		else if (!((double)nN <= 0.4 * (double)L)) c = MCOM_CLS_NHEAVY;
		else c = MCOM_CLS_SKETCH;
		uint32_t rep = 0;                                                    // majority base, ties A,T,G,C (:185-201)
		if (c == MCOM_CLS_SKETCH && nN > 0) {
			int mx = nA; if (nT > mx) mx = nT; if (nG > mx) mx = nG; if (nC > mx) mx = nC;
			rep = (mx == nA) ? 0u : (mx == nT) ? 3u : (mx == nG) ? 2u : 1u;
		}
		// substitute: N positions take rep, invalid positions take 0
		uint32_t nsp = nb; nsp = (nsp | (nsp << 4)) & 0x0F0Fu; nsp = (nsp | (nsp << 2)) & 0x3333u; nsp = (nsp | (nsp << 1)) & 0x5555u;
		uint32_t keep = ok | (ok << 1);
		uint32_t fill = (rep & 1 ? nsp : 0u) | (rep & 2 ? (nsp << 1) : 0u);
		uint32_t out16 = (codes & keep) | fill;
		// four lanes make one 64-bit word
		uint64_t word = (uint64_t)out16 << (16 * (j & 3));
		word |= dpp_u64<DPP_XOR1>(word);
		word |= dpp_u64<DPP_XOR2>(word);
		if (live && (j & 3) == 0 && (j >> 2) < W) packed[r * (size_t)W + (j >> 2)] = word;
		if (nmask) {
			uint64_t nw = (uint64_t)nb << (8 * (j & 7));
			nw |= dpp_u64<DPP_XOR1>(nw); nw |= dpp_u64<DPP_XOR2>(nw); nw |= dpp_u64<DPP_HALF_MIRROR>(nw);   // (an OR over eight lanes: the quads already agree)
			if (live && (j & 7) == 0 && (j >> 3) < NW) nmask[r * (size_t)NW + (j >> 3)] = nw;
		}
		if (live && j == 0) { cls[r] = (uint8_t)c; ncnt[r] = (uint16_t)nN; }
	}
}

// The same with SIXTEEN bases per lane, for 128 < L <= 256: 16 lanes per read (four reads per wave instead of two with
// eight bases on 32 lanes, of which L = 150 keeps only 19 busy), one 16-byte load per lane, two lanes per packed word.
__device__ __forceinline__ uint32_t spread16(uint32_t m)                 // bit i of the low 16 -> bit 2i
{
	m = (m | (m << 8)) & 0x00FF00FFu; m = (m | (m << 4)) & 0x0F0F0F0Fu; m = (m | (m << 2)) & 0x33333333u; m = (m | (m << 1)) & 0x55555555u;
	return m;
}
__global__ __launch_bounds__(256) void k_classify_pack16(const uint8_t *__restrict__ ascii, size_t pitch, size_t n,
                                                         int L, int e, uint64_t *__restrict__ packed, int W,
                                                         uint8_t *__restrict__ cls, uint16_t *__restrict__ ncnt,
                                                         uint64_t *__restrict__ nmask, int NW)
{
	constexpr int G = 16, RPW = 64 / G;
	const int lane = threadIdx.x & 63;
	const int sub = lane / G, j = lane % G;
	const size_t wave0 = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	const size_t nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
	const size_t total_bytes = n ? (n - 1) * pitch + (size_t)L : 0;
	for (size_t base = wave0 * RPW; base < n; base += nwaves * RPW) {
		const size_t r = base + sub;
		const bool live = r < n;
		const int c0 = 16 * j;
		int nv = L - c0; nv = nv < 0 ? 0 : (nv > 16 ? 16 : nv);
		if (!live) nv = 0;
		uint32_t d[4] = {0, 0, 0, 0};
		if (nv > 0) {
			const size_t off = r * pitch + (size_t)c0;
			if (off + 16 <= total_bytes) { d[0] = load_u32_any(ascii + off); d[1] = load_u32_any(ascii + off + 4); d[2] = load_u32_any(ascii + off + 8); d[3] = load_u32_any(ascii + off + 12); }
			else for (int q = 0; q < nv; ++q) d[q >> 2] |= (uint32_t)ascii[off + q] << (8 * (q & 3));
		}
		const uint32_t vmask = nv >= 16 ? 0xFFFFu : ((1u << nv) - 1u);
		const uint32_t codes = ascii4_to_codes(d[0]) | (ascii4_to_codes(d[1]) << 8) | (ascii4_to_codes(d[2]) << 16) | (ascii4_to_codes(d[3]) << 24);   // 16 x 2 bits
		const uint32_t nb = (ascii4_nbits(d[0]) | (ascii4_nbits(d[1]) << 4) | (ascii4_nbits(d[2]) << 8) | (ascii4_nbits(d[3]) << 12)) & vmask;       // 16 N flags
		const uint32_t lo = codes & 0x55555555u, hi = (codes >> 1) & 0x55555555u;
		const uint32_t ok = spread16(vmask & ~nb);
		const uint32_t cA = __popc(~lo & ~hi & ok), cC = __popc(lo & ~hi & ok), cG = __popc(~lo & hi & ok), cT = __popc(lo & hi & ok);
		uint32_t acc0 = cA | (cT << 16), acc1 = cG | (cC << 16), accN = __popc(nb);
		acc0 = row16_sum(acc0); acc1 = row16_sum(acc1); accN = row16_sum(accN);   // G = 16 lanes = one DPP row
		const int nA = acc0 & 0xFFFF, nT = acc0 >> 16, nG = acc1 & 0xFFFF, nC = acc1 >> 16, nN = (int)accN;
		int c;                                                               // kthread_reads.c:84-224
		if (nA == L) c = MCOM_CLS_ALLA;
		else if (nT == L) c = MCOM_CLS_ALLT;
		else if (nN == L) c = MCOM_CLS_ALLN;
		else if (nT + nG + nC + nN <= e) c = MCOM_CLS_NEARA;
		else if (nA + nG + nC + nN <= e) c = MCOM_CLS_NEART;
		else if (nA + nT + nG + nC <= e) c = MCOM_CLS_NEARN;
		else if (!((double)nN <= 0.4 * (double)L)) c = MCOM_CLS_NHEAVY;
		else c = MCOM_CLS_SKETCH;
		uint32_t rep = 0;                                                    // majority base, ties A,T,G,C (:185-201)
		if (c == MCOM_CLS_SKETCH && nN > 0) {
			int mx = nA; if (nT > mx) mx = nT; if (nG > mx) mx = nG; if (nC > mx) mx = nC;
			rep = (mx == nA) ? 0u : (mx == nT) ? 3u : (mx == nG) ? 2u : 1u;
		}
		const uint32_t nsp = spread16(nb);
		const uint32_t keep = ok | (ok << 1);
		const uint32_t fill = (rep & 1 ? nsp : 0u) | (rep & 2 ? (nsp << 1) : 0u);
		const uint32_t out32 = (codes & keep) | fill;
		uint64_t word = (uint64_t)out32 << (32 * (j & 1));                   // two lanes make one 64-bit word
		word |= dpp_u64<DPP_XOR1>(word);
		if (live && (j & 1) == 0 && (j >> 1) < W) packed[r * (size_t)W + (j >> 1)] = word;
		if (nmask) {
			uint64_t nw = (uint64_t)nb << (16 * (j & 3));                    // four lanes make one word of N flags
			nw |= dpp_u64<DPP_XOR1>(nw); nw |= dpp_u64<DPP_XOR2>(nw);
			if (live && (j & 3) == 0 && (j >> 2) < NW) nmask[r * (size_t)NW + (j >> 2)] = nw;
		}
		if (live && j == 0) { cls[r] = (uint8_t)c; ncnt[r] = (uint16_t)nN; }
	}
}

// ------------------------------------------------------------------------------------------------
// k_classify_flat (round 3): the same result for reads that lie back to back (pitch == L), as a byte stream.
// The kernels above give a read sixteen lanes: at L = 150 ten of them have bases, every load is an unaligned dword at pitch 150,
// and the counts and the words are put together across lanes -- 37 wave instructions per read, issue-bound at 2.1 TB/s.  Here a
// wave takes 64 consecutive reads = 64 L bytes (a multiple of 16 whatever L is): phase A loads them as aligned 16-byte units,
// lane after lane, turns every unit into 32 bits of codes and 16 N flags and lays them down in LDS as two flat bit streams;
// phase B gives every lane one read: its W words are cut out of the code stream at bit 2 L r (and its N words out of the flag
// stream at bit L r), counted with popcounts, classified, the N positions filled with the majority base, and written -- no
// cross-lane step at all, every lane busy in both phases: ~9 wave instructions per read, which leaves the kernel to HBM.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t spread32_64(uint64_t x)             // bit i of the low 32 bits -> bit 2i
{
	x &= 0xFFFFFFFFull;
	x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
	x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
	x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
	x = (x | (x << 2)) & 0x3333333333333333ull;
	x = (x | (x << 1)) & 0x5555555555555555ull;
	return x;
}
// 64 bits of a flat stream of 32-bit words from bit `bit` on (the stream has two words of padding behind its end)
__device__ __forceinline__ uint64_t stream64(const uint32_t *w, uint32_t bit)
{
	const uint32_t i = bit >> 5, sh = bit & 31u;
	const uint64_t lo = (uint64_t)w[i] | ((uint64_t)w[i + 1] << 32);
	return sh ? (lo >> sh) | ((uint64_t)w[i + 2] << (64 - sh)) : lo;
}
// one read as W words of codes and NW words of N flags: counts, class, N positions filled with the majority base, written
// (kthread_reads.c:55-224) -- the second phase of k_classify_flat, and all of k_classify_packed
template <int W>
__device__ __forceinline__ void cf_classify_words(const uint64_t (&cw)[W], const uint64_t (&nw)[(W + 1) / 2], int L, int e, size_t r, uint64_t *packed,
                                                  uint8_t *__restrict__ cls, uint16_t *__restrict__ ncnt, uint64_t *nmask)
{
	constexpr int NW = (W + 1) / 2;
	int nA = 0, nC = 0, nG = 0, nT = 0, nN = 0;
	uint64_t okm[W];
#pragma unroll
	for (int w = 0; w < W; ++w) {
		const int left = 2 * L - 64 * w;                              // code bits of this word
		const uint64_t valid = left >= 64 ? ~0ull : (left > 0 ? (1ull << left) - 1 : 0ull);
		const uint64_t nsp = spread32_64(nw[w >> 1] >> (32 * (w & 1)));
		okm[w] = ~nsp & 0x5555555555555555ull & valid;
		const uint64_t lo = cw[w] & 0x5555555555555555ull, hi = (cw[w] >> 1) & 0x5555555555555555ull;
		nA += __popcll(~lo & ~hi & okm[w]); nC += __popcll(lo & ~hi & okm[w]); nG += __popcll(~lo & hi & okm[w]); nT += __popcll(lo & hi & okm[w]);
	}
#pragma unroll
	for (int w = 0; w < NW; ++w) nN += __popcll(nw[w]);
	int c;                                                           // kthread_reads.c:84-224
	if (nA == L) c = MCOM_CLS_ALLA;
	else if (nT == L) c = MCOM_CLS_ALLT;
	else if (nN == L) c = MCOM_CLS_ALLN;
	else if (nT + nG + nC + nN <= e) c = MCOM_CLS_NEARA;
	else if (nA + nG + nC + nN <= e) c = MCOM_CLS_NEART;
	else if (nA + nT + nG + nC <= e) c = MCOM_CLS_NEARN;
	else if (!((double)nN <= 0.4 * (double)L)) c = MCOM_CLS_NHEAVY;
	else c = MCOM_CLS_SKETCH;
	uint32_t rep = 0;                                                // majority base, ties A,T,G,C (:185-201)
	if (c == MCOM_CLS_SKETCH && nN > 0) {
		int mx = nA; if (nT > mx) mx = nT; if (nG > mx) mx = nG; if (nC > mx) mx = nC;
		rep = (mx == nA) ? 0u : (mx == nT) ? 3u : (mx == nG) ? 2u : 1u;
	}
#pragma unroll
	for (int w = 0; w < W; ++w) {
		const int left = 2 * L - 64 * w;
		const uint64_t valid = left >= 64 ? ~0ull : (left > 0 ? (1ull << left) - 1 : 0ull);
		const uint64_t nsp = spread32_64(nw[w >> 1] >> (32 * (w & 1))) & valid;
		const uint64_t keep = okm[w] | (okm[w] << 1);
		const uint64_t fill = ((rep & 1) ? nsp : 0ull) | ((rep & 2) ? (nsp << 1) : 0ull);
		packed[r * (size_t)W + w] = (cw[w] & keep) | fill;
	}
	if (nmask) {
#pragma unroll
		for (int w = 0; w < NW; ++w) nmask[r * (size_t)NW + w] = nw[w];
	}
	cls[r] = (uint8_t)c; ncnt[r] = (uint16_t)nN;
}
template <int W>
__global__ __launch_bounds__(256) void k_classify_flat(const uint8_t *__restrict__ ascii, size_t n, int L, int e, uint64_t *__restrict__ packed,
                                                       uint8_t *__restrict__ cls, uint16_t *__restrict__ ncnt, uint64_t *__restrict__ nmask)
{
	constexpr int NW = (W + 1) / 2;
	extern __shared__ uint32_t cf_lds[];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const uint32_t units = 4u * (uint32_t)L;                               // 16-byte units of 64 reads
	uint32_t *codes = cf_lds + (size_t)wv * (units + units / 2 + 8);       // [units + 2]: 32 bits of codes per unit
	uint32_t *nbw = codes + units + 4;                                     // [units / 2 + 2]: 16 N flags per unit, two units per word
	uint16_t *nbh = (uint16_t*)nbw;
	const size_t total = n * (size_t)L;
	const size_t nchunks = (n + 63) / 64;
	const size_t wave0 = (size_t)blockIdx.x * (blockDim.x >> 6) + wv, nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
	if (lane < 4) { codes[units + lane] = 0; if (lane < 2) nbw[units / 2 + lane] = 0; }
	for (size_t ch = wave0; ch < nchunks; ch += nwaves) {
		const size_t byte0 = ch * 64 * (size_t)L;
		// phase A: units of 16 bases
		for (uint32_t u = lane; u < units; u += 64) {
			const size_t off = byte0 + 16 * (size_t)u;
			uint4 d = make_uint4(0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u);   // 'A's behind the end of the data
			if (off + 16 <= total) d = *(const uint4*)(ascii + off);
			else if (off < total) { uint32_t t[4] = {0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u}; for (size_t q = off; q < total; ++q) { const int b = (int)(q - off); t[b >> 2] = (t[b >> 2] & ~(0xFFu << (8 * (b & 3)))) | ((uint32_t)ascii[q] << (8 * (b & 3))); } d = make_uint4(t[0], t[1], t[2], t[3]); }
			codes[u] = ascii4_to_codes(d.x) | (ascii4_to_codes(d.y) << 8) | (ascii4_to_codes(d.z) << 16) | (ascii4_to_codes(d.w) << 24);
			nbh[u] = (uint16_t)(ascii4_nbits(d.x) | (ascii4_nbits(d.y) << 4) | (ascii4_nbits(d.z) << 8) | (ascii4_nbits(d.w) << 12));
		}
		__builtin_amdgcn_s_waitcnt(0);                                       // (one wave: its LDS writes are in order; make them visible to the other lanes)
		__builtin_amdgcn_wave_barrier();
		// phase B: one read per lane
		const size_t r = ch * 64 + (size_t)lane;
		if (r < n) {
			uint64_t cw[W], nw[NW];
#pragma unroll
			for (int w = 0; w < W; ++w) cw[w] = stream64(codes, 2u * (uint32_t)L * (uint32_t)lane + 64u * w);
#pragma unroll
			for (int w = 0; w < NW; ++w) {
				nw[w] = stream64(nbw, (uint32_t)L * (uint32_t)lane + 64u * w);
				const int left = L - 64 * w;
				if (left < 64) nw[w] &= left > 0 ? ((1ull << left) - 1) : 0ull;
			}
			cf_classify_words<W>(cw, nw, L, e, r, packed, cls, ncnt, nmask);
		}
		__builtin_amdgcn_wave_barrier();                                     // the next chunk's phase A overwrites the streams
	}
}

// The same for reads that arrive PACKED: W words of 2-bit codes (an N holds code 0) and NW words of N flags per read, as a parser that
// packs on the host sends them (round 4: 64 bytes per read over PCIe instead of 150).  One read per lane; in and out may be the same arrays.
template <int W>
__global__ __launch_bounds__(256) void k_classify_packed(const uint64_t *in_packed, const uint64_t *in_nmask, size_t n, int L, int e, uint64_t *packed,
                                                         uint8_t *__restrict__ cls, uint16_t *__restrict__ ncnt, uint64_t *nmask)
{
	constexpr int NW = (W + 1) / 2;
	const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n) return;
	const int nwr = (L + 63) / 64;                                               // words of flags a read really has
	uint64_t cw[W], nw[NW];
#pragma unroll
	for (int w = 0; w < W; ++w) cw[w] = in_packed[r * (size_t)W + w];
#pragma unroll
	for (int w = 0; w < NW; ++w) {
		nw[w] = w < nwr ? in_nmask[r * (size_t)nwr + w] : 0ull;
		const int left = L - 64 * w;
		if (left < 64) nw[w] &= left > 0 ? ((1ull << left) - 1) : 0ull;
	}
	cf_classify_words<W>(cw, nw, L, e, r, packed, cls, ncnt, nmask);
}

// ------------------------------------------------------------------------------------------------
// k_sketch_reads: one thread per read, rolling forward / reverse-complement k-mers out of the packed
// row held in registers; 64 reads of a wave run the same control flow (the only divergent branch is the
// rare k-mer that equals its own reverse complement).
//   WIDE = true : 17 <= k <= 31, k-mers in a 64-bit register pair, mask low word is all ones
//   WIDE = false: k <= 16, everything in 32-bit registers
// ------------------------------------------------------------------------------------------------
// ODDK: k is odd, so no k-mer (nor any of the partly filled registers at the start of a read) equals its reverse complement: every
// base counts, the run counter is the position, and the two data-dependent branches of the loop body become one uniform test.
template <int W, bool WIDE, bool ODDK>
__global__ __launch_bounds__(256) void k_sketch_reads(const uint64_t *__restrict__ packed, const uint32_t *__restrict__ rids,
                                                      size_t n, int L, int k, uint32_t rid0, mcom_mm128 *__restrict__ rec)
{
	const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n) return;
	const uint32_t rid = rids ? rids[t] : rid0 + (uint32_t)t;
	const size_t row = rids ? (size_t)rid : t;
	uint64_t w[W];
#pragma unroll
	for (int q = 0; q < W; ++q) w[q] = packed[row * W + q];

	uint64_t best_x = U64MAX; uint32_t best_i = 0, best_z = 0;
	int run = 0;
	if (WIDE) {
		const uint32_t mhi = (uint32_t)((1ull << (2 * k - 32)) - 1);
		const int sh_hi = 2 * (k - 1) - 32;                 // top base of the reverse k-mer, inside the high word
		uint64_t fwd = 0, rev = 0;
		uint32_t vzero = 0, vone = 1;
		asm volatile("" : "+v"(vzero), "+v"(vone));                            // constants that live in registers (operands of the strand select)
		int i = 0;
#pragma unroll
		for (int q = 0; q < W; ++q) {
			uint64_t cur = w[q];
			const int lim = (L - 32 * q) < 32 ? (L - 32 * q) : 32;
			for (int b = 0; b < lim; ++b, ++i) {
				const uint32_t c = (uint32_t)cur & 3u; cur >>= 2;
				fwd = mcom_mask_hi((fwd << 2) | c, mhi);
				rev = (rev >> 2) | ((uint64_t)((3u ^ c) << sh_hi) << 32);
				if (ODDK) {
					if (i >= k - 1) {                                            // uniform
						const bool lt = fwd < rev;                                  // one comparison serves the strand and the choice
						uint32_t z = lt ? vzero : vone;                             // (from registers: a two-operand select on the same vcc,
						asm volatile("" : "+v"(z));                                 // made here and now: the compiler would compare again after the hash)
						const uint64_t h = mcom_hash64_wide(lt ? fwd : rev, mhi);
						const bool better = h < best_x;
						best_x = better ? h : best_x; best_i = better ? ((uint32_t)i << 1 | z) : best_i;
					}
				} else if (fwd != rev) {
					++run;
					if (run >= k) {
						const uint32_t z = fwd < rev ? 0u : 1u;
						const uint64_t h = mcom_hash64_wide(z ? rev : fwd, mhi);
						if (h < best_x) { best_x = h; best_i = (uint32_t)i; best_z = z; }
					}
				}
			}
		}
	} else {
		const uint32_t mask = (k == 16) ? 0xFFFFFFFFu : ((1u << (2 * k)) - 1u);
		const int sh = 2 * (k - 1);
		uint32_t fwd = 0, rev = 0;
		int i = 0;
#pragma unroll
		for (int q = 0; q < W; ++q) {
			uint64_t cur = w[q];
			const int lim = (L - 32 * q) < 32 ? (L - 32 * q) : 32;
			for (int b = 0; b < lim; ++b, ++i) {
				const uint32_t c = (uint32_t)cur & 3u; cur >>= 2;
				fwd = ((fwd << 2) | c) & mask;
				rev = (rev >> 2) | ((3u ^ c) << sh);
				if (ODDK) {
					if (i >= k - 1) {                                            // uniform
						const uint32_t z = fwd < rev ? 0u : 1u;
						const uint64_t h = mcom_hash64_lo(z ? rev : fwd, mask);
						const bool better = h < best_x;
						best_x = better ? h : best_x; best_i = better ? (uint32_t)i : best_i; best_z = better ? z : best_z;
					}
				} else if (fwd != rev) {
					++run;
					if (run >= k) {
						const uint32_t z = fwd < rev ? 0u : 1u;
						const uint64_t h = mcom_hash64_lo(z ? rev : fwd, mask);
						if (h < best_x) { best_x = h; best_i = (uint32_t)i; best_z = z; }
					}
				}
			}
		}
	}
	mcom_mm128 o;
	o.x = best_x;
	const uint32_t ylow = (WIDE && ODDK) ? best_i : ((best_i << 1) | best_z);       // (that loop keeps position and strand in one register)
	o.y = best_x == U64MAX ? U64MAX : ((uint64_t)rid << 32 | (uint64_t)ylow);
	rec[t] = o;
}

// only class-0 reads carry a record; the others get {MAX, MAX}
__global__ void k_mask_records(const uint8_t *__restrict__ cls, size_t n, mcom_mm128 *__restrict__ rec)
{
	const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t < n && cls[t] != MCOM_CLS_SKETCH) { mcom_mm128 o; o.x = U64MAX; o.y = U64MAX; rec[t] = o; }
}

// the reads of another class than 0, listed (rid << 8 | class): eight class bytes per thread, nearly always one zero word
__global__ void k_special_reads(const uint8_t *__restrict__ cls, size_t n, uint64_t *__restrict__ list, uint32_t cap, uint32_t *__restrict__ count)
{
	const size_t r0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
	if (r0 >= n) return;
	uint64_t w = 0;
	if (r0 + 8 <= n) w = *(const uint64_t*)(cls + r0);
	else for (size_t j = 0; r0 + j < n; ++j) w |= (uint64_t)cls[r0 + j] << (8 * j);
	if (w == 0) return;
	for (int j = 0; j < 8; ++j) {
		const uint32_t c = (uint32_t)(w >> (8 * j)) & 255u;
		if (!c) continue;
		const uint32_t at = atomicAdd(count, 1u);
		if (at < cap) list[at] = ((uint64_t)(r0 + j) << 8) | c;
	}
}
extern "C" int mcom_special_reads(mcom_ctx *ctx, const uint8_t *d_cls, size_t n, uint64_t *d_list, uint32_t cap, uint32_t *d_count)
{
	if (!ctx) return MCOM_E_ARG;
	if (!d_count || (cap && !d_list)) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	MCOM_HIP(ctx, hipMemsetAsync(d_count, 0, 4, ctx->stream));
	if (n == 0) return MCOM_OK;
	if (!d_cls || ((uintptr_t)d_cls & 7)) return mcom_fail(ctx, MCOM_E_ARG, "class array: null or not 8-byte aligned");
	MCOM_LAUNCH(k_special_reads, dim3((unsigned)((n / 8 + 1 + 255) / 256)), dim3(256), 0, ctx->stream, d_cls, n, d_list, cap, d_count);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

// ------------------------------------------------------------------------------------------------
// synthetic reads (minicom_amd/synth.py, plumbing=False); one thread per base
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t sm64(uint64_t z)
{
	z += 0x9E3779B97F4A7C15ull;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

// The repeat-rich genome (kind 1): blocks of 25 kb.  The first 2 kb of every block is a copy of one of n_blocks / 40 family
// sequences (forty copies each, a copy carrying up to five substitutions of its own); one block in 200 holds a tandem repeat of a
// 37-base unit (60 times), one in 400 a poly-A stretch of 400 bases, one in 400 an (AT)150 stretch; everything else is the uniform
// genome.  What a uniform genome never produces: minimizer groups of a thousand reads, long runs of equal minimizers in the contig
// index, 17-mers with forty copies in the Stage-2 index (heavy home lines), long Stage-2 bins.
#define SYN_BLOCK 25000ull
__device__ __forceinline__ unsigned synth_base(uint64_t b0, uint64_t g, int kind, uint64_t n_blocks)
{
	if (kind == 1) {
		const uint64_t blk = g / SYN_BLOCK, off = g - blk * SYN_BLOCK;
		if (off < 2000) {
			const uint64_t fams = n_blocks / 40 ? n_blocks / 40 : 1, fam = blk % fams;
			unsigned b = (unsigned)(sm64(b0 ^ 0x5eedfa11ull ^ (fam * 2000 + off)) & 3);
			const uint64_t m = sm64(b0 + 77 + blk) % 6;                              // this copy's substitutions
			const uint64_t h = sm64(b0 + 99 + g);
			if ((h % 2000) < m) b = (b + 1 + (unsigned)((h >> 20) % 3)) & 3;
			return b;
		}
		if (blk % 200 == 7 && off >= 5000 && off < 5000 + 37 * 60) return (unsigned)(sm64(b0 + 1234 + blk * 64 + (off - 5000) % 37) & 3);
		if (blk % 400 == 11 && off >= 9000 && off < 9400) return 0u;                 // A
		if (blk % 400 == 13 && off >= 12000 && off < 12300) return (off & 1) ? 3u : 0u;   // (AT)n
	}
	return (unsigned)(sm64(b0 + g) & 3);
}

__global__ void k_synth_reads(uint64_t seed, uint64_t G, int L, uint64_t thr, uint64_t first, uint64_t count,
                              uint8_t *__restrict__ out, size_t pitch, int kind)
{
	const uint64_t b0 = sm64(seed + 0), b1 = sm64(seed + 1), b2 = sm64(seed + 2), b3 = sm64(seed + 3);
	const uint64_t total = count * (uint64_t)L;
	for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t q = t / (uint64_t)L; const int o = (int)(t - q * (uint64_t)L);   // output column
		const uint64_t r = first + q;
		const uint64_t start = sm64(b1 + r) % (G - (uint64_t)L + 1);
		const int strand = (int)(sm64(b2 + r) & 1);
		const int i = strand ? L - 1 - o : o;                                              // forward coordinate
		unsigned b = synth_base(b0, start + (uint64_t)i, kind, G / SYN_BLOCK);
		const uint64_t u = sm64(b3 + r * (uint64_t)L + (uint64_t)i);
		if ((u & 0xFFFFFF) < thr) b = (b + 1 + (unsigned)((u >> 24) % 3)) & 3;
		out[q * pitch + o] = "ACGT"[strand ? 3 - b : b];
	}
}

// ------------------------------------------------------------------------------------------------
// host entry points
// ------------------------------------------------------------------------------------------------
template <bool WIDE>
static int launch_sketch(mcom_ctx *ctx, const uint64_t *d_packed, const uint32_t *d_rids, size_t n, int L, int k,
                         uint32_t rid0, mcom_mm128 *d_rec)
{
	const int W = mcom_words_per_read(L);
	const unsigned blocks = (unsigned)((n + 255) / 256);
#define MCOM_CASE(WW) case WW: if (k & 1) MCOM_LAUNCH((k_sketch_reads<WW, WIDE, true>), dim3(blocks), dim3(256), 0, ctx->stream, d_packed, d_rids, n, L, k, rid0, d_rec); \
	else MCOM_LAUNCH((k_sketch_reads<WW, WIDE, false>), dim3(blocks), dim3(256), 0, ctx->stream, d_packed, d_rids, n, L, k, rid0, d_rec); break;
	McomProfScope ps_(ctx, PROF_SKETCH_READS);
	switch (W) { MCOM_CASE(1) MCOM_CASE(2) MCOM_CASE(3) MCOM_CASE(4) MCOM_CASE(5) MCOM_CASE(6) MCOM_CASE(7) MCOM_CASE(8)
	default: return mcom_fail(ctx, MCOM_E_ARG, "read length %d not supported (1..256)", L); }
#undef MCOM_CASE
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

extern "C" int mcom_sketch_reads(mcom_ctx *ctx, const uint64_t *d_packed, const uint32_t *d_rids, size_t n, int L,
                                 int k, uint32_t rid0, mcom_mm128 *d_rec)
{
	if (!ctx) return MCOM_E_ARG;
	if (L < 1 || L > 256) return mcom_fail(ctx, MCOM_E_ARG, "read length %d out of range 1..256", L);
	if (k < 1 || k > 31) return mcom_fail(ctx, MCOM_E_ARG, "k=%d out of range 1..31", k);
	if (n == 0) return MCOM_OK;
	if (!d_packed || !d_rec) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	return k >= 17 ? launch_sketch<true>(ctx, d_packed, d_rids, n, L, k, rid0, d_rec)
	               : launch_sketch<false>(ctx, d_packed, d_rids, n, L, k, rid0, d_rec);
}

// a1 as a batched entry of its own: hash64 (sketch.c:27-37) of n k-mers with the two device forms the sketch kernels use
__global__ void k_hash64(const uint64_t *__restrict__ kmer, size_t n, int k, uint64_t *__restrict__ out)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	if (k >= 17) out[i] = mcom_hash64_wide(kmer[i], (uint32_t)((1ull << (2 * k - 32)) - 1));
	else out[i] = mcom_hash64_lo((uint32_t)kmer[i], k == 16 ? 0xFFFFFFFFu : ((1u << (2 * k)) - 1u));
}
extern "C" int mcom_hash64_batch(mcom_ctx *ctx, const uint64_t *d_kmer, size_t n, int k, uint64_t *d_hash)
{
	if (!ctx) return MCOM_E_ARG;
	if (k < 1 || k > 31) return mcom_fail(ctx, MCOM_E_ARG, "k=%d out of range 1..31", k);
	if (n == 0) return MCOM_OK;
	if (!d_kmer || !d_hash) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	MCOM_LAUNCH(k_hash64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_kmer, n, k, d_hash);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

// kt_for_reads for reads that were packed by the caller (a FASTQ parser that packs on the host: 2 bits per base + one N flag per base
// cross PCIe instead of a byte per base).  d_in_packed [n][W]: codes A0 C1 G2 T3, an N holds 0; d_in_nmask [n][ceil(L/64)]: bit i = base
// i is an N.  Outputs as mcom_process_reads; d_packed / d_nmask may be the input arrays themselves.
extern "C" int mcom_process_reads_packed(mcom_ctx *ctx, const uint64_t *d_in_packed, const uint64_t *d_in_nmask, size_t n, int L, int k, int e,
                                         uint32_t rid0, uint64_t *d_packed, uint8_t *d_cls, uint16_t *d_ncnt, uint64_t *d_nmask, mcom_mm128 *d_rec)
{
	if (!ctx) return MCOM_E_ARG;
	if (L < 1 || L > 256) return mcom_fail(ctx, MCOM_E_ARG, "read length %d out of range 1..256", L);
	if (k < 1 || k > 31) return mcom_fail(ctx, MCOM_E_ARG, "k=%d out of range 1..31", k);
	if (n == 0) return MCOM_OK;
	if (!d_in_packed || !d_in_nmask || !d_packed || !d_cls || !d_ncnt || !d_rec) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	const int W = mcom_words_per_read(L);
	const unsigned blocks = (unsigned)((n + 255) / 256);
	{
		McomProfScope ps_(ctx, PROF_CLASSIFY_PACK);
#define MCOM_CASE(WW) case WW: MCOM_LAUNCH((k_classify_packed<WW>), dim3(blocks), dim3(256), 0, ctx->stream, d_in_packed, d_in_nmask, n, L, e, d_packed, d_cls, d_ncnt, d_nmask); break;
		switch (W) { MCOM_CASE(1) MCOM_CASE(2) MCOM_CASE(3) MCOM_CASE(4) MCOM_CASE(5) MCOM_CASE(6) MCOM_CASE(7) MCOM_CASE(8) default: return mcom_fail(ctx, MCOM_E_ARG, "read length %d not supported", L); }
#undef MCOM_CASE
	}
	MCOM_LAUNCH_CHECK(ctx);
	int rc = mcom_sketch_reads(ctx, d_packed, nullptr, n, L, k, rid0, d_rec);
	if (rc) return rc;
	MCOM_LAUNCH(k_mask_records, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_cls, n, d_rec);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

// process_reads without the sketch: classes, N counts, N masks, packed rows (what mcom_process_reads starts with; a caller that runs
// the sketch of one batch beside the classification of the next -- one is bound by the VALU, the other by HBM -- asks for the parts)
extern "C" int mcom_classify_reads(mcom_ctx *ctx, const uint8_t *d_ascii, size_t pitch, size_t n, int L, int e,
                                   uint64_t *d_packed, uint8_t *d_cls, uint16_t *d_ncnt, uint64_t *d_nmask)
{
	if (!ctx) return MCOM_E_ARG;
	if (L < 1 || L > 256) return mcom_fail(ctx, MCOM_E_ARG, "read length %d out of range 1..256", L);
	if (pitch < (size_t)L) return mcom_fail(ctx, MCOM_E_ARG, "pitch %zu < read length %d", pitch, L);
	if (n == 0) return MCOM_OK;
	if (!d_ascii || !d_packed || !d_cls || !d_ncnt) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	const int W = mcom_words_per_read(L), NW = (L + 63) / 64;
	const int G = 16;                                                       // L <= 128: 8 bases per lane; above: 16 bases per lane
	const size_t waves = (n + (64 / G) - 1) / (64 / G);
	size_t blocks = (waves + 3) / 4;
	const size_t cap = (size_t)ctx->n_cu * 16;
	if (blocks > cap) blocks = cap;
	const bool flat = pitch == (size_t)L && ((uintptr_t)d_ascii & 15) == 0;      // reads back to back, 16-byte aligned: the byte-stream kernel
	if (flat) {
		McomProfScope ps_(ctx, PROF_CLASSIFY_PACK);
		const size_t chunks = (n + 63) / 64;
		size_t fb = (chunks + 3) / 4;
		if (fb > (size_t)ctx->n_cu * 8) fb = (size_t)ctx->n_cu * 8;
		const size_t lds = 4 * (size_t)(4 * L + 2 * L + 8) * 4;
#define MCOM_CASE(WW) case WW: MCOM_LAUNCH((k_classify_flat<WW>), dim3((unsigned)fb), dim3(256), lds, ctx->stream, d_ascii, n, L, e, d_packed, d_cls, d_ncnt, d_nmask); break;
		switch (W) { MCOM_CASE(1) MCOM_CASE(2) MCOM_CASE(3) MCOM_CASE(4) MCOM_CASE(5) MCOM_CASE(6) MCOM_CASE(7) MCOM_CASE(8) default: return mcom_fail(ctx, MCOM_E_ARG, "read length %d not supported", L); }
#undef MCOM_CASE
	} else
	{ McomProfScope ps_(ctx, PROF_CLASSIFY_PACK);
	if (L <= 128) MCOM_LAUNCH((k_classify_pack<16>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_ascii, pitch, n, L, e, d_packed, W, d_cls, d_ncnt, d_nmask, NW);
	else          MCOM_LAUNCH(k_classify_pack16, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_ascii, pitch, n, L, e, d_packed, W, d_cls, d_ncnt, d_nmask, NW); }
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}
// ... and its end: the sketch of the packed rows (mm_sketch_two for the kept reads), records of the other classes blanked
extern "C" int mcom_sketch_classified(mcom_ctx *ctx, const uint64_t *d_packed, const uint8_t *d_cls, size_t n, int L, int k, uint32_t rid0, mcom_mm128 *d_rec)
{
	if (!ctx) return MCOM_E_ARG;
	if (n == 0) return MCOM_OK;
	if (!d_packed || !d_cls || !d_rec) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	int rc = mcom_sketch_reads(ctx, d_packed, nullptr, n, L, k, rid0, d_rec);
	if (rc) return rc;
	MCOM_LAUNCH(k_mask_records, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_cls, n, d_rec);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}
extern "C" int mcom_process_reads(mcom_ctx *ctx, const uint8_t *d_ascii, size_t pitch, size_t n, int L, int k, int e,
                                  uint32_t rid0, uint64_t *d_packed, uint8_t *d_cls, uint16_t *d_ncnt,
                                  uint64_t *d_nmask, mcom_mm128 *d_rec)
{
	if (!ctx) return MCOM_E_ARG;
	if (k < 1 || k > 31) return mcom_fail(ctx, MCOM_E_ARG, "k=%d out of range 1..31", k);
	if (n && !d_rec) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	const int rc = mcom_classify_reads(ctx, d_ascii, pitch, n, L, e, d_packed, d_cls, d_ncnt, d_nmask);
	return rc ? rc : mcom_sketch_classified(ctx, d_packed, d_cls, n, L, k, rid0, d_rec);
}

extern "C" int mcom_synth_reads(mcom_ctx *ctx, uint64_t seed, uint64_t n_reads, int L, int coverage, double sub_rate,
                                uint64_t first, uint64_t count, uint8_t *d_ascii, size_t pitch)
{
	return mcom_synth_reads_genome(ctx, seed, n_reads, L, coverage, sub_rate, 0, first, count, d_ascii, pitch);
}
extern "C" int mcom_synth_reads_genome(mcom_ctx *ctx, uint64_t seed, uint64_t n_reads, int L, int coverage, double sub_rate, int genome_kind,
                                       uint64_t first, uint64_t count, uint8_t *d_ascii, size_t pitch)
{
	if (!ctx) return MCOM_E_ARG;
	if (L < 1 || L > 256 || coverage < 1 || pitch < (size_t)L || genome_kind < 0 || genome_kind > 1) return mcom_fail(ctx, MCOM_E_ARG, "bad synth arguments");
	if (count == 0) return MCOM_OK;
	if (!d_ascii) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	uint64_t G = n_reads * (uint64_t)L / (uint64_t)coverage;
	if (G < (uint64_t)L + 1) G = (uint64_t)L + 1;
	const uint64_t thr = (uint64_t)(sub_rate * (double)(1 << 24));
	uint64_t blocks = (count * (uint64_t)L + 255) / 256;
	const uint64_t cap = (uint64_t)ctx->n_cu * 32;
	if (blocks > cap) blocks = cap;
	MCOM_LAUNCH(k_synth_reads, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, seed, G, L, thr, first, count, d_ascii, pitch, genome_kind);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}


// ---- records that travelled: the minimizer of a read does not change when the read changes rank ---------------------
// After the multi-GPU bucket exchange a rank holds reads that were sketched by their sender; x and the low half of y
// (position<<1 | strand) came along, only the read id is new: the row's index on this rank.
__global__ void k_records_assemble(const uint64_t *__restrict__ x, const uint32_t *__restrict__ ylow, size_t n, uint32_t rid0, mcom_mm128 *__restrict__ rec)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	mcom_mm128 r;
	r.x = x[i];
	r.y = r.x == U64MAX ? U64MAX : (((uint64_t)(rid0 + (uint32_t)i) << 32) | (uint64_t)ylow[i]);
	rec[i] = r;
}

extern "C" int mcom_records_assemble(mcom_ctx *ctx, const uint64_t *d_x, const uint32_t *d_ylow, size_t n, uint32_t rid0, mcom_mm128 *d_rec)
{
	if (!ctx) return MCOM_E_ARG;
	if (n == 0) return MCOM_OK;
	if (!d_x || !d_ylow || !d_rec) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	if ((uint64_t)rid0 + n > (1ull << 32)) return mcom_fail(ctx, MCOM_E_ARG, "read ids exceed 32 bits");
	MCOM_LAUNCH(k_records_assemble, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_x, d_ylow, n, rid0, d_rec);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}
