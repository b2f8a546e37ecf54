// minicom_amd/csrc/flagsort.hip -- the reference's radix_sort_128x with its exact, UNSTABLE element order.
//
// radix_sort_128x (reference ksort.h:108-157, misc.c:22) is an in-place American-flag MSD radix sort on .x, 8 bits
// a level, that finishes ranges of <= 64 elements by insertion sort.  Above 64 elements the cycle-leader
// permutation reorders equal keys, and mm_idx (kthread_idx.c:126, :154-155) hands equal minimizers to find_next in
// exactly that order.  To reproduce the order the algorithm is run as written, by ONE lane per range: the ranges
// are the 2^b index buckets (thousands of them run side by side, one wave each), small enough to be sorted inside
// LDS; a range that does not fit is sorted in HBM by the same code.
#include "mcom_dev.hpp"

#define FS_STACK 1024

struct FsRange { uint32_t b, e, shift; };

// the algorithm on an array `a` (LDS or global); bb/be: 256-entry scratch, stk: pending big ranges
template <class PTR>
__device__ void flag_sort_range(PTR a, uint32_t n, uint32_t *bb, uint32_t *be, FsRange *stk, uint32_t stk_cap, uint32_t *overflow)
{
	auto insertion = [&](uint32_t beg, uint32_t end) {                       // rs_insertsort (ksort.h:112-122)
		for (uint32_t i = beg + 1; i < end; ++i) {
			if (a[i].x < a[i - 1].x) {
				const mcom_mm128 t = a[i]; uint32_t j = i;
				while (j > beg && t.x < a[j - 1].x) { a[j] = a[j - 1]; --j; }
				a[j] = t;
			}
		}
	};
	if (n <= 64) { insertion(0, n); return; }                                // radix_sort (ksort.h:153-157)
	uint32_t sp = 0;
	stk[sp++] = FsRange{0, n, 56};
	while (sp) {
		const FsRange r = stk[--sp];
		const uint32_t beg = r.b, end = r.e, s = r.shift;
		// rs_sort (ksort.h:123-152)
		for (int q = 0; q < 256; ++q) { bb[q] = beg; be[q] = beg; }
		for (uint32_t i = beg; i != end; ++i) ++be[(a[i].x >> s) & 255];
		for (int q = 1; q < 256; ++q) { be[q] += be[q - 1] - beg; bb[q] = be[q - 1]; }
		for (int q = 0; q < 256;) {
			if (bb[q] != be[q]) {
				int l = (int)((a[bb[q]].x >> s) & 255);
				if (l != q) {
					mcom_mm128 hold = a[bb[q]], moved;
					do {
						moved = hold; hold = a[bb[l]]; a[bb[l]++] = moved;
						l = (int)((hold.x >> s) & 255);
					} while (l != q);
					a[bb[q]++] = hold;
				} else ++bb[q];
			} else ++q;
		}
		bb[0] = beg;
		for (int q = 1; q < 256; ++q) bb[q] = be[q - 1];
		if (s) {
			const uint32_t nxt = s > 8 ? s - 8 : 0;
			for (int q = 0; q < 256; ++q) {
				const uint32_t cnt = be[q] - bb[q];
				if (cnt > 64) { if (sp < stk_cap) stk[sp++] = FsRange{bb[q], be[q], nxt}; else *overflow = 1; }
				else if (cnt > 1) insertion(bb[q], be[q]);
			}
		}
	}
}

// one 64-lane workgroup per range [bstart[r], bstart[r+1]); lanes stage the range into LDS and back, lane 0 sorts
__global__ __launch_bounds__(64) void k_flag_sort(mcom_mm128 *__restrict__ rec, const uint32_t *__restrict__ bstart, uint32_t nr,
                                                  uint32_t lds_cap, uint32_t *__restrict__ overflow)
{
	extern __shared__ __align__(16) unsigned char smem[];
	uint32_t *bb = (uint32_t*)smem, *be = bb + 256;
	FsRange *stk = (FsRange*)(be + 256);
	mcom_mm128 *buf = (mcom_mm128*)(stk + FS_STACK);
	const uint32_t r = blockIdx.x;
	if (r >= nr) return;
	const uint32_t beg = bstart[r], end = bstart[r + 1], n = end - beg;
	if (n < 2) return;
	if (n <= lds_cap) {
		for (uint32_t i = threadIdx.x; i < n; i += 64) buf[i] = rec[beg + i];
		__syncthreads();
		if (threadIdx.x == 0) flag_sort_range(buf, n, bb, be, stk, FS_STACK, overflow);
		__syncthreads();
		for (uint32_t i = threadIdx.x; i < n; i += 64) rec[beg + i] = buf[i];
	} else if (threadIdx.x == 0) {
		flag_sort_range(rec + beg, n, bb, be, stk, FS_STACK, overflow);
	}
}

// ---- index buckets: the same algorithm with three bytes per element in LDS, a dozen buckets per CU --------------------------------
// The cycle-leader permutation is one lane's chain of dependent LDS accesses per bucket (it is the reference's order of equal keys,
// it cannot be split), so what counts is how many buckets a CU holds while one lane of each walks: with 8-byte elements (key,
// position) six buckets of ~2900 records.  The walk only needs an element's DIGIT at the current level and its identity, so here
// an element is a digit byte and a 16-bit index (D, I): 13 buckets per CU (34 -> 15 ms per step with the rest of round 2's changes:
// only non-empty bins are visited, runs of in-place elements -- the top byte of a minimum of hashes is nearly always 0 -- are
// skipped 64 at a time by the wave, lane 0 goes from cycle to cycle on its own).  The keys stay in HBM / L2 and are gathered when
// a level's digits are made and when the small ranges are finished:
//   * a range of at most 64 elements is finished by rs_insertsort, which is a STABLE sort of the range as it stands, so every element
//     can find its final rank by itself (elements of the range with a smaller key, plus equal ones before it): one lane per position,
//     the keys of a 192-position window staged in LDS, and the record goes straight to its final place in the output -- no
//     insertion loop, no second index array;
//   * larger ranges go on the stack for the next level, as in the reference (ksort.h:146-151).
#define FST_STACK 96
struct FsTok { uint16_t b, e; uint32_t shift; };                             // a pending range of a bucket of at most 65 535 records
__global__ __launch_bounds__(64) void k_flag_sort_tokens(const mcom_mm128 *__restrict__ in, mcom_mm128 *__restrict__ out,
                                                         const uint32_t *__restrict__ bstart, uint32_t nr, uint32_t lds_cap,
                                                         uint32_t *__restrict__ overflow)
{
	extern __shared__ __align__(16) unsigned char smem[];
	// (round 5: what a bucket needs beside its tokens went from 4 736 to 3 328 bytes -- the LDS a bucket takes decides how many buckets a CU
	// walks at once, and the walk is nothing but latency: the key window of the finishing step lies over the fill pointers, which are dead by
	// then -- bb[q] == be[q] for every bin once the permutation is done --, and a pending range is 8 bytes)
	uint64_t *KW = (uint64_t*)smem;                                          // keys of a window of 192 positions (finish_small only)
	uint32_t *bb = (uint32_t*)smem;                                          // fill pointer of every bin of the current level (permutation only)
	uint32_t *be = (uint32_t*)(KW + 192);                                    // end of every bin of the current level
	FsTok *stk = (FsTok*)(be + 256);
	uint16_t *I = (uint16_t*)(stk + FST_STACK);                              // [cap] which record of the bucket stands at a position
	uint8_t *D = (uint8_t*)(I + lds_cap);                                    // [cap] its digit at the current level
	const uint32_t r = blockIdx.x;
	if (r >= nr) return;
	const int lane = threadIdx.x;
	const uint32_t beg = bstart[r], end = bstart[r + 1], n = end - beg;
	if (n == 0) return;
	if (n > lds_cap || n > 65535u) {                                           // does not fit: the record form, in HBM, by one lane
		for (uint32_t i = lane; i < n; i += 64) out[beg + i] = in[beg + i];
		__threadfence(); __syncthreads();
		if (lane == 0) flag_sort_range(out + beg, n, bb, be, (FsRange*)stk, (uint32_t)(FST_STACK * sizeof(FsTok) / sizeof(FsRange)), overflow);
		return;
	}
	const mcom_mm128 *rec = in + beg;
	for (uint32_t i = lane; i < n; i += 64) I[i] = (uint16_t)i;
	__syncthreads();
	// rs_insertsort of every range of at most 64 positions among the bins [first bin .. ] of [rb, re): bin of position p = D[p]
	// when by_bins, else the one range [rb, re) itself.  The ranks are final: the records are written out.
	// (round 4: a chunk none of whose positions lies in a small bin is passed over -- at the levels where nearly everything shares one
	// digit that is every chunk, and each cost two barriers and a gather of 192 keys; the three keys a lane gathers travel together)
	auto finish_small = [&](uint32_t rb, uint32_t re, bool by_bins) {
		for (uint32_t base = rb; base < re; base += 64) {
			const uint32_t p = base + (uint32_t)lane;
			uint32_t b0 = rb, e0 = re;
			if (by_bins && p < re) { const uint32_t d = D[p]; e0 = be[d]; b0 = d ? be[d - 1] : rb; }
			const bool small = p < re && e0 - b0 <= 64;
			if (!__any(small)) continue;                                         // uniform: the workgroup is one wave
			const uint32_t w0 = base >= rb + 64 ? base - 64 : rb;                // the window starts at most 64 positions before the chunk
			const uint32_t w1 = base + 128 < re ? base + 128 : re;
			__syncthreads();
			{
				const uint32_t q0 = w0 + lane, q1 = q0 + 64, q2 = q0 + 128;
				uint64_t k0 = 0, k1 = 0, k2 = 0;
				if (q0 < w1) k0 = rec[I[q0]].x;
				if (q1 < w1) k1 = rec[I[q1]].x;
				if (q2 < w1) k2 = rec[I[q2]].x;
				if (q0 < w1) KW[q0 - w0] = k0;
				if (q1 < w1) KW[q1 - w0] = k1;
				if (q2 < w1) KW[q2 - w0] = k2;
			}
			__syncthreads();
			if (small) {
				const uint64_t mine = KW[p - w0];
				uint32_t rank = b0;
				for (uint32_t q = b0; q < e0; ++q) { const uint64_t k = KW[q - w0]; rank += (k < mine || (k == mine && q < p)) ? 1u : 0u; }
				out[beg + rank] = rec[I[p]];
			}
		}
	};
	if (n <= 64) { finish_small(0, n, false); return; }                        // radix_sort (ksort.h:153-157)
	uint32_t sp = 1;                                                           // uniform: every lane keeps the same count
	if (lane == 0) stk[0] = FsTok{0, (uint16_t)n, 56};
	__syncthreads();
	while (sp) {
		const FsTok rg = stk[--sp];
		const uint32_t rb = rg.b, re = rg.e, s = rg.shift;
		__syncthreads();                                                       // everybody has read the entry before a push reuses it
		// rs_sort (ksort.h:123-152): digits, histogram and bin bounds by all lanes
		for (int q = lane; q < 256; q += 64) be[q] = 0;
		__syncthreads();
		for (uint32_t i0 = rb + lane; i0 < re; i0 += 256) {                      // four gathers in flight per lane
			uint64_t kx[4];
#pragma unroll
			for (int u = 0; u < 4; ++u) { const uint32_t i = i0 + 64u * u; kx[u] = i < re ? rec[I[i]].x : 0ull; }
#pragma unroll
			for (int u = 0; u < 4; ++u) { const uint32_t i = i0 + 64u * u; if (i < re) { const uint32_t d = (uint32_t)(kx[u] >> s) & 255u; D[i] = (uint8_t)d; atomicAdd(&be[d], 1u); } }
		}
		__syncthreads();
		{
			const uint32_t c0 = be[4 * lane], c1 = be[4 * lane + 1], c2 = be[4 * lane + 2], c3 = be[4 * lane + 3];
			const uint32_t tot = c0 + c1 + c2 + c3;
			uint32_t incl = tot;
#pragma unroll
			for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d, 64); if (lane >= d) incl += v; }
			uint32_t a = rb + incl - tot;
			bb[4 * lane] = a; a += c0; be[4 * lane] = a;
			bb[4 * lane + 1] = a; a += c1; be[4 * lane + 1] = a;
			bb[4 * lane + 2] = a; a += c2; be[4 * lane + 2] = a;
			bb[4 * lane + 3] = a; a += c3; be[4 * lane + 3] = a;
		}
		__syncthreads();
		// The cycle-leader permutation as written (ksort.h:132-144), with two short cuts that change nothing: only the bins that hold
		// something are visited, and an element at bb[q] whose digit is q is "in place" -- the reference does ++bb[q] and nothing else,
		// nothing ever writes into bin q ahead of bb[q] (a cycle closes AT bb[q]) -- so the first element of a bin that is not in place
		// is found by all lanes at once.  Lane 0 then goes from cycle to cycle on its own and asks the wave again only after eight
		// in-place elements in a row.  The cycles themselves stay with one lane: they are the reference's order of equal keys.
		// (Round 5, measured and dropped: lane 0 going from BIN to bin on its own as well -- a level has ~100 bins that hold something, each
		// costs two barriers and a scan -- 11.5 -> 14.5 ms per step: what the wave does per bin is scalar code, what lane 0 does instead is not.)
		for (int q0 = 0; q0 < 256; q0 += 64) {
			uint64_t live = __ballot(bb[q0 + lane] != be[q0 + lane]);             // uniform
			while (live) {
				const int q = q0 + __ffsll((unsigned long long)live) - 1;
				live &= live - 1;
				for (;;) {
					const uint32_t b0 = bb[q], e0 = be[q];                           // uniform
					if (b0 == e0) break;
					uint32_t found = e0;
					for (uint32_t p0 = b0; p0 < e0; p0 += 64) {
						const uint32_t p = p0 + (uint32_t)lane;
						const bool neq = p < e0 && (int)D[p] != q;
						const uint64_t m = __ballot(neq);
						if (m) { found = p0 + (uint32_t)__ffsll((unsigned long long)m) - 1u; break; }
					}
					if (lane == 0) {
						uint32_t at = found, inplace = 0;
						while (at != e0 && inplace < 8) {
							int l = (int)D[at];
							if (l == q) { ++at; ++inplace; continue; }                  // in place: ++bb[q] (ksort.h:143)
							inplace = 0;
							uint8_t hd = D[at]; uint16_t hi = I[at];
							do {
								const uint8_t md = hd; const uint16_t mi = hi;
								const uint32_t pos = atomicAdd(&bb[l], 1u);                  // (one LDS operation for the read and the ++)
								hd = D[pos]; hi = I[pos];
								D[pos] = md; I[pos] = mi;
								l = (int)hd;
							} while (l != q);
							D[at] = hd; I[at] = hi;
							++at;
						}
						bb[q] = at;
					}
					__syncthreads();
					if (found == e0) break;
				}
			}
		}
		__syncthreads();
		// the bins: above 64 elements to the next level, the others finished (at the last level every bin is one key: in order as it is)
		if (s) {
			const uint32_t nxt = s > 8 ? s - 8 : 0;
			for (int q0 = 0; q0 < 256; q0 += 64) {
				const int q = q0 + lane;
				const uint32_t b0 = q ? be[q - 1] : rb, e0 = be[q], cnt = e0 - b0;
				const bool big = cnt > 64;
				const uint64_t bm = __ballot(big);
				if (big) {
					const uint32_t at = sp + (uint32_t)__popcll(bm & (lane == 0 ? 0ull : (~0ull >> (64 - lane))));
					if (at < FST_STACK) stk[at] = FsTok{(uint16_t)b0, (uint16_t)e0, nxt}; else *overflow = 1;
				}
				sp += (uint32_t)__popcll(bm);
			}
			if (sp > FST_STACK) sp = FST_STACK;
			finish_small(rb, re, true);
		} else {
			for (uint32_t p = rb + lane; p < re; p += 64) out[beg + p] = rec[I[p]];
		}
		__syncthreads();
	}
}

// every bucket [d_bstart[r], d_bstart[r+1]) of d_in (all of whose x share their low low_bits bits = r) in the
// reference's order, to d_out
int mcom_flag_sort_buckets(mcom_ctx *ctx, const mcom_mm128 *d_in, mcom_mm128 *d_out, const uint32_t *d_bstart, uint32_t nr, int low_bits,
                           uint32_t max_range, uint32_t *d_overflow)
{
	if (nr == 0) return MCOM_OK;
	(void)low_bits;                                                            // (the records of a bucket share their low bits: their order by x is the order of the keys)
	const size_t fixed = 192 * 8 + 256 * 4 + FST_STACK * sizeof(FsTok);
	size_t cap = max_range < 64 ? 64 : max_range;
	const size_t lds_max = 150 * 1024;
	if (cap > 65535) cap = 65535;
	cap = (cap + 7) & ~(size_t)7;                                            // the digit bytes start behind the 16-bit indices: keep that 8-byte aligned
	if (fixed + cap * 3 > lds_max) cap = ((lds_max - fixed) / 3) & ~(size_t)7;
	const size_t lds = fixed + cap * 3;
	MCOM_HIP(ctx, hipFuncSetAttribute((const void*)k_flag_sort_tokens, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	MCOM_LAUNCH(k_flag_sort_tokens, dim3(nr), dim3(64), lds, ctx->stream, d_in, d_out, d_bstart, nr, (uint32_t)cap, d_overflow);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

// starts of the runs of equal (x & mask) in an array sorted by that value: bstart[v] = first index with value >= v, v = 0 .. nb (one
// thread per v, binary search: see k_seg_bounds in sort.hip)
__global__ void k_bucket_starts(const mcom_mm128 *__restrict__ rec, size_t n, uint64_t mask, uint32_t nb, uint32_t *__restrict__ bstart)
{
	const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
	if (v > nb) return;
	size_t lo = 0, hi = n;
	while (lo < hi) { const size_t mid = (lo + hi) >> 1; if ((rec[mid].x & mask) < (uint64_t)v) lo = mid + 1; else hi = mid; }
	bstart[v] = (uint32_t)lo;
}

// sorts every range of d_rec given by d_bstart[0..nr] into the reference's order
int mcom_flag_sort_ranges(mcom_ctx *ctx, mcom_mm128 *d_rec, const uint32_t *d_bstart, uint32_t nr, uint32_t max_range, uint32_t *d_overflow)
{
	if (nr == 0) return MCOM_OK;
	const size_t fixed = 2 * 256 * 4 + FS_STACK * sizeof(FsRange);
	size_t cap = max_range;
	const size_t lds_max = 150 * 1024;
	if (fixed + cap * sizeof(mcom_mm128) > lds_max) cap = (lds_max - fixed) / sizeof(mcom_mm128);
	const size_t lds = fixed + cap * sizeof(mcom_mm128);
	MCOM_HIP(ctx, hipFuncSetAttribute((const void*)k_flag_sort, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	MCOM_LAUNCH(k_flag_sort, dim3(nr), dim3(64), lds, ctx->stream, d_rec, d_bstart, nr, (uint32_t)cap, d_overflow);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

int mcom_bucket_starts(mcom_ctx *ctx, const mcom_mm128 *d_rec, size_t n, int bits, uint32_t *d_bstart)
{
	const uint32_t nb = 1u << bits;
	MCOM_LAUNCH(k_bucket_starts, dim3((nb + 1 + 255) / 256), dim3(256), 0, ctx->stream, d_rec, n, (uint64_t)nb - 1, nb, d_bstart);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

// a5, exact: radix_sort_128x with the reference's element order (one range = the whole array).  Sequential by
// nature: meant for index buckets and tests, not for bulk sorting (mcom_radix_sort_128x is the fast, stable one).
extern "C" int mcom_radix_sort_128x_ref_order(mcom_ctx *ctx, mcom_mm128 *d_a, size_t n)
{
	if (!ctx) return MCOM_E_ARG;
	if (n < 2) return MCOM_OK;
	if (!d_a) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	if (n >= (1ull << 31)) return mcom_fail(ctx, MCOM_E_ARG, "too many records");
	int rc = mcom_ws_reserve(ctx, 64);
	if (rc) return rc;
	uint32_t *d = (uint32_t*)ctx->ws;
	const uint32_t h[3] = {0, (uint32_t)n, 0};
	MCOM_HIP(ctx, hipMemcpyAsync(d, h, 12, hipMemcpyHostToDevice, ctx->stream));
	rc = mcom_flag_sort_ranges(ctx, d_a, d, 1, (uint32_t)n, d + 2);
	if (rc) return rc;
	uint32_t ov = 0;
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &ov, d + 2, 4));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	if (ov) return mcom_fail(ctx, MCOM_E_OVERFLOW, "radix sort emulation ran out of range stack");
	return MCOM_OK;
}
