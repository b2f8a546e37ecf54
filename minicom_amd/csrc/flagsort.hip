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

// ---- index buckets: the same algorithm on 8-byte elements, several buckets per CU ------------------------------------
// All records of an index bucket share their low `low_bits` bits, so inside a bucket the order by x is the order by
// x >> low_bits, which fits 48 bits; with the record's position in the bucket (16 bits) an element is one uint64 and a
// bucket of thousands of records takes half the LDS, i.e. twice the buckets in flight for the latency-bound
// cycle-leader permutation, which stays with one lane.  Histogram, prefix sums and the insertion sorts of the
// sub-ranges (the bulk of the instructions) use all 64 lanes.  The records are moved once, at the end.
#define FSB_STACK 192

__global__ __launch_bounds__(64) void k_flag_sort_bucket(const mcom_mm128 *__restrict__ in, mcom_mm128 *__restrict__ out,
                                                         const uint32_t *__restrict__ bstart, uint32_t nr, int low_bits, uint32_t lds_cap,
                                                         uint32_t *__restrict__ overflow)
{
	extern __shared__ __align__(16) unsigned char smem[];
	uint32_t *bb = (uint32_t*)smem, *be = bb + 256;
	FsRange *stk = (FsRange*)(be + 256);
	uint64_t *E = (uint64_t*)(stk + FSB_STACK);
	const uint32_t r = blockIdx.x;
	if (r >= nr) return;
	const int lane = threadIdx.x;
	const uint32_t beg = bstart[r], end = bstart[r + 1], n = end - beg;
	if (n == 0) return;
	if (n > lds_cap || n > 65536u) {                                           // does not fit: the record form, in HBM
		for (uint32_t i = lane; i < n; i += 64) out[beg + i] = in[beg + i];
		__threadfence(); __syncthreads();
		if (lane == 0) flag_sort_range(out + beg, n, bb, be, stk, FSB_STACK, overflow);
		return;
	}
	for (uint32_t i = lane; i < n; i += 64) E[i] = ((in[beg + i].x >> low_bits) << 16) | (uint64_t)i;
	__syncthreads();
	auto digit = [&](uint64_t e, uint32_t s) -> uint32_t { return (uint32_t)(((((e >> 16) << low_bits) | (uint64_t)r) >> s) & 255); };
	auto insertion = [&](uint32_t b0, uint32_t e0) {                           // rs_insertsort (ksort.h:112-122)
		for (uint32_t i = b0 + 1; i < e0; ++i) {
			if ((E[i] >> 16) < (E[i - 1] >> 16)) {
				const uint64_t t = E[i]; uint32_t j = i;
				while (j > b0 && (t >> 16) < (E[j - 1] >> 16)) { E[j] = E[j - 1]; --j; }
				E[j] = t;
			}
		}
	};
	if (n <= 64) { if (lane == 0) insertion(0, n); }                           // radix_sort (ksort.h:153-157)
	else {
		uint32_t sp = 1;                                                       // uniform: every lane keeps the same count
		if (lane == 0) stk[0] = FsRange{0, n, 56};
		__syncthreads();
		while (sp) {
			const FsRange rg = stk[--sp];
			const uint32_t rb = rg.b, re = rg.e, s = rg.shift;
			__syncthreads();                                                   // everybody has read the entry before a push reuses it
			// rs_sort (ksort.h:123-152): histogram and bucket bounds by all lanes
			for (int q = lane; q < 256; q += 64) be[q] = 0;
			__syncthreads();
			for (uint32_t i = rb + lane; i < re; i += 64) atomicAdd(&be[digit(E[i], s)], 1u);
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
			// The cycle-leader permutation, as written (ksort.h:132-144) -- except that the elements it merely steps over are stepped
			// over 64 at a time: an element at bb[q] whose digit is q is "in place", the reference does ++bb[q] and nothing else,
			// and nothing ever writes into bin q ahead of bb[q] (a cycle closes AT bb[q]), so the first element of the bin that is
			// not in place can be found by all lanes at once.  At the top levels nearly every element is in place (the hashes of a
			// bucket are minima of many k-mer hashes: their top byte is almost always 0), and walking them one by one was half of
			// this kernel's time.  The cycles themselves stay with one lane: they are the reference's order of equal keys.
			// Only the bins that hold something are visited (a barrier round per empty bin was a quarter of a millisecond per level),
			// and lane 0 keeps going on its own from cycle to cycle -- as the reference does -- until it has stepped over eight
			// in-place elements in a row: only then is the wave asked to find the next element that is not in place.
			for (int q0 = 0; q0 < 256; q0 += 64) {
				uint64_t live = __ballot(bb[q0 + lane] != be[q0 + lane]);         // uniform
				while (live) {
					const int q = q0 + __ffsll((unsigned long long)live) - 1;
					live &= live - 1;
					for (;;) {
						const uint32_t b0 = bb[q], e0 = be[q];                       // uniform
						if (b0 == e0) break;
						uint32_t found = e0;
						for (uint32_t p0 = b0; p0 < e0; p0 += 64) {
							const uint32_t p = p0 + (uint32_t)lane;
							const bool neq = p < e0 && (int)digit(E[p < e0 ? p : b0], s) != q;
							const uint64_t m = __ballot(neq);
							if (m) { found = p0 + (uint32_t)__ffsll((unsigned long long)m) - 1u; break; }
						}
						if (lane == 0) {
							uint32_t at = found, inplace = 0;
							while (at != e0 && inplace < 8) {
								int l = (int)digit(E[at], s);
								if (l == q) { ++at; ++inplace; continue; }              // in place: ++bb[q] (ksort.h:143)
								inplace = 0;
								bb[q] = at;
								uint64_t hold = E[at], moved;
								do {
									moved = hold; hold = E[bb[l]]; E[bb[l]++] = moved;
									l = (int)digit(hold, s);
								} while (l != q);
								E[bb[q]++] = hold;
								at = bb[q];
							}
							bb[q] = at;
						}
						__syncthreads();
						if (found == e0) break;
					}
				}
			}
			__syncthreads();
			if (s) {
				const uint32_t nxt = s > 8 ? s - 8 : 0;
				for (int q0 = 0; q0 < 256; q0 += 64) {
					const int q = q0 + lane;
					const uint32_t b0 = q ? be[q - 1] : rb, e0 = be[q], cnt = e0 - b0;
					const bool big = cnt > 64;
					const uint64_t bm = __ballot(big);
					if (big) {
						const uint32_t at = sp + (uint32_t)__popcll(bm & (lane == 0 ? 0ull : (~0ull >> (64 - lane))));
						if (at < FSB_STACK) stk[at] = FsRange{b0, e0, nxt}; else *overflow = 1;
					} else if (cnt > 1) insertion(b0, e0);
					sp += (uint32_t)__popcll(bm);
				}
				if (sp > FSB_STACK) sp = FSB_STACK;
			}
			__syncthreads();
		}
	}
	__syncthreads();
	for (uint32_t i = lane; i < n; i += 64) out[beg + i] = in[beg + (uint32_t)(E[i] & 0xFFFFull)];
}

// every bucket [d_bstart[r], d_bstart[r+1]) of d_in (all of whose x share their low low_bits bits = r) in the
// reference's order, to d_out
int mcom_flag_sort_buckets(mcom_ctx *ctx, const mcom_mm128 *d_in, mcom_mm128 *d_out, const uint32_t *d_bstart, uint32_t nr, int low_bits,
                           uint32_t max_range, uint32_t *d_overflow)
{
	if (nr == 0) return MCOM_OK;
	const size_t fixed = 2 * 256 * 4 + FSB_STACK * sizeof(FsRange);
	size_t cap = max_range < 64 ? 64 : max_range;
	const size_t lds_max = 150 * 1024;
	if (cap > 65536) cap = 65536;
	if (fixed + cap * 8 > lds_max) cap = (lds_max - fixed) / 8;
	const size_t lds = fixed + cap * 8;
	MCOM_HIP(ctx, hipFuncSetAttribute((const void*)k_flag_sort_bucket, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipLaunchKernelGGL(k_flag_sort_bucket, dim3(nr), dim3(64), lds, ctx->stream, d_in, d_out, d_bstart, nr, low_bits, (uint32_t)cap, d_overflow);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

// starts of the runs of equal (x & mask) in an array sorted by that value: bstart[v] = first index with value >= v
__global__ void k_bucket_starts(const mcom_mm128 *__restrict__ rec, size_t n, uint64_t mask, uint32_t nb, uint32_t *__restrict__ bstart)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i > n) return;
	const uint64_t cur = i < n ? (rec[i].x & mask) : (uint64_t)nb;
	const uint64_t prev = i > 0 ? (rec[i - 1].x & mask) + 1 : 0;
	for (uint64_t v = prev; v <= cur && v <= nb; ++v) bstart[v] = (uint32_t)i;   // also fills the empty values in between
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
	hipLaunchKernelGGL(k_flag_sort, dim3(nr), dim3(64), lds, ctx->stream, d_rec, d_bstart, nr, (uint32_t)cap, d_overflow);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

int mcom_bucket_starts(mcom_ctx *ctx, const mcom_mm128 *d_rec, size_t n, int bits, uint32_t *d_bstart)
{
	const uint32_t nb = 1u << bits;
	hipLaunchKernelGGL(k_bucket_starts, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, ctx->stream, d_rec, n, (uint64_t)nb - 1, nb, d_bstart);
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
	MCOM_HIP(ctx, hipMemcpyAsync(&ov, d + 2, 4, hipMemcpyDeviceToHost, ctx->stream));
	MCOM_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (ov) return mcom_fail(ctx, MCOM_E_OVERFLOW, "radix sort emulation ran out of range stack");
	return MCOM_OK;
}
