// minicom_amd/csrc/sort.hip -- minimizer-record sort and grouping for gfx950 (MI355X).
//
// Replaces, for all 2^b buckets of a Stage-1 round at once, the per-bucket radix_sort_128x + run-length
// grouping + cmpcluster ordering at the top of process_bucket (reference kthread_bucket.c:391-446, :44-62;
// misc.c:22, ksort.h:108-157) and the bucket sort of the minimizer index (kthread_idx.c:126).
//
// One stable LSD radix sort (8-bit digits) over a composite key
//      C = [ bucket = x & (2^b-1) | x >> b | (L + k - aligned_pos) ]      (most .. least significant)
// on records that arrive in ascending rid order gives exactly the order the reference produces bucket by
// bucket: buckets ascending, hashes ascending inside a bucket, and inside a group of equal hashes the
// cmpcluster order (aligned position descending, rid ascending).  HBM-bound: every pass streams the
// 16-byte records once for the histogram and once for the scatter, and writes them once.
#include "mcom_dev.hpp"
#include <algorithm>
#include <cmath>

#define RS_THREADS 256
#define RS_ITEMS 8
#define RS_TILE (RS_THREADS * RS_ITEMS)   // 2048 records = 32 KiB staged in LDS (4096: two workgroups per CU, 1.3 waves per SIMD -- SQ pass)

struct KeySpec {
	int mode;       // 0: key = x (64 bits);  1: composite minimizer key;  2: low kbits of x;  3: owner rank of the bucket
	                // (b bucket bits, kbits = number of ranks; records without a minimizer get the digit `ranks`);  4: read id
	int b;          // bucket bits
	int kbits;      // 2 * k of the sketch that produced x
	int L, k_orig;  // for the aligned position of cmpcluster
	int t;          // mode 5: the MSD key is bucket (b bits) || sub (t <= 3 bits), sub = number of thresholds thr[] that x reaches;
	                // records without a minimizer get the largest key
	uint64_t thr[7];
};

// 128-bit composite key as (lo, hi); digit p = bits [8p, 8p+8)
__device__ __forceinline__ void make_key(const KeySpec &ks, uint64_t x, uint64_t y, uint64_t &lo, uint32_t &hi)
{
	if (ks.mode == 0) { lo = x; hi = 0; return; }
	if (ks.mode == 2) { lo = x & ((1ull << ks.kbits) - 1); hi = 0; return; }   // only the low kbits of x (bucket id)
	if (ks.mode == 3) { lo = x == U64MAX ? (uint64_t)ks.kbits : ((x & ((1ull << ks.b) - 1)) * (uint64_t)ks.kbits) >> ks.b; hi = 0; return; }
	if (ks.mode == 4) { lo = y >> 32; hi = 0; return; }
	if (ks.mode == 5) {
		if (x == U64MAX) { lo = (1ull << (ks.b + ks.t)) - 1; hi = 0; return; }
		uint32_t sub = 0;
#pragma unroll
		for (int j = 0; j < 7; ++j) sub += (j < (1 << ks.t) - 1 && x >= ks.thr[j]) ? 1u : 0u;
		lo = ((x & ((1ull << ks.b) - 1)) << ks.t) | sub; hi = 0; return;
	}
	if (x == U64MAX) { lo = U64MAX; hi = 0xFFFFFFFFu; return; }           // records without a minimizer sort last
	const uint64_t bucket = x & ((1ull << ks.b) - 1);
	const uint64_t K = (bucket << (ks.kbits - ks.b)) | (x >> ks.b);
	int pos = (int)(((uint32_t)y) >> 1);
	if (y & 1) pos = ks.L - pos + ks.k_orig - 2;                         // kthread_bucket.c:51-56
	const uint32_t key2 = (uint32_t)(ks.L + ks.k_orig - pos) & 0x1FFu;    // descending aligned position
	lo = (K << 9) | key2;
	hi = (uint32_t)(K >> 55);
}
__device__ __forceinline__ uint32_t key_digit(uint64_t lo, uint32_t hi, int pass)
{
	return pass < 8 ? (uint32_t)(lo >> (8 * pass)) & 255u : (hi >> (8 * (pass - 8))) & 255u;
}

// ---- per-tile digit histogram: hist[digit * nblocks + block] ------------------------------------
__global__ __launch_bounds__(RS_THREADS) void k_radix_hist(const mcom_mm128 *__restrict__ in, size_t n, KeySpec ks, int pass,
                                                           uint32_t *__restrict__ hist, uint32_t nblocks)
{
	__shared__ uint32_t h[256];
	h[threadIdx.x] = 0;
	__syncthreads();
	const size_t base = (size_t)blockIdx.x * RS_TILE;
#pragma unroll 4
	for (int it = 0; it < RS_ITEMS; ++it) {
		const size_t i = base + (size_t)it * RS_THREADS + threadIdx.x;
		if (i < n) {
			const mcom_mm128 r = in[i];
			uint64_t lo; uint32_t hi; make_key(ks, r.x, r.y, lo, hi);
			atomicAdd(&h[key_digit(lo, hi, pass)], 1u);
		}
	}
	__syncthreads();
	hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// ---- stable scatter of one tile --------------------------------------------------------------------
// wave w of the block owns records [w*1024, w*1024+1024) of the tile, 16 chunks of 64 (one per lane).
__global__ __launch_bounds__(RS_THREADS) void k_radix_scatter(const mcom_mm128 *__restrict__ in, mcom_mm128 *__restrict__ out, size_t n,
                                                              KeySpec ks, int pass, const uint32_t *__restrict__ offs, uint32_t nblocks)
{
	__shared__ mcom_mm128 stage[RS_TILE];
	__shared__ uint32_t wcnt[4][256];        // per-wave digit counts, then per-wave start inside the tile
	__shared__ uint32_t tstart[256];         // first local slot of each digit in the staged tile
	__shared__ uint32_t gofs[256];           // global slot of that first local slot
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	for (int q = threadIdx.x; q < 4 * 256; q += RS_THREADS) (&wcnt[0][0])[q] = 0;
	__syncthreads();

	const size_t base = (size_t)blockIdx.x * RS_TILE + (size_t)wv * (RS_TILE / 4);
	mcom_mm128 r[RS_ITEMS];
	uint16_t rank[RS_ITEMS];
	uint8_t dig[RS_ITEMS];
	const uint64_t lt = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
	for (int c = 0; c < RS_ITEMS; ++c) {
		const size_t i = base + (size_t)c * 64 + lane;
		const bool valid = i < n;
		uint32_t d = 0;
		if (valid) {
			r[c] = in[i];
			uint64_t lo; uint32_t hi; make_key(ks, r[c].x, r[c].y, lo, hi);
			d = key_digit(lo, hi, pass);
		}
		uint64_t peers = __ballot(valid);
#pragma unroll
		for (int bit = 0; bit < 8; ++bit) {
			const bool one = (d >> bit) & 1;
			const uint64_t bl = __ballot(one);
			peers &= one ? bl : ~bl;
		}
		uint32_t old = 0;
		const int leader = __ffsll((unsigned long long)peers) - 1;
		if (valid && lane == leader) { old = wcnt[wv][d]; wcnt[wv][d] = old + (uint32_t)__popcll(peers); }
		old = __shfl(old, leader < 0 ? 0 : leader, 64);
		rank[c] = (uint16_t)(old + (uint32_t)__popcll(peers & lt));
		dig[c] = (uint8_t)d;
	}
	__syncthreads();
	{   // digit totals -> exclusive prefix over digits (tstart), per-wave starts, global starts
		const int d = threadIdx.x;
		const uint32_t c0 = wcnt[0][d], c1 = wcnt[1][d], c2 = wcnt[2][d], c3 = wcnt[3][d];
		const uint32_t tot = c0 + c1 + c2 + c3;
		// block exclusive scan of tot over 256 threads
		uint32_t v = tot;
#pragma unroll
		for (int s = 1; s < 64; s <<= 1) { uint32_t t = __shfl_up(v, s, 64); if (lane >= s) v += t; }
		__shared__ uint32_t wsum[4];
		if (lane == 63) wsum[wv] = v;
		__syncthreads();
		uint32_t add = 0;
		for (int q = 0; q < wv; ++q) add += wsum[q];
		const uint32_t excl = v + add - tot;
		tstart[d] = excl;
		wcnt[0][d] = excl; wcnt[1][d] = excl + c0; wcnt[2][d] = excl + c0 + c1; wcnt[3][d] = excl + c0 + c1 + c2;
		gofs[d] = offs[(size_t)d * nblocks + blockIdx.x];
	}
	__syncthreads();
#pragma unroll
	for (int c = 0; c < RS_ITEMS; ++c) {
		const size_t i = base + (size_t)c * 64 + lane;
		if (i < n) stage[wcnt[wv][dig[c]] + rank[c]] = r[c];
	}
	__syncthreads();
	const size_t tile_base = (size_t)blockIdx.x * RS_TILE;
	const uint32_t cnt = (uint32_t)((n - tile_base) < (size_t)RS_TILE ? (n - tile_base) : (size_t)RS_TILE);
	for (uint32_t q = threadIdx.x; q < cnt; q += RS_THREADS) {
		const mcom_mm128 v = stage[q];
		uint64_t lo; uint32_t hi; make_key(ks, v.x, v.y, lo, hi);
		const uint32_t d = key_digit(lo, hi, pass);
		out[(size_t)gofs[d] + (q - tstart[d])] = v;
	}
}


// ---- bucket sort in three passes --------------------------------------------------------------------------------------
// The composite key of a Stage-1 round is 71 bits wide (62 of hash), nine 8-bit LSD passes of 48 bytes per record.  But
// only its top bits need a global pass: two stable passes on the MSD key [bucket | top t bits of the hash] cut the
// records into 2^(14+t) segments of about 1500, in final order; what is left of the key is settled inside a segment by
// one workgroup, in LDS: keys as 64-bit words [x >> b | aligned position], a stable 8-bit LSD sort of 16-bit indices (the
// per-wave ballot ranking of k_radix_scatter), then the records are written once, in order.  3 passes of the records
// instead of 9.  A segment larger than the LDS arrays (a repeat: tens of thousands of reads with ONE minimizer, which no
// prefix can split) is left alone and sorted afterwards by the nine-pass sort, together with the others of its kind.
#define SS_THREADS 256
#define SS_CAP 4096
#define SS_CAP_SMALL 2048                    // half the LDS: six workgroups per CU instead of three (2.6 waves per SIMD with the large form, SQ pass)

__device__ __forceinline__ uint32_t msd_key(const KeySpec &ks, uint64_t x)
{
	uint64_t lo; uint32_t hi; make_key(ks, x, 0, lo, hi);
	return (uint32_t)lo;
}
// seg_start[k] = first record whose MSD key is >= k (records sorted by it); seg_start[nseg] = n.  One thread per k and a binary search
// (round 5; before: one thread per record writing the starts between its neighbour's key and its own -- a rank of several GPUs holds one
// stretch of the keys, and the one thread in front of it wrote tens of thousands of starts in a row: 0.6 ms per launch)
__global__ void k_seg_bounds(const mcom_mm128 *__restrict__ s, size_t n, KeySpec ks, uint32_t nseg, uint32_t *__restrict__ seg_start)
{
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k > nseg) return;
	size_t lo = 0, hi = n;
	while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (msd_key(ks, s[mid].x) < k) lo = mid + 1; else hi = mid; }
	seg_start[k] = (uint32_t)lo;
}

template <int CAP>
__global__ __launch_bounds__(SS_THREADS) void k_segment_sort(const mcom_mm128 *__restrict__ in, mcom_mm128 *__restrict__ out, const uint32_t *__restrict__ seg_start,
                                                             KeySpec full, int sig_bits, uint32_t cap, uint32_t *__restrict__ ovf_count, uint2 *__restrict__ ovf_list,
                                                             const uint2 *__restrict__ seg_list = nullptr /* the segments by their bounds instead of seg_start */)
{
	constexpr int SS_ITEMS = CAP / SS_THREADS;
	__shared__ uint64_t keys[CAP];
	__shared__ uint16_t idx[2][CAP];
	__shared__ uint32_t wcnt[SS_THREADS / 64][256];
	__shared__ uint32_t wsum[SS_THREADS / 64];
	const uint32_t s0 = seg_list ? seg_list[blockIdx.x].x : seg_start[blockIdx.x], n = (seg_list ? seg_list[blockIdx.x].y : seg_start[blockIdx.x + 1]) - s0;
	if (n == 0) return;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	if (n == 1) { if (tid == 0) out[s0] = in[s0]; return; }
	if (n > cap) {                                                             // left in place; the caller sorts these segments afterwards
		if (tid == 0) { const uint32_t at = atomicAdd(ovf_count, 1u); ovf_list[at] = make_uint2(s0, s0 + n); }
		for (uint32_t i = tid; i < n; i += SS_THREADS) out[s0 + i] = in[s0 + i];
		return;
	}
	for (uint32_t i = tid; i < n; i += SS_THREADS) {
		const mcom_mm128 r = in[s0 + i];
		uint64_t lo; uint32_t hi; make_key(full, r.x, r.y, lo, hi);           // [bucket | x >> b | position key]: the low sig_bits are what is left to sort
		keys[i] = lo;
		idx[0][i] = (uint16_t)i;
	}
	__syncthreads();
	{   // digits above the highest bit in which two keys of the segment differ need no pass (the hashes of one sub-range share their top bits)
		uint64_t diff = 0;
		const uint64_t k0 = keys[0];
		for (uint32_t i = tid; i < n; i += SS_THREADS) diff |= keys[i] ^ k0;
		for (int o = 32; o; o >>= 1) diff |= __shfl_xor(diff, o);
		__shared__ uint64_t wdiff[SS_THREADS / 64];
		if (lane == 0) wdiff[wv] = diff;
		__syncthreads();
		diff = 0;
		for (int q = 0; q < SS_THREADS / 64; ++q) diff |= wdiff[q];
		const int top = diff ? 64 - __clzll((long long)diff) : 0;
		if (top < sig_bits) sig_bits = top;
	}
	// wave w ranks elements [w * per, (w + 1) * per), 64 at a time
	const uint32_t per = ((n + SS_THREADS - 1) / SS_THREADS) * 64;
	const int iters = (int)(per >> 6);
	const uint64_t lt = lane ? (~0ull >> (64 - lane)) : 0ull;
	int cur = 0;
	for (int shift = 0; shift < sig_bits; shift += 8) {
		for (int q = tid; q < (SS_THREADS / 64) * 256; q += SS_THREADS) (&wcnt[0][0])[q] = 0;
		__syncthreads();
		uint16_t id[SS_ITEMS], rank[SS_ITEMS]; uint8_t dig[SS_ITEMS];
#pragma unroll
		for (int c = 0; c < SS_ITEMS; ++c) if (c < iters) {                      // wave-uniform: the arrays stay in registers
			const uint32_t i = (uint32_t)wv * per + (uint32_t)c * 64 + lane;
			const bool valid = i < n;
			uint32_t d = 0;
			id[c] = 0;
			if (valid) { id[c] = idx[cur][i]; d = (uint32_t)(keys[id[c]] >> shift) & 255u; }
			uint64_t peers = __ballot(valid);
#pragma unroll
			for (int bit = 0; bit < 8; ++bit) {
				const bool one = (d >> bit) & 1;
				const uint64_t bl = __ballot(one);
				peers &= one ? bl : ~bl;
			}
			uint32_t old = 0;
			const int leader = __ffsll((unsigned long long)peers) - 1;
			if (valid && lane == leader) { old = wcnt[wv][d]; wcnt[wv][d] = old + (uint32_t)__popcll(peers); }
			old = __shfl(old, leader < 0 ? 0 : leader, 64);
			rank[c] = (uint16_t)(old + (uint32_t)__popcll(peers & lt));
			dig[c] = (uint8_t)d;
		}
		__syncthreads();
		{   // digit totals -> exclusive prefix over the 256 digits -> start of every wave's share of a digit
			const int d = tid;
			uint32_t c[SS_THREADS / 64], tot = 0;
#pragma unroll
			for (int w = 0; w < SS_THREADS / 64; ++w) { c[w] = wcnt[w][d]; tot += c[w]; }
			uint32_t v = tot;
#pragma unroll
			for (int s = 1; s < 64; s <<= 1) { const uint32_t t = __shfl_up(v, s, 64); if (lane >= s) v += t; }
			if (lane == 63) wsum[wv] = v;
			__syncthreads();
			uint32_t add = 0;
			for (int q = 0; q < wv; ++q) add += wsum[q];
			uint32_t run = v + add - tot;
#pragma unroll
			for (int w = 0; w < SS_THREADS / 64; ++w) { wcnt[w][d] = run; run += c[w]; }
		}
		__syncthreads();
#pragma unroll
		for (int c = 0; c < SS_ITEMS; ++c) if (c < iters) {
			const uint32_t i = (uint32_t)wv * per + (uint32_t)c * 64 + lane;
			if (i < n) idx[cur ^ 1][wcnt[wv][dig[c]] + rank[c]] = id[c];
		}
		__syncthreads();
		cur ^= 1;
	}
	__syncthreads();
	for (uint32_t j = tid; j < n; j += SS_THREADS) out[s0 + j] = in[s0 + idx[cur][j]];
}

// gather / scatter of the oversized segments: compact[dst[q] + i] <-> arr[list[q].x + i]
__global__ void k_seg_move(mcom_mm128 *__restrict__ arr, mcom_mm128 *__restrict__ compact, const uint2 *__restrict__ list, const uint32_t *__restrict__ dst, int back)
{
	const uint2 sg = list[blockIdx.x];
	const uint32_t d0 = dst[blockIdx.x];
	for (uint32_t i = threadIdx.x; i < sg.y - sg.x; i += blockDim.x) { if (back) arr[sg.x + i] = compact[d0 + i]; else compact[d0 + i] = arr[sg.x + i]; }
}

// (the exclusive scans live in scan.hip: one launch each)
static inline int scan_u32(mcom_ctx *ctx, const uint32_t *in, uint32_t *out, size_t n, uint32_t *scratch) { return mcom_scan_u32(ctx, in, out, n, scratch); }
static inline size_t scan_scratch_elems(size_t n) { return mcom_scan_scratch_elems(n); }

// ---- the sort driver -------------------------------------------------------------------------------
// a and tmp are ping-pong buffers of n records; on return *sorted points at the one holding the result
static int radix_sort_records(mcom_ctx *ctx, mcom_mm128 *a, mcom_mm128 *tmp, size_t n, const KeySpec &ks, int passes,
                              uint32_t *hist, uint32_t *scratch, mcom_mm128 **sorted)
{
	const uint32_t nblocks = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
	mcom_mm128 *src = a, *dst = tmp;
	for (int p = 0; p < passes; ++p) {
		McomProfScope ps_(ctx, PROF_RADIX_PASS);
		MCOM_LAUNCH(k_radix_hist, dim3(nblocks), dim3(RS_THREADS), 0, ctx->stream, src, n, ks, p, hist, nblocks);
		MCOM_LAUNCH_CHECK(ctx);
		int rc = scan_u32(ctx, hist, hist, (size_t)256 * nblocks, scratch);
		if (rc) return rc;
		MCOM_LAUNCH(k_radix_scatter, dim3(nblocks), dim3(RS_THREADS), 0, ctx->stream, src, dst, n, ks, p, hist, nblocks);
		MCOM_LAUNCH_CHECK(ctx);
		mcom_mm128 *t = src; src = dst; dst = t;
	}
	*sorted = src;
	return MCOM_OK;
}

struct SortWs { mcom_mm128 *tmp; uint32_t *hist, *scratch; size_t bytes; };
static size_t sort_ws_layout(size_t n, SortWs *w, char *base)
{
	const size_t nblocks = (n + RS_TILE - 1) / RS_TILE;
	size_t off = 0;
	auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
	size_t o_tmp = take(n * sizeof(mcom_mm128));
	size_t o_hist = take((size_t)256 * nblocks * 4);
	size_t o_scr = take(scan_scratch_elems((size_t)256 * nblocks) * 4 + 1024);
	if (w && base) { w->tmp = (mcom_mm128*)(base + o_tmp); w->hist = (uint32_t*)(base + o_hist); w->scratch = (uint32_t*)(base + o_scr); }
	return off;
}

// internal: stable sort of records by the low `bits` bits of x; ws must hold mcom_sort_ws_bytes(n) bytes.
size_t mcom_sort_ws_bytes(size_t n) { return sort_ws_layout(n, nullptr, nullptr); }
int mcom_sort_by_x(mcom_ctx *ctx, mcom_mm128 *d_a, size_t n, int bits, void *ws)
{
	if (n == 0) return MCOM_OK;
	SortWs w; sort_ws_layout(n, &w, (char*)ws);
	KeySpec ks{0, 0, 64, 0, 0, 0, {0, 0, 0, 0, 0, 0, 0}};
	mcom_mm128 *res = nullptr;
	int rc = radix_sort_records(ctx, d_a, w.tmp, n, ks, (bits + 7) / 8, w.hist, w.scratch, &res);
	if (rc) return rc;
	if (res != d_a) MCOM_HIP(ctx, hipMemcpyAsync(d_a, res, n * sizeof(mcom_mm128), hipMemcpyDeviceToDevice, ctx->stream));
	return MCOM_OK;
}

// internal: stable sort by x & (2^bits - 1) only (records of one bucket keep their input order)
int mcom_sort_by_low_bits(mcom_ctx *ctx, mcom_mm128 *d_a, size_t n, int bits, void *ws)
{
	if (n == 0) return MCOM_OK;
	SortWs w; sort_ws_layout(n, &w, (char*)ws);
	KeySpec ks{2, 0, bits, 0, 0, 0, {0, 0, 0, 0, 0, 0, 0}};
	mcom_mm128 *res = nullptr;
	int rc = radix_sort_records(ctx, d_a, w.tmp, n, ks, (bits + 7) / 8, w.hist, w.scratch, &res);
	if (rc) return rc;
	if (res != d_a) MCOM_HIP(ctx, hipMemcpyAsync(d_a, res, n * sizeof(mcom_mm128), hipMemcpyDeviceToDevice, ctx->stream));
	return MCOM_OK;
}

// the same from a source that is left alone, through d_a and the workspace's record buffer, the result IN THE WORKSPACE (*d_sorted):
// what comes next (the index's bucket sort) reads it there and writes d_a -- no copy in front of the passes, none behind them
int mcom_sort_by_low_bits_into_ws(mcom_ctx *ctx, const mcom_mm128 *d_src, mcom_mm128 *d_a, size_t n, int bits, void *ws, mcom_mm128 **d_sorted)
{
	SortWs w; sort_ws_layout(n, &w, (char*)ws);
	*d_sorted = w.tmp;
	if (n == 0) return MCOM_OK;
	const KeySpec ks{2, 0, bits, 0, 0, 0, {0, 0, 0, 0, 0, 0, 0}};
	const int passes = (bits + 7) / 8;
	const uint32_t nblocks = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
	const mcom_mm128 *src = d_src;
	mcom_mm128 *dst = (passes & 1) ? w.tmp : d_a;                            // so that the last pass lands in the workspace
	for (int p = 0; p < passes; ++p) {
		McomProfScope ps_(ctx, PROF_RADIX_PASS);
		MCOM_LAUNCH(k_radix_hist, dim3(nblocks), dim3(RS_THREADS), 0, ctx->stream, src, n, ks, p, w.hist, nblocks);
		MCOM_LAUNCH_CHECK(ctx);
		int rc = scan_u32(ctx, w.hist, w.hist, (size_t)256 * nblocks, w.scratch);
		if (rc) return rc;
		MCOM_LAUNCH(k_radix_scatter, dim3(nblocks), dim3(RS_THREADS), 0, ctx->stream, src, dst, n, ks, p, w.hist, nblocks);
		MCOM_LAUNCH_CHECK(ctx);
		src = dst; dst = dst == w.tmp ? d_a : w.tmp;
	}
	return MCOM_OK;
}

// ---- multi-GPU exchange -------------------------------------------------------------------
// Bucket beta = x & (2^b - 1) belongs to rank (beta * ranks) >> b: contiguous bucket ranges in rank order, so that
// "rank 0's groups, then rank 1's, ..." is the reference's visiting order (buckets ascending, kthread_bucket.c:531-560).
// One stable pass of the radix kernels on the owner digit; the scanned histogram gives the per-owner counts.
extern "C" int mcom_partition_by_owner(mcom_ctx *ctx, const mcom_mm128 *d_rec, size_t n, int b, int ranks, mcom_mm128 *d_out, uint64_t *h_counts)
{
	if (!ctx || !h_counts || ranks < 1 || ranks > 254 || b < 1 || b > 24) return MCOM_E_ARG;
	for (int q = 0; q < ranks; ++q) h_counts[q] = 0;
	if (n == 0) return MCOM_OK;
	if (!d_rec || !d_out) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	if (n >= (1ull << 32)) return mcom_fail(ctx, MCOM_E_ARG, "more than 2^32-1 records");
	const uint32_t nblocks = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
	const size_t hist_b = ((size_t)256 * nblocks * 4 + 255) & ~(size_t)255;
	int rc = mcom_ws_reserve(ctx, hist_b + scan_scratch_elems((size_t)256 * nblocks) * 4 + 1024);
	if (rc) return rc;
	uint32_t *hist = (uint32_t*)ctx->ws, *scratch = (uint32_t*)((char*)ctx->ws + hist_b);
	KeySpec ks{3, b, ranks, 0, 0, 0, {0, 0, 0, 0, 0, 0, 0}};
	{
		McomProfScope ps_(ctx, PROF_RADIX_PASS);
		MCOM_LAUNCH(k_radix_hist, dim3(nblocks), dim3(RS_THREADS), 0, ctx->stream, d_rec, n, ks, 0, hist, nblocks);
		MCOM_LAUNCH_CHECK(ctx);
		if ((rc = scan_u32(ctx, hist, hist, (size_t)256 * nblocks, scratch))) return rc;
		MCOM_LAUNCH(k_radix_scatter, dim3(nblocks), dim3(RS_THREADS), 0, ctx->stream, d_rec, d_out, n, ks, 0, hist, nblocks);
		MCOM_LAUNCH_CHECK(ctx);
	}
	std::vector<uint32_t> start((size_t)ranks + 1);
	for (int q = 0; q <= ranks; ++q) MCOM_HIP(ctx, mcom_d2h_async(ctx, &start[q], hist + (size_t)q * nblocks, 4));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	for (int q = 0; q < ranks; ++q) h_counts[q] = start[q + 1] - start[q];
	return MCOM_OK;
}

// records by read id ascending (stable): what a rank receives from R senders, each part in rid order, becomes one rid-ordered list
extern "C" int mcom_sort_by_rid(mcom_ctx *ctx, mcom_mm128 *d_a, size_t n)
{
	if (!ctx) return MCOM_E_ARG;
	if (n == 0) return MCOM_OK;
	if (!d_a) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	if (n >= (1ull << 32)) return mcom_fail(ctx, MCOM_E_ARG, "more than 2^32-1 records");
	int rc = mcom_ws_reserve(ctx, sort_ws_layout(n, nullptr, nullptr));
	if (rc) return rc;
	SortWs w; sort_ws_layout(n, &w, (char*)ctx->ws);
	KeySpec ks{4, 0, 32, 0, 0, 0, {0, 0, 0, 0, 0, 0, 0}};
	mcom_mm128 *res = nullptr;
	if ((rc = radix_sort_records(ctx, d_a, w.tmp, n, ks, 4, w.hist, w.scratch, &res))) return rc;
	if (res != d_a) MCOM_HIP(ctx, hipMemcpyAsync(d_a, res, n * sizeof(mcom_mm128), hipMemcpyDeviceToDevice, ctx->stream));
	return MCOM_OK;
}

// ---- records already grouped (the members of one merged contig, of one contig) need no global pass at all ------------------------
// Groups goff[0..ng] of consecutive records, each to be sorted by x (whose high bits are the group's number, so that x
// ascends from group to group): whole groups are packed into tiles of about 1300 records (3000 until round 4) and every tile goes through
// k_segment_sort -- its "highest differing bit" test makes it sort just the key bits plus the few bits in which the group
// numbers of one tile differ.  One read and one write of the records where the LSD sort of (group | key) made five passes.
// A tile swollen by a group of thousands of records goes through the global passes instead.
__global__ void k_tile_starts(const uint64_t *__restrict__ goff, size_t ng, size_t n, uint32_t ntiles, uint32_t *__restrict__ start)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t > ntiles) return;
	if (t == ntiles) { start[t] = (uint32_t)n; return; }
	const uint64_t target = (uint64_t)t * (uint64_t)MCOM_GROUP_TILE;
	size_t lo = 0, hi = ng;                                                  // first group boundary at or behind the target
	while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (goff[mid] < target) lo = mid + 1; else hi = mid; }
	start[t] = (uint32_t)(goff[lo] < n ? goff[lo] : n);
}
int mcom_sort_groups_by_x(mcom_ctx *ctx, mcom_mm128 *d_in, mcom_mm128 *d_out, size_t n, const uint64_t *d_goff, size_t ng, int bits,
                          uint32_t *d_scratch /* MCOM_GROUP_SCRATCH(n) uint32, 8-byte aligned */)
{
	if (n == 0) return MCOM_OK;
	const uint32_t ntiles = (uint32_t)(n / MCOM_GROUP_TILE + 1);
	uint32_t *start = d_scratch, *ovf0 = d_scratch + ((ntiles + 3) & ~1u);    // [count, pad, list of ntiles pairs, ntiles offsets]
	uint2 *ovf_list = (uint2*)(ovf0 + 2);
	uint32_t *ovf_dst = ovf0 + 2 + 2 * (size_t)ntiles;
	MCOM_LAUNCH(k_tile_starts, dim3((ntiles + 1 + 255) / 256), dim3(256), 0, ctx->stream, d_goff, ng, n, ntiles, start);
	uint32_t *ovf = (uint32_t*)mcom_zeroed(ctx, ovf0, 4);                       // the count (a zeroed word of the context's pool)
	if (!ovf) return mcom_fail(ctx, MCOM_E_HIP, "clear");
	const KeySpec ks{0, 0, 64, 0, 0, 0, {0, 0, 0, 0, 0, 0, 0}};
	{
		McomProfScope ps_(ctx, PROF_RADIX_PASS);
		const uint32_t scap = ctx->seg_cap ? ctx->seg_cap : (uint32_t)SS_CAP_SMALL;
		if (scap <= SS_CAP_SMALL) MCOM_LAUNCH(k_segment_sort<SS_CAP_SMALL>, dim3(ntiles), dim3(SS_THREADS), 0, ctx->stream, d_in, d_out, start, ks, 64, scap, ovf, ovf_list);
		else MCOM_LAUNCH(k_segment_sort<SS_CAP>, dim3(ntiles), dim3(SS_THREADS), 0, ctx->stream, d_in, d_out, start, ks, 64, scap, ovf, ovf_list);
	}
	MCOM_LAUNCH_CHECK(ctx);
	uint32_t novf = 0;
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &novf, ovf, 4));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	if (!novf) return MCOM_OK;
	// tiles that hold a group of thousands of records: gathered, sorted by the whole of x with the global passes (x ascends from
	// group to group, so the tiles stay apart and their groups in place), put into the output
	std::vector<uint2> list(novf);
	MCOM_HIP(ctx, hipMemcpy(list.data(), ovf_list, (size_t)novf * sizeof(uint2), hipMemcpyDeviceToHost));
	std::sort(list.begin(), list.end(), [](const uint2 &a, const uint2 &c) { return a.x < c.x; });
	std::vector<uint32_t> dst_off(novf);
	size_t m = 0;
	for (uint32_t q = 0; q < novf; ++q) { dst_off[q] = (uint32_t)m; m += list[q].y - list[q].x; }
	MCOM_HIP(ctx, hipMemcpyAsync(ovf_list, list.data(), (size_t)novf * sizeof(uint2), hipMemcpyHostToDevice, ctx->stream));
	MCOM_HIP(ctx, hipMemcpyAsync(ovf_dst, dst_off.data(), (size_t)novf * 4, hipMemcpyHostToDevice, ctx->stream));
	void *ws2 = nullptr;
	const size_t rec_b = (m * sizeof(mcom_mm128) + 255) & ~(size_t)255;
	if (mcom_dmalloc(&ws2, rec_b + sort_ws_layout(m, nullptr, nullptr)) != hipSuccess) return mcom_fail(ctx, MCOM_E_NOMEM, "workspace for %zu records of oversized groups", m);
	mcom_mm128 *compact = (mcom_mm128*)ws2;
	SortWs w2; sort_ws_layout(m, &w2, (char*)ws2 + rec_b);
	MCOM_LAUNCH(k_seg_move, dim3(novf), dim3(256), 0, ctx->stream, d_in, compact, ovf_list, ovf_dst, 0);
	mcom_mm128 *res = nullptr;
	int rc = radix_sort_records(ctx, compact, w2.tmp, m, ks, (bits + 7) / 8, w2.hist, w2.scratch, &res);
	if (!rc) MCOM_LAUNCH(k_seg_move, dim3(novf), dim3(256), 0, ctx->stream, d_out, res, ovf_list, ovf_dst, 1);
	hipError_t e2 = mcom_stream_sync(ctx);
	mcom_dfree(ws2);
	if (rc) return rc;
	if (e2 != hipSuccess) return mcom_fail(ctx, MCOM_E_HIP, "oversized groups: %s", hipGetErrorString(e2));
	return MCOM_OK;
}

// a5: radix_sort_128x (misc.c:22).  Stable, so equal keys keep their input order; the reference is stable
// only up to 64 elements (ksort.h:155) and leaves equal keys of larger arrays in an order that depends on
// its in-place cycle walk.
extern "C" int mcom_radix_sort_128x(mcom_ctx *ctx, mcom_mm128 *d_a, size_t n)
{
	if (!ctx) return MCOM_E_ARG;
	if (n == 0) return MCOM_OK;
	if (!d_a) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	if (n >= (1ull << 32)) return mcom_fail(ctx, MCOM_E_ARG, "more than 2^32-1 records");
	const size_t need = sort_ws_layout(n, nullptr, nullptr);
	int rc = mcom_ws_reserve(ctx, need);
	if (rc) return rc;
	SortWs w; sort_ws_layout(n, &w, (char*)ctx->ws);
	KeySpec ks{0, 0, 64, 0, 0, 0, {0, 0, 0, 0, 0, 0, 0}};
	mcom_mm128 *res = nullptr;
	rc = radix_sort_records(ctx, d_a, w.tmp, n, ks, 8, w.hist, w.scratch, &res);
	if (rc) return rc;
	if (res != d_a) MCOM_HIP(ctx, hipMemcpyAsync(d_a, res, n * sizeof(mcom_mm128), hipMemcpyDeviceToDevice, ctx->stream));
	return MCOM_OK;
}

// ---- grouping ------------------------------------------------------------------------------------------
// flags per sorted record: bit0 single (group of one), bit1 member of a group >= 2, bit2 first member of such a group
__global__ void k_group_flags(const mcom_mm128 *__restrict__ s, size_t n, uint32_t *__restrict__ f_single,
                              uint32_t *__restrict__ f_member, uint32_t *__restrict__ f_head)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint64_t x = s[i].x;
	const bool valid = x != U64MAX;
	const bool eq_prev = i > 0 && s[i - 1].x == x, eq_next = i + 1 < n && s[i + 1].x == x;
	const bool single = valid && !eq_prev && !eq_next, member = valid && (eq_prev || eq_next);
	f_single[i] = single; f_member[i] = member; f_head[i] = member && !eq_prev;
}
__global__ void k_group_emit(const mcom_mm128 *__restrict__ s, size_t n, const uint32_t *__restrict__ p_single,
                             const uint32_t *__restrict__ p_member, const uint32_t *__restrict__ p_head,
                             uint32_t *__restrict__ singles, uint32_t *__restrict__ single_ord, uint64_t *__restrict__ members,
                             uint32_t *__restrict__ group_off, uint64_t *__restrict__ counts)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const mcom_mm128 r = s[i];
	const bool valid = r.x != U64MAX;
	const bool eq_prev = i > 0 && s[i - 1].x == r.x, eq_next = i + 1 < n && s[i + 1].x == r.x;
	const bool single = valid && !eq_prev && !eq_next, member = valid && (eq_prev || eq_next);
	if (single) { singles[p_single[i]] = (uint32_t)(r.y >> 32); if (single_ord) single_ord[p_single[i]] = p_head[i]; }
	if (member) { members[p_member[i]] = r.y; if (!eq_prev) group_off[p_head[i]] = p_member[i]; }
	if (i == n - 1) {
		const uint32_t ns = p_single[i] + (single ? 1u : 0u), nm = p_member[i] + (member ? 1u : 0u), ng = p_head[i] + ((member && !eq_prev) ? 1u : 0u);
		group_off[ng] = nm;
		counts[1] = ns; counts[2] = ng; counts[3] = nm;
	}
}
__global__ void k_count_valid(const mcom_mm128 *__restrict__ s, size_t n, uint64_t *__restrict__ counts)
{
	// sorted: the records without a minimizer are a suffix; binary search its start
	if (threadIdx.x || blockIdx.x) return;
	size_t lo = 0, hi = n;
	while (lo < hi) { size_t mid = (lo + hi) / 2; if (s[mid].x == U64MAX) hi = mid; else lo = mid + 1; }
	counts[0] = lo;
}

extern "C" int mcom_sort_group(mcom_ctx *ctx, const mcom_mm128 *d_rec, size_t n, int L, int k_orig, int kmer, int b,
                               mcom_mm128 *d_sorted, uint32_t *d_singles, uint32_t *d_single_ord, uint64_t *d_members,
                               uint32_t *d_group_off, uint64_t *h_counts)
{
	if (!ctx) return MCOM_E_ARG;
	if (!h_counts) return mcom_fail(ctx, MCOM_E_ARG, "h_counts is null");
	h_counts[0] = h_counts[1] = h_counts[2] = h_counts[3] = 0;
	if (n == 0) return MCOM_OK;
	if (!d_rec || !d_sorted || !d_singles || !d_members || !d_group_off) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	if (n >= (1ull << 32)) return mcom_fail(ctx, MCOM_E_ARG, "more than 2^32-1 records");
	if (kmer < 1 || kmer > 31 || k_orig < 1 || k_orig > 31 || L < 1 || L > 256) return mcom_fail(ctx, MCOM_E_ARG, "bad k/L");
	if (b < 1 || b > 2 * kmer || b > 24) return mcom_fail(ctx, MCOM_E_ARG, "bucket bits %d not in 1..min(24, 2k)", b);
	// workspace: sort buffers + three flag/prefix arrays + counts
	const size_t sort_bytes = sort_ws_layout(n, nullptr, nullptr);
	const size_t flag_bytes = ((n * 4 + 255) & ~(size_t)255);
	const size_t scr_bytes = ((scan_scratch_elems(n) * 4 + 1024 + 255) & ~(size_t)255);
	// the sort: passes on the MSD key [bucket | top t bits of the hash] until a segment holds about 1500 records, the rest of the
	// key inside the segments (k_segment_sort); the last global pass lands in the workspace, the segment sort in d_sorted
	int t = 0;
	while ((n >> (b + t)) > 2048 && t < 3 && b + t < 24) ++t;
	const int B = b + t, passes = (B + 7) / 8;
	const uint32_t nseg = 1u << B;
	const size_t segs_b = (((size_t)nseg + 2) * 4 + 255) & ~(size_t)255, list_b = ((size_t)nseg * 8 + 255) & ~(size_t)255, dst_b = ((size_t)nseg * 4 + 255) & ~(size_t)255;
	const size_t need = sort_bytes + 3 * flag_bytes + scr_bytes + 256 + segs_b + list_b + dst_b + 256;
	int rc = mcom_ws_reserve(ctx, need);
	if (rc) return rc;
	char *base = (char*)ctx->ws;
	SortWs w; sort_ws_layout(n, &w, base);
	uint32_t *f0 = (uint32_t*)(base + sort_bytes), *f1 = (uint32_t*)(base + sort_bytes + flag_bytes), *f2 = (uint32_t*)(base + sort_bytes + 2 * flag_bytes);
	uint32_t *scr = (uint32_t*)(base + sort_bytes + 3 * flag_bytes);
	uint64_t *d_counts = (uint64_t*)(base + sort_bytes + 3 * flag_bytes + scr_bytes);
	char *segbase = base + sort_bytes + 3 * flag_bytes + scr_bytes + 256;
	uint32_t *seg_start = (uint32_t*)segbase, *ovf_dst = (uint32_t*)(segbase + segs_b + list_b), *ovf_count = (uint32_t*)(segbase + segs_b + list_b + dst_b);
	uint2 *ovf_list = (uint2*)(segbase + segs_b);

	const KeySpec full{1, b, 2 * kmer, L, k_orig, 0, {0, 0, 0, 0, 0, 0, 0}};
	KeySpec msd{5, b, 2 * kmer, L, k_orig, t, {0, 0, 0, 0, 0, 0, 0}};
	{
		// Where to cut a bucket into 2^t sub-ranges of the hash: x is the MINIMUM of the hashes of a read's m = L - kmer + 1 k-mers,
		// far from uniform (its top bits are nearly always zero): P(x > v) = (1 - v / 2^2k)^m, so the j-th of 2^t equal shares
		// ends at 2^2k (1 - (1 - j / 2^t)^(1/m)).  Any ascending thresholds give a correct sort; these give even segments.
		const double m = (double)std::max(1, L - kmer + 1), top = ldexp(1.0, 2 * kmer);
		for (int j = 1; j < (1 << t); ++j) msd.thr[j - 1] = (uint64_t)(top * (1.0 - pow(1.0 - (double)j / (double)(1 << t), 1.0 / m)));
	}
	const int sig_bits = 9 + 2 * kmer - b;                  // everything above the bucket bits: the sub-ranges are cut by value, not by bit
	{
		const uint32_t nblocks = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
		const mcom_mm128 *src = d_rec;
		mcom_mm128 *dst = (passes & 1) ? w.tmp : d_sorted;
		for (int p = 0; p < passes; ++p) {
			McomProfScope ps_(ctx, PROF_RADIX_PASS);
			MCOM_LAUNCH(k_radix_hist, dim3(nblocks), dim3(RS_THREADS), 0, ctx->stream, src, n, msd, p, w.hist, nblocks);
			MCOM_LAUNCH_CHECK(ctx);
			if ((rc = scan_u32(ctx, w.hist, w.hist, (size_t)256 * nblocks, w.scratch))) return rc;
			MCOM_LAUNCH(k_radix_scatter, dim3(nblocks), dim3(RS_THREADS), 0, ctx->stream, src, dst, n, msd, p, w.hist, nblocks);
			MCOM_LAUNCH_CHECK(ctx);
			src = dst; dst = dst == w.tmp ? d_sorted : w.tmp;
		}
		uint32_t novf = 0, scap_used = 0;
		{
			McomProfScope ps_(ctx, PROF_RADIX_PASS);
			MCOM_LAUNCH(k_seg_bounds, dim3((nseg + 1 + 255) / 256), dim3(256), 0, ctx->stream, w.tmp, n, msd, nseg, seg_start);
			ovf_count = (uint32_t*)mcom_zeroed(ctx, ovf_count, 4);
			if (!ovf_count) return mcom_fail(ctx, MCOM_E_HIP, "clear");
			// segments of ~1500 records on average fit the small form; a read set whose average is higher keeps the large one
			const uint32_t scap = ctx->seg_cap ? ctx->seg_cap : ((n >> B) <= 1600 ? (uint32_t)SS_CAP_SMALL : (uint32_t)SS_CAP);
			scap_used = scap;
			if (scap <= SS_CAP_SMALL) MCOM_LAUNCH(k_segment_sort<SS_CAP_SMALL>, dim3(nseg), dim3(SS_THREADS), 0, ctx->stream, w.tmp, d_sorted, seg_start, full, sig_bits, scap, ovf_count, ovf_list);
			else MCOM_LAUNCH(k_segment_sort<SS_CAP>, dim3(nseg), dim3(SS_THREADS), 0, ctx->stream, w.tmp, d_sorted, seg_start, full, sig_bits, scap, ovf_count, ovf_list);
			MCOM_LAUNCH_CHECK(ctx);
		}
		MCOM_HIP(ctx, mcom_d2h_async(ctx, &novf, ovf_count, 4));
		MCOM_HIP(ctx, mcom_stream_sync(ctx));
		if (novf && !ctx->seg_cap && scap_used == (uint32_t)SS_CAP_SMALL && 2 * (size_t)novf <= (size_t)nseg) {
			// the few segments between the small form's capacity and the large one's (a couple of dozen of 65 536 at an average of 1 500):
			// the large form over their list; what it cannot hold either is listed behind them and goes on to the nine-pass sort
			uint32_t *ovf2 = (uint32_t*)mcom_zeroed(ctx, ovf_count, 4);
			if (!ovf2) return mcom_fail(ctx, MCOM_E_HIP, "clear");
			{ McomProfScope ps_(ctx, PROF_RADIX_PASS);
			MCOM_LAUNCH(k_segment_sort<SS_CAP>, dim3(novf), dim3(SS_THREADS), 0, ctx->stream, w.tmp, d_sorted, seg_start, full, sig_bits, (uint32_t)SS_CAP, ovf2, ovf_list + novf, (const uint2*)ovf_list); }
			MCOM_LAUNCH_CHECK(ctx);
			uint32_t novf2 = 0;
			MCOM_HIP(ctx, mcom_d2h_async(ctx, &novf2, ovf2, 4));
			MCOM_HIP(ctx, mcom_stream_sync(ctx));
			ovf_list += novf; novf = novf2;
		}
		ctx->sort_overflow_segments += novf;
		if (novf) {
			// segments beyond the LDS arrays (one minimizer shared by thousands of reads): gathered, sorted by the whole key with the
			// nine-pass sort -- whose order refines the MSD order, so the segments stay apart and in place --, put back
			std::vector<uint2> list(novf);
			MCOM_HIP(ctx, hipMemcpy(list.data(), ovf_list, (size_t)novf * sizeof(uint2), hipMemcpyDeviceToHost));
			std::sort(list.begin(), list.end(), [](const uint2 &a, const uint2 &c) { return a.x < c.x; });
			std::vector<uint32_t> dst_off(novf);
			size_t m = 0;
			for (uint32_t q = 0; q < novf; ++q) { dst_off[q] = (uint32_t)m; m += list[q].y - list[q].x; }
			MCOM_HIP(ctx, hipMemcpyAsync(ovf_list, list.data(), (size_t)novf * sizeof(uint2), hipMemcpyHostToDevice, ctx->stream));
			MCOM_HIP(ctx, hipMemcpyAsync(ovf_dst, dst_off.data(), (size_t)novf * 4, hipMemcpyHostToDevice, ctx->stream));
			void *ws2 = nullptr;
			if (mcom_dmalloc(&ws2, sort_ws_layout(m, nullptr, nullptr)) != hipSuccess) return mcom_fail(ctx, MCOM_E_NOMEM, "workspace for %zu records of oversized segments", m);
			SortWs w2; sort_ws_layout(m, &w2, (char*)ws2);
			MCOM_LAUNCH(k_seg_move, dim3(novf), dim3(256), 0, ctx->stream, d_sorted, w.tmp, ovf_list, ovf_dst, 0);
			mcom_mm128 *res = nullptr;
			rc = radix_sort_records(ctx, w.tmp, w2.tmp, m, full, (2 * kmer + 9 + 7) / 8, w2.hist, w2.scratch, &res);
			if (!rc) MCOM_LAUNCH(k_seg_move, dim3(novf), dim3(256), 0, ctx->stream, d_sorted, res, ovf_list, ovf_dst, 1);
			hipError_t e2 = mcom_stream_sync(ctx);
			mcom_dfree(ws2);
			if (rc) return rc;
			if (e2 != hipSuccess) return mcom_fail(ctx, MCOM_E_HIP, "oversized segments: %s", hipGetErrorString(e2));
		}
	}
	const unsigned blocks = (unsigned)((n + 255) / 256);
	MCOM_LAUNCH(k_group_flags, dim3(blocks), dim3(256), 0, ctx->stream, d_sorted, n, f0, f1, f2);
	MCOM_LAUNCH_CHECK(ctx);
	if ((rc = scan_u32(ctx, f0, f0, n, scr))) return rc;
	if ((rc = scan_u32(ctx, f1, f1, n, scr))) return rc;
	if ((rc = scan_u32(ctx, f2, f2, n, scr))) return rc;
	MCOM_LAUNCH(k_count_valid, dim3(1), dim3(64), 0, ctx->stream, d_sorted, n, d_counts);
	MCOM_LAUNCH(k_group_emit, dim3(blocks), dim3(256), 0, ctx->stream, d_sorted, n, f0, f1, f2, d_singles, d_single_ord, d_members, d_group_off, d_counts);
	MCOM_LAUNCH_CHECK(ctx);
	MCOM_HIP(ctx, mcom_d2h_async(ctx, h_counts, d_counts, 4 * sizeof(uint64_t)));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	return MCOM_OK;
}
