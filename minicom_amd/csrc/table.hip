// minicom_amd/csrc/table.hip -- exact hash table in HBM over the runs of a sorted record array.
// Shared by the Stage-2 dictionaries (realign.hip) and the contig-minimizer index (contigs.hip).
#include "mcom_dev.hpp"

__global__ void k_table_heads(const mcom_mm128 *__restrict__ s, size_t n, uint32_t *__restrict__ head)      // n + 1 entries, the last one 0
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i <= n) head[i] = (i < n && (i == 0 || s[i].x != s[i - 1].x)) ? 1u : 0u;
}

// hpre: exclusive prefix of the head flags.  The run heads are listed first (hidx[h] = position of the h-th head,
// hidx[H] = n), so that the insert kernel runs with every lane busy: it is bound by the latency of its random accesses,
// and with one thread per RECORD only the heads -- a third of the lanes -- had a request in flight.
__global__ void k_table_head_list(const mcom_mm128 *__restrict__ s, size_t n, const uint32_t *__restrict__ hpre, uint32_t *__restrict__ hidx)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i > n) return;
	if (i == n) { hidx[hpre[n]] = (uint32_t)n; return; }                        // hpre[n] = number of heads
	if (i == 0 || s[i].x != s[i - 1].x) hidx[hpre[i]] = (uint32_t)i;
}
__global__ void k_table_insert(const mcom_mm128 *__restrict__ s, const uint32_t *__restrict__ hidx, uint32_t n_heads, uint64_t *__restrict__ slots,
                               uint32_t log2cap, uint32_t *__restrict__ meta /* [1]=maxrun */)
{
	const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
	if (h >= n_heads) return;
	const uint32_t i = hidx[h], cnt = hidx[h + 1] - i;                         // a run ends where the next one starts
	const uint64_t key = s[i].x;
	const uint32_t capm = (1u << log2cap) - 1u;
	uint32_t sl = mcom_slot_of(key, log2cap);
	// a plain read first: it brings the line into L2 (an atomic that misses costs several times one that hits: measured on
	// the contig index) and an occupied slot is passed without an atomic at all
	for (;;) {
		if (*(volatile uint64_t*)&slots[2 * (size_t)sl] == ~0ull &&
		    atomicCAS((unsigned long long*)&slots[2 * (size_t)sl], ~0ull, (unsigned long long)key) == ~0ull) break;
		sl = (sl + 1) & capm;
	}
	slots[2 * (size_t)sl + 1] = (uint64_t)i | ((uint64_t)cnt << 32);
	if (cnt > meta[1]) atomicMax(&meta[1], cnt);             // filtered: one address for every key would serialise on its L2 channel
}

// ---- bucketed build: one workgroup per bucket, the bucket's region of the table assembled in LDS and written in one piece ----
// Round 4: the keys and the longest run of a bucket go to a per-bucket pair with a plain store and a second, tiny launch folds the
// pairs.  Before, every wave added to ONE pair of counters: 65 536 atomics on one cache line take ~4 ns each on its L2 channel and
// every workgroup waited for its turn -- 0.8 ms per build whatever the size of the index (0.9 -> 0.x ms on the largest, 0.8 -> 0.x
// on the smallest of a step).  The bucket's keys are staged in LDS (all of the workgroup's loads in flight at once).
template <bool KEYS>
__global__ __launch_bounds__(256) void k_table_bucket(const mcom_mm128 *__restrict__ s, const uint32_t *__restrict__ bstart, int bbits, uint32_t R,
                                                      uint64_t *__restrict__ slots, uint2 *__restrict__ per_bucket /* { keys, longest run } */,
                                                      uint32_t bucket0, uint32_t start_base)
{
	extern __shared__ unsigned long long reg[];                               // [R][2]: key, start | count << 32; then the keys
	__shared__ uint32_t wsum[4], wmax[4];
	const uint32_t v = bucket0 + blockIdx.x;
	const uint32_t b0 = bstart[v], b1 = bstart[v + 1];
	unsigned long long *keys = reg + 2 * (size_t)R;
	if (KEYS) for (uint32_t i = b0 + threadIdx.x; i < b1; i += 256) keys[i - b0] = s[i].x;
	for (uint32_t q = threadIdx.x; q < 2 * R; q += 256) reg[q] = ~0ull;
	__syncthreads();
	uint32_t heads = 0, longest = 0;
	for (uint32_t i = b0 + threadIdx.x; i < b1; i += 256) {
		const uint64_t key = KEYS ? keys[i - b0] : s[i].x;
		if (i > b0 && (KEYS ? keys[i - 1 - b0] : s[i - 1].x) == key) continue;  // not the head of its run
		uint32_t e = i + 1;
		while (e < b1 && (KEYS ? keys[e - b0] : s[e].x) == key) ++e;
		uint32_t sl = mcom_region_slot(key >> bbits, R);
		while (atomicCAS(&reg[2 * sl], ~0ull, (unsigned long long)key) != ~0ull) sl = sl + 1 == R ? 0 : sl + 1;
		reg[2 * sl + 1] = (unsigned long long)(start_base + i) | ((unsigned long long)(e - i) << 32);
		++heads; longest = e - i > longest ? e - i : longest;
	}
	for (int o = 32; o; o >>= 1) { heads += __shfl_xor(heads, o); const uint32_t t = __shfl_xor(longest, o); longest = t > longest ? t : longest; }
	if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6] = heads; wmax[threadIdx.x >> 6] = longest; }
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t lg = wmax[0]; lg = wmax[1] > lg ? wmax[1] : lg; lg = wmax[2] > lg ? wmax[2] : lg; lg = wmax[3] > lg ? wmax[3] : lg;
		per_bucket[blockIdx.x] = make_uint2(wsum[0] + wsum[1] + wsum[2] + wsum[3], lg);
	}
	ulonglong2 *dst = (ulonglong2*)(slots + 2 * ((size_t)v * R));
	for (uint32_t q = threadIdx.x; q < R; q += 256) dst[q] = make_ulonglong2(reg[2 * q], reg[2 * q + 1]);
}
__global__ __launch_bounds__(1024) void k_table_fold(const uint2 *__restrict__ per_bucket, uint32_t nb, uint32_t *__restrict__ meta /* [0] keys, [1] longest run */,
                                                     uint32_t *__restrict__ host_copy)
{
	__shared__ uint32_t wsum[16], wmax[16];
	uint32_t a = 0, m = 0;
	for (uint32_t i = threadIdx.x; i < nb; i += 1024) { const uint2 c = per_bucket[i]; a += c.x; m = c.y > m ? c.y : m; }
	for (int o = 32; o; o >>= 1) { a += __shfl_xor(a, o); const uint32_t t = __shfl_xor(m, o); m = t > m ? t : m; }
	if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6] = a; wmax[threadIdx.x >> 6] = m; }
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int w = 1; w < 16; ++w) { a += wsum[w]; m = wmax[w] > m ? wmax[w] : m; }
		meta[0] = a; meta[1] = m;
		if (host_copy) { host_copy[0] = a; host_copy[1] = m; }                      // (pinned memory: the read-back needs no copy, scan.hip)
	}
}

// a region holds the fullest bucket at most 0.73 full if every record of it were a key of its own (about half of them are):
// as many slots as that asks for, a multiple of four -- not the next power of two, which wrote and cleared up to twice the table.
// MCOM_E_OVERFLOW: a bucket too large for a region in LDS (8192 slots = 128 KB) -- the global table (mcom_table_build) serves then.
int mcom_table_alloc_bucketed(mcom_ctx *ctx, int bbits, uint32_t max_bucket, McomTable *t)
{
	const uint32_t R = (max_bucket + (max_bucket >> 2) + (max_bucket >> 3) + 16 + 3) & ~3u;
	if (bbits < 1 || R > 8192 || ((size_t)16 * R << bbits) > ((size_t)64 << 30)) return MCOM_E_OVERFLOW;
	t->slots = nullptr; t->numkeys = 0; t->maxrun = 0; t->region = R; t->bbits = (uint32_t)bbits; t->log2cap = 0;
	const size_t bytes = (size_t)16 * R << bbits;
	hipError_t e = mcom_dmalloc(&t->slots, bytes);
	if (e != hipSuccess) { t->slots = nullptr; return mcom_fail(ctx, MCOM_E_NOMEM, "hash table of %zu bytes: %s", bytes, hipGetErrorString(e)); }
	return MCOM_OK;
}
// the regions of buckets [bucket0, bucket1) from a sorted record array whose record 0 is record start_base of the index
int mcom_table_fill_buckets(mcom_ctx *ctx, const mcom_mm128 *sorted, const uint32_t *bstart, uint32_t bucket0, uint32_t bucket1, uint32_t start_base, McomTable *t)
{
	if (bucket0 >= bucket1) return MCOM_OK;
	const uint32_t nb = bucket1 - bucket0;
	int rc = mcom_ws_reserve(ctx, 256 + (size_t)8 * nb);
	if (rc) return rc;
	uint32_t *meta = (uint32_t*)ctx->ws;
	uint2 *per_bucket = (uint2*)((char*)ctx->ws + 256);
	// a region was sized for the fullest bucket (mcom_table_alloc_bucketed): R >= 1.375 x that bucket, so R keys always cover it
	const size_t lds0 = (size_t)16 * t->region, lds1 = lds0 + (size_t)8 * t->region;
	if (lds1 <= 64 * 1024) {
		MCOM_HIP(ctx, hipFuncSetAttribute((const void*)k_table_bucket<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
		MCOM_LAUNCH(k_table_bucket<true>, dim3(nb), dim3(256), lds1, ctx->stream, sorted, bstart, (int)t->bbits, t->region, t->slots, per_bucket, bucket0, start_base);
	} else {
		MCOM_HIP(ctx, hipFuncSetAttribute((const void*)k_table_bucket<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds0));
		MCOM_LAUNCH(k_table_bucket<false>, dim3(nb), dim3(256), lds0, ctx->stream, sorted, bstart, (int)t->bbits, t->region, t->slots, per_bucket, bucket0, start_base);
	}
	uint32_t ring = 0;
	uint32_t *host_copy = (uint32_t*)mcom_ring_slot(ctx, &ring);
	MCOM_LAUNCH(k_table_fold, dim3(1), dim3(1024), 0, ctx->stream, per_bucket, nb, meta, host_copy);
	if (host_copy) mcom_ring_register(ctx, meta, 8, ring);
	MCOM_LAUNCH_CHECK(ctx);
	uint32_t hm[2] = {0, 0};
	MCOM_HIP(ctx, mcom_d2h_async(ctx, hm, meta, 8));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	t->numkeys += hm[0]; t->maxrun = hm[1] > t->maxrun ? hm[1] : t->maxrun;
	return MCOM_OK;
}

void mcom_table_free(McomTable *t)
{
	if (t && t->slots) { mcom_dfree(t->slots); t->slots = nullptr; }
}

int mcom_table_build(mcom_ctx *ctx, const mcom_mm128 *sorted, size_t n, uint32_t *head, uint32_t *scr, uint32_t *meta, McomTable *t)
{
	t->slots = nullptr; t->numkeys = 0; t->maxrun = 0; t->region = 0; t->bbits = 0;
	uint32_t lg = 4;
	while ((1ull << lg) < 2 * n + 16) ++lg;
	t->log2cap = lg;
	hipError_t e = mcom_dmalloc(&t->slots, (size_t)16 << lg);
	if (e != hipSuccess) { t->slots = nullptr; return mcom_fail(ctx, MCOM_E_NOMEM, "hash table of %zu bytes: %s", (size_t)16 << lg, hipGetErrorString(e)); }
	MCOM_HIP(ctx, hipMemsetAsync(t->slots, 0xFF, (size_t)16 << lg, ctx->stream));
	if (n == 0) { MCOM_HIP(ctx, mcom_stream_sync(ctx)); return MCOM_OK; }
	// own buffers (the caller's head / scr arrays are sized for n entries; the head list needs n + 1 and n + 2)
	(void)head; (void)scr;
	const size_t scr_elems = mcom_scan_scratch_elems(n + 1) + 256;
	uint32_t *buf = nullptr;
	if (mcom_dmalloc(&buf, ((n + 1) + (n + 2) + scr_elems) * 4) != hipSuccess) return mcom_fail(ctx, MCOM_E_NOMEM, "run heads");
	uint32_t *hpre = buf, *hidx = buf + (n + 1), *scr2 = hidx + (n + 2);
	const unsigned blocks = (unsigned)((n + 1 + 255) / 256);
	MCOM_LAUNCH(k_table_heads, dim3(blocks), dim3(256), 0, ctx->stream, sorted, n, hpre);
	int rc = mcom_scan_u32(ctx, hpre, hpre, n + 1, scr2);
	uint32_t nh = 0;
	hipError_t e3 = hipSuccess;
	if (!rc) {
		MCOM_LAUNCH(k_table_head_list, dim3(blocks), dim3(256), 0, ctx->stream, sorted, n, hpre, hidx);
		e3 = hipMemsetAsync(meta, 0, 8, ctx->stream);
		if (e3 == hipSuccess) e3 = mcom_d2h_async(ctx, &nh, hpre + n, 4);
		if (e3 == hipSuccess) e3 = mcom_stream_sync(ctx);
	}
	uint32_t hm[2] = {0, 0};
	if (!rc && e3 == hipSuccess && nh) {
		MCOM_LAUNCH(k_table_insert, dim3((nh + 255) / 256), dim3(256), 0, ctx->stream, sorted, hidx, nh, t->slots, lg, meta);
		e3 = hipGetLastError();
		if (e3 == hipSuccess) e3 = mcom_d2h_async(ctx, hm, meta, 8);
		if (e3 == hipSuccess) e3 = mcom_stream_sync(ctx);
	}
	mcom_dfree(buf);
	if (rc) return rc;
	if (e3 != hipSuccess) return mcom_fail(ctx, MCOM_E_HIP, "%s", hipGetErrorString(e3));
	hm[0] = nh;
	t->numkeys = hm[0]; t->maxrun = hm[1];
	return MCOM_OK;
}
