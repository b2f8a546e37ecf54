// minicom_amd/csrc/dist.hip -- small device helpers of the multi-GPU path (the reference has no
// counterpart, it is a shared-memory program).  The exchange itself is host code over RCCL (host/mcom_comm.cpp); the
// partition by owner and the sort by read id are radix passes (sort.hip: mcom_partition_by_owner, mcom_sort_by_rid).
#include "mcom_dev.hpp"

// out[i] = min over q of parts[q * stride + i]: the Stage-2 claim of a singleton is the minimum claim key over all
// (contig, window, direction, dictionary) tuples it passes; every rank holds the minimum over
// its share of the contigs, this folds the R shares of one slice of the singletons.
__global__ void k_min_fold(const unsigned long long *__restrict__ parts, int R, size_t stride, size_t n, unsigned long long *__restrict__ out)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	unsigned long long v = parts[i];
	for (int q = 1; q < R; ++q) { const unsigned long long w = parts[(size_t)q * stride + i]; v = w < v ? w : v; }
	out[i] = v;
}
extern "C" int mcom_min_fold_u64(mcom_ctx *ctx, const uint64_t *d_parts, int n_parts, size_t stride, size_t n, uint64_t *d_out)
{
	if (!ctx || n_parts < 1) return MCOM_E_ARG;
	if (n == 0) return MCOM_OK;
	if (!d_parts || !d_out) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	MCOM_LAUNCH(k_min_fold, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const unsigned long long*)d_parts, n_parts, stride, n, (unsigned long long*)d_out);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

// largest value of a uint16 array (does any read of the shard hold an N? then the N masks have to travel)
__global__ void k_max_u16(const uint16_t *__restrict__ v, size_t n, uint32_t *__restrict__ out)
{
	uint32_t m = 0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = v[i] > m ? v[i] : m;
	for (int o = 32; o; o >>= 1) { const uint32_t t = __shfl_xor(m, o); m = t > m ? t : m; }
	if ((threadIdx.x & 63) == 0 && m > *out) atomicMax(out, m);       // filtered: at most a handful of atomics on the one address
}
extern "C" int mcom_max_u16(mcom_ctx *ctx, const uint16_t *d_v, size_t n, uint32_t *h_max)
{
	if (!ctx || !h_max) return MCOM_E_ARG;
	*h_max = 0;
	if (n == 0) return MCOM_OK;
	if (!d_v) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	int rc = mcom_ws_reserve(ctx, 256);
	if (rc) return rc;
	uint32_t *d = (uint32_t*)ctx->ws;
	MCOM_HIP(ctx, hipMemsetAsync(d, 0, 4, ctx->stream));
	const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
	MCOM_LAUNCH(k_max_u16, dim3(blocks), dim3(256), 0, ctx->stream, d_v, n, d);
	MCOM_LAUNCH_CHECK(ctx);
	MCOM_HIP(ctx, mcom_d2h_async(ctx, h_max, d, 4));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	return MCOM_OK;
}

// A rank sketches contigs [c0, c1) of the replicated set as if they were contigs 0 .. c1-c0-1: ids (the contig index in the high
// word of y) and record offsets are moved to their global values before the all-gather.
__global__ void k_records_rebase(mcom_mm128 *__restrict__ rec, size_t n, unsigned long long id_delta)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) rec[i].y += id_delta;
}
__global__ void k_add_u32(uint32_t *__restrict__ v, size_t n, uint32_t delta)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) v[i] += delta;
}
__global__ void k_add_u64(unsigned long long *__restrict__ v, size_t n, unsigned long long delta)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) v[i] += delta;
}
extern "C" int mcom_offsets_rebase(mcom_ctx *ctx, uint64_t *d_off, size_t n, uint64_t delta)
{
	if (!ctx) return MCOM_E_ARG;
	if (!n || !delta) return MCOM_OK;
	if (!d_off) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	MCOM_LAUNCH(k_add_u64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (unsigned long long*)d_off, n, (unsigned long long)delta);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}
extern "C" int mcom_records_rebase(mcom_ctx *ctx, mcom_mm128 *d_rec, size_t n_rec, uint32_t first_contig, uint32_t *d_roff, size_t n_off, uint32_t first_record)
{
	if (!ctx) return MCOM_E_ARG;
	if (n_rec && first_contig) {
		if (!d_rec) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
		MCOM_LAUNCH(k_records_rebase, dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, ctx->stream, d_rec, n_rec, (unsigned long long)first_contig << 32);
		MCOM_LAUNCH_CHECK(ctx);
	}
	if (n_off && first_record) {
		if (!d_roff) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
		MCOM_LAUNCH(k_add_u32, dim3((unsigned)((n_off + 255) / 256)), dim3(256), 0, ctx->stream, d_roff, n_off, first_record);
		MCOM_LAUNCH_CHECK(ctx);
	}
	return MCOM_OK;
}

// ---- result digest: wrapping sum and xor of the little-endian 64-bit words of a device array (a tail of fewer than 8
// bytes is zero-extended).  bench.py prints it for every step and compares it with the digest of a run whose result was
// checked; two runs with the same digest of strings, members and offsets hold the same contig set.
__global__ void k_digest(const uint8_t *__restrict__ p, size_t bytes, unsigned long long *__restrict__ out)
{
	const size_t nw = bytes >> 3;
	unsigned long long s = 0, x = 0;
	const unsigned long long *w = (const unsigned long long*)p;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += (size_t)gridDim.x * blockDim.x) {
		const unsigned long long v = w[i];
		s += v * (unsigned long long)(2 * (i & 0xFFFFF) + 1); x ^= v;           // the sum is position-weighted: it notices swapped words
	}
	if (blockIdx.x == 0 && threadIdx.x == 0 && (bytes & 7)) {
		unsigned long long v = 0;
		for (size_t b = 0; b < (bytes & 7); ++b) v |= (unsigned long long)p[(nw << 3) + b] << (8 * b);
		s += v * (unsigned long long)(2 * (nw & 0xFFFFF) + 1); x ^= v;
	}
	for (int o = 32; o; o >>= 1) { s += __shfl_xor(s, o); x ^= __shfl_xor(x, o); }
	if ((threadIdx.x & 63) == 0) { unsigned long long *slot = out + 2 * (blockIdx.x & 255); atomicAdd(&slot[0], s); atomicXor(&slot[1], x); }
}
__global__ void k_digest_fold(unsigned long long *__restrict__ out)
{
	unsigned long long s = 0, x = 0;
	for (int i = 0; i < 256; ++i) { s += out[2 * i]; x ^= out[2 * i + 1]; }
	out[512] = s; out[513] = x;
}
extern "C" int mcom_digest(mcom_ctx *ctx, const void *d_data, size_t bytes, uint64_t *h_sum_xor)
{
	if (!ctx || !h_sum_xor) return MCOM_E_ARG;
	h_sum_xor[0] = h_sum_xor[1] = 0;
	if (bytes == 0) return MCOM_OK;
	if (!d_data || ((uintptr_t)d_data & 7)) return mcom_fail(ctx, MCOM_E_ARG, "digest: null or unaligned device pointer");
	int rc = mcom_ws_reserve(ctx, 514 * 8);
	if (rc) return rc;
	unsigned long long *d = (unsigned long long*)ctx->ws;
	MCOM_HIP(ctx, hipMemsetAsync(d, 0, 514 * 8, ctx->stream));
	const size_t nw = bytes >> 3;
	const unsigned blocks = (unsigned)(nw / 256 + 1 < 4096 ? nw / 256 + 1 : 4096);
	MCOM_LAUNCH(k_digest, dim3(blocks), dim3(256), 0, ctx->stream, (const uint8_t*)d_data, bytes, d);
	MCOM_LAUNCH(k_digest_fold, dim3(1), dim3(1), 0, ctx->stream, d);
	MCOM_LAUNCH_CHECK(ctx);
	MCOM_HIP(ctx, mcom_d2h_async(ctx, h_sum_xor, d + 512, 16));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	return MCOM_OK;
}
