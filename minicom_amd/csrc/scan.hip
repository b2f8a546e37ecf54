// minicom_amd/csrc/scan.hip -- exclusive prefix sums of uint32 / uint64 arrays in ONE launch (round 4).
//
// Rounds 1-3 scanned in tiles with a recursion on the tile sums: three to five launches per scan, ~170 scans per step -- 600 of
// the step's 1 800 launches.  Now a scan is one kernel of at most 2 workgroups per CU, every workgroup owning a CONTIGUOUS
// chunk of the array:
//   1. it sums its chunk and publishes the sum as two self-validating words { epoch << 32 | half of the sum };
//   2. it adds up the sums of the chunks before it (wave 0; a word whose epoch is not this launch's is not there yet);
//   3. it scans its chunk, tile by tile, from that base.
// Step 2 waits for other workgroups.  No co-residency is assumed (round 5): a workgroup's chunk is not its blockIdx but a TICKET it draws
// when it starts (one atomic per workgroup), so the chunks in front of it belong to workgroups that started before it -- they are running
// or done, they publish without waiting for anybody, and the wait ends whatever else occupies the card and in whatever order the
// workgroups are dispatched.  (The workgroup that draws the last ticket puts the counter back to zero for the next launch: launches of
// one stream do not overlap, and every context has a counter of its own.)  The wait is BOUNDED all the same: after ~2^22 polls a
// workgroup gives up, raises the context's poison flag (a word of pinned host memory that mcom_stream_sync checks) and finishes with a
// wrong base; the call that synchronises next fails with MCOM_E_HIP instead of the GPU hanging -- which now takes a workgroup that died,
// not one that was kept waiting.  The epoch words make a clear of the scratch between launches unnecessary.
#include "mcom_dev.hpp"

#define SO_THREADS 256
#define SO_MAX_WG 2048

template <class T> struct SoCfg;
template <> struct SoCfg<uint32_t> { static constexpr int PER = 8; };
template <> struct SoCfg<uint64_t> { static constexpr int PER = 4; };

template <class T>
__device__ __forceinline__ void so_load(const T *in, size_t base, size_t n, T (&v)[SoCfg<T>::PER])
{
	constexpr int PER = SoCfg<T>::PER;
	if (base + PER <= n && (((uintptr_t)(in + base)) & 15) == 0) {
		const uint4 *p = (const uint4*)(in + base);
		constexpr int NV = PER * (int)sizeof(T) / 16;
		uint4 q[NV];
#pragma unroll
		for (int u = 0; u < NV; ++u) q[u] = p[u];
		memcpy(v, q, sizeof(v));
	} else {
#pragma unroll
		for (int u = 0; u < PER; ++u) v[u] = base + u < n ? in[base + u] : (T)0;
	}
}
template <class T>
__device__ __forceinline__ void so_store(T *out, size_t base, size_t n, const T (&v)[SoCfg<T>::PER])
{
	constexpr int PER = SoCfg<T>::PER;
	if (base + PER <= n && (((uintptr_t)(out + base)) & 15) == 0) {
		constexpr int NV = PER * (int)sizeof(T) / 16;
		uint4 q[NV];
		memcpy(q, v, sizeof(v));
		uint4 *p = (uint4*)(out + base);
#pragma unroll
		for (int u = 0; u < NV; ++u) p[u] = q[u];
	} else {
#pragma unroll
		for (int u = 0; u < PER; ++u) if (base + u < n) out[base + u] = v[u];
	}
}

template <class T>
__global__ __launch_bounds__(SO_THREADS) void k_scan_one(const T *in, T *out, size_t n, uint32_t tiles_per_wg,
                                                         unsigned long long *__restrict__ parts, uint32_t epoch, unsigned int *__restrict__ poison,
                                                         unsigned long long *__restrict__ total_slot)
{
	constexpr int PER = SoCfg<T>::PER;
	constexpr size_t TILE = (size_t)SO_THREADS * PER;
	__shared__ T wsum[SO_THREADS / 64];
	__shared__ T s_base;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const uint32_t G = gridDim.x;
	uint32_t g = 0;
	if (G > 1) {
		__shared__ uint32_t s_ticket;
		if (tid == 0) {
			unsigned int *ticket = (unsigned int*)(parts + 2 * SO_MAX_WG);
			const uint32_t t = atomicAdd(ticket, 1u);
			if (t == G - 1) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // nobody draws after the last: ready for the next launch
			s_ticket = t;
		}
		__syncthreads();
		g = s_ticket;
	}
	const size_t first = (size_t)g * tiles_per_wg * TILE;
	size_t last = first + (size_t)tiles_per_wg * TILE; if (last > n) last = n;
	T base = 0;
	if (G > 1) {
		// 1. the chunk's sum
		T acc = 0;
		for (size_t t0 = first; t0 < last; t0 += TILE) {
			T v[PER]; so_load<T>(in, t0 + (size_t)tid * PER, last, v);
#pragma unroll
			for (int u = 0; u < PER; ++u) acc += v[u];
		}
#pragma unroll
		for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o, 64);
		if (lane == 0) wsum[wv] = acc;
		__syncthreads();
		if (tid == 0) {
			T tot = 0;
			for (int q = 0; q < SO_THREADS / 64; ++q) tot += wsum[q];
			const unsigned long long e = (unsigned long long)epoch << 32;
			__hip_atomic_store(&parts[2 * g], e | (unsigned long long)((uint64_t)tot & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(&parts[2 * g + 1], e | (unsigned long long)((uint64_t)tot >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		// 2. the sums of the chunks before this one
		if (wv == 0) {
			T before = 0;
			bool gave_up = false;
			for (uint32_t j = (uint32_t)lane; j < g; j += 64) {
				unsigned long long a, b;
				uint32_t polls = 0;
				for (;;) {
					a = __hip_atomic_load(&parts[2 * j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					b = __hip_atomic_load(&parts[2 * j + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					if ((uint32_t)(a >> 32) == epoch && (uint32_t)(b >> 32) == epoch) break;
					if (++polls > (1u << 22)) { gave_up = true; a = b = 0; break; }
					__builtin_amdgcn_s_sleep(4);
				}
				before += (T)(((uint64_t)(b & 0xFFFFFFFFull) << 32) | (uint64_t)(a & 0xFFFFFFFFull));
			}
#pragma unroll
			for (int o = 32; o; o >>= 1) before += __shfl_xor(before, o, 64);
			if (__any(gave_up) && lane == 0) *poison = 1u;
			if (lane == 0) s_base = before;
		}
		__syncthreads();
		base = s_base;
		__syncthreads();
	}
	// 3. the scan of the chunk
	for (size_t t0 = first; t0 < last; t0 += TILE) {
		const size_t at = t0 + (size_t)tid * PER;
		T v[PER]; so_load<T>(in, at, last, v);
		T tot = 0;
#pragma unroll
		for (int u = 0; u < PER; ++u) tot += v[u];
		T inc = tot;
#pragma unroll
		for (int s = 1; s < 64; s <<= 1) { const T t = __shfl_up(inc, s, 64); if (lane >= s) inc += t; }
		if (lane == 63) wsum[wv] = inc;
		__syncthreads();
		T add = 0, all = 0;
#pragma unroll
		for (int q = 0; q < SO_THREADS / 64; ++q) { if (q < wv) add += wsum[q]; all += wsum[q]; }
		T run = base + inc + add - tot;
		T o[PER];
#pragma unroll
		for (int u = 0; u < PER; ++u) { o[u] = run; run += v[u]; }
		so_store<T>(out, at, last, o);
		if (n - 1 >= at && n - 1 < at + PER) *total_slot = (unsigned long long)o[n - 1 - at];   // the last element, for the host (mcom_d2h_async)
		base += all;
		__syncthreads();
	}
}

// the scratch of a context's scans: the published sums (device), the poison flag (pinned host memory the kernels can write)
int mcom_scan_prepare(mcom_ctx *ctx)
{
	if (ctx->scan_parts) return MCOM_OK;
	MCOM_HIP(ctx, hipHostMalloc((void**)&ctx->poison, 64 + 8 * mcom_ctx::SCAN_TOTALS, hipHostMallocMapped));
	*ctx->poison = 0;
	MCOM_HIP(ctx, hipHostGetDevicePointer((void**)&ctx->d_poison, (void*)ctx->poison, 0));
	ctx->scan_tot = (unsigned long long*)((char*)ctx->poison + 64); ctx->d_scan_tot = (unsigned long long*)((char*)ctx->d_poison + 64);
	MCOM_HIP(ctx, hipMalloc((void**)&ctx->scan_parts, (size_t)2 * SO_MAX_WG * 8 + 64));             // (+ the ticket counter)
	MCOM_HIP(ctx, hipMemsetAsync(ctx->scan_parts, 0, (size_t)2 * SO_MAX_WG * 8 + 64, ctx->stream));
	ctx->scan_epoch = 0;
	return MCOM_OK;
}

template <class T>
static int scan_one(mcom_ctx *ctx, const T *in, T *out, size_t n)
{
	if (n == 0) return MCOM_OK;
	int rc = mcom_scan_prepare(ctx);
	if (rc) return rc;
	constexpr size_t TILE = (size_t)SO_THREADS * SoCfg<T>::PER;
	const size_t tiles = (n + TILE - 1) / TILE;
	size_t maxg = (size_t)ctx->n_cu * 2; if (maxg > SO_MAX_WG) maxg = SO_MAX_WG; if (maxg < 1) maxg = 1;
	const size_t per = (tiles + maxg - 1) / maxg;
	const size_t G = (tiles + per - 1) / per;
	if (per >= (1ull << 32)) return mcom_fail(ctx, MCOM_E_ARG, "scan of %zu elements", n);
	if (++ctx->scan_epoch == 0) ctx->scan_epoch = 1;                               // (0 = what the cleared scratch holds)
	const uint32_t slot = ctx->scan_epoch % mcom_ctx::SCAN_TOTALS, gen_before = ctx->launch_gen;
	MCOM_LAUNCH(k_scan_one<T>, dim3((unsigned)G), dim3(SO_THREADS), 0, ctx->stream, in, out, n, (uint32_t)per, ctx->scan_parts, ctx->scan_epoch, ctx->d_poison, ctx->d_scan_tot + slot);
	MCOM_LAUNCH_CHECK(ctx);
	mcom_ring_flush_slot(ctx, slot);                                               // (the ring came round: a read-back still waiting on that word is served now)
	for (mcom_ctx::ScanTotal &t : ctx->scan_last) {
		if (t.slot == slot) t.last = nullptr;                                      // (that total is gone)
		if (t.last && (const char*)t.last >= (const char*)out && (const char*)t.last < (const char*)(out + n)) t.last = nullptr;   // this scan overwrites it (a reused scratch array)
		if (t.gen == gen_before) t.gen = ctx->launch_gen;                          // scans in a row: the earlier ones' totals still stand
	}
	ctx->scan_last[ctx->scan_last_at++ % 8] = mcom_ctx::ScanTotal{(const void*)(out + n - 1), (uint32_t)sizeof(T), slot, ctx->launch_gen};
	return MCOM_OK;
}

// A ring word about to be handed out again may still be what a queued read-back (mcom_d2h_async, PinWait::from) waits for -- 64 scans or
// ring slots between a read-back and its synchronisation: the stream is synchronised and the waiting values delivered before the word is reused.
void mcom_ring_flush_slot(mcom_ctx *ctx, uint32_t slot)
{
	for (const mcom_ctx::PinWait &w : ctx->pin_wait)
		if (w.from == (const void*)(ctx->scan_tot + slot)) { (void)mcom_stream_sync(ctx); return; }
}

// The ring of pinned words for OTHER kernels whose result is one value written by one thread at their end (a fold's totals, the
// claim kernel's round count): mcom_ring_slot before the launch says where the kernel stores the value beside its place in device
// memory, mcom_ring_register after the launch makes mcom_d2h_async find it there -- no copy launch for the read-back.
unsigned long long *mcom_ring_slot(mcom_ctx *ctx, uint32_t *slot)
{
	if (mcom_scan_prepare(ctx)) return nullptr;
	if (++ctx->scan_epoch == 0) ctx->scan_epoch = 1;
	*slot = ctx->scan_epoch % mcom_ctx::SCAN_TOTALS;
	mcom_ring_flush_slot(ctx, *slot);
	for (mcom_ctx::ScanTotal &t : ctx->scan_last) if (t.slot == *slot) t.last = nullptr;
	return ctx->d_scan_tot + *slot;
}
void mcom_ring_register(mcom_ctx *ctx, const void *d_result, uint32_t bytes, uint32_t slot)      // bytes <= 8; right after the launch
{
	ctx->scan_last[ctx->scan_last_at++ % 8] = mcom_ctx::ScanTotal{d_result, bytes, slot, ctx->launch_gen};
}

// exported to the other translation units of the library (the scratch arguments are what rounds 1-3 needed: unused, kept so that
// the callers' workspace layouts stay as they are)
int mcom_scan_u32(mcom_ctx *ctx, const uint32_t *in, uint32_t *out, size_t n, uint32_t *) { return scan_one<uint32_t>(ctx, in, out, n); }
size_t mcom_scan_scratch_elems(size_t) { return 1; }
int mcom_scan64(mcom_ctx *ctx, const uint64_t *in, uint64_t *out, size_t n, uint64_t *) { return scan_one<uint64_t>(ctx, in, out, n); }
size_t mcom_scan64_scratch_elems(size_t) { return 1; }
