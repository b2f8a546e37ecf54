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

void mcom_table_free(McomTable *t)
{
	if (t && t->slots) { mcom_dfree(t->slots); t->slots = nullptr; }
}

int mcom_table_build(mcom_ctx *ctx, const mcom_mm128 *sorted, size_t n, uint32_t *head, uint32_t *scr, uint32_t *meta, McomTable *t)
{
	t->slots = nullptr; t->numkeys = 0; t->maxrun = 0;
	uint32_t lg = 4;
	while ((1ull << lg) < 2 * n + 16) ++lg;
	t->log2cap = lg;
	hipError_t e = mcom_dmalloc(&t->slots, (size_t)16 << lg);
	if (e != hipSuccess) { t->slots = nullptr; return mcom_fail(ctx, MCOM_E_NOMEM, "hash table of %zu bytes: %s", (size_t)16 << lg, hipGetErrorString(e)); }
	MCOM_HIP(ctx, hipMemsetAsync(t->slots, 0xFF, (size_t)16 << lg, ctx->stream));
	if (n == 0) { MCOM_HIP(ctx, hipStreamSynchronize(ctx->stream)); return MCOM_OK; }
	// own buffers (the caller's head / scr arrays are sized for n entries; the head list needs n + 1 and n + 2)
	(void)head; (void)scr;
	const size_t scr_elems = mcom_scan_scratch_elems(n + 1) + 256;
	uint32_t *buf = nullptr;
	if (mcom_dmalloc(&buf, ((n + 1) + (n + 2) + scr_elems) * 4) != hipSuccess) return mcom_fail(ctx, MCOM_E_NOMEM, "run heads");
	uint32_t *hpre = buf, *hidx = buf + (n + 1), *scr2 = hidx + (n + 2);
	const unsigned blocks = (unsigned)((n + 1 + 255) / 256);
	hipLaunchKernelGGL(k_table_heads, dim3(blocks), dim3(256), 0, ctx->stream, sorted, n, hpre);
	int rc = mcom_scan_u32(ctx, hpre, hpre, n + 1, scr2);
	uint32_t nh = 0;
	hipError_t e3 = hipSuccess;
	if (!rc) {
		hipLaunchKernelGGL(k_table_head_list, dim3(blocks), dim3(256), 0, ctx->stream, sorted, n, hpre, hidx);
		e3 = hipMemsetAsync(meta, 0, 8, ctx->stream);
		if (e3 == hipSuccess) e3 = hipMemcpyAsync(&nh, hpre + n, 4, hipMemcpyDeviceToHost, ctx->stream);
		if (e3 == hipSuccess) e3 = hipStreamSynchronize(ctx->stream);
	}
	uint32_t hm[2] = {0, 0};
	if (!rc && e3 == hipSuccess && nh) {
		hipLaunchKernelGGL(k_table_insert, dim3((nh + 255) / 256), dim3(256), 0, ctx->stream, sorted, hidx, nh, t->slots, lg, meta);
		e3 = hipGetLastError();
		if (e3 == hipSuccess) e3 = hipMemcpyAsync(hm, meta, 8, hipMemcpyDeviceToHost, ctx->stream);
		if (e3 == hipSuccess) e3 = hipStreamSynchronize(ctx->stream);
	}
	mcom_dfree(buf);
	if (rc) return rc;
	if (e3 != hipSuccess) return mcom_fail(ctx, MCOM_E_HIP, "%s", hipGetErrorString(e3));
	hm[0] = nh;
	t->numkeys = hm[0]; t->maxrun = hm[1];
	return MCOM_OK;
}
