// minicom_amd/csrc/table.hip -- exact hash table in HBM over the runs of a sorted record array.
// Shared by the Stage-2 dictionaries (realign.hip) and the contig-minimizer index (contigs.hip).
#include "mcom_dev.hpp"

__global__ void k_table_heads(const mcom_mm128 *__restrict__ s, size_t n, uint32_t *__restrict__ head)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) head[i] = (i == 0 || s[i].x != s[i - 1].x) ? 1u : 0u;
}

// hpre: exclusive prefix of the head flags.  Only run heads insert.
__global__ void k_table_insert(const mcom_mm128 *__restrict__ s, size_t n, const uint32_t *__restrict__ hpre, uint64_t *__restrict__ slots,
                               uint32_t log2cap, uint32_t *__restrict__ meta /* [0]=numkeys, [1]=maxrun */)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint64_t key = s[i].x;
	if (i > 0 && s[i - 1].x == key) return;
	size_t e = i + 1;                                       // run length by forward scan (runs are short; long ones are rare)
	while (e < n && s[e].x == key) ++e;
	const uint32_t cnt = (uint32_t)(e - i);
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
	if (e == n) meta[0] = hpre[i] + 1;
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
	const unsigned blocks = (unsigned)((n + 255) / 256);
	hipLaunchKernelGGL(k_table_heads, dim3(blocks), dim3(256), 0, ctx->stream, sorted, n, head);
	MCOM_LAUNCH_CHECK(ctx);
	int rc = mcom_scan_u32(ctx, head, head, n, scr);
	if (rc) return rc;
	MCOM_HIP(ctx, hipMemsetAsync(meta, 0, 8, ctx->stream));
	hipLaunchKernelGGL(k_table_insert, dim3(blocks), dim3(256), 0, ctx->stream, sorted, n, head, t->slots, lg, meta);
	MCOM_LAUNCH_CHECK(ctx);
	uint32_t hm[2] = {0, 0};
	MCOM_HIP(ctx, hipMemcpyAsync(hm, meta, 8, hipMemcpyDeviceToHost, ctx->stream));
	MCOM_HIP(ctx, hipStreamSynchronize(ctx->stream));
	t->numkeys = hm[0]; t->maxrun = hm[1];
	return MCOM_OK;
}
