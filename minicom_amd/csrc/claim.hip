// minicom_amd/csrc/claim.hip -- the first-come claiming of find_next (reference kthread_cb.c:267-343) on the device.
//
// The reference visits the contigs in index order; an unclaimed contig takes the first of its passing candidates
// (in lookup order) that is still unclaimed, and both are flagged (:339-343).  With the passing candidate pairs in
// that visiting order (mcom_find_next_candidates emits them so) this is the greedy maximal matching over the edge
// list in list order: an edge is taken iff neither end is taken by an EARLIER edge.  That matching is unique and can
// be computed in rounds: an edge that is the earliest live edge at both of its ends belongs to it, whatever happens
// elsewhere; taking it kills every other edge at its two ends.  Contig indices are unrelated to positions on the
// genome, so the chains of dependent edges are short and a few dozen rounds finish tens of millions of edges.
#include "mcom_dev.hpp"

// live edges: dead when an end is matched, else they bid (with their list position) for both ends
__global__ void k_claim_bid(const mcom_mm128 *__restrict__ pairs, size_t n, const uint8_t *__restrict__ matched, uint8_t *__restrict__ dead,
                            unsigned int *__restrict__ best, unsigned int *__restrict__ any_live)
{
	const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= n || dead[e]) return;
	const mcom_mm128 pr = pairs[e];
	const uint32_t ci = (uint32_t)(pr.x >> 32), cj = (uint32_t)(pr.y >> 32);
	if (matched[ci] || matched[cj]) { dead[e] = 1; return; }
	atomicMin(&best[ci], (unsigned int)e);
	atomicMin(&best[cj], (unsigned int)e);
	if (*(volatile unsigned int*)any_live == 0) *any_live = 1;            // filtered: millions of stores to one address would queue up on its L2 channel
}
__global__ void k_claim_take(const mcom_mm128 *__restrict__ pairs, size_t n, uint8_t *__restrict__ matched, uint8_t *__restrict__ dead,
                             const unsigned int *__restrict__ best, uint32_t *__restrict__ sel)
{
	const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= n || dead[e]) return;
	const mcom_mm128 pr = pairs[e];
	const uint32_t ci = (uint32_t)(pr.x >> 32), cj = (uint32_t)(pr.y >> 32);
	if (best[ci] == (unsigned int)e && best[cj] == (unsigned int)e) { matched[ci] = 1; matched[cj] = 1; sel[e] = 1; dead[e] = 1; }
}
__global__ void k_claim_jobs(const mcom_mm128 *__restrict__ pairs, size_t n, const uint32_t *__restrict__ sel, const uint32_t *__restrict__ spre,
                             uint32_t *__restrict__ jobs)
{
	const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= n || !sel[e]) return;
	const mcom_mm128 pr = pairs[e];
	uint32_t *j = jobs + 4 * (size_t)spre[e];
	j[0] = (uint32_t)(pr.x >> 32); j[1] = (uint32_t)(pr.y >> 32); j[2] = (uint32_t)pr.x >> 1; j[3] = (uint32_t)pr.y >> 1;
}

extern "C" int mcom_claim_pairs(mcom_ctx *ctx, const mcom_mm128 *d_pairs, size_t n_pairs, size_t n_contigs, int max_rounds, uint32_t *d_jobs,
                                uint8_t *d_flag, uint64_t *h_nj, int *h_rounds)
{
	if (!ctx || !h_nj) return MCOM_E_ARG;
	*h_nj = 0; if (h_rounds) *h_rounds = 0;
	if (n_contigs && !d_flag) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	if (n_contigs) MCOM_HIP(ctx, hipMemsetAsync(d_flag, 0, n_contigs, ctx->stream));
	if (n_pairs == 0) return MCOM_OK;
	if (!d_pairs || !d_jobs || n_pairs >= (1ull << 32) - 1 || n_contigs >= (1ull << 32) - 1) return mcom_fail(ctx, MCOM_E_ARG, "bad claim arguments");
	auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
	const size_t best_b = al(n_contigs * 4), dead_b = al(n_pairs), sel_b = al((n_pairs + 1) * 4), scr_b = al(mcom_scan_scratch_elems(n_pairs + 1) * 4 + 1024);
	int rc = mcom_ws_reserve(ctx, best_b + dead_b + 2 * sel_b + scr_b + 256);
	if (rc) return rc;
	char *base = (char*)ctx->ws;
	unsigned int *best = (unsigned int*)base;
	uint8_t *dead = (uint8_t*)(base + best_b);
	uint32_t *sel = (uint32_t*)(base + best_b + dead_b);
	uint32_t *scr = (uint32_t*)(base + best_b + dead_b + sel_b);
	uint32_t *spre = (uint32_t*)(base + best_b + dead_b + sel_b + scr_b);
	unsigned int *live = (unsigned int*)(base + best_b + dead_b + 2 * sel_b + scr_b);
	MCOM_HIP(ctx, hipMemsetAsync(dead, 0, dead_b + sel_b, ctx->stream));
	const unsigned blocks = (unsigned)((n_pairs + 255) / 256);
	int rounds = 0;
	for (;;) {
		if (rounds >= max_rounds) return mcom_fail(ctx, MCOM_E_OVERFLOW, "claiming did not settle in %d rounds", max_rounds);
		MCOM_HIP(ctx, hipMemsetAsync(best, 0xFF, best_b, ctx->stream));
		MCOM_HIP(ctx, hipMemsetAsync(live, 0, 4, ctx->stream));
		MCOM_LAUNCH(k_claim_bid, dim3(blocks), dim3(256), 0, ctx->stream, d_pairs, n_pairs, d_flag, dead, best, live);
		MCOM_LAUNCH(k_claim_take, dim3(blocks), dim3(256), 0, ctx->stream, d_pairs, n_pairs, d_flag, dead, best, sel);
		MCOM_LAUNCH_CHECK(ctx);
		unsigned int hl = 0;
		MCOM_HIP(ctx, mcom_d2h_async(ctx, &hl, live, 4));
		MCOM_HIP(ctx, mcom_stream_sync(ctx));
		if (!hl) break;
		++rounds;
	}
	if (h_rounds) *h_rounds = rounds;
	// the taken edges in list order = the reference's claiming order
	if ((rc = mcom_scan_u32(ctx, sel, spre, n_pairs + 1, scr))) return rc;          // sel[n_pairs] = 0 from the memset
	uint32_t nj = 0;
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &nj, spre + n_pairs, 4));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	*h_nj = nj;
	if (nj) MCOM_LAUNCH(k_claim_jobs, dim3(blocks), dim3(256), 0, ctx->stream, d_pairs, n_pairs, sel, spre, d_jobs);
	MCOM_LAUNCH_CHECK(ctx);
	MCOM_HIP(ctx, mcom_stream_sync(ctx));                                // the workspace is in use until here
	return MCOM_OK;
}
