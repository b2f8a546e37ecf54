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

__global__ void k_claim_jobs(const mcom_mm128 *__restrict__ pairs, size_t n, const uint32_t *__restrict__ sel, const uint32_t *__restrict__ spre,
                             uint32_t *__restrict__ jobs)
{
	const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= n || !sel[e]) return;
	const mcom_mm128 pr = pairs[e];
	uint32_t *j = jobs + 4 * (size_t)spre[e];
	j[0] = (uint32_t)(pr.x >> 32); j[1] = (uint32_t)(pr.y >> 32); j[2] = (uint32_t)pr.x >> 1; j[3] = (uint32_t)pr.y >> 1;
}

// ---- round 4: all rounds in ONE launch ------------------------------------------------------------------------------------------
// The loop above cost two launches, two clears and a host round trip per round (25 rounds in the benchmark's first merge round, a
// handful in the others: ~150 launches and ~40 round trips per step).  k_claim_all runs the same rounds inside one cooperative launch:
// 4 workgroups per CU (a CU holds 8 of these: every one is resident, which a grid barrier needs), every workgroup owning a fixed
// stripe of the edge list, two grid barriers per round.  A bid is { round << 32 | ~edge } under atomicMax, so the bids of a later
// round override those of an earlier one and `best` needs no clear between rounds.  The barrier is the guide's recipe (each wave's
// stores drained by __syncthreads, one lane's agent-scope release, a counter, that lane's agent-scope acquire, __syncthreads) and its
// wait is bounded: a workgroup that gives up raises the context's poison flag and leaves -- the others leave at their next barrier
// for the same reason -- and the host falls back to the launch-per-round loop.
#define CL_THREADS 256
__device__ __forceinline__ bool cl_barrier(unsigned int *bar, unsigned int G, unsigned int &gen, unsigned int *poison)
{
	__shared__ int ok_s;
	__syncthreads();
	if (threadIdx.x == 0) {
		__threadfence();
		++gen;
		atomicAdd(bar, 1u);
		const unsigned int want = gen * G;
		unsigned int polls = 0; int ok = 1;
		while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
			if (++polls > (1u << 24) || ((polls & 4095u) == 0 && *(volatile unsigned int*)poison)) { *poison = 1u; ok = 0; break; }   // (the flag lives in host memory: looked at rarely)
			__builtin_amdgcn_s_sleep(2);
		}
		__threadfence();
		ok_s = ok;
	}
	__syncthreads();
	return ok_s != 0;
}
__global__ __launch_bounds__(CL_THREADS) void k_claim_all(const mcom_mm128 *__restrict__ pairs, uint32_t n, uint8_t *matched, uint8_t *dead,
                                                         unsigned long long *best, uint32_t *__restrict__ sel, unsigned int *state, int max_rounds, unsigned int *poison)
{
	// state[0] = barrier counter, state[1 + (round & 1)] = "some edge is live in this round", state[3] = rounds done, state[4] = did not settle
	const unsigned int G = gridDim.x;
	const uint32_t stride = G * CL_THREADS;
	unsigned int gen = 0;
	for (int round = 1; ; ++round) {
		const unsigned long long rkey = (unsigned long long)round << 32;
		bool live = false;
		for (uint32_t e = blockIdx.x * CL_THREADS + threadIdx.x; e < n; e += stride) {
			if (dead[e]) continue;
			const mcom_mm128 pr = pairs[e];
			const uint32_t ci = (uint32_t)(pr.x >> 32), cj = (uint32_t)(pr.y >> 32);
			if (matched[ci] || matched[cj]) { dead[e] = 1; continue; }
			const unsigned long long key = rkey | (unsigned long long)(0xFFFFFFFFu - e);
			atomicMax(&best[ci], key);
			atomicMax(&best[cj], key);
			live = true;
		}
		if (__any(live) && (threadIdx.x & 63) == 0 && *(volatile unsigned int*)(state + 1 + (round & 1)) == 0) state[1 + (round & 1)] = 1;
		if (!cl_barrier(state, G, gen, poison)) return;
		const bool any = *(volatile unsigned int*)(state + 1 + (round & 1)) != 0;
		if (!any || round > max_rounds) { if (blockIdx.x == 0 && threadIdx.x == 0) { state[3] = (unsigned int)(round - 1); state[4] = any ? 1u : 0u; } return; }
		for (uint32_t e = blockIdx.x * CL_THREADS + threadIdx.x; e < n; e += stride) {
			if (dead[e]) continue;
			const mcom_mm128 pr = pairs[e];
			const uint32_t ci = (uint32_t)(pr.x >> 32), cj = (uint32_t)(pr.y >> 32);
			const unsigned long long key = rkey | (unsigned long long)(0xFFFFFFFFu - e);
			if (best[ci] == key && best[cj] == key) { matched[ci] = 1; matched[cj] = 1; sel[e] = 1; dead[e] = 1; }
		}
		if (blockIdx.x == 0 && threadIdx.x == 0) state[1 + ((round + 1) & 1)] = 0;           // the next round's flag (nobody reads or writes it before the barrier below)
		if (!cl_barrier(state, G, gen, poison)) return;
	}
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
	int rc = mcom_scan_prepare(ctx);                                               // (the poison flag)
	if (rc) return rc;
	auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
	const size_t best_b = al(n_contigs * 8), dead_b = al(n_pairs), sel_b = al((n_pairs + 1) * 4);
	rc = mcom_ws_reserve(ctx, best_b + dead_b + 2 * sel_b + 512);
	if (rc) return rc;
	char *base = (char*)ctx->ws;
	unsigned long long *best = (unsigned long long*)base;
	uint8_t *dead = (uint8_t*)(base + best_b);
	uint32_t *sel = (uint32_t*)(base + best_b + dead_b);
	uint32_t *spre = (uint32_t*)(base + best_b + dead_b + sel_b);
	unsigned int *state = (unsigned int*)(base + best_b + dead_b + 2 * sel_b);
	MCOM_HIP(ctx, hipMemsetAsync(base, 0, best_b + dead_b + sel_b, ctx->stream));   // best = 0: below every bid; dead, sel = 0
	MCOM_HIP(ctx, hipMemsetAsync(state, 0, 64, ctx->stream));
	const unsigned blocks = (unsigned)((n_pairs + 255) / 256);
	unsigned int hs[5] = {0, 0, 0, 0, 0};
	{
		unsigned grid = (unsigned)ctx->n_cu * 4;
		if (grid > blocks) grid = blocks;
		uint32_t n32 = (uint32_t)n_pairs;
		void *args[] = { (void*)&d_pairs, (void*)&n32, (void*)&d_flag, (void*)&dead, (void*)&best, (void*)&sel, (void*)&state, (void*)&max_rounds, (void*)&ctx->d_poison };
		if (ctx->prof_on) ++ctx->prof_kernels[std::make_pair(ctx->prof_cur, (const void*)&k_claim_all)];
		const hipError_t e = hipLaunchCooperativeKernel((const void*)k_claim_all, dim3(grid), dim3(CL_THREADS), args, 0, ctx->stream);
		if (e != hipSuccess) { (void)hipGetLastError(); return mcom_fail(ctx, MCOM_E_HIP, "claiming: cooperative launch: %s", hipGetErrorString(e)); }
		MCOM_HIP(ctx, mcom_d2h_async(ctx, hs, state, 20));
		MCOM_HIP(ctx, mcom_stream_sync(ctx));
	}
	if (hs[4]) return mcom_fail(ctx, MCOM_E_OVERFLOW, "claiming did not settle in %d rounds", max_rounds);
	if (h_rounds) *h_rounds = (int)hs[3];
	// the taken edges in list order = the reference's claiming order
	if ((rc = mcom_scan_u32(ctx, sel, spre, n_pairs + 1, nullptr))) return rc;      // sel[n_pairs] = 0 from the memset
	uint32_t nj = 0;
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &nj, spre + n_pairs, 4));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	*h_nj = nj;
	if (nj) MCOM_LAUNCH(k_claim_jobs, dim3(blocks), dim3(256), 0, ctx->stream, d_pairs, n_pairs, sel, spre, d_jobs);
	MCOM_LAUNCH_CHECK(ctx);
	MCOM_HIP(ctx, mcom_stream_sync(ctx));                                // the workspace is in use until here
	return MCOM_OK;
}
