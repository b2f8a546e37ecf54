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

// (cap: jobs the caller's array holds -- the kernel also runs on what a claiming launch whose barrier gave up left behind, which
// nobody looks at but which must not be written past the array)
__global__ void k_claim_jobs(const mcom_mm128 *__restrict__ pairs, size_t n, const uint32_t *__restrict__ sel, const uint32_t *__restrict__ spre,
                             uint32_t *__restrict__ jobs, uint32_t cap)
{
	const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= n || !sel[e] || spre[e] >= cap) return;
	const mcom_mm128 pr = pairs[e];
	uint32_t *j = jobs + 4 * (size_t)spre[e];
	j[0] = (uint32_t)(pr.x >> 32); j[1] = (uint32_t)(pr.y >> 32); j[2] = (uint32_t)pr.x >> 1; j[3] = (uint32_t)pr.y >> 1;
}

// ---- round 4: all rounds in ONE launch ------------------------------------------------------------------------------------------
// The loop above cost two launches, two clears and a host round trip per round (25 rounds in the benchmark's first merge round, a
// handful in the others: ~150 launches and ~40 round trips per step).  k_claim_all runs the same rounds inside one launch:
// one workgroup of 1024 threads per CU (every one is resident, which a grid barrier needs), every workgroup owning a fixed
// stripe of the edge list, one grid barrier per round.  A bid is { round << 32 | ~edge } under atomicMax, so the bids of a later
// round override those of an earlier one and `best` needs no clear between rounds.  The barrier is the guide's recipe (each wave's
// stores drained by __syncthreads, one lane's agent-scope release, a counter, that lane's agent-scope acquire, __syncthreads) and its
// wait is bounded: a workgroup that gives up raises the context's poison flag and leaves -- the others leave at their next barrier
// for the same reason -- and mcom_claim_pairs redoes the claiming with the launch-per-round loop below (k_claim_bid / k_claim_take),
// which needs no co-residency.  Residency is NOT guaranteed by the grid's size alone: another process or thread-rank on the card
// running its own persistent kernel, or a CU mask, can keep a workgroup from starting; then the waits run out and the loop takes over.
#define CL_THREADS 1024
#define CL_REL_LINES 16                     // copies of the release word, one cache line each: 16 pollers per line instead of 256 on one
#define CL_STATE_WORDS (64 + CL_REL_LINES * 32)
// One workgroup of 1024 threads per CU: a barrier of 256 arrivals (the first form had 1024 workgroups of 256 threads polling ONE word
// with agent-scope loads while the late arrivals' atomics queued behind the polls on that word's L2 channel: ~50 us per barrier,
// 13 ms per step against 6.2 ms for the launch-per-round loop).  The last to arrive releases everybody through CL_REL_LINES words.
__device__ __forceinline__ bool cl_barrier(unsigned int *state, unsigned int G, unsigned int &gen, unsigned int *poison, unsigned int poll_limit)
{
	__shared__ int ok_s;
	__syncthreads();
	if (threadIdx.x == 0) {
		__threadfence();
		++gen;
		unsigned int *rel = state + 64;
		int ok = 1;
		if (poll_limit == 0) { *poison = 1u; ok = 0; }                           // (test hook, mcom_set_claim_route(2): give up at once)
		else if (atomicAdd(state, 1u) == gen * G - 1u) {
			for (int j = 0; j < CL_REL_LINES; ++j) __hip_atomic_store(rel + 32 * j, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		} else {
			unsigned int *mine = rel + 32 * (blockIdx.x % CL_REL_LINES);
			unsigned int polls = 0;
			while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
				if (++polls > poll_limit || ((polls & 4095u) == 0 && *(volatile unsigned int*)poison)) { *poison = 1u; ok = 0; break; }   // (the flag lives in host memory: looked at rarely)
				__builtin_amdgcn_s_sleep(8);
			}
		}
		__threadfence();
		ok_s = ok;
	}
	__syncthreads();
	return ok_s != 0;
}
// ONE barrier per round: a phase takes the winners of the round before and places the bids of its own round.  Bids of odd and even
// rounds go to two arrays, so a take never meets a bid of the next round.  What the missing barrier between "take" and "bid" allows
// is a stale bid: an edge bids although one of its ends was matched in this very phase.  It cannot be taken (a take looks at the
// matched flags again, a barrier later) and it cannot change the result: an edge that IS taken was the smallest bidder at both ends
// among bids that include every edge still truly alive there, and its ends were free -- the greedy matching's own rule.  A stale bid
// can only make a live edge wait one more round (every edge is stale at most once).
//
// THE TAIL (round 5, second half): a round of the grid costs ~100 us whatever is left to do -- 256 workgroups meeting at the barrier, all
// dead flags read once more -- and most rounds of a step have next to nothing left (43 rounds per step at 100 M reads, a few thousand
// live edges after the first handful).  When a round ends with at most CL_TAIL edges alive, the next one LISTS the edges that bid in it,
// and from there ONE workgroup runs the same phases over the list alone, meeting at __syncthreads; the other workgroups leave.
#define CL_TAIL 16384
// what an edge does in a phase; true: it placed its bids (it is alive).  skip_ci: its bid at the query's end is left out (see below)
// Round 5: "this contig is taken" lives in the bid words themselves -- a take writes CL_TAKEN (above every bid) into both bid arrays
// at both ends -- so a phase reads ONE random word per end (the winner of the round before, or CL_TAKEN) where it read two (that and
// the contig's flag byte).  The flag array is still written, for the caller; nobody here reads it.
#define CL_TAKEN 0xFFFFFFFFFFFFFFFFull
__device__ __forceinline__ bool cl_edge(const mcom_mm128 *__restrict__ pairs, uint32_t e, int round, uint8_t *matched, uint8_t *dead, unsigned long long *bestC,
                                        unsigned long long *bestP, unsigned long long rkey, unsigned long long pkey, uint32_t *__restrict__ sel, uint32_t &bid_ci)
{
	const mcom_mm128 pr = pairs[e];
	const uint32_t ci = (uint32_t)(pr.x >> 32), cj = (uint32_t)(pr.y >> 32);
	const unsigned long long inv = (unsigned long long)(0xFFFFFFFFu - e);
	if (round > 1) {                                                            // (nothing is taken before the first takes: the phase that sees every edge reads nothing here)
		const unsigned long long pi = bestP[ci], pj = bestP[cj];
		if (pi == (pkey | inv) && pj == (pkey | inv)) {                        // the winner of the round before at both ends, and both still free
			matched[ci] = 1; matched[cj] = 1; sel[e] = 1;
			bestP[ci] = CL_TAKEN; bestP[cj] = CL_TAKEN; bestC[ci] = CL_TAKEN; bestC[cj] = CL_TAKEN;
			dead[e] = 1;
			return false;
		}
		if (pi == CL_TAKEN || pj == CL_TAKEN) { dead[e] = 1; return false; }   // (also the stale winner: an end was taken after it had bid)
	}
	// The edges of one query lie one behind the other and share the end ci: an earlier one that bids in this phase is the smaller
	// bidder there whatever this one does, so this one's bid at ci is left out (round 5: a third of the atomics of the first
	// round).  At cj it must bid: it stands in front of later queries' edges there until its own fate is known.
	if (bid_ci != ci) atomicMax(&bestC[ci], rkey | inv);
	atomicMax(&bestC[cj], rkey | inv);
	bid_ci = ci;
	return true;
}
__global__ __launch_bounds__(CL_THREADS) void k_claim_all(const mcom_mm128 *__restrict__ pairs, uint32_t n, uint8_t *matched, uint8_t *dead,
                                                         unsigned long long *best, uint32_t n_contigs, uint32_t *__restrict__ sel, unsigned int *state, int max_rounds,
                                                         unsigned int *poison, unsigned int *__restrict__ host_copy, unsigned int poll_limit, uint32_t *__restrict__ tail_list, uint32_t tail_max)
{
	// state[0] = barrier counter; on a line of their own: state[32 + round % 3] = edges that bid in this round, state[35] = entries of the tail's
	// list, state[36] = rounds with bids, state[37] = did not settle, state[38] = the round the tail began with (0: no tail); state[64 ...] = the release words
	const unsigned int G = gridDim.x;
	const uint32_t stride = G * CL_THREADS, chunks = (n + 15u) >> 4;
	unsigned int gen = 0;
	__shared__ unsigned int wg_live;
	auto finish = [&](int round, bool any) {
		const unsigned int r36 = (unsigned int)(any ? round - 1 : round - 2), r37 = any ? 1u : 0u;
		state[36] = r36; state[37] = r37;
		if (host_copy) { host_copy[0] = r36; host_copy[1] = r37; }               // (pinned memory: the read-back needs no copy, scan.hip)
	};
	int round = 1;
	bool listing = false;                                                        // this round's bidders go to the tail's list
	for (; ; ++round) {
		if (round > 1) {
			const unsigned int alive = *(volatile unsigned int*)(state + 32 + (round - 1) % 3);
			if (!alive || round - 1 >= max_rounds) {
				if (blockIdx.x == 0 && threadIdx.x == 0) finish(round, alive != 0);
				return;
			}
			if (listing) break;                                                        // the list is made: the tail below
			listing = alive <= tail_max;
		}
		if (threadIdx.x == 0) wg_live = 0;
		__syncthreads();
		const unsigned long long rkey = (unsigned long long)round << 32, pkey = (unsigned long long)(round - 1) << 32;
		unsigned long long *bestC = best + (size_t)(round & 1) * n_contigs;
		unsigned long long *bestP = best + (size_t)((round - 1) & 1) * n_contigs;
		uint32_t live = 0;
		// A wave takes 1024 consecutive edges at a time, lane = edge within a row of 64: the records of a row are ONE kilobyte (round 5; a
		// thread used to own sixteen consecutive edges, so that a wave's load touched 64 lines 256 bytes apart, sixteen times over).  The
		// dead flags come first, sixteen per lane in one load (most edges are dead after the first rounds): a block without a live edge
		// costs that load, a row without one nothing.
		const uint32_t lane = threadIdx.x & 63u, wv = (blockIdx.x * CL_THREADS + threadIdx.x) >> 6, n_wv = stride >> 6, blocks1k = (n + 1023u) >> 10;
		for (uint32_t kb = wv; kb < blocks1k; kb += n_wv) {
			const uint32_t e0 = kb << 10, ch = (e0 >> 4) + lane;
			uint4 d16 = make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u);
			if (ch < chunks) d16 = ((const uint4*)dead)[ch];
			const unsigned long long some = __ballot(d16.x != 0x01010101u || d16.y != 0x01010101u || d16.z != 0x01010101u || d16.w != 0x01010101u);
			if (!some) continue;
			uint32_t carry_ci = 0xFFFFFFFFu; bool carry_bid = false;                   // the last edge of the row before (when that row was looked at)
			for (uint32_t j = 0; j < 16; ++j) {
				if (!((some >> (4 * j)) & 0xFull)) { carry_ci = 0xFFFFFFFFu; carry_bid = false; continue; }   // the row's four chunks are dead
				const uint32_t e = e0 + 64u * j + lane;
				bool bids = false;
				uint32_t ci = 0xFFFFFFFFu, cj = 0;
				unsigned long long inv = 0;
				if (e < n && !dead[e]) {                                                  // (the line came with the load above)
					const mcom_mm128 pr = pairs[e];
					ci = (uint32_t)(pr.x >> 32); cj = (uint32_t)(pr.y >> 32);
					inv = (unsigned long long)(0xFFFFFFFFu - e);
					bids = true;
					if (round > 1) {                                                       // (nothing is taken before the first takes)
						const unsigned long long pi = bestP[ci], pj = bestP[cj];
						if (pi == (pkey | inv) && pj == (pkey | inv)) {                   // the winner of the round before at both ends, and both still free
							matched[ci] = 1; matched[cj] = 1; sel[e] = 1;
							bestP[ci] = CL_TAKEN; bestP[cj] = CL_TAKEN; bestC[ci] = CL_TAKEN; bestC[cj] = CL_TAKEN;
							dead[e] = 1; bids = false;
						} else if (pi == CL_TAKEN || pj == CL_TAKEN) { dead[e] = 1; bids = false; }
					}
				}
				// the edges of one query lie one behind the other and share the end ci: when the edge right in front bids in this phase it is
				// the smaller bidder there whatever this one does, and this one's bid at ci is left out (a third of the first round's atomics)
				uint32_t pci = (uint32_t)__shfl_up((int)ci, 1); int pb = __shfl_up(bids ? 1 : 0, 1);
				if (lane == 0) { pci = carry_ci; pb = carry_bid ? 1 : 0; }
				if (bids) {
					if (!(pb && pci == ci)) atomicMax(&bestC[ci], rkey | inv);
					atomicMax(&bestC[cj], rkey | inv);
					++live;
					if (listing) tail_list[atomicAdd(state + 35, 1u)] = e;                  // (at most tail_max of them: who bids now bid in the round before)
				}
				carry_ci = (uint32_t)__shfl((int)ci, 63); carry_bid = __shfl(bids ? 1 : 0, 63) != 0;
			}
		}
		for (int o = 32; o; o >>= 1) live += __shfl_xor(live, o);
		if ((threadIdx.x & 63) == 0 && live) atomicAdd(&wg_live, live);
		__syncthreads();
		if (threadIdx.x == 0 && wg_live) atomicAdd(state + 32 + round % 3, wg_live);
		if (blockIdx.x == 0 && threadIdx.x == 0) state[32 + (round + 1) % 3] = 0;                // the next round's count (nobody else touches it in this phase)
		if (!cl_barrier(state, G, gen, poison, poll_limit)) return;
	}
	// ---- the tail: one workgroup, the listed edges, the same phases
	if (blockIdx.x != 0) return;
	const uint32_t n_list = *(volatile unsigned int*)(state + 35);
	if (threadIdx.x == 0) state[38] = (unsigned int)round;
	for (; ; ++round) {
		// (round's predecessor had bidders -- the loop above or the end of the last turn saw to that -- and the budget holds)
		if (threadIdx.x == 0) wg_live = 0;
		__syncthreads();
		const unsigned long long rkey = (unsigned long long)round << 32, pkey = (unsigned long long)(round - 1) << 32;
		unsigned long long *bestC = best + (size_t)(round & 1) * n_contigs;
		unsigned long long *bestP = best + (size_t)((round - 1) & 1) * n_contigs;
		uint32_t live = 0;
		for (uint32_t i = threadIdx.x; i < n_list; i += CL_THREADS) {
			const uint32_t e = tail_list[i];
			if (((volatile uint8_t*)dead)[e]) continue;
			uint32_t bid_ci = 0xFFFFFFFFu;                                              // (every edge bids at both ends here)
			live += cl_edge(pairs, e, round, matched, dead, bestC, bestP, rkey, pkey, sel, bid_ci) ? 1u : 0u;
		}
		for (int o = 32; o; o >>= 1) live += __shfl_xor(live, o);
		if ((threadIdx.x & 63) == 0 && live) atomicAdd(&wg_live, live);
		__threadfence();                                                             // bids, flags and takes of this phase: out, and nothing stale kept for the next
		__syncthreads();
		const unsigned int alive = wg_live;
		__syncthreads();
		if (!alive || round >= max_rounds) {
			if (threadIdx.x == 0) finish(round + 1, alive != 0);
			return;
		}
	}
}

// ---- the launch-per-round loop (rounds 1-3's form, back as the fallback): one launch places the bids of the live edges, one takes
// the edges that won at both ends; the host reads "some edge bid" after every round.  Bids carry their round, so nothing is cleared
// between rounds.  No workgroup waits for another: this form works whatever else occupies the card.
__global__ void k_claim_bid(const mcom_mm128 *__restrict__ pairs, uint32_t n, const uint8_t *__restrict__ matched, uint8_t *__restrict__ dead,
                            unsigned long long *__restrict__ best, uint32_t round, unsigned int *__restrict__ any)
{
	const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
	bool live = false;
	if (e < n && !dead[e]) {
		const mcom_mm128 pr = pairs[e];
		const uint32_t ci = (uint32_t)(pr.x >> 32), cj = (uint32_t)(pr.y >> 32);
		if (matched[ci] || matched[cj]) dead[e] = 1;
		else {
			const unsigned long long key = ((unsigned long long)round << 32) | (unsigned long long)(0xFFFFFFFFu - e);
			atomicMax(&best[ci], key); atomicMax(&best[cj], key);
			live = true;
		}
	}
	if (__any(live) && (threadIdx.x & 63) == 0 && *(volatile unsigned int*)any == 0) *any = 1u;
}
__global__ void k_claim_take(const mcom_mm128 *__restrict__ pairs, uint32_t n, uint8_t *__restrict__ matched, uint8_t *__restrict__ dead,
                             const unsigned long long *__restrict__ best, uint32_t round, uint32_t *__restrict__ sel)
{
	const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= n || dead[e]) return;
	const mcom_mm128 pr = pairs[e];
	const uint32_t ci = (uint32_t)(pr.x >> 32), cj = (uint32_t)(pr.y >> 32);
	const unsigned long long key = ((unsigned long long)round << 32) | (unsigned long long)(0xFFFFFFFFu - e);
	if (best[ci] == key && best[cj] == key) { matched[ci] = 1; matched[cj] = 1; sel[e] = 1; dead[e] = 1; }   // the earliest live edge at both ends: nobody else writes these flags in this launch
}

// the flags of the contigs and the kernel's own arrays cleared in ONE launch (two or three fills before: the runtime splits a fill whose
// size is not a multiple of four)
__global__ void k_claim_clear(uint8_t *__restrict__ flag, size_t n_flag, uint4 *__restrict__ base16, size_t n16)
{
	const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = t; i < n16; i += stride) base16[i] = make_uint4(0, 0, 0, 0);
	const size_t head = ((16 - ((uintptr_t)flag & 15)) & 15) < n_flag ? ((16 - ((uintptr_t)flag & 15)) & 15) : n_flag;   // bytes in front of the first 16-byte boundary
	const size_t mid16 = (n_flag - head) / 16;
	uint4 *f16 = (uint4*)(flag + head);
	for (size_t i = t; i < mid16; i += stride) f16[i] = make_uint4(0, 0, 0, 0);
	if (t < head) flag[t] = 0;
	const size_t tail0 = head + mid16 * 16;
	if (t < n_flag - tail0) flag[tail0 + t] = 0;
}

extern "C" int mcom_claim_pairs(mcom_ctx *ctx, const mcom_mm128 *d_pairs, size_t n_pairs, size_t n_contigs, int max_rounds, uint32_t *d_jobs,
                                uint8_t *d_flag, uint64_t *h_nj, int *h_rounds)
{
	if (!ctx || !h_nj) return MCOM_E_ARG;
	*h_nj = 0; if (h_rounds) *h_rounds = 0;
	if (n_contigs && !d_flag) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	if (n_pairs == 0) { if (n_contigs) MCOM_HIP(ctx, hipMemsetAsync(d_flag, 0, n_contigs, ctx->stream)); return MCOM_OK; }
	if (!d_pairs || !d_jobs || n_pairs >= (1ull << 32) - 1 || n_contigs >= (1ull << 32) - 1) return mcom_fail(ctx, MCOM_E_ARG, "bad claim arguments");
	int rc = mcom_scan_prepare(ctx);                                               // (the poison flag)
	if (rc) return rc;
	auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
	const size_t best_b = al(2 * n_contigs * 8), dead_b = al(n_pairs + 16), sel_b = al((n_pairs + 1) * 4);
	rc = mcom_ws_reserve(ctx, best_b + dead_b + 2 * sel_b + al(CL_STATE_WORDS * 4) + al(CL_TAIL * 4) + 256);
	if (rc) return rc;
	char *base = (char*)ctx->ws;
	unsigned long long *best = (unsigned long long*)base;
	uint8_t *dead = (uint8_t*)(base + best_b);
	uint32_t *sel = (uint32_t*)(base + best_b + dead_b);
	unsigned int *state = (unsigned int*)(base + best_b + dead_b + sel_b);
	uint32_t *spre = (uint32_t*)(base + best_b + dead_b + sel_b + al(CL_STATE_WORDS * 4));
	uint32_t *tail_list = (uint32_t*)(base + best_b + dead_b + 2 * sel_b + al(CL_STATE_WORDS * 4));
	const unsigned blocks = (unsigned)((n_pairs + 255) / 256);
	unsigned int hs[2] = {0, 0};
	uint32_t nj = 0;
	bool listed = false;                                                          // the jobs are in d_jobs and nj is known
	const uint32_t job_cap = (uint32_t)(n_contigs / 2 + 1);                          // (an edge taken uses up two contigs; the caller's array holds that many jobs)
	const size_t n16 = (best_b + dead_b + sel_b + al(CL_STATE_WORDS * 4)) / 16;
	size_t cb = (n16 + 255) / 256; if (cb > (size_t)ctx->n_cu * 8) cb = (size_t)ctx->n_cu * 8;
	// best = 0: below every bid; dead, sel, state = 0; and the contigs' flags
	MCOM_LAUNCH(k_claim_clear, dim3((unsigned)cb), dim3(256), 0, ctx->stream, d_flag, n_contigs, (uint4*)base, n16);
	bool loop = ctx->claim_route == 1;
	if (!loop) {
		unsigned grid = (unsigned)ctx->n_cu;                                       // (round 5, tried: two workgroups per CU -- they are not resident together, the barrier waits ran out and the loop below took over: 5.8 s)
		const unsigned need = (unsigned)((n_pairs + 16 * CL_THREADS - 1) / (16 * CL_THREADS));
		if (grid > need) grid = need;
		uint32_t n32 = (uint32_t)n_pairs;
		// a plain launch of one workgroup per CU (rocprofv3's kernel trace crashed on a cooperative one).  Every workgroup is resident as long
		// as nothing else holds the CUs; when something does, the barrier waits run out, the poison flag trips and the loop below takes over.
		// (a stale bid can make a live edge wait one more round, every edge at most once: the kernel's own budget is twice the caller's)
		uint32_t ring = 0;
		unsigned int *host_copy = (unsigned int*)mcom_ring_slot(ctx, &ring);
		MCOM_LAUNCH(k_claim_all, dim3(grid), dim3(CL_THREADS), 0, ctx->stream, d_pairs, n32, d_flag, dead, best, (uint32_t)n_contigs, sel, state, 2 * max_rounds + 2, ctx->d_poison, host_copy,
		            ctx->claim_route == 2 ? 0u : (1u << 23), tail_list, ctx->claim_route == 3 ? 0u : (uint32_t)CL_TAIL);
		if (host_copy) mcom_ring_register(ctx, state + 36, 8, ring);
		MCOM_LAUNCH_CHECK(ctx);
		MCOM_HIP(ctx, mcom_d2h_async(ctx, hs, state + 36, 8));
		// The taken edges in list order and the jobs follow at once, before anybody knows how the launch ended: one round trip for
		// { rounds, settled, jobs } instead of three (the poison flag, the count, the end of the workspace's use).  A launch whose barrier
		// gave up leaves flags that mean nothing: what was made of them is made again below.
		if ((rc = mcom_scan_u32(ctx, sel, spre, n_pairs + 1, nullptr))) return rc;     // sel[n_pairs] = 0 from the clear
		MCOM_LAUNCH(k_claim_jobs, dim3(blocks), dim3(256), 0, ctx->stream, d_pairs, n_pairs, sel, spre, d_jobs, job_cap);
		MCOM_LAUNCH_CHECK(ctx);
		MCOM_HIP(ctx, mcom_d2h_async(ctx, &nj, spre + n_pairs, 4));
		bool poisoned = false;
		MCOM_HIP(ctx, mcom_stream_sync_poison(ctx, &poisoned));
		if (poisoned) {                                                               // what the kernel left is partial: start again
			loop = true; hs[0] = hs[1] = 0;
			MCOM_LAUNCH(k_claim_clear, dim3((unsigned)cb), dim3(256), 0, ctx->stream, d_flag, n_contigs, (uint4*)base, n16);
		} else listed = true;
	}
	if (loop) {
		++ctx->claim_fallbacks;
		const uint32_t n32 = (uint32_t)n_pairs;
		int round = 1;
		for (;; ++round) {
			if (round > max_rounds) { hs[0] = (unsigned)max_rounds; hs[1] = 1; break; }
			MCOM_HIP(ctx, hipMemsetAsync(state + 32, 0, 4, ctx->stream));
			MCOM_LAUNCH(k_claim_bid, dim3(blocks), dim3(256), 0, ctx->stream, d_pairs, n32, d_flag, dead, best, (uint32_t)round, state + 32);
			MCOM_LAUNCH(k_claim_take, dim3(blocks), dim3(256), 0, ctx->stream, d_pairs, n32, d_flag, dead, best, (uint32_t)round, sel);
			MCOM_LAUNCH_CHECK(ctx);
			unsigned int any = 0;
			MCOM_HIP(ctx, mcom_d2h_async(ctx, &any, state + 32, 4));
			MCOM_HIP(ctx, mcom_stream_sync(ctx));
			if (!any) { hs[0] = (unsigned)(round - 1); hs[1] = 0; break; }
		}
	}
	if (hs[1]) return mcom_fail(ctx, MCOM_E_OVERFLOW, "claiming did not settle in %d rounds", max_rounds);
	if (h_rounds) *h_rounds = (int)hs[0];
	// the taken edges in list order = the reference's claiming order
	if (!listed) {
		if ((rc = mcom_scan_u32(ctx, sel, spre, n_pairs + 1, nullptr))) return rc;  // sel[n_pairs] = 0 from the clear
		MCOM_HIP(ctx, mcom_d2h_async(ctx, &nj, spre + n_pairs, 4));
		MCOM_HIP(ctx, mcom_stream_sync(ctx));
		if (nj) MCOM_LAUNCH(k_claim_jobs, dim3(blocks), dim3(256), 0, ctx->stream, d_pairs, n_pairs, sel, spre, d_jobs, job_cap);
		MCOM_LAUNCH_CHECK(ctx);
	}
	*h_nj = nj;
	// (no wait for the last kernel: the workspace belongs to this context's stream, and mcom_ws_reserve waits before it lets go of it)
	return MCOM_OK;
}
