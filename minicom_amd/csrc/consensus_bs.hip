// minicom_amd/csrc/consensus_bs.hip -- construct_ref (reference kthread_bucket.c:69-377) for the groups of up to 31 members,
// which are nearly all of them: column counts as BIT-SLICED vertical counters, one lane per 32 columns of one group.
//
// The wave-per-group kernels (consensus.hip) spend ~500 wave instructions per member: a lane owns one column, so a member
// costs a shuffle, a shift and an add per column on 64 lanes, and the groups hold six members on average.  Here a lane owns
// 32 columns of one group (ceil(2L / 32) lanes per group, 64 / that many groups per wave) and takes a member's 32 bases at
// once as one 64-bit word cut out of the packed row:
//     lo / hi bit of every base on the even bit positions; P = [A on even | C on odd bits], Q = [G even | T odd] (one-hot);
//     adding a member is a ripple-carry add of the 1-bit planes P, Q into BS_K-bit vertical counters: 2 word operations
//     per counter bit, for 32 columns at once, whatever the bases are;
//     the majority base per column is a bit-sliced tournament A, C, G, T with strict '>' (ties to the smaller code, :160-170);
//     the mismatches of a member against the first consensus are one xor, one popcount; a rejected member is taken out of
//     the counters again by the mirrored ripple-borrow.
// About 100 lane instructions per member and 32 columns, 6 groups per wave at L = 150: a twentieth of the instructions.
// Groups are handed out in order of their size (a counting sort of the sizes), so the lanes of a wave loop equally long.
// Results are the same arrays as mcom_group_consensus's other kernels write; groups of 32 members or more keep those kernels.
#include "mcom_dev.hpp"

#define BS_K 5                       // counter bits of the group kernel: up to 31 members
#define BS_NMAX 31u
#define BS_KM 7                      // of the merge kernel: up to 127 members reaching one unit (the coverage of the data set decides)
#define BS_EVEN 0x5555555555555555ull

namespace {
template <int K> struct BsCnt { uint64_t p[K], q[K]; };

template <int K> __device__ __forceinline__ void bs_add(BsCnt<K> &c, uint64_t P, uint64_t Q)
{
#pragma unroll
	for (int b = 0; b < K; ++b) { uint64_t t = c.p[b] & P; c.p[b] ^= P; P = t; t = c.q[b] & Q; c.q[b] ^= Q; Q = t; }
}
template <int K> __device__ __forceinline__ void bs_sub(BsCnt<K> &c, uint64_t P, uint64_t Q)
{
#pragma unroll
	for (int b = 0; b < K; ++b) { uint64_t t = ~c.p[b] & P; c.p[b] ^= P; P = t; t = ~c.q[b] & Q; c.q[b] ^= Q; Q = t; }
}
// per column (even bit 2j): lo / hi bit of the majority base, ties to the smaller code; nz: any base counted at all
template <int K> __device__ __forceinline__ void bs_best(const BsCnt<K> &c, uint64_t &lo, uint64_t &hi, uint64_t &nz)
{
	uint64_t best[K], cand[K];
#pragma unroll
	for (int b = 0; b < K; ++b) best[b] = c.p[b] & BS_EVEN;
	lo = 0; hi = 0;
	auto challenge = [&](int code) {
		uint64_t gt = 0, eq = BS_EVEN;
#pragma unroll
		for (int b = K - 1; b >= 0; --b) { gt |= eq & cand[b] & ~best[b]; eq &= ~(cand[b] ^ best[b]); }
#pragma unroll
		for (int b = 0; b < K; ++b) best[b] = (cand[b] & gt) | (best[b] & ~gt);
		lo = (code & 1) ? (lo | gt) : (lo & ~gt);
		hi = (code & 2) ? (hi | gt) : (hi & ~gt);
	};
#pragma unroll
	for (int b = 0; b < K; ++b) cand[b] = (c.p[b] >> 1) & BS_EVEN;
	challenge(1);
#pragma unroll
	for (int b = 0; b < K; ++b) cand[b] = c.q[b] & BS_EVEN;
	challenge(2);
#pragma unroll
	for (int b = 0; b < K; ++b) cand[b] = (c.q[b] >> 1) & BS_EVEN;
	challenge(3);
	nz = 0;
#pragma unroll
	for (int b = 0; b < K; ++b) nz |= best[b];
}
// reverse the order of the 32 bases of a word
__device__ __forceinline__ uint64_t bs_rev(uint64_t x)
{
	x = __brevll(x);
	return ((x >> 1) & BS_EVEN) | ((x & BS_EVEN) << 1);
}
// the even bits of the columns [a, b) of a unit, 0 <= a, b <= 32
__device__ __forceinline__ uint64_t bs_span(int a, int b)
{
	if (b <= a) return 0;
	const uint64_t mb = b >= 32 ? ~0ull : ((1ull << (2 * b)) - 1), ma = (1ull << (2 * a)) - 1;
	return BS_EVEN & mb & ~ma;
}

// ---- the groups in order of their size: bins[n] = groups of n members (n = 32: 32 or more) ------------------------------------
// A workgroup takes a run of 256-group chunks and goes to the global counters once (round 4: with a workgroup per chunk, 30 000
// workgroups queued their atomics on two cache lines -- 0.35 ms of each of these two kernels on 8 M groups).
#define BS_ORDER_GRID 1024
__global__ __launch_bounds__(256) void k_bs_sizes(const uint32_t *__restrict__ goff, uint32_t ng, uint32_t *__restrict__ bins)
{
	__shared__ uint32_t h[33];
	if (threadIdx.x < 33) h[threadIdx.x] = 0;
	__syncthreads();
	const uint32_t chunks = (ng + 255u) / 256u, per = (chunks + gridDim.x - 1) / gridDim.x;
	const uint32_t c0 = blockIdx.x * per, c1 = c0 + per < chunks ? c0 + per : chunks;
	for (uint32_t c = c0; c < c1; ++c) {
		const uint32_t g = c * 256u + threadIdx.x;
		if (g < ng) { const uint32_t n = goff[g + 1] - goff[g]; atomicAdd(&h[n < 32u ? n : 32u], 1u); }
	}
	__syncthreads();
	if (threadIdx.x < 33 && h[threadIdx.x]) atomicAdd(&bins[threadIdx.x], h[threadIdx.x]);
}
// perm: the groups below 32 members, largest first, then the others (which group comes first inside one size does not matter:
// every group writes its own outputs)
__global__ __launch_bounds__(256) void k_bs_order(const uint32_t *__restrict__ goff, uint32_t ng, const uint32_t *__restrict__ bins,
                                                  uint32_t *__restrict__ cursor, uint32_t *__restrict__ perm)
{
	__shared__ uint32_t h[33], base[33];
	if (threadIdx.x < 33) h[threadIdx.x] = 0;
	__syncthreads();
	const uint32_t chunks = (ng + 255u) / 256u, per = (chunks + gridDim.x - 1) / gridDim.x;
	const uint32_t c0 = blockIdx.x * per, c1 = c0 + per < chunks ? c0 + per : chunks;
	for (uint32_t c = c0; c < c1; ++c) {
		const uint32_t g = c * 256u + threadIdx.x;
		if (g < ng) { const uint32_t n = goff[g + 1] - goff[g]; atomicAdd(&h[n < 32u ? n : 32u], 1u); }
	}
	__syncthreads();
	if (threadIdx.x < 33) {
		const uint32_t b = threadIdx.x;
		uint32_t start = 0;
		if (b == 32) { for (uint32_t q = 0; q < 32; ++q) start += bins[q]; }
		else for (uint32_t q = b + 1; q < 32; ++q) start += bins[q];
		base[b] = start + (h[b] ? atomicAdd(&cursor[b], h[b]) : 0u);
		h[b] = 0;
	}
	__syncthreads();
	for (uint32_t c = c0; c < c1; ++c) {
		const uint32_t g = c * 256u + threadIdx.x;
		if (g < ng) { const uint32_t n = goff[g + 1] - goff[g], bin = n < 32u ? n : 32u; perm[base[bin] + atomicAdd(&h[bin], 1u)] = g; }
	}
}

// ---- the consensus of the small groups ---------------------------------------------------------------------------------------
// LG lanes per group (lane u of a group owns columns [32u, 32u + 32)), GPW = 64 / LG groups per wave, one wave per workgroup
__global__ __launch_bounds__(64) void k_group_consensus_bs(const uint64_t *__restrict__ packed, int W, uint64_t *__restrict__ members,
                                                           const uint32_t *__restrict__ goff, const uint32_t *__restrict__ perm, uint32_t nsmall,
                                                           int L, int k_orig, int e, int LG, int GPW,
                                                           uint8_t *__restrict__ keep, uint32_t *__restrict__ nkept,
                                                           uint16_t *__restrict__ svout, uint16_t *__restrict__ reflen,
                                                           uint8_t *__restrict__ refs, int ref_stride)
{
	__shared__ uint32_t S[2][64];            // mismatches of the current member, summed over its group's lanes (two slots, alternating)
	__shared__ uint32_t M[64];               // minimum over a group's lanes
	__shared__ uint64_t CODE[64 + 64];       // second consensus, 2 bits per column: group slot gs owns CODE[gs * (LG + 1) ...], one spare word each
	const int lane = threadIdx.x;
	const int gs = lane / LG, u = lane - gs * LG;
	const uint32_t idx = blockIdx.x * (uint32_t)GPW + (uint32_t)gs;
	const bool valid = gs < GPW && idx < nsmall;
	const int TL = 2 * L;
	const uint32_t g = valid ? perm[idx] : 0u;
	const uint32_t m0 = valid ? goff[g] : 0u;
	const uint32_t n = valid ? goff[g + 1] - m0 : 0u;
	uint32_t nmax = n;
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)nmax, d, 64); nmax = o > nmax ? o : nmax; }
	S[0][lane] = 0; S[1][lane] = 0;
	const int col0 = 32 * u;
	int pos0 = 0;
	if (n) { const uint64_t y = members[m0]; int pos = (int)((uint32_t)y >> 1); if (y & 1) pos = L - pos + k_orig - 2; pos0 = pos; }
	// Member i of the lane's group in three steps, so that the two dependent round trips (member word, then its packed row) of the
	// NEXT members travel while this one is counted: its word y; its offset and the two row words under the lane's columns; the
	// 32 bases cut out of them (x: 2 bits per column) with the columns the read covers (cov, even bits).
	struct Fetch { uint64_t y, wa, wb; int off, s0, sh; bool covered; };
	auto fetch_y = [&](uint32_t i) -> uint64_t { return i < n ? members[m0 + i] : 0ull; };
	auto fetch_rows = [&](uint32_t i, uint64_t y) -> Fetch {
		Fetch f; f.y = y;
		const uint32_t dir = (uint32_t)(y & 1);
		int pos = (int)((uint32_t)y >> 1);
		if (dir) pos = L - pos + k_orig - 2;
		f.off = pos0 - pos;
		f.s0 = col0 - f.off;                                            // read position under the unit's first column
		f.covered = i < n && f.s0 < L && f.s0 + 32 > 0;
		const int t0 = dir ? L - 32 - f.s0 : f.s0;                      // first of the 32 stored bases (may lie before or behind the row)
		const int wj = t0 >> 5;
		f.sh = 2 * (t0 & 31);
		const uint64_t *row = packed + (size_t)(y >> 32) * W;
		f.wa = (f.covered && wj >= 0 && wj < W) ? row[wj] : 0ull;
		f.wb = (f.covered && f.sh && wj + 1 >= 0 && wj + 1 < W) ? row[wj + 1] : 0ull;
		return f;
	};
	auto finish = [&](const Fetch &f, uint64_t &x, uint64_t &cov) {
		x = f.sh ? (f.wa >> f.sh) | (f.wb << (64 - f.sh)) : f.wa;
		if (f.y & 1) x = ~bs_rev(x);                                    // reverse complement (preprocess.c:22-37)
		cov = f.covered ? bs_span(f.s0 < 0 ? -f.s0 : 0, L - f.s0 < 32 ? L - f.s0 : 32) : 0ull;
	};
	auto planes = [&](uint64_t x, uint64_t cov, uint64_t &P, uint64_t &Q) {
		const uint64_t lo = x & BS_EVEN, hi = (x >> 1) & BS_EVEN;
		const uint64_t nh = cov & ~hi, hh = cov & hi;
		P = (nh & ~lo) | ((nh & lo) << 1);
		Q = (hh & ~lo) | ((hh & lo) << 1);
	};
	// ---- pass 1: counts of all members
	BsCnt<BS_K> c;
#pragma unroll
	for (int b = 0; b < BS_K; ++b) { c.p[b] = 0; c.q[b] = 0; }
	{
		uint64_t yn = fetch_y(0);
		Fetch cur = fetch_rows(0, yn);
		yn = fetch_y(1);
		for (uint32_t i = 0; i < nmax; ++i) {
			const Fetch nxt = fetch_rows(i + 1, yn);
			yn = fetch_y(i + 2);
			uint64_t x, cov, P, Q;
			finish(cur, x, cov);
			planes(x, cov, P, Q);
			bs_add(c, P, Q);
			cur = nxt;
		}
	}
	// ---- first consensus: majority base per column; it ends at the first empty column (:172-180)
	uint64_t rlo, rhi, nz;
	bs_best(c, rlo, rhi, nz);
	if (u == 0) M[gs] = (uint32_t)TL;
	__syncthreads();
	{
		const uint64_t empty = ~nz & BS_EVEN;
		if (valid && empty) { const int cnd = col0 + (__ffsll((unsigned long long)empty) - 1) / 2; if (cnd < TL) atomicMin(&M[gs], (uint32_t)cnd); }
	}
	__syncthreads();
	const int ref_len = (int)M[gs];
	const uint64_t inref = bs_span(0, ref_len - col0 < 0 ? 0 : (ref_len - col0 > 32 ? 32 : ref_len - col0));   // the unit's columns inside it
	// ---- pass 2: mismatches against the first consensus (:182-190); the counts of the kept members = all counts minus the rejected members'
	uint32_t nk = 0; int rend = 0;
	uint64_t yn = fetch_y(0);
	Fetch cur = fetch_rows(0, yn);
	yn = fetch_y(1);
	for (uint32_t i = 0; i < nmax; ++i) {
		const Fetch nxt = fetch_rows(i + 1, yn);                        // members[] is rewritten at i only: what travels ahead is still the sketch record
		yn = fetch_y(i + 2);
		uint64_t x, cov, P, Q;
		finish(cur, x, cov);
		const uint64_t y = cur.y; const int off = cur.off;
		const uint64_t lo = x & BS_EVEN, hi = (x >> 1) & BS_EVEN;
		const uint64_t mis = cov & (((lo ^ rlo) | (hi ^ rhi)) | ~inref);
		const uint32_t slot = i & 1u;
		// one wave per workgroup: its LDS operations execute in order, so the sum needs no barrier (which would also wait for the
		// loads travelling ahead) -- only the compiler has to keep the order
		if (mis) atomicAdd(&S[slot][gs], (uint32_t)__popcll(mis));
		__builtin_amdgcn_wave_barrier();
		const uint32_t dif = ((volatile uint32_t*)S[slot])[gs];
		__builtin_amdgcn_wave_barrier();
		if (u == 0) ((volatile uint32_t*)S[slot ^ 1u])[gs] = 0;           // everybody read it an iteration ago
		const bool act = i < n;
		const bool kp = (int)dif <= e;                                   // kthread_bucket.c:189
		if (act && kp) { ++nk; if (off + L > rend) rend = off + L; }
		if (__ballot(act && !kp)) {
			planes(x, (act && !kp) ? cov : 0ull, P, Q);
			bs_sub(c, P, Q);
		}
		if (act && u == 0) { keep[m0 + i] = kp ? 1 : 0; members[m0 + i] = (y >> 32 << 32) | ((uint64_t)off << 1) | (y & 1); }   // :101
		cur = nxt;
	}
	// ---- second consensus over [sv, rend): sv = first column inside the first consensus that a kept member covers
	bs_best(c, rlo, rhi, nz);
	__syncthreads();
	if (u == 0) M[gs] = (uint32_t)ref_len;
	CODE[gs * (LG + 1) + u] = rlo | (rhi << 1);
	if (u == 0) CODE[gs * (LG + 1) + LG] = 0;
	__syncthreads();
	{
		const uint64_t first = nz & inref;
		if (valid && nk && first) atomicMin(&M[gs], (uint32_t)(col0 + (__ffsll((unsigned long long)first) - 1) / 2));
	}
	__syncthreads();
	const int sv = nk ? (int)M[gs] : 0;
	if (valid && nk) {
		uint8_t *out = refs + (size_t)g * ref_stride;
		const uint32_t *w32 = (const uint32_t*)&CODE[gs * (LG + 1)];
		const int n_out = rend - sv;
		for (int d = u; 4 * d < n_out; d += LG) {
			const int bit = 2 * (sv + 4 * d), wi = bit >> 5, sh = bit & 31;
			uint32_t v = w32[wi] >> sh;
			if (sh > 24) v |= w32[wi + 1] << (32 - sh);
			uint32_t ascii = 0;
#pragma unroll
			for (int r = 0; r < 4; ++r) ascii |= ((0x54474341u >> (8 * ((v >> (2 * r)) & 3u))) & 0xFFu) << (8 * r);
			if (4 * d + 4 <= n_out) *(uint32_t*)(out + 4 * d) = ascii;
			else for (int r = 0; 4 * d + r < n_out; ++r) out[4 * d + r] = (uint8_t)(ascii >> (8 * r));
		}
	}
	if (valid && u == 0) { nkept[g] = nk; svout[g] = (uint16_t)sv; reflen[g] = (uint16_t)(nk ? rend - sv : 0); }
}

// ---- construct_ref2 (kthread_cb.c:105-218), the same way: one lane per 32 columns of one job's column range -------------------------
// The members of a job are sorted by offset, so the ones that reach a unit's columns are a run of the list: found by a binary
// search, walked until the first offset behind the unit.  A unit that more than 127 members reach marks its 512-column tile for the
// wave-per-tile kernel (consensus.hip), which then writes the whole tile.
__global__ __launch_bounds__(64) void k_merge_consensus_bs(const uint64_t *__restrict__ packed, int W, const uint64_t *__restrict__ members,
                                                           const uint64_t *__restrict__ joff, const uint64_t *__restrict__ roff,
                                                           const uint32_t *__restrict__ ujob, const uint32_t *__restrict__ uoff, uint32_t n_units, int L,
                                                           uint8_t *__restrict__ refs, const uint32_t *__restrict__ reg_lo, const uint32_t *__restrict__ reg_hi,
                                                           const uint32_t *__restrict__ toff, unsigned int *__restrict__ tflag, uint32_t *__restrict__ tlist,
                                                           unsigned int *__restrict__ tcount, uint32_t depth_cap)
{
	const uint32_t q = blockIdx.x * 64u + threadIdx.x;
	const bool valid = q < n_units;
	const uint32_t j = valid ? ujob[q] : 0u;
	long lo = 0, hi = 0, c0 = 0;
	uint64_t m0 = 0, m1 = 0;
	if (valid) {
		hi = reg_hi ? (long)reg_hi[j] : (long)(roff[j + 1] - roff[j]);
		lo = reg_lo ? (long)reg_lo[j] : 0;
		c0 = lo + 32l * (long)(q - uoff[j]);
		m0 = joff[j]; m1 = joff[j + 1];
	}
	const long ce = c0 + 32 < hi ? c0 + 32 : hi;                          // the unit's columns: [c0, ce)
	// first member whose read reaches column c0: offset + L > c0
	uint64_t a = m0, b = m1;
	while (a < b) { const uint64_t mid = (a + b) >> 1; if ((long)((uint32_t)members[mid] >> 1) + L <= c0) a = mid + 1; else b = mid; }
	BsCnt<BS_KM> c;
#pragma unroll
	for (int t = 0; t < BS_KM; ++t) { c.p[t] = 0; c.q[t] = 0; }
	uint32_t n = 0; bool over = false;
	uint64_t i = a;
	// Round 5: two members deep.  A member costs a chain of two dependent gathers (its word, then two words of its read's row -- a random
	// 64-byte sector of a 4 GB array); with only the next member WORD in flight every member paid the row's latency in full.  Now the row
	// words of member i + 1 and the word of member i + 2 travel while member i is counted.
	auto row_words = [&](uint64_t yy, uint64_t &wa, uint64_t &wb) {
		const long off_ = (long)((uint32_t)yy >> 1);
		const int s0_ = (int)(c0 - off_);
		const int t0_ = (yy & 1) ? L - 32 - s0_ : s0_;
		const int wj_ = t0_ >> 5, sh_ = 2 * (t0_ & 31);
		const uint64_t *row = packed + (size_t)(yy >> 32) * W;
		wa = (wj_ >= 0 && wj_ < W) ? row[wj_] : 0ull;
		wb = (sh_ && wj_ + 1 >= 0 && wj_ + 1 < W) ? row[wj_ + 1] : 0ull;
	};
	uint64_t y = (valid && i < m1) ? members[i] : ~0ull;
	uint64_t y1 = (valid && i + 1 < m1) ? members[i + 1] : ~0ull;
	uint64_t wa = 0, wb = 0;
	if (valid && i < m1 && (long)((uint32_t)y >> 1) < c0 + 32) row_words(y, wa, wb);
	for (;;) {
		const long off = (long)((uint32_t)y >> 1);
		const bool act = valid && !over && i < m1 && off < ce;
		if (!__ballot(act)) break;
		uint64_t wa1 = 0, wb1 = 0, y2 = ~0ull;
		if (act && i + 1 < m1 && (long)((uint32_t)y1 >> 1) < ce) row_words(y1, wa1, wb1);
		if (act && i + 2 < m1) y2 = members[i + 2];
		if (act) {
			if (n == depth_cap) over = true;
			else {
				const uint32_t dir = (uint32_t)(y & 1);
				const int s0 = (int)(c0 - off);                                // read position under the unit's first column (> -32, < L)
				const int t0 = dir ? L - 32 - s0 : s0;
				const int sh = 2 * (t0 & 31);
				uint64_t x = sh ? (wa >> sh) | (wb << (64 - sh)) : wa;
				if (dir) x = ~bs_rev(x);
				const int e0 = L - s0 < (int)(ce - c0) ? L - s0 : (int)(ce - c0);
				const uint64_t cov = bs_span(s0 < 0 ? -s0 : 0, e0);
				const uint64_t xl = x & BS_EVEN, xh = (x >> 1) & BS_EVEN;
				const uint64_t nh = cov & ~xh, hh = cov & xh;
				bs_add(c, (nh & ~xl) | ((nh & xl) << 1), (hh & ~xl) | ((hh & xl) << 1));
				++n;
			}
			++i;
			y = y1; wa = wa1; wb = wb1; y1 = y2;
		}
	}
	if (!valid) return;
	if (over) {
		const uint32_t t = toff[j] + (uint32_t)((c0 - lo) >> 9);
		if (atomicExch(&tflag[t], 1u) == 0u) tlist[atomicAdd(tcount, 1u)] = t;
		return;
	}
	uint64_t rlo, rhi, nz;
	bs_best(c, rlo, rhi, nz);
	const uint64_t code = rlo | (rhi << 1);
	uint8_t *out = refs + roff[j] + c0;
	const int cnt = (int)(ce - c0);
#pragma unroll
	for (int d = 0; d < 8; ++d) {
		if (4 * d >= cnt) break;
		const uint32_t v = (uint32_t)(code >> (8 * d)) & 0xFFu;
		uint32_t ascii = 0;
#pragma unroll
		for (int r = 0; r < 4; ++r) ascii |= ((0x54474341u >> (8 * ((v >> (2 * r)) & 3u))) & 0xFFu) << (8 * r);
		if (4 * d + 4 <= cnt) __builtin_memcpy(out + 4 * d, &ascii, 4);
		else for (int r = 0; 4 * d + r < cnt; ++r) out[4 * d + r] = (uint8_t)(ascii >> (8 * r));
	}
}
}  // namespace

// the groups below 32 members; *d_perm_out (ng uint32, mcom_dmalloc'ed by this call, freed by the caller) holds the other groups
// from index *n_small on
int mcom_group_consensus_small(mcom_ctx *ctx, const uint64_t *d_packed, uint64_t *d_members, const uint32_t *d_group_off, uint32_t n_groups,
                               int L, int k_orig, int e, uint8_t *d_keep, uint32_t *d_nkept, uint16_t *d_sv, uint16_t *d_reflen, uint8_t *d_refs,
                               int ref_stride, uint32_t **d_perm_out, uint32_t *n_small)
{
	*d_perm_out = nullptr; *n_small = 0;
	uint32_t *perm = nullptr;
	if (mcom_dmalloc(&perm, ((size_t)n_groups + 128) * 4) != hipSuccess) return mcom_fail(ctx, MCOM_E_NOMEM, "group order");
	uint32_t *bins = (uint32_t*)mcom_zeroed(ctx, perm + n_groups, 80 * 4), *cursor = bins ? bins + 40 : nullptr;
	hipError_t er = bins ? hipSuccess : hipErrorUnknown;
	const unsigned blocks = std::min<unsigned>((n_groups + 255) / 256, BS_ORDER_GRID);
	if (er == hipSuccess) {
		MCOM_LAUNCH(k_bs_sizes, dim3(blocks), dim3(256), 0, ctx->stream, d_group_off, n_groups, bins);
		MCOM_LAUNCH(k_bs_order, dim3(blocks), dim3(256), 0, ctx->stream, d_group_off, n_groups, bins, cursor, perm);
		er = hipGetLastError();
	}
	uint32_t nbig = 0;
	if (er == hipSuccess) er = mcom_d2h_async(ctx, &nbig, bins + 32, 4);
	if (er == hipSuccess) er = mcom_stream_sync(ctx);
	if (er != hipSuccess) { mcom_dfree(perm); return mcom_fail(ctx, MCOM_E_HIP, "group order: %s", hipGetErrorString(er)); }
	const uint32_t nsmall = n_groups - nbig;
	const int LG = (2 * L + 31) / 32, GPW = 64 / LG;
	if (nsmall) {
		McomProfScope ps_(ctx, PROF_CONSENSUS);
		MCOM_LAUNCH(k_group_consensus_bs, dim3((nsmall + GPW - 1) / GPW), dim3(64), 0, ctx->stream, d_packed, mcom_words_per_read(L), d_members, d_group_off, perm,
		                   nsmall, L, k_orig, e, LG, GPW, d_keep, d_nkept, d_sv, d_reflen, d_refs, ref_stride);
	}
	er = hipGetLastError();
	if (er != hipSuccess) { mcom_dfree(perm); return mcom_fail(ctx, MCOM_E_HIP, "group consensus: %s", hipGetErrorString(er)); }
	*d_perm_out = perm; *n_small = nsmall;
	return MCOM_OK;
}

// units of 32 columns (d_ujob: the job of every unit, d_uoff: first unit of every job); the 512-column tiles d_toff counts per job
// that hold a unit too deep for the counters come back as a list: d_tflag [n tiles, zeroed here], d_tlist [n tiles], *h_nlist
int mcom_merge_consensus_units(mcom_ctx *ctx, const uint64_t *d_packed, const uint64_t *d_members, const uint64_t *d_job_off, const uint64_t *d_ref_off,
                               const uint32_t *d_ujob, const uint32_t *d_uoff, uint32_t n_units, int L, uint8_t *d_refs,
                               const uint32_t *d_reg_lo, const uint32_t *d_reg_hi, const uint32_t *d_toff, uint32_t n_tiles,
                               unsigned int *d_tflag, uint32_t *d_tlist, uint32_t *h_nlist)
{
	*h_nlist = 0;
	if (n_units == 0) return MCOM_OK;
	unsigned int *tcount = d_tflag + n_tiles;                                // one more word behind the flags
	MCOM_HIP(ctx, hipMemsetAsync(d_tflag, 0, ((size_t)n_tiles + 1) * 4, ctx->stream));
	{
		McomProfScope ps_(ctx, PROF_CONSENSUS);
		MCOM_LAUNCH(k_merge_consensus_bs, dim3((n_units + 63) / 64), dim3(64), 0, ctx->stream, d_packed, mcom_words_per_read(L), d_members, d_job_off, d_ref_off,
		                   d_ujob, d_uoff, n_units, L, d_refs, d_reg_lo, d_reg_hi, d_toff, d_tflag, d_tlist, tcount,
		                   ctx->bs_cap && ctx->bs_cap < (1u << BS_KM) ? ctx->bs_cap : (1u << BS_KM) - 1u);
	}
	MCOM_LAUNCH_CHECK(ctx);
	MCOM_HIP(ctx, mcom_d2h_async(ctx, h_nlist, tcount, 4));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	return MCOM_OK;
}
