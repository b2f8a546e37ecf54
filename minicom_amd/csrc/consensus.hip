// minicom_amd/csrc/consensus.hip -- contig consensus on the device.
//
//   mcom_group_consensus : construct_ref (reference kthread_bucket.c:69-377) for every minimizer group of a
//                          Stage-1 round: column counts of the members laid out by their minimizer position,
//                          majority base per column, rejection of members with more than e mismatches, second
//                          consensus from the kept members.
//   mcom_merge_consensus : construct_ref2 (kthread_cb.c:105-218): majority base per column of a merged contig.
//   mcom_minimizer_prefix: the first m minimizers of every contig (what the builders push into the index,
//                          kthread_bucket.c:463, kthread_cb.c:370, :423) taken from the full sketch.
//
// Byte/integer work, LDS count tables, HBM-bound on the gathered packed rows (2 x 8W bytes per member).
#include "mcom_dev.hpp"

// base i (0..L-1) of a read as it lies on the contig: reverse complement when dir = 1 (preprocess.c:22-37)
__device__ __forceinline__ uint32_t obase(const uint64_t *row, int L, uint32_t dir, int i)
{
	const int j = dir ? L - 1 - i : i;
	const uint32_t b = (uint32_t)(row[j >> 5] >> (2 * (j & 31))) & 3u;
	return dir ? 3u - b : b;
}

// the same with the packed row spread over the wave (lane q holds word q): one global load per member instead of
// one per base
__device__ __forceinline__ uint32_t obase_w(uint64_t roww, int L, uint32_t dir, int i, int lane0 = 0)
{
	const int j = dir ? L - 1 - i : i;
	const uint64_t wv = __shfl(roww, lane0 + (j >> 5), 64);
	const uint32_t b = (uint32_t)(wv >> (2 * (j & 31))) & 3u;
	return dir ? 3u - b : b;
}

// ------------------------------------------------------------------------------------------------
// one 64-lane workgroup per group.  Columns: member q starts at off_q = pos0 - pos_q with pos the aligned
// minimizer position (cmpcluster's, kthread_bucket.c:51-56); 0 <= off_q <= L - k, so 2L columns suffice.
// ------------------------------------------------------------------------------------------------
#define GC_MAXCOL 512

// PK = true: two 16-bit counters per LDS word -- half the LDS, twice the groups in flight for a kernel that lives on
// hiding the latency of its dependent loads; groups of 65535 members or more are left to the PK = false launch.
#define GC_BIG 65535u
template <bool PK>
__global__ __launch_bounds__(64) void k_group_consensus(const uint64_t *__restrict__ packed, int W, uint64_t *__restrict__ members,
                                                        const uint32_t *__restrict__ goff, uint32_t ng, int L, int k_orig, int e,
                                                        uint8_t *__restrict__ keep, uint32_t *__restrict__ nkept,
                                                        uint16_t *__restrict__ svout, uint16_t *__restrict__ reflen,
                                                        uint8_t *__restrict__ refs, int ref_stride, unsigned int *__restrict__ big_seen,
                                                        const uint32_t *__restrict__ glist)
{
	extern __shared__ uint32_t gc_lds[];
	const int TL = 2 * L;
	const int TW = PK ? 2 * TL : 4 * TL;        // words of one table
	uint32_t *c1 = gc_lds;                      // counts of all members            [4][TL]
	uint32_t *c2 = gc_lds + TW;                 // counts of the kept members       [4][TL]
	uint8_t *rc = (uint8_t*)(gc_lds + 2 * TW);  // first consensus, 0xFF beyond its end
	if (blockIdx.x >= ng) return;
	const uint32_t g = glist ? glist[blockIdx.x] : blockIdx.x;         // glist: the ng groups to do
	const int lane = threadIdx.x;
	const uint32_t m0 = goff[g], m1 = goff[g + 1];
	if (PK ? (m1 - m0 >= GC_BIG) : (m1 - m0 < GC_BIG)) { if (PK && lane == 0) *big_seen = 1; return; }
	auto cadd = [&](uint32_t *t, int idx) { if (PK) atomicAdd(&t[idx >> 1], 1u << (16 * (idx & 1))); else atomicAdd(&t[idx], 1u); };
	auto cget = [&](const uint32_t *t, int idx) -> uint32_t { return PK ? (t[idx >> 1] >> (16 * (idx & 1))) & 0xFFFFu : t[idx]; };
	auto csub = [&](uint32_t *t, int idx) { if (PK) atomicSub(&t[idx >> 1], 1u << (16 * (idx & 1))); else atomicSub(&t[idx], 1u); };
	for (int c = lane; c < TW; c += 64) c1[c] = 0;
	__syncthreads();
	// pass 1: offsets, first counts
	// Members are taken eight at a time: lane 8i+w loads word w of member i's packed row, so the two dependent global
	// loads (member word, then its row) are paid once per eight members instead of once per member.
	int pos0 = 0;
	const int il = lane >> 3, wl = lane & 7;
	for (uint32_t q0 = m0; q0 < m1; q0 += 8) {
		const uint32_t ql = q0 + (uint32_t)il;
		const uint64_t yl = ql < m1 ? members[ql] : 0ull;
		const uint64_t rowl = (ql < m1 && wl < W) ? packed[(size_t)(yl >> 32) * W + wl] : 0ull;
		const int nm = (int)(m1 - q0 < 8u ? m1 - q0 : 8u);
		for (int i = 0; i < nm; ++i) {
			const uint64_t y = __shfl(yl, 8 * i, 64);
			const uint32_t dir = (uint32_t)(y & 1);
			int pos = (int)((uint32_t)y >> 1);
			if (dir) pos = L - pos + k_orig - 2;
			if (q0 == m0 && i == 0) pos0 = pos;
			const int off = pos0 - pos;
			for (int s0 = 0; s0 < L; s0 += 64) {                            // whole wave in the shuffle, also past the read's end
				const int s = s0 + lane;
				const uint32_t b = obase_w(rowl, L, dir, s < L ? s : 0, 8 * i);
				if (s < L) cadd(c1, b * TL + off + s);
			}
		}
	}
	__syncthreads();
	// The counts of the KEPT members (second consensus) start as a copy of all members' counts; pass 2 takes the rejected
	// members out again.  Nearly every member is kept, so this replaces ~L LDS atomics per member by ~L per rejected member.
	for (int c = lane; c < TW; c += 64) c2[c] = c1[c];
	// first consensus: majority base per column, ties to the smaller code (strict '>'), ends at the first empty column
	for (int c = lane; c < TL; c += 64) {
		uint32_t mx = cget(c1, c); uint8_t b = 0;
		for (int q = 1; q < 4; ++q) { const uint32_t v = cget(c1, q * TL + c); if (v > mx) { mx = v; b = (uint8_t)q; } }
		rc[c] = mx ? b : (uint8_t)0xFF;
	}
	__syncthreads();
	int ref_len = TL;
	for (int c0 = 0; c0 < TL; c0 += 64) {
		const int c = c0 + lane;
		const uint64_t z = __ballot(c < TL && rc[c] == 0xFF);
		if (z) { ref_len = c0 + __ffsll((unsigned long long)z) - 1; break; }
	}
	// pass 2: mismatches against the first consensus, kept members counted again
	uint32_t nk = 0; int rend = 0;
	for (uint32_t q0 = m0; q0 < m1; q0 += 8) {
		const uint32_t ql = q0 + (uint32_t)il;
		const uint64_t yl = ql < m1 ? members[ql] : 0ull;                  // still the sketch records: rewritten below
		const uint64_t rowl = (ql < m1 && wl < W) ? packed[(size_t)(yl >> 32) * W + wl] : 0ull;
		const int nm = (int)(m1 - q0 < 8u ? m1 - q0 : 8u);
		for (int i = 0; i < nm; ++i) {
		const uint32_t q = q0 + (uint32_t)i;
		const uint64_t y = __shfl(yl, 8 * i, 64);
		const uint32_t dir = (uint32_t)(y & 1);
		int pos = (int)((uint32_t)y >> 1);
		if (dir) pos = L - pos + k_orig - 2;
		const int off = pos0 - pos;
		int dif = 0;
		uint32_t bs[4];                                                     // this lane's bases of the member (L <= 256)
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const int s = u * 64 + lane;
			bs[u] = 0;
			if (u * 64 < L) {
				bs[u] = obase_w(rowl, L, dir, s < L ? s : 0, 8 * i);
				const bool mis = s < L && ((off + s >= ref_len) || rc[off + s] != (uint8_t)bs[u]);
				dif += __popcll(__ballot(mis));
			}
		}
		const bool kp = dif <= e;                                          // kthread_bucket.c:189
		if (kp) {
			++nk;
			if (off + L > rend) rend = off + L;
		} else {
#pragma unroll
			for (int u = 0; u < 4; ++u) { const int s = u * 64 + lane; if (s < L) csub(c2, bs[u] * TL + off + s); }
		}
		if (lane == 0) { keep[q] = kp ? 1 : 0; members[q] = (y >> 32 << 32) | ((uint64_t)off << 1) | dir; }   // :101
		}
	}
	__syncthreads();
	// second consensus over [sv, rend): sv = first column (inside the first consensus) any kept member covers
	int sv = 0;
	if (nk) {
		sv = ref_len;
		for (int c0 = 0; c0 < ref_len; c0 += 64) {
			const int c = c0 + lane;
			const bool any = c < ref_len && (cget(c2, c) | cget(c2, TL + c) | cget(c2, 2 * TL + c) | cget(c2, 3 * TL + c)) != 0;
			const uint64_t z = __ballot(any);
			if (z) { sv = c0 + __ffsll((unsigned long long)z) - 1; break; }
		}
		uint8_t *out = refs + (size_t)g * ref_stride;
		for (int c = sv + lane; c < rend; c += 64) {
			uint32_t mx = cget(c2, c); int b = 0;
			for (int q = 1; q < 4; ++q) { const uint32_t v = cget(c2, q * TL + c); if (v > mx) { mx = v; b = q; } }
			out[c - sv] = (uint8_t)"ACGT"[b];
		}
	}
	if (lane == 0) { nkept[g] = nk; svout[g] = (uint16_t)sv; reflen[g] = (uint16_t)(nk ? rend - sv : 0); }
}

// ------------------------------------------------------------------------------------------------
// The same with the column counts in REGISTERS (round 2).  Lane c owns columns c, c + 64, c + 128, ...: four 16-bit counters
// (one per base) in one 64-bit register per column.  A member is walked by column block instead of by read position, so a lane
// always updates its own columns: no LDS table, no LDS atomics, nothing to clear or read back -- the first consensus, the
// mismatch test of pass 2 and the second consensus all work on the lane's registers.  Groups of 65 535 members or more keep
// the LDS kernel with 32-bit counters (k_group_consensus<false>).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t best_base(uint64_t c, uint32_t &mx)
{
	mx = (uint32_t)c & 0xFFFFu; uint32_t b = 0;
#pragma unroll
	for (uint32_t q = 1; q < 4; ++q) { const uint32_t v = (uint32_t)(c >> (16 * q)) & 0xFFFFu; if (v > mx) { mx = v; b = q; } }   // ties to the smaller code (strict '>')
	return b;
}
template <int NU>
__global__ __launch_bounds__(64) void k_group_consensus_reg(const uint64_t *__restrict__ packed, int W, uint64_t *__restrict__ members,
                                                            const uint32_t *__restrict__ goff, uint32_t ng, int L, int k_orig, int e,
                                                            uint8_t *__restrict__ keep, uint32_t *__restrict__ nkept,
                                                            uint16_t *__restrict__ svout, uint16_t *__restrict__ reflen,
                                                            uint8_t *__restrict__ refs, int ref_stride, unsigned int *__restrict__ big_seen,
                                                            const uint32_t *__restrict__ glist)
{
	if (blockIdx.x >= ng) return;
	const uint32_t g = glist ? glist[blockIdx.x] : blockIdx.x;         // glist: the ng groups to do
	const int lane = threadIdx.x;
	const int TL = 2 * L;
	const uint32_t m0 = goff[g], m1 = goff[g + 1];
	if (m1 - m0 >= GC_BIG) { if (lane == 0) *big_seen = 1; return; }
	// Members are taken eight at a time: lane 8i+w holds word w of member i's packed row.  The first GR_HELD batches (32 members:
	// nearly every group) are loaded up front -- all member words together, then all rows together: two dependent round trips
	// for the whole group instead of two per batch -- and stay in registers for pass 2.
	constexpr int GR_HELD = 4;
	const int il = lane >> 3, wl = lane & 7;
	uint64_t ys[GR_HELD], rows[GR_HELD];
#pragma unroll
	for (int h = 0; h < GR_HELD; ++h) { const uint32_t ql = m0 + 8u * h + (uint32_t)il; ys[h] = ql < m1 ? members[ql] : 0ull; }
#pragma unroll
	for (int h = 0; h < GR_HELD; ++h) { const uint32_t ql = m0 + 8u * h + (uint32_t)il; rows[h] = (ql < m1 && wl < W) ? packed[(size_t)(ys[h] >> 32) * W + wl] : 0ull; }
	uint64_t c1[NU];
#pragma unroll
	for (int u = 0; u < NU; ++u) c1[u] = 0;
	int pos0 = 0;
	// pass 1: offsets, counts of all members
	auto count_batch = [&](uint32_t q0, uint64_t yl, uint64_t rowl) {
		const int nm = (int)(m1 - q0 < 8u ? m1 - q0 : 8u);
		for (int i = 0; i < nm; ++i) {
			const uint64_t y = __shfl(yl, 8 * i, 64);
			const uint32_t dir = (uint32_t)(y & 1);
			int pos = (int)((uint32_t)y >> 1);
			if (dir) pos = L - pos + k_orig - 2;
			if (q0 == m0 && i == 0) pos0 = pos;
			const int off = pos0 - pos;
#pragma unroll
			for (int u = 0; u < NU; ++u) {
				if (64 * u + 63 < off || 64 * u >= off + L) continue;           // uniform: the read does not reach this block of columns
				const int sidx = 64 * u + lane - off;
				const bool valid = (unsigned)sidx < (unsigned)L;
				const uint32_t b = obase_w(rowl, L, dir, valid ? sidx : 0, 8 * i);
				if (valid) c1[u] += 1ull << (16 * b);
			}
		}
	};
#pragma unroll
	for (int h = 0; h < GR_HELD; ++h) if (m0 + 8u * h < m1) count_batch(m0 + 8u * h, ys[h], rows[h]);
	for (uint32_t q0 = m0 + 8u * GR_HELD; q0 < m1; q0 += 8) {
		const uint32_t ql = q0 + (uint32_t)il;
		const uint64_t yl = ql < m1 ? members[ql] : 0ull;
		const uint64_t rowl = (ql < m1 && wl < W) ? packed[(size_t)(yl >> 32) * W + wl] : 0ull;
		count_batch(q0, yl, rowl);
	}
	// first consensus: majority base per column, ends at the first empty column
	uint8_t rcu[NU];
	int ref_len = TL;
#pragma unroll
	for (int u = 0; u < NU; ++u) { uint32_t mx; const uint32_t b = best_base(c1[u], mx); rcu[u] = mx ? (uint8_t)b : (uint8_t)0xFF; }
#pragma unroll
	for (int u = NU - 1; u >= 0; --u) {
		const uint64_t z = __ballot(64 * u + lane < TL && rcu[u] == 0xFF);
		if (z) ref_len = 64 * u + __ffsll((unsigned long long)z) - 1;
	}
	// pass 2: mismatches against the first consensus; the kept members' counts = all counts minus the rejected members'
	uint64_t c2[NU];
#pragma unroll
	for (int u = 0; u < NU; ++u) c2[u] = c1[u];
	uint32_t nk = 0; int rend = 0;
	auto judge_batch = [&](uint32_t q0, uint64_t yl, uint64_t rowl) {
		const int nm = (int)(m1 - q0 < 8u ? m1 - q0 : 8u);
		for (int i = 0; i < nm; ++i) {
			const uint32_t q = q0 + (uint32_t)i;
			const uint64_t y = __shfl(yl, 8 * i, 64);
			const uint32_t dir = (uint32_t)(y & 1);
			int pos = (int)((uint32_t)y >> 1);
			if (dir) pos = L - pos + k_orig - 2;
			const int off = pos0 - pos;
			int dif = 0;
			uint64_t sub[NU];
#pragma unroll
			for (int u = 0; u < NU; ++u) {
				sub[u] = 0;
				if (64 * u + 63 < off || 64 * u >= off + L) continue;
				const int col = 64 * u + lane, sidx = col - off;
				const bool valid = (unsigned)sidx < (unsigned)L;
				const uint32_t b = obase_w(rowl, L, dir, valid ? sidx : 0, 8 * i);
				const bool mis = valid && (col >= ref_len || rcu[u] != (uint8_t)b);
				dif += __popcll(__ballot(mis));
				if (valid) sub[u] = 1ull << (16 * b);
			}
			const bool kp = dif <= e;                                          // kthread_bucket.c:189
			if (kp) { ++nk; if (off + L > rend) rend = off + L; }
			else {
#pragma unroll
				for (int u = 0; u < NU; ++u) c2[u] -= sub[u];
			}
			if (lane == 0) { keep[q] = kp ? 1 : 0; members[q] = (y >> 32 << 32) | ((uint64_t)off << 1) | dir; }   // :101
		}
	};
#pragma unroll
	for (int h = 0; h < GR_HELD; ++h) if (m0 + 8u * h < m1) judge_batch(m0 + 8u * h, ys[h], rows[h]);
	for (uint32_t q0 = m0 + 8u * GR_HELD; q0 < m1; q0 += 8) {
		const uint32_t ql = q0 + (uint32_t)il;
		const uint64_t yl = ql < m1 ? members[ql] : 0ull;                  // still the sketch records: rewritten by this pass
		const uint64_t rowl = (ql < m1 && wl < W) ? packed[(size_t)(yl >> 32) * W + wl] : 0ull;
		judge_batch(q0, yl, rowl);
	}
	// second consensus over [sv, rend): sv = first column (inside the first consensus) any kept member covers
	int sv = 0;
	if (nk) {
		sv = ref_len;
#pragma unroll
		for (int u = NU - 1; u >= 0; --u) {
			const uint64_t z = __ballot(64 * u + lane < ref_len && c2[u] != 0);
			if (z) sv = 64 * u + __ffsll((unsigned long long)z) - 1;
		}
		uint8_t *out = refs + (size_t)g * ref_stride;
#pragma unroll
		for (int u = 0; u < NU; ++u) {
			const int c = 64 * u + lane;
			if (c >= sv && c < rend) { uint32_t mx; out[c - sv] = (uint8_t)"ACGT"[best_base(c2[u], mx)]; }
		}
	}
	if (lane == 0) { nkept[g] = nk; svout[g] = (uint16_t)sv; reflen[g] = (uint16_t)(nk ? rend - sv : 0); }
}

// the groups below 32 members, bit-sliced (consensus_bs.hip)
int mcom_group_consensus_small(mcom_ctx *ctx, const uint64_t *d_packed, uint64_t *d_members, const uint32_t *d_group_off, uint32_t n_groups,
                               int L, int k_orig, int e, uint8_t *d_keep, uint32_t *d_nkept, uint16_t *d_sv, uint16_t *d_reflen, uint8_t *d_refs,
                               int ref_stride, uint32_t **d_perm_out, uint32_t *n_small);

extern "C" int mcom_group_consensus(mcom_ctx *ctx, const uint64_t *d_packed, uint64_t *d_members, const uint32_t *d_group_off,
                                    uint32_t n_groups, int L, int k_orig, int e, uint8_t *d_keep, uint32_t *d_nkept,
                                    uint16_t *d_sv, uint16_t *d_reflen, uint8_t *d_refs, int ref_stride)
{
	if (!ctx) return MCOM_E_ARG;
	if (n_groups == 0) return MCOM_OK;
	if (L < 1 || L > 256 || k_orig < 1 || k_orig > 31 || ref_stride < 2 * L) return mcom_fail(ctx, MCOM_E_ARG, "bad consensus arguments");
	if (!d_packed || !d_members || !d_group_off || !d_keep || !d_nkept || !d_sv || !d_reflen || !d_refs) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	int rc = mcom_ws_reserve(ctx, 256);
	if (rc) return rc;
	// nearly every group holds a handful of members: those go through the bit-sliced kernel, several groups per wave; what is left
	// (32 members or more) takes a wave per group
	uint32_t *perm = nullptr, n_small = 0;
	if (ref_stride % 4 == 0 && ((uintptr_t)d_refs & 3) == 0) {
		if ((rc = mcom_group_consensus_small(ctx, d_packed, d_members, d_group_off, n_groups, L, k_orig, e, d_keep, d_nkept, d_sv, d_reflen, d_refs, ref_stride, &perm, &n_small)))
			return rc;
	}
	const uint32_t n_rest = n_groups - n_small;
	const uint32_t *glist = perm ? perm + n_small : nullptr;
	if (n_rest == 0) { MCOM_HIP(ctx, mcom_stream_sync(ctx)); mcom_dfree(perm); return MCOM_OK; }
	unsigned int *big = (unsigned int*)mcom_zeroed(ctx, ctx->ws, 4);
	hipError_t er = big ? hipSuccess : hipErrorUnknown;
	if (er == hipSuccess) {
		McomProfScope ps_(ctx, PROF_CONSENSUS);
#define MCOM_GC(NU) case NU: MCOM_LAUNCH((k_group_consensus_reg<NU>), dim3(n_rest), dim3(64), 0, ctx->stream, d_packed, mcom_words_per_read(L), d_members, \
	                   d_group_off, n_rest, L, k_orig, e, d_keep, d_nkept, d_sv, d_reflen, d_refs, ref_stride, big, glist); break;
		switch ((2 * L + 63) / 64) { MCOM_GC(1) MCOM_GC(2) MCOM_GC(3) MCOM_GC(4) MCOM_GC(5) MCOM_GC(6) MCOM_GC(7) MCOM_GC(8) default: er = hipErrorInvalidValue; }
#undef MCOM_GC
	}
	if (er == hipSuccess) er = hipGetLastError();
	unsigned int hb = 0;
	if (er == hipSuccess) er = mcom_d2h_async(ctx, &hb, big, 4);
	if (er == hipSuccess) er = mcom_stream_sync(ctx);
	if (er == hipSuccess && hb) {                                            // groups of 65535 members or more: 32-bit counters
		McomProfScope ps_(ctx, PROF_CONSENSUS);
		MCOM_LAUNCH((k_group_consensus<false>), dim3(n_rest), dim3(64), (size_t)(8 * 2 * L * 4 + 2 * L + 16), ctx->stream, d_packed, mcom_words_per_read(L), d_members,
		                   d_group_off, n_rest, L, k_orig, e, d_keep, d_nkept, d_sv, d_reflen, d_refs, ref_stride, big, glist);
		er = hipGetLastError();
		if (er == hipSuccess) er = mcom_stream_sync(ctx);
	}
	mcom_dfree(perm);
	if (er != hipSuccess) return mcom_fail(ctx, MCOM_E_HIP, "group consensus: %s", hipGetErrorString(er));
	return MCOM_OK;
}

// ------------------------------------------------------------------------------------------------
// construct_ref2: members of job j are members[joff[j] .. joff[j+1]), sorted by offset (cmpcluster2); the
// consensus of job j goes to refs[roff[j] .. roff[j+1]).  One 64-lane workgroup per (job, tile of MC_TILE columns).
// ------------------------------------------------------------------------------------------------
#define MC_TILE 512

template <bool PK>
__global__ __launch_bounds__(64) void k_merge_consensus(const uint64_t *__restrict__ packed, int W, const uint64_t *__restrict__ members,
                                                        const uint64_t *__restrict__ joff, const uint64_t *__restrict__ roff,
                                                        const uint32_t *__restrict__ tile_job, const uint32_t *__restrict__ tile_idx,
                                                        uint32_t n_tiles, int L, uint8_t *__restrict__ refs,
                                                        const uint32_t *__restrict__ reg_lo, const uint32_t *__restrict__ reg_hi,
                                                        unsigned int *__restrict__ big_seen, const uint32_t *__restrict__ tlist)
{
	__shared__ uint32_t cc[PK ? 2 * MC_TILE : 4 * MC_TILE];                  // PK: two 16-bit counters per word (see k_group_consensus)
	if (blockIdx.x >= n_tiles) return;
	const uint32_t t = tlist ? tlist[blockIdx.x] : blockIdx.x;                // tlist: the n_tiles tiles to do
	const int lane = threadIdx.x;
	const uint32_t j = tile_job[t];
	const uint64_t m0 = joff[j], m1 = joff[j + 1];
	if (PK ? (m1 - m0 >= GC_BIG) : (m1 - m0 < GC_BIG)) { if (PK && lane == 0) *big_seen = 1; return; }
	auto cadd = [&](int idx) { if (PK) atomicAdd(&cc[idx >> 1], 1u << (16 * (idx & 1))); else atomicAdd(&cc[idx], 1u); };
	auto cget = [&](int idx) -> uint32_t { return PK ? (cc[idx >> 1] >> (16 * (idx & 1))) & 0xFFFFu : cc[idx]; };
	// the columns to count: the whole contig, or only the region given for the job (the overlap of the two parents)
	const long len = reg_hi ? (long)reg_hi[j] : (long)(roff[j + 1] - roff[j]);
	const long lo = (reg_lo ? (long)reg_lo[j] : 0) + (long)tile_idx[t] * MC_TILE, hi = lo + MC_TILE < len ? lo + MC_TILE : len;
	for (int c = lane; c < (PK ? 2 : 4) * MC_TILE; c += 64) cc[c] = 0;
	__syncthreads();
	// first member whose read can reach column lo: offset > lo - L (members are sorted by offset)
	uint64_t a = m0, b = m1;
	while (a < b) { const uint64_t mid = (a + b) >> 1; if ((long)((uint32_t)members[mid] >> 1) + L <= lo) a = mid + 1; else b = mid; }
	const int il = lane >> 3, wl = lane & 7;                                 // eight members per round of global loads
	bool past = false;
	for (uint64_t q0 = a; q0 < m1 && !past; q0 += 8) {
		const uint64_t ql = q0 + (uint64_t)il;
		const uint64_t yl = ql < m1 ? members[ql] : 0ull;
		const bool use = ql < m1 && (long)((uint32_t)yl >> 1) < hi;
		const uint64_t rowl = (use && wl < W) ? packed[(size_t)(yl >> 32) * W + wl] : 0ull;
		const int nm = (int)(m1 - q0 < 8ull ? m1 - q0 : 8ull);
		for (int i = 0; i < nm; ++i) {
			const uint64_t y = __shfl(yl, 8 * i, 64);
			const long off = (long)((uint32_t)y >> 1);
			if (off >= hi) { past = true; break; }
			const uint32_t dir = (uint32_t)(y & 1);
			for (int s0 = 0; s0 < L; s0 += 64) {
				const int s = s0 + lane;
				const long c = off + s;
				const uint32_t b = obase_w(rowl, L, dir, s < L ? s : 0, 8 * i);
				if (s < L && c >= lo && c < hi) cadd((int)b * MC_TILE + (int)(c - lo));
			}
		}
	}
	__syncthreads();
	uint8_t *out = refs + roff[j];
	for (long c = lo + lane; c < hi; c += 64) {
		const int i = (int)(c - lo);
		uint32_t mx = cget(i); int bb = 0;
		for (int q = 1; q < 4; ++q) { const uint32_t v = cget(q * MC_TILE + i); if (v > mx) { mx = v; bb = q; } }
		out[c] = (uint8_t)"ACGT"[bb];
	}
}

// The same with the column counts in registers (see k_group_consensus_reg): lane c owns columns lo + c + 64 u of the tile.
__global__ __launch_bounds__(64) void k_merge_consensus_reg(const uint64_t *__restrict__ packed, int W, const uint64_t *__restrict__ members,
                                                            const uint64_t *__restrict__ joff, const uint64_t *__restrict__ roff,
                                                            const uint32_t *__restrict__ tile_job, const uint32_t *__restrict__ tile_idx,
                                                            uint32_t n_tiles, int L, uint8_t *__restrict__ refs,
                                                            const uint32_t *__restrict__ reg_lo, const uint32_t *__restrict__ reg_hi,
                                                            unsigned int *__restrict__ big_seen, const uint32_t *__restrict__ tlist)
{
	constexpr int NU = MC_TILE / 64;
	if (blockIdx.x >= n_tiles) return;
	const uint32_t t = tlist ? tlist[blockIdx.x] : blockIdx.x;                // tlist: the n_tiles tiles to do
	const int lane = threadIdx.x;
	const uint32_t j = tile_job[t];
	const uint64_t m0 = joff[j], m1 = joff[j + 1];
	if (m1 - m0 >= GC_BIG) { if (lane == 0) *big_seen = 1; return; }
	const long len = reg_hi ? (long)reg_hi[j] : (long)(roff[j + 1] - roff[j]);
	const long lo = (reg_lo ? (long)reg_lo[j] : 0) + (long)tile_idx[t] * MC_TILE, hi = lo + MC_TILE < len ? lo + MC_TILE : len;
	uint64_t cc[NU];
#pragma unroll
	for (int u = 0; u < NU; ++u) cc[u] = 0;
	// first member whose read can reach column lo: offset > lo - L (members are sorted by offset)
	uint64_t a = m0, b = m1;
	while (a < b) { const uint64_t mid = (a + b) >> 1; if ((long)((uint32_t)members[mid] >> 1) + L <= lo) a = mid + 1; else b = mid; }
	const int il = lane >> 3, wl = lane & 7;                                 // eight members per round of global loads
	// two batches ahead for the member words, one ahead for the rows: the loads of the next batch travel while this one is counted
	auto load_y = [&](uint64_t q0) -> uint64_t { const uint64_t ql = q0 + (uint64_t)il; return ql < m1 ? members[ql] : 0ull; };
	auto load_row = [&](uint64_t q0, uint64_t yl) -> uint64_t {
		const uint64_t ql = q0 + (uint64_t)il;
		return (ql < m1 && (long)((uint32_t)yl >> 1) < hi && wl < W) ? packed[(size_t)(yl >> 32) * W + wl] : 0ull;
	};
	uint64_t y_cur = load_y(a), y_nxt = load_y(a + 8);
	uint64_t row_cur = load_row(a, y_cur);
	bool past = false;
	for (uint64_t q0 = a; q0 < m1 && !past; q0 += 8) {
		const uint64_t row_nxt = load_row(q0 + 8, y_nxt);
		const uint64_t y_nn = load_y(q0 + 16);
		const uint64_t yl = y_cur, rowl = row_cur;
		const int nm = (int)(m1 - q0 < 8ull ? m1 - q0 : 8ull);
		for (int i = 0; i < nm; ++i) {
			const uint64_t y = __shfl(yl, 8 * i, 64);
			const long off = (long)((uint32_t)y >> 1);
			if (off >= hi) { past = true; break; }
			const uint32_t dir = (uint32_t)(y & 1);
			const long rel = off - lo;                                         // the read covers tile columns [rel, rel + L)
#pragma unroll
			for (int u = 0; u < NU; ++u) {
				if (64 * u + 63 < rel || 64 * u >= rel + L) continue;            // uniform
				const long c = 64 * u + lane;
				const long sidx = c - rel;
				const bool valid = sidx >= 0 && sidx < L && lo + c < hi;
				const uint32_t bb = obase_w(rowl, L, dir, valid ? (int)sidx : 0, 8 * i);
				if (valid) cc[u] += 1ull << (16 * bb);
			}
		}
		y_cur = y_nxt; row_cur = row_nxt; y_nxt = y_nn;
	}
	uint8_t *out = refs + roff[j];
#pragma unroll
	for (int u = 0; u < NU; ++u) {
		const long c = lo + 64 * u + lane;
		if (c < hi) { uint32_t mx; out[c] = (uint8_t)"ACGT"[best_base(cc[u], mx)]; }
	}
}

extern "C" int mcom_merge_consensus(mcom_ctx *ctx, const uint64_t *d_packed, const uint64_t *d_members, const uint64_t *d_job_off,
                                    const uint64_t *d_ref_off, const uint32_t *d_tile_job, const uint32_t *d_tile_idx,
                                    uint32_t n_tiles, int L, uint8_t *d_refs)
{
	if (!ctx) return MCOM_E_ARG;
	if (n_tiles == 0) return MCOM_OK;
	if (L < 1 || L > 256) return mcom_fail(ctx, MCOM_E_ARG, "bad read length");
	if (!d_packed || !d_members || !d_job_off || !d_ref_off || !d_tile_job || !d_tile_idx || !d_refs) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	return mcom_merge_consensus_regions(ctx, d_packed, d_members, d_job_off, d_ref_off, d_tile_job, d_tile_idx, n_tiles, L, d_refs, nullptr, nullptr, nullptr);
}

// the same over one column range per job (merge.hip: only the overlap of the two parents is counted again)
int mcom_merge_consensus_regions(mcom_ctx *ctx, const uint64_t *d_packed, const uint64_t *d_members, const uint64_t *d_job_off, const uint64_t *d_ref_off,
                                 const uint32_t *d_tile_job, const uint32_t *d_tile_idx, uint32_t n_tiles, int L, uint8_t *d_refs,
                                 const uint32_t *d_reg_lo, const uint32_t *d_reg_hi, const uint32_t *d_tlist)
{
	if (n_tiles == 0) return MCOM_OK;
	// the flag lives behind everything the callers keep in the workspace (they reserve their own part first)
	unsigned int *big0 = nullptr;
	MCOM_HIP(ctx, mcom_dmalloc((void**)&big0, 256));
	unsigned int *big = (unsigned int*)mcom_zeroed(ctx, big0, 4);
	if (!big) { mcom_dfree(big0); return mcom_fail(ctx, MCOM_E_HIP, "clear"); }
	{ McomProfScope ps_(ctx, PROF_CONSENSUS);
	MCOM_LAUNCH(k_merge_consensus_reg, dim3(n_tiles), dim3(64), 0, ctx->stream, d_packed, mcom_words_per_read(L), d_members, d_job_off, d_ref_off,
	                   d_tile_job, d_tile_idx, n_tiles, L, d_refs, d_reg_lo, d_reg_hi, big, d_tlist); }
	unsigned int hb = 0;
	hipError_t e1 = mcom_d2h_async(ctx, &hb, big, 4);
	if (e1 == hipSuccess) e1 = mcom_stream_sync(ctx);
	if (e1 == hipSuccess && hb) {                                            // a job of 65535 members or more: 32-bit counters
		McomProfScope ps_(ctx, PROF_CONSENSUS);
		MCOM_LAUNCH((k_merge_consensus<false>), dim3(n_tiles), dim3(64), 0, ctx->stream, d_packed, mcom_words_per_read(L), d_members, d_job_off, d_ref_off,
		                   d_tile_job, d_tile_idx, n_tiles, L, d_refs, d_reg_lo, d_reg_hi, big, d_tlist);
		e1 = mcom_stream_sync(ctx);
	}
	mcom_dfree(big0);
	if (e1 != hipSuccess) return mcom_fail(ctx, MCOM_E_HIP, "merge consensus: %s", hipGetErrorString(e1));
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

// ------------------------------------------------------------------------------------------------
// first m minimizers of every contig out of the full sketch: out_moff[c] = c' start, fixed stride not needed
// ------------------------------------------------------------------------------------------------
// (ord: the contigs in the order they are taken in -- contig ord[c] of the set is the c-th of the output; NULL = the set's own order)
__global__ void k_prefix_counts(const uint32_t *__restrict__ moff, const uint32_t *__restrict__ ord, size_t n, uint32_t m, uint32_t *__restrict__ cnt)
{
	const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (c < n) { const uint32_t i = ord ? ord[c] : (uint32_t)c; const uint32_t k = moff[i + 1] - moff[i]; cnt[c] = k < m ? k : m; }
	if (c == n) cnt[c] = 0;
}
__global__ void k_prefix_copy(const uint32_t *__restrict__ moff, const mcom_mm128 *__restrict__ rec, const uint32_t *__restrict__ ord, size_t n, uint32_t m,
                              const uint32_t *__restrict__ ooff, mcom_mm128 *__restrict__ out)
{
	const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t c = t / m; const uint32_t q = (uint32_t)(t - c * m);
	if (c >= n) return;
	if (q < ooff[c + 1] - ooff[c]) out[ooff[c] + q] = rec[moff[ord ? ord[c] : (uint32_t)c] + q];
}

extern "C" int mcom_minimizer_prefix(mcom_ctx *ctx, const uint32_t *d_moff, const mcom_mm128 *d_rec, size_t n, uint32_t m,
                                     uint32_t *d_out_moff, mcom_mm128 *d_out, uint64_t *h_total)
{
	return mcom_minimizer_prefix_ord(ctx, d_moff, d_rec, nullptr, n, m, d_out_moff, d_out, h_total);
}
extern "C" int mcom_minimizer_prefix_ord(mcom_ctx *ctx, const uint32_t *d_moff, const mcom_mm128 *d_rec, const uint32_t *d_ord, size_t n, uint32_t m,
                                         uint32_t *d_out_moff, mcom_mm128 *d_out, uint64_t *h_total)
{
	if (!ctx) return MCOM_E_ARG;
	if (h_total) *h_total = 0;
	if (m == 0) return mcom_fail(ctx, MCOM_E_ARG, "m must be positive");
	if (!d_out_moff) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	if (n == 0) { MCOM_HIP(ctx, hipMemsetAsync(d_out_moff, 0, 4, ctx->stream)); return MCOM_OK; }
	if (!d_moff || !d_rec || !d_out) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	const size_t scr_b = (mcom_scan_scratch_elems(n + 1) * 4 + 1024 + 255) & ~(size_t)255;
	int rc = mcom_ws_reserve(ctx, scr_b);
	if (rc) return rc;
	MCOM_LAUNCH(k_prefix_counts, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, ctx->stream, d_moff, d_ord, n, m, d_out_moff);
	MCOM_LAUNCH_CHECK(ctx);
	rc = mcom_scan_u32(ctx, d_out_moff, d_out_moff, n + 1, (uint32_t*)ctx->ws);
	if (rc) return rc;
	const size_t tot = n * (size_t)m;
	MCOM_LAUNCH(k_prefix_copy, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, d_moff, d_rec, d_ord, n, m, d_out_moff, d_out);
	MCOM_LAUNCH_CHECK(ctx);
	if (h_total) {
		uint32_t total = 0;
		MCOM_HIP(ctx, mcom_d2h_async(ctx, &total, d_out_moff + n, 4));
		MCOM_HIP(ctx, mcom_stream_sync(ctx));
		*h_total = total;
	}
	return MCOM_OK;
}
