// minicom_amd/csrc/groups.hip -- what process_bucket does with the outcome of construct_ref (kthread_bucket.c:446-505),
// on the device: groups that keep more than one member become contigs of the device-resident contig set (members
// re-based to the first covered column, :349), every other member of a group is a reject that goes back to the caller
// in the reference's visiting order.  Nothing but the singles and the (few) rejects has to cross PCIe.
#include "mcom_dev.hpp"

static inline size_t g_al256(size_t b) { return (b + 255) & ~(size_t)255; }

// per group: contig slot, members, chars, rejects
__global__ void k_group_sizes(const uint32_t *__restrict__ goff, const uint32_t *__restrict__ nkept, const uint16_t *__restrict__ reflen, size_t ng,
                              uint32_t *__restrict__ slot, uint32_t *__restrict__ msz, uint64_t *__restrict__ rsz, uint32_t *__restrict__ rej)
{
	const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (g > ng) return;
	if (g == ng) { slot[g] = 0; msz[g] = 0; rsz[g] = 0; rej[g] = 0; return; }
	const uint32_t sz = goff[g + 1] - goff[g], nk = nkept[g];
	const bool acc = nk > 1;                                                  // :451
	slot[g] = acc ? 1u : 0u; msz[g] = acc ? nk : 0u; rsz[g] = acc ? (uint64_t)reflen[g] : 0ull;
	rej[g] = (nk == sz && nk > 1) ? 0u : (sz - nk) + (nk == 1 ? 1u : 0u);     // :194-213, :477-498
}

__global__ __launch_bounds__(256) void k_group_emit(const uint64_t *__restrict__ members, const uint32_t *__restrict__ goff, size_t ng,
                                                    const uint8_t *__restrict__ keep, const uint32_t *__restrict__ nkept, const uint16_t *__restrict__ sv,
                                                    const uint16_t *__restrict__ reflen, const uint8_t *__restrict__ refs, int ref_stride,
                                                    const uint32_t *__restrict__ slot, const uint32_t *__restrict__ mof, const uint64_t *__restrict__ rof,
                                                    const uint32_t *__restrict__ rjo, uint64_t n_have, uint64_t chars_have, uint64_t members_have,
                                                    uint8_t *__restrict__ seq, uint64_t *__restrict__ soff, uint64_t *__restrict__ mem, uint64_t *__restrict__ moff,
                                                    uint32_t *__restrict__ rej_rid, uint32_t *__restrict__ rej_group)
{
	// sixteen lanes per group, four groups per wave (a group is six members and two hundred characters: with a wave per group the
	// kernel waited for its dependent loads, PMC 89 % dependency wait); the ballots are cut to the group's sixteen lanes
	const size_t g = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
	if (g >= ng) return;
	const int lane = threadIdx.x & 15, seg = (threadIdx.x & 63) >> 4;
	const uint32_t m0 = goff[g], m1 = goff[g + 1], nk = nkept[g], sz = m1 - m0;
	const uint32_t below = (1u << lane) - 1u;
	auto ballot16 = [&](bool p) -> uint32_t { return (uint32_t)(__ballot(p) >> (16 * seg)) & 0xFFFFu; };
	if (nk > 1) {
		const uint64_t c = n_have + slot[g];
		uint64_t *dst = mem + members_have + mof[g];
		const uint64_t sv2 = (uint64_t)sv[g] << 1;
		uint32_t done = 0;
		for (uint32_t q0 = m0; q0 < m1; q0 += 16) {
			const uint32_t q = q0 + lane;
			const bool kp = q < m1 && keep[q];
			const uint32_t km = ballot16(kp);
			if (kp) dst[done + (uint32_t)__popc(km & below)] = members[q] - sv2;
			done += (uint32_t)__popc(km);
		}
		const uint32_t len = reflen[g];
		const uint8_t *src = refs + g * (size_t)ref_stride;
		uint8_t *out = seq + chars_have + rof[g];
		{                                                                    // eight characters per lane and step, then the last few
			const uint32_t n8 = len >> 3;
			for (uint32_t i = lane; i < n8; i += 16) { uint64_t v; __builtin_memcpy(&v, src + 8 * i, 8); __builtin_memcpy(out + 8 * i, &v, 8); }
			for (uint32_t i = (n8 << 3) + lane; i < len; i += 16) out[i] = src[i];
		}
		if (lane == 0) { moff[c + 1] = members_have + mof[g] + nk; soff[c + 1] = chars_have + rof[g] + len; }
	}
	if (!(nk == sz && nk > 1)) {
		uint32_t at = rjo[g];
		for (uint32_t q0 = m0; q0 < m1; q0 += 16) {                            // the rejected ones first (:194-213)
			const uint32_t q = q0 + lane;
			const bool rj = q < m1 && !keep[q];
			const uint32_t rm = ballot16(rj);
			if (rj) { const uint32_t o = at + (uint32_t)__popc(rm & below); rej_rid[o] = (uint32_t)(members[q] >> 32); rej_group[o] = (uint32_t)g; }
			at += (uint32_t)__popc(rm);
		}
		if (nk == 1)                                                           // a contig of one is dissolved (:477-498)
			for (uint32_t q = m0 + lane; q < m1; q += 16) if (keep[q]) { rej_rid[at] = (uint32_t)(members[q] >> 32); rej_group[at] = (uint32_t)g; }
	}
}

extern "C" int mcom_groups_to_contigs(mcom_ctx *ctx, const uint64_t *d_members, const uint32_t *d_goff, size_t ng, const uint8_t *d_keep,
                                      const uint32_t *d_nkept, const uint16_t *d_sv, const uint16_t *d_reflen, const uint8_t *d_refs, int ref_stride,
                                      uint64_t n_have, uint64_t chars_have, uint64_t members_have, uint8_t *d_seq, uint64_t seq_cap, uint64_t *d_soff,
                                      uint64_t *d_mem, uint64_t mem_cap, uint64_t *d_moff, uint64_t off_cap, uint32_t *d_rej_rid, uint32_t *d_rej_group,
                                      uint64_t rej_cap, uint64_t *h_counts)
{
	if (!ctx || !h_counts) return MCOM_E_ARG;
	h_counts[0] = h_counts[1] = h_counts[2] = h_counts[3] = 0;
	if (ng == 0) return MCOM_OK;
	if (!d_members || !d_goff || !d_keep || !d_nkept || !d_sv || !d_reflen || !d_refs) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	if (ng >= (1ull << 31)) return mcom_fail(ctx, MCOM_E_ARG, "too many groups");
	const size_t e = ng + 1;
	int rc = mcom_ws_reserve(ctx, 3 * g_al256(e * 4) + g_al256(e * 8) + g_al256(mcom_scan_scratch_elems(e) * 4 + 1024) + g_al256(mcom_scan64_scratch_elems(e) * 8) + 256);
	if (rc) return rc;
	char *base = (char*)ctx->ws; size_t o = 0;
	auto take = [&](size_t b) { char *q = base + o; o += g_al256(b); return q; };
	uint32_t *slot = (uint32_t*)take(e * 4), *msz = (uint32_t*)take(e * 4), *rej = (uint32_t*)take(e * 4);
	uint64_t *rsz = (uint64_t*)take(e * 8);
	uint32_t *scr = (uint32_t*)take(mcom_scan_scratch_elems(e) * 4 + 1024);
	uint64_t *scr64 = (uint64_t*)take(mcom_scan64_scratch_elems(e) * 8);
	const unsigned gb = (unsigned)((e + 255) / 256);
	MCOM_LAUNCH(k_group_sizes, dim3(gb), dim3(256), 0, ctx->stream, d_goff, d_nkept, d_reflen, ng, slot, msz, rsz, rej);
	MCOM_LAUNCH_CHECK(ctx);
	if ((rc = mcom_scan_u32(ctx, slot, slot, e, scr)) || (rc = mcom_scan_u32(ctx, msz, msz, e, scr)) || (rc = mcom_scan_u32(ctx, rej, rej, e, scr)) ||
	    (rc = mcom_scan64(ctx, rsz, rsz, e, scr64))) return rc;
	uint32_t h32[3] = {0, 0, 0}; uint64_t chars = 0;
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &h32[0], slot + ng, 4));
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &h32[1], msz + ng, 4));
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &h32[2], rej + ng, 4));
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &chars, rsz + ng, 8));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	h_counts[0] = h32[0]; h_counts[1] = chars; h_counts[2] = h32[1]; h_counts[3] = h32[2];
	if (n_have + h32[0] + 1 > off_cap || chars_have + chars > seq_cap || members_have + h32[1] > mem_cap || h32[2] > rej_cap)
		return mcom_fail(ctx, MCOM_E_OVERFLOW, "contig set buffers too small for %u contigs, %llu chars, %u members, %u rejects", h32[0], (unsigned long long)chars, h32[1], h32[2]);
	if (!d_seq || !d_soff || !d_mem || !d_moff || (h32[2] && (!d_rej_rid || !d_rej_group))) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	if (n_have == 0) { MCOM_HIP(ctx, hipMemsetAsync(d_soff, 0, 8, ctx->stream)); MCOM_HIP(ctx, hipMemsetAsync(d_moff, 0, 8, ctx->stream)); }
	MCOM_LAUNCH(k_group_emit, dim3((unsigned)((ng * 16 + 255) / 256)), dim3(256), 0, ctx->stream, d_members, d_goff, ng, d_keep, d_nkept, d_sv, d_reflen,
	                   d_refs, ref_stride, slot, msz, rsz, rej, n_have, chars_have, members_have, d_seq, d_soff, d_mem, d_moff, d_rej_rid, d_rej_group);
	MCOM_LAUNCH_CHECK(ctx);
	MCOM_HIP(ctx, mcom_stream_sync(ctx));                      // the workspace arrays are in use until here
	return MCOM_OK;
}
