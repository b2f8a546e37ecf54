// minicom_amd/csrc/streams.hip -- the stream files of cluster_dump made on the device.
//
// The reference's print_encode (kthread_dump.c:142-236) walks every member of every contig on the host: the member's string
// (reverse-complemented when its direction bit is set, preprocess.c:22-37), compared base by base with the contig, written as
// run-length mismatch text (dif_char.txt), a 16-bit position delta (beg_pos.bin) and a direction bit (dir.bin); the contigs go to
// ref.bin and the unclustered reads to single.seq four bases per byte (breads.h:232-239).  All of that is a function of data
// that is in HBM when Stage 2 ends -- packed rows, N masks, the contig set -- so it is made there: one thread per member finds
// the mismatches as the set bits of (row XOR contig window) | N mask and emits the text over them (two passes: lengths, a prefix
// sum, the bytes), and the bit-packed streams are one thread per output byte.  What the host writes are finished file images.
#include "mcom_dev.hpp"

// reverse the order of the 32 two-bit groups of a word
__device__ __forceinline__ uint64_t st_rev_groups(uint64_t x)
{
	x = __brevll(x);
	return ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
}
// bit i of the low 32 bits -> bit 2i
__device__ __forceinline__ uint64_t st_spread32(uint64_t x)
{
	x &= 0xFFFFFFFFull;
	x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
	x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
	x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
	x = (x | (x << 2)) & 0x3333333333333333ull;
	x = (x | (x << 1)) & 0x5555555555555555ull;
	return x;
}

// contig c -> its first member's place: cid[q] = c for the members of c, and the member count in front of c's position deltas
// (beg_pos.bin: per contig a 32-bit count, then 16-bit deltas, kthread_dump.c:167-169, :224-225).  Sixteen lanes per contig.
__global__ __launch_bounds__(256) void k_st_contigs(const uint64_t *__restrict__ moff, size_t n_contigs, uint32_t *__restrict__ cid, uint16_t *__restrict__ pos16)
{
	const size_t c = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
	if (c >= n_contigs) return;
	const int lane = threadIdx.x & 15;
	const uint64_t m0 = moff[c], m1 = moff[c + 1];
	if (lane == 0) { const uint32_t n = (uint32_t)(m1 - m0); uint16_t *d = pos16 + 2 * c + m0; d[0] = (uint16_t)n; d[1] = (uint16_t)(n >> 16); }
	for (uint64_t q = m0 + lane; q < m1; q += 16) cid[q] = (uint32_t)c;
}

// The member's read as the reference compares it (N restored, reverse complement when dir), against its window of the contig:
// mm[w] = mismatching bases of word w on the even bits; t[] the oriented read, tn[] its N mask.
template <int W>
__device__ __forceinline__ void st_member(const uint64_t *__restrict__ packed, const uint64_t *__restrict__ nmask, int NW, int L, uint32_t rid, uint32_t pos, bool dir,
                                          const uint64_t *__restrict__ contig, uint64_t (&t)[W], uint64_t (&tn)[(W + 1) / 2], uint64_t (&mm)[W])
{
	constexpr int NWC = (W + 1) / 2;
	const int tail = 2 * L - 64 * (W - 1);
	uint64_t row[W], nm[NWC];
#pragma unroll
	for (int w = 0; w < W; ++w) row[w] = packed[(size_t)rid * W + w];
#pragma unroll
	for (int w = 0; w < NWC; ++w) nm[w] = (nmask && w < NW) ? nmask[(size_t)rid * NW + w] : 0ull;
	if (dir) {
		uint64_t tt[W];
#pragma unroll
		for (int w = 0; w < W; ++w) tt[w] = ~st_rev_groups(row[W - 1 - w]);
		const int drop = 64 * W - 2 * L;
#pragma unroll
		for (int w = 0; w < W; ++w) { const uint64_t a = tt[w], b = w + 1 < W ? tt[w + 1] : 0ull; t[w] = drop ? (a >> drop) | (b << (64 - drop)) : a; }
		uint64_t rr[NWC];
#pragma unroll
		for (int w = 0; w < NWC; ++w) rr[w] = (NW - 1 - w >= 0 && NW - 1 - w < NWC) ? __brevll(nm[NW - 1 - w]) : 0ull;
		const int dropn = 64 * NW - L;
#pragma unroll
		for (int w = 0; w < NWC; ++w) { const uint64_t a = rr[w], b = w + 1 < NWC ? rr[w + 1] : 0ull; tn[w] = dropn ? (a >> dropn) | (b << (64 - dropn)) : a; }
	} else {
#pragma unroll
		for (int w = 0; w < W; ++w) t[w] = row[w];
#pragma unroll
		for (int w = 0; w < NWC; ++w) tn[w] = nm[w];
	}
	if (tail < 64) t[W - 1] &= (1ull << tail) - 1;
	const uint64_t *src = contig + ((2 * (uint64_t)pos) >> 6);
	const int sh = (int)((2 * (uint64_t)pos) & 63);
	uint64_t cur = src[0];
#pragma unroll
	for (int w = 0; w < W; ++w) {
		const uint64_t nxt = src[w + 1];
		uint64_t win = sh ? (cur >> sh) | (nxt << (64 - sh)) : cur;
		cur = nxt;
		if (w == W - 1 && tail < 64) win &= (1ull << tail) - 1;
		const uint64_t x = win ^ t[w];
		uint64_t m = (x | (x >> 1)) & 0x5555555555555555ull;
		m |= st_spread32(tn[w >> 1] >> (32 * (w & 1)));                       // an N never equals the contig's base (kthread_dump.c:178-186)
		if (w == W - 1 && tail < 64) m &= (1ull << tail) - 1;
		mm[w] = m;
	}
}

// print_encode's text of one member (kthread_dump.c:198-221): a run of matches longer than one as its decimal length, a single
// match as the base itself, a mismatch as the read's base, trailing matches dropped, "0" for an identical read; then the newline.
template <int W, bool WRITE>
__device__ __forceinline__ uint32_t st_text(const uint64_t (&t)[W], const uint64_t (&tn)[(W + 1) / 2], const uint64_t (&mm)[W], uint8_t *__restrict__ out)
{
	uint32_t len = 0;
	int prev = -1;
	auto base_at = [&](int p) -> uint8_t {
		if ((tn[p >> 6] >> (p & 63)) & 1ull) return (uint8_t)'N';
		return (uint8_t)"ACGT"[(t[p >> 5] >> (2 * (p & 31))) & 3ull];
	};
#pragma unroll
	for (int w = 0; w < W; ++w) {
		uint64_t m = mm[w];
		while (m) {
			const int b = __ffsll((unsigned long long)m) - 1; m &= m - 1;
			const int p = 32 * w + (b >> 1);
			const int eq = p - prev - 1;
			if (eq > 1) {
				if (eq >= 100) { if (WRITE) out[len] = (uint8_t)('0' + eq / 100); ++len; }
				if (eq >= 10) { if (WRITE) out[len] = (uint8_t)('0' + (eq / 10) % 10); ++len; }
				if (WRITE) out[len] = (uint8_t)('0' + eq % 10); ++len;
			} else if (eq == 1) { if (WRITE) out[len] = base_at(p - 1); ++len; }
			if (WRITE) out[len] = base_at(p); ++len;
			prev = p;
		}
	}
	if (len == 0) { if (WRITE) out[0] = (uint8_t)'0'; len = 1; }
	if (WRITE) out[len] = (uint8_t)'\n';
	return len + 1;
}

// pass 1: text length, position delta
template <int W>
__global__ __launch_bounds__(256) void k_st_len(const uint64_t *__restrict__ packed, const uint64_t *__restrict__ nmask, int NW, int L, const uint64_t *__restrict__ mem,
                                                const uint64_t *__restrict__ moff, const uint32_t *__restrict__ cid, size_t n_members, const uint64_t *__restrict__ cbits,
                                                const uint64_t *__restrict__ coff, uint16_t *__restrict__ pos16, uint64_t *__restrict__ tlen)
{
	const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= n_members) return;
	const uint32_t c = cid[q];
	const uint64_t y = mem[q];
	const uint32_t rid = (uint32_t)(y >> 32), pos = (uint32_t)y >> 1;
	const uint32_t pre = q > moff[c] ? (uint32_t)mem[q - 1] >> 1 : 0u;
	pos16[2 * ((size_t)c + 1) + q] = (uint16_t)(pos - pre);
	uint64_t t[W], tn[(W + 1) / 2], mm[W];
	st_member<W>(packed, nmask, NW, L, rid, pos, (y & 1) != 0, cbits + coff[c], t, tn, mm);
	tlen[q] = st_text<W, false>(t, tn, mm, nullptr);
}
// pass 2: the bytes
template <int W>
__global__ __launch_bounds__(256) void k_st_emit(const uint64_t *__restrict__ packed, const uint64_t *__restrict__ nmask, int NW, int L, const uint64_t *__restrict__ mem,
                                                 const uint32_t *__restrict__ cid, size_t n_members, const uint64_t *__restrict__ cbits, const uint64_t *__restrict__ coff,
                                                 const uint64_t *__restrict__ toff, uint8_t *__restrict__ text)
{
	const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= n_members) return;
	const uint32_t c = cid[q];
	const uint64_t y = mem[q];
	uint64_t t[W], tn[(W + 1) / 2], mm[W];
	st_member<W>(packed, nmask, NW, L, (uint32_t)(y >> 32), (uint32_t)y >> 1, (y & 1) != 0, cbits + coff[c], t, tn, mm);
	(void)st_text<W, true>(t, tn, mm, text + toff[q]);
}
// dir.bin: one bit per member, least significant first (bit_push, breads.h:241-248)
__global__ void k_st_dirs(const uint64_t *__restrict__ mem, size_t n_members, uint8_t *__restrict__ out)
{
	const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (8 * b >= n_members) return;
	unsigned v = 0;
	for (int j = 0; j < 8; ++j) { const size_t q = 8 * b + j; if (q < n_members) v |= (unsigned)(mem[q] & 1ull) << j; }
	out[b] = (uint8_t)v;
}
// ref.bin: the contig strings back to back, four bases per byte (DNA_push, breads.h:232-239; A0 C1 G2 T3, kthread_dump.c:158-160)
__global__ void k_st_refbin(const uint8_t *__restrict__ seq, uint64_t chars, uint8_t *__restrict__ out)
{
	const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (4 * b >= chars) return;
	unsigned v = 0;
	for (int j = 0; j < 4; ++j) { const uint64_t i = 4 * b + j; if (i < chars) { const unsigned ch = seq[i]; v |= (ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : 3u) << (2 * j); } }
	out[b] = (uint8_t)v;
}
// single.seq: the listed reads back to back, four bases per byte (kthread_dump.c:390-417)
__global__ void k_st_singles(const uint64_t *__restrict__ packed, int W, int L, const uint32_t *__restrict__ rids, uint64_t n, uint8_t *__restrict__ out)
{
	const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t bases = n * (uint64_t)L;
	if (4 * b >= bases) return;
	unsigned v = 0;
	for (int j = 0; j < 4; ++j) {
		const uint64_t g = 4 * b + j;
		if (g >= bases) break;
		const uint64_t r = g / (uint64_t)L; const int i = (int)(g - r * (uint64_t)L);
		v |= (unsigned)((packed[(size_t)rids[r] * W + (i >> 5)] >> (2 * (i & 31))) & 3ull) << (2 * j);
	}
	out[b] = (uint8_t)v;
}
// does read rids[i] hold an N?
__global__ void k_st_has_n(const uint64_t *__restrict__ nmask, int NW, const uint32_t *__restrict__ rids, size_t n, uint8_t *__restrict__ flag)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint64_t any = 0;
	for (int w = 0; w < NW; ++w) any |= nmask[(size_t)rids[i] * NW + w];
	flag[i] = any ? 1 : 0;
}

extern "C" int mcom_dump_members(mcom_ctx *ctx, const uint64_t *d_packed, const uint64_t *d_nmask, int L, const uint64_t *d_cbits, const uint64_t *d_coff,
                                 const uint64_t *d_mem, const uint64_t *d_moff, size_t n_contigs, uint64_t n_members, uint8_t *d_pos, uint8_t *d_dir,
                                 uint8_t *d_text, uint64_t text_cap, uint64_t *h_text_bytes)
{
	return mcom_dump_members_at(ctx, d_packed, d_nmask, L, d_cbits, d_coff, d_mem, d_moff, n_contigs, n_members, d_pos, d_dir, d_text, text_cap, h_text_bytes, nullptr, 0, nullptr);
}
extern "C" int mcom_dump_members_at(mcom_ctx *ctx, const uint64_t *d_packed, const uint64_t *d_nmask, int L, const uint64_t *d_cbits, const uint64_t *d_coff,
                                    const uint64_t *d_mem, const uint64_t *d_moff, size_t n_contigs, uint64_t n_members, uint8_t *d_pos, uint8_t *d_dir,
                                    uint8_t *d_text, uint64_t text_cap, uint64_t *h_text_bytes, const uint64_t *h_at_members, int n_at, uint64_t *h_text_at)
{
	if (!ctx || !h_text_bytes) return MCOM_E_ARG;
	*h_text_bytes = 0;
	if (L < 1 || L > 256) return mcom_fail(ctx, MCOM_E_ARG, "read length %d out of range", L);
	if (n_contigs == 0 || n_members == 0) return MCOM_OK;
	if (!d_packed || !d_cbits || !d_coff || !d_mem || !d_moff || !d_pos || !d_dir) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	if ((uintptr_t)d_pos & 1) return mcom_fail(ctx, MCOM_E_ARG, "the position stream must be 2-byte aligned");
	const int W = mcom_words_per_read(L), NW = (L + 63) / 64;
	auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
	const size_t cid_b = al(n_members * 4), len_b = al((n_members + 1) * 8), scr_b = al(mcom_scan64_scratch_elems(n_members + 1) * 8 + 256);
	char *tmp = nullptr;
	if (mcom_dmalloc(&tmp, cid_b + len_b + scr_b) != hipSuccess) return mcom_fail(ctx, MCOM_E_NOMEM, "stream encoder: %zu bytes of scratch", cid_b + len_b + scr_b);
	struct Guard { mcom_ctx *c; char *p; ~Guard() { (void)hipStreamSynchronize(c->stream); mcom_dfree(p); } } guard{ctx, tmp};
	uint32_t *cid = (uint32_t*)tmp; uint64_t *tlen = (uint64_t*)(tmp + cid_b), *scr = (uint64_t*)(tmp + cid_b + len_b);
	MCOM_LAUNCH(k_st_contigs, dim3((unsigned)((n_contigs * 16 + 255) / 256)), dim3(256), 0, ctx->stream, d_moff, n_contigs, cid, (uint16_t*)d_pos);
	MCOM_LAUNCH_CHECK(ctx);
	const unsigned blocks = (unsigned)((n_members + 255) / 256);
	MCOM_HIP(ctx, hipMemsetAsync(tlen + n_members, 0, 8, ctx->stream));
#define MCOM_CASE(WW) case WW: MCOM_LAUNCH((k_st_len<WW>), dim3(blocks), dim3(256), 0, ctx->stream, d_packed, d_nmask, NW, L, d_mem, d_moff, cid, (size_t)n_members, d_cbits, d_coff, (uint16_t*)d_pos, tlen); break;
	switch (W) { MCOM_CASE(1) MCOM_CASE(2) MCOM_CASE(3) MCOM_CASE(4) MCOM_CASE(5) MCOM_CASE(6) MCOM_CASE(7) MCOM_CASE(8) default: return mcom_fail(ctx, MCOM_E_ARG, "unsupported read length"); }
#undef MCOM_CASE
	MCOM_LAUNCH_CHECK(ctx);
	MCOM_LAUNCH(k_st_dirs, dim3((unsigned)(((n_members + 7) / 8 + 255) / 256)), dim3(256), 0, ctx->stream, d_mem, (size_t)n_members, d_dir);
	MCOM_LAUNCH_CHECK(ctx);
	int rc = mcom_scan64(ctx, tlen, tlen, n_members + 1, scr);
	if (rc) return rc;
	uint64_t total = 0;
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &total, tlen + n_members, 8));
	for (int q = 0; q < n_at && h_at_members && h_text_at; ++q)                   // where the text of given members starts (stream sets cut at contig boundaries; checked at the entry)
		MCOM_HIP(ctx, hipMemcpyAsync(&h_text_at[q], tlen + h_at_members[q], 8, hipMemcpyDeviceToHost, ctx->stream));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	*h_text_bytes = total;
	if (!d_text && !text_cap) return MCOM_OK;                                   // a sizing call: d_pos and d_dir are complete, the text's length known
	if (total > text_cap || !d_text) return mcom_fail(ctx, MCOM_E_OVERFLOW, "mismatch text: %llu bytes, room for %llu", (unsigned long long)total, (unsigned long long)text_cap);
#define MCOM_CASE(WW) case WW: MCOM_LAUNCH((k_st_emit<WW>), dim3(blocks), dim3(256), 0, ctx->stream, d_packed, d_nmask, NW, L, d_mem, cid, (size_t)n_members, d_cbits, d_coff, tlen, d_text); break;
	switch (W) { MCOM_CASE(1) MCOM_CASE(2) MCOM_CASE(3) MCOM_CASE(4) MCOM_CASE(5) MCOM_CASE(6) MCOM_CASE(7) MCOM_CASE(8) default: break; }
#undef MCOM_CASE
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

extern "C" int mcom_dump_refbin(mcom_ctx *ctx, const uint8_t *d_seq, uint64_t chars, uint8_t *d_out)
{
	if (!ctx) return MCOM_E_ARG;
	if (!chars) return MCOM_OK;
	if (!d_seq || !d_out) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	const uint64_t bytes = (chars + 3) / 4;
	if ((bytes + 255) / 256 >= (1ull << 31)) return mcom_fail(ctx, MCOM_E_ARG, "too many contig bases for one launch");
	MCOM_LAUNCH(k_st_refbin, dim3((unsigned)((bytes + 255) / 256)), dim3(256), 0, ctx->stream, d_seq, chars, d_out);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

extern "C" int mcom_dump_singles(mcom_ctx *ctx, const uint64_t *d_packed, const uint32_t *d_rids, uint64_t n, int L, uint8_t *d_out)
{
	if (!ctx) return MCOM_E_ARG;
	if (L < 1 || L > 256) return mcom_fail(ctx, MCOM_E_ARG, "read length %d out of range", L);
	if (!n) return MCOM_OK;
	if (!d_packed || !d_rids || !d_out) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	const uint64_t bytes = (n * (uint64_t)L + 3) / 4;
	if ((bytes + 255) / 256 >= (1ull << 31)) return mcom_fail(ctx, MCOM_E_ARG, "too many unclustered reads for one launch");
	MCOM_LAUNCH(k_st_singles, dim3((unsigned)((bytes + 255) / 256)), dim3(256), 0, ctx->stream, d_packed, mcom_words_per_read(L), L, d_rids, n, d_out);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

extern "C" int mcom_rows_have_n(mcom_ctx *ctx, const uint64_t *d_nmask, const uint32_t *d_rids, size_t n, int L, uint8_t *d_flag)
{
	if (!ctx) return MCOM_E_ARG;
	if (L < 1 || L > 256) return mcom_fail(ctx, MCOM_E_ARG, "read length %d out of range", L);
	if (!n) return MCOM_OK;
	if (!d_nmask || !d_rids || !d_flag) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	MCOM_LAUNCH(k_st_has_n, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_nmask, (L + 63) / 64, d_rids, n, d_flag);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

// ---- the order-preserving and the paired-end file sets (kthread_dump.c:33-138, kthread_dump_pe.c:35-120, :218-619) ----
// Round 3 wrote these two modes with a host loop over every base of every member (~260 s per 100 M reads).  They differ from the
// default mode in the member order inside a contig (cmpcluster3: offset, then read id, kthread_cb.c:72-84), in one more stream per
// member (the ids), and in the pairing streams of the paired-end mode; mismatch text, positions, directions, contigs and singles are
// the default mode's kernels above.
namespace {
__global__ __launch_bounds__(256) void k_o3_fill(const uint64_t *__restrict__ mem, const uint64_t *__restrict__ moff, size_t n_contigs, mcom_mm128 *__restrict__ rec)
{
	const size_t c = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;         // sixteen lanes per contig
	if (c >= n_contigs) return;
	const int lane = threadIdx.x & 15;
	const uint64_t hi = (uint64_t)c << 32;
	for (uint64_t q = moff[c] + lane; q < moff[c + 1]; q += 16) { const uint64_t y = mem[q]; mcom_mm128 r; r.x = hi | (y >> 32); r.y = y; rec[q] = r; }
}
__global__ void k_o3_rekey(const mcom_mm128 *__restrict__ in, size_t n, int kb, mcom_mm128 *__restrict__ out)
{
	const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= n) return;
	const mcom_mm128 r = in[q];
	mcom_mm128 o; o.x = ((r.x >> 32) << kb) | (uint64_t)((uint32_t)r.y >> 1); o.y = r.y;
	out[q] = o;
}
__global__ void k_o3_out(const mcom_mm128 *__restrict__ rec, size_t n, uint64_t *__restrict__ mem2)
{
	const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (q < n) mem2[q] = rec[q].y;
}
__global__ __launch_bounds__(256) void k_st_cid(const uint64_t *__restrict__ moff, size_t n_contigs, uint32_t *__restrict__ cid)
{
	const size_t c = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
	if (c >= n_contigs) return;
	for (uint64_t q = moff[c] + (threadIdx.x & 15); q < moff[c + 1]; q += 16) cid[q] = (uint32_t)c;
}
// ids.bin of the order-preserving mode (kthread_dump.c:116-127): the read id, or -- at the position of the member before -- its
// difference to that member's id (the 16-bit position delta of the stream decides what "the same position" is)
__global__ void k_st_ids_order(const uint64_t *__restrict__ mem, const uint64_t *__restrict__ moff, const uint32_t *__restrict__ cid, size_t n_members, uint32_t *__restrict__ out)
{
	const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= n_members) return;
	const uint64_t y = mem[q];
	const uint32_t rid = (uint32_t)(y >> 32), pos = (uint32_t)y >> 1;
	uint32_t v = rid;
	if (q > moff[cid[q]]) {
		const uint64_t yp = mem[q - 1];
		if ((uint16_t)(pos - ((uint32_t)yp >> 1)) == 0) v = rid - (uint32_t)(yp >> 32);
	}
	out[q] = v;
}
// ids.txt of the paired-end mode (kthread_dump_pe.c:70-74): "%d %u\n" = which file the member came from, its read id
__device__ __forceinline__ int st_digits(uint32_t v) { int d = 1; while (v >= 10) { v /= 10; ++d; } return d; }
__global__ void k_st_idtext_len(const uint64_t *__restrict__ mem, size_t n, uint64_t *__restrict__ len)
{
	const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (q < n) len[q] = 3 + (uint64_t)st_digits((uint32_t)(mem[q] >> 32));
}
__global__ void k_st_idtext_emit(const uint64_t *__restrict__ mem, size_t n, uint32_t half, const uint64_t *__restrict__ off, uint8_t *__restrict__ text)
{
	const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= n) return;
	uint32_t rid = (uint32_t)(mem[q] >> 32);
	uint8_t *o = text + off[q];
	const int d = st_digits(rid);
	o[0] = rid < half ? (uint8_t)'0' : (uint8_t)'1'; o[1] = (uint8_t)' ';
	for (int i = d - 1; i >= 0; --i) { o[2 + i] = (uint8_t)('0' + rid % 10); rid /= 10; }
	o[2 + d] = (uint8_t)'\n';
}
// pairing (kthread_dump_pe.c:270-470, :583-612).  seq = the reads in the order the decoder writes them: the eight lists, then the
// members.  A read of the first file gets its number in that order among first-file reads (mpv); a read of the second file writes the
// number of its mate; a file bit per read says which it is.
__global__ void k_pe_rid(const uint32_t *__restrict__ lists, size_t n_list, const uint64_t *__restrict__ mem, size_t n_members, uint32_t half, uint32_t *__restrict__ first)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i > n_list + n_members) return;
	if (i == n_list + n_members) { first[i] = 0; return; }
	const uint32_t rid = i < n_list ? lists[i] : (uint32_t)(mem[i - n_list] >> 32);
	first[i] = rid < half ? 1u : 0u;
}
__global__ void k_pe_mpv(const uint32_t *__restrict__ lists, size_t n_list, const uint64_t *__restrict__ mem, size_t n_members, uint32_t half, const uint32_t *__restrict__ pre,
                         uint32_t *__restrict__ mpv)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_list + n_members) return;
	const uint32_t rid = i < n_list ? lists[i] : (uint32_t)(mem[i - n_list] >> 32);
	if (rid < half) mpv[rid] = pre[i];
}
__global__ void k_pe_ids(const uint32_t *__restrict__ lists, size_t n_list, const uint64_t *__restrict__ mem, size_t n_members, uint32_t half, const uint32_t *__restrict__ pre,
                         const uint32_t *__restrict__ mpv, uint32_t *__restrict__ ids_sp, uint32_t *__restrict__ ids_0)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_list + n_members) return;
	const uint32_t rid = i < n_list ? lists[i] : (uint32_t)(mem[i - n_list] >> 32);
	if (rid < half) return;
	const uint32_t second_before = (uint32_t)i - pre[i];                          // reads of the second file in front of this one
	const uint32_t mate = rid - half < half ? mpv[rid - half] : 0xFFFFFFFFu;          // (an id beyond the two files: not a read of this job)
	if (i < n_list) ids_sp[second_before] = mate;
	else ids_0[second_before - ((uint32_t)n_list - pre[n_list])] = mate;
}
// file.bin: one bit per read of a range of seq (1 = second file), least significant first (bit_push, breads.h:241-248)
__global__ void k_pe_bits(const uint32_t *__restrict__ first, size_t lo, size_t n, uint8_t *__restrict__ out)
{
	const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (8 * b >= n) return;
	unsigned v = 0;
	for (int j = 0; j < 8; ++j) { const size_t q = 8 * b + j; if (q < n && !first[lo + q]) v |= 1u << j; }
	out[b] = (uint8_t)v;
}
}  // namespace

extern "C" int mcom_members_order3(mcom_ctx *ctx, const uint64_t *d_mem, const uint64_t *d_moff, size_t n_contigs, uint64_t n_members, int key_bits, uint64_t *d_mem2)
{
	if (!ctx) return MCOM_E_ARG;
	if (key_bits < 1 || key_bits > 32) return mcom_fail(ctx, MCOM_E_ARG, "bad key bits");
	if (n_contigs == 0 || n_members == 0) return MCOM_OK;
	if (!d_mem || !d_moff || !d_mem2) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	if (n_members >= (1ull << 32)) return mcom_fail(ctx, MCOM_E_ARG, "more than 2^32-1 members");
	int cb = 1; while ((1ull << cb) < n_contigs) ++cb;
	if (key_bits + cb > 64) return mcom_fail(ctx, MCOM_E_ARG, "contig index and member offset need %d bits", key_bits + cb);
	mcom_mm128 *a = nullptr, *b = nullptr; uint32_t *tiles = nullptr;
	const size_t rec_b = (n_members + 1) * sizeof(mcom_mm128);
	if (mcom_dmalloc(&a, rec_b) != hipSuccess) return mcom_fail(ctx, MCOM_E_NOMEM, "member records");
	if (mcom_dmalloc(&b, rec_b) != hipSuccess) { mcom_dfree(a); return mcom_fail(ctx, MCOM_E_NOMEM, "member records"); }
	if (mcom_dmalloc(&tiles, (MCOM_GROUP_SCRATCH(n_members) + 16) * 4) != hipSuccess) { mcom_dfree(a); mcom_dfree(b); return mcom_fail(ctx, MCOM_E_NOMEM, "member records"); }
	int rc = mcom_ws_reserve(ctx, mcom_sort_ws_bytes(n_members));                  // (the oversized-group route of the grouped sort)
	const unsigned blocks = (unsigned)((n_members + 255) / 256);
	if (!rc) {
		MCOM_LAUNCH(k_o3_fill, dim3((unsigned)((n_contigs * 16 + 255) / 256)), dim3(256), 0, ctx->stream, d_mem, d_moff, n_contigs, a);
		rc = mcom_sort_groups_by_x(ctx, a, b, (size_t)n_members, d_moff, n_contigs, 32 + cb, tiles);                 // by read id inside every contig
	}
	if (!rc) {
		MCOM_LAUNCH(k_o3_rekey, dim3(blocks), dim3(256), 0, ctx->stream, (const mcom_mm128*)b, (size_t)n_members, key_bits, a);
		rc = mcom_sort_groups_by_x(ctx, a, b, (size_t)n_members, d_moff, n_contigs, key_bits + cb, tiles);           // then, stable, by offset
	}
	if (!rc) MCOM_LAUNCH(k_o3_out, dim3(blocks), dim3(256), 0, ctx->stream, (const mcom_mm128*)b, (size_t)n_members, d_mem2);
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) e = mcom_stream_sync(ctx);
	mcom_dfree(a); mcom_dfree(b); mcom_dfree(tiles);
	if (rc) return rc;
	if (e != hipSuccess) return mcom_fail(ctx, MCOM_E_HIP, "%s", hipGetErrorString(e));
	return MCOM_OK;
}

extern "C" int mcom_dump_ids_order(mcom_ctx *ctx, const uint64_t *d_mem, const uint64_t *d_moff, size_t n_contigs, uint64_t n_members, uint32_t *d_ids)
{
	if (!ctx) return MCOM_E_ARG;
	if (n_contigs == 0 || n_members == 0) return MCOM_OK;
	if (!d_mem || !d_moff || !d_ids) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	uint32_t *cid = nullptr;
	if (mcom_dmalloc(&cid, n_members * 4 + 16) != hipSuccess) return mcom_fail(ctx, MCOM_E_NOMEM, "id stream scratch");
	MCOM_LAUNCH(k_st_cid, dim3((unsigned)((n_contigs * 16 + 255) / 256)), dim3(256), 0, ctx->stream, d_moff, n_contigs, cid);
	MCOM_LAUNCH(k_st_ids_order, dim3((unsigned)((n_members + 255) / 256)), dim3(256), 0, ctx->stream, d_mem, d_moff, (const uint32_t*)cid, (size_t)n_members, d_ids);
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) e = mcom_stream_sync(ctx);
	mcom_dfree(cid);
	if (e != hipSuccess) return mcom_fail(ctx, MCOM_E_HIP, "%s", hipGetErrorString(e));
	return MCOM_OK;
}

extern "C" int mcom_dump_ids_text(mcom_ctx *ctx, const uint64_t *d_mem, uint64_t n_members, uint32_t half, uint8_t *d_text, uint64_t text_cap, uint64_t *h_text_bytes)
{
	return mcom_dump_ids_text_at(ctx, d_mem, n_members, half, d_text, text_cap, h_text_bytes, nullptr, 0, nullptr);
}
extern "C" int mcom_dump_ids_text_at(mcom_ctx *ctx, const uint64_t *d_mem, uint64_t n_members, uint32_t half, uint8_t *d_text, uint64_t text_cap, uint64_t *h_text_bytes,
                                     const uint64_t *h_at_members, int n_at, uint64_t *h_text_at)
{
	if (!ctx || !h_text_bytes) return MCOM_E_ARG;
	*h_text_bytes = 0;
	if (n_members == 0) return MCOM_OK;
	if (!d_mem) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	for (int q = 0; q < n_at && h_at_members && h_text_at; ++q)                   // (before any read-back is queued: a return from the middle would leave one pending)
		if (h_at_members[q] > n_members) return mcom_fail(ctx, MCOM_E_ARG, "member %llu of %llu", (unsigned long long)h_at_members[q], (unsigned long long)n_members);
	uint64_t *len = nullptr;
	if (mcom_dmalloc(&len, (n_members + 1) * 8) != hipSuccess) return mcom_fail(ctx, MCOM_E_NOMEM, "id text scratch");
	struct Guard { mcom_ctx *c; uint64_t *p; ~Guard() { (void)hipStreamSynchronize(c->stream); mcom_dfree(p); } } guard{ctx, len};
	const unsigned blocks = (unsigned)((n_members + 255) / 256);
	MCOM_HIP(ctx, hipMemsetAsync(len + n_members, 0, 8, ctx->stream));
	MCOM_LAUNCH(k_st_idtext_len, dim3(blocks), dim3(256), 0, ctx->stream, d_mem, (size_t)n_members, len);
	MCOM_LAUNCH_CHECK(ctx);
	int rc = mcom_scan64(ctx, len, len, n_members + 1, nullptr);
	if (rc) return rc;
	uint64_t total = 0;
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &total, len + n_members, 8));
	for (int q = 0; q < n_at && h_at_members && h_text_at; ++q)
		MCOM_HIP(ctx, hipMemcpyAsync(&h_text_at[q], len + h_at_members[q], 8, hipMemcpyDeviceToHost, ctx->stream));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	*h_text_bytes = total;
	if (!d_text && !text_cap) return MCOM_OK;                                    // a sizing call
	if (total > text_cap || !d_text) return mcom_fail(ctx, MCOM_E_OVERFLOW, "id text: %llu bytes, room for %llu", (unsigned long long)total, (unsigned long long)text_cap);
	MCOM_LAUNCH(k_st_idtext_emit, dim3(blocks), dim3(256), 0, ctx->stream, d_mem, (size_t)n_members, half, (const uint64_t*)len, d_text);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

extern "C" int mcom_dump_pairing(mcom_ctx *ctx, const uint32_t *d_lists, uint64_t n_list, const uint64_t *d_mem, uint64_t n_members, uint32_t half,
                                 uint32_t *d_ids_sp, uint8_t *d_file_sp, uint32_t *d_ids_0, uint8_t *d_file_0, uint64_t *h_counts)
{
	return mcom_dump_pairing_at(ctx, d_lists, n_list, d_mem, n_members, half, d_ids_sp, d_file_sp, d_ids_0, d_file_0, h_counts, nullptr, 0, nullptr);
}
extern "C" int mcom_dump_pairing_at(mcom_ctx *ctx, const uint32_t *d_lists, uint64_t n_list, const uint64_t *d_mem, uint64_t n_members, uint32_t half,
                                    uint32_t *d_ids_sp, uint8_t *d_file_sp, uint32_t *d_ids_0, uint8_t *d_file_0, uint64_t *h_counts,
                                    const uint64_t *h_at_members, int n_at, uint64_t *h_second_at)
{
	if (!ctx || !h_counts) return MCOM_E_ARG;
	h_counts[0] = h_counts[1] = 0;
	const uint64_t N = n_list + n_members;
	if (N == 0) return MCOM_OK;
	if (N >= (1ull << 32) - 1) return mcom_fail(ctx, MCOM_E_ARG, "more than 2^32-2 reads");
	if ((n_list && !d_lists) || (n_members && !d_mem) || !d_ids_sp || !d_file_sp || !d_ids_0 || !d_file_0) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	// every read of the two files is in exactly one place (a list or a contig), so the mate of every second-file read is listed too and
	// every entry of the mate table below is written; an input that breaks this would give pair ids made of uninitialised memory
	if (N != 2 * (uint64_t)half) return mcom_fail(ctx, MCOM_E_ARG, "pairing: %llu listed reads + %llu members, but two files of %u reads", (unsigned long long)n_list, (unsigned long long)n_members, half);
	for (int q = 0; q < n_at && h_at_members && h_second_at; ++q)
		if (h_at_members[q] > n_members) return mcom_fail(ctx, MCOM_E_ARG, "member %llu of %llu", (unsigned long long)h_at_members[q], (unsigned long long)n_members);
	uint32_t *first = nullptr, *pre = nullptr, *mpv = nullptr;
	auto drop = [&]() { if (first) mcom_dfree(first); if (pre) mcom_dfree(pre); if (mpv) mcom_dfree(mpv); };
	if (mcom_dmalloc(&first, (N + 1) * 4) != hipSuccess || mcom_dmalloc(&pre, (N + 1) * 4) != hipSuccess || mcom_dmalloc(&mpv, ((size_t)half + 1) * 4) != hipSuccess) {
		drop(); return mcom_fail(ctx, MCOM_E_NOMEM, "pairing scratch");
	}
	const unsigned blocks = (unsigned)((N + 1 + 255) / 256);
	MCOM_LAUNCH(k_pe_rid, dim3(blocks), dim3(256), 0, ctx->stream, d_lists, (size_t)n_list, d_mem, (size_t)n_members, half, first);
	int rc = mcom_scan_u32(ctx, first, pre, N + 1, nullptr);
	uint32_t cnt[2] = {0, 0};                                                    // first-file reads among the lists, among everything
	hipError_t e = hipSuccess;
	if (!rc) {
		MCOM_LAUNCH(k_pe_mpv, dim3(blocks), dim3(256), 0, ctx->stream, d_lists, (size_t)n_list, d_mem, (size_t)n_members, half, (const uint32_t*)pre, mpv);
		MCOM_LAUNCH(k_pe_ids, dim3(blocks), dim3(256), 0, ctx->stream, d_lists, (size_t)n_list, d_mem, (size_t)n_members, half, (const uint32_t*)pre, (const uint32_t*)mpv, d_ids_sp, d_ids_0);
		if (n_list) MCOM_LAUNCH(k_pe_bits, dim3((unsigned)(((n_list + 7) / 8 + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)first, (size_t)0, (size_t)n_list, d_file_sp);
		if (n_members) MCOM_LAUNCH(k_pe_bits, dim3((unsigned)(((n_members + 7) / 8 + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)first, (size_t)n_list, (size_t)n_members, d_file_0);
		e = hipGetLastError();
		if (e == hipSuccess) e = hipMemcpyAsync(&cnt[0], pre + n_list, 4, hipMemcpyDeviceToHost, ctx->stream);
		if (e == hipSuccess) e = hipMemcpyAsync(&cnt[1], pre + N, 4, hipMemcpyDeviceToHost, ctx->stream);
		for (int q = 0; q < n_at && h_at_members && h_second_at && e == hipSuccess; ++q) {   // (first-file reads in front of member m, turned into second-file ones below)
			h_second_at[q] = 0;
			e = hipMemcpyAsync(&h_second_at[q], pre + n_list + h_at_members[q], 4, hipMemcpyDeviceToHost, ctx->stream);
		}
	}
	if (e == hipSuccess) e = mcom_stream_sync(ctx);
	drop();
	if (rc) return rc;
	if (e != hipSuccess) return mcom_fail(ctx, MCOM_E_HIP, "%s", hipGetErrorString(e));
	for (int q = 0; q < n_at && h_at_members && h_second_at; ++q)                // second-file reads among the members in front of member m: where a set's peids start
		h_second_at[q] = h_at_members[q] - (h_second_at[q] - cnt[0]);
	h_counts[0] = n_list - cnt[0];                                               // second-file reads among the lists: entries of d_ids_sp
	h_counts[1] = n_members - (cnt[1] - cnt[0]);                                 // ... among the members: entries of d_ids_0
	return MCOM_OK;
}

// dir.bin / file.bin of ONE stream set: the bits of members [0, n) of d_mem packed from bit 0 (the reference starts a fresh bit writer per
// thread, kthread_dump.c:370-379; breads.h:241-248).  which = 0: the direction bit; 1: the file bit of the paired-end mode (read id >= half)
namespace {
__global__ void k_st_member_bits(const uint64_t *__restrict__ mem, size_t n, int which, uint32_t half, uint8_t *__restrict__ out)
{
	const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (8 * b >= n) return;
	unsigned v = 0;
	for (int j = 0; j < 8; ++j) { const size_t q = 8 * b + j; if (q < n) { const uint64_t y = mem[q]; v |= (unsigned)(which ? ((uint32_t)(y >> 32) >= half) : (unsigned)(y & 1ull)) << j; } }
	out[b] = (uint8_t)v;
}
}
extern "C" int mcom_dump_member_bits(mcom_ctx *ctx, const uint64_t *d_mem, uint64_t n_members, int which, uint32_t half, uint8_t *d_out)
{
	if (!ctx) return MCOM_E_ARG;
	if (!n_members) return MCOM_OK;
	if (!d_mem || !d_out || which < 0 || which > 1) return mcom_fail(ctx, MCOM_E_ARG, "bad arguments");
	MCOM_LAUNCH(k_st_member_bits, dim3((unsigned)(((n_members + 7) / 8 + 255) / 256)), dim3(256), 0, ctx->stream, d_mem, (size_t)n_members, which, half, d_out);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}
