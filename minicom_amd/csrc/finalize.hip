// minicom_amd/csrc/finalize.hip -- the member lists after Stage 2, assembled on the device (gfx950).
//
// The reference sorts every contig's members at the start of each scan (cmpcluster2: offset, then direction,
// kthread_hash_realign.c:318, kthread_cb.c:54-69) and appends claimed reads behind them (:408-409, :474-475).  After m
// passes contig c therefore holds
//        stable_sort( C(c) + P_1(c) + ... + P_{m-1}(c) )  +  P_m(c)
// with P_i(c) the members pass i appended, in appending order.  Stage 2 itself never reads the lists, so they are
// assembled once, here: every member becomes a record { contig << kb | offset<<1|dir , member }, the members of the
// last pass with the largest key their contig can have, laid out contig by contig as [C(c) | P_1(c) | ... | P_m(c)], and one
// stable sort inside every contig's region (tiles of whole contigs in LDS, sort.hip; the global passes for a tile swollen by
// a contig of thousands of members) yields the lists -- equal keys keep the order C before P_1 before P_2 ..., each in its own order,
// which is what the reference's stable merge sort leaves.
#include "mcom_dev.hpp"

namespace {
__global__ void k_mf_cnt0(const uint64_t *__restrict__ moff, size_t n, unsigned long long *__restrict__ cnt, unsigned long long *__restrict__ cursor)
{
	const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (c > n) return;
	const unsigned long long v = c < n ? moff[c + 1] - moff[c] : 0ull;
	cnt[c] = v; cursor[c] = v;
}
__global__ void k_mf_cnt(const uint32_t *__restrict__ ac, size_t n, unsigned long long *__restrict__ cnt)
{
	const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t < n) atomicAdd(&cnt[ac[t]], 1ull);                                // spread over millions of contigs
}
// the members a contig had before Stage 2, at the head of its final region
__global__ __launch_bounds__(256) void k_mf_base(const uint64_t *__restrict__ mem, const uint64_t *__restrict__ moff, const uint64_t *__restrict__ moff2, size_t n, int kb,
                                                 mcom_mm128 *__restrict__ rec)
{
	const size_t c = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;     // sixteen lanes per contig
	if (c >= n) return;
	const int lane = threadIdx.x & 15;
	const uint64_t a = moff[c], b = moff[c + 1], d = moff2[c];
	const uint64_t hi = (uint64_t)c << kb;
	for (uint64_t q = a + lane; q < b; q += 16) { const uint64_t y = mem[q]; mcom_mm128 r; r.x = hi | (uint64_t)(uint32_t)y; r.y = y; rec[d + (q - a)] = r; }
}
// what one pass appended, behind what the contig holds so far: the pass's list ascends in the contig (the claim key starts with it,
// realign.hip), so a member's place among its contig's appends is its distance from the first entry of that contig
__global__ void k_mf_app(const uint32_t *__restrict__ ac, const uint64_t *__restrict__ am, size_t n, int kb, int last, const uint64_t *__restrict__ moff2,
                         const unsigned long long *__restrict__ cursor, mcom_mm128 *__restrict__ rec)
{
	const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n) return;
	const uint32_t c = ac[t];
	size_t lo = 0, hi = t;                                                   // first entry of contig c
	while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (ac[mid] < c) lo = mid + 1; else hi = mid; }
	const uint64_t y = am[t];
	mcom_mm128 r;
	r.x = ((uint64_t)c << kb) | (last ? ((1ull << kb) - 1) : (uint64_t)(uint32_t)y);
	r.y = y;
	rec[moff2[c] + cursor[c] + (t - lo)] = r;
}
__global__ void k_mf_advance(const uint32_t *__restrict__ ac, size_t n, unsigned long long *__restrict__ cursor)
{
	const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n) return;
	const uint32_t c = ac[t];
	if (t + 1 < n && ac[t + 1] == c) return;                                 // the last entry of a contig moves its cursor
	size_t lo = 0, hi = t;
	while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (ac[mid] < c) lo = mid + 1; else hi = mid; }
	cursor[c] += (unsigned long long)(t - lo + 1);
}
__global__ void k_mf_out(const mcom_mm128 *__restrict__ rec, size_t n, uint64_t *__restrict__ mem2)
{
	const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t < n) mem2[t] = rec[t].y;
}
}  // namespace

extern "C" int mcom_members_finalize(mcom_ctx *ctx, const uint64_t *d_mem, const uint64_t *d_moff, size_t n_contigs, uint64_t n_members,
                                     const uint32_t *const *d_app_contig, const uint64_t *const *d_app_member, const uint64_t *h_app_n,
                                     int n_passes, int key_bits, uint64_t *d_mem2, uint64_t *d_moff2)
{
	if (!ctx) return MCOM_E_ARG;
	if (n_passes < 1 || !h_app_n || key_bits < 1 || key_bits > 32) return mcom_fail(ctx, MCOM_E_ARG, "bad finalize arguments");
	if (n_contigs == 0) return MCOM_OK;
	if (!d_mem || !d_moff || !d_mem2 || !d_moff2) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	int cb = 1; while ((1ull << cb) < n_contigs) ++cb;
	if (key_bits + cb > 64) return mcom_fail(ctx, MCOM_E_ARG, "contig index and member offset need %d bits", key_bits + cb);
	uint64_t total = n_members;
	for (int i = 0; i < n_passes; ++i) {
		total += h_app_n[i];
		if (h_app_n[i] && (!d_app_contig || !d_app_member || !d_app_contig[i] || !d_app_member[i])) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	}
	if (total >= (1ull << 32)) return mcom_fail(ctx, MCOM_E_ARG, "more than 2^32-1 members");
	// Every record goes straight to its contig's final region (counts, scan, fill), so that the lists only have to be sorted
	// inside their regions: tiles of whole contigs in LDS (sort.hip), one read and one write of the records.
	mcom_mm128 *rec = nullptr; unsigned long long *cnt = nullptr;
	const size_t scr_elems = mcom_scan64_scratch_elems(n_contigs + 1);
	const size_t tile_words = (MCOM_GROUP_SCRATCH(total) + 1) / 2;
	if (mcom_dmalloc(&rec, (total + 1) * sizeof(mcom_mm128)) != hipSuccess) return mcom_fail(ctx, MCOM_E_NOMEM, "member records");
	if (mcom_dmalloc(&cnt, (2 * (n_contigs + 1) + scr_elems + tile_words + 64) * 8) != hipSuccess) { mcom_dfree(rec); return mcom_fail(ctx, MCOM_E_NOMEM, "member counts"); }
	unsigned long long *cursor = cnt + n_contigs + 1, *scr = cursor + n_contigs + 1;
	uint32_t *tiles = (uint32_t*)(scr + scr_elems + 8);
	int rc = mcom_ws_reserve(ctx, mcom_sort_ws_bytes(total));
	if (rc) { mcom_dfree(rec); mcom_dfree(cnt); return rc;	}
	const unsigned cblocks = (unsigned)((n_contigs + 1 + 255) / 256);
	MCOM_LAUNCH(k_mf_cnt0, dim3(cblocks), dim3(256), 0, ctx->stream, d_moff, n_contigs, cnt, cursor);
	for (int i = 0; i < n_passes; ++i)
		if (h_app_n[i]) MCOM_LAUNCH(k_mf_cnt, dim3((unsigned)((h_app_n[i] + 255) / 256)), dim3(256), 0, ctx->stream, d_app_contig[i], (size_t)h_app_n[i], cnt);
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) {
		rc = mcom_scan64(ctx, (const uint64_t*)cnt, d_moff2, n_contigs + 1, (uint64_t*)scr);
		if (!rc) {
			MCOM_LAUNCH(k_mf_base, dim3((unsigned)((n_contigs * 16 + 255) / 256)), dim3(256), 0, ctx->stream, d_mem, d_moff, d_moff2, n_contigs, key_bits, rec);
			for (int i = 0; i < n_passes; ++i) {
				if (!h_app_n[i]) continue;
				const unsigned blocks = (unsigned)((h_app_n[i] + 255) / 256);
				MCOM_LAUNCH(k_mf_app, dim3(blocks), dim3(256), 0, ctx->stream, d_app_contig[i], d_app_member[i], (size_t)h_app_n[i], key_bits,
				                   i == n_passes - 1 ? 1 : 0, d_moff2, cursor, rec);
				MCOM_LAUNCH(k_mf_advance, dim3(blocks), dim3(256), 0, ctx->stream, d_app_contig[i], (size_t)h_app_n[i], cursor);
			}
			e = hipGetLastError();
		}
		if (!rc && e == hipSuccess) {
			const mcom_mm128 *sorted = (const mcom_mm128*)ctx->ws;
			rc = mcom_sort_groups_by_x(ctx, rec, (mcom_mm128*)ctx->ws, (size_t)total, d_moff2, n_contigs, key_bits + cb, tiles);
			if (!rc) {
				MCOM_LAUNCH(k_mf_out, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, sorted, (size_t)total, d_mem2);
				e = hipGetLastError();
			}
		}
	}
	if (e == hipSuccess) e = mcom_stream_sync(ctx);               // the temporaries go back to the pool
	mcom_dfree(rec); mcom_dfree(cnt);
	if (e != hipSuccess) return mcom_fail(ctx, MCOM_E_HIP, "%s", hipGetErrorString(e));
	return rc;
}
