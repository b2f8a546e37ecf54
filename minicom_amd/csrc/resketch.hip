// minicom_amd/csrc/resketch.hip -- the minimizers of a merged contig without sketching all of it (gfx950).
//
// combine_cluster sketches every merged contig from scratch (mm_sketch_lh_ori at kthread_cb.c:232-236).  The scan of
// sketch.c:116-165 keeps a ring of the last w entries and the newest smallest of them: what it emits while storing
// entry t is a function of entries t-w .. t (contigs.hip), so whether the k-mer ending at base p becomes a record is
// decided by the bases p-w-k+1 .. p+w, plus the position of the string's two ends when they are that close.  A merged
// contig differs from its parents only inside their overlap [lo, hi) (merge.hip, k_job_regions); therefore
//     records of the first parent with   p <= lo - w - 3                        are records of the merged contig,
//     records of the parent that reaches the end, at merged position  p >= hi + w + k + 1,  likewise,
// and only the k-mers ending in between are decided again, by a sketch of the segment around the overlap that is long
// enough for the scan to be in its steady state wherever a kept position is concerned (a few bases of slack on every
// bound).  Work per merge round drops from the merged contigs' total length to (overlap + 4w + 2k) per merge.
// Only for odd k: with even k palindromic k-mers store no entry (sketch.c:133) and the ring reaches further back.
#include "mcom_dev.hpp"
#include <algorithm>

int mcom_sketch_strings(mcom_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_off, const uint64_t *d_off_end, uint64_t chars_bound,
                        const uint32_t *d_ids, size_t n, int w, int k, uint32_t max_per_contig, uint32_t *d_moff, mcom_mm128 *d_out, size_t cap,
                        uint64_t *h_total);

namespace {
struct Job { uint32_t ci, cj, pos_ori, pos; };
struct RsPlan {
	uint32_t f, tp;          // first parent (offset 0); parent that supplies the records behind the overlap
	uint32_t tail_off;       // its offset in the merged contig
	int32_t s0;              // start of the sketched segment in the merged contig
	int32_t keep_lo, keep_hi;// records of the segment kept: merged position in [keep_lo, keep_hi]
	int32_t left_hi;         // records of f kept: position <= left_hi
	int32_t tail_lo;         // records of tp kept: merged position >= tail_lo
};
struct RsCut { uint32_t l0, nl, m0, nm, t0, nt; };

__device__ __forceinline__ uint32_t rec_pos(const mcom_mm128 &r) { return (uint32_t)r.y >> 1; }

__global__ __launch_bounds__(256) void k_rs_plan(const Job *__restrict__ jobs, size_t nj, const uint64_t *__restrict__ soff, const uint64_t *__restrict__ soff2,
                                                 int w, int k, RsPlan *__restrict__ plan, uint64_t *__restrict__ seg_start, uint64_t *__restrict__ seg_end,
                                                 unsigned long long *__restrict__ seg_chars)
{
	unsigned long long mine = 0;
	// (a grid of at most 1024 workgroups and one atomic each: a wave per 64 jobs put 47 000 atomics on one address in the first round)
	for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < nj; j += (size_t)gridDim.x * blockDim.x) {
		const Job J = jobs[j];
		const bool afirst = J.pos_ori >= J.pos;
		const uint32_t f = afirst ? J.ci : J.cj, s = afirst ? J.cj : J.ci;
		const int64_t sh = afirst ? (int64_t)J.pos_ori - J.pos : (int64_t)J.pos - J.pos_ori;
		const int64_t lf = (int64_t)(soff[f + 1] - soff[f]), ls = (int64_t)(soff[s + 1] - soff[s]);
		const int64_t m = (int64_t)(soff2[j + 1] - soff2[j]);
		const int64_t lo = sh < lf ? sh : lf;
		int64_t hi = lf < sh + ls ? lf : sh + ls;
		if (hi < lo) hi = lo;
		int64_t left_hi = lo - w - 3, keep_lo = lo - w - 2, keep_hi = hi + w + k, tail_lo = hi + w + k + 1;
		int64_t s0 = keep_lo - w - k - 2, s1 = keep_hi + w + 4;
		if (s0 <= 0) { s0 = 0; keep_lo = 0; left_hi = -1; }                  // the segment starts where the contig starts
		if (s1 >= m) { s1 = m; keep_hi = m; tail_lo = m + 1; }               // ... ends where it ends
		if ((s1 - s0) * 4 >= 3 * m) { s0 = 0; s1 = m; keep_lo = 0; left_hi = -1; keep_hi = m; tail_lo = m + 1; }   // not worth the stitching
		RsPlan P;
		P.f = f; P.tp = hi < lf ? f : s; P.tail_off = hi < lf ? 0u : (uint32_t)sh;
		P.s0 = (int32_t)s0; P.keep_lo = (int32_t)keep_lo; P.keep_hi = (int32_t)keep_hi; P.left_hi = (int32_t)left_hi; P.tail_lo = (int32_t)tail_lo;
		plan[j] = P;
		seg_start[j] = soff2[j] + (uint64_t)s0; seg_end[j] = soff2[j] + (uint64_t)s1;
		mine += (unsigned long long)(s1 - s0);
	}
	__shared__ unsigned long long wg_sum;
	if (threadIdx.x == 0) wg_sum = 0;
	for (int o = 32; o; o >>= 1) mine += __shfl_xor(mine, o);
	__syncthreads();
	if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&wg_sum, mine);
	__syncthreads();
	if (threadIdx.x == 0 && wg_sum) atomicAdd(seg_chars, wg_sum);
}

// first index in [a, b) whose position is >= v
__device__ __forceinline__ uint32_t lower_pos(const mcom_mm128 *__restrict__ rec, uint32_t a, uint32_t b, int64_t v, int64_t add)
{
	while (a < b) { const uint32_t mid = a + ((b - a) >> 1); if ((int64_t)rec_pos(rec[mid]) + add < v) a = mid + 1; else b = mid; }
	return a;
}

__global__ void k_rs_count(const RsPlan *__restrict__ plan, size_t nj, const mcom_mm128 *__restrict__ rec, const uint32_t *__restrict__ roff,
                           const mcom_mm128 *__restrict__ srec, const uint32_t *__restrict__ smoff, RsCut *__restrict__ cut, uint32_t *__restrict__ cnt)
{
	const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j > nj) return;
	if (j == nj) { cnt[j] = 0; return; }
	const RsPlan P = plan[j];
	RsCut C;
	C.l0 = roff[P.f];
	C.nl = P.left_hi < 0 ? 0u : lower_pos(rec, roff[P.f], roff[P.f + 1], (int64_t)P.left_hi + 1, 0) - C.l0;
	C.m0 = lower_pos(srec, smoff[j], smoff[j + 1], P.keep_lo, P.s0);
	C.nm = lower_pos(srec, C.m0, smoff[j + 1], (int64_t)P.keep_hi + 1, P.s0) - C.m0;
	C.t0 = lower_pos(rec, roff[P.tp], roff[P.tp + 1], P.tail_lo, P.tail_off);
	C.nt = roff[P.tp + 1] - C.t0;
	cut[j] = C;
	cnt[j] = C.nl + C.nm + C.nt;
}

__global__ __launch_bounds__(256) void k_rs_write(const RsPlan *__restrict__ plan, const RsCut *__restrict__ cut, size_t nj, const mcom_mm128 *__restrict__ rec,
                                                  const mcom_mm128 *__restrict__ srec, const uint32_t *__restrict__ roff2, mcom_mm128 *__restrict__ out, uint32_t id_base)
{
	const size_t j = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;     // sixteen lanes per job: a merged contig holds a few dozen records
	if (j >= nj) return;
	const int lane = threadIdx.x & 15;
	const RsPlan P = plan[j]; const RsCut C = cut[j];
	const uint64_t id = (uint64_t)(id_base + (uint32_t)j) << 32;             // the contig index (include/mcom.h, "contig ids")
	mcom_mm128 *dst = out + roff2[j];
	for (uint32_t t = lane; t < C.nl; t += 16) { mcom_mm128 r = rec[C.l0 + t]; r.y = id | (r.y & 0xFFFFFFFFull); dst[t] = r; }
	dst += C.nl;
	for (uint32_t t = lane; t < C.nm; t += 16) {
		mcom_mm128 r = srec[C.m0 + t];
		r.y = id | (uint64_t)((((uint32_t)r.y >> 1) + (uint32_t)P.s0) << 1) | (r.y & 1ull);
		dst[t] = r;
	}
	dst += C.nm;
	for (uint32_t t = lane; t < C.nt; t += 16) {
		mcom_mm128 r = rec[C.t0 + t];
		r.y = id | (uint64_t)((((uint32_t)r.y >> 1) + P.tail_off) << 1) | (r.y & 1ull);
		dst[t] = r;
	}
}

struct DevBlock {
	void *p = nullptr;
	~DevBlock() { if (p) mcom_dfree(p); }
	template <class T> T *get(size_t n) { if (p) { mcom_dfree(p); p = nullptr; } return mcom_dmalloc(&p, n * sizeof(T) + 256) == hipSuccess ? (T*)p : nullptr; }
};
}  // namespace

extern "C" int mcom_resketch_merged(mcom_ctx *ctx, const uint32_t *d_jobs, size_t nj, const uint64_t *d_soff, const mcom_mm128 *d_rec,
                                    const uint32_t *d_roff, const uint8_t *d_seq2, const uint64_t *d_soff2, uint64_t merged_chars, int w, int k,
                                    uint32_t *d_roff2, mcom_mm128 *d_rec2, size_t cap2, uint64_t *h_total, uint64_t *h_sketched_chars)
{
	return mcom_resketch_merged_at(ctx, d_jobs, nj, d_soff, d_rec, d_roff, d_seq2, d_soff2, merged_chars, w, k, 0, d_roff2, d_rec2, cap2, h_total, h_sketched_chars);
}
extern "C" int mcom_resketch_merged_at(mcom_ctx *ctx, const uint32_t *d_jobs, size_t nj, const uint64_t *d_soff, const mcom_mm128 *d_rec,
                                       const uint32_t *d_roff, const uint8_t *d_seq2, const uint64_t *d_soff2, uint64_t merged_chars, int w, int k, uint32_t id_base,
                                       uint32_t *d_roff2, mcom_mm128 *d_rec2, size_t cap2, uint64_t *h_total, uint64_t *h_sketched_chars)
{
	if (!ctx || !h_total) return MCOM_E_ARG;
	*h_total = 0;
	if (h_sketched_chars) *h_sketched_chars = 0;
	if (k < 1 || k > 31 || !(k & 1) || w < 1 || w > 128) return mcom_fail(ctx, MCOM_E_ARG, "w=%d (1..128) or k=%d (odd, 1..31) out of range", w, k);
	if (!d_roff2) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	if (nj == 0) { MCOM_HIP(ctx, hipMemsetAsync(d_roff2, 0, 4, ctx->stream)); return MCOM_OK; }
	if ((uint64_t)id_base + nj >= (1ull << 32) - 1) return mcom_fail(ctx, MCOM_E_ARG, "more than 2^32 - 2 contigs: record ids overflow");
	if (!d_jobs || !d_soff || !d_rec || !d_roff || !d_seq2 || !d_soff2 || !d_rec2) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	DevBlock b_plan, b_s0, b_s1, b_cnt, b_moff, b_srec, b_cut;
	RsPlan *plan = b_plan.get<RsPlan>(nj);
	uint64_t *seg_start = b_s0.get<uint64_t>(nj + 1), *seg_end = b_s1.get<uint64_t>(nj + 1);
	unsigned long long *d_chars = b_cnt.get<unsigned long long>(1);
	uint32_t *smoff = b_moff.get<uint32_t>(nj + 2);
	RsCut *cut = b_cut.get<RsCut>(nj);
	if (!plan || !seg_start || !seg_end || !d_chars || !smoff || !cut) return mcom_fail(ctx, MCOM_E_NOMEM, "resketch buffers");
	d_chars = (unsigned long long*)mcom_zeroed(ctx, d_chars, 8);
	if (!d_chars) return mcom_fail(ctx, MCOM_E_HIP, "clear");
	MCOM_LAUNCH(k_rs_plan, dim3((unsigned)std::min<size_t>((nj + 255) / 256, 1024)), dim3(256), 0, ctx->stream, (const Job*)d_jobs, nj, d_soff, d_soff2, w, k, plan, seg_start,
	                   seg_end, d_chars);
	MCOM_LAUNCH_CHECK(ctx);
	unsigned long long seg_chars = 0;
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &seg_chars, d_chars, 8));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	if (seg_chars > merged_chars) return mcom_fail(ctx, MCOM_E_ARG, "segments of %llu bases in contigs of %llu", seg_chars, (unsigned long long)merged_chars);
	if (h_sketched_chars) *h_sketched_chars = seg_chars;
	uint64_t stotal = 0;
	size_t scap = std::max<size_t>(1024, seg_chars / 8 + nj);
	mcom_mm128 *srec = nullptr;
	for (int attempt = 0;; ++attempt) {
		srec = b_srec.get<mcom_mm128>(scap);
		if (!srec) return mcom_fail(ctx, MCOM_E_NOMEM, "segment records");
		const int rc = mcom_sketch_strings(ctx, d_seq2, seg_start, seg_end, seg_chars, nullptr, nj, w, k, 0, smoff, srec, scap, &stotal);
		if (rc == MCOM_E_OVERFLOW && attempt == 0) { scap = stotal; continue; }
		if (rc) return rc;
		break;
	}
	int rc = mcom_ws_reserve(ctx, (mcom_scan_scratch_elems(nj + 1) + 256) * 4 + 1024);
	if (rc) return rc;
	MCOM_LAUNCH(k_rs_count, dim3((unsigned)((nj + 1 + 255) / 256)), dim3(256), 0, ctx->stream, plan, nj, d_rec, d_roff, srec, smoff, cut, d_roff2);
	MCOM_LAUNCH_CHECK(ctx);
	if ((rc = mcom_scan_u32(ctx, d_roff2, d_roff2, nj + 1, (uint32_t*)ctx->ws))) return rc;
	uint32_t total = 0;
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &total, d_roff2 + nj, 4));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	*h_total = total;
	if (total > cap2) return mcom_fail(ctx, MCOM_E_OVERFLOW, "%u minimizers but room for %zu", total, cap2);
	MCOM_LAUNCH(k_rs_write, dim3((unsigned)((nj * 16 + 255) / 256)), dim3(256), 0, ctx->stream, plan, cut, nj, d_rec, srec, d_roff2, d_rec2, id_base);
	MCOM_LAUNCH_CHECK(ctx);
	MCOM_HIP(ctx, mcom_stream_sync(ctx));                          // the temporaries go back to the pool
	return MCOM_OK;
}
