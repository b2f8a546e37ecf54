// minicom_amd/csrc/merge.hip -- the contig set of combine_cluster kept on the device between merge rounds (gfx950).
//
// A merge round of the reference (kthread_cb.c:570-630) = find_next (candidates, first-come claiming, member merge,
// construct_ref2) followed by cp_cluster (merged contigs first, untouched ones behind them).  Only the claiming is
// order dependent; everything else is data movement over the members (8 B each) and the consensus strings, which is
// done here so that neither leaves HBM between rounds:
//   mcom_contig_layout      packed-word offsets and lengths of a contig set
//   mcom_merge_members      member lists of the claimed pairs, shifted and in cmpcluster2 order       (:297-325, :107)
//   mcom_merge_consensus_jobs  construct_ref2 of every merged list, tiles made on the device          (:327)
//   mcom_contigs_carry      the untouched contigs copied behind the merged ones                        (:397-434)
//   mcom_records_carry      their minimizers re-labelled for the next round instead of sketched again
#include "mcom_dev.hpp"

// (the exclusive scans live in scan.hip: one launch each)
static inline int scan64(mcom_ctx *ctx, const uint64_t *in, uint64_t *out, size_t n, uint64_t *scratch) { return mcom_scan64(ctx, in, out, n, scratch); }
static inline size_t scan64_scratch_elems(size_t n) { return mcom_scan64_scratch_elems(n); }

// bump allocator over the context workspace
struct WsCut {
	char *base; size_t off;
	template <class T> T *take(size_t n) { T *p = (T*)(base + off); off += (n * sizeof(T) + 255) & ~(size_t)255; return p; }
};
static inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

extern "C" int mcom_scan_u64(mcom_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, size_t n)
{
	if (!ctx) return MCOM_E_ARG;
	if (n == 0) return MCOM_OK;
	if (!d_in || !d_out) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	int rc = mcom_ws_reserve(ctx, al256(scan64_scratch_elems(n) * 8) + 256);
	if (rc) return rc;
	return scan64(ctx, d_in, d_out, n, (uint64_t*)ctx->ws);
}

// ---- layout ----------------------------------------------------------------------------------------------------------
__global__ void k_layout(const uint64_t *__restrict__ soff, size_t n, uint64_t *__restrict__ words, uint32_t *__restrict__ clen)
{
	const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (c > n) return;
	if (c == n) { words[c] = 0; return; }
	const uint64_t len = soff[c + 1] - soff[c];
	clen[c] = (uint32_t)len;
	words[c] = (2 * len + 63) / 64 + 1;                                    // one padding word behind every contig
}

extern "C" int mcom_contig_layout(mcom_ctx *ctx, const uint64_t *d_soff, size_t n, uint64_t *d_coff_words, uint32_t *d_clen, uint64_t *h_total_words)
{
	if (!ctx || !h_total_words) return MCOM_E_ARG;
	*h_total_words = 0;
	if (!d_coff_words) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	if (n == 0) { MCOM_HIP(ctx, hipMemsetAsync(d_coff_words, 0, 8, ctx->stream)); return MCOM_OK; }
	if (!d_soff || !d_clen) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	int rc = mcom_ws_reserve(ctx, al256(scan64_scratch_elems(n + 1) * 8) + 256);
	if (rc) return rc;
	MCOM_LAUNCH(k_layout, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, ctx->stream, d_soff, n, d_coff_words, d_clen);
	MCOM_LAUNCH_CHECK(ctx);
	if ((rc = scan64(ctx, d_coff_words, d_coff_words, n + 1, (uint64_t*)ctx->ws))) return rc;
	MCOM_HIP(ctx, mcom_d2h_async(ctx, h_total_words, d_coff_words + n, 8));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	return MCOM_OK;
}

// the largest value of a workgroup into *dst: one atomic per workgroup at most (every thread of the workgroup calls this).  Round 4:
// a per-thread "if (v > *dst) atomicMax(dst, v)" let thousands of threads through while the maximum crept up -- lengths differ --
// and the atomics of one address queue on its L2 channel at ~4 ns each: 0.4 ms for 2 M contigs, 0.55 ms for 3 M merged ones.
__device__ __forceinline__ void block_max_to(unsigned long long v, unsigned long long *dst)
{
	__shared__ unsigned long long wg_max;
	if (threadIdx.x == 0) wg_max = 0;
	for (int o = 32; o; o >>= 1) { const unsigned long long t = __shfl_xor(v, o); v = t > v ? t : v; }
	__syncthreads();
	if ((threadIdx.x & 63) == 0 && v) atomicMax(&wg_max, v);
	__syncthreads();
	if (threadIdx.x == 0 && wg_max > *dst) atomicMax(dst, wg_max);
}

// windows of L bases per contig (kthread_hash_realign.c:320: j < strlen(ref) - readlen + 1) and their running total
__global__ void k_window_counts(const uint64_t *__restrict__ soff, size_t n, int L, uint64_t *__restrict__ nw, unsigned long long *__restrict__ maxlen)
{
	const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t len = 0;
	if (c == n) nw[c] = 0;
	else if (c < n) {
		len = soff[c + 1] - soff[c];
		nw[c] = len >= (uint64_t)L ? len - (uint64_t)L + 1 : 0;
	}
	block_max_to(len, maxlen);
}

extern "C" int mcom_window_layout(mcom_ctx *ctx, const uint64_t *d_soff, size_t n, int L, uint64_t *d_woff, uint64_t *h_n_windows, uint64_t *h_maxlen)
{
	if (!ctx || !h_n_windows) return MCOM_E_ARG;
	*h_n_windows = 0;
	if (h_maxlen) *h_maxlen = 0;
	if (!d_woff || L < 1) return mcom_fail(ctx, MCOM_E_ARG, "bad window layout arguments");
	if (n == 0) { MCOM_HIP(ctx, hipMemsetAsync(d_woff, 0, 8, ctx->stream)); return MCOM_OK; }
	if (!d_soff) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	int rc = mcom_ws_reserve(ctx, al256(scan64_scratch_elems(n + 1) * 8) + 512);
	if (rc) return rc;
	unsigned long long *mx = (unsigned long long*)mcom_zeroed(ctx, (char*)ctx->ws + al256(scan64_scratch_elems(n + 1) * 8), 8);
	if (!mx) return mcom_fail(ctx, MCOM_E_HIP, "clear");
	MCOM_LAUNCH(k_window_counts, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, ctx->stream, d_soff, n, L, d_woff, mx);
	MCOM_LAUNCH_CHECK(ctx);
	if ((rc = scan64(ctx, d_woff, d_woff, n + 1, (uint64_t*)ctx->ws))) return rc;
	unsigned long long hm = 0;
	MCOM_HIP(ctx, mcom_d2h_async(ctx, h_n_windows, d_woff + n, 8));
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &hm, mx, 8));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	if (h_maxlen) *h_maxlen = hm;
	return MCOM_OK;
}

// ---- member merge ------------------------------------------------------------------------------------------------
struct Job { uint32_t ci, cj, pos_ori, pos; };

__global__ void k_job_counts(const Job *__restrict__ jobs, size_t nj, const uint64_t *__restrict__ moff, uint64_t *__restrict__ cnt)
{
	const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j > nj) return;
	if (j == nj) { cnt[j] = 0; return; }
	const Job J = jobs[j];
	cnt[j] = (moff[J.ci + 1] - moff[J.ci]) + (moff[J.cj + 1] - moff[J.cj]);
}

// one wave per job: the list of the contig whose anchor lies further right first, the other one shifted behind it
// (:302-325); the sort key is the job and the low word of the member (offset<<1 | dir = cmpcluster2's order)
__global__ __launch_bounds__(256) void k_job_fill(const Job *__restrict__ jobs, size_t nj, const uint64_t *__restrict__ mem,
                                                  const uint64_t *__restrict__ moff, const uint64_t *__restrict__ jmoff, int kb,
                                                  mcom_mm128 *__restrict__ rec, unsigned int *__restrict__ err)
{
	const size_t j = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;     // sixteen lanes per job: a job is two dozen members
	if (j >= nj) return;
	const int lane = threadIdx.x & 15;
	const Job J = jobs[j];
	const bool afirst = J.pos_ori >= J.pos;
	const uint32_t f = afirst ? J.ci : J.cj, s = afirst ? J.cj : J.ci;
	const uint64_t sh = (uint64_t)(afirst ? J.pos_ori - J.pos : J.pos - J.pos_ori) << 1;
	const uint64_t f0 = moff[f], nf = moff[f + 1] - f0, s0 = moff[s], ns = moff[s + 1] - s0;
	mcom_mm128 *dst = rec + jmoff[j];
	const uint64_t jk = (uint64_t)j << kb, kmask = (1ull << kb) - 1;
	bool bad = false;
	for (uint64_t t = lane; t < nf; t += 16) {
		const uint64_t y = mem[f0 + t];
		const uint64_t key = (uint32_t)y;
		bad |= key > kmask;
		mcom_mm128 r; r.x = jk | (key & kmask); r.y = y; dst[t] = r;
	}
	for (uint64_t t = lane; t < ns; t += 16) {
		const uint64_t y = mem[s0 + t] + sh;
		const uint64_t key = (uint32_t)y;
		bad |= key > kmask;
		mcom_mm128 r; r.x = jk | (key & kmask); r.y = y; dst[nf + t] = r;
	}
	if (bad) atomicOr(err, 1u);
}
__global__ void k_job_emit(const mcom_mm128 *__restrict__ rec, size_t n, uint64_t *__restrict__ jm)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) jm[i] = rec[i].y;
}
// rend = the last member's offset + L: members are sorted and all reads have one length (construct_ref2 :112-113)
__global__ void k_job_len(const mcom_mm128 *__restrict__ rec, const uint64_t *__restrict__ jmoff, size_t nj, int L, uint64_t *__restrict__ len,
                          unsigned long long *__restrict__ maxlen)
{
	const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t v = 0;
	if (j == nj) len[j] = 0;
	else if (j < nj) {
		v = (uint64_t)((uint32_t)rec[jmoff[j + 1] - 1].y >> 1) + (uint64_t)L;
		len[j] = v;
	}
	block_max_to(v, maxlen);
}

extern "C" int mcom_merge_members(mcom_ctx *ctx, const uint64_t *d_mem, const uint64_t *d_moff, const uint32_t *d_jobs, size_t nj, int L,
                                  int key_bits, uint64_t *d_jm, uint64_t *d_jmoff, uint64_t *d_jroff, uint64_t *h_totals)
{
	return mcom_merge_members_cap(ctx, d_mem, d_moff, d_jobs, nj, L, key_bits, d_jm, ~(uint64_t)0, d_jmoff, d_jroff, h_totals);
}
extern "C" int mcom_merge_members_cap(mcom_ctx *ctx, const uint64_t *d_mem, const uint64_t *d_moff, const uint32_t *d_jobs, size_t nj, int L,
                                      int key_bits, uint64_t *d_jm, uint64_t jm_cap, uint64_t *d_jmoff, uint64_t *d_jroff, uint64_t *h_totals)
{
	if (!ctx || !h_totals) return MCOM_E_ARG;
	h_totals[0] = h_totals[1] = h_totals[2] = 0;
	if (nj == 0) {
		if (d_jmoff) MCOM_HIP(ctx, hipMemsetAsync(d_jmoff, 0, 8, ctx->stream));
		if (d_jroff) MCOM_HIP(ctx, hipMemsetAsync(d_jroff, 0, 8, ctx->stream));
		return MCOM_OK;
	}
	if (!d_mem || !d_moff || !d_jobs || !d_jm || !d_jmoff || !d_jroff) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	if (L < 1 || L > 256 || key_bits < 2 || key_bits > 29) return mcom_fail(ctx, MCOM_E_ARG, "bad merge arguments");
	int jb = 1; while ((1ull << jb) < nj) ++jb;
	if (nj >= (1ull << 31)) return mcom_fail(ctx, MCOM_E_ARG, "too many merges");
	const Job *jobs = (const Job*)d_jobs;
	int rc = mcom_ws_reserve(ctx, al256(scan64_scratch_elems(nj + 1) * 8) + 1024);
	if (rc) return rc;
	const unsigned jblocks = (unsigned)((nj + 1 + 255) / 256);
	MCOM_LAUNCH(k_job_counts, dim3(jblocks), dim3(256), 0, ctx->stream, jobs, nj, d_moff, d_jmoff);
	MCOM_LAUNCH_CHECK(ctx);
	if ((rc = scan64(ctx, d_jmoff, d_jmoff, nj + 1, (uint64_t*)ctx->ws))) return rc;
	uint64_t total = 0;
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &total, d_jmoff + nj, 8));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	if (total >= (1ull << 32)) return mcom_fail(ctx, MCOM_E_ARG, "too many members in one merge round");
	if (total > jm_cap) { h_totals[0] = total; return mcom_fail(ctx, MCOM_E_OVERFLOW, "%llu merged members but room for %llu", (unsigned long long)total, (unsigned long long)jm_cap); }
	const size_t rec_b = al256(total * sizeof(mcom_mm128));
	if ((rc = mcom_ws_reserve(ctx, rec_b + mcom_sort_ws_bytes(total) + al256(scan64_scratch_elems(nj + 1) * 8) + al256(MCOM_GROUP_SCRATCH(total) * 4) + 1024))) return rc;
	WsCut w{(char*)ctx->ws, 0};
	mcom_mm128 *rec = w.take<mcom_mm128>(total);
	void *sortws = w.take<char>(mcom_sort_ws_bytes(total));
	uint64_t *scr = w.take<uint64_t>(scan64_scratch_elems(nj + 1));
	uint32_t *tiles = w.take<uint32_t>(MCOM_GROUP_SCRATCH(total));
	unsigned long long *meta = (unsigned long long*)mcom_zeroed(ctx, w.take<unsigned long long>(4), 32);   // [0] maxlen, [1] error flag
	if (!meta) return mcom_fail(ctx, MCOM_E_HIP, "clear");
	MCOM_LAUNCH(k_job_fill, dim3((unsigned)((nj * 16 + 255) / 256)), dim3(256), 0, ctx->stream, jobs, nj, d_mem, d_moff, d_jmoff, key_bits, rec,
	                   (unsigned int*)(meta + 1));
	MCOM_LAUNCH_CHECK(ctx);
	{
		// the records lie job by job already: tiles of whole jobs, sorted in LDS (sort.hip); the sort workspace starts with a record buffer
		mcom_mm128 *sorted = (mcom_mm128*)sortws;
		if ((rc = mcom_sort_groups_by_x(ctx, rec, sorted, total, d_jmoff, nj, key_bits + jb, tiles))) return rc;
		rec = sorted;
	}
	MCOM_LAUNCH(k_job_emit, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, rec, (size_t)total, d_jm);
	MCOM_LAUNCH(k_job_len, dim3(jblocks), dim3(256), 0, ctx->stream, rec, d_jmoff, nj, L, d_jroff, meta);
	MCOM_LAUNCH_CHECK(ctx);
	if ((rc = scan64(ctx, d_jroff, d_jroff, nj + 1, scr))) return rc;
	unsigned long long hm[2] = {0, 0}; uint64_t chars = 0;
	MCOM_HIP(ctx, mcom_d2h_async(ctx, hm, meta, 16));
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &chars, d_jroff + nj, 8));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	if (hm[1]) return mcom_fail(ctx, MCOM_E_ARG, "member offset beyond %d key bits", key_bits);
	h_totals[0] = total; h_totals[1] = chars; h_totals[2] = hm[0];
	return MCOM_OK;
}

// ---- consensus tiles made on the device -----------------------------------------------------------------------------
#define MC_TILE 512
__global__ void k_tile_counts(const uint64_t *__restrict__ jroff, size_t nj, uint32_t *__restrict__ cnt, uint32_t *__restrict__ ucnt)
{
	const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j > nj) return;
	cnt[j] = j == nj ? 0u : (uint32_t)((jroff[j + 1] - jroff[j] + MC_TILE - 1) / MC_TILE);
	ucnt[j] = j == nj ? 0u : (uint32_t)((jroff[j + 1] - jroff[j] + 31) / 32);
}
// tjob / tidx: job and index inside the job of every tile (or NULL, NULL: ujob only, for the units of 32 columns)
__global__ void k_tile_fill(const uint32_t *__restrict__ toff, size_t nj, uint32_t *__restrict__ tjob, uint32_t *__restrict__ tidx)
{
	const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= nj) return;
	const uint32_t a = toff[j], b = toff[j + 1];
	for (uint32_t t = a; t < b; ++t) { tjob[t] = (uint32_t)j; if (tidx) tidx[t] = t - a; }
}

// Outside the overlap of its two parents a merged contig's columns see exactly the members one parent had, so the
// majority there is the parent's consensus character: only the overlap [shift, min(len_first, shift + len_second)) is
// counted again (k_merge_consensus over that column range), the rest is copied from the parents' strings.
__global__ void k_job_regions(const Job *__restrict__ jobs, size_t nj, const uint64_t *__restrict__ soff, uint32_t *__restrict__ olo, uint32_t *__restrict__ ohi,
                              uint32_t *__restrict__ tcnt, uint32_t *__restrict__ ucnt)
{
	const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j > nj) return;
	if (j == nj) { tcnt[j] = 0; ucnt[j] = 0; return; }
	const Job J = jobs[j];
	const bool afirst = J.pos_ori >= J.pos;
	const uint32_t f = afirst ? J.ci : J.cj, s = afirst ? J.cj : J.ci;
	const uint64_t sh = afirst ? J.pos_ori - J.pos : J.pos - J.pos_ori;
	const uint64_t lf = soff[f + 1] - soff[f], ls = soff[s + 1] - soff[s];
	const uint64_t lo = sh < lf ? sh : lf, hi = lf < sh + ls ? lf : sh + ls;
	olo[j] = (uint32_t)lo; ohi[j] = (uint32_t)(hi > lo ? hi : lo);
	tcnt[j] = (uint32_t)((ohi[j] - olo[j] + MC_TILE - 1) / MC_TILE);
	ucnt[j] = (uint32_t)((ohi[j] - olo[j] + 31) / 32);
}
__global__ __launch_bounds__(256) void k_merge_copy(const Job *__restrict__ jobs, size_t nj, const uint8_t *__restrict__ seq, const uint64_t *__restrict__ soff,
                                                    const uint64_t *__restrict__ jroff, const uint32_t *__restrict__ olo, const uint32_t *__restrict__ ohi,
                                                    uint8_t *__restrict__ refs)
{
	const size_t j = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;     // sixteen lanes per job (see k_keep_copy)
	if (j >= nj) return;
	const int lane = threadIdx.x & 15;
	const Job J = jobs[j];
	const bool afirst = J.pos_ori >= J.pos;
	const uint32_t f = afirst ? J.ci : J.cj, s = afirst ? J.cj : J.ci;
	const uint64_t sh = afirst ? J.pos_ori - J.pos : J.pos - J.pos_ori;
	const uint64_t lf = soff[f + 1] - soff[f];
	const uint8_t *sf = seq + soff[f], *ss = seq + soff[s];
	const uint64_t len = jroff[j + 1] - jroff[j];
	uint8_t *out = refs + jroff[j];
	const uint64_t lo = olo[j], hi = ohi[j];
	// eight characters per lane and step (unaligned 8-byte accesses), the last few of a stretch one by one
	auto copy8 = [&](uint8_t *dst, const uint8_t *src, uint64_t cnt) {
		const uint64_t n8 = cnt >> 3;
		for (uint64_t t = lane; t < n8; t += 16) { uint64_t v; __builtin_memcpy(&v, src + 8 * t, 8); __builtin_memcpy(dst + 8 * t, &v, 8); }
		for (uint64_t t = (n8 << 3) + lane; t < cnt; t += 16) dst[t] = src[t];
	};
	copy8(out, sf, lo);                                                      // before the overlap: the first parent alone
	if (hi < len) {                                                          // behind it: whichever parent reaches there
		if (hi < lf) copy8(out + hi, sf + hi, len - hi);                      // (the second one ends inside the first: len = lf)
		else copy8(out + hi, ss + (hi - sh), len - hi);
	}
}

extern "C" int mcom_merge_consensus_jobs(mcom_ctx *ctx, const uint64_t *d_packed, const uint64_t *d_jm, const uint64_t *d_jmoff,
                                         const uint64_t *d_jroff, size_t nj, uint64_t total_chars, int L, uint8_t *d_refs,
                                         const uint32_t *d_jobs, const uint8_t *d_seq, const uint64_t *d_soff)
{
	if (!ctx) return MCOM_E_ARG;
	if (nj == 0) return MCOM_OK;
	if (!d_packed || !d_jm || !d_jmoff || !d_jroff || !d_refs) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	const bool regions = d_jobs && d_seq && d_soff;                          // parents known: count the overlaps only
	const size_t max_tiles = (size_t)(total_chars / MC_TILE) + nj + 1, max_units = (size_t)(total_chars / 32) + nj + 1;
	if (max_units >= (1ull << 32)) return mcom_fail(ctx, MCOM_E_ARG, "too many consensus tiles");
	int rc = mcom_ws_reserve(ctx, 4 * al256((nj + 1) * 4) + al256(mcom_scan_scratch_elems(nj + 1) * 4 + 1024) + 4 * al256((max_tiles + 1) * 4) + al256(max_units * 4) + 256);
	if (rc) return rc;
	WsCut w{(char*)ctx->ws, 0};
	uint32_t *toff = w.take<uint32_t>(nj + 1), *uoff = w.take<uint32_t>(nj + 1), *olo = w.take<uint32_t>(nj + 1), *ohi = w.take<uint32_t>(nj + 1);
	uint32_t *scr = w.take<uint32_t>(mcom_scan_scratch_elems(nj + 1) + 256);
	uint32_t *tjob = w.take<uint32_t>(max_tiles + 1), *tidx = w.take<uint32_t>(max_tiles + 1), *tflag = w.take<uint32_t>(max_tiles + 1), *tlist = w.take<uint32_t>(max_tiles + 1);
	uint32_t *ujob = w.take<uint32_t>(max_units);
	const unsigned jblocks = (unsigned)((nj + 1 + 255) / 256);
	if (regions) MCOM_LAUNCH(k_job_regions, dim3(jblocks), dim3(256), 0, ctx->stream, (const Job*)d_jobs, nj, d_soff, olo, ohi, toff, uoff);
	else MCOM_LAUNCH(k_tile_counts, dim3(jblocks), dim3(256), 0, ctx->stream, d_jroff, nj, toff, uoff);
	MCOM_LAUNCH_CHECK(ctx);
	if ((rc = mcom_scan_u32(ctx, toff, toff, nj + 1, scr))) return rc;
	if ((rc = mcom_scan_u32(ctx, uoff, uoff, nj + 1, scr))) return rc;
	uint32_t nt = 0, nu = 0;
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &nt, toff + nj, 4));
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &nu, uoff + nj, 4));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	if (nt > max_tiles || nu > max_units) return mcom_fail(ctx, MCOM_E_ARG, "tile count %u above its bound", nt);
	if (nu) {
		// units of 32 columns through the bit-sliced kernel; the tiles it hands back (a unit that more than 127 members reach) through
		// the wave-per-tile kernel
		MCOM_LAUNCH(k_tile_fill, dim3(jblocks), dim3(256), 0, ctx->stream, uoff, nj, ujob, (uint32_t*)nullptr);
		MCOM_LAUNCH(k_tile_fill, dim3(jblocks), dim3(256), 0, ctx->stream, toff, nj, tjob, tidx);
		MCOM_LAUNCH_CHECK(ctx);
		uint32_t nlist = 0;
		if ((rc = mcom_merge_consensus_units(ctx, d_packed, d_jm, d_jmoff, d_jroff, ujob, uoff, nu, L, d_refs, regions ? olo : nullptr, regions ? ohi : nullptr,
		                                     toff, nt, tflag, tlist, &nlist))) return rc;
		if (nlist && (rc = mcom_merge_consensus_regions(ctx, d_packed, d_jm, d_jmoff, d_jroff, tjob, tidx, nlist, L, d_refs, regions ? olo : nullptr, regions ? ohi : nullptr, tlist)))
			return rc;
	}
	if (regions) {
		MCOM_LAUNCH(k_merge_copy, dim3((unsigned)((nj * 16 + 255) / 256)), dim3(256), 0, ctx->stream, (const Job*)d_jobs, nj, d_seq, d_soff, d_jroff, olo, ohi, d_refs);
		MCOM_LAUNCH_CHECK(ctx);
	}
	MCOM_HIP(ctx, mcom_stream_sync(ctx));                        // the workspace arrays are in use until here
	return MCOM_OK;
}

// ---- updateSingle on the device: the ids whose flag is zero, in order (preprocess.c:243-255) ----------------------------
__global__ void k_live_flags(const uint8_t *__restrict__ flag, size_t n, uint32_t *__restrict__ kf)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i <= n) kf[i] = (i < n && !flag[i]) ? 1u : 0u;
}
__global__ void k_live_scatter(const uint32_t *__restrict__ ids, const uint8_t *__restrict__ flag, const uint32_t *__restrict__ at, size_t n, uint32_t *__restrict__ out)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n && !flag[i]) out[at[i]] = ids[i];
}

extern "C" int mcom_compact_live(mcom_ctx *ctx, const uint32_t *d_ids, const uint8_t *d_flag, size_t n, uint32_t *d_out, uint64_t *h_n_out)
{
	if (!ctx || !h_n_out) return MCOM_E_ARG;
	*h_n_out = 0;
	if (n == 0) return MCOM_OK;
	if (!d_ids || !d_flag || !d_out) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	if (n >= (1ull << 32) - 1) return mcom_fail(ctx, MCOM_E_ARG, "too many ids");
	int rc = mcom_ws_reserve(ctx, al256((n + 1) * 4) + al256(mcom_scan_scratch_elems(n + 1) * 4 + 1024) + 256);
	if (rc) return rc;
	WsCut w{(char*)ctx->ws, 0};
	uint32_t *at = w.take<uint32_t>(n + 1);
	uint32_t *scr = w.take<uint32_t>(mcom_scan_scratch_elems(n + 1) + 256);
	MCOM_LAUNCH(k_live_flags, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, ctx->stream, d_flag, n, at);
	MCOM_LAUNCH_CHECK(ctx);
	if ((rc = mcom_scan_u32(ctx, at, at, n + 1, scr))) return rc;
	MCOM_LAUNCH(k_live_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_ids, d_flag, at, n, d_out);
	MCOM_LAUNCH_CHECK(ctx);
	uint32_t cnt = 0;
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &cnt, at + n, 4));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	*h_n_out = cnt;
	return MCOM_OK;
}

// the few entries whose flag is 1 or 2 (near-poly-A / -T singletons of a Stage-2 pass, bbhashdict.c:177-216): {index, id, flag}
__global__ void k_list_flagged(const uint32_t *__restrict__ ids, const uint8_t *__restrict__ flag, size_t n, uint32_t *__restrict__ out, uint32_t cap, uint32_t *__restrict__ count)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint8_t f = flag[i];
	if (f != 1 && f != 2) return;
	const uint32_t at = atomicAdd(count, 1u);
	if (at < cap) { out[3 * (size_t)at] = (uint32_t)i; out[3 * (size_t)at + 1] = ids[i]; out[3 * (size_t)at + 2] = f; }
}
extern "C" int mcom_list_flagged(mcom_ctx *ctx, const uint32_t *d_ids, const uint8_t *d_flag, size_t n, uint32_t *d_out, uint32_t cap, uint32_t *d_count)
{
	if (!ctx) return MCOM_E_ARG;
	if (!d_count || (cap && !d_out)) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	if (n == 0) { MCOM_HIP(ctx, hipMemsetAsync(d_count, 0, 4, ctx->stream)); return MCOM_OK; }
	if (!d_ids || !d_flag) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	MCOM_HIP(ctx, hipMemsetAsync(d_count, 0, 4, ctx->stream));
	MCOM_LAUNCH(k_list_flagged, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_ids, d_flag, n, d_out, cap, d_count);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

// ---- untouched contigs ------------------------------------------------------------------------------------------------
__global__ void k_keep_flags(const uint8_t *__restrict__ flag, size_t n, uint32_t *__restrict__ kf)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i <= n) kf[i] = (i < n && !flag[i]) ? 1u : 0u;
}
__global__ void k_keep_index(const uint8_t *__restrict__ flag, const uint32_t *__restrict__ kpos, size_t n, uint32_t *__restrict__ keepidx)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n && !flag[i]) keepidx[kpos[i]] = (uint32_t)i;
}
__global__ void k_keep_sizes(const uint32_t *__restrict__ keepidx, size_t nkeep, const uint64_t *__restrict__ soff, const uint64_t *__restrict__ moff,
                             uint64_t *__restrict__ ss, uint64_t *__restrict__ ms)
{
	const size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (u > nkeep) return;
	if (u == nkeep) { ss[u] = 0; ms[u] = 0; return; }
	const uint32_t i = keepidx[u];
	ss[u] = soff[i + 1] - soff[i]; ms[u] = moff[i + 1] - moff[i];
}
// entries nj+1 .. nj+nkeep of the new offset arrays (entry nj, the end of the merged part, is already there)
__global__ void k_keep_offsets(const uint64_t *__restrict__ ss, const uint64_t *__restrict__ ms, size_t nkeep, size_t nj,
                               uint64_t *__restrict__ soff2, uint64_t *__restrict__ moff2)
{
	const size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
	if (u > nkeep) return;
	soff2[nj + u] = soff2[nj] + ss[u]; moff2[nj + u] = moff2[nj] + ms[u];
}
__global__ __launch_bounds__(256) void k_keep_copy(const uint32_t *__restrict__ keepidx, size_t nkeep, size_t nj, const uint8_t *__restrict__ seq,
                                                   const uint64_t *__restrict__ soff, const uint64_t *__restrict__ mem, const uint64_t *__restrict__ moff,
                                                   uint8_t *__restrict__ seq2, const uint64_t *__restrict__ soff2, uint64_t *__restrict__ mem2,
                                                   const uint64_t *__restrict__ moff2)
{
	// sixteen lanes per contig, four contigs per wave: the kernel waits for its dependent loads (index, offsets, then the data:
	// PMC 93 % dependency wait), and a contig is a few hundred bytes -- four of them in flight per wave instead of one
	const size_t u = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
	if (u >= nkeep) return;
	const int lane = threadIdx.x & 15;
	const uint32_t i = keepidx[u];
	const uint64_t s0 = soff[i], sl = soff[i + 1] - s0, d0 = soff2[nj + u];
	const uint64_t m0 = moff[i], ml = moff[i + 1] - m0, e0 = moff2[nj + u];
	// eight characters per lane and step (unaligned 8-byte accesses are fine on this hardware), the last few one by one
	const uint64_t n8 = sl >> 3;
	for (uint64_t t = lane; t < n8; t += 16) { uint64_t v; __builtin_memcpy(&v, seq + s0 + 8 * t, 8); __builtin_memcpy(seq2 + d0 + 8 * t, &v, 8); }
	for (uint64_t t = (n8 << 3) + lane; t < sl; t += 16) seq2[d0 + t] = seq[s0 + t];
	for (uint64_t t = lane; t < ml; t += 16) mem2[e0 + t] = mem[m0 + t];
}

extern "C" int mcom_contigs_carry(mcom_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_soff, const uint64_t *d_mem, const uint64_t *d_moff, size_t n,
                                  const uint8_t *d_flag, size_t nj, size_t nkeep, uint8_t *d_seq2, uint64_t *d_soff2, uint64_t *d_mem2,
                                  uint64_t *d_moff2, uint32_t *d_keepidx, uint64_t *h_totals)
{
	if (!ctx || !h_totals) return MCOM_E_ARG;
	h_totals[0] = h_totals[1] = 0;
	if (!d_soff2 || !d_moff2) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	if (nkeep) {
		if (!d_seq || !d_soff || !d_mem || !d_moff || !d_flag || !d_seq2 || !d_mem2 || !d_keepidx || nkeep > n) return mcom_fail(ctx, MCOM_E_ARG, "bad carry arguments");
		int rc = mcom_ws_reserve(ctx, 2 * al256((n + 1) * 4) + al256(mcom_scan_scratch_elems(n + 1) * 4 + 1024) + 2 * al256((nkeep + 1) * 8) +
		                                  al256(scan64_scratch_elems(nkeep + 1) * 8) + 256);
		if (rc) return rc;
		WsCut w{(char*)ctx->ws, 0};
		uint32_t *kf = w.take<uint32_t>(n + 1);
		uint32_t *scr = w.take<uint32_t>(mcom_scan_scratch_elems(n + 1) + 256);
		uint64_t *ss = w.take<uint64_t>(nkeep + 1), *ms = w.take<uint64_t>(nkeep + 1);
		uint64_t *scr64 = w.take<uint64_t>(scan64_scratch_elems(nkeep + 1));
		const unsigned nb = (unsigned)((n + 1 + 255) / 256), kb = (unsigned)((nkeep + 1 + 255) / 256);
		MCOM_LAUNCH(k_keep_flags, dim3(nb), dim3(256), 0, ctx->stream, d_flag, n, kf);
		MCOM_LAUNCH_CHECK(ctx);
		if ((rc = mcom_scan_u32(ctx, kf, kf, n + 1, scr))) return rc;
		uint32_t have = 0;
		MCOM_HIP(ctx, mcom_d2h_async(ctx, &have, kf + n, 4));
		MCOM_HIP(ctx, mcom_stream_sync(ctx));
		if (have != nkeep) return mcom_fail(ctx, MCOM_E_ARG, "%u contigs unflagged but %zu announced", have, nkeep);
		MCOM_LAUNCH(k_keep_index, dim3(nb), dim3(256), 0, ctx->stream, d_flag, kf, n, d_keepidx);
		MCOM_LAUNCH(k_keep_sizes, dim3(kb), dim3(256), 0, ctx->stream, d_keepidx, nkeep, d_soff, d_moff, ss, ms);
		MCOM_LAUNCH_CHECK(ctx);
		if ((rc = scan64(ctx, ss, ss, nkeep + 1, scr64)) || (rc = scan64(ctx, ms, ms, nkeep + 1, scr64))) return rc;
		MCOM_LAUNCH(k_keep_offsets, dim3(kb), dim3(256), 0, ctx->stream, ss, ms, nkeep, nj, d_soff2, d_moff2);
		MCOM_LAUNCH(k_keep_copy, dim3((unsigned)((nkeep * 16 + 255) / 256)), dim3(256), 0, ctx->stream, d_keepidx, nkeep, nj, d_seq, d_soff, d_mem, d_moff,
		                   d_seq2, d_soff2, d_mem2, d_moff2);
		MCOM_LAUNCH_CHECK(ctx);
	}
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &h_totals[0], d_soff2 + nj + nkeep, 8));
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &h_totals[1], d_moff2 + nj + nkeep, 8));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	return MCOM_OK;
}


// the contigs d_idx[0 .. n_idx) of a set, in that order, as a set of their own (strings, member lists, both offset arrays from 0): what
// the store of the merge rounds is turned into when the rounds are over -- the one copy that is left of cp_cluster's
extern "C" int mcom_contigs_gather(mcom_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_soff, const uint64_t *d_mem, const uint64_t *d_moff,
                                   const uint32_t *d_idx, size_t n_idx, uint8_t *d_seq2, uint64_t *d_soff2, uint64_t *d_mem2, uint64_t *d_moff2, uint64_t *h_totals)
{
	if (!ctx || !h_totals) return MCOM_E_ARG;
	h_totals[0] = h_totals[1] = 0;
	if (!d_soff2 || !d_moff2) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	MCOM_HIP(ctx, hipMemsetAsync(d_soff2, 0, 8, ctx->stream));
	MCOM_HIP(ctx, hipMemsetAsync(d_moff2, 0, 8, ctx->stream));
	if (n_idx == 0) return MCOM_OK;
	if (!d_seq || !d_soff || !d_mem || !d_moff || !d_idx || !d_seq2 || !d_mem2) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	int rc = mcom_ws_reserve(ctx, 2 * al256((n_idx + 1) * 8) + al256(scan64_scratch_elems(n_idx + 1) * 8) + 256);
	if (rc) return rc;
	WsCut w{(char*)ctx->ws, 0};
	uint64_t *ss = w.take<uint64_t>(n_idx + 1), *ms = w.take<uint64_t>(n_idx + 1);
	uint64_t *scr64 = w.take<uint64_t>(scan64_scratch_elems(n_idx + 1));
	const unsigned kb = (unsigned)((n_idx + 1 + 255) / 256);
	MCOM_LAUNCH(k_keep_sizes, dim3(kb), dim3(256), 0, ctx->stream, d_idx, n_idx, d_soff, d_moff, ss, ms);
	MCOM_LAUNCH_CHECK(ctx);
	if ((rc = scan64(ctx, ss, ss, n_idx + 1, scr64)) || (rc = scan64(ctx, ms, ms, n_idx + 1, scr64))) return rc;
	MCOM_LAUNCH(k_keep_offsets, dim3(kb), dim3(256), 0, ctx->stream, ss, ms, n_idx, (size_t)0, d_soff2, d_moff2);
	MCOM_LAUNCH(k_keep_copy, dim3((unsigned)((n_idx * 16 + 255) / 256)), dim3(256), 0, ctx->stream, d_idx, n_idx, (size_t)0, d_seq, d_soff, d_mem, d_moff,
	                   d_seq2, d_soff2, d_mem2, d_moff2);
	MCOM_LAUNCH_CHECK(ctx);
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &h_totals[0], d_soff2 + n_idx, 8));
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &h_totals[1], d_moff2 + n_idx, 8));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	return MCOM_OK;
}

// ---- merge rounds that leave the set where it is (round 5) -------------------------------------------------------------------------
// cp_cluster (kthread_cb.c:397-434) copies every unmerged contig into the other buffer each round; the order of the new list --
// merged contigs in claiming order, then the others in their order -- is all that later code needs from that.  Here the contig set is
// an append-only store (strings, member lists, minimizer records, packed words: offset arrays with one more entry per contig ever
// made, a contig's index never changes) and the list is an array of indices into it (`ord`).  A round appends its merged contigs and
// makes the next list; nothing is copied but the list.
__global__ void k_off_append64(const uint64_t *__restrict__ rel, size_t n, uint64_t base, uint64_t *__restrict__ dst)
{
	const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j <= n) dst[j] = base + rel[j];
}
__global__ void k_off_append32(const uint32_t *__restrict__ rel, size_t n, uint32_t base, uint32_t *__restrict__ dst)
{
	const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j <= n) dst[j] = base + rel[j];
}
extern "C" int mcom_offsets_append(mcom_ctx *ctx, const uint64_t *d_rel, size_t n, uint64_t base, uint64_t *d_dst)
{
	if (!ctx) return MCOM_E_ARG;
	if (!d_rel || !d_dst) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	MCOM_LAUNCH(k_off_append64, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, ctx->stream, d_rel, n, base, d_dst);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}
extern "C" int mcom_offsets_append_u32(mcom_ctx *ctx, const uint32_t *d_rel, size_t n, uint32_t base, uint32_t *d_dst)
{
	if (!ctx) return MCOM_E_ARG;
	if (!d_rel || !d_dst) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	MCOM_LAUNCH(k_off_append32, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, ctx->stream, d_rel, n, base, d_dst);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}
__global__ void k_ord_flags(const uint32_t *__restrict__ ord, const uint8_t *__restrict__ flag, size_t n, uint32_t *__restrict__ kf)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i <= n) kf[i] = (i < n && !flag[ord ? ord[i] : (uint32_t)i]) ? 1u : 0u;
}
__global__ void k_ord_next(const uint32_t *__restrict__ ord, const uint8_t *__restrict__ flag, const uint32_t *__restrict__ kpos, size_t n, uint32_t first_new, size_t nj,
                           uint32_t *__restrict__ ord2)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < nj) ord2[i] = first_new + (uint32_t)i;
	if (i < n) { const uint32_t c = ord ? ord[i] : (uint32_t)i; if (!flag[c]) ord2[nj + kpos[i]] = c; }
}
extern "C" int mcom_order_next(mcom_ctx *ctx, const uint32_t *d_ord, size_t n, const uint8_t *d_flag, uint32_t first_new, size_t nj, uint32_t *d_ord2, uint64_t *h_nkeep)
{
	if (!ctx || !h_nkeep) return MCOM_E_ARG;
	*h_nkeep = 0;
	if (n + nj == 0) return MCOM_OK;
	if (!d_ord2 || (n && !d_flag)) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	if (n + nj >= (1ull << 32) - 1 || (uint64_t)first_new + nj >= (1ull << 32) - 1) return mcom_fail(ctx, MCOM_E_ARG, "more than 2^32 - 2 contigs");
	uint32_t *kf = nullptr;
	if (mcom_dmalloc(&kf, (n + 1) * 4) != hipSuccess) return mcom_fail(ctx, MCOM_E_NOMEM, "order scratch");
	struct G { mcom_ctx *c; uint32_t *p; ~G() { (void)hipStreamSynchronize(c->stream); mcom_dfree(p); } } g{ctx, kf};
	const size_t m = n > nj ? n : nj;
	MCOM_LAUNCH(k_ord_flags, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, ctx->stream, d_ord, d_flag, n, kf);
	MCOM_LAUNCH_CHECK(ctx);
	int rc = mcom_scan_u32(ctx, kf, kf, n + 1, nullptr);
	if (rc) return rc;
	uint32_t keep = 0;
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &keep, kf + n, 4));
	MCOM_LAUNCH(k_ord_next, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, d_ord, d_flag, (const uint32_t*)kf, n, first_new, nj, d_ord2);
	MCOM_LAUNCH_CHECK(ctx);
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	*h_nkeep = keep;
	return MCOM_OK;
}

// ---- minimizers of untouched contigs: same records, new contig id ---------------------------------------------------
__global__ void k_carry_counts(const uint32_t *__restrict__ keepidx, size_t nkeep, const uint32_t *__restrict__ roff, uint32_t *__restrict__ cnt)
{
	const size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (u > nkeep) return;
	if (u == nkeep) { cnt[u] = 0; return; }
	const uint32_t i = keepidx[u];
	cnt[u] = roff[i + 1] - roff[i];
}
__global__ __launch_bounds__(256) void k_carry_copy(const uint32_t *__restrict__ keepidx, size_t nkeep, const mcom_mm128 *__restrict__ rec,
                                                    const uint32_t *__restrict__ roff, const uint32_t *__restrict__ sc, uint32_t first_id, uint32_t base,
                                                    mcom_mm128 *__restrict__ rec2, uint32_t *__restrict__ roff2)
{
	const size_t u = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;   // sixteen lanes per contig (see k_keep_copy)
	if (u > nkeep) return;
	const int lane = threadIdx.x & 15;
	if (lane == 0) roff2[first_id + u] = base + sc[u];
	if (u == nkeep) return;
	const uint32_t i = keepidx[u];
	const uint32_t r0 = roff[i], cnt = roff[i + 1] - r0, d0 = base + sc[u];
	const uint64_t id = (uint64_t)(first_id + (uint32_t)u) << 32;            // the contig index (include/mcom.h, "contig ids")
	for (uint32_t t = lane; t < cnt; t += 16) { mcom_mm128 r = rec[r0 + t]; r.y = id | (r.y & 0xFFFFFFFFull); rec2[d0 + t] = r; }
}

extern "C" int mcom_records_carry(mcom_ctx *ctx, const mcom_mm128 *d_rec, const uint32_t *d_roff, const uint32_t *d_keepidx, size_t nkeep,
                                  uint32_t first_id, uint32_t base, mcom_mm128 *d_rec2, size_t cap2, uint32_t *d_roff2, uint64_t *h_total)
{
	if (!ctx || !h_total) return MCOM_E_ARG;
	*h_total = base;
	if (!d_roff2) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	if ((uint64_t)first_id + nkeep >= (1ull << 32) - 1) return mcom_fail(ctx, MCOM_E_ARG, "more than 2^32 - 2 contigs: record ids overflow");
	if (nkeep == 0) { MCOM_HIP(ctx, hipMemcpyAsync(d_roff2 + first_id, &base, 4, hipMemcpyHostToDevice, ctx->stream)); MCOM_HIP(ctx, mcom_stream_sync(ctx)); return MCOM_OK; }
	if (!d_rec || !d_roff || !d_keepidx || !d_rec2) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	int rc = mcom_ws_reserve(ctx, al256((nkeep + 1) * 4) + al256(mcom_scan_scratch_elems(nkeep + 1) * 4 + 1024) + 256);
	if (rc) return rc;
	WsCut w{(char*)ctx->ws, 0};
	uint32_t *sc = w.take<uint32_t>(nkeep + 1);
	uint32_t *scr = w.take<uint32_t>(mcom_scan_scratch_elems(nkeep + 1) + 256);
	MCOM_LAUNCH(k_carry_counts, dim3((unsigned)((nkeep + 1 + 255) / 256)), dim3(256), 0, ctx->stream, d_keepidx, nkeep, d_roff, sc);
	MCOM_LAUNCH_CHECK(ctx);
	if ((rc = mcom_scan_u32(ctx, sc, sc, nkeep + 1, scr))) return rc;
	uint32_t kept = 0;
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &kept, sc + nkeep, 4));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	*h_total = (uint64_t)base + kept;
	if ((uint64_t)base + kept > cap2) return mcom_fail(ctx, MCOM_E_OVERFLOW, "%llu minimizers but room for %zu", (unsigned long long)base + kept, cap2);
	MCOM_LAUNCH(k_carry_copy, dim3((unsigned)(((nkeep + 1) * 16 + 255) / 256)), dim3(256), 0, ctx->stream, d_keepidx, nkeep, d_rec, d_roff, sc, first_id, base,
	                   d_rec2, d_roff2);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}
