// minicom_amd/csrc/realign.hip -- Stage-2 realignment kernels for gfx950 (MI355X).
//
// Replaces realign_hash (reference kthread_hash_realign.c:569): singleRead2bitset (bbhashdict.c:127-227),
// constructdictionary_realign (kthread_hash_realign.c:3-140, with BooPHF.h's MPHF) and the window scan
// realign_hash_search (kthread_hash_realign.c:316-508).
//
//  * Dictionaries: for every dictionary j the 2*len_j-bit key of each singleton is sorted together with its
//    singleton index (stable, so a bin lists its reads in ascending index like read_id[] does), the runs of
//    equal keys become bins, and an open-addressing hash table in HBM maps key -> (bin start, bin size).
//    The reference's MPHF is only a key -> bin id map whose every hit is re-verified against the bin's key
//    (kthread_hash_realign.c:385-386), so any exact map gives the same results.
//  * Window scan: one thread per contig window.  The reference claims a read at the first window (contig
//    order, window order, forward before reverse, dictionary order) whose test it passes, then deletes it
//    from every dictionary.  The tests themselves do not depend on that state, so the claim of a read is
//    the MINIMUM over all passing (contig, window, dir, dict) tuples: one 64-bit atomicMin per passing
//    candidate, no locks, same result as the sequential scan (exception: bins larger than maxsearch, see
//    DESIGN.md).  Random-access HBM-bound.
#include "cindex.hpp"
#include <cstring>

struct mcom_dicts {
	int L, W, nd;
	int ds[MAXDICT], kl[MAXDICT];           // first base and length (bases) of every key
	size_t n_sg;
	uint64_t *slots[MAXDICT];               // hash tables: pairs {key, start | count << 32}, EMPTY key = ~0
	uint32_t log2cap[MAXDICT];
	uint32_t *ids[MAXDICT];                 // singleton indices, bin after bin, ascending inside a bin
	uint32_t numkeys[MAXDICT], maxbin[MAXDICT];
};

struct DictDev {
	int nd, W, L;
	int ds[MAXDICT], kl[MAXDICT];
	const uint64_t *slots[MAXDICT];
	uint32_t log2cap[MAXDICT];
	const uint32_t *ids[MAXDICT];
};

extern "C" int mcom_dict_layout(int L, int ininumdict, int *start, int *end)
{
	if (L < 1 || L > 256 || !start || !end) return MCOM_E_ARG;
	int len[MAXDICT], st[MAXDICT];
	const int nd = dict_layout(L, ininumdict, st, len);
	for (int i = 0; i < nd; ++i) { start[i] = st[i]; end[i] = st[i] + len[i] - 1; }
	return nd;
}

// bits [2*start, 2*start + 2*len) of a packed row
__device__ __forceinline__ uint64_t bits_key(const uint64_t *w, int start, int len)
{
	const int bo = 2 * start, wi = bo >> 6, sh = bo & 63;
	uint64_t v = w[wi] >> sh;
	if (sh && sh + 2 * len > 64) v |= w[wi + 1] << (64 - sh);
	return v & ((1ull << (2 * len)) - 1);
}

// ---- gather packed rows of the singletons ------------------------------------------------------------
__global__ void k_gather_rows(const uint64_t *__restrict__ packed, const uint32_t *__restrict__ rids, size_t n, int W,
                              uint64_t *__restrict__ out)
{
	const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n * (size_t)W) return;
	const size_t i = t / W; const int q = (int)(t - i * W);
	out[t] = packed[(size_t)rids[i] * W + q];
}

extern "C" int mcom_gather_rows(mcom_ctx *ctx, const uint64_t *d_packed, const uint32_t *d_rids, size_t n, int L, uint64_t *d_out)
{
	if (!ctx) return MCOM_E_ARG;
	if (n == 0) return MCOM_OK;
	if (!d_packed || !d_rids || !d_out || L < 1 || L > 256) return mcom_fail(ctx, MCOM_E_ARG, "bad gather arguments");
	const int W = mcom_words_per_read(L);
	const size_t tot = n * (size_t)W;
	MCOM_LAUNCH(k_gather_rows, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, d_packed, d_rids, n, W, d_out);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

// ---- near-poly-A / poly-T filter of singleRead2bitset (bbhashdict.c:157-222) -----------------------
// flag 1: goes to the A list, 2: to the T list, 0: stays.  The 2-bit distance is the popcount against
// all-A (zero) / all-T (ones); the text length is taken on the read with its N restored.
__device__ __forceinline__ int ndigits_dev(int v) { return v >= 100 ? 3 : (v >= 10 ? 2 : 1); }

__global__ void k_poly_filter(const uint64_t *__restrict__ bits, const uint64_t *__restrict__ nmask, const uint32_t *__restrict__ rids,
                              size_t n, int L, int W, int NW, int thr, uint8_t *__restrict__ flag)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint64_t *b = bits + i * (size_t)W;
	int dA = 0, dT = 0;
	for (int q = 0; q < W; ++q) {
		const int nb = (2 * L - 64 * q) < 64 ? (2 * L - 64 * q) : 64;
		const uint64_t m = nb >= 64 ? ~0ull : ((1ull << nb) - 1);
		dA += __popcll(b[q]); dT += __popcll((~b[q]) & m);
	}
	uint8_t f = 0;
	const bool nearA = dA <= thr, nearT = !nearA && dT <= thr;
	if (nearA || nearT) {
		const uint32_t want = nearA ? 0u : 3u;
		const uint64_t *nm = nmask ? nmask + (size_t)rids[i] * NW : nullptr;
		int len = 0, eq = 0;
		for (int t = 0; t < L; ++t) {
			const uint32_t c = (uint32_t)(b[t >> 5] >> (2 * (t & 31))) & 3u;
			const bool isn = nm && ((nm[t >> 6] >> (t & 63)) & 1);
			if (isn || c != want) { if (eq > 0) { len += ndigits_dev(eq); eq = 0; } ++len; }
			else ++eq;
		}
		if (len == 0) len = 1;
		if ((double)len <= (double)L * 0.4) f = nearA ? 1 : 2;
	}
	flag[i] = f;
}

extern "C" int mcom_poly_filter(mcom_ctx *ctx, const uint64_t *d_sgbits, const uint64_t *d_nmask, const uint32_t *d_rids,
                                size_t n_sg, int L, int thr, uint8_t *d_flag)
{
	if (!ctx) return MCOM_E_ARG;
	if (n_sg == 0) return MCOM_OK;
	if (!d_sgbits || !d_flag || L < 1 || L > 256 || (d_nmask && !d_rids)) return mcom_fail(ctx, MCOM_E_ARG, "bad poly filter arguments");
	MCOM_LAUNCH(k_poly_filter, dim3((unsigned)((n_sg + 255) / 256)), dim3(256), 0, ctx->stream, d_sgbits, d_nmask, d_rids, n_sg, L,
	                   mcom_words_per_read(L), (L + 63) / 64, thr, d_flag);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

// ---- dictionary build ------------------------------------------------------------------------------------
__global__ void k_dict_keys(const uint64_t *__restrict__ bits, size_t n, int W, int start, int len, mcom_mm128 *__restrict__ rec)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	mcom_mm128 r; r.x = bits_key(bits + i * (size_t)W, start, len); r.y = i;
	rec[i] = r;
}
__global__ void k_dict_ids(const mcom_mm128 *__restrict__ s, size_t n, uint32_t *__restrict__ ids)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) ids[i] = (uint32_t)s[i].y;
}

extern "C" void mcom_dicts_free(mcom_ctx *ctx, mcom_dicts *d)
{
	if (!d) return;
	if (ctx) (void)mcom_stream_sync(ctx);
	for (int j = 0; j < MAXDICT; ++j) { if (d->slots[j]) mcom_dfree(d->slots[j]); if (d->ids[j]) mcom_dfree(d->ids[j]); }
	delete d;
}

extern "C" int mcom_dicts_build(mcom_ctx *ctx, const uint64_t *d_sgbits, size_t n_sg, int L, int ininumdict, mcom_dicts **out)
{
	if (!ctx || !out) return MCOM_E_ARG;
	*out = nullptr;
	if (L < 1 || L > 256) return mcom_fail(ctx, MCOM_E_ARG, "read length %d out of range", L);
	if (n_sg && !d_sgbits) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	if (n_sg >= (1ull << 31)) return mcom_fail(ctx, MCOM_E_ARG, "too many singletons");
	mcom_dicts *d = new mcom_dicts();
	std::memset(d, 0, sizeof *d);
	d->L = L; d->W = mcom_words_per_read(L); d->n_sg = n_sg;
	d->nd = dict_layout(L, ininumdict, d->ds, d->kl);
	const size_t n = n_sg;
	// workspace: records + sort workspace + heads + scan scratch + meta
	const size_t rec_b = ((n * sizeof(mcom_mm128)) + 255) & ~(size_t)255;
	const size_t sort_b = mcom_sort_ws_bytes(n);
	const size_t head_b = ((n * 4) + 255) & ~(size_t)255;
	const size_t scr_b = ((mcom_scan_scratch_elems(n) * 4 + 1024) + 255) & ~(size_t)255;
	int rc = mcom_ws_reserve(ctx, rec_b + sort_b + head_b + scr_b + 256);
	if (rc) { delete d; return rc; }
	char *base = (char*)ctx->ws;
	mcom_mm128 *rec = (mcom_mm128*)base;
	void *sortws = base + rec_b;
	uint32_t *head = (uint32_t*)(base + rec_b + sort_b);
	uint32_t *scr = (uint32_t*)(base + rec_b + sort_b + head_b);
	uint32_t *meta = (uint32_t*)(base + rec_b + sort_b + head_b + scr_b);
	const unsigned blocks = (unsigned)((n + 255) / 256);
	for (int j = 0; j < d->nd; ++j) {
		McomProfScope ps_(ctx, PROF_DICT_BUILD);
		hipError_t e2 = mcom_dmalloc(&d->ids[j], (n ? n : 1) * 4);
		if (e2 != hipSuccess) { d->ids[j] = nullptr; mcom_dicts_free(ctx, d); return mcom_fail(ctx, MCOM_E_NOMEM, "dictionary %d: out of device memory", j); }
		if (n) {
			MCOM_LAUNCH(k_dict_keys, dim3(blocks), dim3(256), 0, ctx->stream, d_sgbits, n, d->W, d->ds[j], d->kl[j], rec);
			rc = mcom_sort_by_x(ctx, rec, n, 2 * d->kl[j], sortws);
			if (rc) { mcom_dicts_free(ctx, d); return rc; }
			MCOM_LAUNCH(k_dict_ids, dim3(blocks), dim3(256), 0, ctx->stream, rec, n, d->ids[j]);
		}
		McomTable t;
		rc = mcom_table_build(ctx, rec, n, head, scr, meta, &t);
		d->slots[j] = t.slots; d->log2cap[j] = t.log2cap; d->numkeys[j] = t.numkeys; d->maxbin[j] = t.maxrun;
		if (rc) { mcom_dicts_free(ctx, d); return rc; }
	}
	hipError_t e = mcom_stream_sync(ctx);
	if (e != hipSuccess) { mcom_dicts_free(ctx, d); return mcom_fail(ctx, MCOM_E_HIP, "dict build: %s", hipGetErrorString(e)); }
	*out = d;
	return MCOM_OK;
}

extern "C" int mcom_dicts_info(const mcom_dicts *d, int *nd, uint32_t *numkeys, uint32_t *maxbin)
{
	if (!d) return MCOM_E_ARG;
	if (nd) *nd = d->nd;
	for (int j = 0; j < d->nd; ++j) { if (numkeys) numkeys[j] = d->numkeys[j]; if (maxbin) maxbin[j] = d->maxbin[j]; }
	return MCOM_OK;
}

// ---- batched lookup (bphf->lookup + findpos on an untouched dictionary, bbhashdict.c:33-43) ----------
#define dict_find mcom_table_find
__global__ void k_dict_lookup(const uint64_t *__restrict__ slots, uint32_t log2cap, const uint64_t *__restrict__ keys, size_t n,
                              uint32_t *__restrict__ start, uint32_t *__restrict__ count)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint32_t s = 0, c = 0;
	dict_find(slots, log2cap, keys[i], s, c);
	start[i] = s; count[i] = c;
}
extern "C" int mcom_dicts_lookup(mcom_ctx *ctx, const mcom_dicts *d, int dict, const uint64_t *d_keys, size_t n,
                                 uint32_t *d_start, uint32_t *d_count)
{
	if (!ctx || !d) return MCOM_E_ARG;
	if (dict < 0 || dict >= d->nd) return mcom_fail(ctx, MCOM_E_ARG, "dictionary %d out of range", dict);
	if (n == 0) return MCOM_OK;
	MCOM_LAUNCH(k_dict_lookup, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d->slots[dict], d->log2cap[dict], d_keys, n, d_start, d_count);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}
extern "C" int mcom_dicts_ids(mcom_ctx *ctx, const mcom_dicts *d, int dict, uint32_t *d_ids_out)
{
	if (!ctx || !d) return MCOM_E_ARG;
	if (dict < 0 || dict >= d->nd) return mcom_fail(ctx, MCOM_E_ARG, "dictionary %d out of range", dict);
	if (d->n_sg) MCOM_HIP(ctx, hipMemcpyAsync(d_ids_out, d->ids[dict], d->n_sg * 4, hipMemcpyDeviceToDevice, ctx->stream));
	return MCOM_OK;
}

// ---- window scan ---------------------------------------------------------------------------------------------
// reverse the order of the 32 bases of a word (2-bit groups)
__device__ __forceinline__ uint64_t rev_groups(uint64_t x)
{
	x = __brevll(x);                                                     // reverses bits: groups reversed, bits inside a group swapped
	return ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
}

// encode_byte's length estimate from a per-base mismatch mask (kthread_hash_realign.c:283-314); rev: walk
// the window from its last base to its first (the reference reverse-complements the read instead)
template <int W>
__device__ __forceinline__ bool encode_ok(const uint64_t (&mm)[W], int L, bool rev)
{
	int len = 0, eq = 0;
	for (int t = 0; t < L; ++t) {
		const int p = rev ? L - 1 - t : t;
		const bool mis = (mm[p >> 5] >> (2 * (p & 31))) & 1;
		if (mis) { if (eq > 1) { len += ndigits_dev(eq); eq = 0; } else len += eq; ++len; }
		else ++eq;
	}
	if (len == 0) len = 1;
	return (double)len <= (double)L * 0.4;
}

template <int W>
__global__ __launch_bounds__(256) void k_realign_windows(DictDev dd, const uint64_t *__restrict__ sgbits, const uint8_t *__restrict__ sgflag,
                                                         const uint64_t *__restrict__ cbits, const uint64_t *__restrict__ coff,
                                                         const uint64_t *__restrict__ woff, uint32_t n_contigs, uint64_t n_windows,
                                                         int thr, int maxsearch, unsigned long long *__restrict__ claim,
                                                         unsigned long long *__restrict__ stats)
{
	const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= n_windows) return;
	// contig of this window: last c with woff[c] <= g
	uint32_t lo = 0, hi = n_contigs;
	while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (woff[mid] <= g) lo = mid; else hi = mid; }
	const uint32_t c = lo;
	const uint64_t jj = g - woff[c];
	const int L = dd.L;
	// window bits
	uint64_t win[W], rwin[W];
	{
		const uint64_t *src = cbits + coff[c] + ((2 * jj) >> 6);
		const int sh = (int)((2 * jj) & 63);
		uint64_t cur = src[0];
#pragma unroll
		for (int q = 0; q < W; ++q) {
			// the last source word may lie beyond the window's own bits but is inside the contig's padded storage
			const uint64_t nxt = src[q + 1];
			win[q] = sh ? (cur >> sh) | (nxt << (64 - sh)) : cur;
			cur = nxt;
		}
		const int tail = 2 * L - 64 * (W - 1);
		if (tail < 64) win[W - 1] &= (1ull << tail) - 1;
		// reverse complement: reverse all 32*W groups, complement, then drop the 32*W - L leading pad groups
		uint64_t t[W];
#pragma unroll
		for (int q = 0; q < W; ++q) t[q] = ~rev_groups(win[W - 1 - q]);
		const int drop = 64 * W - 2 * L;                                 // < 64
#pragma unroll
		for (int q = 0; q < W; ++q) {
			const uint64_t a = t[q], b = q + 1 < W ? t[q + 1] : 0ull;
			rwin[q] = drop ? (a >> drop) | (b << (64 - drop)) : a;
		}
		if (tail < 64) rwin[W - 1] &= (1ull << tail) - 1;
	}
	unsigned long long n_look = 0, n_hit = 0, n_cand = 0;
	for (int dir = 0; dir < 2; ++dir) {
		const uint64_t *q = dir ? rwin : win;
		for (int l = 0; l < dd.nd; ++l) {
			if (dir && dd.ds[l] <= 0) continue;                          // kthread_hash_realign.c:440
			++n_look;
			const uint64_t key = bits_key(q, dd.ds[l], dd.kl[l]);
			uint32_t start, count;
			if (!dict_find(dd.slots[l], dd.log2cap[l], key, start, count)) continue;
			++n_hit;
			const uint32_t nscan = count < (uint32_t)maxsearch ? count : (uint32_t)maxsearch;
			const unsigned long long ck = ((unsigned long long)c << 33) | ((unsigned long long)jj << 5) | ((unsigned long long)dir << 4) | (unsigned long long)l;
			for (uint32_t u = 0; u < nscan; ++u) {
				const uint32_t sg = dd.ids[l][start + count - 1 - u];     // from the bin's end, descending (:388)
				if (sgflag[sg]) continue;
				++n_cand;
				const uint64_t *rb = sgbits + (size_t)sg * W;
				uint64_t mm[W]; int dist = 0;
#pragma unroll
				for (int w = 0; w < W; ++w) { const uint64_t x = q[w] ^ rb[w]; dist += __popcll(x); mm[w] = (x | (x >> 1)) & 0x5555555555555555ull; }
				if (dist > thr) continue;
				if (!dir) { if (!encode_ok<W>(mm, L, false)) continue; }                         // :393
				else if (thr > 24 && !encode_ok<W>(mm, L, true)) continue;                      // :461
				atomicMin(&claim[sg], ck);
			}
		}
	}
	if (stats) {
		// wave-level reduction would be cheaper; these are diagnostics only
		atomicAdd(&stats[0], n_look); if (n_hit) atomicAdd(&stats[1], n_hit); if (n_cand) atomicAdd(&stats[2], n_cand);
	}
}

__global__ void k_window_counts(const uint32_t *__restrict__ clen, uint32_t n, int L, uint32_t *__restrict__ cnt)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) cnt[i] = clen[i] >= (uint32_t)L ? clen[i] - (uint32_t)L + 1 : 0u;
}

extern "C" int mcom_realign_pass(mcom_ctx *ctx, const mcom_dicts *d, const uint64_t *d_sgbits, const uint8_t *d_sgflag,
                                 const uint64_t *d_cbits, const uint64_t *d_coff, const uint64_t *d_woff, uint32_t n_contigs,
                                 uint64_t n_windows, int thr, int maxsearch, uint64_t *d_claim, uint64_t *d_stats)
{
	if (!ctx || !d) return MCOM_E_ARG;
	if (d->n_sg) MCOM_HIP(ctx, hipMemsetAsync(d_claim, 0xFF, d->n_sg * 8, ctx->stream));
	if (d_stats) MCOM_HIP(ctx, hipMemsetAsync(d_stats, 0, 3 * 8, ctx->stream));
	if (n_windows == 0 || n_contigs == 0 || d->n_sg == 0) return MCOM_OK;
	if (!d_sgbits || !d_sgflag || !d_cbits || !d_coff || !d_woff || !d_claim) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	if (n_contigs >= (1u << 31)) return mcom_fail(ctx, MCOM_E_ARG, "too many contigs for the claim key");
	DictDev dd; std::memset(&dd, 0, sizeof dd);
	dd.nd = d->nd; dd.W = d->W; dd.L = d->L;
	for (int j = 0; j < d->nd; ++j) { dd.ds[j] = d->ds[j]; dd.kl[j] = d->kl[j]; dd.slots[j] = d->slots[j]; dd.log2cap[j] = d->log2cap[j]; dd.ids[j] = d->ids[j]; }
	const uint64_t blocks = (n_windows + 255) / 256;
	if (blocks >= (1ull << 31)) return mcom_fail(ctx, MCOM_E_ARG, "too many windows for one launch");
#define MCOM_CASE(WW) case WW: MCOM_LAUNCH((k_realign_windows<WW>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, dd, d_sgbits, d_sgflag, d_cbits, d_coff, d_woff, n_contigs, n_windows, thr, maxsearch, (unsigned long long*)d_claim, (unsigned long long*)d_stats); break;
	McomProfScope ps_(ctx, PROF_REALIGN_WINDOWS);
	switch (d->W) { MCOM_CASE(1) MCOM_CASE(2) MCOM_CASE(3) MCOM_CASE(4) MCOM_CASE(5) MCOM_CASE(6) MCOM_CASE(7) MCOM_CASE(8)
	default: return mcom_fail(ctx, MCOM_E_ARG, "unsupported read length"); }
#undef MCOM_CASE
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

// ==== read-driven form of the same pass =======================================================================
// The window scan asks, for every (window, dir, dict), "which singletons carry this key?": ~15 random probes per
// contig base although only the singletons (a few % of the reads) can ever answer.  Because the claim of a read is
// a MINIMUM over its passing (contig, window, dir, dict) tuples, the join can be driven from the other side with the
// same result: index every klen-mer of the (fixed) Stage-2 contigs once, then let every singleton look its own
// 2*nd - 1 keys up (forward key of dictionary l at contig position jj + ds[l]; for the reverse direction the
// reverse complement of the key at jj + L - ds[l] - klen) and verify exactly the tuples the window scan would have
// verified.  Probes per pass drop from (2nd-1) * windows to (2nd-1) * singletons.
//
// Index: multi-map in HBM, lines of 8 words = one 64-B sector: word 0 counts the line's inserts (minus one), words
// 1-7 are slots; a slot is tag<<52 | contig<<28 | position with a 12-bit tag of the key: a hit is only a candidate, the verification re-checks the key bits exactly (the XOR of read
// and window is zero over the key's bases), so a tag collision costs one wasted verification and nothing else.
// A key's probe sequence is line h1, h1+s, h1+2s, ... with an odd key-dependent stride s (so a repeat with 10^5
// copies does not bury its neighbours the way linear probing would); an entry moves on to the next line of its
// sequence only when the line's counter says it is full, and counters only grow, hence every copy of a key lies in
// the lines of its sequence up to and including the first one that saw at most seven inserts.
// (geometry, hashing and the build of the index: cindex.hpp / cindex.hip)

// which dictionaries may still see a singleton when bins are cut at maxsearch: the window scan walks a bin from its
// end and stops after maxsearch entries (:388), flagged reads included
__global__ void k_dict_eligible(const uint64_t *__restrict__ sgbits, int W, const uint32_t *__restrict__ ids, size_t n,
                                const uint64_t *__restrict__ slots, uint32_t log2cap, int start, int len, int l, uint32_t maxsearch,
                                uint32_t *__restrict__ elig)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t sg = ids[i];
	uint32_t s = 0, c = 0;
	if (!dict_find(slots, log2cap, bits_key(sgbits + (size_t)sg * W, start, len), s, c)) return;
	if ((uint64_t)s + c - 1 - i < maxsearch) atomicOr(&elig[sg], 1u << l);
}

extern "C" int mcom_dicts_eligible(mcom_ctx *ctx, const mcom_dicts *d, const uint64_t *d_sgbits, int maxsearch, uint32_t *d_elig)
{
	if (!ctx || !d) return MCOM_E_ARG;
	if (d->n_sg == 0) return MCOM_OK;
	if (!d_sgbits || !d_elig || maxsearch < 1) return mcom_fail(ctx, MCOM_E_ARG, "bad eligibility arguments");
	MCOM_HIP(ctx, hipMemsetAsync(d_elig, 0, d->n_sg * 4, ctx->stream));
	for (int l = 0; l < d->nd; ++l)
		MCOM_LAUNCH(k_dict_eligible, dim3((unsigned)((d->n_sg + 255) / 256)), dim3(256), 0, ctx->stream, d_sgbits, d->W, d->ids[l], d->n_sg,
		                   d->slots[l], d->log2cap[l], d->ds[l], d->kl[l], l, (uint32_t)maxsearch, d_elig);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

// encode_ok with the work proportional to the mismatches: between two mismatches the scan only counts matches
// (kthread_hash_realign.c:283-314), so it is enough to visit the set bits of the mismatch mask in scan order
template <int W>
__device__ __forceinline__ bool encode_ok_sparse(const uint64_t (&mm)[W], int L, bool rev)
{
	int len = 0, eq = 0, prev = -1;
#pragma unroll
	for (int u = 0; u < W; ++u) {
		uint64_t m = mm[rev ? W - 1 - u : u];
		const int wbase = 32 * (rev ? W - 1 - u : u);
		while (m) {
			int p;
			if (rev) { const int b = 63 - __clzll((long long)m); m &= ~(1ull << b); p = L - 1 - (wbase + (b >> 1)); }
			else { const int b = __ffsll((unsigned long long)m) - 1; m &= m - 1; p = wbase + (b >> 1); }
			eq += p - prev - 1; prev = p;
			if (eq > 1) { len += ndigits_dev(eq); eq = 0; } else len += eq;
			++len;
		}
	}
	if (len == 0) len = 1;
	return (double)len <= (double)L * 0.4;
}

// The L bases of a packed contig from base jj on, as a packed row; dir: their reverse complement (what the reference compares a
// reverse-strand read with, kthread_hash_realign.c:446-461)
template <int W>
__device__ __forceinline__ void contig_window(const uint64_t *__restrict__ contig, uint64_t jj, int L, bool dir, uint64_t (&win)[W])
{
	const uint64_t *src = contig + ((2 * jj) >> 6);
	const int sh = (int)((2 * jj) & 63);
	uint64_t cur = src[0];
#pragma unroll
	for (int w = 0; w < W; ++w) { const uint64_t nxt = src[w + 1]; win[w] = sh ? (cur >> sh) | (nxt << (64 - sh)) : cur; cur = nxt; }
	const int tail = 2 * L - 64 * (W - 1);
	if (tail < 64) win[W - 1] &= (1ull << tail) - 1;
	if (dir) {
		uint64_t tt[W];
#pragma unroll
		for (int w = 0; w < W; ++w) tt[w] = ~rev_groups(win[W - 1 - w]);
		const int drop = 64 * W - 2 * L;
#pragma unroll
		for (int w = 0; w < W; ++w) { const uint64_t a = tt[w], b = w + 1 < W ? tt[w + 1] : 0ull; win[w] = drop ? (a >> drop) | (b << (64 - drop)) : a; }
		if (tail < 64) win[W - 1] &= (1ull << tail) - 1;
	}
}

// G lanes per singleton, lane q < 2*nd handles (dir = q / nd, dict = q % nd).  A lane first collects the contig
// positions that carry its key (a few at most), then all lanes verify their i-th candidate together: the expensive
// part runs converged instead of once per slot of the probe loop.
#define RR_CAND 4
// TUP: singletons with mark[sg] set (members of a bin longer than maxsearch) do not take part in the minimum: every
// tuple they pass is appended to `tuples` {claim key, singleton} for the replay of those bins on the host
// SHARED: the index is shared out by key over several GPUs (its own instantiation: the one-GPU kernel sits at 94 registers, five
// waves per SIMD, and the ownership test's few more would leave it four -- 19 -> 21.5 ms)
// QPL (a kernel of its own again, lesson 6): (direction, dictionary) pairs per LANE.  On several GPUs a rank owns 1 / R of the keys, so with a
// lane per pair seven lanes in eight of an 8-rank job drop out after the ownership test and a wave runs as long as ever for an eighth of
// the lookups (profiles/r05_dist_kernels.txt: the eight ranks together spent 4 x the one-GPU kernel's time).  With QPL pairs per lane -- 16 / QPL
// lanes per singleton, a lane going through its pairs one after the other -- the waves are as many as the lookups that are left.
template <int W, int G, bool TUP, bool SHARED, int QPL>
__global__ __launch_bounds__(256) void k_realign_reads(CixGeom g, const unsigned long long *__restrict__ keys,
                                                       const uint64_t *__restrict__ sgbits, const uint8_t *__restrict__ sgflag,
                                                       const uint32_t *__restrict__ elig, size_t n_sg, const uint64_t *__restrict__ cbits,
                                                       const uint64_t *__restrict__ coff, const uint64_t *__restrict__ woff, int thr,
                                                       unsigned long long *__restrict__ claim, unsigned long long *__restrict__ stats,
                                                       const uint8_t *__restrict__ mark, ulonglong2 *__restrict__ tuples,
                                                       unsigned long long tup_cap, unsigned long long *__restrict__ tup_count)
{
	constexpr int GL = G / QPL;                                                          // lanes per singleton
	const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t sg = t / GL;
	const int q0 = (int)(t % GL);
	uint32_t n_look = 0, n_cand = 0, n_pass = 0;                                          // (a lane's own counts: small)
	const int L = g.L;
	const bool sg_live = sg < n_sg && !sgflag[sg];
	const uint32_t el = (sg_live && elig) ? elig[sg] : 0xFFFFFFFFu;
	uint64_t row[W];
#pragma unroll
	for (int w = 0; w < W; ++w) row[w] = 0;
	bool have_row = false;
#pragma unroll 1
	for (int it = 0; it < QPL; ++it) {
	const int q = q0 + it * GL;
	const int dir = q / g.nd, l = q - dir * g.nd;
	bool live = sg_live && q < 2 * g.nd && !(dir && g.ds[l] <= 0);                      // kthread_hash_realign.c:440
	if (live && !((el >> l) & 1u)) live = false;
	const bool marked = TUP && live && mark[sg];
	const int off = dir ? L - g.ds[l] - g.klen : g.ds[l];
	// what one candidate (contig, position of the key) amounts to
	auto verify = [&](uint64_t v) {
		const uint32_t c = (uint32_t)((v & ((1ull << CIX_TAG_SHIFT) - 1)) >> g.pbits);
		const int64_t jj = (int64_t)(v & ((1ull << g.pbits) - 1)) - off;
		if (jj < 0 || (uint64_t)jj >= woff[c + 1] - woff[c]) return;
		++n_cand;
		uint64_t win[W], x[W];
		contig_window<W>(cbits + coff[c], (uint64_t)jj, L, dir != 0, win);
		int dist = 0;
#pragma unroll
		for (int w = 0; w < W; ++w) { x[w] = win[w] ^ row[w]; dist += __popcll(x[w]); }
		if (dist > thr || bits_key(x, g.ds[l], g.klen) != 0) return;                         // a tag is not the key: exact check here
		// a lower dictionary that also sees this read at this window claims the same tuple with a smaller key
		// (not for a marked singleton: whether the lower dictionary sees it there is decided by the replay)
		if (!marked)
			for (int l2 = 0; l2 < l; ++l2)
				if (((el >> l2) & 1u) && (!dir || g.ds[l2] > 0) && bits_key(x, g.ds[l2], g.klen) == 0) return;
		uint64_t mm[W];
#pragma unroll
		for (int w = 0; w < W; ++w) mm[w] = (x[w] | (x[w] >> 1)) & 0x5555555555555555ull;
		if (!dir) { if (!encode_ok_sparse<W>(mm, L, false)) return; }                          // :393
		else if (thr > 24 && !encode_ok_sparse<W>(mm, L, true)) return;                       // :461
		++n_pass;
		const unsigned long long ck = ((unsigned long long)c << 33) | ((unsigned long long)jj << 5) | ((unsigned long long)dir << 4) | (unsigned long long)l;
		if (marked) {
			const unsigned long long at = atomicAdd(tup_count, 1ull);
			if (at < tup_cap) tuples[at] = make_ulonglong2(ck, (unsigned long long)sg);
		} else atomicMin(&claim[sg], ck);
	};

	uint64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0; int nc = 0;
	if (live) {
		const uint64_t *rb = sgbits + sg * (size_t)W;
		constexpr bool shared = SHARED;                                                    // the index is shared out by key: most keys are another rank's,
		if (!shared && !have_row) {                                                        // and the row is loaded only once the key turns out to be this share's
#pragma unroll
			for (int w = 0; w < W; ++w) row[w] = rb[w];
			have_row = true;
		}
		uint64_t key = shared ? bits_key(rb, g.ds[l], g.klen) : bits_key(row, g.ds[l], g.klen);
		const uint64_t kmask = (1ull << (2 * g.klen)) - 1;
		if (dir) key = (~(rev_groups(key) >> (64 - 2 * g.klen))) & kmask;
		uint32_t own, part, h16;
		cix_hash(key, SHARED ? g.n_owners : 1u, g.n_parts, own, part, h16);
		const uint32_t nl = g.n_lines;
		const unsigned long long *lines = keys + CIX_HEAD_WORDS;
		const unsigned long long *L0 = lines + (size_t)part * nl * 8;
		uint32_t line = cix_home(h16, nl);
		const unsigned long long tag = cix_tag(key);
		// (multi-GPU: the index is shared out by key; a key of another share is looked up on its owner's GPU, with this singleton's row
		// replicated there, and the claim keys of all shares are MIN-reduced)
		const bool mine = !SHARED || own == g.owner;
		n_look += mine;
		if (shared && mine && !have_row) {
#pragma unroll
			for (int w = 0; w < W; ++w) row[w] = rb[w];
			have_row = true;
		}
		// the home line, then the lines behind it while entries were pushed on; when the home line says its keys are heavy (a repeat
		// with more copies than a few lines hold) their entries are in a run of lines in the extension area, read afterwards
		unsigned long long heavy = 0;
		for (bool more = mine, home = true; more; home = false) {
			const unsigned long long *kl = L0 + (size_t)line * 8;
			unsigned long long ks[8];
#pragma unroll
			for (int s = 0; s < 8; ++s) ks[s] = kl[s];
			const unsigned long long filled = ks[0] & 0xFFull;                            // entries in this line
			more = (ks[0] & CIX_MORE) != 0;                                               // some were pushed on to the next line
			if (home && (ks[0] & CIX_HEAVY)) heavy = ks[0];
#pragma unroll
			for (int s = 1; s < 8; ++s) {
				if ((unsigned long long)s > filled) break;
				if ((ks[s] >> CIX_TAG_SHIFT) != tag) continue;
				const uint64_t v = ks[s];
				if (nc == 0) c0 = v; else if (nc == 1) c1 = v; else if (nc == 2) c2 = v; else if (nc == 3) c3 = v;
				else verify(v);                                                                // a repeat: more copies than registers
				++nc;
			}
			line = line + 1 == nl ? 0 : line + 1;
		}
		if (heavy) {
			const unsigned long long *X = lines + ((size_t)g.n_parts * nl + (heavy >> 32)) * 8;
			const uint32_t rl = (uint32_t)(heavy >> 10) & 0x3FFFFFu;
#pragma unroll 1
			for (uint32_t j = 0; j < rl; ++j) {
				const uint32_t cn = (uint32_t)X[(size_t)j * 8] & 0xFFu;
#pragma unroll 1
				for (uint32_t s = 1; s <= cn; ++s) {
					const unsigned long long v = X[(size_t)j * 8 + s];
					if ((v >> CIX_TAG_SHIFT) != tag) continue;
					if (nc == 0) c0 = v; else if (nc == 1) c1 = v; else if (nc == 2) c2 = v; else if (nc == 3) c3 = v;
					else verify(v);
					++nc;
				}
			}
		}
	}
#pragma unroll 1
	for (int i = 0; i < RR_CAND; ++i) {
		if (!__any(nc > i)) break;
		if (nc > i) verify(i == 0 ? c0 : i == 1 ? c1 : i == 2 ? c2 : c3);
	}
	}
	if (stats) {
		// the three counts of a lane travel as ONE word through one reduction, and a workgroup sends one set of atomics (three reductions and
		// three atomics per wave cost 0.8 of the kernel's 19 ms).  Fields of 16 / 28 / 20 bits for lookups / verified / passing: a lane's
		// counts are clamped (255, 2^20 - 1, 4095 -- a lane does a handful of lookups, but a heavy repeat key gives it thousands of
		// candidates) so that the sums over the 256 lanes of the workgroup cannot carry into the neighbouring field
		__shared__ unsigned long long wg_sum;
		if (threadIdx.x == 0) wg_sum = 0;
		const unsigned long long f_look = n_look < 255u ? n_look : 255u, f_cand = n_cand < 0xFFFFFu ? n_cand : 0xFFFFFu, f_pass = n_pass < 4095u ? n_pass : 4095u;
		unsigned long long pk = f_look | (f_cand << 16) | (f_pass << 44);
		for (int o = 32; o; o >>= 1) pk += __shfl_xor(pk, o);
		__syncthreads();
		if ((threadIdx.x & 63) == 0) atomicAdd(&wg_sum, pk);
		__syncthreads();
		if (threadIdx.x == 0) {
			// 1024 sets of counters: millions of atomics on three addresses would serialise on one L2 channel
			unsigned long long *st = stats + 4 * (blockIdx.x & 1023);
			const unsigned long long a = wg_sum & 0xFFFFu, b = (wg_sum >> 16) & 0xFFFFFFFu, c = wg_sum >> 44;
			atomicAdd(&st[0], a); if (b) atomicAdd(&st[1], b); if (c) atomicAdd(&st[2], c);
		}
	}
}

// The lookups of a SHARE of the keys (multi-GPU), a thread per OWNED (singleton, pair) task.  With QPL pairs per lane (above) the lanes
// of a wave own their keys in different turns of the loop, so a wave still walks the lookup's chain of dependent loads in nearly every
// turn (profiles/r05_dist_kernels.txt: 5.25 ms per rank of eight against 1.5 for an eighth of the one-GPU kernel).  Here a workgroup goes
// through its singletons in batches of 256 / (G / 4) (a lane computes four keys of a singleton and tests who owns them: arithmetic on one
// row, no table access), the owned pairs are pushed on a stack in LDS, and whenever 256 tasks are there the 256 threads take one each:
// key again, table lines, candidates, verification -- exactly the steps of k_realign_reads -- with every lane of the wave at work.  What is
// left when the workgroup's singletons are through is done by as many threads as there are tasks.  A lane pushes its pairs in a row and
// the lanes of a singleton are neighbours, so the tasks of one singleton -- one row, mostly one contig window -- stay side by side.
#define RO_PAIRS 4
template <int W, int G, bool TUP>
__global__ __launch_bounds__(256) void k_realign_owned(CixGeom g, const unsigned long long *__restrict__ keys,
                                                       const uint64_t *__restrict__ sgbits, const uint8_t *__restrict__ sgflag,
                                                       const uint32_t *__restrict__ elig, size_t n_sg, uint32_t sg_per_wg,
                                                       const uint64_t *__restrict__ cbits, const uint64_t *__restrict__ coff,
                                                       const uint64_t *__restrict__ woff, int thr, unsigned long long *__restrict__ claim,
                                                       unsigned long long *__restrict__ stats, const uint8_t *__restrict__ mark,
                                                       ulonglong2 *__restrict__ tuples, unsigned long long tup_cap,
                                                       unsigned long long *__restrict__ tup_count)
{
	constexpr int GL = G / RO_PAIRS;                                                     // lanes per singleton while the keys are made
	constexpr uint32_t SGB = 256 / GL;                                                   // singletons per batch
	__shared__ uint32_t stack[256 * RO_PAIRS + 256];                                     // tasks: (singleton - the workgroup's first) << 5 | pair
	__shared__ uint32_t s_top;
	__shared__ unsigned long long wg_sum[3];
	const int L = g.L;
	const size_t sg_first = (size_t)blockIdx.x * sg_per_wg;
	const size_t sg_end = sg_first + sg_per_wg < n_sg ? sg_first + sg_per_wg : n_sg;
	const uint64_t kmask = (1ull << (2 * g.klen)) - 1;
	uint32_t n_look = 0, n_cand = 0, n_pass = 0;
	if (threadIdx.x == 0) { s_top = 0; wg_sum[0] = wg_sum[1] = wg_sum[2] = 0; }
	__syncthreads();

	// one task: the body of k_realign_reads for a pair whose key is this share's
	auto task = [&](uint32_t tk, bool valid) {
		const size_t sg = sg_first + (tk >> 5);
		const int q = (int)(tk & 31u);
		const int dir = q / g.nd, l = q - dir * g.nd;
		const uint32_t el = (valid && elig) ? elig[sg] : 0xFFFFFFFFu;
		const bool marked = TUP && valid && mark[sg];
		const int off = dir ? L - g.ds[l] - g.klen : g.ds[l];
		uint64_t row[W];
		const uint64_t *rb = sgbits + sg * (size_t)W;
#pragma unroll
		for (int w = 0; w < W; ++w) row[w] = valid ? rb[w] : 0ull;
		auto verify = [&](uint64_t v) {
			const uint32_t c = (uint32_t)((v & ((1ull << CIX_TAG_SHIFT) - 1)) >> g.pbits);
			const int64_t jj = (int64_t)(v & ((1ull << g.pbits) - 1)) - off;
			if (jj < 0 || (uint64_t)jj >= woff[c + 1] - woff[c]) return;
			++n_cand;
			uint64_t win[W], x[W];
			contig_window<W>(cbits + coff[c], (uint64_t)jj, L, dir != 0, win);
			int dist = 0;
#pragma unroll
			for (int w = 0; w < W; ++w) { x[w] = win[w] ^ row[w]; dist += __popcll(x[w]); }
			if (dist > thr || bits_key(x, g.ds[l], g.klen) != 0) return;
			if (!marked)
				for (int l2 = 0; l2 < l; ++l2)
					if (((el >> l2) & 1u) && (!dir || g.ds[l2] > 0) && bits_key(x, g.ds[l2], g.klen) == 0) return;
			uint64_t mm[W];
#pragma unroll
			for (int w = 0; w < W; ++w) mm[w] = (x[w] | (x[w] >> 1)) & 0x5555555555555555ull;
			if (!dir) { if (!encode_ok_sparse<W>(mm, L, false)) return; }                      // kthread_hash_realign.c:393
			else if (thr > 24 && !encode_ok_sparse<W>(mm, L, true)) return;                   // :461
			++n_pass;
			const unsigned long long ck = ((unsigned long long)c << 33) | ((unsigned long long)jj << 5) | ((unsigned long long)dir << 4) | (unsigned long long)l;
			if (marked) {
				const unsigned long long at = atomicAdd(tup_count, 1ull);
				if (at < tup_cap) tuples[at] = make_ulonglong2(ck, (unsigned long long)sg);
			} else atomicMin(&claim[sg], ck);
		};
		uint64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0; int nc = 0;
		if (valid) {
			uint64_t key = bits_key(row, g.ds[l], g.klen);
			if (dir) key = (~(rev_groups(key) >> (64 - 2 * g.klen))) & kmask;
			uint32_t own, part, h16;
			cix_hash(key, g.n_owners, g.n_parts, own, part, h16);
			const uint32_t nl = g.n_lines;
			const unsigned long long *lines = keys + CIX_HEAD_WORDS;
			const unsigned long long *L0 = lines + (size_t)part * nl * 8;
			uint32_t line = cix_home(h16, nl);
			const unsigned long long tag = cix_tag(key);
			++n_look;
			unsigned long long heavy = 0;
			for (bool more = true, home = true; more; home = false) {
				const unsigned long long *kl = L0 + (size_t)line * 8;
				unsigned long long ks[8];
#pragma unroll
				for (int s = 0; s < 8; ++s) ks[s] = kl[s];
				const unsigned long long filled = ks[0] & 0xFFull;
				more = (ks[0] & CIX_MORE) != 0;
				if (home && (ks[0] & CIX_HEAVY)) heavy = ks[0];
#pragma unroll
				for (int s = 1; s < 8; ++s) {
					if ((unsigned long long)s > filled) break;
					if ((ks[s] >> CIX_TAG_SHIFT) != tag) continue;
					const uint64_t v = ks[s];
					if (nc == 0) c0 = v; else if (nc == 1) c1 = v; else if (nc == 2) c2 = v; else if (nc == 3) c3 = v;
					else verify(v);                                                            // a repeat: more copies than registers
					++nc;
				}
				line = line + 1 == nl ? 0 : line + 1;
			}
			if (heavy) {
				const unsigned long long *X = lines + ((size_t)g.n_parts * nl + (heavy >> 32)) * 8;
				const uint32_t rl = (uint32_t)(heavy >> 10) & 0x3FFFFFu;
#pragma unroll 1
				for (uint32_t j = 0; j < rl; ++j) {
					const uint32_t cn = (uint32_t)X[(size_t)j * 8] & 0xFFu;
#pragma unroll 1
					for (uint32_t s = 1; s <= cn; ++s) {
						const unsigned long long v = X[(size_t)j * 8 + s];
						if ((v >> CIX_TAG_SHIFT) != tag) continue;
						if (nc == 0) c0 = v; else if (nc == 1) c1 = v; else if (nc == 2) c2 = v; else if (nc == 3) c3 = v;
						else verify(v);
						++nc;
					}
				}
			}
		}
#pragma unroll 1
		for (int i = 0; i < RR_CAND; ++i) {                                                   // the lanes of a wave verify their i-th candidate together
			if (!__any(nc > i)) break;
			if (nc > i) verify(i == 0 ? c0 : i == 1 ? c1 : i == 2 ? c2 : c3);
		}
	};

#pragma unroll 1
	for (size_t base = sg_first; base < sg_end; base += SGB) {                               // (bounds are the workgroup's: every thread takes every turn)
		const size_t sg = base + threadIdx.x / GL;
		const int q0 = (int)(threadIdx.x % GL);
		uint32_t mine = 0;                                                                    // bit it: pair q0 + it * GL is this share's
		if (sg < sg_end && !sgflag[sg]) {
			const uint32_t el = elig ? elig[sg] : 0xFFFFFFFFu;
			uint64_t row[W];
			const uint64_t *rb = sgbits + sg * (size_t)W;
#pragma unroll
			for (int w = 0; w < W; ++w) row[w] = rb[w];
#pragma unroll
			for (int it = 0; it < RO_PAIRS; ++it) {
				const int q = q0 + it * GL;
				const int dir = q / g.nd, l = q - dir * g.nd;
				if (q >= 2 * g.nd || (dir && g.ds[l] <= 0) || !((el >> l) & 1u)) continue;      // kthread_hash_realign.c:440
				uint64_t key = bits_key(row, g.ds[l], g.klen);
				if (dir) key = (~(rev_groups(key) >> (64 - 2 * g.klen))) & kmask;
				uint32_t own, part, h16;
				cix_hash(key, g.n_owners, g.n_parts, own, part, h16);
				if (own == g.owner) mine |= 1u << it;
			}
		}
		// room on the stack: a prefix sum inside the wave, one LDS atomic per wave
		const uint32_t cnt = (uint32_t)__popc(mine);
		uint32_t incl = cnt;
		const int lane = (int)(threadIdx.x & 63);
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(incl, o); if (lane >= o) incl += up; }
		uint32_t wbase = 0;
		if (lane == 63 && incl) wbase = atomicAdd(&s_top, incl);
		wbase = __shfl(wbase, 63);
		uint32_t at = wbase + incl - cnt;
		const uint32_t rel = (uint32_t)(sg - sg_first) << 5;
#pragma unroll
		for (int it = 0; it < RO_PAIRS; ++it)
			if ((mine >> it) & 1u) stack[at++] = rel | (uint32_t)(q0 + it * GL);
		__syncthreads();
		uint32_t top = s_top;
		while (top >= 256u) {                                                                 // (top is the same in every thread)
			const uint32_t tk = stack[top - 256u + threadIdx.x];
			top -= 256u;
			__syncthreads();                                                                  // every task read before the next push lands on its place
			if (threadIdx.x == 0) s_top = top;
			task(tk, true);
		}
		__syncthreads();
	}
	{
		const uint32_t top = s_top;                                                          // (< 256; every thread goes in: the verification is a wave's)
		const bool valid = threadIdx.x < top;
		task(valid ? stack[threadIdx.x] : 0u, valid);
	}
	if (stats) {
		// a thread has gone through the tasks of many singletons: its counts are whole words here, three reductions per workgroup in all
		unsigned long long a = n_look, b = n_cand, c = n_pass;
		for (int o = 32; o; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); c += __shfl_xor(c, o); }
		if ((threadIdx.x & 63) == 0) { atomicAdd(&wg_sum[0], a); atomicAdd(&wg_sum[1], b); atomicAdd(&wg_sum[2], c); }
		__syncthreads();
		if (threadIdx.x == 0) {
			unsigned long long *st = stats + 4 * (blockIdx.x & 1023);
			atomicAdd(&st[0], wg_sum[0]); if (wg_sum[1]) atomicAdd(&st[1], wg_sum[1]); if (wg_sum[2]) atomicAdd(&st[2], wg_sum[2]);
		}
	}
}

__global__ void k_stats_fold(const unsigned long long *__restrict__ sets, unsigned long long *__restrict__ out)
{
	const int c = threadIdx.x;                                                       // 3 threads
	unsigned long long s = 0;
	for (int i = 0; i < 1024; ++i) s += sets[4 * i + c];
	out[c] = s;
}

static int realign_reads_launch(mcom_ctx *ctx, const uint64_t *d_keys, uint64_t geom, const uint64_t *d_sgbits,
                                const uint8_t *d_sgflag, const uint32_t *d_elig, size_t n_sg, const uint64_t *d_cbits,
                                const uint64_t *d_coff, const uint64_t *d_woff, uint32_t n_contigs, int L, int ininumdict, int thr,
                                uint64_t *d_claim, uint64_t *d_stats, const uint8_t *d_mark, ulonglong2 *d_tuples, uint64_t cap,
                                unsigned long long *d_count)
{
	if (!ctx) return MCOM_E_ARG;
	CixGeom g;
	if (L < 1 || L > 256 || cix_geom(L, ininumdict, g)) return mcom_fail(ctx, MCOM_E_ARG, "bad dictionary layout");
	cix_unpack(geom, g);
	g.pbits = cix_pbits(n_contigs);
	if (n_sg) MCOM_HIP(ctx, hipMemsetAsync(d_claim, 0xFF, n_sg * 8, ctx->stream));
	if (d_stats) MCOM_HIP(ctx, hipMemsetAsync(d_stats, 0, 3 * 8, ctx->stream));
	if (n_contigs == 0 || n_sg == 0) return MCOM_OK;
	unsigned long long *sets = nullptr;
	if (d_stats) {
		int rcw = mcom_ws_reserve(ctx, 1024 * 4 * 8);
		if (rcw) return rcw;
		sets = (unsigned long long*)ctx->ws;
		MCOM_HIP(ctx, hipMemsetAsync(sets, 0, 1024 * 4 * 8, ctx->stream));
	}
	if (!d_keys || !d_sgbits || !d_sgflag || !d_cbits || !d_coff || !d_woff || !d_claim || g.n_parts < 1 || g.n_lines < 1)
		return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	const int W = mcom_words_per_read(L);
	const int G = 2 * g.nd <= 16 ? 16 : 32;
	const uint64_t blocks = (n_sg * (uint64_t)G + 255) / 256;
	if (blocks >= (1ull << 31)) return mcom_fail(ctx, MCOM_E_ARG, "too many singletons for one launch");
#define MCOM_ARGS g, (const unsigned long long*)d_keys, d_sgbits, d_sgflag, d_elig, n_sg, d_cbits, d_coff, d_woff, thr, (unsigned long long*)d_claim, sets, d_mark, d_tuples, (unsigned long long)cap, d_count
	// a share of the keys: workgroups of a few thousand singletons each (the tasks a workgroup is left with at its end fill part of one turn)
	uint32_t sg_per_wg = 64;
	if (g.n_owners > 1) {
		const uint64_t want_wgs = (uint64_t)(ctx->n_cu > 0 ? ctx->n_cu : 256) * 20;
		uint64_t per = (n_sg + want_wgs - 1) / want_wgs;
		per = (per + 63) & ~63ull;                                                         // whole batches (64 or 32 singletons)
		sg_per_wg = (uint32_t)(per < 512 ? 512 : per > 65536 ? 65536 : per);
	}
	const uint64_t owned_blocks = (n_sg + sg_per_wg - 1) / sg_per_wg;
#define MCOM_RA_LAUNCH(WW, GG, TT) do { \
	if (g.n_owners > 1 && ctx->lookup_route == 0) MCOM_LAUNCH((k_realign_owned<WW, GG, TT>), dim3((unsigned)owned_blocks), dim3(256), 0, ctx->stream, \
		g, (const unsigned long long*)d_keys, d_sgbits, d_sgflag, d_elig, n_sg, sg_per_wg, d_cbits, d_coff, d_woff, thr, (unsigned long long*)d_claim, sets, d_mark, d_tuples, (unsigned long long)cap, d_count); \
	else if (g.n_owners >= 8) MCOM_LAUNCH((k_realign_reads<WW, GG, TT, true, 8>), dim3((unsigned)((n_sg * (uint64_t)(GG / 8) + 255) / 256)), dim3(256), 0, ctx->stream, MCOM_ARGS); \
	else if (g.n_owners >= 3) MCOM_LAUNCH((k_realign_reads<WW, GG, TT, true, 4>), dim3((unsigned)((n_sg * (uint64_t)(GG / 4) + 255) / 256)), dim3(256), 0, ctx->stream, MCOM_ARGS); \
	else if (g.n_owners > 1) MCOM_LAUNCH((k_realign_reads<WW, GG, TT, true, 2>), dim3((unsigned)((n_sg * (uint64_t)(GG / 2) + 255) / 256)), dim3(256), 0, ctx->stream, MCOM_ARGS); \
	else MCOM_LAUNCH((k_realign_reads<WW, GG, TT, false, 1>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, MCOM_ARGS); } while (0)
#define MCOM_CASE(WW) case WW: \
	if (d_mark) { if (G == 16) MCOM_RA_LAUNCH(WW, 16, true); else MCOM_RA_LAUNCH(WW, 32, true); } \
	else if (G == 16) MCOM_RA_LAUNCH(WW, 16, false); else MCOM_RA_LAUNCH(WW, 32, false); break;
	McomProfScope ps_(ctx, PROF_REALIGN_READS);
	switch (W) { MCOM_CASE(1) MCOM_CASE(2) MCOM_CASE(3) MCOM_CASE(4) MCOM_CASE(5) MCOM_CASE(6) MCOM_CASE(7) MCOM_CASE(8)
	default: return mcom_fail(ctx, MCOM_E_ARG, "unsupported read length"); }
#undef MCOM_CASE
#undef MCOM_RA_LAUNCH
#undef MCOM_ARGS
	MCOM_LAUNCH_CHECK(ctx);
	if (d_stats) MCOM_LAUNCH(k_stats_fold, dim3(1), dim3(3), 0, ctx->stream, sets, (unsigned long long*)d_stats);
	return MCOM_OK;
}

extern "C" int mcom_realign_pass_reads(mcom_ctx *ctx, const uint64_t *d_keys, uint64_t n_parts, const uint64_t *d_sgbits,
                                       const uint8_t *d_sgflag, const uint32_t *d_elig, size_t n_sg, const uint64_t *d_cbits,
                                       const uint64_t *d_coff, const uint64_t *d_woff, uint32_t n_contigs, int L, int ininumdict, int thr,
                                       uint64_t *d_claim, uint64_t *d_stats)
{
	return realign_reads_launch(ctx, d_keys, n_parts, d_sgbits, d_sgflag, d_elig, n_sg, d_cbits, d_coff, d_woff, n_contigs, L, ininumdict, thr,
	                            d_claim, d_stats, nullptr, nullptr, 0, nullptr);
}

// ---- a14 as a batched entry of its own: the cost test of kthread_hash_realign.c:283-314 for n (read, contig window) pairs, with the
// device functions the pass itself uses (contig_window, encode_ok_sparse)
template <int W>
__global__ void k_encode_byte(const uint64_t *__restrict__ rows, const uint64_t *__restrict__ cbits, const uint64_t *__restrict__ coff,
                              const uint32_t *__restrict__ contig, const uint32_t *__restrict__ pos, const uint8_t *__restrict__ dirs, size_t n, int L,
                              uint8_t *__restrict__ ok)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint64_t win[W], mm[W];
	const bool dir = dirs[i] != 0;
	contig_window<W>(cbits + coff[contig[i]], (uint64_t)pos[i], L, dir, win);
#pragma unroll
	for (int w = 0; w < W; ++w) { const uint64_t x = win[w] ^ rows[i * (size_t)W + w]; mm[w] = (x | (x >> 1)) & 0x5555555555555555ull; }
	ok[i] = encode_ok_sparse<W>(mm, L, dir) ? 1 : 0;
}
extern "C" int mcom_encode_byte(mcom_ctx *ctx, const uint64_t *d_rows, const uint64_t *d_cbits, const uint64_t *d_coff, const uint32_t *d_contig,
                                const uint32_t *d_pos, const uint8_t *d_dir, size_t n, int L, uint8_t *d_ok)
{
	if (!ctx) return MCOM_E_ARG;
	if (L < 1 || L > 256) return mcom_fail(ctx, MCOM_E_ARG, "read length %d out of range", L);
	if (n == 0) return MCOM_OK;
	if (!d_rows || !d_cbits || !d_coff || !d_contig || !d_pos || !d_dir || !d_ok) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	const unsigned blocks = (unsigned)((n + 255) / 256);
#define MCOM_CASE(WW) case WW: MCOM_LAUNCH((k_encode_byte<WW>), dim3(blocks), dim3(256), 0, ctx->stream, d_rows, d_cbits, d_coff, d_contig, d_pos, d_dir, n, L, d_ok); break;
	switch (mcom_words_per_read(L)) { MCOM_CASE(1) MCOM_CASE(2) MCOM_CASE(3) MCOM_CASE(4) MCOM_CASE(5) MCOM_CASE(6) MCOM_CASE(7) MCOM_CASE(8)
	default: return mcom_fail(ctx, MCOM_E_ARG, "unsupported read length"); }
#undef MCOM_CASE
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

// ---- bins longer than maxsearch ---------------------------------------------------------------------------------
// The scan walks the LIVE part of a bin from its end for at most maxsearch entries (kthread_hash_realign.c:388,
// bbhashdict.c:33-67), and claimed reads leave every bin after the visit that claimed them (:420-435): a read deep in
// a long bin becomes visible once enough of the reads above it are gone.  That is sequential, but only for the
// members of such bins: they are marked here, the pass hands back every tuple they pass, and the host replays the
// visits of those few reads in order.
__global__ void k_dict_bigbins(const uint64_t *__restrict__ sgbits, int W, const uint32_t *__restrict__ ids, size_t n,
                               const uint64_t *__restrict__ slots, uint32_t log2cap, int start, int len, uint32_t maxsearch,
                               uint32_t *__restrict__ binstart, uint8_t *__restrict__ mark)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t sg = ids[i];
	uint32_t s = 0, c = 0;
	const bool f = dict_find(slots, log2cap, bits_key(sgbits + (size_t)sg * W, start, len), s, c);
	if (f && c > maxsearch) { binstart[sg] = s; mark[sg] = 1; }
	else binstart[sg] = 0xFFFFFFFFu;
}

extern "C" int mcom_dicts_bigbins(mcom_ctx *ctx, const mcom_dicts *d, const uint64_t *d_sgbits, int maxsearch, uint32_t *d_binstart, uint8_t *d_mark)
{
	if (!ctx || !d) return MCOM_E_ARG;
	if (d->n_sg == 0) return MCOM_OK;
	if (!d_sgbits || !d_binstart || !d_mark || maxsearch < 1) return mcom_fail(ctx, MCOM_E_ARG, "bad arguments");
	MCOM_HIP(ctx, hipMemsetAsync(d_mark, 0, d->n_sg, ctx->stream));
	for (int l = 0; l < d->nd; ++l)
		MCOM_LAUNCH(k_dict_bigbins, dim3((unsigned)((d->n_sg + 255) / 256)), dim3(256), 0, ctx->stream, d_sgbits, d->W, d->ids[l], d->n_sg,
		                   d->slots[l], d->log2cap[l], d->ds[l], d->kl[l], (uint32_t)maxsearch, d_binstart + (size_t)l * d->n_sg, d_mark);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

extern "C" int mcom_realign_pass_tuples(mcom_ctx *ctx, const uint64_t *d_keys, uint64_t n_parts, const uint64_t *d_sgbits,
                                        const uint8_t *d_sgflag, const uint8_t *d_mark, size_t n_sg, const uint64_t *d_cbits,
                                        const uint64_t *d_coff, const uint64_t *d_woff, uint32_t n_contigs, int L, int ininumdict, int thr,
                                        uint64_t *d_claim, uint64_t *d_stats, uint64_t *d_tuples, uint64_t cap, uint64_t *h_ntuples)
{
	if (!ctx) return MCOM_E_ARG;
	if (!d_mark || !d_tuples || !h_ntuples) return mcom_fail(ctx, MCOM_E_ARG, "null tuple buffer");
	unsigned long long *d_count = nullptr;
	int rc = MCOM_OK;
	hipError_t e = mcom_dmalloc(&d_count, 8);
	if (e != hipSuccess) return mcom_fail(ctx, MCOM_E_NOMEM, "tuple counter");
	e = hipMemsetAsync(d_count, 0, 8, ctx->stream);
	if (e == hipSuccess) {
		rc = realign_reads_launch(ctx, d_keys, n_parts, d_sgbits, d_sgflag, nullptr, n_sg, d_cbits, d_coff, d_woff, n_contigs, L, ininumdict, thr,
		                          d_claim, d_stats, d_mark, (ulonglong2*)d_tuples, cap, d_count);
		if (!rc) {
			unsigned long long h = 0;
			e = mcom_d2h_async(ctx, &h, d_count, 8);
			if (e == hipSuccess) e = mcom_stream_sync(ctx);
			*h_ntuples = h;
		}
	}
	mcom_dfree(d_count);
	if (e != hipSuccess) return mcom_fail(ctx, MCOM_E_HIP, hipGetErrorString(e));
	return rc;
}

__global__ void k_claims_patch(unsigned long long *__restrict__ claim, const uint32_t *__restrict__ idx, const unsigned long long *__restrict__ val, size_t n)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) claim[idx[i]] = val[i];
}

extern "C" int mcom_claims_patch(mcom_ctx *ctx, uint64_t *d_claim, const uint32_t *d_idx, const uint64_t *d_val, size_t n)
{
	if (!ctx) return MCOM_E_ARG;
	if (n == 0) return MCOM_OK;
	if (!d_claim || !d_idx || !d_val) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	MCOM_LAUNCH(k_claims_patch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (unsigned long long*)d_claim, d_idx,
	                   (const unsigned long long*)d_val, n);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

// ---- claim resolution: the appends of one pass in the order of the sequential scan ---------------------------
// The scan visits (contig, window, dir, dict) ascending and walks a bin from its end, so the members a pass appends
// are ordered by claim key ascending, singleton index descending (:388, :408-409, :474-475).  One stable radix sort
// of {key, index} fed in descending index order gives exactly that.
// Only the singletons that claimed something take part (a tenth of them): flags and a scan compact them, still in descending index
// order, before the sort.
__global__ void k_claim_flags(const unsigned long long *__restrict__ claim, size_t n, uint32_t *__restrict__ f)
{
	const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j <= n) f[j] = (j < n && claim[n - 1 - j] != U64MAX) ? 1u : 0u;
}
__global__ void k_claim_records(const unsigned long long *__restrict__ claim, size_t n, const uint32_t *__restrict__ at, mcom_mm128 *__restrict__ rec)
{
	const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n) return;
	const size_t i = n - 1 - j;
	const unsigned long long ck = claim[i];
	if (ck == U64MAX) return;
	mcom_mm128 r; r.x = ck; r.y = i;
	rec[at[j]] = r;
}
__global__ void k_claim_emit(const mcom_mm128 *__restrict__ rec, size_t n, const uint32_t *__restrict__ rids,
                             uint8_t *__restrict__ flag, uint32_t *__restrict__ app_contig, uint64_t *__restrict__ app_member)
{
	const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n) return;
	const mcom_mm128 r = rec[j];
	const uint64_t jj = (r.x >> 5) & ((1ull << 28) - 1), dir = (r.x >> 4) & 1;
	app_contig[j] = (uint32_t)(r.x >> 33);
	app_member[j] = ((uint64_t)rids[r.y] << 32) | (jj << 1) | dir;
	flag[r.y] = 3;
}

extern "C" int mcom_claims_resolve(mcom_ctx *ctx, const uint64_t *d_claim, const uint32_t *d_rids, size_t n_sg, uint32_t n_contigs,
                                   uint8_t *d_flag, uint32_t *d_app_contig, uint64_t *d_app_member, uint64_t *h_nwon)
{
	if (!ctx || !h_nwon) return MCOM_E_ARG;
	*h_nwon = 0;
	if (n_sg == 0) return MCOM_OK;
	if (!d_claim || !d_rids || !d_flag || !d_app_contig || !d_app_member) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	if (n_sg >= (1ull << 32) - 1) return mcom_fail(ctx, MCOM_E_ARG, "too many singletons");
	int cb = 1; while ((1ull << cb) < (uint64_t)n_contigs + 1 && cb < 31) ++cb;
	const int kb = 33 + cb;                                                  // claim keys are < 2^kb
	auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
	const size_t f_b = al((n_sg + 1) * 4), scr_b = al(mcom_scan_scratch_elems(n_sg + 1) * 4 + 1024);
	int rc = mcom_ws_reserve(ctx, f_b + scr_b);
	if (rc) return rc;
	uint32_t *f = (uint32_t*)ctx->ws, *scr = (uint32_t*)((char*)ctx->ws + f_b);
	const unsigned blocks = (unsigned)((n_sg + 1 + 255) / 256);
	MCOM_LAUNCH(k_claim_flags, dim3(blocks), dim3(256), 0, ctx->stream, (const unsigned long long*)d_claim, n_sg, f);
	MCOM_LAUNCH_CHECK(ctx);
	if ((rc = mcom_scan_u32(ctx, f, f, n_sg + 1, scr))) return rc;
	uint32_t nw = 0;
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &nw, f + n_sg, 4));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	*h_nwon = nw;
	if (nw == 0) return MCOM_OK;
	// the records and the sort workspace live in their own block: the flags above stay where they are
	const size_t rec_b = al((size_t)nw * sizeof(mcom_mm128));
	char *blk = nullptr;
	if (mcom_dmalloc(&blk, rec_b + mcom_sort_ws_bytes(nw)) != hipSuccess) return mcom_fail(ctx, MCOM_E_NOMEM, "claim records");
	mcom_mm128 *rec = (mcom_mm128*)blk;
	MCOM_LAUNCH(k_claim_records, dim3(blocks), dim3(256), 0, ctx->stream, (const unsigned long long*)d_claim, n_sg, f, rec);
	rc = mcom_sort_by_x(ctx, rec, nw, kb, blk + rec_b);
	if (!rc) MCOM_LAUNCH(k_claim_emit, dim3((nw + 255) / 256), dim3(256), 0, ctx->stream, rec, (size_t)nw, d_rids, d_flag, d_app_contig, d_app_member);
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) e = mcom_stream_sync(ctx);
	mcom_dfree(blk);
	if (rc) return rc;
	if (e != hipSuccess) return mcom_fail(ctx, MCOM_E_HIP, "claim resolution: %s", hipGetErrorString(e));
	return MCOM_OK;
}

// ---- can any bin exceed maxsearch?  (cheap screen before building the dictionaries) ---------------------------------
// The read-driven pass needs the dictionaries only to cut bins at maxsearch (:388), which almost never happens.  One
// counter per hashed (dictionary, key) in a table of about one counter per key: a counter is an UPPER bound of its bin
// (collisions only add), so "no counter above maxsearch" proves that no bin is.
// (n_shares > 1: the keys are shared out as the index is, a rank counts the keys of its share only -- every bin is counted whole by one rank)
__global__ void k_bin_screen(const uint64_t *__restrict__ sgbits, size_t n_sg, int W, int nd, CixGeom g, uint32_t log2t, uint32_t maxsearch,
                             unsigned int *__restrict__ table, unsigned int *__restrict__ exceeded, uint32_t n_shares, uint32_t share)
{
	const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t sg = t / (size_t)nd; const int l = (int)(t - sg * (size_t)nd);
	if (sg >= n_sg) return;
	const uint64_t key = bits_key(sgbits + sg * (size_t)W, g.ds[l], g.klen);
	if (n_shares > 1) { uint32_t own, part, h16; cix_hash(key, n_shares, 1u, own, part, h16); if (own != share) return; }
	const uint64_t h = (key * 8 + (uint64_t)l + 1) * 0x9E3779B97F4A7C15ull;
	atomicAdd(&table[h >> (64 - log2t)], 1u);                                // nothing comes back: the counters are looked at afterwards (k_bin_screen_max)
}
__global__ void k_bin_screen_max(const unsigned int *__restrict__ table, size_t n, uint32_t maxsearch, unsigned int *__restrict__ exceeded)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n && table[i] > maxsearch) *exceeded = 1;
}

// ---- the same counters WITHOUT global atomics (round 5) --------------------------------------------------------------------------
// k_bin_screen is 128 M atomics on random words of a 64 MB table (100 M x 150 bp: 16 M singletons x 8 dictionaries): 4.8 ms, served by
// the memory side at 27 G atomics a second whatever the kernel does.  The counters do not have to be in HBM: cut the counter numbers
// into BINS of 2^14 consecutive counters -- what one workgroup holds in LDS -- and count in two steps.  k_screen_scatter: a workgroup
// takes a stretch of the singletons sub-tile by sub-tile (32 K keys: thirty-two per thread), counts the keys of a
// sub-tile by bin and places them with LDS atomics, i.e. puts the 16-bit counter numbers in bin order in LDS and appends every bin's run to the
// REGION that this workgroup owns in that bin (fill pointers in LDS: nothing is reserved in global memory, no atomics leave the CU).
// k_screen_count: one workgroup per bin adds the regions of all workgroups up in LDS counters and looks at the largest.  The keys are
// hashes, so the regions fill evenly; one that overflows all the same (copies of one read: one key a million times) raises bit 1 of
// the flag and mcom_dicts_screen_end runs k_bin_screen instead.  A counter per ~16-32 keys (collisions only add: still an upper bound of
// every bin in it, and ~32 is far below any limit worth asking about).
namespace {
constexpr int SCR_THREADS = 1024, SCR_KPT = 32, SCR_KT = SCR_THREADS * SCR_KPT;      // keys of a sub-tile
constexpr int SCR_MAXBINS = 1024, SCR_CB = 14;                                       // bins at most; counters of a bin = 2^SCR_CB at most
constexpr size_t SCR_LDS_SCATTER = 2 * (size_t)SCR_KT + 4 * (4 * (size_t)SCR_MAXBINS + 16);
}
__global__ __launch_bounds__(1024) void k_screen_scatter(const uint64_t *__restrict__ sgbits, size_t n_sg, int W, int nd, CixGeom g, uint32_t log2t, uint32_t log2bins,
                                                         uint32_t sg_per_wg, uint32_t sg_per_tile, uint32_t capw, uint16_t *__restrict__ out, uint32_t *__restrict__ cnt,
                                                         unsigned int *__restrict__ flag, uint32_t n_shares, uint32_t share)
{
	extern __shared__ __align__(16) unsigned char scr_lds[];
	const uint32_t tid = threadIdx.x, lane = tid & 63u, bins = 1u << log2bins;
	uint16_t *stage = (uint16_t*)scr_lds;
	uint32_t *hist = (uint32_t*)(scr_lds + 2 * (size_t)SCR_KT), *base = hist + bins, *fill = base + bins, *cur = fill + bins, *wsum = cur + bins;   // (sized by the launch: two workgroups per CU up to 512 bins)
	const uint32_t cb = log2t - log2bins, vmask = (1u << cb) - 1u;
	const size_t sg0 = (size_t)blockIdx.x * sg_per_wg, sg1 = sg0 + sg_per_wg < n_sg ? sg0 + sg_per_wg : n_sg;
	if (tid < bins) fill[tid] = 0;
	bool over = false;
	for (size_t ts = sg0; ts < sg1; ts += sg_per_tile) {
		const uint32_t nk = (uint32_t)((ts + sg_per_tile < sg1 ? ts + sg_per_tile : sg1) - ts) * (uint32_t)nd;
		if (tid < bins) hist[tid] = 0;
		__syncthreads();
		// (the counter number of a key is made twice -- once to count its bin, once to place it: thirty-two of them do not fit a thread's
		// registers beside everything else, the first form spilled a kilobyte per lane -- the rows come out of the L1 the second time)
		auto counter_of = [&](uint32_t q) -> uint32_t {
			const uint32_t sl = q / (uint32_t)nd; const int l = (int)(q - sl * (uint32_t)nd);
			const uint64_t key = bits_key(sgbits + (ts + sl) * (size_t)W, g.ds[l], g.klen);
			if (n_shares > 1) { uint32_t own, part, h16; cix_hash(key, n_shares, 1u, own, part, h16); if (own != share) return 0xFFFFFFFFu; }
			const uint64_t h = (key * 8 + (uint64_t)l + 1) * 0x9E3779B97F4A7C15ull;          // (k_bin_screen's counter)
			return (uint32_t)(h >> (64 - log2t));
		};
#pragma unroll 4
		for (int r = 0; r < SCR_KPT; ++r) {
			const uint32_t q = (uint32_t)r * SCR_THREADS + tid;
			if (q < nk) { const uint32_t idx = counter_of(q); if (idx != 0xFFFFFFFFu) atomicAdd(&hist[idx >> cb], 1u); }
		}
		__syncthreads();
		{	// first place of every bin in the staging area (an entry per thread: bins <= 1024); does the region hold the run?
			const uint32_t v = tid < bins ? hist[tid] : 0u;
			uint32_t inc = v;
#pragma unroll
			for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d); if (lane >= (uint32_t)d) inc += t; }
			if (lane == 63u) wsum[tid >> 6] = inc;
			__syncthreads();
			uint32_t pre = 0;
			for (uint32_t w = 0; w < (tid >> 6); ++w) pre += wsum[w];
			if (tid < bins) { base[tid] = pre + inc - v; cur[tid] = pre + inc - v; if (fill[tid] + v > capw) over = true; }
		}
		__syncthreads();
#pragma unroll 4
		for (int r = 0; r < SCR_KPT; ++r) {
			const uint32_t q = (uint32_t)r * SCR_THREADS + tid;
			if (q < nk) { const uint32_t idx = counter_of(q); if (idx != 0xFFFFFFFFu) stage[atomicAdd(&cur[idx >> cb], 1u)] = (uint16_t)(idx & vmask); }
		}
		__syncthreads();
		for (uint32_t b = tid >> 6; b < bins; b += SCR_THREADS / 64) {                 // a wave per bin: runs leave as 128-byte stores
			const uint32_t n = hist[b], f = fill[b], s0 = base[b];
			if (f + n <= capw) {
				uint16_t *dst = out + ((size_t)b * gridDim.x + blockIdx.x) * capw + f;
				for (uint32_t j = lane; j < n; j += 64) dst[j] = stage[s0 + j];
			}
		}
		__syncthreads();
		if (tid < bins) { const uint32_t f = fill[tid] + hist[tid]; fill[tid] = f < capw ? f : capw; }
	}
	if (tid < bins) cnt[(size_t)tid * gridDim.x + blockIdx.x] = over ? 0u : fill[tid];
	if (over) atomicOr(flag, 2u);
}
__global__ __launch_bounds__(1024) void k_screen_count(const uint16_t *__restrict__ out, const uint32_t *__restrict__ cnt, uint32_t n_wg, uint32_t capw, uint32_t cb,
                                                       uint32_t maxsearch, unsigned int *__restrict__ flag)
{
	extern __shared__ __align__(16) unsigned char scr_lds[];
	uint32_t *ctr = (uint32_t*)scr_lds;
	const uint32_t tid = threadIdx.x, lane = tid & 63u, nctr = 1u << cb, b = blockIdx.x;
	for (uint32_t i = tid; i < nctr; i += SCR_THREADS) ctr[i] = 0;
	__syncthreads();
	for (uint32_t w = tid >> 6; w < n_wg; w += SCR_THREADS / 64) {
		const uint32_t n = cnt[(size_t)b * n_wg + w];
		const uint16_t *src = out + ((size_t)b * n_wg + w) * capw;
		for (uint32_t j = 8u * lane; j < n; j += 512u) {                             // eight numbers per lane and step (capw is a multiple of 8)
			const uint4 v = *(const uint4*)(src + j);
			const uint32_t ww[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
			for (uint32_t u = 0; u < 8; ++u) if (j + u < n) atomicAdd(&ctr[(ww[u >> 1] >> (16u * (u & 1u))) & 0xFFFFu], 1u);
		}
	}
	__syncthreads();
	uint32_t m = 0;
	for (uint32_t i = tid; i < nctr; i += SCR_THREADS) m = ctr[i] > m ? ctr[i] : m;
	if (m > maxsearch) atomicOr(flag, 1u);
}

// the global-atomics route: table + flag in the workspace
static int screen_atomics(mcom_ctx *ctx, const uint64_t *d_sgbits, size_t n_sg, int L, const CixGeom &g, int maxsearch, int n_shares, int share)
{
	const uint64_t nkeys = (uint64_t)n_sg * (uint64_t)g.nd;
	const uint64_t nmine = nkeys / (uint64_t)n_shares + 1;
	// a counter per ~8 keys: collisions only add (the count stays an upper bound of every bin in it) and ~8 is far below any limit worth
	// asking about, while the table (64 MB at 16 M singletons instead of 512) stays in the Infinity Cache, where the atomics are served
	uint32_t lg = 10; while (lg < 30 && (8ull << lg) < nmine) ++lg;
	int rc = mcom_ws_reserve(ctx, ((size_t)4 << lg) + 256);
	if (rc) return rc;
	unsigned int *table = (unsigned int*)ctx->ws;
	unsigned int *flag = (unsigned int*)((char*)ctx->ws + ((size_t)4 << lg));
	MCOM_HIP(ctx, hipMemsetAsync(table, 0, ((size_t)4 << lg) + 4, ctx->stream));
	const uint64_t blocks = (nkeys + 255) / 256;
	if (blocks >= (1ull << 31)) return mcom_fail(ctx, MCOM_E_ARG, "too many singletons for one launch");
	{ McomProfScope ps_(ctx, PROF_DICT_BUILD);
	MCOM_LAUNCH(k_bin_screen, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_sgbits, n_sg, mcom_words_per_read(L), g.nd, g, lg, (uint32_t)maxsearch, table, flag, (uint32_t)n_shares, (uint32_t)share);
	MCOM_LAUNCH(k_bin_screen_max, dim3((unsigned)((((size_t)1 << lg) + 255) / 256)), dim3(256), 0, ctx->stream, table, (size_t)1 << lg, (uint32_t)maxsearch, flag); }
	MCOM_LAUNCH_CHECK(ctx);
	ctx->screen_flag = flag;
	return MCOM_OK;
}

// begin: the counting kernels are put on the context's stream and the call returns; end: waits for them and reads the answer.
// (mcom_dicts_screen is the two in one.  The pipeline can run the screen on a second stream beside the contig index build.)
extern "C" int mcom_dicts_screen_begin_shared(mcom_ctx *ctx, const uint64_t *d_sgbits, size_t n_sg, int L, int ininumdict, int maxsearch, int n_shares, int share)
{
	if (!ctx) return MCOM_E_ARG;
	if (n_shares < 1 || share < 0 || share >= n_shares) return mcom_fail(ctx, MCOM_E_ARG, "bad share");
	ctx->screen_flag = nullptr;
	if (n_sg == 0) return MCOM_OK;
	CixGeom g;
	if (!d_sgbits || L < 1 || L > 256 || maxsearch < 1 || cix_geom(L, ininumdict, g)) return mcom_fail(ctx, MCOM_E_ARG, "bad screen arguments");
	ctx->screen_args = mcom_ctx::ScreenArgs{d_sgbits, n_sg, L, ininumdict, maxsearch, n_shares, share};
	if (ctx->screen_route == 1) return screen_atomics(ctx, d_sgbits, n_sg, L, g, maxsearch, n_shares, share);
	const uint64_t nkeys = (uint64_t)n_sg * (uint64_t)g.nd;
	const uint64_t nmine = nkeys / (uint64_t)n_shares + 1;
	uint32_t lg = 10; while (lg < (uint32_t)SCR_CB + 10u && (16ull << lg) < nmine) ++lg;      // (1024 bins of 2^14 counters at most: 16 M counters)
	const uint32_t log2bins = lg > (uint32_t)SCR_CB ? lg - (uint32_t)SCR_CB : 0u, bins = 1u << log2bins, cb = lg - log2bins;
	const uint32_t sg_per_tile = (uint32_t)(SCR_KT / g.nd);
	uint64_t n_wg = (n_sg + sg_per_tile - 1) / sg_per_tile;
	if (n_wg > 2ull * (uint64_t)ctx->n_cu) n_wg = 2ull * (uint64_t)ctx->n_cu;
	const uint64_t sg_per_wg = (n_sg + n_wg - 1) / n_wg;
	if (sg_per_wg >= (1ull << 31) / (uint64_t)g.nd) return screen_atomics(ctx, d_sgbits, n_sg, L, g, maxsearch, n_shares, share);
	// a region: 5/4 of an even share of the workgroup's keys + 256 (the keys are hashes: a run of 32 K / bins keys per sub-tile, +- its root)
	uint64_t capw = ctx->screen_route == 2 ? 8 : ((sg_per_wg * (uint64_t)g.nd / bins) * 5 / 4 + 256 + 7) & ~7ull;
	const size_t out_b = ((size_t)bins * n_wg * capw * 2 + 255) & ~(size_t)255, cnt_b = ((size_t)bins * n_wg * 4 + 255) & ~(size_t)255;
	int rc = mcom_ws_reserve(ctx, out_b + cnt_b + 256);
	if (rc) return rc;
	uint16_t *out = (uint16_t*)ctx->ws;
	uint32_t *cnt = (uint32_t*)((char*)ctx->ws + out_b);
	unsigned int *flag = (unsigned int*)((char*)ctx->ws + out_b + cnt_b);
	MCOM_HIP(ctx, hipMemsetAsync(flag, 0, 4, ctx->stream));
	static bool attr_set = false;
	if (!attr_set) {
		MCOM_HIP(ctx, hipFuncSetAttribute((const void*)k_screen_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SCR_LDS_SCATTER));
		MCOM_HIP(ctx, hipFuncSetAttribute((const void*)k_screen_count, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4u << SCR_CB)));
		attr_set = true;
	}
	{ McomProfScope ps_(ctx, PROF_DICT_BUILD);
	MCOM_LAUNCH(k_screen_scatter, dim3((unsigned)n_wg), dim3(SCR_THREADS), 2 * (size_t)SCR_KT + 4 * (4 * (size_t)bins + 16), ctx->stream, d_sgbits, n_sg, mcom_words_per_read(L), g.nd, g, lg, log2bins,
	            (uint32_t)sg_per_wg, sg_per_tile, (uint32_t)capw, out, cnt, flag, (uint32_t)n_shares, (uint32_t)share);
	MCOM_LAUNCH(k_screen_count, dim3(bins), dim3(SCR_THREADS), (size_t)4 << cb, ctx->stream, (const uint16_t*)out, (const uint32_t*)cnt, (uint32_t)n_wg, (uint32_t)capw, cb,
	            (uint32_t)maxsearch, flag); }
	MCOM_LAUNCH_CHECK(ctx);
	ctx->screen_flag = flag;
	return MCOM_OK;
}
extern "C" int mcom_dicts_screen_begin(mcom_ctx *ctx, const uint64_t *d_sgbits, size_t n_sg, int L, int ininumdict, int maxsearch)
{
	return mcom_dicts_screen_begin_shared(ctx, d_sgbits, n_sg, L, ininumdict, maxsearch, 1, 0);
}
extern "C" int mcom_dicts_screen_end(mcom_ctx *ctx, int *h_may_exceed)
{
	if (!ctx || !h_may_exceed) return MCOM_E_ARG;
	*h_may_exceed = 0;
	if (!ctx->screen_flag) return MCOM_OK;
	unsigned int hf = 0;
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &hf, ctx->screen_flag, 4));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	ctx->screen_flag = nullptr;
	if (hf & 2u) {                                                                  // a region overflowed: the counters were not all made -- the other route
		const mcom_ctx::ScreenArgs a = ctx->screen_args;
		CixGeom g;
		if (cix_geom(a.L, a.ininumdict, g)) return mcom_fail(ctx, MCOM_E_ARG, "bad screen arguments");
		++ctx->screen_fallbacks;
		const int rc = screen_atomics(ctx, a.sgbits, a.n_sg, a.L, g, a.maxsearch, a.n_shares, a.share);
		if (rc) return rc;
		MCOM_HIP(ctx, mcom_d2h_async(ctx, &hf, ctx->screen_flag, 4));
		MCOM_HIP(ctx, mcom_stream_sync(ctx));
		ctx->screen_flag = nullptr;
	}
	*h_may_exceed = (hf & 1u) ? 1 : 0;
	return MCOM_OK;
}
extern "C" int mcom_dicts_screen(mcom_ctx *ctx, const uint64_t *d_sgbits, size_t n_sg, int L, int ininumdict, int maxsearch, int *h_may_exceed)
{
	if (!ctx || !h_may_exceed) return MCOM_E_ARG;
	*h_may_exceed = 0;
	const int rc = mcom_dicts_screen_begin(ctx, d_sgbits, n_sg, L, ininumdict, maxsearch);
	return rc ? rc : mcom_dicts_screen_end(ctx, h_may_exceed);
}

// ====================================================================================================================================
// Round 5: Stage 2 as a PARTITION-LOCAL JOIN -- no table, no random line in HBM.
//
// mcom_realign_pass_reads sends 193 M lookups (100 M x 150 bp, first pass) into a 16.8 GB table: random 64-byte lines at 13 G lines/s,
// two thirds of what the card delivers for that pattern over that footprint (address translation: tools/ubench/random_lines.hip).  The
// entries of that table are already SORTED BY PARTITION when mcom_cindex_place starts placing them (mcom_cindex_partition), and a key's
// partition is a function of the key alone.  So the singletons' keys are sorted the same way -- { partition | home bits, tag | singleton |
// lane } tuples through the same two radix passes -- and a workgroup per partition joins the two lists in LDS: the partition's few
// thousand queries go into an LDS hash table keyed by the 28 bits (home bits, tag) the index would have compared, the partition's
// ~12 k entries stream past it, and every match is a candidate that is verified exactly as k_realign_reads verifies it (window, distance,
// the exact key, the "a lower dictionary sees the same window" rule, encode_byte).  What moves through HBM is streams (entries 12 bytes
// each, queries 12 bytes each, twice for their two passes) plus the verification's gathers, which the table design pays as well.
//
// Three things the table design did beside the lookups come for free here:
//   * the placement of the entries into the table (k_cx_assemble_sorted: 5 ms, 16.8 GB) is not needed;
//   * the dictionary screen (k_bin_screen: 4.8 ms of random atomics) -- "does any (dictionary, key) bin hold more than maxsearch
//     singletons" -- is answered inside the join: the queries of one bin sit in one partition's LDS table, a query counts the queries with
//     its own (home bits, tag, lane) there, an upper bound of its bin (two keys may share the 28 bits).  A count above maxsearch, a
//     partition with more queries than the LDS table takes, or more singletons than the tuple format numbers: *h_status = 1 and the caller
//     builds the table after all (mcom_cindex_assemble from the same sorted entries) and runs the passes as before;
//   * the LATER PASSES (preprocess.c:197-232: thr = e, e + S, ...) need no second look at the index: a (singleton, contig window) pair's
//     test depends on the threshold only through "distance <= thr" and the rule about which encode_byte tests apply (:393, :461), so the
//     first pass keeps every candidate that fails only for its distance (<= maxthr) as a DEFERRED tuple { claim key, read id, distance,
//     the two encode_byte answers }, and a later pass is a kernel over those tuples (mcom_realign_deferred) -- a few million at most:
//     a 17-mer hit that is not at the read's true place has a distance around 0.75 L, far above maxthr = L / 2.
// Same claims as mcom_realign_pass_reads, pass by pass (tests/test_gpu_realign.py compares the two on the reference's fixtures and on
// sets with repeats; the pipeline tests compare whole runs with the reference's dumps and the oracle).
// ====================================================================================================================================
#define RJ_THREADS 512                      // two workgroups per CU (76 KB of LDS each at LS = 13): one's loads travel while the other works in LDS
// LDS hash slots of a partition's queries: 2^LS of key + value (64 KB at LS = 13, 128 KB at 14: the host picks by the mean number of
// queries per partition); the table takes 0.69 of them
#define RJ_QUEUE 1280                       // candidates of a workgroup on their way to the list (a batch of 2048 entries adds ~430); 79 KB of LDS with the table: two workgroups per CU
#define RJ_CHUNK 2048u                      // candidates a workgroup reserves room for at a time
#define RJ_EMPTY 0xFFFFFFFFu

// the queries: one tuple per (singleton, lane) with a lane for every (direction, dictionary) the scan asks (kthread_hash_realign.c:440:
// not the reverse direction of a dictionary that starts at base 0), flagged singletons included (they count in the bins)
__global__ void k_rj_queries(CixGeom g, const uint64_t *__restrict__ sgbits, size_t n_sg, int W, uint32_t lane_mask, int n_lanes, int lg_lanes,
                             uint32_t *__restrict__ qkey, uint64_t *__restrict__ qslot)
{
	const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t sg = t >> lg_lanes;
	const int q = (int)(t & ((1u << lg_lanes) - 1u));
	if (sg >= n_sg || !((lane_mask >> q) & 1u)) return;
	const int dir = q / g.nd, l = q - dir * g.nd;
	const uint64_t *rb = sgbits + sg * (size_t)W;
	uint64_t key = bits_key(rb, g.ds[l], g.klen);
	const uint64_t kmask = (1ull << (2 * g.klen)) - 1;
	if (dir) key = (~(rev_groups(key) >> (64 - 2 * g.klen))) & kmask;
	uint32_t own, part, h16;
	cix_hash(key, 1u, g.n_parts, own, part, h16);
	const size_t at = sg * (size_t)n_lanes + (size_t)__popc(lane_mask & ((1u << q) - 1u));
	qkey[at] = (part << 16) | h16;
	qslot[at] = (cix_tag(key) << CIX_TAG_SHIFT) | ((uint64_t)sg << 5) | (uint64_t)q;
}

template <int LS> __device__ __forceinline__ uint32_t rj_hash(uint32_t kid) { return (kid * 0x9E3779B1u) >> (32 - LS); }

// One workgroup per partition joins; the matches leave as a list of candidates { entry's slot word, singleton << 5 | lane } that a
// second kernel verifies one per thread.  (The first form verified inside the join kernel: 1024 threads per workgroup cap a thread at
// 128 registers, the verification -- five-word window, its reverse complement, the mismatch words -- spilled 400 bytes per thread
// to scratch and the kernel took 37 ms.  Split, the join is an LDS kernel with a handful of registers and the verification runs at
// the occupancy its registers allow, as it does in k_realign_reads.)
template <int LS>
__global__ __launch_bounds__(RJ_THREADS) void k_rj_join(CixGeom g, const uint32_t *__restrict__ ekey, const unsigned long long *__restrict__ eslot,
                                                        const uint32_t *__restrict__ epstart, const uint32_t *__restrict__ qkey,
                                                        const unsigned long long *__restrict__ qslot, const uint32_t *__restrict__ qpstart,
                                                        uint32_t maxsearch, unsigned long long *__restrict__ cand_v, uint32_t *__restrict__ cand_q,
                                                        unsigned long long cand_cap, unsigned long long *__restrict__ cand_count, unsigned int *__restrict__ status)
{
	constexpr uint32_t RJ_SLOTS = 1u << LS, RJ_QMAX = (RJ_SLOTS / 16) * 11;
	__shared__ uint32_t K[RJ_SLOTS], V[RJ_SLOTS];
	__shared__ unsigned long long QC[RJ_QUEUE];
	__shared__ uint32_t QV[RJ_QUEUE];
	__shared__ uint32_t q_n, w_left;
	__shared__ unsigned long long q_base, w_base;
	const uint32_t part = blockIdx.x;
	const uint32_t q0 = qpstart[part], nq = qpstart[part + 1] - q0;
	const uint32_t e0 = epstart[part], ne = epstart[part + 1] - e0;
	const int tid = threadIdx.x;
	if (nq == 0) return;                                                          // nobody asks for a key of this partition
	if (nq > RJ_QMAX) { if (tid == 0) *status = 1u; return; }
	for (uint32_t i = tid; i < RJ_SLOTS; i += RJ_THREADS) K[i] = RJ_EMPTY;
	if (tid == 0) { q_n = 0; w_left = 0; w_base = 0; }
	__syncthreads();
	// (four tuples of a thread in flight at a time: with one, every round of the loop waited for its own pair of loads)
	for (uint32_t i0 = tid; i0 < nq; i0 += 4 * RJ_THREADS) {
		uint32_t k4[4]; unsigned long long s4[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) { const uint32_t i = i0 + u * RJ_THREADS; if (i < nq) { k4[u] = qkey[q0 + i]; s4[u] = qslot[q0 + i]; } }
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const uint32_t i = i0 + u * RJ_THREADS;
			if (i >= nq) break;
			const uint32_t kid = ((k4[u] & 0xFFFFu) << 12) | (uint32_t)(s4[u] >> CIX_TAG_SHIFT);
			uint32_t h = rj_hash<LS>(kid);
			while (atomicCAS(&K[h], RJ_EMPTY, kid) != RJ_EMPTY) h = (h + 1) & (RJ_SLOTS - 1);
			V[h] = (uint32_t)s4[u];                                                      // singleton << 5 | lane
		}
	}
	__syncthreads();
	// the screen: a forward query counts the queries of its own (28 bits, lane) -- at least its bin of dictionary l.  (Walking from the
	// slot a query hashes to covers every query with its 28 bits: they all lie in that run of occupied slots.)
	for (uint32_t i = tid; i < RJ_SLOTS; i += RJ_THREADS) {
		const uint32_t kid = K[i];
		if (kid == RJ_EMPTY) continue;
		const uint32_t code = V[i] & 31u;
		if ((int)code >= g.nd) continue;
		uint32_t same = 0;
		for (uint32_t h = rj_hash<LS>(kid); K[h] != RJ_EMPTY; h = (h + 1) & (RJ_SLOTS - 1)) same += (K[h] == kid && (V[h] & 31u) == code);
		if (same > maxsearch) *status = 1u;
	}
	// the queue leaves for the candidate list: one reservation per workgroup and hand-over
	// (room in the list is reserved RJ_CHUNK candidates at a time: one returning atomic on the list's counter per hand-over -- 600 000 of
	// them from 65 000 workgroups, each waited for between two barriers -- made this kernel 30 ms; what a workgroup leaves unused of
	// its last chunk stays marked empty and the verification skips it)
	auto drain = [&]() {
		__syncthreads();
		const uint32_t n = q_n < RJ_QUEUE ? q_n : RJ_QUEUE;
		if (tid == 0) {
			if (w_left < n) { w_base = atomicAdd(cand_count, (unsigned long long)RJ_CHUNK); w_left = RJ_CHUNK; }
			q_base = w_base; w_base += n; w_left -= n;
		}
		__syncthreads();
		const unsigned long long b0 = q_base;
		for (uint32_t c = tid; c < n; c += RJ_THREADS) if (b0 + c < cand_cap) { cand_v[b0 + c] = QC[c]; cand_q[b0 + c] = QV[c]; }
		__syncthreads();
		if (tid == 0) q_n = 0;
		__syncthreads();
	};
	// (a batch = four entries per thread, and the next batch's are loaded before this one is probed: the loop is a chain of
	// load -> LDS -> barrier otherwise, with sixteen waves on the CU to hide it)
	constexpr uint32_t RJ_BATCH = 4 * RJ_THREADS;
	uint32_t kn[4]; unsigned long long vn4[4];
#pragma unroll
	for (int u = 0; u < 4; ++u) { const uint32_t i = tid + u * RJ_THREADS; kn[u] = 0; vn4[u] = 0; if (i < ne) { kn[u] = ekey[e0 + i]; vn4[u] = eslot[e0 + i]; } }
	for (uint32_t base = 0; base < ne; base += RJ_BATCH) {
		uint32_t kc[4]; unsigned long long vc[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) { kc[u] = kn[u]; vc[u] = vn4[u]; }
#pragma unroll
		for (int u = 0; u < 4; ++u) { const uint32_t i = base + RJ_BATCH + tid + u * RJ_THREADS; if (i < ne) { kn[u] = ekey[e0 + i]; vn4[u] = eslot[e0 + i]; } }
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const uint32_t i = base + tid + u * RJ_THREADS;
			if (i >= ne) break;
			const unsigned long long v = vc[u];
			const uint32_t kid = ((kc[u] & 0xFFFFu) << 12) | (uint32_t)(v >> CIX_TAG_SHIFT);
			for (uint32_t h = rj_hash<LS>(kid), kk; (kk = K[h]) != RJ_EMPTY; h = (h + 1) & (RJ_SLOTS - 1)) {
				if (kk != kid) continue;
				const uint32_t pos = atomicAdd(&q_n, 1u);
				if (pos < RJ_QUEUE) { QC[pos] = v; QV[pos] = V[h]; }
				else {                                                                       // (the queue is full -- a key that thousands of singletons share: straight to the list)
					const unsigned long long at = atomicAdd(cand_count, 1ull);
					if (at < cand_cap) { cand_v[at] = v; cand_q[at] = V[h]; }
				}
			}
		}
		// (the queue is handed over after every second batch, whatever it holds: two batches add ~860 candidates on average, and a fixed
		// schedule needs no barrier to agree on)
		if ((base / RJ_BATCH) & 1u) drain();
	}
	drain();
}

// one candidate per thread: defer = { claim key, read id | distance << 32 | encode_byte forward ok << 40 | reverse ok << 41 }
template <int W>
__global__ __launch_bounds__(256) void k_rj_verify(CixGeom g, const unsigned long long *__restrict__ cand_v, const uint32_t *__restrict__ cand_q, size_t n_cand_in,
                                                   const uint64_t *__restrict__ sgbits, const uint32_t *__restrict__ rids,
                                                   const uint64_t *__restrict__ cbits, const ulonglong2 *__restrict__ cgeo,
                                                   int thr, int maxthr, unsigned long long *__restrict__ claim, unsigned long long *__restrict__ stats,
                                                   ulonglong2 *__restrict__ defer, unsigned long long defer_cap, unsigned long long *__restrict__ defer_count)
{
	// (measured and dropped: two launches over the list, the candidates of dictionary 0 first so that their claims drop the other
	// dictionaries' candidates for the same window by key alone -- 4.8 + 9.2 ms against 13.1 in one launch)
	const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t n_cand = 0, n_pass = 0;
	const int L = g.L;
	if (t < n_cand_in) do {
		const uint32_t qv = cand_q[t];
		if (qv == RJ_EMPTY) break;                                                           // (the unused end of a workgroup's last chunk)
		const uint32_t sg = qv >> 5;
		const int q = (int)(qv & 31u);
		const int dir = q / g.nd, l = q - dir * g.nd;
		const unsigned long long v = cand_v[t];
		const int off = dir ? L - g.ds[l] - g.klen : g.ds[l];
		const uint32_t c = (uint32_t)((v & ((1ull << CIX_TAG_SHIFT) - 1)) >> g.pbits);
		const int64_t jj = (int64_t)(v & ((1ull << g.pbits) - 1)) - off;
		if (jj < 0) break;
		// The claim key this candidate would bid is known before anything is loaded.  A read at its true place is found by most of its
		// dictionaries -- seven or eight candidates for ONE window, spread over the whole list (their keys hash to different partitions) --
		// and only the smallest key counts: a candidate whose key is not below the singleton's claim so far can change nothing.  (Its
		// deferred tuple is not needed either: a singleton with a claim leaves the list after this pass.)  One 8-byte gather instead of
		// the singleton's row, the contig's offsets and its window for about two candidates in three.
		const unsigned long long ck = ((unsigned long long)c << 33) | ((unsigned long long)jj << 5) | ((unsigned long long)dir << 4) | (unsigned long long)l;
		// (a flagged singleton -- near-poly-A / -T at this threshold -- carries the claim 0 while this kernel runs: no key is below it, and
		// its flag needs no gather of its own; k_rj_unflag puts UINT64_MAX back.  The kernel runs at the card's rate for random 64-byte
		// lines from a gigabyte: every line saved per candidate counts)
		if (__hip_atomic_load(&claim[sg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= ck) break;
		const ulonglong2 cg = cgeo[c];                                                       // { first word of the packed contig, its windows }
		if ((uint64_t)jj >= cg.y) break;
		++n_cand;
		uint64_t win[W], x[W];
		const uint64_t *rb = sgbits + (size_t)sg * W;
		contig_window<W>(cbits + cg.x, (uint64_t)jj, L, dir != 0, win);
		int dist = 0;
#pragma unroll
		for (int w = 0; w < W; ++w) { x[w] = win[w] ^ rb[w]; dist += __popcll(x[w]); }
		if (dist > maxthr || bits_key(x, g.ds[l], g.klen) != 0) break;                       // a tag is not the key: exact check here
		bool lower = false;
		for (int l2 = 0; l2 < l; ++l2)
			if ((!dir || g.ds[l2] > 0) && bits_key(x, g.ds[l2], g.klen) == 0) lower = true;   // a lower dictionary claims the same tuple with a smaller key
		if (lower) break;
		uint64_t mm[W];
#pragma unroll
		for (int w = 0; w < W; ++w) mm[w] = (x[w] | (x[w] >> 1)) & 0x5555555555555555ull;
		if (dist <= thr) {
			if (!dir) { if (!encode_ok_sparse<W>(mm, L, false)) break; }                          // :393
			else if (thr > 24 && !encode_ok_sparse<W>(mm, L, true)) break;                       // :461
			++n_pass;
			atomicMin(&claim[sg], ck);
			break;
		}
		// fails for its distance alone at this threshold: a later pass decides (the read may be claimed or flagged by then)
		const bool okf = !dir ? encode_ok_sparse<W>(mm, L, false) : true;
		const bool okr = dir ? encode_ok_sparse<W>(mm, L, true) : true;
		if (!dir && !okf) break;                                                            // (the forward test does not depend on the threshold)
		const unsigned long long at = atomicAdd(defer_count, 1ull);
		if (at < defer_cap) defer[at] = make_ulonglong2(ck, (unsigned long long)rids[sg] | ((unsigned long long)dist << 32) | ((unsigned long long)okf << 40) | ((unsigned long long)okr << 41));
	} while (0);
	if (stats) {
		__shared__ unsigned long long st2[2];
		if (threadIdx.x < 2) st2[threadIdx.x] = 0;
		__syncthreads();
		unsigned long long b = n_cand, c = n_pass;
		for (int o = 32; o; o >>= 1) { b += __shfl_xor(b, o); c += __shfl_xor(c, o); }
		if ((threadIdx.x & 63) == 0) { if (b) atomicAdd(&st2[0], b); if (c) atomicAdd(&st2[1], c); }
		__syncthreads();
		if (threadIdx.x < 2 && st2[threadIdx.x]) atomicAdd(&stats[4 * (blockIdx.x & 1023) + 1 + threadIdx.x], st2[threadIdx.x]);
	}
}
// flagged singletons: claim 0 while the verification runs (set = true), UINT64_MAX afterwards; and the contigs' { first word, windows }
__global__ void k_rj_unflag(const uint8_t *__restrict__ sgflag, size_t n_sg, bool set, unsigned long long *__restrict__ claim)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n_sg && sgflag[i]) claim[i] = set ? 0ull : ~0ull;
}
__global__ void k_rj_cgeo(const uint64_t *__restrict__ coff, const uint64_t *__restrict__ woff, uint32_t n, ulonglong2 *__restrict__ cgeo)
{
	const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c < n) cgeo[c] = make_ulonglong2(coff[c], woff[c + 1] - woff[c]);
}
// the lookups of the pass (statistics): queries of unflagged singletons
__global__ void k_rj_count(const uint8_t *__restrict__ sgflag, size_t n_sg, int n_lanes, unsigned long long *__restrict__ stats)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	unsigned long long v = (i < n_sg && !sgflag[i]) ? (unsigned long long)n_lanes : 0ull;
	for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
	if ((threadIdx.x & 63) == 0 && v) atomicAdd(&stats[4 * (blockIdx.x & 1023)], v);
}

__global__ void k_rj_map(const uint32_t *__restrict__ rids, size_t n, uint32_t *__restrict__ map)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) map[rids[i]] = (uint32_t)i;
}
__global__ void k_rj_deferred(const ulonglong2 *__restrict__ defer, size_t n, const uint32_t *__restrict__ map, const uint8_t *__restrict__ sgflag, int thr,
                              unsigned long long *__restrict__ claim, unsigned long long *__restrict__ stats)
{
	const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	unsigned long long np = 0;
	if (t < n) {
		const ulonglong2 d = defer[t];
		const uint32_t cur = map[(uint32_t)d.y];
		const int dist = (int)((d.y >> 32) & 0xFFull);
		const bool dir = (d.x >> 4) & 1ull;
		if (cur != 0xFFFFFFFFu && !sgflag[cur] && dist <= thr && (!dir || thr <= 24 || ((d.y >> 41) & 1ull))) { atomicMin(&claim[cur], d.x); np = 1; }
	}
	for (int o = 32; o; o >>= 1) np += __shfl_xor(np, o);
	if (stats && (threadIdx.x & 63) == 0 && np) atomicAdd(&stats[2], np);
}

extern "C" int mcom_realign_join(mcom_ctx *ctx, uint64_t geom, const uint32_t *d_ekey, const uint64_t *d_eslot, const uint32_t *d_epstart,
                                 const uint64_t *d_sgbits, const uint8_t *d_sgflag, const uint32_t *d_rids, size_t n_sg, const uint64_t *d_cbits,
                                 const uint64_t *d_coff, const uint64_t *d_woff, uint32_t n_contigs, int L, int ininumdict, int thr, int maxthr, int maxsearch,
                                 uint64_t *d_claim, uint64_t *d_stats, uint64_t *d_defer, uint64_t defer_cap, uint64_t *h_n_defer, int *h_status)
{
	if (!ctx || !h_n_defer || !h_status) return MCOM_E_ARG;
	*h_n_defer = 0; *h_status = 0;
	CixGeom g;
	if (L < 1 || L > 256 || cix_geom(L, ininumdict, g)) return mcom_fail(ctx, MCOM_E_ARG, "bad dictionary layout");
	cix_unpack(geom, g);
	g.pbits = cix_pbits(n_contigs);
	if (g.n_owners != 1) return mcom_fail(ctx, MCOM_E_ARG, "the join works on a whole index: one share");
	if (n_sg) MCOM_HIP(ctx, hipMemsetAsync(d_claim, 0xFF, n_sg * 8, ctx->stream));
	if (d_stats) MCOM_HIP(ctx, hipMemsetAsync(d_stats, 0, 3 * 8, ctx->stream));
	if (n_contigs == 0 || n_sg == 0) return MCOM_OK;
	if (!d_ekey || !d_eslot || !d_epstart || !d_sgbits || !d_sgflag || !d_rids || !d_cbits || !d_coff || !d_woff || !d_claim || (defer_cap && !d_defer) || maxsearch < 1 || maxthr < thr || maxthr > 255)
		return mcom_fail(ctx, MCOM_E_ARG, "bad join arguments");
	if (n_sg >= (1ull << 27) || 2 * g.nd > 32) { *h_status = 1; return MCOM_OK; }                  // (the tuple holds singleton << 5 | lane in 32 bits)
	const int W = mcom_words_per_read(L);
	uint32_t lane_mask = 0; int n_lanes = 0;
	for (int q = 0; q < 2 * g.nd; ++q) { const int dir = q / g.nd, l = q - dir * g.nd; if (!(dir && g.ds[l] <= 0)) { lane_mask |= 1u << q; ++n_lanes; } }
	const uint64_t nq = (uint64_t)n_sg * (uint64_t)n_lanes;
	if (nq >= (1ull << 32)) { *h_status = 1; return MCOM_OK; }
	auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
	const size_t key_b = al((size_t)nq * 4 + 4), slot_b = al((size_t)nq * 8 + 8), ps_b = al(((size_t)g.n_parts + 2) * 4);
	char *tmp = nullptr;
	if (mcom_dmalloc(&tmp, 2 * key_b + 2 * slot_b + ps_b + 4096 * 8 + 512) != hipSuccess) return mcom_fail(ctx, MCOM_E_NOMEM, "join: %zu bytes of query tuples", 2 * key_b + 2 * slot_b);
	struct Guard { mcom_ctx *c; char *p; ~Guard() { (void)hipStreamSynchronize(c->stream); mcom_dfree(p); } } guard{ctx, tmp};
	uint32_t *qkA = (uint32_t*)tmp, *qkB = (uint32_t*)(tmp + key_b);
	uint64_t *qsA = (uint64_t*)(tmp + 2 * key_b), *qsB = (uint64_t*)(tmp + 2 * key_b + slot_b);
	uint32_t *qps = (uint32_t*)(tmp + 2 * key_b + 2 * slot_b);
	unsigned long long *sets = (unsigned long long*)(tmp + 2 * key_b + 2 * slot_b + ps_b);          // 1024 x 4 statistics words, then the deferred count and the status
	unsigned long long *d_cnt = sets + 4096;
	unsigned int *d_status = (unsigned int*)(d_cnt + 1);
	MCOM_HIP(ctx, hipMemsetAsync(sets, 0, 4096 * 8 + 32, ctx->stream));            // (+ deferred count, status, candidate count)
	{
		const int lg_lanes = 2 * g.nd <= 16 ? 4 : 5;
		const uint64_t blocks = ((n_sg << lg_lanes) + 255) / 256;
		if (blocks >= (1ull << 31)) { *h_status = 1; return MCOM_OK; }
		McomProfScope ps_(ctx, PROF_REALIGN_READS);
		MCOM_LAUNCH(k_rj_queries, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, g, d_sgbits, n_sg, W, lane_mask, n_lanes, lg_lanes, qkA, qsA);
		MCOM_LAUNCH_CHECK(ctx);
	}
	const uint32_t *qk = nullptr; const uint64_t *qs = nullptr;
	int rc = mcom_cindex_partition(ctx, qkA, qsA, nq, 0, qkB, qsB, L, ininumdict, geom, qps, &qk, &qs);
	if (rc) return rc;
	// the candidate list: room for 1.5 x the queries (chunks are not filled to the end; more candidates than that -- a repeat whose key
	// thousands of singletons and thousands of contig positions share -- and the table route takes over), marked empty
	const unsigned long long ccap = nq + nq / 2 + RJ_CHUNK;
	char *ctmp = nullptr;
	if (mcom_dmalloc(&ctmp, al((size_t)ccap * 8) + al((size_t)ccap * 4) + 256) != hipSuccess) return mcom_fail(ctx, MCOM_E_NOMEM, "join: candidate list");
	Guard cguard{ctx, ctmp};
	unsigned long long *cand_v = (unsigned long long*)ctmp;
	uint32_t *cand_q = (uint32_t*)(ctmp + al((size_t)ccap * 8));
	MCOM_HIP(ctx, hipMemsetAsync(cand_q, 0xFF, (size_t)ccap * 4, ctx->stream));
	unsigned long long *d_ccnt = d_cnt + 2;
	{
		McomProfScope ps_(ctx, PROF_REALIGN_READS);
		const bool wide = nq / g.n_parts + 1 > 4600;                                      // (mean queries per partition: the small table takes 5632)
#define MCOM_RJ_ARGS g, d_ekey, (const unsigned long long*)d_eslot, d_epstart, qk, (const unsigned long long*)qs, (const uint32_t*)qps, (uint32_t)maxsearch, cand_v, cand_q, ccap, d_ccnt, d_status
		if (wide) MCOM_LAUNCH((k_rj_join<14>), dim3(g.n_parts), dim3(RJ_THREADS), 0, ctx->stream, MCOM_RJ_ARGS);
		else MCOM_LAUNCH((k_rj_join<13>), dim3(g.n_parts), dim3(RJ_THREADS), 0, ctx->stream, MCOM_RJ_ARGS);
#undef MCOM_RJ_ARGS
		MCOM_LAUNCH_CHECK(ctx);
	}
	unsigned long long hcc[3] = {0, 0, 0};
	MCOM_HIP(ctx, mcom_d2h_async(ctx, hcc, d_cnt, 24));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	if ((unsigned int)(hcc[1] & 0xFFFFFFFFull) || hcc[2] > ccap) { *h_status = 1; return MCOM_OK; }
	const size_t n_cand = (size_t)hcc[2];
	if (n_cand) {
		ulonglong2 *cgeo = nullptr;
		if (mcom_dmalloc(&cgeo, ((size_t)n_contigs + 1) * 16) != hipSuccess) return mcom_fail(ctx, MCOM_E_NOMEM, "join: contig geometry");
		struct G2 { mcom_ctx *c; ulonglong2 *p; ~G2() { (void)hipStreamSynchronize(c->stream); mcom_dfree(p); } } g2{ctx, cgeo};
		MCOM_LAUNCH(k_rj_cgeo, dim3((n_contigs + 255) / 256), dim3(256), 0, ctx->stream, d_coff, d_woff, n_contigs, cgeo);
		MCOM_LAUNCH(k_rj_unflag, dim3((unsigned)((n_sg + 255) / 256)), dim3(256), 0, ctx->stream, d_sgflag, n_sg, true, (unsigned long long*)d_claim);
		McomProfScope ps_(ctx, PROF_REALIGN_READS);
		const unsigned vb = (unsigned)((n_cand + 255) / 256);
#define MCOM_CASE(WW) case WW: MCOM_LAUNCH((k_rj_verify<WW>), dim3(vb), dim3(256), 0, ctx->stream, g, (const unsigned long long*)cand_v, (const uint32_t*)cand_q, n_cand, d_sgbits, d_rids, \
		d_cbits, (const ulonglong2*)cgeo, thr, maxthr, (unsigned long long*)d_claim, d_stats ? sets : nullptr, (ulonglong2*)d_defer, (unsigned long long)defer_cap, d_cnt); break;
		switch (W) { MCOM_CASE(1) MCOM_CASE(2) MCOM_CASE(3) MCOM_CASE(4) MCOM_CASE(5) MCOM_CASE(6) MCOM_CASE(7) MCOM_CASE(8)
		default: return mcom_fail(ctx, MCOM_E_ARG, "unsupported read length"); }
#undef MCOM_CASE
		MCOM_LAUNCH(k_rj_unflag, dim3((unsigned)((n_sg + 255) / 256)), dim3(256), 0, ctx->stream, d_sgflag, n_sg, false, (unsigned long long*)d_claim);
		MCOM_LAUNCH_CHECK(ctx);
		MCOM_HIP(ctx, mcom_stream_sync(ctx));                                              // (cgeo goes back to the pool)
	}
	if (d_stats) MCOM_LAUNCH(k_rj_count, dim3((unsigned)((n_sg + 255) / 256)), dim3(256), 0, ctx->stream, d_sgflag, n_sg, n_lanes, sets);
	if (d_stats) MCOM_LAUNCH(k_stats_fold, dim3(1), dim3(3), 0, ctx->stream, sets, (unsigned long long*)d_stats);
	unsigned long long hc[2] = {0, 0};
	MCOM_HIP(ctx, mcom_d2h_async(ctx, hc, d_cnt, 16));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	*h_n_defer = hc[0];
	*h_status = (unsigned int)(hc[1] & 0xFFFFFFFFull) ? 1 : 0;
	if (hc[0] > defer_cap) *h_status = 1;                                                           // (more deferred candidates than the caller has room for)
	return MCOM_OK;
}

extern "C" int mcom_realign_deferred(mcom_ctx *ctx, const uint64_t *d_defer, uint64_t n_defer, const uint32_t *d_rids, const uint8_t *d_sgflag, size_t n_sg,
                                     uint32_t *d_map, size_t n_reads, int thr, uint64_t *d_claim, uint64_t *d_stats)
{
	if (!ctx) return MCOM_E_ARG;
	if (n_sg) MCOM_HIP(ctx, hipMemsetAsync(d_claim, 0xFF, n_sg * 8, ctx->stream));
	if (d_stats) MCOM_HIP(ctx, hipMemsetAsync(d_stats, 0, 3 * 8, ctx->stream));
	if (n_sg == 0 || n_defer == 0) return MCOM_OK;
	if (!d_defer || !d_rids || !d_sgflag || !d_map || !d_claim) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	MCOM_HIP(ctx, hipMemsetAsync(d_map, 0xFF, n_reads * 4, ctx->stream));
	MCOM_LAUNCH(k_rj_map, dim3((unsigned)((n_sg + 255) / 256)), dim3(256), 0, ctx->stream, d_rids, n_sg, d_map);
	MCOM_LAUNCH(k_rj_deferred, dim3((unsigned)((n_defer + 255) / 256)), dim3(256), 0, ctx->stream, (const ulonglong2*)d_defer, (size_t)n_defer, (const uint32_t*)d_map, d_sgflag, thr,
	            (unsigned long long*)d_claim, (unsigned long long*)d_stats);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}
