// minicom_amd/csrc/cindex.hip -- the Stage-2 contig index, built by radix partitioning (no scattered insert).
//
// What it is for (realign.hip): every klen-mer of the Stage-2 contigs, findable by key, so that a
// singleton looks up its own 2*nd - 1 keys instead of every contig window asking every dictionary.
//
// Structure: a multi-map of lines of 8 words (64 bytes): word 0 = number of entries in the line (| 0x100 when entries
// were pushed past it), words 1-7 = entries tag12 | contig | position (52 bits between them, cindex.hpp).  The table is cut into partitions of equal size
// (a few thousand lines); a key hashes to a partition and to a HOME line inside it -- an address computed from the key
// alone -- and its entries lie in the home line and, when that is full, in the lines behind it (wrapping inside the
// partition): bucketed linear probing, placed exactly: entries in home order take consecutive slot positions,
// pos = max(7 * home, previous + 1).  A home line with more entries than a few lines' worth (a repeat: 10^5 copies of one
// key) keeps them out of the partition, in a run of lines of its own in the extension area behind the table, and says where.
//
// Build.  Round 1 inserted entry by entry into the whole 17 GB table: one random 64-byte line read and written per entry,
// DRAM-random bound (2 TB/s of line traffic for 0.25 of the roofline by sectors, 0.13 by words).  Now everything streams:
//   1. two radix passes split the entries {partition | home bits, slot word} (12 bytes, two arrays) by partition: the
//      first pass computes them straight from the packed contigs (one thread per contig position), tiles of 4096, LDS-staged
//      so that every digit's run leaves the tile in one piece; the second is the stable pass of sort.hip on two arrays;
//   2. the starts of the partitions in the sorted arrays are found;
//   3. one workgroup per partition places its entries: home counts (LDS atomics), the carry into every line (what the
//      lines before it could not hold: a max-plus scan, wrapped once round the partition), slot positions, and writes
//      the partition's lines -- a region of ~150 KB that one workgroup fills within microseconds, so the writes meet in
//      L2 and every line goes to HBM once.  No memset, no atomics in HBM.
#include "cindex.hpp"
#include <algorithm>

#define CX_THREADS 256
#define CX_ITEMS 24                         // 6144 entries per tile: 16 gave 36.6 ms for the benchmark index, 24 35.6, 8 41.7, 32 45.3 (one workgroup per CU)
#define CX_TILE (CX_THREADS * CX_ITEMS)

// ---- position space (as before): contig c owns [woff[c] + maxoff*c, woff[c+1] + maxoff*(c+1)); positions p = 0 .. nw-1+maxoff
// of a contig with nw > 0 windows are indexed.  first_contig[b] = the contig that owns position 256*b (counted from pos0).
__global__ void k_cindex_blocks(int maxoff, const uint64_t *__restrict__ woff, uint32_t c0, uint32_t c1, uint64_t pos0, uint64_t n_blocks, uint32_t *__restrict__ first_contig)
{
	const uint32_t c = c0 + blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= c1) return;
	const uint64_t a = woff[c] + (uint64_t)maxoff * c - pos0, b = woff[c + 1] + (uint64_t)maxoff * (c + 1) - pos0;
	for (uint64_t blk = (a + 255) >> 8; (blk << 8) < b && blk < n_blocks; ++blk) first_contig[blk] = c;
}

// Workgroups are dealt to the 8 XCDs round robin, and every XCD has an L2 of its own.  Tiles that follow each other write runs that
// follow each other (the same digit's region), and a run of ~24 entries ends inside a 128-byte line: with tile = block, the two
// halves of such a line are written through two different L2s.  So XCD x takes a contiguous eighth of the tiles: the halves meet
// in one L2 and leave it as whole lines.  t >= n_tiles: a spare workgroup (the eighths are rounded up).
__device__ __forceinline__ uint32_t cx_tile(uint32_t b, uint32_t n_tiles) { const uint32_t per = (n_tiles + 7u) >> 3; return (b & 7u) * per + (b >> 3); }
static inline uint32_t cx_grid(uint32_t n_tiles) { return ((n_tiles + 7u) >> 3) << 3; }

struct CxSrc { const uint64_t *cbits, *coff, *woff; const uint32_t *first_contig; uint32_t c1; uint64_t pos0, n_pos; unsigned long long *head; };

// ---- walking the position space.  A thread owns CX_ITEMS CONSECUTIVE positions: it finds its contig once, loads three words of
// the packed string and then rolls the klen-mer two bits per position (round 3 looked every position up on its own: contig search,
// offsets and two word loads per position, twice -- histogram and scatter; the round-4 profile gave 3.6 + 9.9 ms for the two).
struct CxWalk { uint32_t c; bool has; uint64_t p, np, lo, hi; };
__device__ __forceinline__ uint64_t cx_first(const CixGeom &g, const CxSrc &s, uint32_t c) { return s.woff[c] + (uint64_t)g.maxoff * c - s.pos0; }
__device__ __forceinline__ void cx_load(const CixGeom &g, const CxSrc &s, CxWalk &w)
{
	w.has = s.woff[w.c + 1] != s.woff[w.c];
	if (!w.has || w.p >= w.np) return;
	const uint64_t len = (s.woff[w.c + 1] - s.woff[w.c]) + (uint64_t)g.L - 1;     // the contig's length: its words are ceil(2 len / 64) + 1
	const uint64_t q = (2 * len + 63) >> 6, i = (2 * w.p) >> 6;
	const uint64_t *src = s.cbits + s.coff[w.c] + i;
	const int sh = (int)((2 * w.p) & 63);
	const uint64_t w0 = src[0], w1 = src[1], w2 = i + 2 <= q ? src[2] : 0ull;      // (bits past the contig's own words are never part of a key)
	w.lo = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
	w.hi = sh ? (w1 >> sh) | (w2 << (64 - sh)) : w1;
}
__device__ __forceinline__ void cx_seek(const CixGeom &g, const CxSrc &s, uint64_t gi, CxWalk &w)
{
	uint32_t c = s.first_contig[gi >> 8];
	while (c + 1 < s.c1 && cx_first(g, s, c + 1) <= gi) ++c;
	const uint64_t a = cx_first(g, s, c);
	w.c = c; w.p = gi - a; w.np = cx_first(g, s, c + 1) - a;
	cx_load(g, s, w);
}
// the entry of the walk's position (false: a contig without windows), then one position on
// digit: what pass 1 splits by -- the HIGH byte of the partition, or (multi-GPU) the share that owns the key
__device__ __forceinline__ bool cx_step(const CixGeom &g, const CxSrc &s, CxWalk &w, uint32_t &key32, uint64_t &slot, uint32_t &digit)
{
	while (w.p >= w.np) {                                                           // next contig (rare: a contig holds hundreds of positions)
		++w.c; w.p = 0; w.np = cx_first(g, s, w.c + 1) - cx_first(g, s, w.c);
		cx_load(g, s, w);
	}
	bool ok = w.has;
	if (ok && (w.p >> g.pbits)) { if (*(volatile unsigned long long*)(s.head + 1) == 0) s.head[1] = 1; ok = false; }   // reported by the build
	if (ok) {
		const uint64_t key = w.lo & ((1ull << (2 * g.klen)) - 1);
		uint32_t own, part, h16;
		cix_hash(key, g.n_owners, g.n_parts, own, part, h16);
		key32 = (part << 16) | h16;
		digit = g.n_owners > 1 ? own : (part >> 8);
		slot = (cix_tag(key) << CIX_TAG_SHIFT) | ((uint64_t)w.c << g.pbits) | w.p;
	}
	w.lo = (w.lo >> 2) | (w.hi << 62); w.hi >>= 2; ++w.p;
	return ok;
}

// pass 1, histogram: digit = high byte of the partition (owner of the key when the index is shared out)
__global__ __launch_bounds__(CX_THREADS) void k_cx_hist1(CixGeom g, CxSrc s, uint32_t *__restrict__ hist, uint32_t nblocks)
{
	__shared__ uint32_t h[256];
	const uint32_t tile = cx_tile(blockIdx.x, nblocks);
	if (tile >= nblocks) return;
	h[threadIdx.x] = 0;
	__syncthreads();
	uint64_t gi = (uint64_t)tile * CX_TILE + (uint64_t)threadIdx.x * CX_ITEMS;
	if (gi < s.n_pos) {
		CxWalk w; cx_seek(g, s, gi, w);
#pragma unroll 4
		for (int it = 0; it < CX_ITEMS && gi < s.n_pos; ++it, ++gi) {
			uint32_t k32, dg; uint64_t sl;
			if (cx_step(g, s, w, k32, sl, dg)) atomicAdd(&h[dg], 1u);
		}
	}
	__syncthreads();
	hist[(size_t)threadIdx.x * nblocks + tile] = h[threadIdx.x];
}

// pass 1, scatter: the tile's entries grouped by digit in LDS (order inside a digit is free), every digit's run written in one piece
// OWNERS: the index is shared out (digit = owner, kept beside the staged entry: it is not a function of the staged key; the extra
// LDS would cost the one-share kernel its second workgroup per CU)
template <bool OWNERS>
__global__ __launch_bounds__(CX_THREADS) void k_cx_scatter1(CixGeom g, CxSrc s, const uint32_t *__restrict__ offs, uint32_t nblocks,
                                                            uint32_t *__restrict__ out_key, uint64_t *__restrict__ out_slot)
{
	__shared__ uint64_t st_slot[CX_TILE];
	__shared__ uint32_t st_key[CX_TILE];
	__shared__ uint8_t st_dig[OWNERS ? CX_TILE : 4];
	__shared__ uint32_t cnt[256], start[256], gofs[256], wsum[CX_THREADS / 64];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const uint32_t tile = cx_tile(blockIdx.x, nblocks);
	if (tile >= nblocks) return;
	cnt[tid] = 0;
	__syncthreads();
	uint64_t gi = (uint64_t)tile * CX_TILE + (uint64_t)tid * CX_ITEMS;
	uint32_t k32[CX_ITEMS], rank[CX_ITEMS]; uint64_t sl[CX_ITEMS]; bool ok[CX_ITEMS]; uint8_t dg[CX_ITEMS];
	{
		CxWalk w; w.c = 0; w.has = false; w.p = w.np = w.lo = w.hi = 0;
		if (gi < s.n_pos) cx_seek(g, s, gi, w);
#pragma unroll
		for (int it = 0; it < CX_ITEMS; ++it, ++gi) {
			uint32_t d = 0;
			ok[it] = gi < s.n_pos && cx_step(g, s, w, k32[it], sl[it], d);
			dg[it] = (uint8_t)d;
			rank[it] = ok[it] ? atomicAdd(&cnt[d], 1u) : 0u;
		}
	}
	__syncthreads();
	{
		const uint32_t tot = cnt[tid];
		uint32_t v = tot;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(v, d, 64); if (lane >= d) v += t; }
		if (lane == 63) wsum[wv] = v;
		__syncthreads();
		uint32_t add = 0;
		for (int q = 0; q < wv; ++q) add += wsum[q];
		start[tid] = v + add - tot;
		gofs[tid] = offs[(size_t)tid * nblocks + tile];
	}
	__syncthreads();
#pragma unroll
	for (int it = 0; it < CX_ITEMS; ++it) if (ok[it]) {
		const uint32_t at = start[dg[it]] + rank[it];
		st_key[at] = k32[it]; st_slot[at] = sl[it];
		if (OWNERS) st_dig[at] = dg[it];
	}
	__syncthreads();
	const uint32_t total = start[255] + cnt[255];
	for (uint32_t q = tid; q < total; q += CX_THREADS) {
		const uint32_t d = OWNERS ? (uint32_t)st_dig[q] : st_key[q] >> 24;
		const size_t o = (size_t)gofs[d] + (q - start[d]);
		out_key[o] = st_key[q]; out_slot[o] = st_slot[q];
	}
}

// ---- passes over the entry arrays.  MSD order (round 4): first by the high byte of the partition (pass 1 above, or -- for entries
// that arrive in any order -- the flat pass REGION = false here), then INSIDE each of those 256 regions by the low byte (REGION =
// true: tiles never straddle two regions).  No pass needs to be stable (the order inside a partition is free), so ranks come from
// LDS atomics as in pass 1 (round 3's second pass was the LSD-stable one: eight ballots per entry), and the starts of the
// partitions fall out of the second pass's scanned histogram (no search pass).
// rs[257]: first entry of every region; tp[257]: first tile of every region (tiles of CX_TILE entries, the last of a region short)
struct CxTiles { uint32_t rs[257], tp[257]; };
__global__ void k_cx_regions(const uint32_t *__restrict__ key, uint32_t n, CxTiles *__restrict__ T)
{
	__shared__ uint32_t tl[257];
	const uint32_t v = threadIdx.x;                                                    // 0 .. 256: first index whose high byte is >= v
	uint32_t lo = 0, hi = n;
	if (v >= 256) lo = n;
	else while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if ((key[mid] >> 24) < v) lo = mid + 1; else hi = mid; }
	T->rs[v] = lo; tl[v] = lo;
	__syncthreads();
	if (v == 0) {
		uint32_t run = 0;
		for (int r = 0; r < 256; ++r) { T->tp[r] = run; run += (tl[r + 1] - tl[r] + CX_TILE - 1) / CX_TILE; }
		T->tp[256] = run;
	}
}
// where a block of the region passes works: region r, tile t of it, entries [first, first + count); false: a spare block
struct CxWhere { uint32_t r, t, nt, first, count; };
__device__ __forceinline__ bool cx_where(const CxTiles *__restrict__ T, uint32_t b, uint32_t *sh_tp, CxWhere &w)
{
	for (uint32_t q = threadIdx.x; q < 257; q += blockDim.x) sh_tp[q] = T->tp[q];
	__syncthreads();
	if (b >= sh_tp[256]) return false;
	uint32_t lo = 0, hi = 255;                                                         // the last region whose first tile is <= b (empty regions share a first tile with the next)
	while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (sh_tp[mid] <= b) lo = mid; else hi = mid - 1; }
	// (tp[lo + 1] > b >= tp[lo]: the region is not empty)
	w.r = lo; w.t = b - sh_tp[lo]; w.nt = sh_tp[lo + 1] - sh_tp[lo];
	const uint32_t a = T->rs[lo], e = T->rs[lo + 1];
	w.first = a + w.t * CX_TILE;
	w.count = e - w.first < (uint32_t)CX_TILE ? e - w.first : (uint32_t)CX_TILE;
	return true;
}
template <bool REGION>
__global__ __launch_bounds__(CX_THREADS) void k_cx_hist2(const uint32_t *__restrict__ key, uint32_t n, const CxTiles *__restrict__ T, uint32_t *__restrict__ hist, uint32_t nblocks)
{
	__shared__ uint32_t h[256], sh_tp[257];
	const uint32_t b = cx_tile(blockIdx.x, nblocks);
	if (b >= nblocks) return;
	h[threadIdx.x] = 0;
	uint32_t first, count; size_t at;
	if (REGION) {
		CxWhere w;
		if (!cx_where(T, b, sh_tp, w)) { hist[(size_t)256 * b + threadIdx.x] = 0; return; }   // spare tiles clear the spare slots
		first = w.first; count = w.count; at = (size_t)256 * sh_tp[w.r] + (size_t)threadIdx.x * w.nt + w.t;
	} else {
		__syncthreads();
		first = b * (uint32_t)CX_TILE; count = n - first < (uint32_t)CX_TILE ? n - first : (uint32_t)CX_TILE;
		at = (size_t)threadIdx.x * nblocks + b;
	}
#pragma unroll 4
	for (int it = 0; it < CX_ITEMS; ++it) {
		const uint32_t i = (uint32_t)it * CX_THREADS + threadIdx.x;
		if (i < count) atomicAdd(&h[(key[first + i] >> (REGION ? 16 : 24)) & 255u], 1u);
	}
	__syncthreads();
	hist[at] = h[threadIdx.x];
}
template <bool REGION>
__global__ __launch_bounds__(CX_THREADS) void k_cx_scatter2(const uint32_t *__restrict__ in_key, const uint64_t *__restrict__ in_slot, uint32_t n, const CxTiles *__restrict__ T,
                                                            const uint32_t *__restrict__ offs, uint32_t nblocks, uint32_t *__restrict__ out_key, uint64_t *__restrict__ out_slot)
{
	__shared__ uint64_t st_slot[CX_TILE];
	__shared__ uint32_t st_key[CX_TILE];
	__shared__ uint32_t cnt[256], start[256], gofs[256], wsum[CX_THREADS / 64], sh_tp[257];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	constexpr int SHIFT = REGION ? 16 : 24;
	const uint32_t b = cx_tile(blockIdx.x, nblocks);
	if (b >= nblocks) return;
	cnt[tid] = 0;
	uint32_t first, count; size_t at;
	if (REGION) {
		CxWhere w;
		if (!cx_where(T, b, sh_tp, w)) return;
		first = w.first; count = w.count; at = (size_t)256 * sh_tp[w.r] + (size_t)tid * w.nt + w.t;
	} else {
		__syncthreads();
		first = b * (uint32_t)CX_TILE; count = n - first < (uint32_t)CX_TILE ? n - first : (uint32_t)CX_TILE;
		at = (size_t)tid * nblocks + b;
	}
	uint32_t k32[CX_ITEMS], rank[CX_ITEMS]; uint64_t sl[CX_ITEMS];
#pragma unroll
	for (int it = 0; it < CX_ITEMS; ++it) {
		const uint32_t i = (uint32_t)it * CX_THREADS + tid;
		if (i < count) { k32[it] = in_key[first + i]; sl[it] = in_slot[first + i]; }
	}
#pragma unroll
	for (int it = 0; it < CX_ITEMS; ++it) {
		const uint32_t i = (uint32_t)it * CX_THREADS + tid;
		rank[it] = i < count ? atomicAdd(&cnt[(k32[it] >> SHIFT) & 255u], 1u) : 0u;
	}
	__syncthreads();
	{
		const uint32_t tot = cnt[tid];
		uint32_t v = tot;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(v, d, 64); if (lane >= d) v += t; }
		if (lane == 63) wsum[wv] = v;
		__syncthreads();
		uint32_t add = 0;
		for (int q = 0; q < wv; ++q) add += wsum[q];
		start[tid] = v + add - tot;
		gofs[tid] = offs[at];
	}
	__syncthreads();
#pragma unroll
	for (int it = 0; it < CX_ITEMS; ++it) {
		const uint32_t i = (uint32_t)it * CX_THREADS + tid;
		if (i < count) { const uint32_t p = start[(k32[it] >> SHIFT) & 255u] + rank[it]; st_key[p] = k32[it]; st_slot[p] = sl[it]; }
	}
	__syncthreads();
	for (uint32_t q = tid; q < count; q += CX_THREADS) {
		const uint32_t k = st_key[q], d = (k >> SHIFT) & 255u;
		const size_t o = (size_t)gofs[d] + (q - start[d]);
		out_key[o] = k; out_slot[o] = st_slot[q];
	}
}

// first entry of every partition, from the scanned histogram of the region pass: partition v = region v >> 8, digit v & 255
__global__ void k_cx_pstart(const CxTiles *__restrict__ T, const uint32_t *__restrict__ offs, uint32_t n, uint32_t n_parts, uint32_t *__restrict__ pstart)
{
	const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
	if (v > n_parts) return;
	const uint32_t r = v >> 8, d = v & 255u;
	uint32_t at = n;
	if (v < n_parts && r < 256) {
		const uint32_t nt = T->tp[r + 1] - T->tp[r];
		at = nt ? offs[(size_t)256 * T->tp[r] + (size_t)d * nt] : T->rs[r];
	}
	pstart[v] = at;
}
// ---- placement: one workgroup per partition ------------------------------------------------------------------------------
#define CA_THREADS 512
#define CA_BATCH 8
__global__ __launch_bounds__(CA_THREADS) void k_cx_assemble(const uint32_t *__restrict__ key, const uint64_t *__restrict__ slot, const uint32_t *__restrict__ pstart,
                                                            uint32_t n_parts, uint32_t NL, unsigned long long *__restrict__ head, uint64_t ext_cap,
                                                            const uint8_t *__restrict__ redo)
{
	extern __shared__ uint32_t sm[];
	if (!redo[blockIdx.x]) return;                              // k_cx_assemble_sorted did this partition
	uint32_t *cnt = sm;                                     // [NL] entries per home line, later the running counter of the placement
	int32_t *carry = (int32_t*)(sm + NL);                   // [NL] entries carried into the line
	uint32_t *runb = sm + 2 * (size_t)NL;                   // [NL] first extension line of a heavy home line, 0xFFFFFFFF = light
	__shared__ int32_t agg_sum[CA_THREADS / 64], agg_min[CA_THREADS / 64];
	__shared__ uint32_t red[CA_THREADS / 64];
	__shared__ int32_t s_c0;
	const uint32_t part = blockIdx.x;
	const uint32_t s0 = pstart[part], n = pstart[part + 1] - s0;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	unsigned long long *lines = head + CIX_HEAD_WORDS;
	unsigned long long *L0 = lines + (size_t)part * NL * 8;
	unsigned long long *X0 = lines + (size_t)n_parts * NL * 8;                  // extension area
	for (uint32_t l = tid; l < NL; l += CA_THREADS) { cnt[l] = 0; runb[l] = 0xFFFFFFFFu; }
	__syncthreads();
	// (the loads of a batch are issued together: with one load in flight per thread the kernel waited on HBM latency)
	for (uint32_t i0 = tid; i0 < n; i0 += CA_THREADS * CA_BATCH) {
		uint32_t kk[CA_BATCH];
#pragma unroll
		for (int q = 0; q < CA_BATCH; ++q) { const uint32_t i = i0 + q * CA_THREADS; kk[q] = i < n ? key[s0 + i] : 0u; }
#pragma unroll
		for (int q = 0; q < CA_BATCH; ++q) if (i0 + q * CA_THREADS < n) atomicAdd(&cnt[cix_home(kk[q] & 0xFFFFu, NL)], 1u);
	}
	__syncthreads();
	// heavy home lines: more than T entries; T falls until the light ones fit the partition with room to spare
	uint32_t T = 8 * CIX_WAYS;
	const uint32_t room = (uint32_t)(((uint64_t)CIX_WAYS * NL * 15) / 16);
	for (;;) {
		uint32_t sum = 0;
		for (uint32_t l = tid; l < NL; l += CA_THREADS) sum += cnt[l] <= T ? cnt[l] : 0u;
		for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o);
		if (lane == 0) red[wv] = sum;
		__syncthreads();
		sum = 0;
		for (int q = 0; q < CA_THREADS / 64; ++q) sum += red[q];
		__syncthreads();
		if (sum <= room || T == 0) break;
		T >>= 1;
	}
	for (uint32_t l = tid; l < NL; l += CA_THREADS) if (cnt[l] > T) {
		const uint32_t rl = (cnt[l] + CIX_WAYS - 1) / CIX_WAYS;
		const unsigned long long at = atomicAdd(&head[0], (unsigned long long)rl);
		if (at + rl <= ext_cap && rl < (1u << 22) && at + rl < (1ull << 32)) {
			runb[l] = (uint32_t)at;
			for (uint32_t j = 0; j < rl; ++j) X0[(size_t)(at + j) * 8] = (unsigned long long)std::min(CIX_WAYS, cnt[l] - j * CIX_WAYS);
		} else runb[l] = 0xFFFFFFFEu;                                          // no room: the caller sees head[0] > ext_cap and builds again with more
	}
	__syncthreads();
	// carry into line l = what the lines before it could not hold: x_0 = 0, x_{l+1} = max(0, x_l + c_l - 7), c = light counts.  With S =
	// prefix sums of (c - 7): x_l = S_{l-1} - min_{j <= l} S_{j-1}; wrapped once round the partition: x'_l = max(x_l, x_NL + S_{l-1}).
	auto light = [&](uint32_t l) -> int32_t { return runb[l] == 0xFFFFFFFFu ? (int32_t)cnt[l] : 0; };
	const uint32_t per = (NL + CA_THREADS - 1) / CA_THREADS;
	const uint32_t lo = std::min(NL, (uint32_t)tid * per), hi = std::min(NL, lo + per);
	int32_t sum = 0, mn = 0;
	for (uint32_t l = lo; l < hi; ++l) { mn = l == lo ? 0 : std::min(mn, sum); sum += light(l) - (int32_t)CIX_WAYS; }
	// exclusive scan over the threads of the pairs (sum, min of the running sums before a line): (s1, m1) o (s2, m2) = (s1 + s2, min(m1, s1 + m2))
	int32_t before = 0, minbefore = 0;
	{
		int32_t ps = sum, pm = mn;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const int32_t s1 = __shfl_up(ps, d, 64), m1 = __shfl_up(pm, d, 64);
			if (lane >= d) { pm = std::min(m1, s1 + pm); ps = s1 + ps; }
		}
		if (lane == 63) { agg_sum[wv] = ps; agg_min[wv] = pm; }
		__syncthreads();
		int32_t ws = 0, wm = 0;                                                  // everything before this wave
		for (int q = 0; q < wv; ++q) { wm = std::min(wm, ws + agg_min[q]); ws += agg_sum[q]; }
		const int32_t is = ws + ps, im = std::min(wm, ws + pm);                  // inclusive at this thread
		const int32_t es = __shfl_up(is, 1, 64), em = __shfl_up(im, 1, 64);      // exclusive: the thread before (the wave's prefix for lane 0)
		before = lane ? es : ws;
		minbefore = std::min(0, lane ? em : wm);
	}
	{
		int32_t S = before, M = std::min(minbefore, before);
		for (uint32_t l = lo; l < hi; ++l) { M = std::min(M, S); carry[l] = S - M; S += light(l) - (int32_t)CIX_WAYS; }
		if (hi == NL && lo < hi) s_c0 = std::max(0, S - std::min(M, S));          // carry out of the last line
	}
	__syncthreads();
	const int32_t c0 = s_c0;
	if (c0 > 0) {
		int32_t S = before;
		for (uint32_t l = lo; l < hi; ++l) { carry[l] = std::max(carry[l], c0 + S); S += light(l) - (int32_t)CIX_WAYS; }
	}
	__syncthreads();
	// word 0 of every line, then the entries: the k-th entry of light home h takes slot position 7 h + carry[h] + k, the k-th of
	// a heavy one slot k of its run
	for (uint32_t l = tid; l < NL; l += CA_THREADS) {
		const int32_t have = carry[l] + light(l);
		unsigned long long w0 = (unsigned long long)std::min<int32_t>(have, (int32_t)CIX_WAYS) | (have > (int32_t)CIX_WAYS ? CIX_MORE : 0ull);
		if (runb[l] < 0xFFFFFFFEu) w0 |= CIX_HEAVY | ((unsigned long long)((cnt[l] + CIX_WAYS - 1) / CIX_WAYS) << 10) | ((unsigned long long)runb[l] << 32);
		L0[(size_t)l * 8] = w0;
	}
	__syncthreads();
	for (uint32_t l = tid; l < NL; l += CA_THREADS) cnt[l] = 0;
	__syncthreads();
	for (uint32_t i0 = tid; i0 < n; i0 += CA_THREADS * CA_BATCH) {
		uint32_t kk[CA_BATCH]; uint64_t ss[CA_BATCH];
#pragma unroll
		for (int q = 0; q < CA_BATCH; ++q) { const uint32_t i = i0 + q * CA_THREADS; kk[q] = i < n ? key[s0 + i] : 0u; ss[q] = i < n ? slot[s0 + i] : 0ull; }
#pragma unroll
		for (int q = 0; q < CA_BATCH; ++q) if (i0 + q * CA_THREADS < n) {
			const uint32_t h = cix_home(kk[q] & 0xFFFFu, NL);
			const uint32_t k = atomicAdd(&cnt[h], 1u);
			const uint32_t rb = runb[h];
			if (rb == 0xFFFFFFFFu) {
				const uint32_t pos = CIX_WAYS * h + (uint32_t)carry[h] + k;
				uint32_t line = pos / CIX_WAYS;
				const uint32_t sl = pos - line * CIX_WAYS;
				if (line >= NL) line -= NL;
				L0[(size_t)line * 8 + 1 + sl] = ss[q];
			} else if (rb != 0xFFFFFFFEu) X0[((size_t)rb + k / CIX_WAYS) * 8 + 1 + k % CIX_WAYS] = ss[q];
		}
	}
}

// ---- placement, sorted: the same result with every line written whole ------------------------------------------------------
// k_cx_assemble stores every slot word where it belongs: 13 000 eight-byte stores to 13 000 different lines per partition, and the
// store path of a CU takes one line per transaction whatever its size -- that, not HBM, bounded it.  Here the slot words of
// a partition are first sorted by home line in LDS (a counting sort: the counts are there anyway); since slot positions rise
// with the home line, line l then holds the sorted entries [first(l) - carry(l), ... ) and ONE thread writes its 64 bytes:
// a wave writes 4 KB in a row.  Heavy homes (rare) keep the scattered stores.  Partitions with more entries or lines than the
// LDS arrays hold, or with many heavy homes, are left to k_cx_assemble (redo list).
#define CS_THREADS 1024
#define CS_CAP 14336                        // entries of a partition whose slot words fit in LDS
#define CS_NL 4096                          // lines
#define CS_HEAVY 32
__global__ __launch_bounds__(CS_THREADS) void k_cx_assemble_sorted(const uint32_t *__restrict__ key, const uint64_t *__restrict__ slot, const uint32_t *__restrict__ pstart,
                                                                   uint32_t n_parts, uint32_t NL, unsigned long long *__restrict__ head, uint64_t ext_cap,
                                                                   uint8_t *__restrict__ redo, uint32_t cap)
{
	__shared__ unsigned long long sorted[CS_CAP];
	__shared__ uint32_t cnt[CS_NL];                         // entries per home line, later the running counter of the sort
	__shared__ uint16_t first[CS_NL + 1];                   // start of the home line's light entries in `sorted`
	__shared__ uint16_t carry[CS_NL];                       // entries carried into the line
	__shared__ int32_t agg_sum[CS_THREADS / 64], agg_min[CS_THREADS / 64];
	__shared__ uint32_t red[CS_THREADS / 64], hv_line[CS_HEAVY], hv_base[CS_HEAVY], hv_cnt[CS_HEAVY];
	__shared__ uint32_t n_heavy;
	__shared__ int32_t s_c0;
	const uint32_t part = blockIdx.x;
	const uint32_t s0 = pstart[part], n = pstart[part + 1] - s0;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	if (n > cap || NL > CS_NL) { if (tid == 0) redo[part] = 1; return; }
	unsigned long long *lines = head + CIX_HEAD_WORDS;
	unsigned long long *L0 = lines + (size_t)part * NL * 8;
	unsigned long long *X0 = lines + (size_t)n_parts * NL * 8;
	// every entry of the partition is loaded NOW: 14 keys and 14 slot words per thread, all in flight at once; the keys are used
	// twice (counts, sort) and the slot words only after the scans, so their latency hides behind everything in between (round 3
	// loaded key by key and slot by slot inside the two loops, and the keys twice)
	constexpr int CS_PER = CS_CAP / CS_THREADS;
	uint32_t kk[CS_PER]; uint64_t ss[CS_PER];
#pragma unroll
	for (int q = 0; q < CS_PER; ++q) { const uint32_t i = (uint32_t)q * CS_THREADS + tid; kk[q] = i < n ? key[s0 + i] : 0u; }
#pragma unroll
	for (int q = 0; q < CS_PER; ++q) { const uint32_t i = (uint32_t)q * CS_THREADS + tid; ss[q] = i < n ? slot[s0 + i] : 0ull; }
	for (uint32_t l = tid; l < NL; l += CS_THREADS) cnt[l] = 0;
	if (tid == 0) n_heavy = 0;
	__syncthreads();
#pragma unroll
	for (int q = 0; q < CS_PER; ++q) { if ((uint32_t)q * CS_THREADS + tid < n) { kk[q] = cix_home(kk[q] & 0xFFFFu, NL); atomicAdd(&cnt[kk[q]], 1u); } }   // (kk = home line from here on)
	__syncthreads();
	uint32_t T = 8 * CIX_WAYS;
	const uint32_t room = (uint32_t)(((uint64_t)CIX_WAYS * NL * 15) / 16);
	for (;;) {
		uint32_t sum = 0;
		for (uint32_t l = tid; l < NL; l += CS_THREADS) sum += cnt[l] <= T ? cnt[l] : 0u;
		for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o);
		if (lane == 0) red[wv] = sum;
		__syncthreads();
		sum = 0;
		for (int q = 0; q < CS_THREADS / 64; ++q) sum += red[q];
		__syncthreads();
		if (sum <= room || T == 0) break;
		T >>= 1;
	}
	for (uint32_t l = tid; l < NL; l += CS_THREADS) if (cnt[l] > T) {
		const uint32_t at = atomicAdd(&n_heavy, 1u);
		if (at < CS_HEAVY) { hv_line[at] = l; hv_cnt[at] = cnt[l]; }
	}
	__syncthreads();
	if (n_heavy > CS_HEAVY) { if (tid == 0) redo[part] = 1; return; }
	for (uint32_t q = tid; q < n_heavy; q += CS_THREADS) {
		const uint32_t rl = (hv_cnt[q] + CIX_WAYS - 1) / CIX_WAYS;
		const unsigned long long at = atomicAdd(&head[0], (unsigned long long)rl);
		if (at + rl <= ext_cap && rl < (1u << 22) && at + rl < (1ull << 32)) {
			hv_base[q] = (uint32_t)at;
			for (uint32_t j = 0; j < rl; ++j) X0[(size_t)(at + j) * 8] = (unsigned long long)std::min(CIX_WAYS, hv_cnt[q] - j * CIX_WAYS);
		} else hv_base[q] = 0xFFFFFFFEu;
		cnt[hv_line[q]] = 0;                                                    // its entries are not the partition's
	}
	__syncthreads();
	// per thread a chunk of lines: sums for `first` (prefix of the light counts) and for the carry (max-plus scan, see k_cx_assemble)
	const uint32_t per = (NL + CS_THREADS - 1) / CS_THREADS;
	const uint32_t lo = std::min(NL, (uint32_t)tid * per), hi = std::min(NL, lo + per);
	int32_t sum = 0, mn = 0; uint32_t tot = 0;
	for (uint32_t l = lo; l < hi; ++l) { mn = l == lo ? 0 : std::min(mn, sum); sum += (int32_t)cnt[l] - (int32_t)CIX_WAYS; tot += cnt[l]; }
	int32_t before = 0, minbefore = 0; uint32_t tbefore = 0;
	{
		int32_t ps = sum, pm = mn; uint32_t pt = tot;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const int32_t s1 = __shfl_up(ps, d, 64), m1 = __shfl_up(pm, d, 64); const uint32_t t1 = __shfl_up(pt, d, 64);
			if (lane >= d) { pm = std::min(m1, s1 + pm); ps = s1 + ps; pt += t1; }
		}
		if (lane == 63) { agg_sum[wv] = ps; agg_min[wv] = pm; red[wv] = pt; }
		__syncthreads();
		int32_t ws = 0, wm = 0; uint32_t wt = 0;
		for (int q = 0; q < wv; ++q) { wm = std::min(wm, ws + agg_min[q]); ws += agg_sum[q]; wt += red[q]; }
		const int32_t is = ws + ps, im = std::min(wm, ws + pm); const uint32_t it = wt + pt;
		const int32_t es = __shfl_up(is, 1, 64), em = __shfl_up(im, 1, 64); const uint32_t et = __shfl_up(it, 1, 64);
		before = lane ? es : ws;
		minbefore = std::min(0, lane ? em : wm);
		tbefore = lane ? et : wt;
		if (tid == CS_THREADS - 1) first[NL] = (uint16_t)it;
	}
	{
		int32_t S = before, M = std::min(minbefore, before); uint32_t f = tbefore;
		for (uint32_t l = lo; l < hi; ++l) { M = std::min(M, S); carry[l] = (uint16_t)(S - M); first[l] = (uint16_t)f; S += (int32_t)cnt[l] - (int32_t)CIX_WAYS; f += cnt[l]; }
		if (hi == NL && lo < hi) s_c0 = std::max(0, S - std::min(M, S));
	}
	__syncthreads();
	const int32_t c0 = s_c0;
	if (c0 > 0) {
		int32_t S = before;
		for (uint32_t l = lo; l < hi; ++l) { carry[l] = (uint16_t)std::max((int32_t)carry[l], c0 + S); S += (int32_t)cnt[l] - (int32_t)CIX_WAYS; }
	}
	__syncthreads();
	for (uint32_t l = tid; l < NL; l += CS_THREADS) cnt[l] = 0;
	__syncthreads();
	// the counting sort of the slot words; a heavy home's entries go straight to their run
#pragma unroll
	for (int u = 0; u < CS_PER; ++u) if ((uint32_t)u * CS_THREADS + tid < n) {
		const uint32_t h = kk[u];
		const uint32_t k = atomicAdd(&cnt[h], 1u);
		uint32_t hq = CS_HEAVY;
		for (uint32_t q = 0; q < n_heavy; ++q) if (hv_line[q] == h) hq = q;
		if (hq == CS_HEAVY) sorted[first[h] + k] = ss[u];
		else if (hv_base[hq] != 0xFFFFFFFEu) X0[((size_t)hv_base[hq] + k / CIX_WAYS) * 8 + 1 + k % CIX_WAYS] = ss[u];
	}
	__syncthreads();
	// four lanes per line, sixteen bytes each: a wave's store is one run of 1 KB (round 3 let one thread write its line's four
	// quarters one after the other -- 64 lanes, 64 lines, four partial-line stores each)
	const uint32_t nlight = first[NL];
	const uint32_t j2 = (uint32_t)(tid & 3) * 2;                                   // this lane's two words of the line: j2, j2 + 1
	for (uint32_t l = (uint32_t)tid >> 2; l < NL; l += CS_THREADS / 4) {
		const uint32_t cl = (uint32_t)first[l + 1] - first[l];
		const uint32_t have = (uint32_t)carry[l] + cl, occ = std::min(have, CIX_WAYS);
		int32_t f = (int32_t)first[l] - (int32_t)carry[l];
		if (f < 0) f += (int32_t)nlight;                                          // entries carried round the end of the partition
		unsigned long long w[2];
#pragma unroll
		for (uint32_t e = 0; e < 2; ++e) {
			const uint32_t word = j2 + e;
			if (word == 0) {
				w[e] = (unsigned long long)occ | (have > CIX_WAYS ? CIX_MORE : 0ull);
				for (uint32_t q = 0; q < n_heavy; ++q) if (hv_line[q] == l && hv_base[q] != 0xFFFFFFFEu)
					w[e] |= CIX_HEAVY | ((unsigned long long)((hv_cnt[q] + CIX_WAYS - 1) / CIX_WAYS) << 10) | ((unsigned long long)hv_base[q] << 32);
			} else {
				const uint32_t j = word - 1;
				uint32_t at = (uint32_t)f + j;
				if (at >= nlight) at -= nlight;
				w[e] = j < occ ? sorted[at] : 0ull;
			}
		}
		*(ulonglong2*)(L0 + (size_t)l * 8 + j2) = make_ulonglong2(w[0], w[1]);
	}
}

// ---- host ----------------------------------------------------------------------------------------------------------------
// sizes for `ranks` equal shares of the one index over a set with n_windows windows (ranks = 1: the whole index)
extern "C" int mcom_cindex_plan_shared(uint64_t n_windows, uint32_t n_contigs, int L, int ininumdict, int ranks, int rank, uint64_t *n_entries, uint64_t *n_share,
                                       uint64_t *geom, uint64_t *n_words)
{
	CixGeom g;
	if (L < 1 || L > 256 || cix_geom(L, ininumdict, g) || ranks < 1 || ranks > (int)CIX_MAX_OWNERS || rank < 0 || rank >= ranks) return MCOM_E_ARG;
	const uint64_t ne = n_windows + (uint64_t)n_contigs * (uint64_t)g.maxoff;          // an upper bound: contigs without windows hold no entry
	// a share receives the keys of 1 / ranks of the hash range: its expected number of entries, with room for the spread
	const uint64_t share = ranks == 1 ? ne : ne / (uint64_t)ranks + ne / (uint64_t)ranks / 16 + 65536;
	if (share >= (1ull << 32)) return MCOM_E_ARG;
	uint64_t P = share / 12288;                                                        // ~12 k entries = ~3500 lines = ~220 KB per partition
	if (P < 1) P = 1;
	if (P > CIX_MAX_PARTS) P = CIX_MAX_PARTS;
	uint64_t NL = (2 * ((share + P - 1) / P) + 6) / 7;                                 // entries / 3.5: half full -- a lookup nearly always ends in its home line (round 4, measured at 0.73 full: placement 5.2 -> 4.4 ms, but the lookups of the passes 19.0 -> 24.9 ms)
	if (NL < 16) NL = 16;
	if (NL > CIX_MAX_LINES) return MCOM_E_ARG;                                         // the counters of a partition live in LDS (2.75 G entries per share)
	const uint64_t ext = std::max<uint64_t>(1024, P * NL / 16);
	if ((P * NL + ext) >= (1ull << 32)) return MCOM_E_ARG;
	if (n_entries) *n_entries = ne;
	if (n_share) *n_share = share;
	if (geom) *geom = cix_pack((uint32_t)P, (uint32_t)NL, (uint32_t)ranks, (uint32_t)rank);
	if (n_words) *n_words = CIX_HEAD_WORDS + 8 * (P * NL + ext);
	return MCOM_OK;
}
extern "C" int mcom_cindex_plan(uint64_t n_windows, uint32_t n_contigs, int L, int ininumdict, uint64_t *n_entries, uint64_t *geom, uint64_t *n_words)
{
	return mcom_cindex_plan_shared(n_windows, n_contigs, L, ininumdict, 1, 0, n_entries, nullptr, geom, n_words);
}

static inline size_t cx_al(size_t b) { return (b + 255) & ~(size_t)255; }

// Step 1 of the build: the entries of contigs [c0, c1) -- { partition << 16 | home bits } in d_key, slot words in d_slot, room for
// `cap` each -- grouped by the share that owns their key (h_counts[q] entries for share q, in share order); with one share they
// are grouped by the HIGH byte of their partition instead, which is the first radix pass of mcom_cindex_place.
extern "C" int mcom_cindex_entries(mcom_ctx *ctx, const uint64_t *d_cbits, const uint64_t *d_coff, const uint64_t *d_woff, uint32_t n_contigs,
                                   uint32_t c0, uint32_t c1, int L, int ininumdict, uint64_t geom, uint32_t *d_key, uint64_t *d_slot, uint64_t cap,
                                   uint64_t *h_counts)
{
	if (!ctx || !h_counts) return MCOM_E_ARG;
	CixGeom g;
	if (L < 1 || L > 256 || cix_geom(L, ininumdict, g)) return mcom_fail(ctx, MCOM_E_ARG, "bad dictionary layout");
	cix_unpack(geom, g);
	for (uint32_t q = 0; q < g.n_owners; ++q) h_counts[q] = 0;
	if (g.n_parts < 1 || g.n_lines < 1) return mcom_fail(ctx, MCOM_E_ARG, "bad contig index geometry");
	if (c0 > c1 || c1 > n_contigs) return mcom_fail(ctx, MCOM_E_ARG, "bad contig range");
	g.pbits = cix_pbits(n_contigs);
	if (c0 == c1) return MCOM_OK;
	if (!d_cbits || !d_coff || !d_woff || !d_key || !d_slot) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	McomProfScope ps_(ctx, PROF_CINDEX_BUILD);
	uint64_t w01[2] = {0, 0};
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &w01[0], d_woff + c0, 8));
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &w01[1], d_woff + c1, 8));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	const uint64_t n_pos = (w01[1] - w01[0]) ? (w01[1] - w01[0]) + (uint64_t)(c1 - c0) * (uint64_t)g.maxoff : 0;
	if (n_pos >= (1ull << 32)) return mcom_fail(ctx, MCOM_E_ARG, "too many contig positions");
	if (n_pos > cap) return mcom_fail(ctx, MCOM_E_OVERFLOW, "contig index entries: room for %llu, %llu positions", (unsigned long long)cap, (unsigned long long)n_pos);
	if (!n_pos) return MCOM_OK;
	const uint32_t nblocks = (uint32_t)((n_pos + CX_TILE - 1) / CX_TILE);
	const uint64_t blocks256 = (n_pos + 255) / 256;
	const size_t hist_b = cx_al((size_t)256 * nblocks * 4 + 64), scr_b = cx_al(mcom_scan_scratch_elems((size_t)256 * nblocks + 2) * 4 + 1024), map_b = cx_al((size_t)blocks256 * 4 + 4);
	char *tmp = nullptr;
	if (mcom_dmalloc(&tmp, hist_b + scr_b + map_b + 2 * 256) != hipSuccess) return mcom_fail(ctx, MCOM_E_NOMEM, "contig index: scratch");
	struct Guard { mcom_ctx *c; char *p; ~Guard() { (void)hipStreamSynchronize(c->stream); mcom_dfree(p); } } guard{ctx, tmp};
	uint32_t *hist = (uint32_t*)tmp, *scr = (uint32_t*)(tmp + hist_b), *first_contig = (uint32_t*)(tmp + hist_b + scr_b);
	unsigned long long *head = (unsigned long long*)(tmp + hist_b + scr_b + map_b);   // word 1: a contig too long for the position field
	MCOM_HIP(ctx, hipMemsetAsync(head, 0, 16, ctx->stream));
	const uint64_t pos0 = w01[0] + (uint64_t)g.maxoff * c0;
	MCOM_LAUNCH(k_cindex_blocks, dim3((c1 - c0 + 255) / 256), dim3(256), 0, ctx->stream, g.maxoff, d_woff, c0, c1, pos0, blocks256, first_contig);
	MCOM_LAUNCH_CHECK(ctx);
	const CxSrc src{d_cbits, d_coff, d_woff, first_contig, c1, pos0, n_pos, head};
	MCOM_LAUNCH(k_cx_hist1, dim3(cx_grid(nblocks)), dim3(CX_THREADS), 0, ctx->stream, g, src, hist, nblocks);
	MCOM_LAUNCH_CHECK(ctx);
	MCOM_HIP(ctx, hipMemsetAsync(hist + (size_t)256 * nblocks, 0, 4, ctx->stream));
	int rc;
	if ((rc = mcom_scan_u32(ctx, hist, hist, (size_t)256 * nblocks + 1, scr))) return rc;        // the extra element becomes the number of entries
	if (g.n_owners > 1) MCOM_LAUNCH(k_cx_scatter1<true>, dim3(cx_grid(nblocks)), dim3(CX_THREADS), 0, ctx->stream, g, src, hist, nblocks, d_key, d_slot);
	else MCOM_LAUNCH(k_cx_scatter1<false>, dim3(cx_grid(nblocks)), dim3(CX_THREADS), 0, ctx->stream, g, src, hist, nblocks, d_key, d_slot);
	MCOM_LAUNCH_CHECK(ctx);
	// the first entry of every share = the scanned count of (digit q, block 0)
	std::vector<uint32_t> st(g.n_owners + 1, 0);
	if (g.n_owners > 1) { for (uint32_t q = 1; q < g.n_owners; ++q) MCOM_HIP(ctx, hipMemcpyAsync(&st[q], hist + (size_t)q * nblocks, 4, hipMemcpyDeviceToHost, ctx->stream)); }
	MCOM_HIP(ctx, hipMemcpyAsync(&st[g.n_owners], hist + (size_t)256 * nblocks, 4, hipMemcpyDeviceToHost, ctx->stream));
	uint64_t hd[2] = {0, 0};
	MCOM_HIP(ctx, hipMemcpyAsync(hd, head, 16, hipMemcpyDeviceToHost, ctx->stream));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	if (hd[1]) return mcom_fail(ctx, MCOM_E_ARG, "contig index: a contig of this set of %u contigs is longer than 2^%d bases", n_contigs, g.pbits);
	for (uint32_t q = 0; q < g.n_owners; ++q) h_counts[q] = (uint64_t)st[q + 1] - st[q];
	return MCOM_OK;
}

// Step 2a: n_ent entries in any order (grouped = 0) or grouped by the partition's HIGH byte (grouped != 0, what mcom_cindex_entries
// leaves with one share) -> sorted by partition, in two passes (one when grouped).  d_key / d_slot are overwritten, d_key_tmp / d_slot_tmp
// are scratch of n_ent entries each; the sorted arrays are one of the two pairs (*d_key_sorted, *d_slot_sorted) and the entries of
// partition v are [d_pstart[v], d_pstart[v + 1]) (d_pstart: n_parts + 1 words on the device).  Round 5: a step of its own, because the
// Stage-2 join (realign_join.hip) works on the sorted entries themselves and never needs the table of step 2b.
extern "C" int mcom_cindex_partition(mcom_ctx *ctx, uint32_t *d_key, uint64_t *d_slot, uint64_t n_ent64, int grouped, uint32_t *d_key_tmp, uint64_t *d_slot_tmp,
                                     int L, int ininumdict, uint64_t geom, uint32_t *d_pstart, const uint32_t **d_key_sorted, const uint64_t **d_slot_sorted)
{
	if (!ctx || !d_key_sorted || !d_slot_sorted) return MCOM_E_ARG;
	CixGeom g;
	if (L < 1 || L > 256 || cix_geom(L, ininumdict, g)) return mcom_fail(ctx, MCOM_E_ARG, "bad dictionary layout");
	cix_unpack(geom, g);
	if (!d_pstart || g.n_parts < 1 || g.n_lines < 1) return mcom_fail(ctx, MCOM_E_ARG, "bad contig index buffers");
	if (n_ent64 >= (1ull << 32)) return mcom_fail(ctx, MCOM_E_ARG, "too many contig index entries for one share");
	if (n_ent64 && (!d_key || !d_slot || !d_key_tmp || !d_slot_tmp)) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	const uint32_t n_ent = (uint32_t)n_ent64;
	McomProfScope ps_(ctx, PROF_CINDEX_BUILD);
	const uint32_t nbA = std::max<uint32_t>(1, (uint32_t)(((size_t)n_ent + CX_TILE - 1) / CX_TILE));   // tiles of the flat pass
	const uint32_t nbR = nbA + 256;                                                                       // at most: every region ends in a short tile
	const size_t hist_b = cx_al((size_t)256 * nbR * 4 + 64), scr_b = cx_al(mcom_scan_scratch_elems((size_t)256 * nbR + 2) * 4 + 1024);
	char *tmp = nullptr;
	if (mcom_dmalloc(&tmp, hist_b + scr_b + cx_al(sizeof(CxTiles)) + 256) != hipSuccess) return mcom_fail(ctx, MCOM_E_NOMEM, "contig index: scratch");
	struct Guard { mcom_ctx *c; char *p; ~Guard() { (void)hipStreamSynchronize(c->stream); mcom_dfree(p); } } guard{ctx, tmp};
	uint32_t *hist = (uint32_t*)tmp, *scr = (uint32_t*)(tmp + hist_b);
	CxTiles *tiles = (CxTiles*)(tmp + hist_b + scr_b);
	const uint32_t *skey = d_key; const uint64_t *sslot = d_slot;                         // the arrays that end up sorted by partition
	int rc;
	uint32_t *ik = d_key, *ok = d_key_tmp; uint64_t *is = d_slot, *os = d_slot_tmp;
	if (n_ent && !grouped) {                                                              // any order -> by the partition's high byte
		MCOM_LAUNCH(k_cx_hist2<false>, dim3(cx_grid(nbA)), dim3(CX_THREADS), 0, ctx->stream, ik, n_ent, (const CxTiles*)nullptr, hist, nbA);
		MCOM_LAUNCH_CHECK(ctx);
		if ((rc = mcom_scan_u32(ctx, hist, hist, (size_t)256 * nbA, scr))) return rc;
		MCOM_LAUNCH(k_cx_scatter2<false>, dim3(cx_grid(nbA)), dim3(CX_THREADS), 0, ctx->stream, ik, is, n_ent, (const CxTiles*)nullptr, hist, nbA, ok, os);
		MCOM_LAUNCH_CHECK(ctx);
		std::swap(ik, ok); std::swap(is, os);
	}
	// inside every region by the partition's low byte
	MCOM_LAUNCH(k_cx_regions, dim3(1), dim3(257), 0, ctx->stream, ik, n_ent, tiles);
	MCOM_LAUNCH_CHECK(ctx);
	if (n_ent) {
		MCOM_LAUNCH(k_cx_hist2<true>, dim3(cx_grid(nbR)), dim3(CX_THREADS), 0, ctx->stream, ik, n_ent, (const CxTiles*)tiles, hist, nbR);
		MCOM_LAUNCH_CHECK(ctx);
		if ((rc = mcom_scan_u32(ctx, hist, hist, (size_t)256 * nbR, scr))) return rc;
		MCOM_LAUNCH(k_cx_scatter2<true>, dim3(cx_grid(nbR)), dim3(CX_THREADS), 0, ctx->stream, ik, is, n_ent, (const CxTiles*)tiles, hist, nbR, ok, os);
		MCOM_LAUNCH_CHECK(ctx);
		std::swap(ik, ok); std::swap(is, os);
		skey = ik; sslot = is;
	}
	MCOM_LAUNCH(k_cx_pstart, dim3((g.n_parts + 1 + 255) / 256), dim3(256), 0, ctx->stream, (const CxTiles*)tiles, hist, n_ent, g.n_parts, d_pstart);
	MCOM_LAUNCH_CHECK(ctx);
	*d_key_sorted = skey; *d_slot_sorted = sslot;
	return MCOM_OK;                                                                     // (the guard waits for the stream: the scratch is in use until then)
}

// Step 2b: the table from entries sorted by partition (step 2a): one workgroup per partition.
extern "C" int mcom_cindex_assemble(mcom_ctx *ctx, const uint32_t *d_key_sorted, const uint64_t *d_slot_sorted, const uint32_t *d_pstart, int L, int ininumdict,
                                    uint64_t geom, uint64_t *d_keys, uint64_t n_words)
{
	if (!ctx) return MCOM_E_ARG;
	CixGeom g;
	if (L < 1 || L > 256 || cix_geom(L, ininumdict, g)) return mcom_fail(ctx, MCOM_E_ARG, "bad dictionary layout");
	cix_unpack(geom, g);
	const uint64_t main_lines = (uint64_t)g.n_parts * g.n_lines;
	if (!d_keys || !d_pstart || g.n_parts < 1 || g.n_lines < 1 || n_words < CIX_HEAD_WORDS + 8 * (main_lines + 1)) return mcom_fail(ctx, MCOM_E_ARG, "bad contig index buffers");
	const uint64_t ext_cap = (n_words - CIX_HEAD_WORDS) / 8 - main_lines;
	McomProfScope ps_(ctx, PROF_CINDEX_BUILD);
	MCOM_HIP(ctx, hipMemsetAsync(d_keys, 0, CIX_HEAD_WORDS * 8, ctx->stream));
	uint8_t *redo = nullptr;
	if (mcom_dmalloc(&redo, (size_t)g.n_parts + 256) != hipSuccess) return mcom_fail(ctx, MCOM_E_NOMEM, "contig index: scratch");
	struct Guard { mcom_ctx *c; uint8_t *p; ~Guard() { (void)hipStreamSynchronize(c->stream); mcom_dfree(p); } } guard{ctx, redo};
	const size_t lds = (size_t)3 * g.n_lines * 4;
	if (lds > 150 * 1024) return mcom_fail(ctx, MCOM_E_ARG, "contig index: partitions of %u lines", g.n_lines);
	MCOM_HIP(ctx, hipMemsetAsync(redo, 0, g.n_parts, ctx->stream));
	MCOM_LAUNCH(k_cx_assemble_sorted, dim3(g.n_parts), dim3(CS_THREADS), 0, ctx->stream, d_key_sorted, d_slot_sorted, d_pstart, g.n_parts, g.n_lines, (unsigned long long*)d_keys, ext_cap, redo,
	                   ctx->cix_cap_set ? std::min<uint32_t>(ctx->cix_cap, CS_CAP) : (uint32_t)CS_CAP);
	MCOM_LAUNCH_CHECK(ctx);
	MCOM_HIP(ctx, hipFuncSetAttribute((const void*)k_cx_assemble, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	MCOM_LAUNCH(k_cx_assemble, dim3(g.n_parts), dim3(CA_THREADS), lds, ctx->stream, d_key_sorted, d_slot_sorted, d_pstart, g.n_parts, g.n_lines, (unsigned long long*)d_keys, ext_cap, redo);
	MCOM_LAUNCH_CHECK(ctx);
	uint64_t used = 0;
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &used, d_keys, 8));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	if (used > ext_cap) return mcom_fail(ctx, MCOM_E_OVERFLOW, "contig index: %llu extension lines needed, room for %llu", (unsigned long long)used, (unsigned long long)ext_cap);
	return MCOM_OK;
}

// Step 2 = 2a + 2b: this share's table from its n_ent entries (what mcom_cindex_entries made, or what the other ranks sent: any order).
extern "C" int mcom_cindex_place(mcom_ctx *ctx, uint32_t *d_key, uint64_t *d_slot, uint64_t n_ent64, int grouped, uint32_t *d_key_tmp, uint64_t *d_slot_tmp,
                                 int L, int ininumdict, uint64_t geom, uint64_t *d_keys, uint64_t n_words)
{
	if (!ctx) return MCOM_E_ARG;
	CixGeom g;
	if (L < 1 || L > 256 || cix_geom(L, ininumdict, g)) return mcom_fail(ctx, MCOM_E_ARG, "bad dictionary layout");
	cix_unpack(geom, g);
	const uint64_t main_lines = (uint64_t)g.n_parts * g.n_lines;
	if (!d_keys || g.n_parts < 1 || g.n_lines < 1 || n_words < CIX_HEAD_WORDS + 8 * (main_lines + 1)) return mcom_fail(ctx, MCOM_E_ARG, "bad contig index buffers");
	uint32_t *pstart = nullptr;
	if (mcom_dmalloc(&pstart, ((size_t)g.n_parts + 2) * 4) != hipSuccess) return mcom_fail(ctx, MCOM_E_NOMEM, "contig index: scratch");
	struct Guard { mcom_ctx *c; uint32_t *p; ~Guard() { (void)hipStreamSynchronize(c->stream); mcom_dfree(p); } } guard{ctx, pstart};
	const uint32_t *skey = nullptr; const uint64_t *sslot = nullptr;
	int rc = mcom_cindex_partition(ctx, d_key, d_slot, n_ent64, grouped, d_key_tmp, d_slot_tmp, L, ininumdict, geom, pstart, &skey, &sslot);
	if (rc) return rc;
	return mcom_cindex_assemble(ctx, skey, sslot, pstart, L, ininumdict, geom, d_keys, n_words);
}

// both steps for one GPU: 24 bytes per entry of temporaries from the library's block pool
extern "C" int mcom_cindex_build(mcom_ctx *ctx, const uint64_t *d_cbits, const uint64_t *d_coff, const uint64_t *d_woff, uint32_t n_contigs,
                                 uint64_t n_windows, int L, int ininumdict, uint64_t geom, uint64_t *d_keys, uint64_t n_words)
{
	if (!ctx) return MCOM_E_ARG;
	CixGeom g;
	if (L < 1 || L > 256 || cix_geom(L, ininumdict, g)) return mcom_fail(ctx, MCOM_E_ARG, "bad dictionary layout");
	cix_unpack(geom, g);
	if (g.n_owners != 1) return mcom_fail(ctx, MCOM_E_ARG, "mcom_cindex_build makes the whole index: one share");
	const uint64_t cap = n_windows + (uint64_t)n_contigs * (uint64_t)g.maxoff + 1;
	if (cap >= (1ull << 32)) return mcom_fail(ctx, MCOM_E_ARG, "too many contig positions");
	const size_t key_b = cx_al((size_t)cap * 4 + 4), slot_b = cx_al((size_t)cap * 8 + 8);
	char *tmp = nullptr;
	if (mcom_dmalloc(&tmp, 2 * key_b + 2 * slot_b) != hipSuccess) return mcom_fail(ctx, MCOM_E_NOMEM, "contig index: %zu bytes of temporaries", 2 * key_b + 2 * slot_b);
	struct Guard { mcom_ctx *c; char *p; ~Guard() { (void)hipStreamSynchronize(c->stream); mcom_dfree(p); } } guard{ctx, tmp};
	uint32_t *keyA = (uint32_t*)tmp, *keyB = (uint32_t*)(tmp + key_b);
	uint64_t *slotA = (uint64_t*)(tmp + 2 * key_b), *slotB = (uint64_t*)(tmp + 2 * key_b + slot_b);
	uint64_t cnt = 0;
	int rc = mcom_cindex_entries(ctx, d_cbits, d_coff, d_woff, n_contigs, 0, n_contigs, L, ininumdict, geom, keyA, slotA, cap, &cnt);
	if (rc) return rc;
	return mcom_cindex_place(ctx, keyA, slotA, cnt, 1, keyB, slotB, L, ininumdict, geom, d_keys, n_words);
}
