// tools/experiments/sketchflat.hip -- NOT PART OF THE PRODUCT (not compiled into libmcom_hip.so): a measured dead end kept for reference.
// mm_sketch_lh_ori (reference sketch.c:116-165) for the regular case, one thread per
// contig POSITION instead of one wave per contig.
//
// Regular = k odd and no ambiguous base in any contig (what every contig of the pipeline is: consensus strings are
// ACGT, k = 31).  Then the scan of the reference has no hidden state:
//   * a k-mer cannot equal its reverse complement (odd k), and with the registers only partly filled at a contig start
//     it cannot either (the forward register is zero above 2m bits where the reverse one holds complements, and vice
//     versa below 2(k-m) bits), so every position stores exactly one entry: entry index = position, run counter =
//     position + 1, the entry is real (has a hash) from position k-1 on;
//   * what storing entry t emits depends on the hashes of positions t-w .. t only (contigs.hip: phase 2).
// So a block takes TILE consecutive positions of the concatenated contigs plus a halo of w + k - 1 characters before
// them, packs the characters to 2 bits (ballots + scalar bit spreading), lets every thread cut its k-mer out of the
// packed words and hash it, builds the sparse table over the TILE + w entries (newest smallest entry of any window,
// "equal hash in the window" flag) and evaluates the reference's statements per position.  Window queries are clamped
// to the contig's first position: entries before it are the ring's initial all-ones fill.  Counts per block, one scan,
// the same kernel again writes the records at their final place (position order = contig order) and the per-contig
// offsets (one pass: the records go to a temporary array first, block by block).  Irregular input (even k, an ambiguous base, a per-contig limit) goes to the wave-per-contig kernel.
//
// MEASURED (round 1, 100 M reads): bit-exact, and as fast as the wave-per-contig kernel but not faster (80 ms per step
// either way: 8 VALU wave-instructions per position here -- hash 3, sparse table 3, emission rules 2 -- against 15 there,
// but at a lower issue rate).  Removed from the library in round 2; a cheaper window minimum than the
// log2(w)-level table is what would make it win.
#include "../../minicom_amd/csrc/mcom_dev.hpp"
#include <vector>
#include <algorithm>

#define FTH 64                             // threads per block
#define FPER 4                             // positions per thread
#define FT (FTH * FPER)                    // positions per block
#define FMAXW 128
#define FE (FT + FMAXW)                    // entries per block at most
#define FH (FMAXW + 32)                    // halo characters at most (w + k - 1 <= 158)

__device__ __forceinline__ uint64_t f_rev_groups64(uint64_t x)
{
	x = __brevll(x);
	return ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
}
__device__ __forceinline__ uint64_t f_spread32(uint64_t x)
{
	x &= 0xFFFFFFFFull;
	x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
	x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
	x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
	x = (x | (x << 2)) & 0x3333333333333333ull;
	x = (x | (x << 1)) & 0x5555555555555555ull;
	return x;
}
__device__ __forceinline__ int f_nt4(uint8_t ch) { const uint8_t u = ch & 0xDF; return u == 'A' ? 0 : u == 'C' ? 1 : u == 'G' ? 2 : u == 'T' ? 3 : 4; }

// any character that is not ACGT (either case)?
__global__ void k_flat_check(const uint8_t *__restrict__ seq, uint64_t n_chars, unsigned int *__restrict__ bad)
{
	const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
	if (i >= n_chars) return;
	bool b = false;
	if (i + 16 <= n_chars && (((uintptr_t)(seq + i)) & 15) == 0) {
		const uint4 v = *(const uint4*)(seq + i);
		const uint32_t ws[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const uint32_t u = ws[q] & 0xDFDFDFDFu;
			auto eq = [](uint32_t x, uint32_t pat) { const uint32_t z = x ^ pat; return ~(((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z | 0x7F7F7F7Fu); };
			const uint32_t ok = eq(u, 0x41414141u) | eq(u, 0x43434343u) | eq(u, 0x47474747u) | eq(u, 0x54545454u);
			b |= ok != 0x80808080u;
		}
	} else for (uint64_t j = i; j < n_chars && j < i + 16; ++j) b |= f_nt4(seq[j]) > 3;
	if (b) *bad = 1;
}

// FTH threads handle FT = FTH * FPER consecutive positions (position g0 + i * FTH + tid for i < FPER): small blocks so
// that many of them are in flight per CU -- a block is a chain of dependent global loads (contig search, characters,
// offsets) whose latency only other blocks can hide.
struct FlatChunk { uint32_t start, count; };

__global__ __launch_bounds__(FTH) void k_sketch_flat(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ off, const uint32_t *__restrict__ ids,
                                                     uint32_t n, uint64_t n_chars, int w, int k, FlatChunk *__restrict__ chunks,
                                                     uint32_t *__restrict__ mloc, mcom_mm128 *__restrict__ tmp, uint64_t arena_cap, uint32_t arena_mask,
                                                     unsigned long long *__restrict__ cursors)
{
	__shared__ uint64_t PW[(FT + FH) / 32 + 2];     // 2-bit packed characters of [g0 - H, g0 + FT)
	__shared__ uint64_t EX[FE];                     // entry e <-> position g0 - w + e: hash, U64MAX when not real
	__shared__ uint32_t EP[FE];                     // pos<<1 | strand, 0xFFFFFFFF when not real
	__shared__ uint8_t ST[8][FE];                   // sparse table as in contigs.hip
	__shared__ uint32_t srch[16];
	__shared__ uint32_t wsum[FTH / 64 + 1];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const uint64_t g0 = (uint64_t)blockIdx.x * FT;
	const int H = w + k - 1;
	const uint64_t mask = (1ull << (2 * k)) - 1;
	// ---- characters -> packed words (issued first: they do not depend on the contig search).  Group q = 64 positions
	// starting at g0 - H + 64 q; H need not be a multiple of 64.
	const int ngrp = (FT + H + 63) / 64;
	for (int q = wv; q < ngrp; q += FTH / 64) {
		const int64_t p = (int64_t)g0 - H + 64 * q + lane;
		const int c = (p >= 0 && (uint64_t)p < n_chars) ? f_nt4(seq[p]) & 3 : 0;
		const uint64_t b0 = __ballot(c & 1), b1 = __ballot(c & 2);
		if (lane == 0) {
			PW[2 * q] = f_spread32(b0) | (f_spread32(b1) << 1);
			PW[2 * q + 1] = f_spread32(b0 >> 32) | (f_spread32(b1 >> 32) << 1);
		}
	}
	const uint32_t c0 = mcom_block_search(n, [&](uint32_t c) { return off[c] <= g0; }, srch);   // the contig that holds g0
	__syncthreads();
	const uint64_t s0 = off[c0];
	// ---- hashes of the entries: e in [0, FT + w) <-> position g0 - w + e
	const int NE = FT + w;
	for (int e = tid; e < NE; e += FTH) {
		const int64_t q = (int64_t)g0 - w + e;
		uint64_t x = U64MAX; uint32_t pp = 0xFFFFFFFFu;
		uint64_t qs = 0; bool in = false;
		if (e < w) { in = q >= (int64_t)s0; qs = s0; }                       // halo: only the part inside the contig of g0 matters
		else if ((uint64_t)q < n_chars) {                                    // a tile position: its own contig, a few steps from c0
			uint32_t cc = c0;
			while (cc + 1 < n && off[cc + 1] <= (uint64_t)q) ++cc;
			in = true; qs = off[cc];
		}
		if (in) {
			const uint64_t prel = (uint64_t)q - qs;
			if (prel >= (uint64_t)(k - 1)) {
				const uint64_t first = (uint64_t)(q - (k - 1) - ((int64_t)g0 - H));   // bit position / 2 of the k-mer's first base in PW
				const int sft = 2 * (int)(first & 31);
				uint64_t V = PW[first >> 5] >> sft;
				if (sft) V |= PW[(first >> 5) + 1] << (64 - sft);
				V &= mask;
				const uint64_t rev = (~V) & mask;
				const uint64_t fwd = f_rev_groups64(V) >> (64 - 2 * k);
				const uint32_t z = fwd < rev ? 0u : 1u;                          // fwd != rev: k is odd
				x = mcom_hash64(z ? rev : fwd, mask);
				pp = ((uint32_t)prel << 1) | z;
			}
		}
		EX[e] = x; EP[e] = pp;
	}
	__syncthreads();
	// ---- sparse table over the entries
	int LG = 0; while ((2 << LG) <= w) ++LG;
	auto lvl = [&](int j, int e, bool &dup) -> int {
		if (j == 0) { dup = false; return e; }
		const uint8_t v = ST[j][e]; dup = (v & 128) != 0; return e + (v & 127);
	};
	for (int j = 1; j <= LG; ++j) {
		const int h = 1 << (j - 1);
		for (int e = tid; e + 2 * h <= NE; e += FTH) {
			bool da, db;
			const int a = lvl(j - 1, e, da), b = lvl(j - 1, e + h, db);
			const uint64_t xa = EX[a], xb = EX[b];
			const int win = xa < xb ? a : b;                                   // equal: b, the newer
			const bool dup = xa == xb ? true : (xa < xb ? da : db);
			ST[j][e] = (uint8_t)((win - e) | (dup ? 128 : 0));
		}
		__syncthreads();
	}
	// ---- what storing an entry emits (sketch.c:138-161), FPER positions per thread.  First sweep: counts and every
	// entry's first record (nearly always its only one); then one atomicAdd reserves the block's room in the temporary
	// array (1024 arenas, as in contigs.hip); second sweep: write.
	uint32_t mine_r[FPER], excl_r[FPER], fp_r[FPER], c_r[FPER]; uint64_t fx_r[FPER];
	uint32_t run_base = 0;
	auto row_eval = [&](int row, auto &&put_fn) {
		const uint64_t g = g0 + (uint64_t)row * FTH + tid;
		const uint32_t c = c_r[row];
		const uint64_t cs = off[c], ce = off[c + 1];
		const int te = w + row * FTH + tid;                                   // my entry
		const int64_t csrel = (int64_t)cs - ((int64_t)g0 - w);                // entry of my contig's first position
		const int first_e = csrel > 0 ? (int)csrel : 0;                       // before it: the ring's initial fill
		const int64_t prel = (int64_t)(g - cs);
		auto EXat = [&](int e) -> uint64_t { return e < first_e ? U64MAX : EX[e]; };
		auto EPat = [&](int e) -> uint32_t { return e < first_e ? 0xFFFFFFFFu : EP[e]; };
		// newest smallest entry of [lo, hi] (hi - lo + 1 <= w), entries before the contig start being all-ones
		auto wquery = [&](int lo, int hi, bool &dup) -> int {
			if (lo < first_e) lo = first_e;                                   // the initial fill never beats a later entry, and never wins a tie
			const int len = hi - lo + 1;
			int j = 0; while ((2 << j) <= len) ++j;
			bool da, db;
			const int a = lvl(j, lo, da), b = lvl(j, hi - (1 << j) + 1, db);
			if (a == b) { dup = da || db; return a; }
			const uint64_t xa = EX[a], xb = EX[b];
			if (xa < xb) { dup = da; return a; }
			dup = xa == xb ? true : db;
			return b;
		};
		const uint64_t cx = EX[te];
		const int64_t run = prel + 1;
		int bidx = te - w; bool bdup = false;                                  // prel == 0: the ring is all initial fill
		if (prel > 0) bidx = wquery(te - w, te - 1, bdup);
		const bool binit = prel == 0;
		const uint64_t bx = binit ? U64MAX : EX[bidx]; const uint32_t bp = binit ? 0xFFFFFFFFu : EP[bidx];
		if (run == w + k - 1 && bdup && bx != U64MAX) {
			for (int e = te - w + 1; e < te; ++e) { const uint64_t x = EXat(e); const uint32_t pp = EPat(e); if (bx == x && pp != bp) put_fn(x, pp); }
		}
		if (cx <= bx) {
			if (run >= w + k) put_fn(bx, bp);
		} else if (bidx == te - w) {
			if (run >= w + k - 1) {
				put_fn(bx, bp);
				bool ndup; const int nidx = wquery(te - w + 1, te, ndup);
				const uint64_t nx = EX[nidx]; const uint32_t np = EP[nidx];
				if (ndup && nx != U64MAX)
					for (int e = te - w + 1; e <= te; ++e) { const uint64_t x = EXat(e); const uint32_t pp = EPat(e); if (nx == x && np != pp) put_fn(x, pp); }
			}
		}
		if ((uint64_t)g + 1 == ce) {                                          // last position: the minimum still held (sketch.c:163-164)
			bool d; const int b = wquery(te - w + 1, te, d);
			if (EX[b] != U64MAX) put_fn(EX[b], EP[b]);
		}
	};
#pragma unroll
	for (int row = 0; row < FPER; ++row) {
		const uint64_t g = g0 + (uint64_t)row * FTH + tid;
		const bool live = g < n_chars;
		uint32_t c = c0;
		if (live) while (c + 1 < n && off[c + 1] <= g) ++c;
		c_r[row] = c;
		uint32_t mine = 0; uint64_t fx = 0; uint32_t fp = 0;
		if (live) row_eval(row, [&](uint64_t x, uint32_t pp) { if (mine == 0) { fx = x; fp = pp; } ++mine; });
		uint32_t incl = mine;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d, 64); if (lane >= d) incl += v; }
		uint32_t add = 0, all;
		if (FTH > 64) {
			__syncthreads();
			if (lane == 63) wsum[wv] = incl;
			__syncthreads();
			all = 0;
			for (int q = 0; q < FTH / 64; ++q) { if (q < wv) add += wsum[q]; all += wsum[q]; }
		} else all = __shfl(incl, 63, 64);
		mine_r[row] = mine; fx_r[row] = fx; fp_r[row] = fp; excl_r[row] = run_base + add + incl - mine;
		if (live && g == off[c]) {                                            // my contig starts here, and so do the empty ones before it
			mloc[c] = excl_r[row];
			for (uint32_t cc = c; cc > 0 && off[cc - 1] == off[c]; --cc) mloc[cc - 1] = excl_r[row];
		}
		run_base += all;
	}
	// room for the block's records
	__shared__ unsigned long long start_s;
	const uint32_t arena = blockIdx.x & arena_mask;
	if (tid == 0) {
		unsigned long long st = 0;
		if (run_base) st = atomicAdd(&cursors[arena], (unsigned long long)run_base);
		start_s = st;
		FlatChunk ck; ck.start = (uint32_t)((uint64_t)arena * arena_cap + st); ck.count = run_base; chunks[blockIdx.x] = ck;
	}
	__syncthreads();
	const unsigned long long st = start_s;
	if (st + run_base > arena_cap) return;                                    // the arena is full: counted, not written (the caller retries)
	mcom_mm128 *dst = tmp + (uint64_t)arena * arena_cap + st;
#pragma unroll
	for (int row = 0; row < FPER; ++row) {
		if (!mine_r[row]) continue;
		const uint32_t c = c_r[row];
		const uint64_t idhi = (uint64_t)(ids ? ids[c] : (uint32_t)(c << 8)) << 32;
		auto rec = [&](uint64_t x, uint32_t pp) { mcom_mm128 v; v.x = x; v.y = (x == U64MAX && pp == 0xFFFFFFFFu) ? U64MAX : (idhi | pp); return v; };
		if (mine_r[row] == 1) dst[excl_r[row]] = rec(fx_r[row], fp_r[row]);
		else { uint32_t o = excl_r[row]; row_eval(row, [&](uint64_t x, uint32_t pp) { dst[o++] = rec(x, pp); }); }
	}
}

// the records of block b, to their final place (block order = position order = contig order)
__global__ __launch_bounds__(256) void k_flat_gather(const FlatChunk *__restrict__ chunks, const uint32_t *__restrict__ base, size_t nblocks,
                                                     const mcom_mm128 *__restrict__ tmp, mcom_mm128 *__restrict__ out)
{
	const size_t b = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	if (b >= nblocks) return;
	const int lane = threadIdx.x & 63;
	const FlatChunk ck = chunks[b];
	const uint32_t o = base[b];
	for (uint32_t i = lane; i < ck.count; i += 64) out[o + i] = tmp[ck.start + i];
}
__global__ void k_flat_counts(const FlatChunk *__restrict__ chunks, size_t nblocks, uint32_t *__restrict__ cnt)
{
	const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (b <= nblocks) cnt[b] = b < nblocks ? chunks[b].count : 0u;
}
// offsets of the contigs: the base of the block that holds the contig's first position + the prefix inside it
__global__ void k_flat_moff(const uint64_t *__restrict__ off, uint32_t n, uint64_t n_chars, const uint32_t *__restrict__ base, const uint32_t *__restrict__ mloc,
                            uint32_t total, uint32_t *__restrict__ moff)
{
	const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c > n) return;
	if (c == n || off[c] >= n_chars) { moff[c] = total; return; }            // the end, and empty contigs behind the last character
	moff[c] = base[off[c] / FT] + mloc[c];
}

// 0 = done through the flat path (*used = 1) or not applicable (*used = 0, nothing written); else an error
int mcom_sketch_contigs_flat(mcom_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_off, const uint32_t *d_ids, size_t n, uint64_t n_chars,
                             int w, int k, uint32_t *d_moff, mcom_mm128 *d_out, size_t cap, uint64_t *h_total, int *used)
{
	*used = 0;
	if ((k & 1) == 0 || w > FMAXW || w + k - 1 > FH || n_chars == 0 || n == 0) return MCOM_OK;
	const uint64_t nblocks = (n_chars + FT - 1) / FT;
	if (nblocks >= (1ull << 31)) return MCOM_OK;
	auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
	uint32_t arenas = 1; while (arenas < 1024 && (size_t)arenas * 4096 <= nblocks) arenas <<= 1;
	const uint64_t arena_cap = cap / arenas;
	const size_t cnt_b = al((nblocks + 1) * 4), scr_b = al(mcom_scan_scratch_elems(nblocks + 1) * 4 + 1024), chunk_b = al(nblocks * sizeof(FlatChunk)),
	             mloc_b = al((n + 1) * 4), cur_b = al(arenas * 8), tmp_b = al(cap * sizeof(mcom_mm128));
	int rc = mcom_ws_reserve(ctx, 2 * cnt_b + scr_b + chunk_b + mloc_b + cur_b + 256 + tmp_b);
	if (rc) return rc;
	char *base = (char*)ctx->ws; size_t o = 0;
	auto take = [&](size_t b) { char *q = base + o; o += b; return q; };
	uint32_t *cnt = (uint32_t*)take(cnt_b), *bbase = (uint32_t*)take(cnt_b), *scr = (uint32_t*)take(scr_b);
	FlatChunk *chunks = (FlatChunk*)take(chunk_b);
	uint32_t *mloc = (uint32_t*)take(mloc_b);
	unsigned long long *cursors = (unsigned long long*)take(cur_b);
	unsigned int *bad = (unsigned int*)take(256);
	mcom_mm128 *tmp = (mcom_mm128*)take(tmp_b);
	MCOM_HIP(ctx, hipMemsetAsync(cursors, 0, cur_b + 256, ctx->stream));
	hipLaunchKernelGGL(k_flat_check, dim3((unsigned)((n_chars / 16 + 1 + 255) / 256)), dim3(256), 0, ctx->stream, d_seq, n_chars, bad);
	unsigned int hb = 0;
	MCOM_HIP(ctx, hipMemcpyAsync(&hb, bad, 4, hipMemcpyDeviceToHost, ctx->stream));
	MCOM_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (hb) return MCOM_OK;                                                   // an ambiguous base somewhere: the general kernel
	{ McomProfScope ps_(ctx, PROF_SKETCH_CONTIGS);
	hipLaunchKernelGGL(k_sketch_flat, dim3((unsigned)nblocks), dim3(FTH), 0, ctx->stream, d_seq, d_off, d_ids, (uint32_t)n, n_chars, w, k, chunks, mloc, tmp,
	                   arena_cap, arenas - 1, cursors); }
	MCOM_LAUNCH_CHECK(ctx);
	hipLaunchKernelGGL(k_flat_counts, dim3((unsigned)((nblocks + 1 + 255) / 256)), dim3(256), 0, ctx->stream, chunks, (size_t)nblocks, cnt);
	if ((rc = mcom_scan_u32(ctx, cnt, bbase, nblocks + 1, scr))) return rc;
	std::vector<unsigned long long> fill(arenas);
	uint32_t total = 0;
	MCOM_HIP(ctx, hipMemcpyAsync(fill.data(), cursors, arenas * 8, hipMemcpyDeviceToHost, ctx->stream));
	MCOM_HIP(ctx, hipMemcpyAsync(&total, bbase + nblocks, 4, hipMemcpyDeviceToHost, ctx->stream));
	MCOM_HIP(ctx, hipStreamSynchronize(ctx->stream));
	unsigned long long most = 0, sum = 0;
	for (unsigned long long f : fill) { most = std::max(most, f); sum += f; }
	*used = 1;
	if (h_total) *h_total = total;
	if (sum >= (1ull << 32)) return mcom_fail(ctx, MCOM_E_ARG, "too many minimizers");
	if (most > arena_cap) {
		if (h_total) *h_total = (most + 1) * arenas;
		return mcom_fail(ctx, MCOM_E_OVERFLOW, "%llu minimizers but room for %zu", sum, cap);
	}
	hipLaunchKernelGGL(k_flat_moff, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, ctx->stream, d_off, (uint32_t)n, n_chars, bbase, mloc, total, d_moff);
	if (total) hipLaunchKernelGGL(k_flat_gather, dim3((unsigned)((nblocks * 64 + 255) / 256)), dim3(256), 0, ctx->stream, chunks, bbase, (size_t)nblocks, tmp, d_out);
	MCOM_LAUNCH_CHECK(ctx);
	MCOM_HIP(ctx, hipStreamSynchronize(ctx->stream));                          // the workspace arrays are in use until here
	return MCOM_OK;
}
