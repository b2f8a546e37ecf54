// minicom_amd/csrc/contigs.hip -- contig-side kernels of Stage 1 for gfx950 (MI355X).
//
//   mcom_sketch_contigs     : mm_sketch_lh_ori, the minimap-style (w,k)-minimizers of a contig
//                             (reference sketch.c:116-165), one thread per contig
//   mcom_pack_contigs       : ASCII contigs -> 2-bit packed words (the layout mcom_realign_pass reads)
//   mcom_idx_build / _get   : mm_idx_generation / mm_idx_get (kthread_idx.c:116-170, :84-101)
//   mcom_match_pro          : match_pro (kthread_cb.c:36-52) on packed contigs, batched
//   mcom_find_next_candidates: the lookup part of find_next (kthread_cb.c:267-291) for every contig
#include "mcom_dev.hpp"
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <vector>

#define MAXW 128

// A0 C1 G2 T3, anything else 4 (sketch.c:8-25 for the letters that occur)
__device__ __forceinline__ int nt4_of(uint8_t ch)
{
	const uint8_t u = ch & 0xDF;                       // fold case
	return u == 'A' ? 0 : u == 'C' ? 1 : u == 'G' ? 2 : u == 'T' ? 3 : 4;
}

// ------------------------------------------------------------------------------------------------
// mm_sketch_lh_ori, one pass.  At most `limit` minimizers per contig.
//
// One wave (one 64-thread workgroup) per contig, the contig taken in pieces of PIECE bases.  The reference
// scan keeps, besides the run counter, a ring of the last w entries and "the minimum": by construction that
// minimum is always the NEWEST smallest entry of the ring (sketch.c:145-153), i.e. a pure function of the last
// w entries.  What the scan emits when it stores entry t therefore depends only on entry t, on the newest
// smallest entry of the windows ending at t-1 and at t, and (rarely) on equal hashes inside the window: every
// entry can be handled by its own lane.
//   phase 1   the k-mer registers of the reference hold the last k UNAMBIGUOUS bases (an ambiguous base resets the
//             run counter but does not enter the registers, sketch.c:129-132; at the contig start they are partly
//             zero).  The wave keeps those bases 2-bit packed in an LDS ring (bit planes from two ballots, spread
//             with scalar bit tricks), and every lane cuts the k-mer ending at its own base out of the ring:
//             reverse register = ~window, forward register = the window with its base order reversed.  No lane
//             rolls a register over bases that belong to another lane;
//   phase 1b  ballots over the flags give, per position, the run counter (valid, non-palindromic bases since
//             the last ambiguous base) and the ENTRY index (palindromic k-mers store no entry, sketch.c:133);
//             every position writes its entry (hash or "empty", pos<<1|strand, run) into an LDS ring by entry index;
//   phase 1c  every new entry gets the newest smallest entry of the four entries ending at it, then -- from w/4 such
//             blocks and w%4 single entries -- the newest smallest entry of the window of w entries ending at it, plus
//             a flag "another entry of the window has the same hash"; both stay in LDS rings (two dependent LDS
//             steps instead of the log2 w levels of a sparse table, and each window is evaluated once);
//   phase 2   one lane per entry: the reference's statements (sketch.c:138-161) with the window minimum looked
//             up instead of scanned; counts, wave prefix sums, one atomicAdd per piece reserves room in a
//             temporary record array, then the same once more to write.
// A second, cheap kernel strings the pieces of every contig together in contig order (k_sketch_gather).
// ------------------------------------------------------------------------------------------------
#define PIECE 128
#define NGRP (PIECE / 64)
#define ERING 256                          // >= PIECE + MAXW, power of two
#define EMASK (ERING - 1)
#define RBCAP 64                           // records collected in LDS before room is reserved for them (one atomic)

// reverse the order of the 32 bases of a word (2-bit groups)
__device__ __forceinline__ uint64_t rev_groups64(uint64_t x)
{
	x = __brevll(x);
	return ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
}
// bit i of the low 32 bits -> bit 2i
__device__ __forceinline__ uint64_t spread32(uint64_t x)
{
	x &= 0xFFFFFFFFull;
	x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
	x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
	x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
	x = (x | (x << 2)) & 0x3333333333333333ull;
	x = (x | (x << 1)) & 0x5555555555555555ull;
	return x;
}

struct SkChunk { uint32_t start, count; };

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(7, 8))) void k_sketch_contigs(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ off,
                                                       const uint64_t *__restrict__ off_end,
                                                       const uint32_t *__restrict__ ids, size_t n, int w, int k, uint32_t limit,
                                                       const uint32_t *__restrict__ piece_off, SkChunk *__restrict__ chunks,
                                                       mcom_mm128 *__restrict__ tmp, uint64_t arena_cap, uint32_t arena_mask,
                                                       unsigned long long *__restrict__ cursors, uint32_t *__restrict__ cnt)
{
	__shared__ uint64_t CW[8];              // the last <= 256 unambiguous bases, 2-bit packed, base q at bits 2(q%32) of word (q/32)%8
	__shared__ uint64_t EX[ERING];          // per entry (index mod ERING): hash or U64MAX when empty
	__shared__ uint32_t EP[ERING];          // pos<<1|strand, 0xFFFFFFFF when empty
	__shared__ uint16_t ER[ERING];          // run counter after the entry (saturating)
	__shared__ uint8_t M4O[ERING];          // newest smallest entry of entries [e-3, e]: bits 0-1 how far it lies before e;
	                                        //   bit 7: another entry of the four has the same hash
	__shared__ uint8_t WQ[ERING];           // window of w entries ending at e: bits 0-6 distance of its newest smallest entry
	                                        //   before e, bit 7: another entry of the window has the same hash
	__shared__ mcom_mm128 RB[RBCAP];        // records waiting for room in the temporary array
	const size_t t = blockIdx.x;
	if (t >= n) return;
	const int lane = threadIdx.x;
	const uint8_t *s = seq + off[t];
	const int len = (int)((off_end ? off_end[t] : off[t + 1]) - off[t]);   // off_end: string t is [off[t], off_end[t]), a segment
	const uint64_t idhi = (uint64_t)(ids ? ids[t] : (uint32_t)t) << 32;
	const uint64_t mask = (1ull << (2 * k)) - 1;
	SkChunk *my_chunks = chunks + piece_off[t];
	// room in the temporary array comes from one of arena_mask+1 arenas, each with its own cursor: a single cursor would
	// serialise tens of millions of atomics on one L2 channel
	unsigned long long *cursor = cursors + (t & arena_mask);
	const uint64_t arena0 = (uint64_t)(t & arena_mask) * arena_cap;
	int ent_in = 0;                        // entries stored before the current piece   (wave uniform)
	uint32_t run_in = 0;                    // run counter before the current piece       (wave uniform)
	uint32_t ne_base = 0;                   // minimizers emitted before the current piece (wave uniform)
	uint64_t qbase = 0;                     // unambiguous bases before the current group (wave uniform)
	const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));   // bits 0..lane-1

	// The rings start out as the scan's initial fill (all-ones entries): entry e >= -128 lives in slot e & EMASK, and a
	// negative e is only looked at while the entries written so far are below 128, so no access needs a guard.
	for (int i = lane; i < ERING; i += 64) { EX[i] = U64MAX; EP[i] = 0xFFFFFFFFu; M4O[i] = 128; WQ[i] = 0; }
	auto EXat = [&](int e) -> uint64_t { return EX[e & EMASK]; };
	auto EPat = [&](int e) -> uint32_t { return EP[e & EMASK]; };
	// newest smallest entry of the window of w entries ending at te (the reference's "min" after storing te) and whether
	// another entry of the window has the same hash: the w entries are w%4 single ones (the oldest) and w/4 blocks of
	// four whose newest smallest entries M4O points at, visited oldest first so that equal hashes leave the newest
	auto wscan = [&](int te, bool &dup) -> int {
		uint64_t x = U64MAX; int idx = te - w; dup = false;
		bool first = true;
		const int r = w & 3, q = w >> 2;
		for (int i = 0; i < r; ++i) {
			const int e = te - w + 1 + i;
			const uint64_t xe = EXat(e);
			if (first || xe <= x) { dup = !first && xe == x; x = xe; idx = e; }
			first = false;
		}
		// four blocks at a time: their eight LDS reads are in flight together, a loop of single blocks would wait for each
		for (int b0 = q - 1; b0 >= 0; b0 -= 4) {
			uint64_t xs[4]; uint8_t os[4];
#pragma unroll
			for (int j = 0; j < 4; ++j) { const int b = b0 - j < 0 ? 0 : b0 - j; os[j] = M4O[(te - 4 * b) & EMASK]; }
#pragma unroll
			for (int j = 0; j < 4; ++j) { const int b = b0 - j < 0 ? 0 : b0 - j; xs[j] = EX[(te - 4 * b - (os[j] & 3)) & EMASK]; }
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				if (b0 - j < 0) break;
				const int e = te - 4 * (b0 - j);
				const uint64_t xb = xs[j]; const uint8_t o = os[j];
				const int ib = e - (o & 3); const bool db = (o & 128) != 0;
				if (first || xb <= x) { dup = (!first && xb == x) ? true : db; x = xb; idx = ib; }
				first = false;
			}
		}
		return idx;
	};
	auto wquery = [&](int te, bool &dup) -> int {                    // the same, looked up (te >= 0, scanned already)
		const uint8_t v = WQ[te & EMASK]; dup = (v & 128) != 0; return te - (v & 127);
	};
	auto make_rec = [&](uint64_t x, uint32_t pp) -> mcom_mm128 {
		mcom_mm128 v; v.x = x; v.y = (x == U64MAX && pp == 0xFFFFFFFFu) ? U64MAX : (idhi | pp); return v;
	};
	// what storing entry te makes the scan emit (sketch.c:138-161); put(x, p) in emission order
	auto entry_emits = [&](int te, auto &&put) {
		const uint64_t cx = EX[te & EMASK];
		const int run = ER[te & EMASK];
		int bidx = -w; bool bdup = false;
		if (te > 0) bidx = wquery(te - 1, bdup);
		const uint64_t bx = EXat(bidx); const uint32_t bp = EPat(bidx);
		if (run == w + k - 1 && bdup && bx != U64MAX) {               // first full window: older copies of the minimum
			for (int e = te - w + 1; e < te; ++e) { const uint64_t x = EXat(e); const uint32_t pp = EPat(e); if (bx == x && pp != bp) put(x, pp); }
		}
		if (cx <= bx) {
			if (run >= w + k) put(bx, bp);
		} else if (bidx == te - w) {                                  // the minimum has just left the window
			if (run >= w + k - 1) {
				put(bx, bp);
				bool ndup; const int nidx = wquery(te, ndup);
				const uint64_t nx = EXat(nidx); const uint32_t np = EPat(nidx);
				if (ndup && nx != U64MAX)
					for (int e = te - w + 1; e <= te; ++e) { const uint64_t x = EXat(e); const uint32_t pp = EPat(e); if (nx == x && np != pp) put(x, pp); }
			}
		}
	};

	// Global round trips are what a wave of this kernel waits for (PMC: 60 % of its cycles in s_waitcnt, LDS conflicts
	// negligible), so the bases of the next piece are fetched while this one is worked on, and records collect in LDS
	// until RBCAP of them (or the contig's end) make one reservation worthwhile instead of one per piece.
	uint32_t rb_n = 0; int nchunk = 0;                                   // wave uniform
	auto flush = [&]() {
		if (!rb_n) return;
		unsigned long long st = 0;
		if (lane == 0) st = atomicAdd(cursor, (unsigned long long)rb_n);
		st = __shfl(st, 0, 64);
		if (lane == 0) { SkChunk ck; ck.start = (uint32_t)(arena0 + st); ck.count = rb_n; my_chunks[nchunk] = ck; }
		++nchunk;
		if (st + rb_n <= arena_cap) for (uint32_t i = lane; i < rb_n; i += 64) tmp[arena0 + st + i] = RB[i];
		rb_n = 0;
	};
	uint8_t nx[NGRP];
#pragma unroll
	for (int g = 0; g < NGRP; ++g) { const int p = g * 64 + lane; nx[g] = p < len ? s[p] : 0; }
	for (int ps = 0; ps < len && ne_base < limit; ps += PIECE) {
		const int pe = ps + PIECE < len ? ps + PIECE : len;
		__syncthreads();                                   // everybody is done with the rings of the previous piece
		uint8_t ch[NGRP];
#pragma unroll
		for (int g = 0; g < NGRP; ++g) { ch[g] = nx[g]; const int p = ps + PIECE + g * 64 + lane; nx[g] = p < len ? s[p] : 0; }
		// ---- phase 1a: this piece's unambiguous bases into the packed ring
		int cc[NGRP]; uint64_t QQ[NGRP];
#pragma unroll
		for (int g = 0; g < NGRP; ++g) {
			const int p = ps + g * 64 + lane;
			const int c = p < pe ? nt4_of(ch[g]) : 5;                     // 4 = ambiguous, 5 = beyond the end
			const bool valid = c < 4;
			const uint64_t vM = __ballot(valid);
			const int nv = __popcll(vM);
			cc[g] = c; QQ[g] = qbase + (uint64_t)__popcll(vM & below);
			if (nv) {
				const int sft = 2 * (int)(qbase & 31); const uint64_t w0 = qbase >> 5;
				if (vM == (nv == 64 ? ~0ull : ((1ull << nv) - 1))) {       // a run of bases from lane 0 on: bit planes -> 2-bit words
					const uint64_t b0 = __ballot(valid && (c & 1)), b1 = __ballot(valid && (c & 2));
					const uint64_t lo = spread32(b0) | (spread32(b1) << 1), hi = spread32(b0 >> 32) | (spread32(b1 >> 32) << 1);
					if (lane == 0) {
						if (sft == 0) { CW[w0 & 7] = lo; CW[(w0 + 1) & 7] = hi; }
						else {
							CW[w0 & 7] = (CW[w0 & 7] & ((1ull << sft) - 1)) | (lo << sft);
							CW[(w0 + 1) & 7] = (lo >> (64 - sft)) | (hi << sft);
							CW[(w0 + 2) & 7] = hi >> (64 - sft);
						}
					}
				} else {                                                   // ambiguous bases inside the group (rare): one OR per base
					if (lane == 0) {
						if (sft) CW[w0 & 7] &= (1ull << sft) - 1;
						for (uint64_t wi = (qbase + 31) >> 5; wi <= (qbase + (uint64_t)nv - 1) >> 5; ++wi) CW[wi & 7] = 0;
					}
					__syncthreads();
					if (valid) atomicOr((unsigned long long*)&CW[(QQ[g] >> 5) & 7], (unsigned long long)c << (2 * (QQ[g] & 31)));
					__syncthreads();
				}
			}
			qbase += (uint64_t)nv;
		}
		__syncthreads();
		// ---- phase 1 + 1b: k-mer, hash, run counter and entry index of every position; entries into the ring
		int ent_run = ent_in; uint32_t run_run = run_in;
#pragma unroll
		for (int g = 0; g < NGRP; ++g) {
			const int p = ps + g * 64 + lane;
			const int c = cc[g];
			uint8_t f = 2; uint64_t x = U64MAX;                           // beyond the end: stores nothing, counts nothing
			if (c < 4) {
				const uint64_t Q = QQ[g];
				const int m = Q + 1 < (uint64_t)k ? (int)(Q + 1) : k;       // bases in the registers
				const uint64_t first = Q + 1 - (uint64_t)m;
				const int sft = 2 * (int)(first & 31);
				uint64_t V = CW[(first >> 5) & 7] >> sft;
				if (sft) V |= CW[((first >> 5) + 1) & 7] << (64 - sft);
				const uint64_t mm = (1ull << (2 * m)) - 1;
				V &= mm;
				const uint64_t rev = ((~V) & mm) << (2 * (k - m));
				const uint64_t fwd = rev_groups64(V) >> (64 - 2 * m);
				if (fwd != rev) { const uint32_t z = fwd < rev ? 0u : 1u; f = (uint8_t)z; x = mcom_hash64(z ? rev : fwd, mask); }
			} else if (c == 4) f = 4;
			const bool isn = (f & 4) != 0, inc = (f & 6) == 0;
			const uint64_t nM = __ballot(isn), incM = __ballot(inc), entM = nM | incM;
			const uint64_t lowm = lane == 63 ? ~0ull : ((2ull << lane) - 1);   // bits 0..lane
			const uint64_t nlow = nM & lowm;
			uint32_t run;
			if (nlow) { const int hb = 63 - __clzll((long long)nlow); const uint64_t above = hb == 63 ? 0ull : ~((2ull << hb) - 1); run = (uint32_t)__popcll(incM & lowm & above); }
			else run = run_run + (uint32_t)__popcll(incM & lowm);
			if (isn || inc) {
				const int te = ent_run + (int)__popcll(entM & (lowm >> 1));
				const bool real = inc && run >= (uint32_t)k;
				EX[te & EMASK] = real ? x : U64MAX;
				EP[te & EMASK] = real ? (((uint32_t)p << 1) | (f & 1u)) : 0xFFFFFFFFu;
				ER[te & EMASK] = (uint16_t)(run > 0xFFFFu ? 0xFFFFu : run);
			}
			// carry to the next group (uniform)
			if (nM) { const int hb = 63 - __clzll((long long)nM); run_run = (uint32_t)__popcll(hb == 63 ? 0ull : (incM >> (hb + 1))); }
			else run_run += (uint32_t)__popcll(incM);
			ent_run += (int)__popcll(entM);
		}
		__syncthreads();
		// ---- phase 1c: minima of blocks of four for the new entries, then their window minima
#pragma unroll
		for (int g = 0; g < NGRP; ++g) {
			const int te = ent_in + 64 * g + lane;
			if (te < ent_run) {
				uint64_t x = EXat(te - 3); int o = 3; bool dup = false;
#pragma unroll
				for (int d = 2; d >= 0; --d) { const uint64_t xe = EXat(te - d); if (xe <= x) { dup = xe == x; x = xe; o = d; } }
				M4O[te & EMASK] = (uint8_t)(o | (dup ? 128 : 0));
			}
		}
		__syncthreads();
#pragma unroll
		for (int g = 0; g < NGRP; ++g) {
			const int te = ent_in + 64 * g + lane;
			if (te < ent_run) { bool d; const int i = wscan(te, d); WQ[te & EMASK] = (uint8_t)((te - i) | (d ? 128 : 0)); }
		}
		__syncthreads();
		// ---- phase 2: one lane per entry (at most PIECE entries per piece), emission order = entry order
		uint32_t mine[NGRP], excl[NGRP], gbase[NGRP], piece_total = 0;
		uint64_t fx[NGRP]; uint32_t fp[NGRP];                              // an entry's first record: nearly always its only one
#pragma unroll
		for (int g = 0; g < NGRP; ++g) {
			const int te = ent_in + 64 * g + lane;
			uint32_t m = 0;
			fx[g] = 0; fp[g] = 0;
			if (te < ent_run) entry_emits(te, [&](uint64_t x, uint32_t pp) { if (m == 0) { fx[g] = x; fp[g] = pp; } ++m; });
			// an entry emits one record at most unless equal hashes meet in a window: then (rarely) a real prefix sum
			const uint64_t anyM = __ballot(m != 0);
			uint32_t ex, tot;
			if (__ballot(m > 1) == 0) { ex = (uint32_t)__popcll(anyM & below); tot = (uint32_t)__popcll(anyM); }
			else {
				uint32_t incl = m;
#pragma unroll
				for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d, 64); if (lane >= d) incl += v; }
				ex = incl - m; tot = __shfl(incl, 63, 64);
			}
			mine[g] = m; excl[g] = ex; gbase[g] = piece_total;
			piece_total += tot;
		}
		const uint32_t room = limit - ne_base;
		const uint32_t take = piece_total < room ? piece_total : room;
		// a piece's records go to the LDS buffer; a piece with more than the buffer holds (tiny w) reserves its own room
		const bool direct = take > RBCAP;
		if (direct || rb_n + take > RBCAP) flush();
		unsigned long long start = 0;
		if (direct) {
			if (lane == 0) start = atomicAdd(cursor, (unsigned long long)take);
			start = __shfl(start, 0, 64);
			if (lane == 0) { SkChunk ck; ck.start = (uint32_t)(arena0 + start); ck.count = take; my_chunks[nchunk] = ck; }
			++nchunk;
		}
		const bool fits = start + take <= arena_cap;
		auto emit = [&](uint32_t rel, const mcom_mm128 &v) {
			if (rel >= take) return;
			if (!direct) RB[rb_n + rel] = v;
			else if (fits) tmp[arena0 + start + rel] = v;
		};
#pragma unroll
		for (int g = 0; g < NGRP; ++g) {
			if (!mine[g]) continue;
			uint32_t rel = gbase[g] + excl[g];
			if (mine[g] == 1) { emit(rel, make_rec(fx[g], fp[g])); continue; }
			entry_emits(ent_in + 64 * g + lane, [&](uint64_t x, uint32_t pp) { emit(rel, make_rec(x, pp)); ++rel; });
		}
		if (!direct) rb_n += take;
		ne_base = ne_base + piece_total < ne_base ? 0xFFFFFFFFu : ne_base + piece_total;
		ent_in = ent_run; run_in = run_run;
	}
	__syncthreads();
	// the minimum still held after the last entry is written out (sketch.c:163-164)
	if (ne_base < limit && ent_in > 0) {
		// the window minimum of the last piece covers the window ending at the last entry
		bool d; const int b = wquery(ent_in - 1, d);
		const uint64_t bx = EXat(b);
		if (bx != U64MAX) {
			if (rb_n == RBCAP) flush();
			if (lane == 0) RB[rb_n] = make_rec(bx, EPat(b));
			++rb_n; ++ne_base;
		}
	}
	flush();
	if (lane == 0) cnt[t] = ne_base < limit ? ne_base : limit;
}

__global__ void k_sketch_slots(const uint64_t *__restrict__ off, const uint64_t *__restrict__ off_end, size_t n, uint32_t *__restrict__ slots)
{
	const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t > n) return;
	slots[t] = t == n ? 0u : (uint32_t)(((off_end ? off_end[t] : off[t + 1]) - off[t] + PIECE - 1) / PIECE) + 1u;
}
// the chunks of contig t, in piece order, to out[moff[t] ...)
__global__ __launch_bounds__(256) void k_sketch_gather(const uint32_t *__restrict__ piece_off, const SkChunk *__restrict__ chunks,
                                                       const mcom_mm128 *__restrict__ tmp, size_t n, const uint32_t *__restrict__ moff,
                                                       mcom_mm128 *__restrict__ out)
{
	const size_t t = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	if (t >= n) return;
	const int lane = threadIdx.x & 63;
	uint32_t dst = moff[t];
	const uint32_t end = moff[t + 1];
	for (uint32_t c = piece_off[t]; c < piece_off[t + 1] && dst < end; ++c) {
		const SkChunk ck = chunks[c];
		for (uint32_t i = lane; i < ck.count; i += 64) out[dst + i] = tmp[ck.start + i];
		dst += ck.count;
	}
}

// d_off_end = NULL: string t is [d_off[t], d_off[t+1]) and their total is read from d_off[n]; otherwise string t is the
// segment [d_off[t], d_off_end[t]) and chars_bound bounds the sum of their lengths
int mcom_sketch_strings(mcom_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_off, const uint64_t *d_off_end, uint64_t chars_bound,
                        const uint32_t *d_ids, size_t n, int w, int k, uint32_t max_per_contig, uint32_t *d_moff, mcom_mm128 *d_out, size_t cap,
                        uint64_t *h_total);
int mcom_sketch_strings_scan(mcom_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_off, const uint64_t *d_off_end, uint64_t chars,
                             const uint32_t *d_ids, size_t n, int w, int k, uint32_t limit, uint32_t *d_moff, mcom_mm128 *d_out, size_t cap,
                             uint64_t *h_total);

extern "C" int mcom_sketch_contigs(mcom_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_off, const uint32_t *d_ids, size_t n,
                                   int w, int k, uint32_t max_per_contig, uint32_t *d_moff, mcom_mm128 *d_out, size_t cap,
                                   uint64_t *h_total)
{
	return mcom_sketch_strings(ctx, d_seq, d_off, nullptr, 0, d_ids, n, w, k, max_per_contig, d_moff, d_out, cap, h_total);
}

int mcom_sketch_strings(mcom_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_off, const uint64_t *d_off_end, uint64_t chars_bound,
                        const uint32_t *d_ids, size_t n, int w, int k, uint32_t max_per_contig, uint32_t *d_moff, mcom_mm128 *d_out, size_t cap,
                        uint64_t *h_total)
{
	if (!ctx) return MCOM_E_ARG;
	if (h_total) *h_total = 0;
	if (k < 1 || k > 31 || w < 1 || w > MAXW) return mcom_fail(ctx, MCOM_E_ARG, "w=%d (1..%d) or k=%d (1..31) out of range", w, MAXW, k);
	if (n >= (1ull << 31)) return mcom_fail(ctx, MCOM_E_ARG, "too many contigs");
	if (!d_moff) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	if (n == 0) { MCOM_HIP(ctx, hipMemsetAsync(d_moff, 0, 4, ctx->stream)); return MCOM_OK; }
	if (!d_seq || !d_off) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	if (cap && !d_out) return mcom_fail(ctx, MCOM_E_ARG, "null output pointer");
	const uint32_t limit = max_per_contig ? max_per_contig : 0xFFFFFFFFu;
	uint64_t chars = chars_bound;
	if (!d_off_end) {
		MCOM_HIP(ctx, mcom_d2h_async(ctx, &chars, d_off + n, 8));
		MCOM_HIP(ctx, mcom_stream_sync(ctx));
	}
	// millions of short strings: one lane per string (sketch_scan.hip); wide windows and very long strings stay with the wave per string,
	// and so does a handful of strings (the last merge rounds sketch a few hundred segments: a lane needs ~95 ns per base, 0.35 ms
	// for the longest whatever their number, where 8192 waves are all resident and go through it 128 bases at a time)
	if (!ctx->sketch_wave_only && (n > 8192 || ctx->sketch_lane_always)) {
		const int rs = mcom_sketch_strings_scan(ctx, d_seq, d_off, d_off_end, chars, d_ids, n, w, k, limit, d_moff, d_out, cap, h_total);
		if (rs != -1) return rs;
	}
	const uint64_t max_slots = chars / PIECE + 2 * (uint64_t)n + 1;
	if (max_slots >= (1ull << 32)) return mcom_fail(ctx, MCOM_E_ARG, "too many contig pieces");
	auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
	uint32_t arenas = 1; while (arenas < 1024 && (size_t)arenas * 8192 <= n) arenas <<= 1;
	const uint64_t arena_cap = cap / arenas;
	const size_t slot_b = al((n + 1) * 4), scr_b = al(mcom_scan_scratch_elems(n + 1) * 4 + 1024), chunk_b = al(max_slots * sizeof(SkChunk)), cur_b = al(arenas * 8),
	             tmp_b = al(cap * sizeof(mcom_mm128));
	int rc = mcom_ws_reserve(ctx, slot_b + scr_b + chunk_b + cur_b + tmp_b);
	if (rc) return rc;
	char *base = (char*)ctx->ws;
	uint32_t *piece_off = (uint32_t*)base;
	uint32_t *scr = (uint32_t*)(base + slot_b);
	SkChunk *chunks = (SkChunk*)(base + slot_b + scr_b);
	unsigned long long *cursor = (unsigned long long*)(base + slot_b + scr_b + chunk_b);
	mcom_mm128 *tmp = (mcom_mm128*)(base + slot_b + scr_b + chunk_b + cur_b);
	MCOM_LAUNCH(k_sketch_slots, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, ctx->stream, d_off, d_off_end, n, piece_off);
	MCOM_LAUNCH_CHECK(ctx);
	if ((rc = mcom_scan_u32(ctx, piece_off, piece_off, n + 1, scr))) return rc;
	MCOM_HIP(ctx, hipMemsetAsync(chunks, 0, chunk_b + cur_b, ctx->stream));             // chunk table and the cursors behind it
	{ McomProfScope ps_(ctx, PROF_SKETCH_CONTIGS);
	MCOM_LAUNCH(k_sketch_contigs, dim3((unsigned)n), dim3(64), 0, ctx->stream, d_seq, d_off, d_off_end, d_ids, n, w, k, limit, piece_off, chunks, tmp,
	                   arena_cap, arenas - 1, cursor, d_moff); }
	MCOM_LAUNCH_CHECK(ctx);
	// counts are in d_moff[0..n), d_moff[n] = 0, then an exclusive scan over n+1 entries leaves the total in d_moff[n]
	MCOM_HIP(ctx, hipMemsetAsync(d_moff + n, 0, 4, ctx->stream));
	if ((rc = mcom_scan_u32(ctx, d_moff, d_moff, n + 1, scr))) return rc;
	std::vector<unsigned long long> fill(arenas);
	MCOM_HIP(ctx, mcom_d2h_async(ctx, fill.data(), cursor, arenas * 8));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	unsigned long long total = 0, most = 0;
	for (unsigned long long f : fill) { total += f; most = std::max(most, f); }
	if (h_total) *h_total = total;
	if (total >= (1ull << 32)) return mcom_fail(ctx, MCOM_E_ARG, "too many minimizers");
	if (most > arena_cap) {                           // room that is enough for every arena: the fullest one times their number
		if (h_total) *h_total = (most + 1) * arenas;
		return mcom_fail(ctx, MCOM_E_OVERFLOW, "%llu minimizers but room for %zu", total, cap);
	}
	if (total == 0) return MCOM_OK;
	MCOM_LAUNCH(k_sketch_gather, dim3((unsigned)((n * 64 + 255) / 256)), dim3(256), 0, ctx->stream, piece_off, chunks, tmp, n, d_moff, d_out);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

// ------------------------------------------------------------------------------------------------
// pack: one thread per output word; contig c owns words [coff[c], coff[c+1]) of which the last is padding
// ------------------------------------------------------------------------------------------------
// A block of 256 threads makes 256 consecutive words.  The contig offsets the block needs (at most 130 contigs: a
// contig owns at least two words) and the characters behind its words (one contiguous range of the concatenation, at
// most 8 KB) are staged in LDS with coalesced loads; a thread then finds its contig by a search in LDS and packs 32
// characters with word-wide bit tricks instead of 22 dependent global loads and a byte loop.
#define PK_T 256
__global__ __launch_bounds__(PK_T) void k_pack_contigs(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ off,
                                                       const uint64_t *__restrict__ coff, uint32_t n, uint64_t total_words,
                                                       uint64_t *__restrict__ cbits, uint32_t n_first, uint64_t w_lo = 0, uint64_t w_hi = ~0ull)
{
	__shared__ uint64_t CO[PK_T / 2 + 4], OF[PK_T / 2 + 4];
	__shared__ uint32_t SB[PK_T * 8 + 16];
	__shared__ uint32_t srch[16];
	__shared__ uint64_t lo_s, hi_s;
	const uint64_t g0 = w_lo + (uint64_t)blockIdx.x * PK_T;                    // (w_lo, w_hi: the words [w_lo, w_hi) only -- a rank's share of the set)
	if (n_first < n) {                                                       // only the first n_first contigs: the words below coff[n_first]
		total_words = coff[n_first]; n = n_first;
		if (g0 >= total_words) return;
	}
	const uint64_t off_n = off[n];                                           // (read before total_words shrinks to the share's end: the end of the concatenation)
	const uint64_t all_words = total_words;
	if (w_hi < total_words) total_words = w_hi;
	if (g0 >= total_words) return;
	const uint64_t g = g0 + threadIdx.x;
	const uint32_t c0 = mcom_block_search(n, [&](uint32_t c) { return coff[c] <= g0; }, srch);
	if (threadIdx.x == 0) { lo_s = 0; hi_s = 0; }
	__syncthreads();
	const int NC = PK_T / 2 + 2;
	for (int t = threadIdx.x; t < NC; t += PK_T) {
		const uint64_t c = (uint64_t)c0 + t;
		CO[t] = c < n ? coff[c] : (c == n ? all_words : ~0ull);           // coff may hold n entries only
		OF[t] = c <= n ? off[c] : off[n];
	}
	__syncthreads();
	// my contig: last t with CO[t] <= g
	uint64_t mybyte = 0, myend = 0; bool live = g < total_words;
	int nb = 0;
	if (live) {
		int lo = 0, hi = NC;
		while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (CO[mid] <= g) lo = mid; else hi = mid; }
		const uint64_t wi = g - CO[lo], len = OF[lo + 1] - OF[lo];
		const uint64_t b0 = wi * 32;
		nb = b0 < len ? (int)(len - b0 < 32 ? len - b0 : 32) : 0;
		mybyte = OF[lo] + (b0 < len ? b0 : len); myend = mybyte + (uint64_t)nb;
	}
	const uint64_t last_live = total_words - g0 < PK_T ? total_words - g0 - 1 : PK_T - 1;
	if (threadIdx.x == 0) lo_s = mybyte;
	if (threadIdx.x == last_live) hi_s = myend;
	__syncthreads();
	const uint64_t start4 = lo_s & ~3ull;
	const uint64_t ndw = hi_s > start4 ? (hi_s - start4 + 3) / 4 + 1 : 0;            // one more word for the funnel shift
	const uint64_t seq_dw = (off_n + 3) / 4;                                           // do not read past the concatenation
	const uint32_t *seq32 = (const uint32_t*)(seq + start4);                           // the base pointer is 4-byte aligned
	for (uint64_t i = threadIdx.x; i < ndw; i += PK_T) SB[i] = (start4 / 4 + i) < seq_dw ? seq32[i] : 0u;
	__syncthreads();
	if (!live) return;
	uint64_t v = 0;
	if (nb) {
		const uint32_t boff = (uint32_t)(mybyte - start4), d = boff >> 2, sh = (boff & 3) * 8;
#pragma unroll
		for (int q = 0; q < 8; ++q) {
			const uint32_t a = SB[d + q], b = SB[d + q + 1];
			const uint32_t w4 = sh ? (a >> sh) | (b << (32 - sh)) : a;                  // four characters
			const uint32_t u = w4 & 0xDFDFDFDFu;                                       // fold case
			// per byte: 0x80 where the character is A, C, G or T
			auto eq = [](uint32_t x, uint32_t pat) { const uint32_t z = x ^ pat; return ~(((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z | 0x7F7F7F7Fu); };
			const uint32_t ok = eq(u, 0x41414141u) | eq(u, 0x43434343u) | eq(u, 0x47474747u) | eq(u, 0x54545454u);
			uint32_t c = ((w4 >> 1) ^ (w4 >> 2)) & 0x03030303u;                        // A0 C1 G2 T3
			c &= (ok >> 7) * 3u;                                                       // anything else packs as A
			c = (c | (c >> 6)) & 0x000F000Fu;
			c = (c | (c >> 12)) & 0xFFu;
			v |= (uint64_t)c << (8 * q);
		}
		if (nb < 32) v &= (1ull << (2 * nb)) - 1;
	}
	cbits[g] = v;
}

extern "C" int mcom_pack_contigs_words(mcom_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_off, const uint64_t *d_coff, uint32_t n,
                                       uint64_t total_words, uint64_t *d_cbits, uint64_t w_lo, uint64_t w_hi)
{
	if (!ctx) return MCOM_E_ARG;
	if (w_hi > total_words) w_hi = total_words;
	if (n == 0 || total_words == 0 || w_lo >= w_hi) return MCOM_OK;
	if (!d_seq || !d_off || !d_coff || !d_cbits) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	const uint64_t blocks = (w_hi - w_lo + 255) / 256;
	if (blocks >= (1ull << 31)) return mcom_fail(ctx, MCOM_E_ARG, "too many words");
	if (((uintptr_t)d_seq & 3) != 0) return mcom_fail(ctx, MCOM_E_ARG, "contig strings must start at a 4-byte boundary");
	MCOM_LAUNCH(k_pack_contigs, dim3((unsigned)blocks), dim3(PK_T), 0, ctx->stream, d_seq, d_off, d_coff, n, total_words, d_cbits, n, w_lo, w_hi);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}
extern "C" int mcom_pack_contigs(mcom_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_off, const uint64_t *d_coff, uint32_t n,
                                 uint64_t total_words, uint64_t *d_cbits)
{
	return mcom_pack_contigs_words(ctx, d_seq, d_off, d_coff, n, total_words, d_cbits, 0, total_words);
}

// ------------------------------------------------------------------------------------------------
// unpack: the strings of n contigs from their packed words -- the inverse of k_pack_contigs for strings of upper-case ACGT, which is
// what every consensus kernel writes.  Several GPUs send a new contig ONCE, as packed words (a quarter of a byte per base instead of
// the string AND the words: host/mcom_pipeline.cpp, "merged contigs"), and every rank makes the strings it did not build itself.
// Sixteen lanes per contig, a lane writes aligned 16-byte pieces of the concatenation (one store; the sixteen lanes 256 bytes in a row):
// no search for "whose byte is this" -- a first form, one block per 4 KB of the concatenation with a block-wide search through the offsets
// in front, took 22 - 35 ms over the eight ranks of a 64 M-read job whatever its block size, and this form with the end pieces read and written
// byte by byte 28; 15 as it stands (1.9 ms per rank for 2.5 GB of strings).  The piece at either end of a contig, shared with its neighbour, is
// made in registers like any other and stored byte by byte, each contig its own bytes.  Bytes outside [off[0], off[n]) are not touched.
// ------------------------------------------------------------------------------------------------
// eight 2-bit codes (the low 16 bits of x) as eight characters, code i in byte i
__device__ __forceinline__ uint64_t up_chars8(uint64_t x)
{
	uint64_t v = x & 0xFFFFull;
	v = (v | (v << 24)) & 0x000000FF000000FFull;
	v = (v | (v << 12)) & 0x000F000F000F000Full;
	v = (v | (v << 6)) & 0x0303030303030303ull;
	const uint64_t b0 = v & 0x0101010101010101ull, b1 = (v >> 1) & 0x0101010101010101ull, bb = b0 & b1;
	// A 0x41, C 0x43 (+2), G 0x47 (+6), T 0x54 (+2 +6 +11): no byte carries into its neighbour
	return 0x4141414141414141ull + (b0 << 1) + (b1 << 1) + (b1 << 2) + bb + (bb << 1) + (bb << 3);
}
__global__ __launch_bounds__(256) void k_unpack_contigs(const uint64_t *__restrict__ cbits, const uint64_t *__restrict__ coff, const uint64_t *__restrict__ off,
                                                        uint32_t n, uint8_t *__restrict__ seq)
{
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t c = t >> 4;
	if (c >= n) return;
	const uint64_t s = off[c], e = off[c + 1];
	const uint64_t *w = cbits + coff[c];
	for (uint64_t gb = (s & ~15ull) + 16 * (t & 15); gb < e; gb += 256) {
		// the sixteen characters of this piece in two registers, whether the contig covers all of it or not: the piece at a contig's head
		// starts d bytes in front of the contig (its codes moved up to their bytes' places), the one at its tail reads into the padding word
		uint64_t codes;
		if (gb >= s) {
			const uint64_t idx = gb - s;
			const int sh = (int)(idx & 31) * 2;
			const uint64_t w0 = w[idx >> 5];
			codes = sh > 32 ? (w0 >> sh) | (w[(idx >> 5) + 1] << (64 - sh)) : w0 >> sh;
		} else codes = w[0] << (2 * (int)(s - gb));
		const uint64_t c_lo = up_chars8(codes), c_hi = up_chars8(codes >> 16);
		if (gb >= s && gb + 16 <= e) *(ulonglong2*)(seq + gb) = make_ulonglong2(c_lo, c_hi);
		else {
			const int i_lo = gb >= s ? 0 : (int)(s - gb), i_hi = gb + 16 <= e ? 16 : (int)(e - gb);
#pragma unroll
			for (int i = 0; i < 16; ++i)
				if (i >= i_lo && i < i_hi) seq[gb + i] = (uint8_t)((i < 8 ? c_lo : c_hi) >> (8 * (i & 7)));
		}
	}
}

extern "C" int mcom_unpack_contigs(mcom_ctx *ctx, const uint64_t *d_cbits, const uint64_t *d_coff, const uint64_t *d_off, uint32_t n,
                                   uint64_t byte_lo, uint64_t byte_hi, uint8_t *d_seq)
{
	if (!ctx) return MCOM_E_ARG;
	if (n == 0 || byte_hi <= byte_lo) return MCOM_OK;
	if (!d_cbits || !d_coff || !d_off || !d_seq) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	if (((uintptr_t)d_seq & 15) != 0) return mcom_fail(ctx, MCOM_E_ARG, "contig strings must start at a 16-byte boundary");
	const uint64_t blocks = ((uint64_t)n * 16 + 255) / 256;
	MCOM_LAUNCH(k_unpack_contigs, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_cbits, d_coff, d_off, n, d_seq);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

// A merge round changes the merged contigs only: they (the first n_first of the new set) are packed, the packed words of the
// untouched ones (contig nj + u of the new set = contig keepidx[u] of the old one) are copied -- a quarter of a byte per base
// instead of a byte.
__global__ __launch_bounds__(256) void k_packed_carry(const uint64_t *__restrict__ cbits_old, const uint64_t *__restrict__ coff_old,
                                                      const uint32_t *__restrict__ keepidx, size_t nkeep, size_t first,
                                                      const uint64_t *__restrict__ coff_new, uint64_t *__restrict__ cbits_new, uint64_t total_words)
{
	if (blockIdx.x == 0 && threadIdx.x < 2) cbits_new[total_words + threadIdx.x] = 0;   // the two words behind the set (a window at its very end reads past it)
	const size_t u = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;       // sixteen lanes per contig
	if (u >= nkeep) return;
	const int lane = threadIdx.x & 15;
	const uint32_t i = keepidx[u];
	const uint64_t s0 = coff_old[i], cnt = coff_old[i + 1] - s0, d0 = coff_new[first + u];
	for (uint64_t t = lane; t < cnt; t += 16) cbits_new[d0 + t] = cbits_old[s0 + t];
}
extern "C" int mcom_pack_contigs_merged(mcom_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_off, const uint64_t *d_coff, uint32_t n,
                                        uint64_t total_words, uint32_t n_first, const uint64_t *d_cbits_old, const uint64_t *d_coff_old,
                                        const uint32_t *d_keepidx, uint64_t *d_cbits)
{
	if (!ctx) return MCOM_E_ARG;
	if (n == 0 || total_words == 0) return MCOM_OK;
	if (n_first > n) return mcom_fail(ctx, MCOM_E_ARG, "bad pack arguments");
	if (!d_seq || !d_off || !d_coff || !d_cbits || (n_first < n && (!d_cbits_old || !d_coff_old || !d_keepidx))) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	const uint64_t blocks = (total_words + 255) / 256;
	if (blocks >= (1ull << 31)) return mcom_fail(ctx, MCOM_E_ARG, "too many words");
	if (((uintptr_t)d_seq & 3) != 0) return mcom_fail(ctx, MCOM_E_ARG, "contig strings must start at a 4-byte boundary");
	if (n_first) MCOM_LAUNCH(k_pack_contigs, dim3((unsigned)blocks), dim3(PK_T), 0, ctx->stream, d_seq, d_off, d_coff, n, total_words, d_cbits, n_first);
	const size_t nkeep = n - n_first;
	if (nkeep) MCOM_LAUNCH(k_packed_carry, dim3((unsigned)((nkeep * 16 + 255) / 256)), dim3(256), 0, ctx->stream, d_cbits_old, d_coff_old, d_keepidx, nkeep,
	                              (size_t)n_first, d_coff, d_cbits, total_words);
	else MCOM_HIP(ctx, hipMemsetAsync(d_cbits + total_words, 0, 16, ctx->stream));
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

// ------------------------------------------------------------------------------------------------
// minimizer index: records sorted by x (stable: equal minimizers keep their push order, which is the order
// the reference's bucket sort keeps for buckets of <= 64 entries, ksort.h:155) + exact hash table
// ------------------------------------------------------------------------------------------------
struct mcom_idx {
	size_t n;
	mcom_mm128 *rec;     // sorted records
	McomTable tab;
	// a part being built (mcom_idx_sort_part .. mcom_idx_table_part): its place in rec, the starts of its buckets relative to it
	int k = 0, b = 0;
	size_t part_n = 0, part_base = 0;
	uint32_t *part_bst = nullptr;
};

extern "C" void mcom_idx_destroy(mcom_ctx *ctx, mcom_idx *mi)
{
	if (!mi) return;
	// (no wait: the blocks go back to the pool when the context's stream is next synchronised -- the lookups that read them may still be
	// on their way, and the pool is shared with other contexts)
	if (ctx) { mcom_dfree_later(ctx, mi->rec); mcom_dfree_later(ctx, mi->part_bst); mcom_dfree_later(ctx, mi->tab.slots); mi->tab.slots = nullptr; }
	else { if (mi->rec) mcom_dfree(mi->rec); if (mi->part_bst) mcom_dfree(mi->part_bst); mcom_table_free(&mi->tab); }
	delete mi;
}

int mcom_flag_sort_ranges(mcom_ctx *ctx, mcom_mm128 *d_rec, const uint32_t *d_bstart, uint32_t nr, uint32_t max_range, uint32_t *d_overflow);
int mcom_bucket_starts(mcom_ctx *ctx, const mcom_mm128 *d_rec, size_t n, int bits, uint32_t *d_bstart);
int mcom_flag_sort_buckets(mcom_ctx *ctx, const mcom_mm128 *d_in, mcom_mm128 *d_out, const uint32_t *d_bstart, uint32_t nr, int low_bits,
                           uint32_t max_range, uint32_t *d_overflow);

// ---- the index in steps: a caller with several GPUs builds it by bucket range and all-gathers the parts (include/mcom.h) ----
extern "C" int mcom_idx_create(mcom_ctx *ctx, size_t n, int k, int b, mcom_idx **out)
{
	if (!ctx || !out) return MCOM_E_ARG;
	*out = nullptr;
	if (k < 1 || k > 31) return mcom_fail(ctx, MCOM_E_ARG, "k=%d out of range", k);
	if (b < 0 || b > 20 || b > 2 * k) return mcom_fail(ctx, MCOM_E_ARG, "bucket bits %d out of range", b);
	if (n >= (1ull << 32)) return mcom_fail(ctx, MCOM_E_ARG, "too many records");
	mcom_idx *mi = new mcom_idx();
	mi->n = n; mi->rec = nullptr; mi->tab.slots = nullptr; mi->tab.region = 0; mi->tab.bbits = 0; mi->tab.log2cap = 0; mi->tab.numkeys = 0; mi->tab.maxrun = 0;
	mi->k = k; mi->b = b;
	hipError_t e = mcom_dmalloc(&mi->rec, (n ? n : 1) * sizeof(mcom_mm128));
	if (e != hipSuccess) { mi->rec = nullptr; mcom_idx_destroy(ctx, mi); return mcom_fail(ctx, MCOM_E_NOMEM, "index records: %s", hipGetErrorString(e)); }
	if (b > 0) {
		e = mcom_dmalloc(&mi->part_bst, (((size_t)1 << b) + 2) * 4);
		if (e != hipSuccess) { mi->part_bst = nullptr; mcom_idx_destroy(ctx, mi); return mcom_fail(ctx, MCOM_E_NOMEM, "index bucket starts: %s", hipGetErrorString(e)); }
	}
	*out = mi;
	return MCOM_OK;
}

// the records of some buckets (all of them: the whole index), in the order the reference pushes them, sorted into rec[base ...):
// bucket by bucket, every bucket in radix_sort_128x's exact element order (kthread_idx.c:126)
extern "C" int mcom_idx_sort_part(mcom_ctx *ctx, mcom_idx *mi, const mcom_mm128 *d_rec, size_t n, size_t base_rec, uint32_t *h_max_bucket)
{
	if (!ctx || !mi) return MCOM_E_ARG;
	if (h_max_bucket) *h_max_bucket = 0;
	if (base_rec + n > mi->n) return mcom_fail(ctx, MCOM_E_ARG, "index part beyond the index");
	if (n && !d_rec) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	mi->part_n = n; mi->part_base = base_rec;
	const int b = mi->b, k = mi->k;
	const size_t sort_b = mcom_sort_ws_bytes(n);
	int rc = mcom_ws_reserve(ctx, sort_b + 256);
	if (rc) return rc;
	char *base = (char*)ctx->ws;
	mcom_mm128 *part = mi->rec + base_rec;
	if (b == 0) {
		if (n) MCOM_HIP(ctx, hipMemcpyAsync(part, d_rec, n * sizeof(mcom_mm128), hipMemcpyDeviceToDevice, ctx->stream));
		return n ? mcom_sort_by_x(ctx, part, n, 2 * k, base) : MCOM_OK;                    // stable: equal minimizers keep their input order
	}
	const uint32_t nb = 1u << b;
	if (!n) { MCOM_HIP(ctx, hipMemsetAsync(mi->part_bst, 0, ((size_t)nb + 1) * 4, ctx->stream)); return MCOM_OK; }
	// records go to bucket x & (2^b-1) in input order (kthread_bucket.c:468-473), every bucket is then sorted by radix_sort_128x.
	// Round 4: the passes read the caller's records and leave their result in the workspace, the bucket sort reads it there and
	// writes the index (two copies of all records per build before: into the index in front of the passes, out of the workspace behind the sort)
	mcom_mm128 *byb = nullptr;                                                    // the records by bucket (in the sort workspace)
	if (d_rec == part) return mcom_fail(ctx, MCOM_E_ARG, "the records of an index part must not lie in the index itself");
	rc = mcom_sort_by_low_bits_into_ws(ctx, d_rec, part, n, b, base, &byb);
	if (!rc) rc = mcom_bucket_starts(ctx, byb, n, b, mi->part_bst);
	if (rc) return rc;
	std::vector<uint32_t> hb(nb + 1);
	MCOM_HIP(ctx, mcom_d2h_async(ctx, hb.data(), mi->part_bst, (nb + 1) * 4));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	uint32_t mx = 0;
	for (uint32_t q = 0; q < nb; ++q) mx = std::max(mx, hb[q + 1] - hb[q]);
	if (h_max_bucket) *h_max_bucket = mx;
	uint32_t *ovf = (uint32_t*)mcom_zeroed(ctx, base + sort_b, 4);
	if (!ovf) return mcom_fail(ctx, MCOM_E_HIP, "clear");
	if (2 * k - b <= 48) rc = mcom_flag_sort_buckets(ctx, byb, part, mi->part_bst, nb, b, mx, ovf);   // compact elements: x >> b fits 48 bits
	else {
		MCOM_HIP(ctx, hipMemcpyAsync(part, byb, n * sizeof(mcom_mm128), hipMemcpyDeviceToDevice, ctx->stream));
		rc = mcom_flag_sort_ranges(ctx, part, mi->part_bst, nb, mx, ovf);
	}
	if (rc) return rc;
	uint32_t ov = 0;
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &ov, ovf, 4));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	if (ov) return mcom_fail(ctx, MCOM_E_OVERFLOW, "index bucket sort ran out of range stack");
	return MCOM_OK;
}

int mcom_table_alloc_bucketed(mcom_ctx *ctx, int bbits, uint32_t max_bucket, McomTable *t);
int mcom_table_fill_buckets(mcom_ctx *ctx, const mcom_mm128 *sorted, const uint32_t *bstart, uint32_t bucket0, uint32_t bucket1, uint32_t start_base, McomTable *t);

// The table over the sorted part: the regions of buckets [bucket0, bucket1) (those the part holds), sized for the fullest bucket of
// the WHOLE index (max_bucket_all; every builder passes the same).  MCOM_E_OVERFLOW: a bucket too large for a region in LDS -- the
// caller puts all sorted records together and calls mcom_idx_table_global.
extern "C" int mcom_idx_table_part(mcom_ctx *ctx, mcom_idx *mi, uint32_t max_bucket_all, uint32_t bucket0, uint32_t bucket1)
{
	if (!ctx || !mi) return MCOM_E_ARG;
	if (mi->b < 1) return mcom_fail(ctx, MCOM_E_ARG, "an index without buckets has one global table");
	if (bucket0 > bucket1 || bucket1 > (1u << mi->b)) return mcom_fail(ctx, MCOM_E_ARG, "bad bucket range");
	if (!mi->tab.slots) { int rc = mcom_table_alloc_bucketed(ctx, mi->b, max_bucket_all, &mi->tab); if (rc) return rc; }
	return mcom_table_fill_buckets(ctx, mi->rec + mi->part_base, mi->part_bst, bucket0, bucket1, (uint32_t)mi->part_base, &mi->tab);
}
// ... or the regions of ALL buckets over the whole sorted array, once the parts of all builders are there (round 5: a region is sized for
// the fullest bucket of the index and a quarter full on average, so the regions were three quarters of what the builders sent each
// other -- 3 of 13.8 GB per rank and step at eight ranks; making them from the records is 0.15 ms per build).  Same MCOM_E_OVERFLOW.
extern "C" int mcom_idx_table_all(mcom_ctx *ctx, mcom_idx *mi, uint32_t max_bucket_all)
{
	if (!ctx || !mi) return MCOM_E_ARG;
	if (mi->b < 1) return mcom_fail(ctx, MCOM_E_ARG, "an index without buckets has one global table");
	if (!mi->tab.slots) { int rc = mcom_table_alloc_bucketed(ctx, mi->b, max_bucket_all, &mi->tab); if (rc) return rc; }
	int rc = mcom_bucket_starts(ctx, mi->rec, mi->n, mi->b, mi->part_bst);           // (the array is bucket-major: the parts are bucket ranges in order)
	if (rc) return rc;
	mi->part_base = 0; mi->part_n = mi->n;
	return mcom_table_fill_buckets(ctx, mi->rec, mi->part_bst, 0, 1u << mi->b, 0, &mi->tab);
}
extern "C" int mcom_idx_table_global(mcom_ctx *ctx, mcom_idx *mi)
{
	if (!ctx || !mi) return MCOM_E_ARG;
	mcom_table_free(&mi->tab);
	const size_t n = mi->n;
	const size_t head_b = ((n * 4) + 255) & ~(size_t)255, scr_b = ((mcom_scan_scratch_elems(n) * 4 + 1024) + 255) & ~(size_t)255;
	int rc = mcom_ws_reserve(ctx, head_b + scr_b + 256);
	if (rc) return rc;
	char *base = (char*)ctx->ws;
	return mcom_table_build(ctx, mi->rec, n, (uint32_t*)base, (uint32_t*)(base + head_b), (uint32_t*)(base + head_b + scr_b), &mi->tab);
}
// what the parts of other builders are received into: the sorted records [n], the table (slots of 16 bytes, *region per bucket;
// *region = 0: one global table, nothing to exchange)
extern "C" int mcom_idx_buffers(mcom_idx *mi, mcom_mm128 **d_rec, uint64_t **d_slots, uint32_t *region)
{
	if (!mi) return MCOM_E_ARG;
	if (d_rec) *d_rec = mi->rec;
	if (d_slots) *d_slots = mi->tab.slots;
	if (region) *region = mi->tab.region;
	return MCOM_OK;
}

extern "C" int mcom_idx_build(mcom_ctx *ctx, const mcom_mm128 *d_rec, size_t n, int k, int b, mcom_idx **out)
{
	if (!ctx || !out) return MCOM_E_ARG;
	*out = nullptr;
	if (n && !d_rec) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	mcom_idx *mi = nullptr;
	int rc = mcom_idx_create(ctx, n, k, b, &mi);
	if (rc) return rc;
	uint32_t mx = 0;
	rc = mcom_idx_sort_part(ctx, mi, d_rec, n, 0, &mx);
	if (!rc) {
		if (n && b > 0) { rc = mcom_idx_table_part(ctx, mi, mx, 0, 1u << b); if (rc == MCOM_E_OVERFLOW) rc = mcom_idx_table_global(ctx, mi); }
		else rc = mcom_idx_table_global(ctx, mi);
	}
	if (rc) { mcom_idx_destroy(ctx, mi); return rc; }
	*out = mi;
	return MCOM_OK;
}

__global__ void k_idx_get(const uint64_t *__restrict__ slots, uint32_t log2cap, uint32_t region, uint32_t bbits, const uint64_t *__restrict__ x, size_t n,
                          uint32_t *__restrict__ start, uint32_t *__restrict__ count)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint32_t s = 0, c = 0;
	if (x[i] != U64MAX) mcom_table_find_any(slots, log2cap, region, bbits, x[i], s, c);
	start[i] = s; count[i] = c;
}

extern "C" int mcom_idx_get(mcom_ctx *ctx, const mcom_idx *mi, const uint64_t *d_x, size_t n, uint32_t *d_start, uint32_t *d_count)
{
	if (!ctx || !mi) return MCOM_E_ARG;
	if (n == 0) return MCOM_OK;
	if (!d_x || !d_start || !d_count) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	MCOM_LAUNCH(k_idx_get, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, mi->tab.slots, mi->tab.log2cap, mi->tab.region, mi->tab.bbits, d_x, n, d_start, d_count);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

extern "C" int mcom_idx_records(mcom_ctx *ctx, const mcom_idx *mi, mcom_mm128 *d_out, size_t *n)
{
	if (!ctx || !mi) return MCOM_E_ARG;
	if (n) *n = mi->n;
	if (d_out && mi->n) MCOM_HIP(ctx, hipMemcpyAsync(d_out, mi->rec, mi->n * sizeof(mcom_mm128), hipMemcpyDeviceToDevice, ctx->stream));
	return MCOM_OK;
}

// ------------------------------------------------------------------------------------------------
// match_pro on packed contigs: mismatches over the whole overlap of A and B anchored at A[i] ~ B[j]
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t bits_at(const uint64_t *w, uint64_t bitoff)
{
	const uint64_t wi = bitoff >> 6; const int sh = (int)(bitoff & 63);
	return sh ? (w[wi] >> sh) | (w[wi + 1] << (64 - sh)) : w[wi];      // relies on the padding word after every contig
}
__device__ __forceinline__ uint32_t match_pro_packed(const uint64_t *A, uint32_t lenA, const uint64_t *B, uint32_t lenB, int i_, int j_,
                                                     uint32_t stop_above)
{
	const long d = (long)i_ - (long)j_;                                    // A[p] is compared with B[p - d]
	const long lo = d > 0 ? d : 0;
	long hi = (long)lenB + d; if (hi > (long)lenA) hi = (long)lenA;
	uint32_t mis = 0;
	for (long p = lo; p < hi; p += 32) {
		const long nb = hi - p < 32 ? hi - p : 32;
		uint64_t x = bits_at(A, 2 * (uint64_t)p) ^ bits_at(B, 2 * (uint64_t)(p - d));
		x = (x | (x >> 1)) & 0x5555555555555555ull;
		if (nb < 32) x &= (1ull << (2 * nb)) - 1;
		mis += (uint32_t)__popcll(x);
		if (mis > stop_above) break;
	}
	return mis;
}

__global__ void k_match_pro(const uint64_t *__restrict__ cbits, const uint64_t *__restrict__ coff, const uint32_t *__restrict__ clen,
                            const uint32_t *__restrict__ a, const uint32_t *__restrict__ pa, const uint32_t *__restrict__ b,
                            const uint32_t *__restrict__ pb, size_t n, uint32_t *__restrict__ out)
{
	const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n) return;
	out[t] = match_pro_packed(cbits + coff[a[t]], clen[a[t]], cbits + coff[b[t]], clen[b[t]], (int)pa[t], (int)pb[t], 0xFFFFFFFFu);
}

extern "C" int mcom_match_pro(mcom_ctx *ctx, const uint64_t *d_cbits, const uint64_t *d_coff, const uint32_t *d_clen,
                              const uint32_t *d_a, const uint32_t *d_pos_a, const uint32_t *d_b, const uint32_t *d_pos_b, size_t n,
                              uint32_t *d_mismatch)
{
	if (!ctx) return MCOM_E_ARG;
	if (n == 0) return MCOM_OK;
	if (!d_cbits || !d_coff || !d_clen || !d_a || !d_pos_a || !d_b || !d_pos_b || !d_mismatch) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	MCOM_LAUNCH(k_match_pro, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_cbits, d_coff, d_clen, d_a, d_pos_a, d_b, d_pos_b, n, d_mismatch);
	MCOM_LAUNCH_CHECK(ctx);
	return MCOM_OK;
}

// ------------------------------------------------------------------------------------------------
// find_next, lookup part: for query minimizer q of contig i (all minimizers, in order) and every index hit
// in index order: other contig != i, same strand bit, match_pro <= cbthr  -> one passing candidate.
// The flags the reference also tests (:286) change while it merges, so they are left to the caller, who
// walks each contig's candidates in this order and takes the first whose partner is still free.
// ------------------------------------------------------------------------------------------------
// Round 4, second cut: a query minimizer of a contig that came through the round before unmerged (id >= n_new) can only pass with a
// NEW contig (k_fn_eval's comment), i.e. through a key that a new contig has in the index.  k_fn_newkeys marks those keys in a bit
// map (about 16 bits per such key, 128 KB .. 8 MB: L2 resident); such a query
// probes the table only where the map says so and lists no pair otherwise -- the pairs left out are exactly pairs k_fn_eval would
// fail without looking.
__device__ __forceinline__ uint32_t fn_key_bit(uint64_t x, int kbits) { return (uint32_t)((x * 0x9E3779B97F4A7C15ull) >> (64 - kbits)); }
// Which contigs are new in this round: ids [new_lo, new_lo + new_span) (new_span = 0: all of them, the first round).  The copying form of
// the merge rounds puts the new contigs at the head of the list (new_lo = 0, new_span = their number); the form that leaves the set where
// it is gives them the ids behind everybody else's (new_lo = first new id, new_span = 2^32 - 1).
__device__ __forceinline__ bool fn_is_new(uint32_t id, uint32_t new_lo, uint32_t new_span) { return id >= new_lo && (uint32_t)(id - new_lo) < new_span; }
__global__ void k_fn_newkeys(const mcom_mm128 *__restrict__ irec, size_t n, uint32_t new_lo, uint32_t new_span, int kbits, uint32_t *__restrict__ keymap)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const mcom_mm128 t = irec[i];
	if (fn_is_new((uint32_t)(t.y >> 32), new_lo, new_span)) { const uint32_t h = fn_key_bit(t.x, kbits); atomicOr(&keymap[h >> 5], 1u << (h & 31)); }
}
__global__ void k_fn_counts(const uint64_t *__restrict__ slots, uint32_t log2cap, uint32_t region, uint32_t bbits, const mcom_mm128 *__restrict__ q, size_t nq,
                            uint32_t new_lo, uint32_t new_span, int kbits, const uint32_t *__restrict__ keymap, uint32_t *__restrict__ hits, uint32_t *__restrict__ first)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nq) return;
	if (i == 0) hits[nq] = 0;                     // (the scan over nq + 1 counts leaves the number of pairs there)
	const mcom_mm128 t = q[i];
	uint32_t s = 0, c = 0;
	bool look = t.x != U64MAX;
	if (look && new_span && !fn_is_new((uint32_t)(t.y >> 32), new_lo, new_span)) { const uint32_t h = fn_key_bit(t.x, kbits); look = (keymap[h >> 5] >> (h & 31)) & 1u; }
	if (look) mcom_table_find_any(slots, log2cap, region, bbits, t.x, s, c);
	hits[i] = c; if (c) first[i] = s;              // the later passes read these instead of probing the table again (first: only where there are hits)
}
// The same over a set whose contigs are NOT stored in visiting order (round 5: a merge round leaves the unmerged contigs where they are and
// appends the merged ones; `ord` lists the contigs in visiting order, k_fn_qcounts + a scan give the first query number of each): sixteen
// lanes per contig walk its records; a query with hits leaves its y in qy (the pair kernels read it there: the records themselves are
// not in query order).
__global__ void k_fn_qcounts(const uint32_t *__restrict__ roff, const uint32_t *__restrict__ ord, size_t n, uint32_t *__restrict__ cnt)
{
	const size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (u > n) return;
	if (u == n) { cnt[u] = 0; return; }
	const uint32_t c = ord ? ord[u] : (uint32_t)u;
	cnt[u] = roff[c + 1] - roff[c];
}
__global__ __launch_bounds__(256) void k_fn_counts_ord(const uint64_t *__restrict__ slots, uint32_t log2cap, uint32_t region, uint32_t bbits,
                                                       const mcom_mm128 *__restrict__ rec, const uint32_t *__restrict__ roff, const uint32_t *__restrict__ ord,
                                                       const uint32_t *__restrict__ qoff, size_t n, uint32_t nq, uint32_t new_lo, uint32_t new_span, int kbits,
                                                       const uint32_t *__restrict__ keymap, uint32_t *__restrict__ hits, uint32_t *__restrict__ first,
                                                       uint64_t *__restrict__ qy)
{
	const size_t u = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
	if (u >= n) return;
	const int lane = threadIdx.x & 15;
	if (u == 0 && lane == 0) hits[nq] = 0;
	const uint32_t c = ord ? ord[u] : (uint32_t)u;
	const uint32_t r0 = roff[c], cnt = roff[c + 1] - r0, q0 = qoff[u];
	const bool old_contig = new_span && !fn_is_new(c, new_lo, new_span);
	// four records of the contig in flight per lane (a lane's loop is a chain of load -> probe -> store otherwise: the flat kernel has a
	// thread per query and the scheduler's other waves to hide that behind)
	for (uint32_t t0 = lane; t0 < cnt; t0 += 64) {
		mcom_mm128 v[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) { const uint32_t t = t0 + 16u * u; if (t < cnt) v[u] = rec[r0 + t]; else { v[u].x = U64MAX; v[u].y = 0; } }
		bool look[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			look[u] = v[u].x != U64MAX;
			if (look[u] && old_contig) { const uint32_t h = fn_key_bit(v[u].x, kbits); look[u] = (keymap[h >> 5] >> (h & 31)) & 1u; }
		}
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const uint32_t t = t0 + 16u * u;
			if (t >= cnt) break;
			uint32_t s = 0, k = 0;
			if (look[u]) mcom_table_find_any(slots, log2cap, region, bbits, v[u].x, s, k);
			hits[q0 + t] = k;
			if (k) { first[q0 + t] = s; qy[q0 + t] = v[u].y; }
		}
	}
}
// Round 4: the evaluation runs one thread per (query, hit) PAIR.  Three queries in four have no hit, and the others between one
// and thousands: with a thread per query (rounds 1-3) a wave ran as long as its busiest lane while most lanes had nothing to do, and
// every lane walked its hits one dependent gather after the other.  k_fn_expand names the query of every pair (the only loop over a
// query's hits left: plain stores), then every lane of k_fn_eval has one pair to test and k_fn_emit one to write.
__global__ void k_fn_expand(const uint32_t *__restrict__ pair_off, size_t nq, uint32_t *__restrict__ pair_q)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nq) return;
	const uint32_t p0 = pair_off[i], p1 = pair_off[i + 1];
	for (uint32_t p = p0; p < p1; ++p) pair_q[p] = (uint32_t)i;
}
// n_new: contigs [0, n_new) are new in this merge round (the merged ones head the list, cp_cluster order); 0 = all of them.  A pair of
// two contigs that both came through the round before unmerged cannot pass: it was a candidate then, with the same strings and the
// same positions, and a passing pair of two contigs that both stay unclaimed does not exist (the first of the two to be visited would
// have taken the other, kthread_cb.c:286-343) -- so it failed match_pro then and fails it now.
__global__ void k_fn_eval(const uint32_t *__restrict__ first, const mcom_mm128 *__restrict__ irec,
                          const mcom_mm128 *__restrict__ q, const uint64_t *__restrict__ qy, const uint32_t *__restrict__ pair_off, const uint32_t *__restrict__ pair_q, uint32_t n_pairs,
                          const uint64_t *__restrict__ cbits, const uint64_t *__restrict__ coff, const uint32_t *__restrict__ clen,
                          int cbthr, uint32_t new_lo, uint32_t new_span, uint32_t *__restrict__ pass)
{
	const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= n_pairs) return;
	const uint32_t i = pair_q[p];
	const uint32_t u = p - pair_off[i], s = first[i];
	const uint64_t my = qy ? qy[i] : q[i].y, y = irec[s + u].y;
	const uint32_t ci = (uint32_t)(my >> 32), pos_ori = (uint32_t)my >> 1;   // the id IS the contig index here (include/mcom.h, "contig ids")
	const uint32_t cj = (uint32_t)(y >> 32), pos = (uint32_t)y >> 1;
	uint32_t ok = 0;
	if (cj != ci && ((my ^ y) & 1) == 0 && (new_span == 0 || fn_is_new(ci, new_lo, new_span) || fn_is_new(cj, new_lo, new_span))) {
		const uint32_t mis = match_pro_packed(cbits + coff[ci], clen[ci], cbits + coff[cj], clen[cj], (int)pos_ori, (int)pos, (uint32_t)cbthr);
		ok = mis <= (uint32_t)cbthr;
	}
	pass[p] = ok;
	if (p == 0) pass[n_pairs] = 0;                                               // the scan over n_pairs + 1 flags leaves the number of passing pairs here
}
__global__ void k_fn_emit(const uint32_t *__restrict__ first, const mcom_mm128 *__restrict__ irec,
                          const mcom_mm128 *__restrict__ q, const uint64_t *__restrict__ qy, const uint32_t *__restrict__ pair_off, const uint32_t *__restrict__ pair_q,
                          const uint32_t *__restrict__ pass_pre, uint32_t n_pairs,
                          mcom_mm128 *__restrict__ out)
{
	const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= n_pairs) return;
	const uint32_t here = pass_pre[p], nxt = pass_pre[p + 1];                       // (n_pairs + 1 scanned flags)
	if (nxt == here) return;
	const uint32_t i = pair_q[p];
	mcom_mm128 v; v.x = qy ? qy[i] : q[i].y; v.y = irec[first[i] + (p - pair_off[i])].y;   // x = query y (contig i, pos_ori, dir), y = hit y
	out[here] = v;
}

extern "C" int mcom_find_next_candidates(mcom_ctx *ctx, const mcom_idx *mi, const mcom_mm128 *d_query, size_t n_query,
                                         const uint64_t *d_cbits, const uint64_t *d_coff, const uint32_t *d_clen, int cbthr,
                                         mcom_mm128 *d_out, size_t cap, uint64_t *h_counts)
{
	return mcom_find_next_candidates_new(ctx, mi, d_query, n_query, d_cbits, d_coff, d_clen, cbthr, 0, d_out, cap, h_counts);
}
// the two forms share everything behind the hit counts: queries given in visiting order (d_query) or contigs given in visiting order
// over records that lie elsewhere (d_roff + d_ord over d_query as the record array)
static int find_next_impl(mcom_ctx *ctx, const mcom_idx *mi, const mcom_mm128 *d_query, size_t n_query, const uint32_t *d_roff, const uint32_t *d_ord, size_t n_contigs,
                          const uint64_t *d_cbits, const uint64_t *d_coff, const uint32_t *d_clen, int cbthr, uint32_t new_lo, uint32_t new_span, uint32_t n_new_hint,
                          mcom_mm128 *d_out, size_t cap, uint64_t *h_counts)
{
	if (!ctx || !mi) return MCOM_E_ARG;
	if (h_counts) { h_counts[0] = h_counts[1] = 0; }
	const bool by_contig = d_roff != nullptr;
	if (by_contig ? n_contigs == 0 : n_query == 0) return MCOM_OK;
	if (!d_query || !d_cbits || !d_coff || !d_clen) return mcom_fail(ctx, MCOM_E_ARG, "null device pointer");
	uint32_t *qoff = nullptr;
	// (the blocks go back to the pool at the context's next synchronisation: the last kernels that read them are still on their way when
	// this returns -- round 5: the call ended with a wait of its own for nothing but these frees)
	struct FirstGuard { mcom_ctx *c; uint32_t *p; ~FirstGuard() { mcom_dfree_later(c, p); } };
	if (by_contig) {                                                             // first query number of every contig, in visiting order
		if (mcom_dmalloc(&qoff, (n_contigs + 1) * 4) != hipSuccess) return mcom_fail(ctx, MCOM_E_NOMEM, "candidate buffers");
	}
	FirstGuard qoff_guard{ctx, qoff};
	if (by_contig) {
		MCOM_LAUNCH(k_fn_qcounts, dim3((unsigned)((n_contigs + 1 + 255) / 256)), dim3(256), 0, ctx->stream, d_roff, d_ord, n_contigs, qoff);
		MCOM_LAUNCH_CHECK(ctx);
		int rc0 = mcom_scan_u32(ctx, qoff, qoff, n_contigs + 1, nullptr);
		if (rc0) return rc0;
		uint32_t nq = 0;
		MCOM_HIP(ctx, mcom_d2h_async(ctx, &nq, qoff + n_contigs, 4));
		MCOM_HIP(ctx, mcom_stream_sync(ctx));
		n_query = nq;
		if (n_query == 0) return MCOM_OK;
	}
	if (n_query >= (1ull << 31)) return mcom_fail(ctx, MCOM_E_ARG, "too many query minimizers");
	// pass 1: hits per query -> pair offsets
	const size_t nq1 = n_query + 1;
	const size_t hit_b = (nq1 * 4 + 255) & ~(size_t)255;
	const size_t scr1_b = (mcom_scan_scratch_elems(nq1) * 4 + 1024 + 255) & ~(size_t)255;
	// ~16 map bits per key of a new contig (a contig has about six keys in the index), 8 MB at most: a map that stays in the L2s beats a
	// sparser one that does not (measured: 32 bits per key up to 32 MB made the rounds with a million new contigs 7 % slower)
	int kbits = 20;
	while (kbits < 26 && ((uint64_t)1 << kbits) < (uint64_t)n_new_hint * 6 * 16) ++kbits;
	const size_t map_b = new_span ? (size_t)1 << (kbits - 3) : 0;
	int rc = mcom_ws_reserve(ctx, hit_b + scr1_b + map_b);
	if (rc) return rc;
	uint32_t *hits = (uint32_t*)ctx->ws;
	uint32_t *keymap = new_span ? (uint32_t*)((char*)ctx->ws + hit_b + scr1_b) : nullptr;
	uint32_t *first = nullptr, *pair_off = nullptr;                            // pair_off: the scanned counts, in an allocation of its own (the workspace may move below)
	uint64_t *qy = nullptr;
	if (mcom_dmalloc(&first, nq1 * 4) != hipSuccess) return mcom_fail(ctx, MCOM_E_NOMEM, "candidate buffers");
	FirstGuard first_guard{ctx, first};
	if (mcom_dmalloc(&pair_off, nq1 * 4) != hipSuccess) return mcom_fail(ctx, MCOM_E_NOMEM, "candidate buffers");
	FirstGuard off_guard{ctx, pair_off};
	if (by_contig && mcom_dmalloc(&qy, nq1 * 8) != hipSuccess) return mcom_fail(ctx, MCOM_E_NOMEM, "candidate buffers");
	FirstGuard qy_guard{ctx, (uint32_t*)qy};
	const unsigned qb = (unsigned)((n_query + 255) / 256);
	if (new_span) {
		MCOM_HIP(ctx, hipMemsetAsync(keymap, 0, map_b, ctx->stream));
		if (mi->n) MCOM_LAUNCH(k_fn_newkeys, dim3((unsigned)((mi->n + 255) / 256)), dim3(256), 0, ctx->stream, mi->rec, mi->n, new_lo, new_span, kbits, keymap);
	}
	if (by_contig)
		MCOM_LAUNCH(k_fn_counts_ord, dim3((unsigned)((n_contigs * 16 + 255) / 256)), dim3(256), 0, ctx->stream, mi->tab.slots, mi->tab.log2cap, mi->tab.region, mi->tab.bbits,
		            d_query, d_roff, d_ord, (const uint32_t*)qoff, n_contigs, (uint32_t)n_query, new_lo, new_span, kbits, keymap, hits, first, qy);
	else
		MCOM_LAUNCH(k_fn_counts, dim3(qb), dim3(256), 0, ctx->stream, mi->tab.slots, mi->tab.log2cap, mi->tab.region, mi->tab.bbits, d_query, n_query, new_lo, new_span, kbits, keymap, hits, first);
	MCOM_LAUNCH_CHECK(ctx);
	rc = mcom_scan_u32(ctx, hits, pair_off, nq1, (uint32_t*)((char*)ctx->ws + hit_b));
	if (rc) return rc;
	uint32_t n_pairs = 0;
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &n_pairs, pair_off + n_query, 4));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	if (h_counts) h_counts[0] = n_pairs;
	if (n_pairs == 0) return MCOM_OK;
	// pass 2: evaluate pairs
	uint32_t *pass = nullptr, *pair_q = nullptr;
	const size_t scr2_b = (mcom_scan_scratch_elems(n_pairs) * 4 + 1024 + 255) & ~(size_t)255;
	hipError_t e = mcom_dmalloc(&pass, ((size_t)n_pairs + 1) * 4);
	if (e == hipSuccess) e = mcom_dmalloc(&pair_q, (size_t)n_pairs * 4);
	auto cleanup = [&]() { mcom_dfree_later(ctx, pass); mcom_dfree_later(ctx, pair_q); };
	if (e != hipSuccess) { cleanup(); return mcom_fail(ctx, MCOM_E_NOMEM, "candidate buffers: %s", hipGetErrorString(e)); }
	hipError_t e1 = hipSuccess;
	rc = mcom_ws_reserve(ctx, scr2_b);
	if (rc) { cleanup(); return rc; }
	const unsigned pb = (unsigned)(((size_t)n_pairs + 255) / 256);
	{ McomProfScope ps_(ctx, PROF_FIND_NEXT);
	MCOM_LAUNCH(k_fn_expand, dim3(qb), dim3(256), 0, ctx->stream, pair_off, n_query, pair_q);
	MCOM_LAUNCH(k_fn_eval, dim3(pb), dim3(256), 0, ctx->stream, first, mi->rec, d_query, (const uint64_t*)qy, pair_off, pair_q, n_pairs, d_cbits, d_coff, d_clen, cbthr, new_lo, new_span, pass); }
	uint32_t n_pass = 0;
	rc = mcom_scan_u32(ctx, pass, pass, (size_t)n_pairs + 1, (uint32_t*)ctx->ws);
	if (rc) { cleanup(); return rc; }
	hipError_t e1b = mcom_d2h_async(ctx, &n_pass, pass + n_pairs, 4);               // (the scan's own total: no copy, scan.hip)
	if (e1b == hipSuccess) e1b = mcom_stream_sync(ctx);
	if (e1b != hipSuccess) { cleanup(); return mcom_fail(ctx, MCOM_E_HIP, "candidate scan: %s", hipGetErrorString(e1b)); }
	if (h_counts) h_counts[1] = n_pass;
	if (n_pass > cap) { cleanup(); return mcom_fail(ctx, MCOM_E_OVERFLOW, "%u passing candidates but room for %zu", n_pass, cap); }
	if (n_pass) {
		if (!d_out) { cleanup(); return mcom_fail(ctx, MCOM_E_ARG, "null output pointer"); }
		MCOM_LAUNCH(k_fn_emit, dim3(pb), dim3(256), 0, ctx->stream, first, mi->rec, d_query, (const uint64_t*)qy, pair_off, pair_q, pass, n_pairs, d_out);
		e1 = hipGetLastError();                                                     // (d_out is complete when the stream gets there: no wait here)
		if (e1 != hipSuccess) { cleanup(); return mcom_fail(ctx, MCOM_E_HIP, "candidate emit: %s", hipGetErrorString(e1)); }
	}
	cleanup();
	return MCOM_OK;
}
extern "C" int mcom_find_next_candidates_new(mcom_ctx *ctx, const mcom_idx *mi, const mcom_mm128 *d_query, size_t n_query,
                                             const uint64_t *d_cbits, const uint64_t *d_coff, const uint32_t *d_clen, int cbthr, uint32_t n_new,
                                             mcom_mm128 *d_out, size_t cap, uint64_t *h_counts)
{
	return find_next_impl(ctx, mi, d_query, n_query, nullptr, nullptr, 0, d_cbits, d_coff, d_clen, cbthr, 0, n_new, n_new, d_out, cap, h_counts);
}
extern "C" int mcom_find_next_candidates_ord(mcom_ctx *ctx, const mcom_idx *mi, const mcom_mm128 *d_rec, const uint32_t *d_roff, const uint32_t *d_ord, size_t n_contigs,
                                             const uint64_t *d_cbits, const uint64_t *d_coff, const uint32_t *d_clen, int cbthr, uint32_t first_new, uint32_t n_new,
                                             mcom_mm128 *d_out, size_t cap, uint64_t *h_counts)
{
	if (!d_roff) return ctx ? mcom_fail(ctx, MCOM_E_ARG, "null device pointer") : MCOM_E_ARG;
	return find_next_impl(ctx, mi, d_rec, 0, d_roff, d_ord, n_contigs, d_cbits, d_coff, d_clen, cbthr, first_new, n_new ? 0xFFFFFFFFu : 0u, n_new, d_out, cap, h_counts);
}
