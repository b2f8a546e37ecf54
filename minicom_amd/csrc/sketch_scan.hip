// minicom_amd/csrc/sketch_scan.hip -- mm_sketch_lh_ori (reference sketch.c:116-165) with ONE LANE PER STRING.
//
// The wave-per-contig kernel (contigs.hip) spreads the positions of one contig over the lanes and pays for it with ballots,
// prefix sums, block minima and LDS rings shared by the wave: ~1100 wave instructions per 64 positions.  The strings sketched
// in a merge round are millions of short ones (the 7.8 M first contigs of ~200 bases, then the segments around the overlaps of
// merged contigs), so here every lane runs the reference's own sequential scan over its own string -- k-mer registers, run
// counter, the ring of the last w entries (in LDS, slot-major: a lane always hits its own banks), the current minimum -- and
// 64 strings advance one position per iteration: ~150 wave instructions per 64 positions.  The one loop of the reference that
// would ruin this is the rescan of the ring when the minimum leaves the window: every ~w positions per lane, i.e. in most
// iterations for SOME lane of the wave, and w entries long.  The scan's minimum is always the newest smallest entry of the ring
// (sketch.c:145-153), and the ring is, at any moment, the lap being written (slots 0..slot, newest) plus what is left of the
// lap before (slots slot+1..w-1): so the minimum is the better of the running minimum of this lap (registers) and the minimum
// of the suffix slot+1..w-1 of the last lap -- a table of w suffix minima made once per lap, when the slot wraps, by all lanes
// at once (van Herk / Gil-Werman).  A rescan becomes two LDS reads.
// Strings are handed out longest first in bins of similar length, so the lanes of a wave finish together.
//
// Output without a second scan: every string gets room for ~1.5x the expected number of minimizers in a temporary array,
// counts everything it would emit, the counts are scanned and the records gathered; a string that emitted more than its room
// is scanned once more, straight into its final place.
#include "mcom_dev.hpp"
#include <algorithm>
#include <type_traits>


namespace {
#define SSC_BINS 1024
#define SSC_STRIDE 64

__device__ __forceinline__ uint32_t ssc_len(const uint64_t *off, const uint64_t *off_end, size_t t)
{
	return (uint32_t)((off_end ? off_end[t] : off[t + 1]) - off[t]);
}

// room, length bin and the longest string.  A workgroup takes a run of 256-string chunks and adds its counts to the global bins
// ONCE (round 4: with a workgroup per chunk, 30 000 workgroups x ~15 occupied bins queued on four cache lines -- the kernel's whole
// 0.36 ms on 8 M strings; the same for the cursors of k_ssc_order below).
#define SSC_GRID 1024
__global__ __launch_bounds__(256) void k_ssc_prepare(const uint64_t *__restrict__ off, const uint64_t *__restrict__ off_end, uint32_t n, int w,
                                                     uint32_t *__restrict__ room, uint32_t *__restrict__ bins, uint32_t *__restrict__ longest, uint32_t *__restrict__ cnt)
{
	__shared__ uint32_t h[SSC_BINS];
	__shared__ uint32_t mx;
	for (int q = threadIdx.x; q < SSC_BINS; q += 256) h[q] = 0;
	if (threadIdx.x == 0) mx = 0;
	__syncthreads();
	const uint32_t chunks = n / 256u + 1u, per = (chunks + gridDim.x - 1) / gridDim.x;      // (n + 1 entries: room[n] = 0)
	const uint32_t c0 = blockIdx.x * per, c1 = c0 + per < chunks ? c0 + per : chunks;
	uint32_t lmx = 0;
	for (uint32_t c = c0; c < c1; ++c) {
		const uint32_t t = c * 256u + threadIdx.x;
		if (t < n) {
			const uint32_t len = ssc_len(off, off_end, t);
			room[t] = len ? 3u * len / (uint32_t)(w + 1) + 6u : 0u;
			atomicAdd(&h[len >> 2 < SSC_BINS ? len >> 2 : SSC_BINS - 1], 1u);
			lmx = len > lmx ? len : lmx;
		} else if (t == n) { room[t] = 0; cnt[t] = 0; }                     // (the scans over n + 1 entries find their last one cleared)
	}
	if (lmx) atomicMax(&mx, lmx);
	__syncthreads();
	for (int q = threadIdx.x; q < SSC_BINS; q += 256) if (h[q]) atomicAdd(&bins[q], h[q]);
	if (threadIdx.x == 0 && mx > *longest) atomicMax(longest, mx);
}
// start[b]: the strings in longer bins
__global__ void k_ssc_starts(const uint32_t *__restrict__ bins, uint32_t *__restrict__ start)
{
	if (threadIdx.x || blockIdx.x) return;
	uint32_t a = 0;
	for (int q = SSC_BINS - 1; q >= 0; --q) { start[q] = a; a += bins[q]; }
}
// perm: the strings, longest bin first (a workgroup counts its run of chunks, reserves its room in every bin once, then places)
__global__ __launch_bounds__(256) void k_ssc_order(const uint64_t *__restrict__ off, const uint64_t *__restrict__ off_end, uint32_t n,
                                                   const uint32_t *__restrict__ start, uint32_t *__restrict__ cursor, uint32_t *__restrict__ perm)
{
	__shared__ uint32_t h[SSC_BINS], base[SSC_BINS];
	for (int q = threadIdx.x; q < SSC_BINS; q += 256) h[q] = 0;
	__syncthreads();
	const uint32_t chunks = (n + 255u) / 256u, per = (chunks + gridDim.x - 1) / gridDim.x;
	const uint32_t c0 = blockIdx.x * per, c1 = c0 + per < chunks ? c0 + per : chunks;
	for (uint32_t c = c0; c < c1; ++c) {
		const uint32_t t = c * 256u + threadIdx.x;
		if (t < n) { const uint32_t len = ssc_len(off, off_end, t); atomicAdd(&h[len >> 2 < SSC_BINS ? len >> 2 : SSC_BINS - 1], 1u); }
	}
	__syncthreads();
	for (int q = threadIdx.x; q < SSC_BINS; q += 256) { base[q] = h[q] ? start[q] + atomicAdd(&cursor[q], h[q]) : 0u; h[q] = 0; }
	__syncthreads();
	for (uint32_t c = c0; c < c1; ++c) {
		const uint32_t t = c * 256u + threadIdx.x;
		if (t < n) {
			const uint32_t len = ssc_len(off, off_end, t), bin = len >> 2 < SSC_BINS ? len >> 2 : SSC_BINS - 1;
			perm[base[bin] + atomicAdd(&h[bin], 1u)] = t;
		}
	}
}

// lane q of workgroup b scans string list[64 b + q] (list = NULL: the string of that index); at most room[t] records go to
// dst[base[t] ...], all of them are counted in cnt[t] (capped at `limit`)
// ODDK: k is odd, so no k-mer equals its reverse complement and every position of the string stores exactly one entry: the
// position of a ring entry follows from its slot, its strand bit rides in bit 62 of the hash word (hashes have 2k <= 62 bits),
// and the ring of positions (a sixth of the LDS, i.e. one more wave per CU) is not needed.
#define SSC_ZBIT (1ull << 62)
// WIDE: k >= 17, the hash in its form for 64-bit k-mers (mcom_hash64_wide: high-word mask, shift-add chains)
template <bool ODDK, bool WIDE>
__global__ __launch_bounds__(64) void k_sketch_scan(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ off, const uint64_t *__restrict__ off_end,
                                                    const uint32_t *__restrict__ ids, const uint32_t *__restrict__ list, uint32_t nlist,
                                                    int w, int k, uint32_t limit, const uint32_t *__restrict__ base, const uint32_t *__restrict__ room,
                                                    int room_is_count, mcom_mm128 *__restrict__ dst, uint32_t *__restrict__ cnt)
{
	extern __shared__ __align__(8) unsigned char ssc_lds[];
	uint64_t *RX = (uint64_t*)ssc_lds;                                     // [w][SSC_STRIDE]: hash of the entry, U64MAX when empty
	uint16_t *RY = (uint16_t*)(RX + (size_t)w * SSC_STRIDE);               // [w][64]: pos<<1 | strand, 0xFFFF when empty (not ODDK)
	uint8_t *SM = ODDK ? (uint8_t*)RY : (uint8_t*)(RY + (size_t)w * 64);                         // [w][64]: newest smallest slot of the last lap's slots j..w-1; bit 7: its hash occurs again there
	const int lane = threadIdx.x;
	const uint32_t li = blockIdx.x * 64u + (uint32_t)lane;
	const bool have = li < nlist;
	const uint32_t t = have ? (list ? list[li] : li) : 0u;
	const uint32_t len = have ? ssc_len(off, off_end, t) : 0u;
	const uint8_t *s = seq + (have ? off[t] : 0);
	const uint64_t idhi = (uint64_t)(ids ? (have ? ids[t] : 0u) : (uint32_t)t) << 32;
	const uint32_t mybase = have ? base[t] : 0u;
	const uint32_t myroom = have ? (room_is_count ? cnt[t] : room[t]) : 0u;
	uint32_t maxlen = len;
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)maxlen, d, 64); maxlen = o > maxlen ? o : maxlen; }
	for (int j = 0; j < w; ++j) { RX[j * SSC_STRIDE + lane] = U64MAX; if (!ODDK) RY[j * 64 + lane] = 0xFFFFu; SM[j * 64 + lane] = (uint8_t)((w - 1) | 0x80); }
	const uint64_t mask = (1ull << (2 * k)) - 1;
	const int shift1 = 2 * (k - 1);
	uint64_t fwd = 0, rev = 0;
	int run = 0, slot = 0, best_slot = 0;
	uint64_t best_x = U64MAX; uint32_t best_y = 0xFFFFFFFFu;
	uint64_t lap_x = U64MAX; int lap_slot = 0; bool lap_dup = false;           // newest smallest entry of the lap being written
	uint32_t ne = 0;
	auto y32 = [](uint16_t v) -> uint32_t { return v == 0xFFFFu ? 0xFFFFFFFFu : (uint32_t)v; };
	uint32_t step = 0;                                                         // the current position (for ring_y)
	auto ring_x = [&](int j) -> uint64_t { const uint64_t v = RX[j * SSC_STRIDE + lane]; return ODDK ? (v == U64MAX ? v : v & ~SSC_ZBIT) : v; };
	// pos<<1 | strand of the entry in slot j (0xFFFFFFFF when empty); xw: the ring word of that slot
	auto ring_y = [&](int j, uint64_t xw) -> uint32_t {
		if (!ODDK) return y32(RY[j * 64 + lane]);
		if (xw == U64MAX) return 0xFFFFFFFFu;
		const int age = slot - j < 0 ? slot - j + w : slot - j;               // stored that many positions ago
		return ((step - (uint32_t)age) << 1) | (uint32_t)((xw >> 62) & 1u);
	};
	auto put = [&](uint64_t x, uint32_t y) {
		if (ne < limit) {
			if (ne < myroom) { mcom_mm128 v; v.x = x; v.y = (x == U64MAX && y == 0xFFFFFFFFu) ? U64MAX : (idhi | y); dst[(size_t)mybase + ne] = v; }
			++ne;
		}
	};
	auto put_if = [&](bool cond, uint64_t x, uint32_t y) {
		const bool counted = cond && ne < limit;
		if (counted && ne < myroom) { mcom_mm128 v; v.x = x; v.y = (x == U64MAX && y == 0xFFFFFFFFu) ? U64MAX : (idhi | y); dst[(size_t)mybase + ne] = v; }
		ne += counted ? 1u : 0u;
	};
	// every other entry of the ring, oldest first, that has the minimum's hash but is not the minimum itself (sketch.c:140-143, :156-160)
	auto put_equals = [&](bool with_current) {
		for (int j = slot + 1; j < w; ++j) { const uint64_t xw = RX[j * SSC_STRIDE + lane], x = ODDK && xw != U64MAX ? xw & ~SSC_ZBIT : xw; if (x == best_x) { const uint32_t y = ring_y(j, xw); if (y != best_y) put(x, y); } }
		const int e = with_current ? slot + 1 : slot;
		for (int j = 0; j < e; ++j) { const uint64_t xw = RX[j * SSC_STRIDE + lane], x = ODDK && xw != U64MAX ? xw & ~SSC_ZBIT : xw; if (x == best_x) { const uint32_t y = ring_y(j, xw); if (y != best_y) put(x, y); } }
	};
	// Eight characters from position p on: one unaligned 8-byte load whatever p is -- a string of at least eight characters is
	// read at min(p, len - 8) and shifted when the characters are taken, a shorter one (all of it in the first chunk, read byte by
	// byte up front) reads a harmless word -- so that no branch joins behind the load and it stays in flight until its characters
	// are needed, eight iterations later.
	const bool tiny = len < 8;
	auto issue8 = [&](uint32_t p, uint32_t &sh) -> uint64_t {
		const uint32_t a = tiny ? 0u : (p + 8 <= len ? p : len - 8);
		const uint8_t *src = tiny ? (const uint8_t*)off : s + a;
		uint64_t v; __builtin_memcpy(&v, src, 8);
		sh = tiny ? 64u : 8 * (p - a);
		return v;
	};
	uint64_t chunk = 0;
	uint32_t ahead_sh = 0;
	uint64_t ahead = issue8(0, ahead_sh);
	uint64_t first8 = 0;
	if (tiny) for (uint32_t q = 0; q < len; ++q) first8 |= (uint64_t)s[q] << (8 * q);
	// One position of the scan, in three strengths (k odd only for the upper two): 0 = as written; 1 = every lane of the wave is inside
	// its string and reads A, C, G or T for the whole chunk of eight, so the tests on that and the run counter's cases drop out; 2 = the
	// lanes are past their first full window as well (run >= w + k), so the thresholds on the run counter and the first-window rule
	// drop out too -- the steady state of most iterations.
	uint32_t lap_y = 0xFFFFFFFFu;                                              // pos<<1 | strand of the lap's minimum
	auto scan_step = [&](const uint32_t i, auto level_tag) {
		constexpr int LV = decltype(level_tag)::value;
		const uint32_t ch = (uint32_t)chunk & 0xFFu; chunk >>= 8;
		step = i;
		const uint32_t u = ch & 0xDFu;                                       // fold case
		const bool in = LV >= 1 || i < len;
		const bool base = LV >= 1 || (in && (u == 'A' || u == 'C' || u == 'G' || u == 'T'));
		const uint64_t c = ((ch >> 1) ^ (ch >> 2)) & 3u;                     // A0 C1 G2 T3
		// Straight-line code with selects: a wave that is alone on its SIMD (the rings fill the LDS) pays for every taken branch with
		// an instruction-fetch bubble, and this loop body used to hold nineteen of them.
		const uint64_t nf = (fwd << 2 | c) & mask, nr = (rev >> 2) | ((3ull ^ c) << shift1);
		fwd = base ? nf : fwd; rev = base ? nr : rev;
		const bool pal = LV == 0 && base && fwd == rev;                      // a k-mer equal to its reverse complement stores nothing (:133)
		const bool stored = in && !pal;                                      // an ambiguous base stores an empty entry and resets the run
		const bool fwd_lt = fwd < rev;
		const uint32_t z = fwd_lt ? 0u : 1u;
		run = LV >= 1 ? run + 1 : (base ? (pal ? run : run + 1) : (in ? 0 : run));
		const bool real = LV == 2 || (base && !pal && run >= k);
		const uint64_t hx = WIDE ? mcom_hash64_wide(fwd_lt ? fwd : rev, (uint32_t)(mask >> 32)) : mcom_hash64(fwd_lt ? fwd : rev, mask);
		const uint64_t cx = real ? hx : U64MAX;
		const uint32_t cy = real ? ((i << 1) | z) : 0xFFFFFFFFu;
		if (stored) { RX[slot * SSC_STRIDE + lane] = (ODDK && real) ? (cx | (z ? SSC_ZBIT : 0ull)) : cx; if (!ODDK) RY[slot * 64 + lane] = (uint16_t)cy; }
		{
			const bool lt = stored && (slot == 0 || cx < lap_x), eq = stored && !lt && cx == lap_x;
			lap_x = lt ? cx : lap_x; lap_slot = (lt || eq) ? slot : lap_slot; lap_dup = lt ? false : (eq ? true : lap_dup);
			lap_y = (lt || eq) ? cy : lap_y;
		}
		if (LV < 2) {
			const bool firstwin = stored && run == w + k - 1;
			if (__builtin_expect(__ballot(firstwin) != 0, 0)) { if (firstwin) put_equals(false); }   // first full window: earlier copies of the minimum (:139-144)
		}
		const bool newmin = stored && cx <= best_x;                          // '<=': the rightmost of equal hashes wins
		const bool left = stored && !newmin && slot == best_slot;            // the minimum has just left the window
		put_if(LV == 2 ? (newmin || left) : ((newmin && run >= w + k) || (left && run >= w + k - 1)), best_x, best_y);
		// the reference scans slot+1..w-1, then 0..slot, with '>=': the last smallest entry in that order = the better of the lap's
		// minimum and the minimum of what is left of the lap before
		uint64_t nx = lap_x; int ns = lap_slot; bool dup = lap_dup; uint32_t ny = lap_y;
		if (left && slot + 1 < w) {
			const uint32_t sm = SM[(slot + 1) * 64 + lane];
			const int sj = (int)(sm & 63u);
			const uint64_t sxw = RX[sj * SSC_STRIDE + lane];
			const uint64_t sx = ODDK ? (sxw == U64MAX ? sxw : sxw & ~SSC_ZBIT) : sxw;
			const bool older = sx < lap_x;
			dup = older ? (sm & 0x80u) != 0 : (dup || sx == lap_x);
			if (older) { nx = sx; ns = sj; ny = ring_y(sj, sxw); }
		}
		best_x = newmin ? cx : (left ? nx : best_x);
		best_y = newmin ? cy : (left ? ny : best_y);
		best_slot = newmin ? slot : (left ? ns : best_slot);
		const bool again = left && dup && (LV == 2 || run >= w + k - 1);
		if (__builtin_expect(__ballot(again) != 0, 0)) { if (again) put_equals(true); }           // identical k-mers of the new minimum (:155-161)
		const bool wrapped = stored && slot + 1 == w;
		slot = stored ? (wrapped ? 0 : slot + 1) : slot;
		if (__builtin_expect(__ballot(wrapped) != 0, 0)) {
			if (wrapped) {                                                   // the lap is complete: its suffix minima, newest (highest slot) first among equals
				uint64_t mx = ring_x(w - 1); uint32_t ms = (uint32_t)(w - 1);
				SM[(w - 1) * 64 + lane] = (uint8_t)ms;
				for (int j = w - 2; j >= 0; j -= 4) {                           // four reads in flight: the wave has nobody to hide its LDS latency behind
					uint64_t xs[4];
#pragma unroll
					for (int q = 0; q < 4; ++q) xs[q] = j - q >= 0 ? ring_x(j - q) : U64MAX;
#pragma unroll
					for (int q = 0; q < 4; ++q) if (j - q >= 0) {
						if (xs[q] < mx) { mx = xs[q]; ms = (uint32_t)(j - q); } else if (xs[q] == mx) ms |= 0x80u;
						SM[(j - q) * 64 + lane] = (uint8_t)ms;
					}
				}
			}
		}
	};
	// 0x80 in every byte of x that equals the byte of pat
	auto eq8 = [](uint64_t x, uint64_t pat) -> uint64_t { const uint64_t zz = x ^ pat; return ~(((zz & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | zz | 0x7F7F7F7F7F7F7F7Full); };
	for (uint32_t i0 = 0; i0 < maxlen; i0 += 8) {
		chunk = ahead_sh < 64 ? ahead >> ahead_sh : (i0 == 0 ? first8 : 0ull);   // the next eight characters travel while these are scanned
		ahead = issue8(i0 + 8, ahead_sh);
		bool clean = false;
		if (ODDK) {
			const uint64_t uc = chunk & 0xDFDFDFDFDFDFDFDFull;
			const uint64_t ok = eq8(uc, 0x4141414141414141ull) | eq8(uc, 0x4343434343434343ull) | eq8(uc, 0x4747474747474747ull) | eq8(uc, 0x5454545454545454ull);
			clean = i0 + 8 <= len && ok == 0x8080808080808080ull;
		}
		if (ODDK && __all(clean)) {
			if (__all(run >= w + k)) {
#pragma unroll 1
				for (uint32_t q = 0; q < 8; ++q) scan_step(i0 + q, std::integral_constant<int, 2>{});
			} else {
#pragma unroll 1
				for (uint32_t q = 0; q < 8; ++q) scan_step(i0 + q, std::integral_constant<int, 1>{});
			}
		} else {
#pragma unroll 1
			for (uint32_t q = 0; q < 8 && i0 + q < maxlen; ++q) scan_step(i0 + q, std::integral_constant<int, 0>{});
		}
	}
	if (best_x != U64MAX) put(best_x, best_y);                               // the minimum still held (:163-164)
	if (have) cnt[t] = ne;
}

// ---- the same scan with a ring of 32-bit hash PREFIXES (k odd) ---------------------------------------------------------------
// The ring of 8-byte hashes is what fills the LDS: w x 64 x 8 bytes per wave, six waves per CU, 1.5 per SIMD -- and a wave that
// is nearly alone on its SIMD pays every latency in full (PMC: 35 % of its cycles waiting, VALU pipe a quarter busy).  What the
// scan does with a ring entry is compare it: with the minimum of the lap, with another entry when the suffix minima are made,
// with the minimum when equal k-mers are looked for.  The top PB bits of the hash (and the strand bit) decide nearly every
// comparison; when two prefixes are equal the hashes are RECOMPUTED from the strings (the position of a ring entry follows from
// its slot, the k characters ending there give the k-mer: hash64 is a bijection, equal hashes are equal k-mers), so every
// decision is the one the 64-bit ring makes.  Ties are one comparison in 2^30 between different k-mers, and every comparison
// between equal ones (repeats inside a window: rare, and costing a hundred instructions each).  The minimum itself is kept as
// (prefix, position) too, so what is emitted is the position: the hash of an emitted record is made afterwards from the string
// (in k_ssc_gather) -- 0.07 records per position.  Default: 14-bit prefixes in 16-bit words, a quarter of the LDS (19 waves per CU
// by LDS, 6 per SIMD by registers); measured on 8 M strings of 200: 64-bit ring 11.8 ms, 30-bit prefixes 9.3, 14-bit 8.0, 12-bit 9.4.
// RT: the ring word, uint32_t (prefixes of up to 30 bits) or uint16_t (up to 14: a tie every 16 384 comparisons, still next to nothing,
// for a ring of a quarter of the 64-bit one's size)
template <bool WIDE, class RT>
__global__ __launch_bounds__(64) void k_sketch_scan32(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ off, const uint64_t *__restrict__ off_end,
                                                      const uint32_t *__restrict__ ids, const uint32_t *__restrict__ list, uint32_t nlist,
                                                      int w, int k, uint32_t limit, const uint32_t *__restrict__ base, const uint32_t *__restrict__ room,
                                                      int room_is_count, mcom_mm128 *__restrict__ dst, uint32_t *__restrict__ cnt, int pb)
{
	extern __shared__ __align__(8) unsigned char ssc_lds[];
	RT *RP = (RT*)ssc_lds;                                                 // [w][64]: prefix << 1 | strand, all ones when empty
	constexpr uint32_t EW = (uint32_t)(RT)~(RT)0, SSC_EMPTY32 = EW >> 1;     // the empty ring word and its key (above every prefix)
	uint8_t *SM = (uint8_t*)(RP + (size_t)w * 64);                         // [w][64]: newest smallest slot of the last lap's slots j..w-1; bit 7: its hash occurs again there
	const int lane = threadIdx.x;
	const uint32_t li = blockIdx.x * 64u + (uint32_t)lane;
	const bool have = li < nlist;
	const uint32_t t = have ? (list ? list[li] : li) : 0u;
	const uint32_t len = have ? ssc_len(off, off_end, t) : 0u;
	const uint8_t *s = seq + (have ? off[t] : 0);
	const uint64_t idhi = (uint64_t)(ids ? (have ? ids[t] : 0u) : (uint32_t)t) << 32;
	const uint32_t mybase = have ? base[t] : 0u;
	const uint32_t myroom = have ? (room_is_count ? cnt[t] : room[t]) : 0u;
	uint32_t maxlen = len;
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)maxlen, d, 64); maxlen = o > maxlen ? o : maxlen; }
	for (int j = 0; j < w; ++j) { RP[j * 64 + lane] = (RT)EW; SM[j * 64 + lane] = (uint8_t)((w - 1) | 0x80); }
	const uint64_t mask = (1ull << (2 * k)) - 1;
	const int shift1 = 2 * (k - 1);
	const int psh = 2 * k > pb ? 2 * k - pb : 0;                           // the prefix: the top pb bits of the 2k-bit hash
	uint64_t fwd = 0, rev = 0;
	int run = 0, slot = 0, best_slot = 0;
	uint32_t best_p = SSC_EMPTY32, best_y = 0xFFFFFFFFu;
	uint32_t lap_p = SSC_EMPTY32; int lap_slot = 0; bool lap_dup = false; uint32_t lap_y = 0xFFFFFFFFu;
	uint32_t ne = 0, step = 0;
	// the hash of the k-mer that ends at position pos (all of its k characters are bases: only such entries are real)
	auto full_at = [&](uint32_t pos) -> uint64_t {
		uint64_t f = 0, r = 0;
		for (int q = k - 1; q >= 0; --q) {
			const uint32_t ch = s[pos - (uint32_t)q];
			const uint64_t c = ((ch >> 1) ^ (ch >> 2)) & 3u;
			f = (f << 2 | c) & mask; r = (r >> 2) | ((3ull ^ c) << shift1);
		}
		const uint64_t canon = f < r ? f : r;
		return WIDE ? mcom_hash64_wide(canon, (uint32_t)(mask >> 32)) : mcom_hash64(canon, mask);
	};
	// position of the entry in slot j, seen at a storing step whose entry went into slot `cur`
	auto pos_of = [&](int j, int cur) -> uint32_t { const int age = cur - j < 0 ? cur - j + w : cur - j; return step - (uint32_t)age; };
	// exact order of two entries given as (prefix, pos<<1|strand): -1 / 0 / +1 as the first hash is smaller / equal / larger
	auto cmp_entries = [&](uint32_t pa, uint32_t ya, uint32_t pb_, uint32_t yb) -> int {
		if (pa != pb_) return pa < pb_ ? -1 : 1;
		if (pa == SSC_EMPTY32) return 0;
		const uint64_t xa = full_at(ya >> 1), xb = full_at(yb >> 1);
		return xa < xb ? -1 : (xa == xb ? 0 : 1);
	};
	auto put_y = [&](uint32_t y) {
		if (ne < limit) {
			if (ne < myroom) { mcom_mm128 v; v.x = y == 0xFFFFFFFFu ? U64MAX : 0ull; v.y = y == 0xFFFFFFFFu ? U64MAX : (idhi | y); dst[(size_t)mybase + ne] = v; }
			++ne;
		}
	};
	auto put_if = [&](bool cond, uint32_t y) {
		const bool counted = cond && ne < limit;
		if (counted && ne < myroom) { mcom_mm128 v; v.x = y == 0xFFFFFFFFu ? U64MAX : 0ull; v.y = y == 0xFFFFFFFFu ? U64MAX : (idhi | y); dst[(size_t)mybase + ne] = v; }
		ne += counted ? 1u : 0u;
	};
	// every other entry of the ring, oldest first, that has the minimum's hash but is not the minimum itself (sketch.c:140-143, :156-160);
	// cur: the slot of the newest entry
	auto put_equals = [&](bool with_current) {
		if (best_p == SSC_EMPTY32) return;                                   // (empty entries carry y = all ones, as the minimum then does: never put)
		const uint64_t bx = full_at(best_y >> 1);
		auto one = [&](int j) {
			const uint32_t e = RP[j * 64 + lane];
			if ((e >> 1) != best_p || e == EW) return;
			const uint32_t y = (pos_of(j, slot) << 1) | (e & 1u);
			if (y != best_y && full_at(y >> 1) == bx) put_y(y);
		};
		for (int j = slot + 1; j < w; ++j) one(j);
		const int e = with_current ? slot + 1 : slot;
		for (int j = 0; j < e; ++j) one(j);
	};
	const bool tiny = len < 8;
	auto issue8 = [&](uint32_t p, uint32_t &sh) -> uint64_t {
		const uint32_t a = tiny ? 0u : (p + 8 <= len ? p : len - 8);
		const uint8_t *src = tiny ? (const uint8_t*)off : s + a;
		uint64_t v; __builtin_memcpy(&v, src, 8);
		sh = tiny ? 64u : 8 * (p - a);
		return v;
	};
	uint64_t chunk = 0;
	uint32_t ahead_sh = 0;
	uint64_t ahead = issue8(0, ahead_sh);
	uint64_t first8 = 0;
	if (tiny) for (uint32_t q = 0; q < len; ++q) first8 |= (uint64_t)s[q] << (8 * q);
	auto scan_step = [&](const uint32_t i, auto level_tag) {
		constexpr int LV = decltype(level_tag)::value;
		const uint32_t ch = (uint32_t)chunk & 0xFFu; chunk >>= 8;
		step = i;
		const uint32_t u = ch & 0xDFu;
		const bool in = LV >= 1 || i < len;
		const bool base_ = LV >= 1 || (in && (u == 'A' || u == 'C' || u == 'G' || u == 'T'));
		const uint64_t c = ((ch >> 1) ^ (ch >> 2)) & 3u;
		const uint64_t nf = (fwd << 2 | c) & mask, nr = (rev >> 2) | ((3ull ^ c) << shift1);
		fwd = base_ ? nf : fwd; rev = base_ ? nr : rev;
		const bool stored = in;                                              // (k odd: no k-mer is its own reverse complement)
		const bool fwd_lt = fwd < rev;
		const uint32_t z = fwd_lt ? 0u : 1u;
		run = LV >= 1 ? run + 1 : (base_ ? run + 1 : (in ? 0 : run));
		const bool real = LV == 2 || (base_ && run >= k);
		const uint64_t hx = WIDE ? mcom_hash64_wide(fwd_lt ? fwd : rev, (uint32_t)(mask >> 32)) : mcom_hash64(fwd_lt ? fwd : rev, mask);
		const uint32_t cp = real ? (uint32_t)(hx >> psh) : SSC_EMPTY32;
		const uint32_t cy = real ? ((i << 1) | z) : 0xFFFFFFFFu;
		if (stored) RP[slot * 64 + lane] = (RT)(real ? ((cp << 1) | z) : EW);
		{
			// the lap's minimum: '<' replaces, '==' marks a duplicate and moves to the newer entry
			bool lt = stored && (slot == 0 || cp < lap_p), eq = false;
			const bool tie = stored && slot != 0 && cp == lap_p;
			if (__builtin_expect(__ballot(tie) != 0, 0)) {
				if (tie) { if (cp == SSC_EMPTY32) eq = true; else { const uint64_t lx = full_at(lap_y >> 1); lt = hx < lx; eq = hx == lx; } }
			}
			lap_p = lt ? cp : lap_p; lap_slot = (lt || eq) ? slot : lap_slot; lap_dup = lt ? false : (eq ? true : lap_dup);
			lap_y = (lt || eq) ? cy : lap_y;
		}
		if (LV < 2) {
			const bool firstwin = stored && run == w + k - 1;
			if (__builtin_expect(__ballot(firstwin) != 0, 0)) { if (firstwin) put_equals(false); }
		}
		bool newmin = stored && cp < best_p;                                  // '<=': the rightmost of equal hashes wins
		{
			const bool tie = stored && cp == best_p;
			if (__builtin_expect(__ballot(tie) != 0, 0)) {
				if (tie) newmin = cp == SSC_EMPTY32 ? true : hx <= full_at(best_y >> 1);
			}
		}
		const bool left = stored && !newmin && slot == best_slot;            // the minimum has just left the window
		put_if(LV == 2 ? (newmin || left) : ((newmin && run >= w + k) || (left && run >= w + k - 1)), best_y);
		uint32_t np = lap_p; int ns = lap_slot; bool dup = lap_dup; uint32_t ny = lap_y;
		if (left && slot + 1 < w) {
			const uint32_t sm = SM[(slot + 1) * 64 + lane];
			const int sj = (int)(sm & 63u);
			const uint32_t e = RP[sj * 64 + lane];
			const uint32_t sp = e >> 1;                                        // (the empty word's key is SSC_EMPTY32)
			const uint32_t sy = e == EW ? 0xFFFFFFFFu : ((pos_of(sj, slot) << 1) | (e & 1u));
			const int c3 = cmp_entries(sp, sy, lap_p, lap_y);
			const bool older = c3 < 0;
			dup = older ? (sm & 0x80u) != 0 : (dup || c3 == 0);
			if (older) { np = sp; ns = sj; ny = sy; }
		}
		best_p = newmin ? cp : (left ? np : best_p);
		best_y = newmin ? cy : (left ? ny : best_y);
		best_slot = newmin ? slot : (left ? ns : best_slot);
		const bool again = left && dup && (LV == 2 || run >= w + k - 1);
		if (__builtin_expect(__ballot(again) != 0, 0)) { if (again) put_equals(true); }
		const bool wrapped = stored && slot + 1 == w;
		const int cur = slot;                                                // the slot this step's entry went into
		slot = stored ? (wrapped ? 0 : slot + 1) : slot;
		if (__builtin_expect(__ballot(wrapped) != 0, 0)) {
			if (wrapped) {                                                   // the lap is complete: its suffix minima, newest (highest slot) first among equals
				auto key = [&](int j) -> uint32_t { return (uint32_t)RP[j * 64 + lane] >> 1; };
				auto yof = [&](int j) -> uint32_t { const uint32_t e = RP[j * 64 + lane]; return e == EW ? 0xFFFFFFFFu : ((pos_of(j, cur) << 1) | (e & 1u)); };
				uint32_t mx = key(w - 1); uint32_t ms = (uint32_t)(w - 1);
				SM[(w - 1) * 64 + lane] = (uint8_t)ms;
				for (int j = w - 2; j >= 0; j -= 4) {
					uint32_t xs[4];
#pragma unroll
					for (int q = 0; q < 4; ++q) xs[q] = j - q >= 0 ? key(j - q) : SSC_EMPTY32;
#pragma unroll
					for (int q = 0; q < 4; ++q) if (j - q >= 0) {
						if (xs[q] < mx) { mx = xs[q]; ms = (uint32_t)(j - q); }
						else if (xs[q] == mx) {
							const int c3 = xs[q] == SSC_EMPTY32 ? 0 : cmp_entries(xs[q], yof(j - q), mx, yof((int)(ms & 63u)));
							if (c3 < 0) ms = (uint32_t)(j - q); else if (c3 == 0) ms |= 0x80u;
						}
						SM[(j - q) * 64 + lane] = (uint8_t)ms;
					}
				}
			}
		}
	};
	auto eq8 = [](uint64_t x, uint64_t pat) -> uint64_t { const uint64_t zz = x ^ pat; return ~(((zz & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | zz | 0x7F7F7F7F7F7F7F7Full); };
	for (uint32_t i0 = 0; i0 < maxlen; i0 += 8) {
		chunk = ahead_sh < 64 ? ahead >> ahead_sh : (i0 == 0 ? first8 : 0ull);
		ahead = issue8(i0 + 8, ahead_sh);
		const uint64_t uc = chunk & 0xDFDFDFDFDFDFDFDFull;
		const uint64_t ok = eq8(uc, 0x4141414141414141ull) | eq8(uc, 0x4343434343434343ull) | eq8(uc, 0x4747474747474747ull) | eq8(uc, 0x5454545454545454ull);
		const bool clean = i0 + 8 <= len && ok == 0x8080808080808080ull;
		if (__all(clean)) {
			if (__all(run >= w + k)) {
#pragma unroll 1
				for (uint32_t q = 0; q < 8; ++q) scan_step(i0 + q, std::integral_constant<int, 2>{});
			} else {
#pragma unroll 1
				for (uint32_t q = 0; q < 8; ++q) scan_step(i0 + q, std::integral_constant<int, 1>{});
			}
		} else {
#pragma unroll 1
			for (uint32_t q = 0; q < 8 && i0 + q < maxlen; ++q) scan_step(i0 + q, std::integral_constant<int, 0>{});
		}
	}
	if (best_p != SSC_EMPTY32) put_y(best_y);                                // the minimum still held (:163-164)
	if (have) cnt[t] = ne;
}

// the hashes of the records k_sketch_scan32 emitted (x = 0 placeholders): from the string, four lanes per string.  The k characters
// that end at the record's position come as four unaligned 8-byte loads; eight characters pack into sixteen bits with three
// shift-or-mask steps, the forward k-mer is the group-reversed little-endian word and the reverse complement its complement.
__device__ __forceinline__ uint64_t ssc_pack8(uint64_t x)
{
	uint64_t c = ((x >> 1) ^ (x >> 2)) & 0x0303030303030303ull;          // A0 C1 G2 T3 in the low two bits of every byte
	c = (c | (c >> 6)) & 0x000F000F000F000Full;
	c = (c | (c >> 12)) & 0x000000FF000000FFull;
	return (c | (c >> 24)) & 0xFFFFull;                                  // character j of the eight at bits 2j
}
template <bool WIDE>
__device__ __forceinline__ uint64_t ssc_hash_at(const uint8_t *__restrict__ s, uint32_t pos, int k, uint64_t mask)
{
		// the 32 characters that end at pos (k <= 31 of them count; a string shorter than 32 is read from its start and shifted)
		uint64_t le;
		if (pos >= 31) {
			uint64_t w4[4];
			__builtin_memcpy(w4, s + (pos - 31), 32);
			le = ssc_pack8(w4[0]) | (ssc_pack8(w4[1]) << 16) | (ssc_pack8(w4[2]) << 32) | (ssc_pack8(w4[3]) << 48);   // character pos - 31 + j at bits 2j
			le >>= 2 * (32 - k);                                          // character pos - k + 1 + j at bits 2j
		} else {
			le = 0;
			for (int j = 0; j < k; ++j) { const uint32_t ch = s[pos + 1 - (uint32_t)k + (uint32_t)j]; le |= (uint64_t)(((ch >> 1) ^ (ch >> 2)) & 3u) << (2 * j); }
		}
		le &= mask;
		const uint64_t r = ~le & mask;                                   // reverse complement as the scan rolls it: the newest character on top, complemented
		uint64_t f = __brevll(le);                                       // forward k-mer: the oldest character on top
		f = ((f >> 1) & 0x5555555555555555ull) | ((f & 0x5555555555555555ull) << 1);
		f >>= 64 - 2 * k;
		const uint64_t canon = f < r ? f : r;
		return WIDE ? mcom_hash64_wide(canon, (uint32_t)(mask >> 32)) : mcom_hash64(canon, mask);
}
template <bool WIDE>
__global__ __launch_bounds__(256) void k_ssc_fill_x(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ off, const uint32_t *__restrict__ list, uint32_t nlist,
                                                    const uint32_t *__restrict__ moff, int k, mcom_mm128 *__restrict__ out)
{
	const uint32_t g = blockIdx.x * 256u + threadIdx.x;
	const uint32_t li = g >> 2, q = g & 3u;
	if (li >= nlist) return;
	const uint32_t t = list ? list[li] : li;
	const uint8_t *s = seq + off[t];
	const uint64_t mask = (1ull << (2 * k)) - 1;
	for (uint32_t i = moff[t] + q; i < moff[t + 1]; i += 4) {
		const uint64_t y = out[i].y;
		if (y != U64MAX) out[i].x = ssc_hash_at<WIDE>(s, (uint32_t)y >> 1, k, mask);
	}
}

// the records of every string from its room to its place; strings that emitted more than their room are listed
// FILL: the records come from k_sketch_scan32 with their hashes still to be made (1: k < 17, 2: k >= 17)
template <int FILL>
__global__ __launch_bounds__(256) void k_ssc_gather(const mcom_mm128 *__restrict__ tmp, const uint32_t *__restrict__ base, const uint32_t *__restrict__ room,
                                                    const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ moff, uint32_t n,
                                                    mcom_mm128 *__restrict__ out, uint32_t *__restrict__ over_list, uint32_t *__restrict__ n_over,
                                                    const uint8_t *__restrict__ seq, const uint64_t *__restrict__ off, int k)
{
	// four lanes per string
	const uint32_t g = blockIdx.x * 256u + threadIdx.x;
	const uint32_t t = g >> 2, q = g & 3u;
	if (t >= n) return;
	const uint32_t c = cnt[t], r = room[t];
	if (c > r) { if (q == 0) over_list[atomicAdd(n_over, 1u)] = t; return; }
	const mcom_mm128 *src = tmp + base[t];
	mcom_mm128 *dstp = out + moff[t];
	if (FILL == 0) { for (uint32_t i = q; i < c; i += 4) dstp[i] = src[i]; return; }
	const uint8_t *s = seq + off[t];
	const uint64_t mask = (1ull << (2 * k)) - 1;
	for (uint32_t i = q; i < c; i += 4) {
		mcom_mm128 v = src[i];
		if (v.y != U64MAX) v.x = ssc_hash_at<FILL == 2>(s, (uint32_t)v.y >> 1, k, mask);
		dstp[i] = v;
	}
}
}  // namespace

// -1: not applicable (window above 64 entries or a string of 32768 characters or more: the caller uses the wave-per-string kernel)
int mcom_sketch_strings_scan(mcom_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_off, const uint64_t *d_off_end, uint64_t chars,
                             const uint32_t *d_ids, size_t n, int w, int k, uint32_t limit, uint32_t *d_moff, mcom_mm128 *d_out, size_t cap,
                             uint64_t *h_total)
{
	if (w > 64 || n >= (1ull << 31)) return -1;
	auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
	const uint32_t nn = (uint32_t)n;
	const size_t n4 = al((n + 1) * 4);
	const size_t scr_b = al(mcom_scan_scratch_elems(n + 1) * 4 + 1024);
	const size_t head = 5 * n4 + scr_b + al((3 * SSC_BINS + 16) * 4);
	const uint64_t tmp_bound = 3 * chars / (uint64_t)(w + 1) + 6 * (uint64_t)n + 64;  // the rooms of all strings
	if (tmp_bound >= (1ull << 32)) return -1;
	int rc = mcom_ws_reserve(ctx, head + al(tmp_bound * sizeof(mcom_mm128)));
	if (rc) return rc;
	char *b0 = (char*)ctx->ws;
	uint32_t *room = (uint32_t*)b0, *base = (uint32_t*)(b0 + n4), *perm = (uint32_t*)(b0 + 2 * n4), *cnt = (uint32_t*)(b0 + 3 * n4), *over = (uint32_t*)(b0 + 4 * n4);
	uint32_t *scr = (uint32_t*)(b0 + 5 * n4);
	uint32_t *bins = (uint32_t*)(b0 + 5 * n4 + scr_b), *cursor = bins + SSC_BINS, *start = cursor + SSC_BINS, *misc = start + SSC_BINS;   // misc[0] longest, [1] overflowed strings
	mcom_mm128 *tmp = (mcom_mm128*)(b0 + head);
	MCOM_HIP(ctx, hipMemsetAsync(bins, 0, (3 * SSC_BINS + 16) * 4, ctx->stream));
	MCOM_LAUNCH(k_ssc_prepare, dim3(std::min<uint32_t>((nn + 1 + 255) / 256, SSC_GRID)), dim3(256), 0, ctx->stream, d_off, d_off_end, nn, w, room, bins, misc, cnt);
	MCOM_LAUNCH_CHECK(ctx);
	if ((rc = mcom_scan_u32(ctx, room, base, n + 1, scr))) return rc;
	MCOM_LAUNCH(k_ssc_starts, dim3(1), dim3(64), 0, ctx->stream, bins, start);
	MCOM_LAUNCH(k_ssc_order, dim3(std::min<uint32_t>((nn + 255) / 256, SSC_GRID)), dim3(256), 0, ctx->stream, d_off, d_off_end, nn, start, cursor, perm);
	uint32_t h2[2] = {0, 0};
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &h2[0], misc, 4));
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &h2[1], base + n, 4));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	if (h2[0] >= 32768u) return -1;
	if (h2[1] > tmp_bound) return mcom_fail(ctx, MCOM_E_HIP, "sketch rooms %u above their bound", h2[1]);
	const bool oddk = (k & 1) != 0, wide = k >= 17;
	const bool ring32 = oddk && !ctx->sketch_ring64;                         // the ring of hash prefixes: a half or a quarter of the LDS
	const bool ring16 = ring32 && ctx->sketch_prefix_bits <= 14 && !ctx->sketch_ring32_only;             // 16-bit ring words hold prefixes of up to 14 bits
	const int pb = ctx->sketch_prefix_bits;
#define SSC_LAUNCH(blocks, ...) do { \
	if (ring16 && wide) MCOM_LAUNCH((k_sketch_scan32<true, uint16_t>), dim3(blocks), dim3(64), lds, ctx->stream, __VA_ARGS__, pb); \
	else if (ring16) MCOM_LAUNCH((k_sketch_scan32<false, uint16_t>), dim3(blocks), dim3(64), lds, ctx->stream, __VA_ARGS__, pb); \
	else if (ring32 && wide) MCOM_LAUNCH((k_sketch_scan32<true, uint32_t>), dim3(blocks), dim3(64), lds, ctx->stream, __VA_ARGS__, pb); \
	else if (ring32) MCOM_LAUNCH((k_sketch_scan32<false, uint32_t>), dim3(blocks), dim3(64), lds, ctx->stream, __VA_ARGS__, pb); \
	else if (oddk && wide) MCOM_LAUNCH((k_sketch_scan<true, true>), dim3(blocks), dim3(64), lds, ctx->stream, __VA_ARGS__); \
	else if (oddk) MCOM_LAUNCH((k_sketch_scan<true, false>), dim3(blocks), dim3(64), lds, ctx->stream, __VA_ARGS__); \
	else if (wide) MCOM_LAUNCH((k_sketch_scan<false, true>), dim3(blocks), dim3(64), lds, ctx->stream, __VA_ARGS__); \
	else MCOM_LAUNCH((k_sketch_scan<false, false>), dim3(blocks), dim3(64), lds, ctx->stream, __VA_ARGS__); } while (0)
	const size_t lds = ring16 ? (size_t)w * 64 * 2 + (size_t)w * 64 : ring32 ? (size_t)w * 64 * 4 + (size_t)w * 64 : (size_t)w * SSC_STRIDE * 8 + (oddk ? 0 : (size_t)w * 64 * 2) + (size_t)w * 64;
	{
		McomProfScope ps_(ctx, PROF_SKETCH_CONTIGS);
		SSC_LAUNCH((nn + 63) / 64, d_seq, d_off, d_off_end, d_ids, perm, nn, w, k, limit, base, room, 0, tmp, cnt);
	}
	MCOM_LAUNCH_CHECK(ctx);
	ctx->sketch_strings += n;
	if ((rc = mcom_scan_u32(ctx, cnt, d_moff, n + 1, scr))) return rc;                // (cnt[n] = 0: k_ssc_prepare)
	uint32_t total = 0;
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &total, d_moff + n, 4));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	if (h_total) *h_total = total;
	if (total > cap) return mcom_fail(ctx, MCOM_E_OVERFLOW, "%u minimizers but room for %zu", total, cap);
	if (total == 0) return MCOM_OK;
	{
		const dim3 gb((unsigned)(((size_t)nn * 4 + 255) / 256));
		if (!ring32) MCOM_LAUNCH(k_ssc_gather<0>, gb, dim3(256), 0, ctx->stream, tmp, base, room, cnt, d_moff, nn, d_out, over, misc + 1, d_seq, d_off, k);
		else if (wide) MCOM_LAUNCH(k_ssc_gather<2>, gb, dim3(256), 0, ctx->stream, tmp, base, room, cnt, d_moff, nn, d_out, over, misc + 1, d_seq, d_off, k);
		else MCOM_LAUNCH(k_ssc_gather<1>, gb, dim3(256), 0, ctx->stream, tmp, base, room, cnt, d_moff, nn, d_out, over, misc + 1, d_seq, d_off, k);
	}
	MCOM_LAUNCH_CHECK(ctx);
	uint32_t n_over = 0;
	MCOM_HIP(ctx, mcom_d2h_async(ctx, &n_over, misc + 1, 4));
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	if (n_over) {                                                            // denser than their room: once more, into their final places
		McomProfScope ps_(ctx, PROF_SKETCH_CONTIGS);
		SSC_LAUNCH((n_over + 63) / 64, d_seq, d_off, d_off_end, d_ids, over, n_over, w, k, limit, d_moff, room, 1, d_out, cnt);
		MCOM_LAUNCH_CHECK(ctx);
		if (ring32) {                                                        // (the gather made the hashes of the others; these went straight into place)
			const unsigned fb = (unsigned)(((size_t)n_over * 4 + 255) / 256);
			if (wide) MCOM_LAUNCH(k_ssc_fill_x<true>, dim3(fb), dim3(256), 0, ctx->stream, d_seq, d_off, over, n_over, d_moff, k, d_out);
			else MCOM_LAUNCH(k_ssc_fill_x<false>, dim3(fb), dim3(256), 0, ctx->stream, d_seq, d_off, over, n_over, d_moff, k, d_out);
			MCOM_LAUNCH_CHECK(ctx);
		}
		MCOM_HIP(ctx, mcom_stream_sync(ctx));
	}
	return MCOM_OK;
}
