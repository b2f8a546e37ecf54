// minicom_amd/csrc/mcom_dev.hpp -- shared host/device helpers for libmcom_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <map>
#include <string>
#include <vector>
#include "../../include/mcom.h"

// kernel classes timed by the optional in-library profiler (mcom_prof_*): HIP events on the launch stream
enum McomProfId { PROF_CLASSIFY_PACK = 0, PROF_SKETCH_READS, PROF_RADIX_PASS, PROF_SKETCH_CONTIGS, PROF_FIND_NEXT,
                  PROF_DICT_BUILD, PROF_REALIGN_WINDOWS, PROF_CONSENSUS, PROF_CINDEX_BUILD, PROF_REALIGN_READS, PROF_COUNT };
struct McomProfSpan { int id; hipEvent_t a, b; };

struct mcom_ctx {
	int device;
	hipStream_t stream;
	std::string err;
	// grow-only device workspace
	void *ws; size_t ws_bytes;
	int n_cu;
	// profiler
	bool prof_on = false;
	std::vector<McomProfSpan> prof_open;
	double prof_ms[PROF_COUNT] = {0};
	uint64_t prof_calls[PROF_COUNT] = {0};
	// which kernels were launched, how often, and inside which timed class (PROF_COUNT = outside any): keyed by the host address of
	// the instantiation (MCOM_LAUNCH), so that a caller can ask for the EXACT kernels a class's time belongs to
	int prof_cur = PROF_COUNT;
	std::map<std::pair<int, const void *>, uint64_t> prof_kernels;
	// bucket sort: capacity of an in-LDS segment (0 = the kernel's own 4096; tests lower it to reach the fallback on small
	// inputs) and how many segments went through the fallback so far
	uint32_t seg_cap = 0; uint64_t sort_overflow_segments = 0;
	// strings sketched by the lane-per-string kernel so far (64 per wave: bench.py's issue roofline)
	uint64_t sketch_strings = 0;
	// contig index: entries of a partition above which the scattered placement is used (tests lower it; 0 entries = always)
	bool cix_cap_set = false; uint32_t cix_cap = 0;
	// merge consensus: members reaching one unit of 32 columns above which its tile goes to the wave-per-tile kernel (0 = the counters' 127)
	uint32_t bs_cap = 0;
	// contig sketch: true = always the wave-per-string kernel (tests compare the two)
	bool sketch_wave_only = false;
	bool sketch_lane_always = false;                        // (test hook: the lane-per-string kernel for a handful of strings too)
	// ... 2 = the lane-per-string kernel with the ring of 64-bit hashes even where the ring of 32-bit prefixes applies (k odd); and the
	// prefix width of the latter (tests narrow it to a few bits so that prefix ties, one in 2^30 otherwise, happen all the time)
	bool sketch_ring64 = false, sketch_ring32_only = false; int sketch_prefix_bits = 14;   // (measured: 14 bits in 16-bit words 8.0 ms, 30 bits in 32-bit words 9.3, 12 bits 9.4, 64-bit ring 11.8)
	// one-launch scans (scan.hip): published chunk sums, the launch counter that validates them, and the poison flag a kernel raises
	// when a bounded wait ran out (pinned host word; mcom_stream_sync turns it into an error)
	unsigned long long *scan_parts = nullptr; uint32_t scan_epoch = 0; volatile unsigned int *poison = nullptr; unsigned int *d_poison = nullptr;
	// ... and the LAST element of every scan's output, which its kernel also stores into a ring of pinned host words: a caller that reads
	// that element back (the total: nearly every scan is followed by exactly that) finds it there after the next synchronisation --
	// mcom_d2h_async recognises the address and issues no copy (127 of a step's 308 device-to-host copies in round 3's profile)
	enum { SCAN_TOTALS = 64 };
	unsigned long long *scan_tot = nullptr, *d_scan_tot = nullptr;               // host / device view of the ring (after the poison word)
	// (an entry stands only while no other kernel has been launched and the stream has not been synchronised since its scan: launch_gen)
	struct ScanTotal { const void *last; uint32_t bytes, slot, gen; } scan_last[8] = {}; uint32_t scan_last_at = 0, launch_gen = 1;
	// first-come claiming (claim.hip): 0 = the one-launch kernel with the launch-per-round loop behind it, 1 = the loop at once, 2 = the
	// one-launch kernel's first barrier gives up (test hook: the poison flag trips and the loop takes over); how often the loop ran
	int claim_route = 0; uint64_t claim_fallbacks = 0;
	unsigned int *screen_flag = nullptr;                                      // mcom_dicts_screen_begin .. _end
	// the screen's two routes (realign.hip): 0 = keys binned by counter range and counted in LDS, with the global-atomics kernel behind it
	// for a set whose keys pile up in one bin's region; 1 = the global-atomics kernel at once; 2 = regions of a few keys (test hook: the
	// binned route overflows and the other one takes over); the arguments _end needs to run the other route; how often it did
	// blocks of mcom_dmalloc that kernels in flight still use: given back to the pool by the next mcom_stream_sync (mcom_dfree_later) --
	// the pool is shared by every context of the process, so a block may only go back once its stream has passed its last user
	std::vector<void*> free_later;
	int screen_route = 0; uint64_t screen_fallbacks = 0;
	int lookup_route = 0;                                                           // mcom_set_lookup_route (include/mcom_test.h)
	struct ScreenArgs { const uint64_t *sgbits; size_t n_sg; int L, ininumdict, maxsearch, n_shares, share; } screen_args = {};
	// a pool of zeroed words for the counters kernels add to (overflow counts, maxima, totals): handed out front to back and cleared as
	// a whole when it is used up, instead of one 4-byte fill launch in front of every such kernel (mcom_zeroed, api.hip)
	enum { ZPOOL_BYTES = 64 * 1024 };
	unsigned char *zpool = nullptr; size_t zpool_used = 0; int zpool_half = 0;
	// small results on their way to the host (mcom_d2h_async): a page of pinned memory and who waits for what
	struct PinWait { void *dst; size_t off, bytes; const void *from = nullptr; };   // from != nullptr: the value waits there (a scan's total), not in the page
	unsigned char *pin = nullptr; size_t pin_off = 0;
	std::vector<PinWait> pin_wait;
};

// events of the profiler come from a process-wide free list (creating and destroying a pair per launch cost more than recording them)
hipEvent_t mcom_prof_event_get();
void mcom_prof_event_put(hipEvent_t e);
// brackets one kernel launch (or a short launch sequence) with events when the profiler is on
struct McomProfScope {
	mcom_ctx *ctx; int idx;
	McomProfScope(mcom_ctx *c, int id) : ctx(c), idx(-1) {
		if (!c->prof_on) return;
		McomProfSpan s; s.id = id;
		if (!(s.a = mcom_prof_event_get())) return;
		if (!(s.b = mcom_prof_event_get())) { mcom_prof_event_put(s.a); return; }
		(void)hipEventRecord(s.a, c->stream);
		c->prof_open.push_back(s); idx = (int)c->prof_open.size() - 1;
		prev = c->prof_cur; c->prof_cur = id;
	}
	~McomProfScope() { if (idx >= 0) { (void)hipEventRecord(ctx->prof_open[idx].b, ctx->stream); ctx->prof_cur = prev; } }
	int prev = PROF_COUNT;
};
// Every kernel launch of the library goes through MCOM_LAUNCH: hipLaunchKernelGGL, plus -- while the profiler is on -- a tally of
// the instantiation under the class whose scope is open.  mcom_prof_kernels turns the addresses into names with the runtime's own
// table (hipKernelNameRefByPtr, demangled): the spelling rocprofv3 prints for the same kernel, template arguments included.
#define MCOM_LAUNCH(kernel, grid, block, lds, stream, ...) do { \
	++ctx->launch_gen; \
	if (ctx->prof_on) ++ctx->prof_kernels[std::make_pair(ctx->prof_cur, (const void*)&kernel)]; \
	hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__); } while (0)

int mcom_fail(mcom_ctx *ctx, int code, const char *fmt, ...);
// A count or a flag back to the host costs a round trip; into pageable memory (a local variable) the runtime stages it and the
// round trip takes 27 us instead of 15 (tools/ubench/d2h_latency.cpp), several hundred times per job.  mcom_d2h_async sends up to
// 256 bytes through a pinned page of the context; mcom_stream_sync -- which replaces hipStreamSynchronize on the context's stream
// everywhere in the library -- hands them to their destinations.  (mcom_fail drops what is still on its way.)
hipError_t mcom_d2h_async(mcom_ctx *ctx, void *dst, const void *src, size_t bytes);
hipError_t mcom_stream_sync(mcom_ctx *ctx);
hipError_t mcom_stream_sync_poison(mcom_ctx *ctx, bool *poisoned);
int mcom_ws_reserve(mcom_ctx *ctx, size_t bytes);
// `bytes` (<= 4096, rounded up to 8) of zeroed device memory for a kernel of the context's stream to count into; the caller reads what it
// needs back before it returns (the pool has two halves, cleared and taken in turn: a word keeps its value for thousands of further requests).  Without the pool: `fallback` cleared by a
// fill, as before.  nullptr: the fill failed.
void *mcom_zeroed(mcom_ctx *ctx, void *fallback, size_t bytes);
void mcom_ring_flush_slot(mcom_ctx *ctx, uint32_t slot);
unsigned long long *mcom_ring_slot(mcom_ctx *ctx, uint32_t *slot);                      // scan.hip: a kernel's one-value result straight into pinned memory
void mcom_ring_register(mcom_ctx *ctx, const void *d_result, uint32_t bytes, uint32_t slot);
// recycled device blocks for the library's own objects (api.hip)
hipError_t mcom_dmalloc(void **out, size_t bytes);
void mcom_dfree(void *p);
void mcom_dfree_later(mcom_ctx *ctx, void *p);                                           // ... once the context's stream has been synchronised
template <class T> static inline hipError_t mcom_dmalloc(T **out, size_t bytes) { return mcom_dmalloc((void**)out, bytes); }

// internal helpers shared between translation units (sort.hip)
int mcom_scan_u32(mcom_ctx *ctx, const uint32_t *in, uint32_t *out, size_t n, uint32_t *scratch);
size_t mcom_scan_scratch_elems(size_t n);
int mcom_scan_prepare(mcom_ctx *ctx);                    // the scratch of the one-launch scans and the poison flag (scan.hip)
size_t mcom_sort_ws_bytes(size_t n);
int mcom_sort_by_x(mcom_ctx *ctx, mcom_mm128 *d_a, size_t n, int bits, void *ws);
int mcom_sort_by_low_bits(mcom_ctx *ctx, mcom_mm128 *d_a, size_t n, int bits, void *ws);
int mcom_sort_by_low_bits_into_ws(mcom_ctx *ctx, const mcom_mm128 *d_src, mcom_mm128 *d_a, size_t n, int bits, void *ws, mcom_mm128 **d_sorted);
// grouped records (x < 2^bits ascends from group to group) sorted by x inside their groups, in tiles of whole groups (sort.hip)
#define MCOM_GROUP_TILE 1280                 // records per tile of mcom_sort_groups_by_x: with a group of up to ~750 behind it a tile fits the small segment sort (2048)
#define MCOM_GROUP_SCRATCH(n) (4 * ((size_t)(n) / MCOM_GROUP_TILE + 1) + 16)
int mcom_sort_groups_by_x(mcom_ctx *ctx, mcom_mm128 *d_in, mcom_mm128 *d_out, size_t n, const uint64_t *d_goff, size_t ng, int bits, uint32_t *d_scratch);
// consensus of one column range per job (consensus.hip)
// d_tlist: the n_tiles tiles to do out of the tile arrays (NULL: all of them)
int mcom_merge_consensus_regions(mcom_ctx *ctx, const uint64_t *d_packed, const uint64_t *d_members, const uint64_t *d_job_off, const uint64_t *d_ref_off,
                                 const uint32_t *d_tile_job, const uint32_t *d_tile_idx, uint32_t n_tiles, int L, uint8_t *d_refs,
                                 const uint32_t *d_reg_lo, const uint32_t *d_reg_hi, const uint32_t *d_tlist);
// the same by units of 32 columns, bit-sliced (consensus_bs.hip); tiles holding a unit that too many members reach come back as a list
int mcom_merge_consensus_units(mcom_ctx *ctx, const uint64_t *d_packed, const uint64_t *d_members, const uint64_t *d_job_off, const uint64_t *d_ref_off,
                               const uint32_t *d_ujob, const uint32_t *d_uoff, uint32_t n_units, int L, uint8_t *d_refs,
                               const uint32_t *d_reg_lo, const uint32_t *d_reg_hi, const uint32_t *d_toff, uint32_t n_tiles,
                               unsigned int *d_tflag, uint32_t *d_tlist, uint32_t *h_nlist);
// 64-bit exclusive scan (merge.hip): scratch of mcom_scan64_scratch_elems(n) uint64
size_t mcom_scan64_scratch_elems(size_t n);
int mcom_scan64(mcom_ctx *ctx, const uint64_t *in, uint64_t *out, size_t n, uint64_t *scratch);

#define MCOM_HIP(ctx, call) do { hipError_t e__ = (call); if (e__ != hipSuccess) \
	return mcom_fail(ctx, MCOM_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); } while (0)
#define MCOM_LAUNCH_CHECK(ctx) MCOM_HIP(ctx, hipGetLastError())
// A fill or a copy INTO device memory on the context's stream can overwrite the last element of a scan whose total waits in the pinned
// ring (scan.hip): like a kernel launch it ends the validity of those entries.  (`ctx` is in scope wherever the library moves device data.)
#define hipMemsetAsync(...) (++ctx->launch_gen, hipMemsetAsync(__VA_ARGS__))
#define hipMemcpyAsync(dst_, src_, n_, kind_, st_) (((kind_) != hipMemcpyDeviceToHost ? ++ctx->launch_gen : 0u), hipMemcpyAsync(dst_, src_, n_, kind_, st_))

#define U64MAX 0xFFFFFFFFFFFFFFFFull

// hash64: invertible mix on 2k bits (reference sketch.c:27-37)
__device__ __forceinline__ uint64_t mcom_hash64(uint64_t key, uint64_t mask)
{
	key = (~key + (key << 21)) & mask;
	key ^= key >> 24;
	key = (key + (key << 3) + (key << 8)) & mask;
	key ^= key >> 14;
	key = (key + (key << 2) + (key << 4)) & mask;
	key ^= key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}

// The same mix for 17 <= k <= 31 written for the machine: the low word of the mask is all ones, so only the high word is masked
// (mhi = the mask's high word), and the two multiplications by small constants (x265, x21 -- the reference writes them as shifts
// and adds, the compiler folds them into 64 x 32-bit multiply-adds, six v_mad_u64_u32 per k-mer) are chains of v_lshl_add_u64
// (gfx940+: (a << s) + c, s <= 4, one instruction): 265 = 9 + 16 * 16, 21 = 5 + 16.
template <int S> __device__ __forceinline__ uint64_t mcom_lshl_add64(uint64_t a, uint64_t c)
{
	uint64_t d;
	asm("v_lshl_add_u64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "n"(S), "v"(c));
	return d;
}
__device__ __forceinline__ uint64_t mcom_mask_hi(uint64_t v, uint32_t mhi) { return (v & 0xFFFFFFFFull) | ((uint64_t)((uint32_t)(v >> 32) & mhi) << 32); }
// key * C + add (mod 2^64) for a 32-bit constant: one 32 x 32 -> 64 multiply-add for the low word, one 32-bit multiply and an add
// for the high word.  (tools/ubench/valu_rates.hip: v_mad_u64_u32 and v_mul_lo_u32 issue like every other three-operand or 64-bit
// instruction on gfx950, 4.4 cycles per wave, two-operand 32-bit instructions in 2.4.)
__device__ __forceinline__ uint64_t mcom_mul64_c32(uint64_t key, uint32_t C, uint64_t add)
{
	const uint64_t p = (uint64_t)(uint32_t)key * C + add;
	const uint32_t hi = (uint32_t)(key >> 32) * C + (uint32_t)(p >> 32);
	return ((uint64_t)hi << 32) | (uint32_t)p;
}
__device__ __forceinline__ uint64_t mcom_hash64_wide(uint64_t key, uint32_t mhi)
{
	key = mcom_mask_hi(mcom_mul64_c32(key, 0x1FFFFFu, ~0ull), mhi);                 // ~key + (key << 21) = key * (2^21 - 1) - 1
	key ^= key >> 24;
	key = mcom_mask_hi(mcom_mul64_c32(key, 265u, 0), mhi);                          // key + (key << 3) + (key << 8)
	key ^= key >> 14;
	{ const uint64_t k5 = mcom_lshl_add64<2>(key, key); key = mcom_mask_hi(mcom_lshl_add64<4>(key, k5), mhi); }                    // key + (key << 2) + (key << 4)
	key ^= key >> 28;
	key = mcom_mask_hi(key + (key << 31), mhi);
	return key;
}

// 32-bit form of the same mix for 2k <= 32 (mask fits in one register)
__device__ __forceinline__ uint32_t mcom_hash64_lo(uint32_t key, uint32_t mask)
{
	// every shift of the 64-bit form that can move bits across bit 32 is dead once key < 2^32:
	// (key << s) & mask keeps only the low 32 bits, key >> s never imports high bits.
	key = (~key + (key << 21)) & mask;
	key ^= key >> 24;
	key = (key + (key << 3) + (key << 8)) & mask;
	key ^= key >> 14;
	key = (key + (key << 2) + (key << 4)) & mask;
	key ^= key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}

static inline int mcom_words_per_read(int L) { return (2 * L + 63) / 64; }

#ifdef __HIPCC__
// Largest c in [0, n) with le(c) true, for a predicate that is true for a prefix of [0, n) and for c = 0.  Called by
// ALL threads of a block (blockDim.x a multiple of 64, <= 1024): every round the threads probe blockDim.x evenly
// spaced candidates at once, so a search over millions of contig offsets costs three dependent global loads instead
// of the twenty-odd of a one-thread binary search -- which, paid once per block behind a barrier, was what limited
// the position-space kernels.  scratch: 16 uint32 of LDS.
template <class F> __device__ __forceinline__ uint32_t mcom_block_search(uint32_t n, F le, uint32_t *scratch)
{
	uint32_t lo = 0, hi = n;
	const uint32_t T = blockDim.x, tid = threadIdx.x, nwv = T >> 6;
	while (hi - lo > 1) {
		const uint32_t step = (hi - lo + T - 1) / T;
		const uint64_t idx = (uint64_t)lo + (uint64_t)(tid + 1) * step;
		const bool pred = idx < hi && le((uint32_t)idx);
		const uint64_t m = __ballot(pred);
		if ((tid & 63) == 0) scratch[tid >> 6] = (uint32_t)__popcll(m);
		__syncthreads();
		uint32_t cnt = 0;
		for (uint32_t i = 0; i < nwv; ++i) cnt += scratch[i];
		__syncthreads();
		lo += cnt * step;
		if (lo + step < hi) hi = lo + step;
	}
	return lo;
}
#endif

// ---- exact key -> (run start, run length) map over a sorted record array (table.hip) ---------------------
// Open addressing, linear probing, 16-byte slots {key, start | count << 32}, EMPTY key = ~0, load <= 0.5.
// Bucketed form (the contig-minimizer index, whose records are sorted by bucket x & (2^bbits - 1) first): bucket v owns the
// region of `region` slots (any number, a multiple of four: 64-byte lines) starting at v * region, a key probes inside its bucket's
// region only -- so the table is BUILT bucket by bucket in LDS and written once, in order (table.hip), instead of by scattered
// atomicCAS, and it is as large as the fullest bucket asks for, not the next power of two.  region = 0:
// one global table.
struct McomTable {
	uint64_t *slots; uint32_t log2cap;
	uint32_t numkeys, maxrun;
	uint32_t region, bbits;
};
// sorted: n records sorted by x (runs of equal x are the bins); head/scr: scratch of n and
// mcom_scan_scratch_elems(n)+256 uint32; meta: 2 uint32 on the device.  Synchronous.
int mcom_table_build(mcom_ctx *ctx, const mcom_mm128 *sorted, size_t n, uint32_t *head, uint32_t *scr, uint32_t *meta, McomTable *t);
void mcom_table_free(McomTable *t);

__device__ __forceinline__ uint32_t mcom_slot_of(uint64_t key, uint32_t log2cap)
{
	return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> (64 - log2cap));
}
__device__ __forceinline__ bool mcom_table_find(const uint64_t *slots, uint32_t log2cap, uint64_t key, uint32_t &start, uint32_t &count)
{
	const uint32_t capm = (1u << log2cap) - 1u;
	uint32_t sl = mcom_slot_of(key, log2cap);
	for (;;) {
		const uint64_t k = slots[2 * (size_t)sl];
		if (k == key) { const uint64_t v = slots[2 * (size_t)sl + 1]; start = (uint32_t)v; count = (uint32_t)(v >> 32); return true; }
		if (k == ~0ull) return false;
		sl = (sl + 1) & capm;
	}
}
__device__ __forceinline__ uint32_t mcom_region_slot(uint64_t key, uint32_t region)
{
	return (uint32_t)((((key * 0x9E3779B97F4A7C15ull) >> 32) * (uint64_t)region) >> 32);
}
__device__ __forceinline__ bool mcom_table_find_any(const uint64_t *slots, uint32_t log2cap, uint32_t region, uint32_t bbits, uint64_t key, uint32_t &start, uint32_t &count)
{
	if (!region) return mcom_table_find(slots, log2cap, key, start, count);
	const uint64_t base = (key & ((1ull << bbits) - 1)) * (uint64_t)region;
	uint32_t sl = mcom_region_slot(key >> bbits, region);
	for (;;) {
		const uint64_t k = slots[2 * (base + sl)];
		if (k == key) { const uint64_t v = slots[2 * (base + sl) + 1]; start = (uint32_t)v; count = (uint32_t)(v >> 32); return true; }
		if (k == ~0ull) return false;
		sl = sl + 1 == region ? 0 : sl + 1;
	}
}
