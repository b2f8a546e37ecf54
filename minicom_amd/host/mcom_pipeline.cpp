// minicom_amd/host/mcom_pipeline.cpp -- host driver of the MI355X-native minicom hot path.
//
// Restates the control flow of the reference's pre_process (preprocess.c:39-241) and of its stage drivers,
// with every hot loop replaced by a call into libmcom_hip.so (include/mcom.h), including the contig consensus
// (mcom_group_consensus / mcom_merge_consensus).  What stays on the host is what the reference keeps
// sequential by design: the order in which singletons and contigs are appended, first-come pair claiming, and
// the order in which Stage-2 claims are appended.  Contigs live in flat arrays (members and consensus strings
// back to back + offsets), so no stage allocates per contig.  Citations are file:line into yuansliu/minicom src/.
#include "../../include/mcom.h"
#include "../../include/mcom_host.h"
#include "mcom_fastq.hpp"
#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cinttypes>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int NB_BITS = 14;                 // MM_IDX_DEF_B, minicommain.c:175
constexpr uint64_t U64MAX = ~0ull;

// Process-wide pool of page-locked host blocks.  Pinning memory (hipHostMalloc) and first-touching fresh pages are
// both expensive, and every large host array here is a DMA source or target, so blocks are pinned once and reused
// by later pipelines of the same process (the pool never returns memory to the system).
class PinnedPool {
	std::mutex mu;
	std::multimap<size_t, void*> free_;
public:
	void *get(size_t bytes, size_t &cap) {
		const size_t unit = (size_t)2 << 20;
		cap = ((bytes + bytes / 8 + unit - 1) / unit) * unit;
		{
			std::lock_guard<std::mutex> g(mu);
			auto it = free_.lower_bound(cap);
			if (it != free_.end() && it->first <= 2 * cap + unit) { void *p = it->second; cap = it->first; free_.erase(it); return p; }
		}
		void *p = nullptr;
		if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) p = nullptr;
		return p;
	}
	void put(void *p, size_t cap) { if (!p) return; std::lock_guard<std::mutex> g(mu); free_.emplace(cap, p); }
};
PinnedPool &pinned_pool() { static PinnedPool *pool = new PinnedPool(); return *pool; }

// minimal vector over pooled pinned memory: no value initialisation, contents kept on growth
template <class T> class PinVec {
	T *p_ = nullptr; size_t n_ = 0, cap_ = 0;      // cap_ in bytes
public:
	PinVec() {}
	PinVec(const PinVec&) = delete; PinVec &operator=(const PinVec&) = delete;
	PinVec(PinVec &&o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_) { o.p_ = nullptr; o.n_ = o.cap_ = 0; }
	PinVec &operator=(PinVec &&o) noexcept { if (this != &o) { release(); p_ = o.p_; n_ = o.n_; cap_ = o.cap_; o.p_ = nullptr; o.n_ = o.cap_ = 0; } return *this; }
	~PinVec() { release(); }
	void release() { pinned_pool().put(p_, cap_); p_ = nullptr; n_ = cap_ = 0; }
	bool resize(size_t n) {
		if (n * sizeof(T) > cap_) {
			size_t nc = 0; T *q = (T*)pinned_pool().get(std::max<size_t>(n * sizeof(T), 64), nc);
			if (!q) return false;
			if (n_) memcpy(q, p_, n_ * sizeof(T));
			pinned_pool().put(p_, cap_);
			p_ = q; cap_ = nc;
		}
		n_ = n; return true;
	}
	void clear() { n_ = 0; }
	size_t size() const { return n_; }
	T *data() { return p_; } const T *data() const { return p_; }
	T &operator[](size_t i) { return p_[i]; } const T &operator[](size_t i) const { return p_[i]; }
	void swap(PinVec &o) { std::swap(p_, o.p_); std::swap(n_, o.n_); std::swap(cap_, o.cap_); }
};

// Large host arrays that every pipeline makes anew (the singleton list and its flags: 4 + 1 bytes per unclustered read) come from a
// process-wide pool of malloc'ed blocks too: a fresh 64 MB vector costs its page faults when it is first written and the unmapping
// of its pages when it is freed -- 8 ms per step at 100 M reads, most of it in the pipeline's destructor.
class HostPool {
	std::mutex mu;
	std::multimap<size_t, void*> free_;
public:
	void *get(size_t bytes) {                                              // the block carries its capacity in the 64 bytes before it
		const size_t unit = (size_t)1 << 20, cap = ((bytes + bytes / 8 + unit - 1) / unit) * unit;
		{
			std::lock_guard<std::mutex> g(mu);
			auto it = free_.lower_bound(cap);
			if (it != free_.end() && it->first <= 2 * cap + unit) { void *p = it->second; free_.erase(it); return p; }
		}
		char *raw = (char*)malloc(cap + 64);
		if (!raw) throw std::bad_alloc();
		*(size_t*)raw = cap;
		return raw + 64;
	}
	void put(void *p) { if (!p) return; const size_t cap = *(size_t*)((char*)p - 64); std::lock_guard<std::mutex> g(mu); free_.emplace(cap, p); }
	// the free blocks go back to the C library (mcomh_pool_trim: a process that has run jobs of many sizes keeps a block of every size otherwise)
	void trim() { std::lock_guard<std::mutex> g(mu); for (auto &kv : free_) free((char*)kv.second - 64); free_.clear(); }
};
HostPool &host_pool() { static HostPool *pool = new HostPool(); return *pool; }
template <class T> struct PoolAlloc {
	using value_type = T;
	PoolAlloc() = default;
	template <class U> PoolAlloc(const PoolAlloc<U>&) {}
	static constexpr size_t SMALL = (size_t)1 << 20;
	T *allocate(size_t n) { const size_t b = n * sizeof(T); if (b < SMALL) { void *q = malloc(b ? b : 1); if (!q) throw std::bad_alloc(); return (T*)q; } return (T*)host_pool().get(b); }
	void deallocate(T *q, size_t n) { if (n * sizeof(T) < SMALL) free(q); else host_pool().put(q); }
	template <class U> bool operator==(const PoolAlloc<U>&) const { return true; }
	template <class U> bool operator!=(const PoolAlloc<U>&) const { return false; }
};
using U32Pooled = std::vector<uint32_t, PoolAlloc<uint32_t>>;
using U8Pooled = std::vector<uint8_t, PoolAlloc<uint8_t>>;

// all contigs of one stage: members (rid<<32 | offset<<1 | dir, breads.h:49-58) and consensus strings, flat
struct ContigSet {
	PinVec<uint64_t> mem; std::vector<uint64_t> moff{0};
	PinVec<char> ref;
	std::vector<uint64_t> roff{0};
	size_t n() const { return moff.size() - 1; }
	void clear() { mem.clear(); ref.clear(); moff.assign(1, 0); roff.assign(1, 0); }
	size_t msize(size_t i) const { return (size_t)(moff[i + 1] - moff[i]); }
	size_t rsize(size_t i) const { return (size_t)(roff[i + 1] - roff[i]); }
};

// Process-wide pool of device blocks, per device: hipMalloc / hipFree of multi-GB buffers cost milliseconds each and
// a pipeline needs dozens of them; later pipelines of the same process reuse the blocks.
class DevicePool {
	std::mutex mu;
	std::multimap<std::pair<int, size_t>, void*> free_;
public:
	void *get(size_t bytes, size_t &cap) {
		int dev = 0; (void)hipGetDevice(&dev);
		const size_t unit = (size_t)2 << 20;
		cap = ((bytes + bytes / 8 + unit - 1) / unit) * unit;
		{
			std::lock_guard<std::mutex> g(mu);
			auto it = free_.lower_bound(std::make_pair(dev, cap));
			if (it != free_.end() && it->first.first == dev && it->first.second <= 2 * cap + unit) { void *p = it->second; cap = it->first.second; free_.erase(it); return p; }
		}
		void *p = nullptr;
		if (hipMalloc(&p, cap) != hipSuccess) {
			// out of memory: give the pooled blocks back -- this pool's, then the library's -- and try once more.  (The failed call's error is
			// taken off the runtime's books: left there, the next kernel-launch check would report an out-of-memory that was recovered from.)
			(void)hipGetLastError();
			trim();
			if (hipMalloc(&p, cap) != hipSuccess) { (void)hipGetLastError(); mcom_pool_trim(); if (hipMalloc(&p, cap) != hipSuccess) { (void)hipGetLastError(); p = nullptr; } }
		}
		return p;
	}
	void trim() {
		std::vector<void*> drop;
		{ std::lock_guard<std::mutex> g(mu); for (auto &kv : free_) drop.push_back(kv.second); free_.clear(); }
		for (void *q : drop) (void)hipFree(q);
		(void)hipGetLastError();
	}
	void put(void *p, size_t cap) { if (!p) return; int dev = 0; (void)hipGetDevice(&dev); std::lock_guard<std::mutex> g(mu); free_.emplace(std::make_pair(dev, cap), p); }
};
static void device_pool_trim_hook();
DevicePool &device_pool() { static DevicePool *pool = [] { mcom_set_oom_hook(&device_pool_trim_hook); return new DevicePool(); }(); return *pool; }
static void device_pool_trim_hook() { device_pool().trim(); }                    // (what the library calls when IT runs out of device memory)

template <class T> struct DevBuf {
	T *p = nullptr; size_t cap = 0;            // cap in elements
	size_t bytes_ = 0;
	DevBuf() = default;
	DevBuf(const DevBuf&) = delete;
	DevBuf &operator=(const DevBuf&) = delete;
	DevBuf(DevBuf &&o) noexcept { swap(o); }
	DevBuf &operator=(DevBuf &&o) noexcept { if (this != &o) { device_pool().put(p, bytes_); p = nullptr; cap = 0; bytes_ = 0; swap(o); } return *this; }
	~DevBuf() { device_pool().put(p, bytes_); }
	void swap(DevBuf &o) { std::swap(p, o.p); std::swap(cap, o.cap); std::swap(bytes_, o.bytes_); }
	// reserve that keeps the first `keep` elements
	bool grow(size_t n, size_t keep, hipStream_t st) {
		if (n <= cap) return true;
		DevBuf nb;
		if (!nb.reserve(n + n / 4)) return false;
		if (keep && p) { if (hipMemcpyAsync(nb.p, p, keep * sizeof(T), hipMemcpyDeviceToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return false; }
		swap(nb);
		return true;
	}
	bool reserve(size_t n) {
		if (n <= cap) return true;
		device_pool().put(p, bytes_); p = nullptr; cap = 0; bytes_ = 0;
		size_t got = 0;
		p = (T*)device_pool().get(std::max<size_t>(n * sizeof(T), 256), got);
		if (!p) return false;
		bytes_ = got; cap = got / sizeof(T); return true;
	}
};

// a contig set on the device (include/mcom.h, "the contig set of combine_cluster")
struct DevSet {
	DevBuf<uint8_t> seq; DevBuf<uint64_t> soff, mem, moff; DevBuf<mcom_mm128> rec; DevBuf<uint32_t> roff;
	size_t n = 0; uint64_t chars = 0, members = 0, nrec = 0;
	void swap(DevSet &o) {
		seq.swap(o.seq); soff.swap(o.soff); mem.swap(o.mem); moff.swap(o.moff); rec.swap(o.rec); roff.swap(o.roff);
		std::swap(n, o.n); std::swap(chars, o.chars); std::swap(members, o.members); std::swap(nrec, o.nrec);
	}
};

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// static split of [0, n) over nt threads
template <class F> void parallel_for(int nt, size_t n, F &&fn)
{
	if (nt > 1 && n < 4096) nt = 1;
	if (nt <= 1) { fn(0, (size_t)0, n); return; }
	std::vector<std::thread> th;
	const size_t chunk = (n + (size_t)nt - 1) / (size_t)nt;
	for (int t = 0; t < nt; ++t) {
		const size_t b = std::min(n, (size_t)t * chunk), e = std::min(n, b + chunk);
		if (b >= e) break;
		th.emplace_back([&fn, t, b, e]() { fn(t, b, e); });
	}
	for (auto &t : th) t.join();
}

// cmpcluster2: offset ascending, then direction (kthread_cb.c:54-69); the reference's qsort is glibc's merge sort
bool less_cluster2(uint64_t a, uint64_t b)
{
	const int pa = (int)((uint32_t)a >> 1), pb = (int)((uint32_t)b >> 1);
	if (pa != pb) return pa < pb;
	return (int)(a & 1) < (int)(b & 1);
}

} // namespace

struct mcomh_pipeline {
	mcom_ctx *ctx = nullptr;
	hipStream_t stream = nullptr;
	std::string err;
	size_t n = 0; int L = 0, W = 0, NW = 0;
	int k = 0, e = 0, m = 0, rw = 0, cbthr = 0, max_rounds = 0, step = 0, maxthr = 0, numdict = 0, maxsearch = 500, maxsearch_forced = 0;
	bool resketch = true;
	int host_threads = 1;
	// device
	DevBuf<uint8_t> d_ascii_own; const uint8_t *d_ascii = nullptr; size_t pitch = 0;
	const uint8_t *stream_host = nullptr;    // reads in caller-owned host memory, uploaded chunk by chunk (mcomh_create_streamed)
	const uint64_t *ext_packed = nullptr;    // packed-row input (mcomh_create_packed): no classification stage
	uint8_t *d_ascii_adopted = nullptr;                                // a character matrix made by mcomh_create_from_fastq: freed with the pipeline
	bool packed_in_place = false;                                      // d_packed / d_nmask hold the reads as a packing parser sent them (codes + N flags): kt_for_reads classifies them in place
	const uint64_t *ext_x = nullptr; const uint32_t *ext_ylow = nullptr;   // ... whose minimizers came along (mcomh_set_records)
	DevBuf<uint64_t> d_packed, d_nmask; DevBuf<uint8_t> d_cls; DevBuf<uint16_t> d_ncnt; DevBuf<mcom_mm128> d_rec;
	// host
	std::vector<uint8_t> h_ascii;            // only when the reads came from the host (needed for the N dump)
	PinVec<uint8_t> h_cls; bool h_cls_valid = false;     // (the class array on the host: only when the list of special reads overflowed, or for mcomh_dump_stages)
	DevBuf<uint64_t> d_special; PinVec<uint64_t> h_special;
	std::vector<uint32_t> allA, allT, allN, fpA, fpT, fpN, Nfile;
	U32Pooled sg;
	U8Pooled sg_flag;
	bool sg_flag_zero = false;                                        // sg_flag was cleared and nobody has written to it since (updateSingle need not look through 16 M zeros)
	// the flags of the last Stage-2 pass as they came from the device (0 live, 1 / 2 near-poly, 3 claimed); sg_flag (1 = gone) is
	// made from them only when somebody looks (ensure_sg_flag): a pass does not wait for a 16 M-entry host loop
	PinVec<uint8_t> raw_flags; bool raw_flags_valid = false;
	// second stream: transfers that nobody on the main stream waits for (the read classes going to the host, the singleton list
	// coming up for Stage 2); in-stream they held the next kernels back for milliseconds
	hipStream_t copy_stream = nullptr; hipEvent_t ev_main = nullptr, ev_sg = nullptr;
	// a second library context on the copy stream: the first Stage-2 pass gathers the singletons' rows and runs the dictionary screen
	// there, beside the contig index build on the main stream (atomics against streaming writes); what it leaves for the pass
	mcom_ctx *ctx2 = nullptr; hipEvent_t ev_early = nullptr; int device = 0; bool prof_on = false;
	struct Early { bool on = false; size_t n_sg = 0; DevBuf<uint32_t> sg; DevBuf<uint64_t> sgbits; } early;
	PinVec<uint32_t> sg_pin; bool sg_uploaded = false;
	ContigSet C, Cnext;                      // Cnext: the other half of a double buffer, kept to reuse its memory
	// Stage 2 never reads the member lists, so the appends of the passes stay on the device (contig, member; in the
	// reference's appending order) and are folded into the lists by materialize() (mcom_members_finalize) when Stage 2 ends
	// or somebody looks at the members.  A pass that appended nothing has an entry too: its scan sorted the contigs.
	struct Appended { DevBuf<uint32_t> contig; DevBuf<uint64_t> member; size_t n = 0; };
	std::vector<Appended> pend;
	size_t n_pending = 0;
	// The contig set lives on the device from the bucket stage on (strings, members, offsets); the host gets the offsets
	// when a stage needs them and everything when somebody asks for it (accessors, stage dumps, the stream writer).
	DevSet dC;
	bool dC_valid = false, hostC_valid = true, host_off_valid = true;
	uint64_t maxlen = 0;                     // longest contig (bounds the member offsets)
	// updateSingle (preprocess.c:243-255) happens on the device at the end of a pass: the next pass finds its singleton ids
	// in d_sg_live, the host its compacted list in sg_next
	DevBuf<uint32_t> d_sg_live; size_t n_sg_live = 0; bool sg_live_valid = false;
	U32Pooled sg_next; bool sg_next_valid = false;
	// the singleton list of the bucket stage is put together by a host thread beside combine_cluster's GPU work
	std::thread sg_thread, cls_thread;       // (and the class lists of kt_for_reads beside the bucket stage)
	hipEvent_t ev_cls = nullptr; std::atomic<bool> cls_failed{false};        // (written by the class-list thread)
	void join_cls() { if (cls_thread.joinable()) cls_thread.join(); }
	void join_sg() { join_cls(); if (sg_thread.joinable()) sg_thread.join(); }
	// contigs of the current stage on the device
	DevBuf<uint8_t> d_cseq; DevBuf<uint64_t> d_coff_chars, d_coff_words, d_cbits, d_woff; DevBuf<uint32_t> d_clen;
	// the packed form of the set after a merge round is made from the round's own (merged contigs packed, the others' words copied):
	// the second set of buffers, and whether d_cbits / d_coff_words / d_clen describe the set in dC
	DevBuf<uint64_t> d_coff_words_alt, d_cbits_alt; DevBuf<uint32_t> d_clen_alt;
	bool cbits_for_dC = false;
	// several GPUs: the bucket stage's contigs travel as packed words, straight into d_cbits (kt_for_bucket); combine_cluster finds them there
	bool prepacked = false; uint64_t prepacked_words = 0;
	std::vector<uint64_t> h_coff_words;
	uint64_t total_words = 0, n_windows = 0;
	DevBuf<uint64_t> d_cix_keys; uint64_t cix_geom = 0;    // klen-mer index of the Stage-2 contigs (mcom_cindex_build): this rank's share
	int full_consensus = 0;                                           // 1: count every column of a merged contig (A/B switch)
	bool read_batches = false;                                        // kt_for_reads in batches over two streams (A/B switch: mcomh_params.read_batches = 1)
	bool overlap_screen = false;                                      // true: the first Stage-2 pass's row gather + screen on the copy stream (A/B switch)
	int stream_sets = 1;                                              // stream sets cluster_dump writes (the reference: one per thread)
	bool host_dump = false;                                           // true: cluster_dump's default mode on the host, as the -p / paired-end modes (A/B switch)
	int window_scan = 0;                                              // 1: window-driven kernel (mcom_realign_pass) instead
	bool stage2_uploaded = false;
	// Stage 2 as a partition-local join (round 5, csrc/realign.hip): the index entries sorted by partition, kept until the first pass has
	// joined them with the singletons' keys (jn_entries); what that pass defers is all the later passes need (jn_deferred)
	bool stage2_table = true;                                          // false: the partition-local join (mcomh_params.stage2_join)
	DevBuf<uint32_t> jn_keyA, jn_keyB, jn_pstart, jn_map; DevBuf<uint64_t> jn_slotA, jn_slotB, jn_defer;
	const uint32_t *jn_ek = nullptr; const uint64_t *jn_es = nullptr;
	bool jn_entries = false, jn_deferred = false; uint64_t jn_ndefer = 0, jn_nwords = 0;
	bool screen_clear = false;                                             // a pass of this Stage 2 proved that no dictionary bin exceeds maxsearch
	std::map<std::string, double> stat;
	// multi-GPU (include/mcom_host.h, mcomh_create_dist): this rank holds reads [rid0, rid0 + n_local) of n; packed rows,
	// classes, N masks and -- from the bucket stage on -- the contig set are replicated on every rank
	mcomh_comm *comm = nullptr; int rank = 0, world = 1;
	size_t n_local = 0; uint64_t rid0 = 0;
	std::vector<uint64_t> shard_lo;                                        // [world + 1] first read of every rank
	std::vector<uint64_t> sg_round_len;                                    // this rank's singles + rejects per bucket round
	bool sg_gathered = true;
	uint32_t sp_cap = 1u << 20, sp_first = 4096;                          // room of the special-read list on the device / entries copied with the count (mcomh_test_special_capacity)
	long inject_at = -1, flag_exchanges = 0;                             // test hook: fail right before the inject_at-th flag exchange (mcomh_test_inject_failure)
	bool peers_know = false;                                          // multi-GPU: a failure has been announced to (or learnt from) the other ranks
	uint32_t cix_c0 = 0, cix_c1 = 0;                                       // contigs whose 17-mers this rank indexes in Stage 2

	int fail(int code, const char *fmt, ...) {
		char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
		err = buf; return code;
	}
	int gpu(int rc) { if (rc) err = std::string("libmcom_hip: ") + mcom_last_error(ctx); return rc; }
	int hipc(hipError_t e_, const char *what) { if (e_ != hipSuccess) return fail(MCOM_E_HIP, "%s: %s", what, hipGetErrorString(e_)); return 0; }
	template <class T> int d2h(T *dst, const T *src, size_t cnt, const char *what) {
		return cnt ? hipc(hipMemcpyAsync(dst, src, cnt * sizeof(T), hipMemcpyDeviceToHost, stream), what) : 0;
	}
	template <class T> int h2d(T *dst, const T *src, size_t cnt, const char *what) {
		return cnt ? hipc(hipMemcpyAsync(dst, src, cnt * sizeof(T), hipMemcpyHostToDevice, stream), what) : 0;
	}
	int sync(const char *what) { return hipc(hipStreamSynchronize(stream), what); }
};

using P = mcomh_pipeline;
// the clock of the stage timers: wall time NOT spent inside exchanges (multi-GPU; mcomh_comm_seconds covers staging copies and the
// wait for the peers), so that a stage's timers say what this rank itself was busy with.  One GPU: the wall clock.
static double busy_now(const P *p) { return now_ms() - (p->comm ? 1e3 * mcomh_comm_seconds(p->comm) : 0.0); }
static int materialize(P *p);
static int ensure_packed_contigs(P *p);
static void ensure_sg_flag(P *p)
{
	if (!p->raw_flags_valid) return;
	const size_t n = p->sg.size();
	p->sg_flag.resize(n);
	const uint8_t *f = p->raw_flags.data(); uint8_t *sf = p->sg_flag.data();
	for (size_t i = 0; i < n; ++i) sf[i] = f[i] ? 1 : 0;                       // flagged or claimed
	p->raw_flags_valid = false; p->sg_flag_zero = false;
}
static int ensure_host_contigs(P *p, bool wait_data = true);
static const char ACGT[] = "ACGT";

// ---- multi-GPU plumbing: everything goes through the one all-to-all of host/mcom_comm.cpp ---------------------------
static int comm_rc(P *p, int rc) { if (rc) p->err = std::string("communicator: ") + mcomh_comm_last_error(p->comm); return rc; }
// Failure protocol.  A rank that hits an error between two collectives and simply returned would leave the others waiting in the
// next one for ever (round 3's advisor finding: contig index build, merge-round index, merge shares ...).  So every exchange of the
// pipeline starts with one flag word per rank, exchanged on its own (always the same size, so that any two such exchanges pair): "I have failed".  A rank that fails
// in its local work announces it with ONE such flag exchange when its stage function returns (stage_exit below); that exchange pairs
// with the flag exchange at the head of the collective the others are about to enter, they see the flag, skip the collective and
// return too.  Every rank issues the same sequence of flag exchanges up to the failure, so they always pair.
// the flag exchange: one word per rank, always the same size, so that any two of them pair -- a rank's announcement with whatever
// collective the others are about to enter.  fail_flag: this rank has failed.
static int flags(P *p, bool fail_flag)
{
	const int R = p->world;
	if (!fail_flag && ++p->flag_exchanges == p->inject_at) return p->fail(MCOM_E_ARG, "injected failure in front of flag exchange %ld (test hook)", p->inject_at);
	std::vector<uint64_t> buf((size_t)R, 0), off(R), bytes(R);
	for (int q = 0; q < R; ++q) { off[q] = (uint64_t)q * 8; bytes[q] = 8; }
	buf[(size_t)p->rank] = fail_flag ? 1 : 0;
	int rc = comm_rc(p, mcomh_comm_allgatherv(p->comm, nullptr, buf.data(), off.data(), bytes.data(), 0, p->stream));
	if (rc) { p->peers_know = true; return rc; }                              // (a broken transport: nothing more can be said to anybody)
	int failed = -1;
	for (int q = 0; q < R; ++q) if (buf[(size_t)q] && failed < 0) failed = q;
	if (failed >= 0) {
		p->peers_know = true;
		if (!fail_flag) return p->fail(MCOM_E_HIP, "rank %d failed in its part of this stage", failed);
	}
	return MCOM_OK;
}
// k values of every rank, rank-major: all[q * k + i]
static int gather_host(P *p, const uint64_t *mine, size_t k, std::vector<uint64_t> &all)
{
	const int R = p->world;
	int rc = flags(p, false);
	if (rc) return rc;
	all.assign((size_t)R * k, 0);
	if (!k) return MCOM_OK;
	std::vector<uint64_t> off(R), bytes(R);
	for (int q = 0; q < R; ++q) { off[q] = (uint64_t)q * k * 8; bytes[q] = k * 8; }
	memcpy(all.data() + (size_t)p->rank * k, mine, k * 8);
	return comm_rc(p, mcomh_comm_allgatherv(p->comm, nullptr, all.data(), off.data(), bytes.data(), 0, p->stream));
}
// the flag exchange alone: in front of a collective that moves device data
static int guard(P *p) { return flags(p, false); }
// what a stage function of a multi-GPU pipeline returns through: a rank that failed on its own tells the others
static int stage_exit(P *p, int rc)
{
	if (rc && p->comm && !p->peers_know) { const std::string e = p->err; (void)flags(p, true); p->err = e; p->peers_know = true; }
	return rc;
}
// sums / minima / maxima of a few counters (op 0 / 1 / 2), on the flag-carrying exchange
static int allreduce_host(P *p, uint64_t *vals, size_t n, int op)
{
	std::vector<uint64_t> all;
	int rc = gather_host(p, vals, n, all);
	if (rc) return rc;
	for (size_t i = 0; i < n; ++i) {
		uint64_t v = all[i];
		for (int q = 1; q < p->world; ++q) { const uint64_t x = all[(size_t)q * n + i]; v = op == 0 ? v + x : op == 1 ? std::min(v, x) : std::max(v, x); }
		vals[i] = v;
	}
	return MCOM_OK;
}
// all-gather of a device array: rank q's part is elements [first[q], first[q] + cnt[q]); send = NULL: mine is in place.
// guarded = false: the exchange directly follows another one, with nothing in between that can fail
// bytes this rank has sent so far (the exchanges' statistics b_x_*: what each of them sends per step)
static double xbytes(P *p) { uint64_t b = 0; if (p->comm) mcomh_comm_stats(p->comm, &b, nullptr); return (double)b; }
template <class T> static int gatherv(P *p, T *buf, const std::vector<uint64_t> &first, const std::vector<uint64_t> &cnt, const T *send = nullptr, bool guarded = true)
{
	const int R = p->world;
	int rc;
	if (guarded && (rc = guard(p))) return rc;
	std::vector<uint64_t> off(R), bytes(R);
	for (int q = 0; q < R; ++q) { off[q] = first[q] * sizeof(T); bytes[q] = cnt[q] * sizeof(T); }
	return comm_rc(p, mcomh_comm_allgatherv(p->comm, send, buf, off.data(), bytes.data(), 1, p->stream));
}
// all-to-all of device data (offsets and sizes in bytes)
static int alltoallv_dev(P *p, const void *send, const uint64_t *so, const uint64_t *sb, void *recv, const uint64_t *ro, const uint64_t *rb, bool guarded = true)
{
	int rc;
	if (guarded && (rc = guard(p))) return rc;
	return comm_rc(p, mcomh_comm_alltoallv(p->comm, send, so, sb, recv, ro, rb, 1, p->stream));
}
// Stage-2 claim keys: every rank holds the minimum over ITS contigs; the claim is the minimum over all.
// Reduce-scatter by all-to-all (rank q folds slice q of everybody), then all-gather of the folded slices: direct
// peer-to-peer traffic on every link, not a ring.
static int dist_min_claims(P *p, uint64_t *d_claim, size_t n)
{
	const int R = p->world, me = p->rank;
	if (R == 1 || n == 0) return MCOM_OK;
	std::vector<uint64_t> lo(R + 1), so(R), sb(R), ro(R), rb(R), first(R), cnt(R);
	for (int q = 0; q <= R; ++q) lo[q] = (uint64_t)n * q / R;
	const uint64_t len = lo[me + 1] - lo[me];
	DevBuf<uint64_t> parts;
	if (!parts.reserve((size_t)R * len + 1)) return p->fail(MCOM_E_NOMEM, "claim shares");
	for (int q = 0; q < R; ++q) { so[q] = lo[q] * 8; sb[q] = (lo[q + 1] - lo[q]) * 8; ro[q] = (uint64_t)q * len * 8; rb[q] = len * 8; first[q] = lo[q]; cnt[q] = lo[q + 1] - lo[q]; }
	int rc = alltoallv_dev(p, d_claim, so.data(), sb.data(), parts.p, ro.data(), rb.data());
	if (rc) return rc;
	if ((rc = p->gpu(mcom_min_fold_u64(p->ctx, parts.p, R, (size_t)len, (size_t)len, d_claim + lo[me])))) return rc;
	if ((rc = gatherv(p, d_claim, first, cnt))) return rc;
	return p->sync("claim minimum");
}

// ----------------------------------------------------------------------------------------------------
// construction
// ----------------------------------------------------------------------------------------------------
extern "C" int mcomh_create(mcomh_pipeline **out, int device, void *hip_stream, const uint8_t *host_reads,
                            const uint8_t *d_reads, size_t pitch, size_t n, int L, const mcomh_params *pp)
{
	if (!out) return MCOM_E_ARG;
	*out = nullptr;
	if ((host_reads == nullptr) == (d_reads == nullptr) && n) return MCOM_E_ARG;
	if (L < 1 || L > 256) return MCOM_E_ARG;
	mcomh_params z; memset(&z, 0, sizeof z);
	if (!pp) pp = &z;
	P *p = new P();
	p->stream = (hipStream_t)hip_stream;
	int rc = mcom_create(&p->ctx, device, hip_stream);
	if (rc) { delete p; return rc; }
	p->device = device;
	p->n = n; p->L = L; p->W = (2 * L + 63) / 64; p->NW = (L + 63) / 64;
	p->k = pp->k > 0 ? pp->k : (L < 80 ? 17 : 31);                                  // minicommain.c:92-114
	p->e = pp->e > 0 ? pp->e : 4;
	p->m = pp->m > 0 ? pp->m : 6;
	p->cbthr = pp->cbthr > 0 ? pp->cbthr : 2 * p->e;                                // :122-126
	p->max_rounds = (pp->max_rounds > 0 && pp->max_rounds < 35) ? pp->max_rounds : 35; // :127-129
	p->step = pp->step > 0 ? pp->step : (p->e > 10 ? 5 : p->e);                     // :130-137
	p->maxthr = pp->maxthr > 0 ? pp->maxthr : L / 2;                                // :140-143
	p->rw = L >= 70 ? L / 2 - p->k : 3;                                             // preprocess.c:89-107
	if (pp->w > 0) p->rw = pp->w;
	p->numdict = pp->numdict;
	p->window_scan = pp->window_scan == 1;
	p->full_consensus = pp->full_consensus == 1;
	p->resketch = pp->full_sketch != 1;
	p->host_dump = pp->host_dump == 1;
	p->stream_sets = pp->stream_sets > 1 ? std::min(pp->stream_sets, 4096) : 1;
	p->overlap_screen = pp->overlap_screen == 1;
	p->read_batches = pp->read_batches == 1;
	p->stage2_table = pp->stage2_join != 1;
	p->maxsearch_forced = pp->maxsearch > 0 ? pp->maxsearch : 0;
	p->host_threads = pp->host_threads > 0 ? pp->host_threads : 1;
	if (p->k > 31 || p->k < 11 || p->rw < 1 || p->rw > 128) { mcomh_destroy(p); return MCOM_E_ARG; }
	if (hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&p->ev_main, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&p->ev_sg, hipEventDisableTiming) != hipSuccess) { mcomh_destroy(p); return MCOM_E_HIP; }   // (every failure exit releases what exists)
	// the copy stream's own library context (Stage 2 runs the singletons' rows and the dictionary screen there beside the index build):
	// made with the pipeline, not in front of the first pass (0.8 ms of an idle GPU between the stages); a failure here is not one yet
	if (mcom_create(&p->ctx2, p->device, p->copy_stream) != MCOM_OK) p->ctx2 = nullptr;
	if (hipEventCreateWithFlags(&p->ev_early, hipEventDisableTiming) != hipSuccess) { p->ev_early = nullptr; (void)hipGetLastError(); }
	if (host_reads) {
		p->pitch = (size_t)L;
		p->h_ascii.assign(host_reads, host_reads + n * (size_t)L);
		if (!p->d_ascii_own.reserve(n * (size_t)L + 16)) { mcomh_destroy(p); return MCOM_E_NOMEM; }
		if (n && hipMemcpyAsync(p->d_ascii_own.p, host_reads, n * (size_t)L, hipMemcpyHostToDevice, p->stream) != hipSuccess) { mcomh_destroy(p); return MCOM_E_HIP; }
		p->d_ascii = p->d_ascii_own.p;
	} else { p->d_ascii = d_reads; p->pitch = pitch; }
	*out = p;
	return MCOM_OK;
}

extern "C" int mcomh_create_streamed(mcomh_pipeline **out, int device, void *hip_stream, const uint8_t *host_reads, size_t n, int L, const mcomh_params *pp)
{
	if (!out) return MCOM_E_ARG;
	*out = nullptr;
	if (n && !host_reads) return MCOM_E_ARG;
	static const uint8_t dummy = 0;
	int rc = mcomh_create(out, device, hip_stream, nullptr, n ? host_reads : &dummy, (size_t)L, n, L, pp);   // parameter resolution; no upload
	if (rc) return rc;
	(*out)->d_ascii = nullptr;
	(*out)->stream_host = host_reads;
	return MCOM_OK;
}

extern "C" int mcomh_create_dist(mcomh_pipeline **out, int device, void *hip_stream, mcomh_comm *comm, const uint8_t *host_reads,
                                 const uint8_t *d_reads, size_t pitch, size_t n_local, uint64_t rid0, uint64_t n_total, int L, const mcomh_params *pp)
{
	if (!out) return MCOM_E_ARG;
	*out = nullptr;
	if (!comm || n_total >= (1ull << 32) || rid0 + n_local > n_total) return MCOM_E_ARG;
	if (mcomh_comm_world(comm) > 256) { fprintf(stderr, "mcomh_create_dist: %d ranks; the Stage-2 index is shared out among at most 256 owners\n", mcomh_comm_world(comm)); return MCOM_E_ARG; }
	int rc = mcomh_create(out, device, hip_stream, host_reads, d_reads, pitch, n_local, L, pp);
	if (rc) return rc;
	P *p = *out;
	p->comm = comm; p->rank = mcomh_comm_rank(comm); p->world = mcomh_comm_world(comm);
	p->n_local = n_local; p->rid0 = rid0; p->n = (size_t)n_total;
	std::vector<uint8_t>().swap(p->h_ascii);                                // a rank holds only its shard: no stage dump here
	// the shards must be contiguous, in rank order, and cover [0, n_total): then "rank-major" is "rid ascending"
	const uint64_t mine[2] = {rid0, (uint64_t)n_local};
	std::vector<uint64_t> all;
	if ((rc = gather_host(p, mine, 2, all))) { mcomh_destroy(p); *out = nullptr; return rc; }
	p->shard_lo.assign((size_t)p->world + 1, 0);
	uint64_t at = 0; bool ok = true;
	for (int q = 0; q < p->world; ++q) { if (all[2 * q] != at) ok = false; p->shard_lo[q] = at; at += all[2 * q + 1]; }
	p->shard_lo[p->world] = at;
	if (!ok || at != n_total) { fprintf(stderr, "mcomh_create_dist: the shards are not contiguous in rank order or do not cover %llu reads\n", (unsigned long long)n_total); mcomh_destroy(p); *out = nullptr; return MCOM_E_ARG; }
	return MCOM_OK;
}

extern "C" int mcomh_create_packed(mcomh_pipeline **out, int device, void *hip_stream, const uint64_t *d_packed, size_t n, int L,
                                   const mcomh_params *pp)
{
	if (!out) return MCOM_E_ARG;
	*out = nullptr;
	if (n && !d_packed) return MCOM_E_ARG;
	static const uint8_t dummy = 0;
	// reuse the parameter resolution of mcomh_create with a placeholder device pointer, then switch the input
	int rc = mcomh_create(out, device, hip_stream, nullptr, n ? (const uint8_t*)d_packed : &dummy, (size_t)L, n, L, pp);
	if (rc) return rc;
	(*out)->d_ascii = nullptr;
	(*out)->ext_packed = d_packed;
	return MCOM_OK;
}

// File(s) -> pipeline (round 4).  A plain four-line FASTQ file is parsed by all cores at once, PACKED by the parser threads and sent
// straight into the pipeline's own row arrays: 2 bits per base + one N flag per base over PCIe (64 bytes per read at L = 150 instead
// of 150), no character matrix in HBM at all; classes, N counts and the majority-base substitution are made on the device
// (mcom_process_reads_packed).  Anything else (gzip, FASTA, sequences over several lines) goes through the sequential reader and the
// character matrix as before.  path2: the mates' file (bseq_read_second, preprocess.c:52-75: read n/2 + i is the mate of read i).
extern "C" int mcomh_create_from_fastq(mcomh_pipeline **out, int device, void *hip_stream, const char *path1, const char *path2, const mcomh_params *pp,
                                       char *err, size_t err_cap)
{
	auto fail = [&](int code, const char *msg) { if (err && err_cap) snprintf(err, err_cap, "%s", msg); return code; };
	if (!out || !path1) return MCOM_E_ARG;
	*out = nullptr;
	if (hipSetDevice(device) != hipSuccess) return fail(MCOM_E_HIP, "no usable GPU");
	{
		int L = 0;
		const double t_ix = now_ms();
		const size_t cap1 = mcom_fastq_stream_cap(path1, &L), cap2 = path2 && cap1 ? mcom_fastq_stream_cap(path2, &L) : 0;
		if (cap1 && (!path2 || cap2)) {
			static const uint8_t dummy = 0;
			int rc = mcomh_create(out, device, hip_stream, nullptr, &dummy, (size_t)L, 0, L, pp);    // parameter resolution; the number of reads follows
			if (rc) return fail(rc, "cannot create the pipeline");
			P *p = *out;
			p->d_ascii = nullptr;
			auto giveup = [&](int code, const char *msg) { mcomh_destroy(p); *out = nullptr; return fail(code, msg); };
			DevBuf<uint64_t> tp, tn;                                                 // the rows at their provisional places
			const size_t cap = cap1 + cap2;
			if (!tp.reserve(cap * p->W + 1) || !tn.reserve(cap * p->NW + 1)) return giveup(MCOM_E_NOMEM, "read buffers");
			McomFastqStream *s1 = nullptr, *s2 = nullptr;
			struct Drop { McomFastqStream *&a, *&b; ~Drop() { mcom_fastq_stream_free(a); mcom_fastq_stream_free(b); } } drop{s1, s2};
			int st = mcom_fastq_stream_packed(path1, L, device, p->copy_stream, tp.p, tn.p, cap1, &s1);
			if (st == 1 && path2) st = mcom_fastq_stream_packed(path2, L, device, p->copy_stream, tp.p + cap1 * p->W, tn.p + cap1 * p->NW, cap2, &s2);
			if (st < 0) return giveup(st, st == MCOM_E_NOMEM ? "out of memory" : "upload failed");
			if (st == 1) {
				const size_t n1 = mcom_fastq_stream_total(s1), n2 = path2 ? mcom_fastq_stream_total(s2) : 0;
				if (path2 && n1 != n2) { if (err && err_cap) snprintf(err, err_cap, "the two files hold %zu and %zu reads", n1, n2); mcomh_destroy(p); *out = nullptr; return MCOM_E_ARG; }
				const size_t n = n1 + n2;
				const double t_up = now_ms();
				p->n = n;
				if (!p->d_packed.reserve(n * p->W + 1) || !p->d_nmask.reserve(n * p->NW + 1)) return giveup(MCOM_E_NOMEM, "read buffers");
				// the gaps between the pieces close: one device-to-device copy per piece and array
				size_t row = 0;
				bool ok = true;
				for (int f = 0; f < (path2 ? 2 : 1) && ok; ++f) {
					const McomFastqStream *s = f ? s2 : s1;
					const size_t np = mcom_fastq_stream_pieces(s, nullptr, nullptr), base = f ? cap1 : 0;
					std::vector<size_t> prov(np), cnt(np);
					(void)mcom_fastq_stream_pieces(s, prov.data(), cnt.data());
					for (size_t q = 0; q < np && ok; ++q) {
						if (!cnt[q]) continue;
						ok = hipMemcpyAsync(p->d_packed.p + row * p->W, tp.p + (base + prov[q]) * p->W, cnt[q] * (size_t)p->W * 8, hipMemcpyDeviceToDevice, p->stream) == hipSuccess &&
						     hipMemcpyAsync(p->d_nmask.p + row * p->NW, tn.p + (base + prov[q]) * p->NW, cnt[q] * (size_t)p->NW * 8, hipMemcpyDeviceToDevice, p->stream) == hipSuccess;
						row += cnt[q];
					}
				}
				if (ok) ok = hipStreamSynchronize(p->stream) == hipSuccess;           // (the provisional arrays go back to the pool below)
				if (!ok || row != n) return giveup(MCOM_E_HIP, "upload failed");
				p->packed_in_place = true;
				extern double g_fastq_ms[3];
				p->stat["t_fastq_index"] = 0; p->stat["t_fastq_upload"] = t_up - t_ix; p->stat["t_fastq_close_gaps"] = now_ms() - t_up;
				p->stat["t_fastq_setup_max"] = g_fastq_ms[0]; p->stat["t_fastq_pack_max"] = g_fastq_ms[1]; p->stat["t_fastq_wait_max"] = g_fastq_ms[2];
				return MCOM_OK;
			}
			mcomh_destroy(p); *out = nullptr;                                        // not this layout after all, or a character outside ACGTN: the sequential reader words the message
		}
	}
	uint8_t *d = nullptr; size_t n = 0; int L = 0;
	int rc = path2 ? mcomh_fastq_pair_to_device(path1, path2, device, &L, 0, &d, &n, err, err_cap) : mcomh_fastq_to_device(path1, device, &L, 0, &d, &n, err, err_cap);
	if (rc) return rc;
	rc = mcomh_create(out, device, hip_stream, nullptr, d, (size_t)L, n, L, pp);
	if (rc) { mcomh_device_free(d); return fail(rc, "cannot create the pipeline"); }
	(*out)->d_ascii_adopted = d;                                                  // released with the pipeline
	return MCOM_OK;
}

// test hook (include/mcom_test.h): this rank fails, as if its local work had, right before its k-th flag exchange
extern "C" int mcomh_test_inject_failure(mcomh_pipeline *p, long k) { if (!p) return MCOM_E_ARG; p->inject_at = k; return MCOM_OK; }
extern "C" long mcomh_test_flag_exchanges(const mcomh_pipeline *p) { return p ? p->flag_exchanges : 0; }
extern "C" int mcomh_test_special_capacity(mcomh_pipeline *p, uint32_t first, uint32_t cap) { if (!p || !cap || !first) return MCOM_E_ARG; p->sp_first = first; p->sp_cap = cap; return MCOM_OK; }

extern "C" int mcomh_set_records(mcomh_pipeline *p, const uint64_t *d_x, const uint32_t *d_ylow)
{
	if (!p) return MCOM_E_ARG;
	if (!p->ext_packed && p->n) return p->fail(MCOM_E_ARG, "records can only be handed over with packed rows");
	if (p->n && (!d_x || !d_ylow)) return p->fail(MCOM_E_ARG, "null device pointer");
	p->ext_x = d_x; p->ext_ylow = d_ylow;
	return MCOM_OK;
}

extern "C" void mcomh_destroy(mcomh_pipeline *p)
{
	if (!p) return;
	p->join_sg();
	(void)hipStreamSynchronize(p->stream);
	if (p->d_ascii_adopted) { (void)hipFree(p->d_ascii_adopted); p->d_ascii_adopted = nullptr; }
	if (p->copy_stream) { (void)hipStreamSynchronize(p->copy_stream); (void)hipStreamDestroy(p->copy_stream); }
	if (p->ev_main) (void)hipEventDestroy(p->ev_main);
	if (p->ev_sg) (void)hipEventDestroy(p->ev_sg);
	if (p->ev_cls) (void)hipEventDestroy(p->ev_cls);
	if (p->ev_early) (void)hipEventDestroy(p->ev_early);
	if (p->ctx2) mcom_destroy(p->ctx2);
	if (p->ctx) mcom_destroy(p->ctx);
	delete p;
}

extern "C" void mcomh_pool_trim(void) { device_pool().trim(); mcom_pool_trim(); host_pool().trim(); }

extern "C" const char *mcomh_last_error(const mcomh_pipeline *p) { return p ? p->err.c_str() : "null pipeline"; }

// ----------------------------------------------------------------------------------------------------
// kt_for_reads                                                             kthread_reads.c:247, :40-230
// ----------------------------------------------------------------------------------------------------
static int kt_for_reads_impl(mcomh_pipeline *p);
extern "C" int mcomh_kt_for_reads(mcomh_pipeline *p) { return p ? stage_exit(p, kt_for_reads_impl(p)) : MCOM_E_ARG; }
static int kt_for_reads_impl(mcomh_pipeline *p)
{
	if (!p) return MCOM_E_ARG;
	const double t0 = busy_now(p);
	const size_t n = p->n;                                              // all reads of the job
	const bool dist = p->comm != nullptr;
	const size_t nl = dist ? p->n_local : n;                            // ... and those this rank classifies and sketches
	const uint64_t r0 = dist ? p->rid0 : 0;
	if (dist && (p->ext_packed || p->packed_in_place)) return p->fail(MCOM_E_ARG, "packed-row input is a single-GPU entry");
	if (!p->d_packed.reserve(n * p->W + 1) || !p->d_nmask.reserve(n * p->NW + 1) || !p->d_cls.reserve(n + 1) ||
	    !p->d_ncnt.reserve(nl + 1) || !p->d_rec.reserve(nl + 1)) return p->fail(MCOM_E_NOMEM, "read buffers");
	int rc;
	if (p->packed_in_place) {
		// the parser packed on the host (mcomh_create_from_fastq): codes and N flags are in place, the device does what is left of process_reads
		rc = p->gpu(mcom_process_reads_packed(p->ctx, p->d_packed.p, p->d_nmask.p, n, p->L, p->k, p->e, 0, p->d_packed.p, p->d_cls.p, p->d_ncnt.p, p->d_nmask.p, p->d_rec.p));
	} else if (p->ext_packed) {
		// packed rows handed over by the caller: every read is a kept read (class 0) without N
		if (n && ((rc = p->hipc(hipMemcpyAsync(p->d_packed.p, p->ext_packed, n * (size_t)p->W * 8, hipMemcpyDeviceToDevice, p->stream), "copy packed rows")) ||
		          (rc = p->hipc(hipMemsetAsync(p->d_cls.p, 0, n, p->stream), "clear")) ||
		          (rc = p->hipc(hipMemsetAsync(p->d_nmask.p, 0, n * (size_t)p->NW * 8, p->stream), "clear")))) return rc;
		if (p->ext_x && p->ext_ylow) rc = p->gpu(mcom_records_assemble(p->ctx, p->ext_x, p->ext_ylow, n, 0, p->d_rec.p));
		else rc = p->gpu(mcom_sketch_reads(p->ctx, p->d_packed.p, nullptr, n, p->L, p->k, 0, p->d_rec.p));
		if (!rc && p->ext_x) rc = p->sync("records");                         // the caller may release its arrays now
	} else if (p->stream_host) {
		// host-to-host: two staging blocks; the copy engine fills one while the kernels read the other
		const size_t CH = (size_t)4 << 20;
		DevBuf<uint8_t> stage[2];
		hipStream_t cs = nullptr; hipEvent_t up[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr};
		rc = MCOM_OK;
		if (!stage[0].reserve(std::min(CH, n) * (size_t)p->L + 16) || !stage[1].reserve(std::min(CH, n) * (size_t)p->L + 16)) rc = p->fail(MCOM_E_NOMEM, "staging blocks");
		if (!rc) rc = p->hipc(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking), "copy stream");
		for (int b = 0; b < 2 && !rc; ++b) { rc = p->hipc(hipEventCreateWithFlags(&up[b], hipEventDisableTiming), "event"); if (!rc) rc = p->hipc(hipEventCreateWithFlags(&done[b], hipEventDisableTiming), "event"); }
		size_t c = 0;
		for (size_t lo = 0; lo < n && !rc; lo += CH, ++c) {
			const int b = (int)(c & 1);
			const size_t cnt = std::min(CH, n - lo);
			if (c >= 2) rc = p->hipc(hipStreamWaitEvent(cs, done[b], 0), "wait");                    // the kernels of chunk c-2 have read this block
			if (!rc) rc = p->hipc(hipMemcpyAsync(stage[b].p, p->stream_host + lo * (size_t)p->L, cnt * (size_t)p->L, hipMemcpyHostToDevice, cs), "upload reads");
			if (!rc) rc = p->hipc(hipEventRecord(up[b], cs), "event");
			if (!rc) rc = p->hipc(hipStreamWaitEvent(p->stream, up[b], 0), "wait");
			if (!rc) rc = p->gpu(mcom_process_reads(p->ctx, stage[b].p, (size_t)p->L, cnt, p->L, p->k, p->e, (uint32_t)lo, p->d_packed.p + lo * p->W, p->d_cls.p + lo, p->d_ncnt.p + lo,
			                                        p->d_nmask.p + lo * p->NW, p->d_rec.p + lo));
			if (!rc) rc = p->hipc(hipEventRecord(done[b], p->stream), "event");
		}
		if (!rc) rc = p->sync("upload reads");
		if (cs) { (void)hipStreamSynchronize(cs); (void)hipStreamDestroy(cs); }
		for (int b = 0; b < 2; ++b) { if (up[b]) (void)hipEventDestroy(up[b]); if (done[b]) (void)hipEventDestroy(done[b]); }
		p->stat["h2d_chunks"] += (double)c;
	} else {
		// rows, classes and N masks are indexed by the global read id: a rank writes its shard's part of the whole arrays
		// Round 5, behind a switch (mcomh_params.read_batches = 1): classification + packing stream the reads at HBM's rate (5 ms per 100 M
		// reads), the sketch is bound by the VALU (13 ms): the reads go in a few batches over TWO streams, the classification of a batch
		// starting when that of the batch before has ended -- beside that batch's sketch.  Same arrays, batch by batch.  Measured on one
		// box, steps alternating (tools/ab_read_batches.py): 178.7 against 179.5 ms -- the kernels do run side by side (a batch's sketch takes
		// twice its time alone) but the card gets little more done per millisecond, so one launch each stays the default.
		const size_t NB = 4;
		size_t bsz = ((nl + NB - 1) / NB + 63) & ~(size_t)63;                          // (whole 64-read units: the byte-stream kernel's 16-byte alignment holds for every batch)
		const bool two = !dist && p->ctx2 && p->read_batches && nl >= ((size_t)1 << 22) && p->pitch == (size_t)p->L && (((uintptr_t)p->d_ascii) & 15) == 0 && nl < ((size_t)1 << 32);
		if (!two) {
			rc = p->gpu(mcom_process_reads(p->ctx, p->d_ascii, p->pitch, nl, p->L, p->k, p->e, (uint32_t)r0, p->d_packed.p + r0 * p->W, p->d_cls.p + r0, p->d_ncnt.p,
			                               p->d_nmask.p + r0 * p->NW, p->d_rec.p));
		} else {
			hipEvent_t evc[NB] = {}; hipEvent_t ev_end = nullptr;
			rc = MCOM_OK;
			for (size_t c = 0; c < NB && !rc; ++c) rc = p->hipc(hipEventCreateWithFlags(&evc[c], hipEventDisableTiming), "event");
			if (!rc) rc = p->hipc(hipEventCreateWithFlags(&ev_end, hipEventDisableTiming), "event");
			// (the copy stream may still carry the last step's copies of this pipeline: nothing of this step is on it yet)
			if (!rc) rc = p->hipc(hipEventRecord(p->ev_main, p->stream), "event");
			if (!rc) rc = p->hipc(hipStreamWaitEvent(p->copy_stream, p->ev_main, 0), "wait");   // the reads are where the main stream left them
			size_t c = 0;
			for (size_t lo = 0; lo < nl && !rc; lo += bsz, ++c) {
				const size_t cnt = std::min(bsz, nl - lo);
				mcom_ctx *cx = (c & 1) ? p->ctx2 : p->ctx;
				hipStream_t st = (c & 1) ? p->copy_stream : p->stream;
				if (c) rc = p->hipc(hipStreamWaitEvent(st, evc[c - 1], 0), "wait");              // one classification at a time
				if (!rc) { rc = mcom_classify_reads(cx, p->d_ascii + lo * p->pitch, p->pitch, cnt, p->L, p->e, p->d_packed.p + lo * p->W, p->d_cls.p + lo, p->d_ncnt.p + lo, p->d_nmask.p + lo * p->NW);
				           if (rc) p->err = std::string("libmcom_hip: ") + mcom_last_error(cx); }
				if (!rc) rc = p->hipc(hipEventRecord(evc[c], st), "event");
				if (!rc) { rc = mcom_sketch_classified(cx, p->d_packed.p + lo * p->W, p->d_cls.p + lo, cnt, p->L, p->k, (uint32_t)lo, p->d_rec.p + lo);
				           if (rc) p->err = std::string("libmcom_hip: ") + mcom_last_error(cx); }
			}
			if (!rc) rc = p->hipc(hipEventRecord(ev_end, p->copy_stream), "event");
			if (!rc) rc = p->hipc(hipStreamWaitEvent(p->stream, ev_end, 0), "wait");          // everything behind this sees all batches
			if (rc) { (void)hipStreamSynchronize(p->copy_stream); (void)hipStreamSynchronize(p->stream); }
			for (size_t q = 0; q < NB; ++q) if (evc[q]) (void)hipEventDestroy(evc[q]);
			if (ev_end) (void)hipEventDestroy(ev_end);
			p->stat["read_batches"] += (double)c;
		}
	}
	if (rc) return rc;
	if (dist) {
		// Every later stage reads rows by read id (group consensus, merged consensus, the Stage-2 singletons), and a group or a
		// contig holds reads of any shard: the packed rows (and the classes, 1 byte per read) are replicated once, here.
		// The N masks travel only when some read of the job holds an N.
		const int R = p->world;
		std::vector<uint64_t> first(R), cnt(R), fW(R), cW(R), fN(R), cN(R);
		for (int q = 0; q < R; ++q) { first[q] = p->shard_lo[q]; cnt[q] = p->shard_lo[q + 1] - p->shard_lo[q]; fW[q] = first[q] * p->W; cW[q] = cnt[q] * p->W; fN[q] = first[q] * p->NW; cN[q] = cnt[q] * p->NW; }
		const double tx = now_ms(), bx0 = xbytes(p);
		if ((rc = gatherv(p, p->d_packed.p, fW, cW)) || (rc = gatherv(p, p->d_cls.p, first, cnt, (const uint8_t*)nullptr, false))) return rc;
		uint32_t mx = 0;
		if ((rc = p->gpu(mcom_max_u16(p->ctx, p->d_ncnt.p, nl, &mx)))) return rc;
		uint64_t any = mx;
		if ((rc = allreduce_host(p, &any, 1, 2))) return rc;
		if (any) { if ((rc = gatherv(p, p->d_nmask.p, fN, cN))) return rc; }
		else {
			if (r0 && (rc = p->hipc(hipMemsetAsync(p->d_nmask.p, 0, r0 * p->NW * 8, p->stream), "clear"))) return rc;
			if (r0 + nl < n && (rc = p->hipc(hipMemsetAsync(p->d_nmask.p + (r0 + nl) * p->NW, 0, (n - r0 - nl) * p->NW * 8, p->stream), "clear"))) return rc;
		}
		p->stat["t_x_reads"] += now_ms() - tx; p->stat["b_x_reads"] += xbytes(p) - bx0;
	}
	// The special reads (another class than 0: a handful per million on real data) are listed on the device and the LIST travels,
	// on the copy stream, while the bucket stage starts; a thread sorts it into the seven id lists (rid order) -- nobody reads those
	// before Stage 2.  Rounds 1-3 sent the class array itself: 100 MB of PCIe per 100 M reads beside the first sort pass, which ran
	// 1.8 instead of 0.4 ms under it.  A list that outgrows its room (a file of N reads) falls back to the array.
	p->join_cls();
	p->cls_failed = false;
	p->h_cls_valid = false;
	for (std::vector<uint32_t> *v : {&p->allA, &p->allT, &p->allN, &p->fpA, &p->fpT, &p->fpN, &p->Nfile}) v->clear();   // a second call must not append twice
	if (!p->ev_cls && (rc = p->hipc(hipEventCreateWithFlags(&p->ev_cls, hipEventDisableTiming), "event"))) return rc;
	const uint32_t SP_CAP = p->sp_cap, SP_FIRST = std::min(p->sp_first, p->sp_cap);
	if (!p->d_special.reserve((size_t)SP_CAP + 1) || !p->h_special.resize((size_t)SP_FIRST + 1)) return p->fail(MCOM_E_NOMEM, "special reads");
	uint32_t *d_count = (uint32_t*)(p->d_special.p + SP_CAP);
	if ((rc = p->gpu(mcom_special_reads(p->ctx, p->d_cls.p, n, p->d_special.p, SP_CAP, d_count)))) return rc;
	if ((rc = p->hipc(hipEventRecord(p->ev_main, p->stream), "event")) || (rc = p->hipc(hipStreamWaitEvent(p->copy_stream, p->ev_main, 0), "wait")) ||
	    (rc = p->hipc(hipMemcpyAsync(p->h_special.data() + SP_FIRST, d_count, 8, hipMemcpyDeviceToHost, p->copy_stream), "copy special reads")) ||
	    (rc = p->hipc(hipMemcpyAsync(p->h_special.data(), p->d_special.p, (size_t)SP_FIRST * 8, hipMemcpyDeviceToHost, p->copy_stream), "copy special reads")) ||
	    (rc = p->hipc(hipEventRecord(p->ev_cls, p->copy_stream), "event"))) return rc;
	p->cls_thread = std::thread([p, n, SP_CAP, SP_FIRST]() {
		if (hipSetDevice(p->device) != hipSuccess || hipEventSynchronize(p->ev_cls) != hipSuccess) { p->cls_failed = true; return; }
		const uint32_t count = (uint32_t)p->h_special[SP_FIRST];
		auto put = [p](size_t r, uint32_t c) {
			switch (c) {
			case MCOM_CLS_ALLA: p->allA.push_back((uint32_t)r); break;
			case MCOM_CLS_ALLT: p->allT.push_back((uint32_t)r); break;
			case MCOM_CLS_ALLN: p->allN.push_back((uint32_t)r); break;
			case MCOM_CLS_NEARA: p->fpA.push_back((uint32_t)r); break;
			case MCOM_CLS_NEART: p->fpT.push_back((uint32_t)r); break;
			case MCOM_CLS_NEARN: p->fpN.push_back((uint32_t)r); break;
			case MCOM_CLS_NHEAVY: p->Nfile.push_back((uint32_t)r); break;
			default: break;
			}
		};
		if (count <= SP_CAP) {
			std::vector<uint64_t> all;
			const uint64_t *lst = p->h_special.data();
			if (count > SP_FIRST) {                                                // (the copy stream is this thread's until the event of the next call)
				all.resize(count);
				if (hipMemcpyAsync(all.data(), p->d_special.p, (size_t)count * 8, hipMemcpyDeviceToHost, p->copy_stream) != hipSuccess ||
				    hipStreamSynchronize(p->copy_stream) != hipSuccess) { p->cls_failed = true; return; }
				lst = all.data();
			}
			std::vector<uint64_t> srt(lst, lst + count);
			std::sort(srt.begin(), srt.end());                                      // rid order = file order
			for (uint64_t e : srt) put((size_t)(e >> 8), (uint32_t)e & 255u);
			return;
		}
		// more special reads than the list holds: the class array itself
		if (!p->h_cls.resize(n + 8)) { p->cls_failed = true; return; }
		memset(p->h_cls.data() + n, 0, 8);
		if (hipMemcpyAsync(p->h_cls.data(), p->d_cls.p, n, hipMemcpyDeviceToHost, p->copy_stream) != hipSuccess || hipStreamSynchronize(p->copy_stream) != hipSuccess) { p->cls_failed = true; return; }
		p->h_cls_valid = true;
		for (size_t r = 0; r < n; ++r) {
			if ((r & 7) == 0) {                                                   // most reads are class 0: skip eight at a time
				uint64_t w8; memcpy(&w8, p->h_cls.data() + r, 8);
				if (w8 == 0) { r += 7; continue; }
			}
			put(r, p->h_cls[r]);
		}
	});
	p->stat["t_reads"] += busy_now(p) - t0;
	return MCOM_OK;
}

// ----------------------------------------------------------------------------------------------------
// kt_for_bucket: Stage-1 rounds                                               kthread_bucket.c:562-629
// ----------------------------------------------------------------------------------------------------
static int kt_for_bucket_impl(mcomh_pipeline *p);
extern "C" int mcomh_kt_for_bucket(mcomh_pipeline *p) { return p ? stage_exit(p, kt_for_bucket_impl(p)) : MCOM_E_ARG; }
static int kt_for_bucket_impl(mcomh_pipeline *p)
{
	if (!p) return MCOM_E_ARG;
	const double t0 = busy_now(p);
	const int L = p->L;
	const int RS = (2 * L + 15) & ~15;                       // stride of one group's consensus on the device
	// Multi-GPU: the records of a round go to the owner of their bucket (bucket ranges in rank order); a rank then does what
	// the single GPU does, on its buckets, and the new contigs of all ranks are all-gathered into the replicated set in
	// rank order -- which is the reference's visiting order (buckets ascending, kthread_bucket.c:531-560).
	const bool dist = p->comm != nullptr;
	const int R = p->world, me = p->rank;
	size_t n_cur = dist ? p->n_local : p->n;
	DevBuf<mcom_mm128> d_cur, d_sorted, d_part, d_recv; DevBuf<uint32_t> d_singles_ab[2], d_sord_ab[2], d_goff, d_rids, d_nkept; DevBuf<uint64_t> d_members;
	// the singles of a round (tens of MB in round 1) go to the host on the copy stream while the next round runs: two sets of
	// buffers, a set is written again only after its copy has finished
	hipEvent_t ev_set[2] = {nullptr, nullptr}; bool set_busy[2] = {false, false};
	struct EvGuard { hipEvent_t *e; ~EvGuard() { for (int q = 0; q < 2; ++q) if (e[q]) (void)hipEventDestroy(e[q]); } } ev_guard{ev_set};
	for (int q = 0; q < 2; ++q) if (hipEventCreateWithFlags(&ev_set[q], hipEventDisableTiming) != hipSuccess) return p->fail(MCOM_E_HIP, "event");
	DevBuf<uint8_t> d_keep, d_refs; DevBuf<uint16_t> d_sv, d_reflen;
	const mcom_mm128 *cur = p->d_rec.p;                      // round 1 works on the records of kt_for_reads
	std::vector<uint32_t> resk;
	struct SgRound { PinVec<uint32_t> singles, sord, rej, rejg; size_t ns = 0, nrej = 0; bool last = false; };
	std::vector<SgRound> sg_rounds;
	if (p->sg_thread.joinable()) p->sg_thread.join();        // (not join_sg(): the class-list thread of kt_for_reads is still waiting for its copy)
	if (dist) p->sg.clear();
	size_t n_sg_total = p->sg.size();
	DevBuf<uint32_t> d_rej, d_rejg;
	// the contigs are built on the device (p->dC) and stay there for combine_cluster; the host copy is made on demand
	p->C.clear();
	p->dC.n = 0; p->dC.chars = 0; p->dC.members = 0; p->dC.nrec = 0;
	p->dC_valid = true; p->hostC_valid = false; p->host_off_valid = false; p->cbits_for_dC = false;
	p->prepacked = p->comm != nullptr; p->prepacked_words = 0;
	DevBuf<uint64_t> cw_l, bits_l, cw_all; DevBuf<uint32_t> clen_l, clen_all;
	DevSet &D = p->dC;
	int rc;
	if (dist) {                                              // offset entry 0 of the replicated set: nobody's contig writes it
		if (!D.soff.reserve(2) || !D.moff.reserve(2)) return p->fail(MCOM_E_NOMEM, "contig set");
		if ((rc = p->hipc(hipMemsetAsync(D.soff.p, 0, 8, p->stream), "clear")) || (rc = p->hipc(hipMemsetAsync(D.moff.p, 0, 8, p->stream), "clear"))) return rc;
	}
	int last_rounds = 0; long pre = 0;
	for (int r = 1;; ++r) {
		if (p->k - r <= 9) ++last_rounds;                                           // :584-585
		if (r == p->max_rounds - 1) ++last_rounds;
		const bool last = last_rounds != 0;
		const int kmer_in = p->k - (r - 1);                  // k the incoming records were sketched with
		const int kmer_next = p->k - r;                       // k for the rejects of this round (:592)
		resk.clear();
		if (dist) {
			// the exchange of the round: round 1 moves every kept read's record, later rounds the re-sketched rejects
			// (kthread_bucket.c:205-212, :489-496 push them into the other bucket set)
			const double tx = now_ms(), bx0 = xbytes(p);
			std::vector<uint64_t> cnt(R, 0), all;
			if (!d_part.reserve(n_cur + 1)) return p->fail(MCOM_E_NOMEM, "exchange buffers");
			if ((rc = p->gpu(mcom_partition_by_owner(p->ctx, cur, n_cur, NB_BITS, R, d_part.p, cnt.data()))) || (rc = gather_host(p, cnt.data(), R, all))) return rc;
			std::vector<uint64_t> so(R), sb(R), ro(R), rb(R);
			uint64_t a = 0, b = 0;
			for (int q = 0; q < R; ++q) { so[q] = a * 16; sb[q] = cnt[q] * 16; a += cnt[q]; ro[q] = b * 16; rb[q] = all[(size_t)q * R + me] * 16; b += all[(size_t)q * R + me]; }
			if (!d_recv.reserve(b + 1)) return p->fail(MCOM_E_NOMEM, "exchange buffers");
			if ((rc = alltoallv_dev(p, d_part.p, so.data(), sb.data(), d_recv.p, ro.data(), rb.data()))) return rc;
			// what arrives is ordered by sender, each part by rid.  Round 1: shards are rid ranges in rank order, so this is
			// rid order already; later rounds: one stable sort by rid restores what mcom_sort_group's tie rule expects
			if (r > 1 && (rc = p->gpu(mcom_sort_by_rid(p->ctx, d_recv.p, (size_t)b)))) return rc;
			cur = d_recv.p; n_cur = (size_t)b;
			p->stat["x_records"] += (double)b;
			p->stat["t_x_records"] += now_ms() - tx; p->stat["b_x_records"] += xbytes(p) - bx0;
		}
		size_t ns = 0, ng = 0, nm = 0, nrej = 0;
		const double tg = busy_now(p);
		if (r == 1) p->stat["t_bk_pre"] += tg - t0;
		const int set = r & 1;
		DevBuf<uint32_t> &d_singles = d_singles_ab[set], &d_sord = d_sord_ab[set];
		if (set_busy[set]) { if ((rc = p->hipc(hipStreamWaitEvent(p->stream, ev_set[set], 0), "wait"))) return rc; set_busy[set] = false; }
		if (n_cur) {
			if (!d_sorted.reserve(n_cur) || !d_singles.reserve(n_cur) || !d_sord.reserve(n_cur) || !d_members.reserve(n_cur) || !d_goff.reserve(n_cur / 2 + 2))
				return p->fail(MCOM_E_NOMEM, "round buffers");
			uint64_t cnts[4];
			if ((rc = p->gpu(mcom_sort_group(p->ctx, cur, n_cur, L, p->k, kmer_in, NB_BITS, d_sorted.p, d_singles.p, d_sord.p, d_members.p, d_goff.p, cnts)))) return rc;
			ns = cnts[1]; ng = cnts[2]; nm = cnts[3];
			p->stat["t_bk_sort"] += busy_now(p) - tg;
			if (!d_keep.reserve(nm + 1) || !d_nkept.reserve(ng + 1) || !d_sv.reserve(ng + 1) || !d_reflen.reserve(ng + 1) || !d_refs.reserve(ng * (size_t)RS + 16))
				return p->fail(MCOM_E_NOMEM, "consensus buffers");
			// construct_ref of every group on the device (:446)
			if ((rc = p->gpu(mcom_group_consensus(p->ctx, p->d_packed.p, d_members.p, d_goff.p, (uint32_t)ng, L, p->k, p->e, d_keep.p, d_nkept.p, d_sv.p, d_reflen.p, d_refs.p, RS)))) return rc;
			if (!d_rej.reserve(nm + 1) || !d_rejg.reserve(nm + 1)) return p->fail(MCOM_E_NOMEM, "reject buffers");
		}
		// groups that stay contigs (more than one member kept, :451) join the device-resident contig set; the others'
		// members come back as rejects, in visiting order
		uint64_t gc[4] = {0, 0, 0, 0};
		if (!dist) {
			if (n_cur) {
				for (int attempt = 0; attempt < 2; ++attempt) {
					rc = mcom_groups_to_contigs(p->ctx, d_members.p, d_goff.p, ng, d_keep.p, d_nkept.p, d_sv.p, d_reflen.p, d_refs.p, RS, D.n, D.chars, D.members,
					                            D.seq.p, D.seq.cap, D.soff.p, D.mem.p, D.mem.cap, D.moff.p, std::min(D.soff.cap, D.moff.cap), d_rej.p, d_rejg.p, d_rej.cap, gc);
					if (rc != MCOM_E_OVERFLOW) break;
					// (round 1 makes nearly all the contigs: room for what combine_cluster will append behind them -- its merge rounds leave the
					// set where it is and add the merged contigs: about 1.5 x the strings and, since nearly every member sits in a contig that
					// merges, 3 x the member lists in all -- so that no round has to move the store)
					const uint64_t slack_c = r == 1 ? (D.chars + gc[1]) * 13 / 10 : 0, slack_m = r == 1 ? (D.members + gc[2]) * 7 / 2 : 0, slack_n = r == 1 ? (D.n + gc[0]) * 13 / 10 : 0;
					if (!D.seq.grow(D.chars + gc[1] + slack_c + 16, D.chars, p->stream) || !D.mem.grow(D.members + gc[2] + slack_m + 1, D.members, p->stream) ||
					    !D.soff.grow(D.n + gc[0] + slack_n + 2, D.n + 1, p->stream) || !D.moff.grow(D.n + gc[0] + slack_n + 2, D.n + 1, p->stream)) return p->fail(MCOM_E_NOMEM, "contig set");
				}
				if (rc) return p->gpu(rc);
				D.n += gc[0]; D.chars += gc[1]; D.members += gc[2];
			}
			nrej = gc[3];
			n_sg_total += ns + (last ? nrej : 0);
		} else {
			// sizes first (the call reports them with MCOM_E_OVERFLOW when given no room), so that every rank knows where its
			// contigs go in the replicated set: behind those of the lower ranks of this round
			if (ng) {
				rc = mcom_groups_to_contigs(p->ctx, d_members.p, d_goff.p, ng, d_keep.p, d_nkept.p, d_sv.p, d_reflen.p, d_refs.p, RS, 0, 0, 0,
				                            nullptr, 0, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr, 0, gc);
				if (rc != MCOM_E_OVERFLOW && rc != MCOM_OK) return p->gpu(rc);
			}
			const uint64_t mine[5] = {gc[0], gc[1], gc[2], (uint64_t)ns, gc[3]};
			std::vector<uint64_t> all;
			if ((rc = gather_host(p, mine, 5, all))) return rc;
			std::vector<uint64_t> fn(R), cn(R), fc(R), cc(R), fm(R), cm(R);
			uint64_t tn = 0, tc = 0, tm = 0;
			for (int q = 0; q < R; ++q) {
				fn[q] = D.n + 1 + tn; cn[q] = all[5 * q]; fc[q] = D.chars + tc; cc[q] = all[5 * q + 1]; fm[q] = D.members + tm; cm[q] = all[5 * q + 2];
				tn += cn[q]; tc += cc[q]; tm += cm[q];
				n_sg_total += all[5 * q + 3] + (last ? all[5 * q + 4] : 0);
			}
			const uint64_t slack_c = r == 1 ? (D.chars + tc) * 13 / 10 : 0, slack_m = r == 1 ? (D.members + tm) * 7 / 2 : 0, slack_n = r == 1 ? (D.n + tn) * 13 / 10 : 0;   // (see the single-GPU branch)
			if (!D.seq.grow(D.chars + tc + slack_c + 16, D.chars, p->stream) || !D.mem.grow(D.members + tm + slack_m + 1, D.members, p->stream) ||
			    !D.soff.grow(D.n + tn + slack_n + 2, D.n + 1, p->stream) || !D.moff.grow(D.n + tn + slack_n + 2, D.n + 1, p->stream)) return p->fail(MCOM_E_NOMEM, "contig set");
			if (ng) {
				uint64_t gc2[4];
				if ((rc = p->gpu(mcom_groups_to_contigs(p->ctx, d_members.p, d_goff.p, ng, d_keep.p, d_nkept.p, d_sv.p, d_reflen.p, d_refs.p, RS, fn[me] - 1, fc[me], fm[me],
				                                        D.seq.p, D.seq.cap, D.soff.p, D.mem.p, D.mem.cap, D.moff.p, std::min(D.soff.cap, D.moff.cap), d_rej.p, d_rejg.p, d_rej.cap, gc2)))) return rc;
				if (gc2[0] != gc[0] || gc2[1] != gc[1] || gc2[2] != gc[2] || gc2[3] != gc[3]) return p->fail(MCOM_E_ARG, "contig counts changed between the two calls");
			}
			// The strings do not travel: the PACKED WORDS of this rank's new contigs do (a quarter of a byte per base), straight to their place in
			// d_cbits -- a contig owns whole words, so the rounds' shares one behind the other are the layout combine_cluster makes of the set --
			// and every rank unpacks the strings the others built (mcom_unpack_contigs).
			uint64_t twl = 0;
			if (cn[me]) {
				const uint64_t start = fc[me];
				if ((rc = p->h2d(D.soff.p + fn[me] - 1, &start, 1, "upload"))) return rc;   // (the end of the rank below: it arrives with the gather, the layout needs it now; mcom_contig_layout waits for the stream)
				if (!cw_l.reserve(cn[me] + 2) || !clen_l.reserve(cn[me] + 2)) return p->fail(MCOM_E_NOMEM, "contig set");
				if ((rc = p->gpu(mcom_contig_layout(p->ctx, D.soff.p + fn[me] - 1, cn[me], cw_l.p, clen_l.p, &twl)))) return rc;
				if (!bits_l.reserve(twl + 2)) return p->fail(MCOM_E_NOMEM, "contig set");
				if ((rc = p->gpu(mcom_pack_contigs(p->ctx, D.seq.p, D.soff.p + fn[me] - 1, cw_l.p, (uint32_t)cn[me], twl, bits_l.p)))) return rc;
			}
			std::vector<uint64_t> allw, fw(R), cwq(R);
			if ((rc = gather_host(p, &twl, 1, allw))) return rc;
			uint64_t tw_all = 0;
			for (int q = 0; q < R; ++q) { fw[q] = p->prepacked_words + tw_all; cwq[q] = allw[q]; tw_all += cwq[q]; }
			// (the first round asks for what combine_cluster will want for its merge rounds, 2.4 x the set: one allocation instead of two and a copy)
			if (!p->d_cbits.grow((size_t)((p->prepacked_words + tw_all) * (r == 1 ? 24 : 10) / 10 + 2), (size_t)p->prepacked_words, p->stream)) return p->fail(MCOM_E_NOMEM, "packed contigs");
			const double tx = now_ms(), bx0 = xbytes(p);
			if ((rc = gatherv(p, p->d_cbits.p, fw, cwq, (const uint64_t*)bits_l.p)) || (rc = gatherv(p, D.soff.p, fn, cn, (const uint64_t*)nullptr, false)) || (rc = gatherv(p, D.mem.p, fm, cm, (const uint64_t*)nullptr, false)) || (rc = gatherv(p, D.moff.p, fn, cn, (const uint64_t*)nullptr, false))) return rc;
			p->stat["t_x_contigs"] += now_ms() - tx; p->stat["b_x_contigs"] += xbytes(p) - bx0;
			if (tn) {
				uint64_t tw2 = 0;
				if (!cw_all.reserve(tn + 2) || !clen_all.reserve(tn + 2)) return p->fail(MCOM_E_NOMEM, "contig set");
				if ((rc = p->gpu(mcom_contig_layout(p->ctx, D.soff.p + D.n, tn, cw_all.p, clen_all.p, &tw2)))) return rc;
				if (tw2 != tw_all) return p->fail(MCOM_E_ARG, "new contigs: %llu packed words gathered, the layout has %llu", (unsigned long long)tw_all, (unsigned long long)tw2);
				// (this rank's own strings are where mcom_groups_to_contigs wrote them: the contigs in front of them and behind them are unpacked)
				const uint64_t i0 = fn[me] - 1 - D.n, i1 = i0 + cn[me];
				if (i0 && (rc = p->gpu(mcom_unpack_contigs(p->ctx, p->d_cbits.p + p->prepacked_words, cw_all.p, D.soff.p + D.n, (uint32_t)i0, D.chars, fc[me], D.seq.p)))) return rc;
				if (tn > i1 && (rc = p->gpu(mcom_unpack_contigs(p->ctx, p->d_cbits.p + p->prepacked_words, cw_all.p + i1, D.soff.p + D.n + i1, (uint32_t)(tn - i1), fc[me] + cc[me], D.chars + tc, D.seq.p)))) return rc;
				p->prepacked_words += tw_all;
			}
			D.n += tn; D.chars += tc; D.members += tm;
			nrej = gc[3];
		}
		if (n_cur || dist) {
			SgRound Rd; Rd.ns = ns; Rd.nrej = nrej; Rd.last = last;
			if (!Rd.singles.resize(ns) || !Rd.sord.resize(ns) || !Rd.rej.resize(nrej) || !Rd.rejg.resize(nrej)) return p->fail(MCOM_E_NOMEM, "round lists");
			// the rejects first (the next round waits for them), then the singles on the copy stream: started the other way round, the
			// few KB of rejects queued behind 2 x 64 MB of singles on the way to the host (0.7 ms of the main stream per round)
			if ((rc = p->d2h(Rd.rej.data(), d_rej.p, nrej, "copy")) || (rc = p->d2h(Rd.rejg.data(), d_rejg.p, nrej, "copy")) || (rc = p->sync("round copy"))) return rc;
			if (ns) {
				if ((rc = p->hipc(hipMemcpyAsync(Rd.singles.data(), d_singles.p, ns * 4, hipMemcpyDeviceToHost, p->copy_stream), "copy")) ||
				    (rc = p->hipc(hipMemcpyAsync(Rd.sord.data(), d_sord.p, ns * 4, hipMemcpyDeviceToHost, p->copy_stream), "copy")) ||
				    (rc = p->hipc(hipEventRecord(ev_set[set], p->copy_stream), "event"))) return rc;
				set_busy[set] = true;
			}
			p->stat["t_gpu"] += busy_now(p) - tg;
			p->stat["t_bk_gpu"] += busy_now(p) - tg;
			// the next round only needs the rejects; where singles and rejects go in the singleton list is settled later
			if (!last) resk.assign(Rd.rej.data(), Rd.rej.data() + nrej);
			sg_rounds.push_back(std::move(Rd));
		}
		p->stat["rounds"] += 1;
		if (last_rounds) ++last_rounds;                                             // :594
		const long cr = (long)p->dC.members;
		if (cr - pre < 100) ++last_rounds;                                          // :614-618
		pre = cr;
		if (last_rounds > 1) break;
		// rejected reads are sketched again with a shorter k (:205, :489); ascending rid keeps the sort's tie rule
		std::sort(resk.begin(), resk.end());
		n_cur = resk.size();
		p->stat["resketch"] += (double)n_cur;
		if (n_cur) {
			if (!d_rids.reserve(n_cur) || !d_cur.reserve(n_cur)) return p->fail(MCOM_E_NOMEM, "re-sketch buffers");
			if ((rc = p->h2d(d_rids.p, resk.data(), n_cur, "upload rids"))) return rc;
			if ((rc = p->gpu(mcom_sketch_reads(p->ctx, p->d_packed.p, d_rids.p, n_cur, L, kmer_next, 0, d_cur.p)))) return rc;
			if ((rc = p->sync("re-sketch"))) return rc;
		}
		cur = d_cur.p;
	}
	// singletons and rejects in the reference's visiting order (process_bucket, :398-505): a single whose ordinal is g was
	// visited before group g.  Nobody needs the list before Stage 2: a thread builds it beside combine_cluster.
	// (multi-GPU: this rank's part, round by round; combine_cluster puts the ranks' parts together, dist_gather_sg)
	p->join_sg();
	{
		const double tw = now_ms();
		if ((rc = p->hipc(hipStreamSynchronize(p->copy_stream), "round lists"))) return rc;
		p->stat["t_bk_lists_wait"] += now_ms() - tw;
	}
	p->sg_round_len.assign(sg_rounds.size(), 0);
	p->sg_gathered = !dist;
	p->sg_thread = std::thread([p, n_sg_total](std::vector<SgRound> rounds) {
		p->sg.reserve(n_sg_total);
		size_t ri = 0;
		for (const SgRound &Rd : rounds) {
			const size_t before = p->sg.size();
			size_t si = 0;
			for (size_t u = 0; u < Rd.nrej; ++u) {
				size_t sj = si;
				while (sj < Rd.ns && Rd.sord[sj] <= Rd.rejg[u]) ++sj;                   // groups of one (:402-413)
				p->sg.insert(p->sg.end(), Rd.singles.data() + si, Rd.singles.data() + sj);
				si = sj;
				if (Rd.last) p->sg.push_back(Rd.rej[u]);
			}
			p->sg.insert(p->sg.end(), Rd.singles.data() + si, Rd.singles.data() + Rd.ns);
			p->sg_round_len[ri++] = p->sg.size() - before;
		}
		// a page-locked copy, so that the list goes up to the device beside the Stage-2 set-up instead of in front of the first pass
		if (p->sg_pin.resize(p->sg.size()) && !p->sg.empty()) memcpy(p->sg_pin.data(), p->sg.data(), p->sg.size() * 4);
	}, std::move(sg_rounds));
	if (n_sg_total <= 5000000) p->maxsearch = 2000;                                 // preprocess.c:169-172
	if (p->maxsearch_forced > 0) p->maxsearch = p->maxsearch_forced;
	p->stat["n_sg0"] = (double)n_sg_total;
	p->stat["contigs_bucket"] = (double)p->dC.n;
	p->stat["t_bucket"] += busy_now(p) - t0;
	return MCOM_OK;
}

// Multi-GPU: the singleton list of the job = round by round, rank by rank (the reference visits a round's buckets in
// ascending order, and the ranks own ascending bucket ranges), each part as the rank's own thread ordered it.
static int dist_gather_sg(P *p)
{
	if (!p->comm || p->sg_gathered) return MCOM_OK;
	const int R = p->world, me = p->rank;
	const size_t nr = p->sg_round_len.size();
	int rc;
	uint64_t mn = nr, mx = nr;
	if ((rc = allreduce_host(p, &mn, 1, 1)) || (rc = allreduce_host(p, &mx, 1, 2))) return rc;
	if (mn != mx) return p->fail(MCOM_E_ARG, "the ranks ran different numbers of bucket rounds");
	std::vector<uint64_t> all;
	if ((rc = gather_host(p, p->sg_round_len.data(), nr, all))) return rc;
	std::vector<uint64_t> first(R), tot(R, 0);
	uint64_t total = 0;
	for (int q = 0; q < R; ++q) { for (size_t r = 0; r < nr; ++r) tot[q] += all[(size_t)q * nr + r]; first[q] = total; total += tot[q]; }
	if (tot[me] != p->sg.size() || p->sg_pin.size() != p->sg.size()) return p->fail(MCOM_E_ARG, "singleton list: %zu entries, the rounds say %llu", p->sg.size(), (unsigned long long)tot[me]);
	// The parts meet on the device (this rank's goes up from its page-locked copy), are put into visiting order there -- one
	// device-to-device copy per (round, rank) -- and the list comes back once, into page-locked memory: Stage 2 finds it in HBM
	// already, the host keeps its copy for the accessors.
	DevBuf<uint32_t> d_flat;
	if (!d_flat.reserve(total + 1) || !p->d_sg_live.reserve(total + 1)) return p->fail(MCOM_E_NOMEM, "singleton list");
	if ((rc = p->h2d(d_flat.p + first[me], p->sg_pin.data(), p->sg.size(), "upload singletons")) || (rc = gatherv(p, d_flat.p, first, tot))) return rc;
	std::vector<uint64_t> cursor(first);
	uint64_t at = 0;
	for (size_t r = 0; r < nr; ++r)
		for (int q = 0; q < R; ++q) {
			const uint64_t len = all[(size_t)q * nr + r];
			if (len && (rc = p->hipc(hipMemcpyAsync(p->d_sg_live.p + at, d_flat.p + cursor[q], len * 4, hipMemcpyDeviceToDevice, p->stream), "order singletons"))) return rc;
			cursor[q] += len; at += len;
		}
	if (!p->sg_pin.resize(total)) return p->fail(MCOM_E_NOMEM, "singleton list");
	if ((rc = p->d2h(p->sg_pin.data(), p->d_sg_live.p, total, "copy singletons")) || (rc = p->sync("copy singletons"))) return rc;
	p->sg.assign(p->sg_pin.data(), p->sg_pin.data() + total);
	p->n_sg_live = total; p->sg_live_valid = total != 0;
	p->sg_gathered = true;
	return MCOM_OK;
}

// ----------------------------------------------------------------------------------------------------
// contigs on the device: ASCII concat + char offsets, packed words + word offsets, lengths
// ----------------------------------------------------------------------------------------------------
static int upload_contigs(P *p, const ContigSet &C)
{
	const size_t n = C.n();
	p->h_coff_words.assign(n + 1, 0);
	std::vector<uint32_t> len(n);
	for (size_t i = 0; i < n; ++i) { len[i] = (uint32_t)C.rsize(i); p->h_coff_words[i + 1] = p->h_coff_words[i] + (2 * C.rsize(i) + 63) / 64 + 1; }
	p->total_words = p->h_coff_words[n];
	if (!p->d_cseq.reserve(C.ref.size() + 16) || !p->d_coff_chars.reserve(n + 1) || !p->d_coff_words.reserve(n + 1) || !p->d_clen.reserve(n + 1) ||
	    !p->d_cbits.reserve(p->total_words + 2)) return p->fail(MCOM_E_NOMEM, "contig buffers");
	int rc;
	if ((rc = p->h2d(p->d_cseq.p, (const uint8_t*)C.ref.data(), C.ref.size(), "upload contigs")) ||
	    (rc = p->h2d(p->d_coff_chars.p, C.roff.data(), n + 1, "upload offsets")) ||
	    (rc = p->h2d(p->d_coff_words.p, p->h_coff_words.data(), n + 1, "upload offsets")) ||
	    (rc = p->h2d(p->d_clen.p, len.data(), n, "upload lengths"))) return rc;
	if (n) {
		if ((rc = p->hipc(hipMemsetAsync(p->d_cbits.p, 0, (p->total_words + 2) * 8, p->stream), "clear"))) return rc;
		if ((rc = p->gpu(mcom_pack_contigs(p->ctx, p->d_cseq.p, p->d_coff_chars.p, p->d_coff_words.p, (uint32_t)n, p->total_words, p->d_cbits.p)))) return rc;
	}
	return p->sync("upload contigs");            // `len` must outlive the async copy
}

// mm_sketch_lh_ori of every uploaded contig (all minimizers)
static int sketch_contigs(P *p, size_t n, size_t total_chars, DevBuf<uint32_t> &moff, DevBuf<mcom_mm128> &out, uint64_t &total)
{
	total = 0;
	p->stat["sketch_bases"] += (double)total_chars;                // one launch per call
	if (!moff.reserve(n + 2)) return p->fail(MCOM_E_NOMEM, "minimizer offsets");
	size_t cap = std::max<size_t>(1024, total_chars / 8 + n);
	for (int attempt = 0; attempt < 2; ++attempt) {
		if (!out.reserve(cap)) return p->fail(MCOM_E_NOMEM, "minimizer records");
		int rc = mcom_sketch_contigs(p->ctx, p->d_cseq.p, p->d_coff_chars.p, nullptr, n, p->rw, p->k, 0, moff.p, out.p, out.cap, &total);
		if (rc == MCOM_E_OVERFLOW) { cap = total; continue; }
		return p->gpu(rc);
	}
	return p->fail(MCOM_E_OVERFLOW, "minimizer buffer");
}


// mm_sketch_lh_ori of contigs [0, n_first) of S into S.rec[0 ..) / S.roff[0 .. n_first]; room records stay free behind them
static int sketch_first(P *p, DevSet &S, size_t n_first, uint64_t chars_first, size_t room, uint64_t &total)
{
	total = 0;
	p->stat["sketch_bases"] += (double)chars_first;                // one launch per call
	if (!S.roff.reserve(std::max(S.n + 2, S.soff.cap))) return p->fail(MCOM_E_NOMEM, "minimizer offsets");   // (room for the contigs the merge rounds append, as the other offset arrays)
	size_t cap = std::max<size_t>(1024, chars_first / 8 + n_first);
	for (int attempt = 0; attempt < 2; ++attempt) {
		if (!S.rec.reserve(cap + room)) return p->fail(MCOM_E_NOMEM, "minimizer records");
		int rc = mcom_sketch_contigs(p->ctx, S.seq.p, S.soff.p, nullptr, n_first, p->rw, p->k, 0, S.roff.p, S.rec.p, cap, &total);
		if (rc == MCOM_E_OVERFLOW) { cap = total; continue; }
		if (!rc) p->stat["sketch_records"] += (double)total;
		return p->gpu(rc);
	}
	return p->fail(MCOM_E_OVERFLOW, "minimizer buffer");
}

// Multi-GPU form of sketch_first: the contig set is replicated, so a rank sketches its share of the contigs (as if they
// were contigs 0, 1, ...), moves ids and offsets to their global values and the shares are all-gathered in contig order.
static int sketch_first_dist(P *p, DevSet &S, size_t n_first, uint64_t chars_first, size_t room, uint64_t &total)
{
	total = 0;
	const int R = p->world, me = p->rank;
	const size_t c0 = n_first * (size_t)me / R, c1 = n_first * (size_t)(me + 1) / R, nloc = c1 - c0;
	if (!S.roff.reserve(std::max(S.n + 2, S.soff.cap))) return p->fail(MCOM_E_NOMEM, "minimizer offsets");
	DevBuf<mcom_mm128> tmp; DevBuf<uint32_t> toff;
	if (!toff.reserve(nloc + 2)) return p->fail(MCOM_E_NOMEM, "minimizer offsets");
	size_t cap = std::max<size_t>(1024, chars_first / 8 / R + nloc + 1024);
	uint64_t tl = 0;
	int rc = MCOM_OK;
	for (int attempt = 0; attempt < 2; ++attempt) {
		if (!tmp.reserve(cap)) return p->fail(MCOM_E_NOMEM, "minimizer records");
		rc = mcom_sketch_contigs(p->ctx, S.seq.p, S.soff.p + c0, nullptr, nloc, p->rw, p->k, 0, toff.p, tmp.p, cap, &tl);
		if (rc == MCOM_E_OVERFLOW) { cap = tl; continue; }
		break;
	}
	if (rc) return p->gpu(rc);
	uint64_t hs[2] = {0, 0};
	if (nloc && ((rc = p->d2h(&hs[0], S.soff.p + c0, 1, "copy")) || (rc = p->d2h(&hs[1], S.soff.p + c1, 1, "copy")) || (rc = p->sync("copy")))) return rc;
	p->stat["sketch_bases"] += (double)(hs[1] - hs[0]); p->stat["sketch_records"] += (double)tl;
	std::vector<uint64_t> all, first(R), cnt(R), fo(R), co(R);
	if ((rc = gather_host(p, &tl, 1, all))) return rc;
	for (int q = 0; q < R; ++q) { first[q] = total; cnt[q] = all[q]; total += all[q]; fo[q] = n_first * (size_t)q / R; co[q] = n_first * (size_t)(q + 1) / R - fo[q]; }
	if (total >= (1ull << 32)) return p->fail(MCOM_E_ARG, "more than 2^32-1 minimizer records");
	if (!S.rec.reserve(total + room + 1)) return p->fail(MCOM_E_NOMEM, "minimizer records");
	if ((rc = p->gpu(mcom_records_rebase(p->ctx, tmp.p, tl, (uint32_t)c0, toff.p, nloc, (uint32_t)first[me])))) return rc;
	const double tx = now_ms(), bx0 = xbytes(p);
	if ((rc = gatherv(p, S.rec.p, first, cnt, tmp.p)) || (rc = gatherv(p, S.roff.p, fo, co, toff.p, false))) return rc;
	p->stat["t_x_sketch"] += now_ms() - tx; p->stat["b_x_sketch"] += xbytes(p) - bx0;
	const uint32_t t32 = (uint32_t)total;
	if ((rc = p->h2d(S.roff.p + n_first, &t32, 1, "upload")) || (rc = p->sync("upload"))) return rc;
	return MCOM_OK;
}

// Multi-GPU form of mm_idx_generation (kthread_idx.c:116-170 is a loop over independent buckets): a rank sorts the records of its
// bucket range (ascending ranges in rank order, as in the bucket stage) and builds their table regions; sorted records and regions
// are all-gathered into the replicated index, which every rank then queries for its share of the contigs.
// (round 5: rec_m holds the index records of THIS RANK'S SHARE of the list only -- tm_loc of them, in list order; they are partitioned by the
// rank that owns their bucket and travel there, where the shares arrive in rank order = list order = the order mm_idx_generation pushes
// in.  Before, every rank extracted and partitioned the records of the whole list: 2.3 ms per rank of an 8-rank job.)
static int build_index_dist(P *p, const mcom_mm128 *rec_m, uint64_t tm_loc, mcom_idx **out)
{
	const int R = p->world, me = p->rank;
	*out = nullptr;
	int rc;
	DevBuf<mcom_mm128> send, part;
	if (!send.reserve(tm_loc + 1)) return p->fail(MCOM_E_NOMEM, "index records");
	std::vector<uint64_t> mine(R, 0), all, cnt(R, 0), first(R, 0);
	if ((rc = p->gpu(mcom_partition_by_owner(p->ctx, rec_m, tm_loc, NB_BITS, R, send.p, mine.data())))) return rc;
	{ uint64_t t = 0; for (int q = 0; q < R; ++q) t += mine[q]; if (t != tm_loc) return p->fail(MCOM_E_ARG, "index records without a minimizer"); }
	if ((rc = gather_host(p, mine.data(), R, all))) return rc;
	uint64_t tm = 0;
	for (int q = 0; q < R; ++q) { for (int s2 = 0; s2 < R; ++s2) cnt[q] += all[(size_t)s2 * R + q]; first[q] = tm; tm += cnt[q]; }
	{
		std::vector<uint64_t> so(R), sb(R), ro(R), rb(R);
		uint64_t a = 0, b = 0;
		for (int q = 0; q < R; ++q) { so[q] = a * 16; sb[q] = mine[q] * 16; a += mine[q]; ro[q] = b * 16; rb[q] = all[(size_t)q * R + me] * 16; b += all[(size_t)q * R + me]; }
		if (b != cnt[me]) return p->fail(MCOM_E_ARG, "index exchange: counts disagree");
		if (!part.reserve(b + 1)) return p->fail(MCOM_E_NOMEM, "index records");
		const double tx = now_ms(), bx0 = xbytes(p);
		if ((rc = alltoallv_dev(p, send.p, so.data(), sb.data(), part.p, ro.data(), rb.data()))) return rc;
		p->stat["t_x_index"] += now_ms() - tx; p->stat["b_x_index"] += xbytes(p) - bx0;
	}
	mcom_idx *mi = nullptr;
	if ((rc = p->gpu(mcom_idx_create(p->ctx, tm, p->k, NB_BITS, &mi)))) return rc;
	uint32_t mx = 0;
	rc = p->gpu(mcom_idx_sort_part(p->ctx, mi, part.p, cnt[me], first[me], &mx));
	uint64_t mxa = mx;
	if (!rc) rc = allreduce_host(p, &mxa, 1, 2);
	if (rc) { mcom_idx_destroy(p->ctx, mi); return rc; }
	// the sorted parts are all-gathered; the table regions are made from them on every rank (round 5: the regions, sized for the fullest
	// bucket and a quarter full, were 3 GB of the 13.8 a rank sent per step)
	mcom_mm128 *d_rec = nullptr; uint64_t *d_slots = nullptr; uint32_t region = 0;
	mcom_idx_buffers(mi, &d_rec, &d_slots, &region);
	(void)d_slots; (void)region;
	const double tx = now_ms(), bx0 = xbytes(p);
	rc = gatherv(p, d_rec, first, cnt);
	p->stat["t_x_index"] += now_ms() - tx; p->stat["b_x_index"] += xbytes(p) - bx0;
	bool global_table = false;
	if (!rc) {
		rc = mcom_idx_table_all(p->ctx, mi, (uint32_t)mxa);
		if (rc == MCOM_E_OVERFLOW) { global_table = true; rc = MCOM_OK; }                 // (the same on every rank: mxa is)
		else if (rc) p->gpu(rc);
	}
	if (!rc && global_table) { rc = p->gpu(mcom_idx_table_global(p->ctx, mi)); p->stat["idx_global_tables"] += 1; }
	if (rc) { mcom_idx_destroy(p->ctx, mi); return rc; }
	*out = mi;
	return MCOM_OK;
}

// ----------------------------------------------------------------------------------------------------
// combine_cluster: merge rounds                                                kthread_cb.c:570-630
// The contig set (consensus strings, members, minimizers) lives on the device for the whole stage; per round only the
// passing candidate pairs come to the host, for the first-come claiming, and the claimed pairs go back.
// ----------------------------------------------------------------------------------------------------
static int combine_cluster_impl(mcomh_pipeline *p);
extern "C" int mcomh_combine_cluster(mcomh_pipeline *p) { return p ? stage_exit(p, combine_cluster_impl(p)) : MCOM_E_ARG; }
static int combine_cluster_impl(mcomh_pipeline *p)
{
	if (!p) return MCOM_E_ARG;
	const double t0 = busy_now(p);
	const int L = p->L;
	long pre = 0;
	int rc;
	ContigSet &C = p->C;
	// Round 5: the contig set of this stage is an append-only STORE (S: strings, member lists, minimizer records; the packed words in
	// p->d_cbits & co.).  A contig's index in it never changes -- so neither do the ids in its records -- and the list of a round is
	// `ord`, indices into the store in visiting order (nullptr: the store's own order, the first round).  cp_cluster's copies of the
	// unmerged contigs (kthread_cb.c:397-434) are gone: a round appends what it merged and makes the next list (include/mcom.h,
	// "Merge rounds without cp_cluster's copies"); when the rounds are over the list is gathered into an ordinary set, once.
	DevSet S, Tm;                                                            // Tm: a rank's share of a round's merged contigs (multi-GPU)
	DevBuf<uint32_t> ord, ord2, moff_m, d_jobs, roff_t, d_ids, clen_l; DevBuf<uint64_t> jmoff_t, jroff_t, cw_t, cw_l, bits_l;
	DevBuf<mcom_mm128> rec_m, d_pairs, d_pairs_loc; DevBuf<uint8_t> d_flag;
	PinVec<mcom_mm128> pairs; PinVec<uint8_t> flag;
	struct Job { uint32_t ci, cj, pos_ori, pos; };
	PinVec<Job> jobs;
	double tl = busy_now(p);
	auto lap = [&](const char *nm) { const double t = busy_now(p); p->stat[nm] += t - tl; tl = t; };
	// the set of the bucket stage is on the device already (kt_for_bucket); a set that only exists on the host is uploaded
	uint64_t maxlen = 2 * (uint64_t)L;                                        // a group's consensus spans at most 2L columns
	if (p->dC_valid) { S.swap(p->dC); p->dC_valid = false; }
	else {
		S.n = C.n(); S.chars = C.ref.size(); S.members = C.mem.size();
		for (size_t i = 0; i < S.n; ++i) maxlen = std::max<uint64_t>(maxlen, C.rsize(i));
		if (S.n) {
			if (!S.seq.reserve(S.chars + 16) || !S.soff.reserve(S.n + 1) || !S.mem.reserve(S.members + 1) || !S.moff.reserve(S.n + 1)) return p->fail(MCOM_E_NOMEM, "contig set");
			if ((rc = p->h2d(S.seq.p, (const uint8_t*)C.ref.data(), S.chars, "upload contigs")) || (rc = p->h2d(S.soff.p, C.roff.data(), S.n + 1, "upload offsets")) ||
			    (rc = p->h2d(S.mem.p, C.mem.data(), S.members, "upload members")) || (rc = p->h2d(S.moff.p, C.moff.data(), S.n + 1, "upload offsets")) ||
			    (rc = p->sync("upload contig set"))) return rc;
		}
	}
	const uint64_t chars0 = S.chars, members0 = S.members;                    // a merged string is no longer than its parents together: bounds of every later list
	if (S.n) {
		const double tg = busy_now(p);
		lap("t_cb_upload");
		if ((rc = p->comm ? sketch_first_dist(p, S, S.n, S.chars, 0, S.nrec) : sketch_first(p, S, S.n, S.chars, 0, S.nrec))) return rc;   // find_next's own sketch (:234)
		// room for the records the merge rounds append (about 1.3 x the first sketch in all): made once, here
		if (S.rec.cap < S.nrec * 23 / 10 && !S.rec.grow((size_t)(S.nrec * 23 / 10 + 1024), (size_t)S.nrec, p->stream)) return p->fail(MCOM_E_NOMEM, "record store");
		lap("t_cb_sketch");
		p->stat["t_gpu"] += busy_now(p) - tg;
	}
	bool packed_ready = false;                                                 // d_cbits & co. describe the store
	bool listed = false;                                                       // `ord` is in use (some round has merged something)
	p->cbits_for_dC = false;
	size_t n_live = S.n;                                                       // contigs of the current list
	uint32_t first_new = 0, n_new = 0;                                         // the contigs the round before made: indices [first_new, first_new + n_new) (0: the first round)
	if (p->comm && n_live) {                                                   // (multi-GPU: shares are ranges of the list, so the list is spelled out from the start)
		uint64_t nk = 0;
		if (!ord.reserve(n_live + 1)) return p->fail(MCOM_E_NOMEM, "contig list");
		if ((rc = p->gpu(mcom_order_next(p->ctx, nullptr, 0, nullptr, 0, n_live, ord.p, &nk)))) return rc;
		listed = true;
	}
	// the packed form of n contigs (strings in S.seq, `tw` words in all) at `out`: on several GPUs a rank packs its share of the WORDS and the
	// shares are all-gathered (every rank packing everything was 2 ms per rank of an 8-rank job: profiles/r05_dist_kernels.txt)
	auto pack_words = [&](const uint64_t *soff, const uint64_t *cw, size_t n, uint64_t tw, uint64_t *out) -> int {
		if (!p->comm || tw < 4096) return p->gpu(mcom_pack_contigs(p->ctx, S.seq.p, soff, cw, (uint32_t)n, tw, out));
		const int R = p->world, me = p->rank;
		std::vector<uint64_t> first(R), cnt(R);
		for (int q = 0; q < R; ++q) { first[q] = tw * (uint64_t)q / R; cnt[q] = tw * (uint64_t)(q + 1) / R - first[q]; }
		int rc2 = p->gpu(mcom_pack_contigs_words(p->ctx, S.seq.p, soff, cw, (uint32_t)n, tw, out, first[me], first[me] + cnt[me]));
		if (rc2) return rc2;
		const double tx = now_ms(), bx0 = xbytes(p);
		rc2 = gatherv(p, out, first, cnt);
		p->stat["t_x_packed"] += now_ms() - tx; p->stat["b_x_packed"] += xbytes(p) - bx0;
		return rc2;
	};
	// room behind the end of a store array, made by growing it (the data in front is kept); the bucket stage leaves some (arena_slack)
	auto room64 = [&](DevBuf<uint64_t> &b, uint64_t used, uint64_t more) { return b.grow((size_t)(used + more), (size_t)used, p->stream); };
	for (;;) {
		const size_t n = n_live, n_store = S.n;
		const uint32_t *lst = listed ? ord.p : nullptr;
		uint64_t n_pass = 0;
		if (n) {
			const double tg = busy_now(p);
			uint64_t tw = 0, tm = 0;
			if (!packed_ready) {
				if (!p->d_coff_words.reserve(std::max(n_store + 1, S.soff.cap)) || !p->d_clen.reserve(std::max(n_store + 1, S.soff.cap))) return p->fail(MCOM_E_NOMEM, "contig layout");   // (with the room the store's offset arrays have)
				if ((rc = p->gpu(mcom_contig_layout(p->ctx, S.soff.p, n_store, p->d_coff_words.p, p->d_clen.p, &tw)))) return rc;
				p->total_words = tw;
				if (p->comm && p->prepacked && p->prepacked_words == tw && p->d_cbits.cap >= tw + 2) {
					// several GPUs: the bucket stage's contigs travelled as packed words and are in their places (kt_for_bucket)
					if (!p->d_cbits.grow((size_t)(tw * 24 / 10 + 2), (size_t)tw, p->stream)) return p->fail(MCOM_E_NOMEM, "packed contigs");
					if ((rc = p->hipc(hipMemsetAsync(p->d_cbits.p + tw, 0, 16, p->stream), "clear"))) return rc;
				} else {
					if (!p->d_cbits.reserve(tw * 24 / 10 + 2)) return p->fail(MCOM_E_NOMEM, "packed contigs");
					if ((rc = p->hipc(hipMemsetAsync(p->d_cbits.p, 0, (tw + 2) * 8, p->stream), "clear"))) return rc;
					if ((rc = pack_words(S.soff.p, p->d_coff_words.p, n_store, tw, p->d_cbits.p))) return rc;
				}
				p->prepacked = false;
				packed_ready = true;
			}
			// the first m minimizers are what the contig builders pushed into mi[index] (kthread_bucket.c:463, :370-380, :423-432)
			if (!moff_m.reserve(n + 2) || !rec_m.reserve(n * (size_t)p->m + 16)) return p->fail(MCOM_E_NOMEM, "index records");
			{
				// (several GPUs: the records of this rank's share of the list; build_index_dist sends them to their buckets' owners)
				const size_t a0 = p->comm ? n * (size_t)p->rank / (size_t)p->world : 0, a1 = p->comm ? n * (size_t)(p->rank + 1) / (size_t)p->world : n;
				if ((rc = p->gpu(mcom_minimizer_prefix_ord(p->ctx, S.roff.p, S.rec.p, lst ? lst + a0 : nullptr, a1 - a0, (uint32_t)p->m, moff_m.p, rec_m.p, &tm)))) return rc;
			}
			lap("t_cb_pack");
			mcom_idx *mi = nullptr;
			if ((rc = p->comm ? build_index_dist(p, rec_m.p, tm, &mi) : p->gpu(mcom_idx_build(p->ctx, rec_m.p, tm, p->k, NB_BITS, &mi)))) return rc;   // mm_idx_generation (:580)
			lap("t_cb_idx");
			uint64_t hc[2] = {0, 0};
			if (!p->comm) {
				size_t cap = std::max<size_t>(1024, (size_t)S.nrec);                      // (the store's records: at least the list's, which bounded the pairs in every run so far; more: the call says how many)
				for (int attempt = 0; attempt < 2; ++attempt) {
					if (!d_pairs.reserve(cap)) { mcom_idx_destroy(p->ctx, mi); return p->fail(MCOM_E_NOMEM, "candidate pairs"); }
					// (no list yet: the store is the list and its records are the queries in visiting order -- one thread per query)
					rc = listed ? mcom_find_next_candidates_ord(p->ctx, mi, S.rec.p, S.roff.p, lst, n, p->d_cbits.p, p->d_coff_words.p, p->d_clen.p, p->cbthr, first_new, n_new, d_pairs.p, d_pairs.cap, hc)
					            : mcom_find_next_candidates_new(p->ctx, mi, S.rec.p, (size_t)S.nrec, p->d_cbits.p, p->d_coff_words.p, p->d_clen.p, p->cbthr, 0, d_pairs.p, d_pairs.cap, hc);
					if (rc == MCOM_E_OVERFLOW) { cap = hc[1]; continue; }
					break;
				}
			} else {
				// multi-GPU: every rank holds the whole index (6 minimizers per contig) and evaluates the queries of its share of
				// the list; the passing pairs come out in visiting order, so the shares concatenate in rank order
				const int R = p->world, me = p->rank;
				const size_t c0 = n * (size_t)me / R, c1 = n * (size_t)(me + 1) / R;
				size_t cap = std::max<size_t>(1024, (size_t)S.nrec / (size_t)R + 1024);
				for (int attempt = 0; attempt < 2; ++attempt) {
					if (!d_pairs_loc.reserve(cap)) { mcom_idx_destroy(p->ctx, mi); return p->fail(MCOM_E_NOMEM, "candidate pairs"); }
					rc = mcom_find_next_candidates_ord(p->ctx, mi, S.rec.p, S.roff.p, ord.p + c0, c1 - c0, p->d_cbits.p, p->d_coff_words.p, p->d_clen.p, p->cbthr, first_new, n_new, d_pairs_loc.p, d_pairs_loc.cap, hc);
					if (rc == MCOM_E_OVERFLOW) { cap = hc[1]; continue; }
					break;
				}
				if (rc) { p->gpu(rc); mcom_idx_destroy(p->ctx, mi); return rc; }
				std::vector<uint64_t> all, first(R), cnt(R);
				if ((rc = gather_host(p, hc, 2, all))) { mcom_idx_destroy(p->ctx, mi); return rc; }
				hc[0] = hc[1] = 0;
				for (int q = 0; q < R; ++q) { first[q] = hc[1]; cnt[q] = all[2 * q + 1]; hc[0] += all[2 * q]; hc[1] += all[2 * q + 1]; }
				if (!d_pairs.reserve(hc[1] + 1)) { mcom_idx_destroy(p->ctx, mi); return p->fail(MCOM_E_NOMEM, "candidate pairs"); }
				const double tx = now_ms(), bx0 = xbytes(p);
				rc = gatherv(p, d_pairs.p, first, cnt, d_pairs_loc.p);
				p->stat["t_x_pairs"] += now_ms() - tx; p->stat["b_x_pairs"] += xbytes(p) - bx0;
				if (rc) { mcom_idx_destroy(p->ctx, mi); return rc; }
			}
			mcom_idx_destroy(p->ctx, mi);
			if (rc) return p->gpu(rc);
			lap("t_cb_findnext");
			n_pass = hc[1];
			p->stat["t_gpu"] += busy_now(p) - tg;
			p->stat["cand_pairs"] += (double)hc[0];
		}
		// first-come claiming in contig order (find_next :267-343 at one thread) = the greedy matching over the pair list,
		// settled in rounds on the device; a list that does not settle goes through the sequential loop on the host.  The flags are
		// indexed like everything else: by the contig's index in the store.
		size_t nj = 0;
		bool on_device = false;
		if (n) {
			if (!d_jobs.reserve(4 * (n_store / 2 + 1)) || !d_flag.reserve(n_store + 16)) return p->fail(MCOM_E_NOMEM, "claim buffers");
			uint64_t njv = 0; int rounds = 0;
			rc = mcom_claim_pairs(p->ctx, d_pairs.p, n_pass, n_store, 4096, d_jobs.p, d_flag.p, &njv, &rounds);
			if (rc == MCOM_OK) { on_device = true; nj = (size_t)njv; p->stat["claim_rounds"] += rounds; }
			else if (rc != MCOM_E_OVERFLOW) return p->gpu(rc);
		}
		if (!on_device) {
		if (!pairs.resize(n_pass)) return p->fail(MCOM_E_NOMEM, "candidate pairs");
		if ((rc = p->d2h(pairs.data(), d_pairs.p, n_pass, "copy candidates")) || (rc = p->sync("candidates"))) return rc;
		if (!flag.resize(n_store) || !jobs.resize(n / 2 + 1)) return p->fail(MCOM_E_NOMEM, "claim buffers");
		if (n_store) memset(flag.data(), 0, n_store);
		for (size_t q = 0; q < n_pass;) {
			const uint32_t ci = (uint32_t)(pairs[q].x >> 32);
			size_t qe = q;
			while (qe < n_pass && (uint32_t)(pairs[qe].x >> 32) == ci) ++qe;
			if (!flag[ci]) {
				for (size_t u = q; u < qe; ++u) {
					const uint32_t cj = (uint32_t)(pairs[u].y >> 32);
					if (flag[cj]) continue;
					jobs[nj++] = Job{ci, cj, (uint32_t)pairs[u].x >> 1, (uint32_t)pairs[u].y >> 1};
					flag[ci] = flag[cj] = 1;                                                // :339-343
					break;
				}
			}
			q = qe;
		}
		if (nj && ((rc = p->h2d(d_jobs.p, (const uint32_t*)jobs.data(), 4 * nj, "upload claimed pairs")) || (rc = p->h2d(d_flag.p, flag.data(), n_store, "upload flags")) ||
		           (rc = p->sync("upload claims")))) return rc;
		}
		lap("t_claim");
		if (nj) {
			const double tg = busy_now(p);
			const size_t nkeep = n - 2 * nj, nn = nj + nkeep;
			if ((uint64_t)n_store + nj >= (1ull << 32) - 2) return p->fail(MCOM_E_ARG, "more than 2^32 contigs made in the merge rounds");
			// the offset arrays of the store gain nj entries
			if (!S.soff.grow(n_store + nj + 2, n_store + 1, p->stream) || !S.moff.grow(n_store + nj + 2, n_store + 1, p->stream) ||
			    !S.roff.grow(n_store + nj + 2, n_store + 1, p->stream) || !p->d_coff_words.grow(n_store + nj + 2, n_store + 1, p->stream) ||
			    !p->d_clen.grow(n_store + nj + 2, n_store, p->stream) || !jmoff_t.reserve(nj + 2) || !jroff_t.reserve(nj + 2) || !roff_t.reserve(nj + 2) || !cw_t.reserve(nj + 2))
				return p->fail(MCOM_E_NOMEM, "merge buffers");
			// merged member lists (:297-325) in cmpcluster2 order as construct_ref2 sorts them first (:107)
			int kb = 2; while ((1ull << kb) < 4 * maxlen + 4) ++kb;
			if (kb > 29) return p->fail(MCOM_E_ARG, "contig of %llu bases: member offsets need more than 28 bits", (unsigned long long)maxlen);
			uint64_t tot[3] = {0, 0, 0};
			uint64_t tn = 0;                                                             // minimizer records of the merged contigs
			uint64_t words_gathered = 0, own_chars_at = 0, own_chars = 0; bool words_in_place = false;   // (several GPUs: the round's packed words arrive with the gather)
			const bool rs = p->resketch && (p->k & 1);
			if (!p->comm) {
				for (int attempt = 0;; ++attempt) {
					rc = mcom_merge_members_cap(p->ctx, S.mem.p, S.moff.p, d_jobs.p, nj, L, kb, S.mem.p + S.members, S.mem.cap - S.members, jmoff_t.p, jroff_t.p, tot);
					if (rc == MCOM_E_OVERFLOW && attempt == 0) { if (!room64(S.mem, S.members, tot[0] + 1)) return p->fail(MCOM_E_NOMEM, "member store"); p->stat["store_grows"] += 1; continue; }
					if (rc) return p->gpu(rc);
					break;
				}
				maxlen = std::max(maxlen, tot[2]);
				lap("t_merge_members");
				// construct_ref2 of every merged contig (:327), written behind the strings of the store
				if (S.seq.cap < S.chars + tot[1] + 16) { if (!S.seq.grow((size_t)(S.chars + tot[1] + 16), (size_t)S.chars, p->stream)) return p->fail(MCOM_E_NOMEM, "string store"); p->stat["store_grows"] += 1; }
				if ((rc = p->gpu(mcom_merge_consensus_jobs(p->ctx, p->d_packed.p, S.mem.p + S.members, jmoff_t.p, jroff_t.p, nj, tot[1], L, S.seq.p + S.chars,
				                                           p->full_consensus ? nullptr : d_jobs.p, S.seq.p, S.soff.p)))) return rc;
				lap("t_merge_cons");
				// their minimizers: only around the overlaps, the parents' records carry the rest (csrc/resketch.hip); even k: sketched whole
				uint64_t sk = 0;
				if (rs) {
					for (int attempt = 0;; ++attempt) {
						rc = mcom_resketch_merged_at(p->ctx, d_jobs.p, nj, S.soff.p, S.rec.p, S.roff.p, S.seq.p + S.chars, jroff_t.p, tot[1], p->rw, p->k, (uint32_t)n_store,
						                             roff_t.p, S.rec.p ? S.rec.p + S.nrec : nullptr, S.rec.cap > S.nrec ? S.rec.cap - (size_t)S.nrec : 0, &tn, &sk);
						if (rc == MCOM_E_OVERFLOW && attempt == 0) { if (!S.rec.grow((size_t)(S.nrec + tn + 1), (size_t)S.nrec, p->stream)) return p->fail(MCOM_E_NOMEM, "record store"); p->stat["store_grows"] += 1; continue; }
						if (rc) return p->gpu(rc);
						break;
					}
					p->stat["resketch_saved_bases"] += (double)(tot[1] - sk);
				} else {
					uint64_t nk = 0;
					if (!d_ids.reserve(nj + 1)) return p->fail(MCOM_E_NOMEM, "contig ids");
					if ((rc = p->gpu(mcom_order_next(p->ctx, nullptr, 0, nullptr, (uint32_t)n_store, nj, d_ids.p, &nk)))) return rc;
					size_t cap = std::max<size_t>(1024, tot[1] / 8 + nj);
					for (int attempt = 0;; ++attempt) {
						if (S.rec.cap < S.nrec + cap && !S.rec.grow((size_t)(S.nrec + cap), (size_t)S.nrec, p->stream)) return p->fail(MCOM_E_NOMEM, "record store");
						rc = mcom_sketch_contigs(p->ctx, S.seq.p + S.chars, jroff_t.p, d_ids.p, nj, p->rw, p->k, 0, roff_t.p, S.rec.p + S.nrec, cap, &tn);
						if (rc == MCOM_E_OVERFLOW && attempt == 0) { cap = tn; continue; }
						if (rc) return p->gpu(rc);
						break;
					}
					sk = tot[1];
				}
				p->stat["sketch_bases"] += (double)sk; p->stat["sketch_records"] += (double)tn;
				if (S.nrec + tn >= (1ull << 32)) return p->fail(MCOM_E_ARG, "more than 2^32-1 minimizer records in the store");
				// the store's offset arrays: entries n_store .. n_store + nj
				if ((rc = p->gpu(mcom_offsets_append(p->ctx, jmoff_t.p, nj, S.members, S.moff.p + n_store))) || (rc = p->gpu(mcom_offsets_append(p->ctx, jroff_t.p, nj, S.chars, S.soff.p + n_store))) ||
				    (rc = p->gpu(mcom_offsets_append_u32(p->ctx, roff_t.p, nj, (uint32_t)S.nrec, S.roff.p + n_store)))) return rc;
				lap("t_cb_sketch");
			} else {
				// Multi-GPU: the merges of a round are independent (find_next :297-381 works on one claimed pair): a rank merges its share of
				// the claimed pairs -- member lists, construct_ref2, the sketch around the overlap -- in buffers of its own, and the shares are
				// all-gathered straight behind the end of every rank's store, in claiming order.
				const int R = p->world, me = p->rank;
				const size_t j0 = nj * (size_t)me / R, j1 = nj * (size_t)(me + 1) / R, njl = j1 - j0;
				DevSet &T = Tm;
				uint64_t tl3[3] = {0, 0, 0}, tnl = 0, sk = 0, twl = 0;
				if (njl) {
					if (!T.mem.reserve(members0 + 1) || !T.moff.reserve(njl + 2) || !T.soff.reserve(njl + 2)) return p->fail(MCOM_E_NOMEM, "merge buffers");
					if ((rc = p->gpu(mcom_merge_members(p->ctx, S.mem.p, S.moff.p, d_jobs.p + 4 * j0, njl, L, kb, T.mem.p, T.moff.p, T.soff.p, tl3)))) return rc;
					if (!T.seq.reserve(tl3[1] + 16)) return p->fail(MCOM_E_NOMEM, "merge buffers");
					if ((rc = p->gpu(mcom_merge_consensus_jobs(p->ctx, p->d_packed.p, T.mem.p, T.moff.p, T.soff.p, njl, tl3[1], L, T.seq.p, p->full_consensus ? nullptr : d_jobs.p + 4 * j0, S.seq.p, S.soff.p)))) return rc;
					if (!T.roff.reserve(njl + 2)) return p->fail(MCOM_E_NOMEM, "minimizer offsets");
					size_t cap = std::max<size_t>(1024, tl3[1] / 8 + njl);
					if (rs) {
						for (int attempt = 0;; ++attempt) {
							if (!T.rec.reserve(cap)) return p->fail(MCOM_E_NOMEM, "minimizer records");
							rc = mcom_resketch_merged_at(p->ctx, d_jobs.p + 4 * j0, njl, S.soff.p, S.rec.p, S.roff.p, T.seq.p, T.soff.p, tl3[1], p->rw, p->k, (uint32_t)(n_store + j0), T.roff.p, T.rec.p, cap, &tnl, &sk);
							if (rc == MCOM_E_OVERFLOW && attempt == 0) { cap = tnl; continue; }
							if (rc) return p->gpu(rc);
							break;
						}
						p->stat["resketch_saved_bases"] += (double)(tl3[1] - sk);
					} else {
						uint64_t nk = 0;
						if (!d_ids.reserve(njl + 1)) return p->fail(MCOM_E_NOMEM, "contig ids");
						if ((rc = p->gpu(mcom_order_next(p->ctx, nullptr, 0, nullptr, (uint32_t)(n_store + j0), njl, d_ids.p, &nk)))) return rc;
						for (int attempt = 0;; ++attempt) {
							if (!T.rec.reserve(cap)) return p->fail(MCOM_E_NOMEM, "minimizer records");
							rc = mcom_sketch_contigs(p->ctx, T.seq.p, T.soff.p, d_ids.p, njl, p->rw, p->k, 0, T.roff.p, T.rec.p, cap, &tnl);
							if (rc == MCOM_E_OVERFLOW && attempt == 0) { cap = tnl; continue; }
							if (rc) return p->gpu(rc);
							break;
						}
						sk = tl3[1];
					}
					p->stat["sketch_bases"] += (double)sk; p->stat["sketch_records"] += (double)tnl;
					// The strings of the share do not travel: their PACKED WORDS do (a quarter of a byte per base -- and the packed form is
					// needed on every rank anyway, so the strings went on top of it: 2.4 of 11.5 GB per rank and step at 64 M reads over eight
					// ranks), and every rank unpacks the strings the others built (mcom_unpack_contigs, below).  A contig owns whole words, so the
					// layouts of the shares, one behind the other, are the layout of the round's contigs.
					if (!cw_l.reserve(njl + 2) || !clen_l.reserve(njl + 2)) return p->fail(MCOM_E_NOMEM, "merge buffers");
					if ((rc = p->gpu(mcom_contig_layout(p->ctx, T.soff.p, njl, cw_l.p, clen_l.p, &twl)))) return rc;
					if (!bits_l.reserve(twl + 2)) return p->fail(MCOM_E_NOMEM, "merge buffers");
					if ((rc = p->gpu(mcom_pack_contigs(p->ctx, T.seq.p, T.soff.p, cw_l.p, (uint32_t)njl, twl, bits_l.p)))) return rc;
				}
				lap("t_merge_local");
				const uint64_t mine[5] = {tl3[0], tl3[1], tl3[2], tnl, twl};
				std::vector<uint64_t> all, fj(R), cj(R), fm(R), cm(R), fc(R), cc(R), fr(R), cr(R), fw(R), cwq(R);
				if ((rc = gather_host(p, mine, 5, all))) return rc;
				for (int q = 0; q < R; ++q) {
					fj[q] = n_store + nj * (size_t)q / R; cj[q] = nj * (size_t)(q + 1) / R - nj * (size_t)q / R;
					fm[q] = S.members + tot[0]; cm[q] = all[5 * q]; tot[0] += cm[q]; fc[q] = S.chars + tot[1]; cc[q] = all[5 * q + 1]; tot[1] += cc[q];
					tot[2] = std::max(tot[2], all[5 * q + 2]); fr[q] = S.nrec + tn; cr[q] = all[5 * q + 3]; tn += cr[q];
					fw[q] = p->total_words + words_gathered; cwq[q] = all[5 * q + 4]; words_gathered += cwq[q];
				}
				if (p->d_cbits.cap < p->total_words + words_gathered + 2) { if (!p->d_cbits.grow((size_t)(p->total_words + words_gathered + 2), (size_t)p->total_words, p->stream)) return p->fail(MCOM_E_NOMEM, "packed store"); p->stat["store_grows"] += 1; }
				words_in_place = true; own_chars_at = fc[me]; own_chars = cc[me];
				maxlen = std::max(maxlen, tot[2]);
				if (S.nrec + tn >= (1ull << 32)) return p->fail(MCOM_E_ARG, "more than 2^32-1 minimizer records in the store");
				// offsets of the share move to their places in the store, then everything travels
				if (njl && ((rc = p->gpu(mcom_offsets_rebase(p->ctx, T.moff.p, njl, fm[me]))) || (rc = p->gpu(mcom_offsets_rebase(p->ctx, T.soff.p, njl, fc[me]))) ||
				            (rc = p->gpu(mcom_records_rebase(p->ctx, T.rec.p, 0, 0, T.roff.p, njl, (uint32_t)fr[me]))))) return rc;
				if (!room64(S.mem, S.members, tot[0] + 1) || !S.seq.grow((size_t)(S.chars + tot[1] + 16), (size_t)S.chars, p->stream) || !S.rec.grow((size_t)(S.nrec + tn + 1), (size_t)S.nrec, p->stream))
					return p->fail(MCOM_E_NOMEM, "contig store");
				const double tx = now_ms(), bx0 = xbytes(p);
				const uint64_t em = S.members + tot[0], ec = S.chars + tot[1]; const uint32_t er = (uint32_t)(S.nrec + tn);
				if ((rc = gatherv(p, S.mem.p, fm, cm, T.mem.p)) || (rc = gatherv(p, S.moff.p, fj, cj, T.moff.p, false)) || (rc = gatherv(p, p->d_cbits.p, fw, cwq, bits_l.p, false)) || (rc = gatherv(p, S.soff.p, fj, cj, T.soff.p, false)) ||
				    (rc = gatherv(p, S.rec.p, fr, cr, T.rec.p, false)) || (rc = gatherv(p, S.roff.p, fj, cj, T.roff.p, false))) return rc;
				if ((rc = p->h2d(S.moff.p + n_store + nj, &em, 1, "upload")) || (rc = p->h2d(S.soff.p + n_store + nj, &ec, 1, "upload")) || (rc = p->h2d(S.roff.p + n_store + nj, &er, 1, "upload"))) return rc;
				if ((rc = p->sync("merged contigs"))) return rc;                             // (em / ec / er live on this stack frame)
				p->stat["t_x_merged"] += now_ms() - tx; p->stat["b_x_merged"] += xbytes(p) - bx0;
				lap("t_merge_gather");
			}
			// the packed form of the new contigs, behind the packed store
			{
				uint64_t tw2 = 0;
				if ((rc = p->gpu(mcom_contig_layout(p->ctx, S.soff.p + n_store, nj, cw_t.p, p->d_clen.p + n_store, &tw2)))) return rc;
				if (words_in_place && tw2 != words_gathered) return p->fail(MCOM_E_ARG, "merged contigs: %llu packed words gathered, the layout has %llu", (unsigned long long)words_gathered, (unsigned long long)tw2);
				if (p->d_cbits.cap < p->total_words + tw2 + 2) { if (!p->d_cbits.grow((size_t)(p->total_words + tw2 + 2), (size_t)p->total_words, p->stream)) return p->fail(MCOM_E_NOMEM, "packed store"); p->stat["store_grows"] += 1; }
				// (several GPUs: the words are there -- they are what travelled -- and the strings are made from them)
				if (words_in_place) {
					// this rank's own strings are copied from where it built them, the others' are unpacked
					const size_t j0 = nj * (size_t)p->rank / (size_t)p->world, j1 = nj * (size_t)(p->rank + 1) / (size_t)p->world;
					if (j1 > j0 && (rc = p->hipc(hipMemcpyAsync(S.seq.p + own_chars_at, Tm.seq.p, own_chars, hipMemcpyDeviceToDevice, p->stream), "copy merged strings"))) return rc;
					if (j0 && (rc = p->gpu(mcom_unpack_contigs(p->ctx, p->d_cbits.p + p->total_words, cw_t.p, S.soff.p + n_store, (uint32_t)j0, S.chars, own_chars_at, S.seq.p)))) return rc;
					if (nj > j1 && (rc = p->gpu(mcom_unpack_contigs(p->ctx, p->d_cbits.p + p->total_words, cw_t.p + j1, S.soff.p + n_store + j1, (uint32_t)(nj - j1), own_chars_at + own_chars, S.chars + tot[1], S.seq.p)))) return rc;
				}
				if ((rc = words_in_place ? MCOM_OK : pack_words(S.soff.p + n_store, cw_t.p, nj, tw2, p->d_cbits.p + p->total_words)) ||
				    (rc = p->hipc(hipMemsetAsync(p->d_cbits.p + p->total_words + tw2, 0, 16, p->stream), "clear")) ||
				    (rc = p->gpu(mcom_offsets_append(p->ctx, cw_t.p, nj, p->total_words, p->d_coff_words.p + n_store)))) return rc;
				p->total_words += tw2;
				lap("t_cb_pack");
			}
			// next list: the merged contigs in claiming order, then the untouched ones in their order (cp_cluster, :397-434)
			uint64_t nk = 0;
			if (!ord2.reserve(nn + 1)) return p->fail(MCOM_E_NOMEM, "contig list");
			if ((rc = p->gpu(mcom_order_next(p->ctx, lst, n, d_flag.p, (uint32_t)n_store, nj, ord2.p, &nk)))) return rc;
			if (nk != nkeep) return p->fail(MCOM_E_ARG, "merge round: %llu contigs unclaimed but %zu expected", (unsigned long long)nk, nkeep);
			ord.swap(ord2); listed = true;
			first_new = (uint32_t)n_store; n_new = (uint32_t)nj;
			S.n = n_store + nj; S.members += tot[0]; S.chars += tot[1]; S.nrec += tn;
			n_live = nn;
			lap("t_cb_copy");
			p->stat["t_gpu"] += busy_now(p) - tg;
		} else packed_ready = packed_ready && n != 0;
		p->stat["merge_rounds"] += 1;
		const long tot = (long)n_live;
		if (std::labs(pre - tot) < 100) break;                                              // :625
		pre = tot;
	}
	// the list becomes an ordinary set (the one copy left of cp_cluster's): it stays on the device, for Stage 2 and beyond; the host
	// learns the offsets when Stage 2 asks for them
	p->maxlen = maxlen;
	p->stat["contigs_combine"] = (double)n_live;
	p->stat["store_contigs"] = (double)S.n; p->stat["store_chars"] = (double)S.chars;
	if (listed && n_live) {
		DevSet F;
		uint64_t t2[2] = {0, 0}, tw = 0;
		if (!F.seq.reserve(chars0 + 16) || !F.mem.reserve(members0 + 1) || !F.soff.reserve(n_live + 1) || !F.moff.reserve(n_live + 1)) return p->fail(MCOM_E_NOMEM, "contig set");
		if ((rc = p->gpu(mcom_contigs_gather(p->ctx, S.seq.p, S.soff.p, S.mem.p, S.moff.p, ord.p, n_live, F.seq.p, F.soff.p, F.mem.p, F.moff.p, t2)))) return rc;
		F.n = n_live; F.chars = t2[0]; F.members = t2[1];
		if (F.members != members0) return p->fail(MCOM_E_ARG, "merge rounds lost members: %llu of %llu", (unsigned long long)F.members, (unsigned long long)members0);
		if (packed_ready) {
			if (!p->d_coff_words_alt.reserve(n_live + 1) || !p->d_clen_alt.reserve(n_live + 1)) return p->fail(MCOM_E_NOMEM, "contig layout");
			if ((rc = p->gpu(mcom_contig_layout(p->ctx, F.soff.p, n_live, p->d_coff_words_alt.p, p->d_clen_alt.p, &tw)))) return rc;
			if (!p->d_cbits_alt.reserve(tw + 2)) return p->fail(MCOM_E_NOMEM, "packed contigs");
			if ((rc = p->gpu(mcom_pack_contigs_merged(p->ctx, F.seq.p, F.soff.p, p->d_coff_words_alt.p, (uint32_t)n_live, tw, 0, p->d_cbits.p, p->d_coff_words.p, ord.p, p->d_cbits_alt.p)))) return rc;
			if ((rc = p->sync("packed set"))) return rc;
			p->d_cbits.swap(p->d_cbits_alt); p->d_coff_words.swap(p->d_coff_words_alt); p->d_clen.swap(p->d_clen_alt);
			p->total_words = tw;
		}
		p->dC.swap(F);
	} else {
		S.nrec = 0;
		p->dC.swap(S);
	}
	p->dC_valid = true;
	p->cbits_for_dC = packed_ready && p->dC.n != 0;                          // Stage 2 takes the packed set as it is
	p->hostC_valid = false; p->host_off_valid = false;
	lap("t_cb_download");
	p->join_sg();
	lap("t_cb_join");
	const bool on_device = p->comm && !p->sg_gathered;                          // (multi-GPU: the ranks' parts meet on the device, the list is there already)
	if (p->comm && (rc = dist_gather_sg(p))) return rc;
	lap("t_cb_sg");
	// the singleton list goes up on the copy stream while Stage 2 packs the contigs and builds its index
	p->sg_uploaded = false;
	if (on_device && p->sg_live_valid) {
		if ((rc = p->hipc(hipEventRecord(p->ev_sg, p->stream), "event"))) return rc;
		p->sg_uploaded = true;
	} else if (!p->sg.empty() && p->sg_pin.size() == p->sg.size() && p->d_sg_live.reserve(p->sg.size())) {
		if ((rc = p->hipc(hipMemcpyAsync(p->d_sg_live.p, p->sg_pin.data(), p->sg.size() * 4, hipMemcpyHostToDevice, p->copy_stream), "upload singletons")) ||
		    (rc = p->hipc(hipEventRecord(p->ev_sg, p->copy_stream), "event"))) return rc;
		p->n_sg_live = p->sg.size(); p->sg_live_valid = true; p->sg_uploaded = true;
	}
	p->raw_flags_valid = false;
	p->sg_flag.assign(p->sg.size(), 0); p->sg_flag_zero = true;                             // preprocess.c:182
	p->stage2_uploaded = false;
	p->screen_clear = false;
	p->stat["t_combine"] += busy_now(p) - t0;
	return MCOM_OK;
}

// ----------------------------------------------------------------------------------------------------
// updateSingle                                                                 preprocess.c:243-255
// ----------------------------------------------------------------------------------------------------
static int update_single_impl(mcomh_pipeline *p);
extern "C" int mcomh_update_single(mcomh_pipeline *p) { return p ? stage_exit(p, update_single_impl(p)) : MCOM_E_ARG; }
static int update_single_impl(mcomh_pipeline *p)
{
	if (!p) return MCOM_E_ARG;
	p->join_sg();
	if (p->sg_next_valid) {                                             // compacted on the device when the pass ended
		p->sg.swap(p->sg_next); p->sg_next_valid = false;
		p->raw_flags_valid = false;
		p->sg_flag.assign(p->sg.size(), 0); p->sg_flag_zero = true;
		return MCOM_OK;
	}
	ensure_sg_flag(p);
	if (!p->sg_uploaded) p->sg_live_valid = false;
	p->sg_uploaded = false;
	const size_t n = p->sg.size();
	if (p->sg_flag.size() != n) { p->sg_flag.assign(n, 0); p->sg_flag_zero = true; return MCOM_OK; }
	if (p->sg_flag_zero) return MCOM_OK;                                 // cleared and untouched (the state after combine_cluster)
	// nothing flagged (the state after combine_cluster): nothing to compact; eight flags per test
	const uint8_t *f = p->sg_flag.data();
	size_t i = 0;
	for (; i + 8 <= n; i += 8) { uint64_t w8; memcpy(&w8, f + i, 8); if (w8) break; }
	if (i + 8 > n) { while (i < n && !f[i]) ++i; if (i == n) return MCOM_OK; }
	// the first flagged entry is at or behind i: compact [i, n) in chunks (count, then copy to each chunk's place)
	const size_t head = i;
	const int nt = std::max(1, std::min(p->host_threads, 16));
	std::vector<size_t> cnt((size_t)nt + 1, 0);
	parallel_for(nt, n - head, [&](int t, size_t b, size_t e) { size_t c = 0; for (size_t q = head + b; q < head + e; ++q) c += !f[q]; cnt[(size_t)t + 1] = c; });
	for (int t = 0; t < nt; ++t) cnt[(size_t)t + 1] += cnt[(size_t)t];
	const size_t nn = head + cnt[(size_t)nt];
	U32Pooled out(nn);
	memcpy(out.data(), p->sg.data(), head * 4);
	const uint32_t *src = p->sg.data();
	parallel_for(nt, n - head, [&](int t, size_t b, size_t e) { uint32_t *dst = out.data() + head + cnt[(size_t)t]; for (size_t q = head + b; q < head + e; ++q) if (!f[q]) *dst++ = src[q]; });
	p->sg.swap(out);
	p->sg_live_valid = false;                                           // the list on the device is the uncompacted one
	p->sg_flag.assign(nn, 0); p->sg_flag_zero = true;
	return MCOM_OK;
}

// The contig set is built and kept on the device; whoever wants it on the host (stage dumps, accessors, the stream
// writer) gets a copy here.  wait_data = false: the offsets are enough (Stage 2 needs the contig lengths).
static int ensure_host_contigs(P *p, bool wait_data)
{
	if (p->hostC_valid || (!wait_data && p->host_off_valid)) return MCOM_OK;
	if (!p->dC_valid) return p->fail(MCOM_E_ARG, "no contig set");
	DevSet &D = p->dC; ContigSet &C = p->C;
	int rc;
	if (!p->host_off_valid) {
		C.moff.assign(D.n + 1, 0); C.roff.assign(D.n + 1, 0);
		if (D.n && ((rc = p->d2h(C.roff.data(), D.soff.p, D.n + 1, "copy offsets")) || (rc = p->d2h(C.moff.data(), D.moff.p, D.n + 1, "copy offsets")) ||
		            (rc = p->sync("copy offsets")))) return rc;
		p->host_off_valid = true;
	}
	if (!wait_data) return MCOM_OK;
	if (!C.mem.resize(D.members) || !C.ref.resize(D.chars)) return p->fail(MCOM_E_NOMEM, "contig set");
	if (D.n && ((rc = p->d2h((uint8_t*)C.ref.data(), D.seq.p, D.chars, "copy contigs")) || (rc = p->d2h(C.mem.data(), D.mem.p, D.members, "copy members")) ||
	            (rc = p->sync("copy contig set")))) return rc;
	p->hostC_valid = true;
	return MCOM_OK;
}

// Folds the pending appends of passes 1..m into the member lists, on the device (mcom_members_finalize): the reference
// sorts a contig at the start of every scan and appends behind it, so after m passes contig c holds
//     stable_sort(C(c) + P_1(c) + ... + P_{m-1}(c)) + P_m(c).
static int materialize(P *p)
{
	const size_t m = p->pend.size();
	if (!m) return MCOM_OK;
	if (!p->dC_valid) return p->fail(MCOM_E_ARG, "no contig set on the device");
	const double t0 = busy_now(p);
	DevSet &D = p->dC;
	std::vector<const uint32_t*> ac(m); std::vector<const uint64_t*> am(m); std::vector<uint64_t> an(m);
	for (size_t i = 0; i < m; ++i) { ac[i] = p->pend[i].contig.p; am[i] = p->pend[i].member.p; an[i] = p->pend[i].n; }
	int kb = 2; while ((1ull << kb) < 4 * std::max<uint64_t>(p->maxlen, 2 * (uint64_t)p->L) + 4) ++kb;
	DevBuf<uint64_t> mem2, moff2;
	if (!mem2.reserve(D.members + p->n_pending + 1) || !moff2.reserve(D.n + 2)) return p->fail(MCOM_E_NOMEM, "member lists");
	int rc = p->gpu(mcom_members_finalize(p->ctx, D.mem.p, D.moff.p, D.n, D.members, ac.data(), am.data(), an.data(), (int)m, kb, mem2.p, moff2.p));
	if (rc) return rc;
	D.mem.swap(mem2); D.moff.swap(moff2);
	D.members += p->n_pending;
	p->pend.clear(); p->n_pending = 0;
	p->hostC_valid = false; p->host_off_valid = false;
	p->stat["t_ra_materialize"] += busy_now(p) - t0;
	return MCOM_OK;
}

// ---- bins longer than maxsearch: replay of their members in visiting order -----------------------------------
// kthread_hash_realign.c:374-435 with findpos / remove of bbhashdict.c:33-67: a visit scans the last maxsearch LIVE
// entries of the bin; the reads it claims leave every bin after the visit.  Reads that sit in no long bin are
// unaffected (all of their bins are scanned completely) and are settled by the kernel's minimum; for the members of
// long bins the kernel hands back every tuple they pass and the visits are replayed here, in key order.
struct Fenwick {
	std::vector<uint32_t> t;
	void init(size_t n) { t.assign(n + 1, 0); }
	void add(size_t i) { for (++i; i < t.size(); i += i & (~i + 1)) ++t[i]; }
	uint32_t prefix(size_t i) const { uint32_t s = 0; for (; i; i -= i & (~i + 1)) s += t[i]; return s; }   // removed among [0, i)
};
static int realign_big_bins(P *p, const mcom_dicts *dicts, const uint64_t *d_sgbits, const uint8_t *d_flag, size_t n_sg, size_t nc, int thr,
                            uint64_t *d_claim, uint64_t *d_st)
{
	int rc, nd = 0; uint32_t nk[16], mb[16];
	mcom_dicts_info(dicts, &nd, nk, mb);
	DevBuf<uint32_t> d_bs; DevBuf<uint8_t> d_mark;
	if (!d_bs.reserve((size_t)nd * n_sg) || !d_mark.reserve(n_sg)) return p->fail(MCOM_E_NOMEM, "long bins");
	if ((rc = p->gpu(mcom_dicts_bigbins(p->ctx, dicts, d_sgbits, p->maxsearch, d_bs.p, d_mark.p)))) return rc;
	std::vector<uint8_t> mark(n_sg);
	std::vector<uint32_t> bs((size_t)nd * n_sg);
	if ((rc = p->d2h(mark.data(), d_mark.p, n_sg, "copy marks")) || (rc = p->d2h(bs.data(), d_bs.p, (size_t)nd * n_sg, "copy bins")) || (rc = p->sync("long bins"))) return rc;
	std::vector<uint32_t> M;                                                           // the marked singletons, ascending
	for (size_t i = 0; i < n_sg; ++i) if (mark[i]) M.push_back((uint32_t)i);
	p->stat["big_bin_reads"] += (double)M.size();
	// every tuple the marked singletons pass
	DevBuf<uint64_t> d_tup;
	uint64_t cap = (1ull << 20) + 64ull * M.size(), nt = 0;
	for (;;) {
		if (!d_tup.reserve(2 * cap)) return p->fail(MCOM_E_NOMEM, "tuples");
		if ((rc = p->gpu(mcom_realign_pass_tuples(p->ctx, p->d_cix_keys.p, p->cix_geom, d_sgbits, d_flag, d_mark.p, n_sg, p->d_cbits.p, p->d_coff_words.p,
		                                          p->d_woff.p, (uint32_t)nc, p->L, p->numdict, thr, d_claim, d_st, d_tup.p, cap, &nt)))) return rc;
		if (nt <= cap) break;
		cap = nt + (nt >> 3);
	}
	if (p->comm) {
		// multi-GPU: this rank saw the tuples against ITS contigs; the unmarked singletons' claims are MIN-reduced, the marked
		// ones' tuples are put together, and every rank replays the same list
		if ((rc = dist_min_claims(p, d_claim, n_sg))) return rc;
		const int R = p->world;
		std::vector<uint64_t> all, first(R), cnt(R);
		if ((rc = gather_host(p, &nt, 1, all))) return rc;
		uint64_t total = 0;
		for (int q = 0; q < R; ++q) { first[q] = 2 * total; cnt[q] = 2 * all[q]; total += all[q]; }
		DevBuf<uint64_t> d_all;
		if (!d_all.reserve(2 * total + 2)) return p->fail(MCOM_E_NOMEM, "tuples");
		if ((rc = gatherv(p, d_all.p, first, cnt, d_tup.p))) return rc;
		d_tup.swap(d_all); nt = total;
	}
	std::vector<std::pair<uint64_t, uint64_t>> tup(nt);
	static_assert(sizeof(std::pair<uint64_t, uint64_t>) == 16, "tuple layout");
	if (nt && ((rc = p->d2h((uint64_t*)tup.data(), d_tup.p, 2 * nt, "copy tuples")) || (rc = p->sync("tuples")))) return rc;
	p->stat["big_bin_tuples"] += (double)nt;
	std::sort(tup.begin(), tup.end());
	// the long bins: members ascending (the order of read_id inside a bin), one tree of removed positions each
	auto dense = [&](uint32_t sg) { return (size_t)(std::lower_bound(M.begin(), M.end(), sg) - M.begin()); };
	struct Bin { std::vector<uint32_t> mem; Fenwick gone; };
	std::vector<Bin> bins;
	std::vector<uint32_t> bin_of((size_t)nd * M.size(), UINT32_MAX), pos_of((size_t)nd * M.size(), 0);
	for (int l = 0; l < nd; ++l) {
		std::vector<std::pair<uint32_t, uint32_t>> v;                                   // (bin start, singleton)
		for (uint32_t sg : M) { const uint32_t b = bs[(size_t)l * n_sg + sg]; if (b != UINT32_MAX) v.emplace_back(b, sg); }
		std::sort(v.begin(), v.end());
		for (size_t a = 0; a < v.size();) {
			size_t e = a; while (e < v.size() && v[e].first == v[a].first) ++e;
			Bin B; B.mem.reserve(e - a);
			for (size_t q = a; q < e; ++q) {
				const size_t d = dense(v[q].second);
				bin_of[(size_t)l * M.size() + d] = (uint32_t)bins.size(); pos_of[(size_t)l * M.size() + d] = (uint32_t)(q - a);
				B.mem.push_back(v[q].second);
			}
			B.gone.init(B.mem.size());
			bins.push_back(std::move(B));
			a = e;
		}
	}
	std::vector<uint64_t> won(M.size(), UINT64_MAX);
	std::vector<size_t> batch;
	const uint32_t ms = (uint32_t)p->maxsearch;
	for (size_t a = 0; a < tup.size();) {
		size_t e = a; while (e < tup.size() && tup[e].first == tup[a].first) ++e;     // one visit of one bin
		const int l = (int)(tup[a].first & 15);
		batch.clear();
		for (size_t q = a; q < e; ++q) {
			const size_t d = dense((uint32_t)tup[q].second);
			if (won[d] != UINT64_MAX) continue;                                           // claimed earlier: gone from the bin (or not re-claimed, :395)
			const uint32_t b = bin_of[(size_t)l * M.size() + d];
			if (b != UINT32_MAX) {
				const Bin &B = bins[b];
				const size_t pos = pos_of[(size_t)l * M.size() + d];
				const uint32_t above = (uint32_t)(B.mem.size() - 1 - pos), gone_above = B.gone.prefix(B.mem.size()) - B.gone.prefix(pos + 1);
				if (above - gone_above >= ms) { p->stat["big_bin_deferred"] += 1; continue; }   // deeper than the scan reaches (:388)
			}
			batch.push_back(d);
		}
		for (size_t d : batch) {                                                          // :420-435, after the visit
			won[d] = tup[a].first;
			for (int l1 = 0; l1 < nd; ++l1) {
				const uint32_t b = bin_of[(size_t)l1 * M.size() + d];
				if (b != UINT32_MAX) bins[b].gone.add(pos_of[(size_t)l1 * M.size() + d]);
			}
		}
		a = e;
	}
	size_t nw = 0; for (uint64_t w : won) nw += w != UINT64_MAX;
	p->stat["big_bin_claims"] += (double)nw;
	if (M.empty()) return MCOM_OK;
	DevBuf<uint32_t> d_idx; DevBuf<uint64_t> d_val;
	if (!d_idx.reserve(M.size()) || !d_val.reserve(M.size())) return p->fail(MCOM_E_NOMEM, "claims of long bins");
	if ((rc = p->h2d(d_idx.p, M.data(), M.size(), "upload claims")) || (rc = p->h2d(d_val.p, won.data(), M.size(), "upload claims"))) return rc;
	if ((rc = p->gpu(mcom_claims_patch(p->ctx, d_claim, d_idx.p, d_val.p, M.size())))) return rc;
	return p->sync("claims of long bins");
}

// the packed form of the contig set in dC (combine_cluster leaves the packed form of its last set behind)
static int ensure_packed_contigs(P *p)
{
	if (p->cbits_for_dC) return MCOM_OK;
	const size_t nc = p->dC.n;
	uint64_t tw = 0;
	int rc;
	if (!p->d_coff_words.reserve(nc + 1) || !p->d_clen.reserve(nc + 1)) return p->fail(MCOM_E_NOMEM, "contig layout");
	if ((rc = p->gpu(mcom_contig_layout(p->ctx, p->dC.soff.p, nc, p->d_coff_words.p, p->d_clen.p, &tw)))) return rc;
	p->total_words = tw;
	if (!p->d_cbits.reserve(tw + 2)) return p->fail(MCOM_E_NOMEM, "packed contigs");
	if (nc && ((rc = p->hipc(hipMemsetAsync(p->d_cbits.p, 0, (tw + 2) * 8, p->stream), "clear")) ||
	           (rc = p->gpu(mcom_pack_contigs(p->ctx, p->dC.seq.p, p->dC.soff.p, p->d_coff_words.p, (uint32_t)nc, tw, p->d_cbits.p))))) return rc;
	p->cbits_for_dC = true;
	return MCOM_OK;
}

// ----------------------------------------------------------------------------------------------------
// realign_hash: one Stage-2 pass                                    kthread_hash_realign.c:569-594
// ----------------------------------------------------------------------------------------------------
static int realign_hash_impl(mcomh_pipeline *p, int thr, long *cluster_reads);
extern "C" int mcomh_realign_hash(mcomh_pipeline *p, int thr, long *cluster_reads) { return p ? stage_exit(p, realign_hash_impl(p, thr, cluster_reads)) : MCOM_E_ARG; }
static int realign_hash_impl(mcomh_pipeline *p, int thr, long *cluster_reads)
{
	if (!p) return MCOM_E_ARG;
	const double t0 = busy_now(p);
	p->join_sg();
	if (p->cls_failed) return p->fail(MCOM_E_HIP, "the read classes did not arrive from the device: the class lists are incomplete");
	if (!p->dC_valid) return p->fail(MCOM_E_ARG, "Stage 2 needs the contig set of kt_for_bucket / combine_cluster on the device");
	const bool sg_sent_up = p->sg_uploaded;                         // (mcomh_update_single clears the flag: it only vouches for the list of combine_cluster)
	{ const double tu = busy_now(p); mcomh_update_single(p); p->stat["t_ra_update"] += busy_now(p) - tu; }                // preprocess.c:203
	const size_t nc = p->dC.n, n_sg = p->sg.size();
	int rc;
	if (!p->stage2_uploaded && !p->window_scan && !p->screen_clear && n_sg && p->sg_live_valid && p->n_sg_live == n_sg && sg_sent_up && !p->early.on && p->overlap_screen) {
		// singletons' rows and the dictionary screen on the copy stream (behind the upload of the singleton list, which went there),
		// while this thread builds the contig index on the main stream
		P::Early &E = p->early;
		bool ok = true;
		if (!p->ctx2) { ok = mcom_create(&p->ctx2, p->device, p->copy_stream) == MCOM_OK; if (ok && p->prof_on) (void)mcom_prof_enable(p->ctx2, 1); }
		if (ok && !p->ev_early) ok = hipEventCreateWithFlags(&p->ev_early, hipEventDisableTiming) == hipSuccess;
		if (ok && E.sgbits.reserve(n_sg * p->W)) {
			E.sg.swap(p->d_sg_live); p->sg_live_valid = false;
			if (mcom_gather_rows(p->ctx2, p->d_packed.p, E.sg.p, n_sg, p->L, E.sgbits.p) == MCOM_OK &&
			    hipEventRecord(p->ev_early, p->copy_stream) == hipSuccess &&
			    mcom_dicts_screen_begin_shared(p->ctx2, E.sgbits.p, n_sg, p->L, p->numdict, p->maxsearch, p->comm ? p->world : 1, p->comm ? p->rank : 0) == MCOM_OK) { E.on = true; E.n_sg = n_sg; p->stat["early_screen"] += 1; }
			else { (void)hipStreamSynchronize(p->copy_stream); p->d_sg_live.swap(E.sg); p->sg_live_valid = true; }   // as before
		}
	}
	if (!p->stage2_uploaded) {                      // contig consensus strings do not change during Stage 2
		if ((rc = ensure_packed_contigs(p))) return rc;            // the set is on the device: laid out and packed there
		if (!p->d_woff.reserve(nc + 2)) return p->fail(MCOM_E_NOMEM, "window offsets");
		uint64_t nwin = 0, mlen = 0;
		if ((rc = p->gpu(mcom_window_layout(p->ctx, p->dC.soff.p, nc, p->L, p->d_woff.p, &nwin, &mlen)))) return rc;
		p->n_windows = nwin; p->maxlen = std::max(p->maxlen, mlen);
		if (!p->window_scan) {
			// Multi-GPU: ONE index over all contigs, shared out BY KEY (include/mcom.h, mcom_cindex_plan_shared): a rank makes the entries of
			// its range of the replicated contig set (equal shares of the windows), the entries travel to the owner of their key, and
			// every rank places what it received into its share of the table.  A pass then looks every singleton up on every rank, but
			// only the keys of the rank's share: 1 / R of the lookups and verifications each, claim keys MIN-reduced (dist_min_claims).
			const int R = p->comm ? p->world : 1, me = p->comm ? p->rank : 0;
			uint32_t c0 = 0, c1 = (uint32_t)nc; uint64_t nwin_mine = p->n_windows;
			if (p->comm && nc) {
				std::vector<uint64_t> hw(nc + 1);
				if ((rc = p->d2h(hw.data(), p->d_woff.p, nc + 1, "copy window offsets")) || (rc = p->sync("copy window offsets"))) return rc;
				auto bound = [&](int q) { return q >= p->world ? (size_t)nc : (size_t)(std::lower_bound(hw.begin(), hw.begin() + nc, p->n_windows * (uint64_t)q / p->world) - hw.begin()); };
				c0 = (uint32_t)bound(p->rank); c1 = (uint32_t)bound(p->rank + 1);
				if (p->rank == 0) c0 = 0;
				nwin_mine = hw[c1] - hw[c0];
			}
			p->cix_c0 = c0; p->cix_c1 = c1;
			uint64_t ne = 0, share = 0, nwords = 0, cap_mine = 0;
			if (mcom_cindex_plan_shared(p->n_windows, (uint32_t)nc, p->L, p->numdict, R, me, &ne, &share, &p->cix_geom, &nwords) ||
			    mcom_cindex_plan(nwin_mine, c1 - c0, p->L, p->numdict, &cap_mine, nullptr, nullptr)) return p->fail(MCOM_E_ARG, "contig index: %llu windows are more than one share of %d holds (2^32 positions, or 65 535 partitions of 12 000 lines = 2.75 G entries)", (unsigned long long)p->n_windows, R);
			DevBuf<uint32_t> keyA, keyB; DevBuf<uint64_t> slotA, slotB;
			const bool use_join = R == 1 && !p->stage2_table && p->maxthr <= 255;
			p->jn_entries = false; p->jn_deferred = false; p->jn_ndefer = 0;
			std::vector<uint64_t> cnt((size_t)R, 0);
			for (int attempt = 0;; ++attempt) {                                 // a repeat-rich set may need a larger extension area for its heavy keys
				if ((!use_join && !p->d_cix_keys.reserve(nwords)) || !keyA.reserve(cap_mine + 1) || !slotA.reserve(cap_mine + 1)) return p->fail(MCOM_E_NOMEM, "contig index");
				if ((rc = p->gpu(mcom_cindex_entries(p->ctx, p->d_cbits.p, p->d_coff_words.p, p->d_woff.p, (uint32_t)nc, c0, c1, p->L, p->numdict, p->cix_geom,
				                                     keyA.p, slotA.p, cap_mine + 1, cnt.data())))) return rc;
				uint64_t n_ent = cnt[0];
				if (R > 1) {
					// the exchange: entries to the owner of their key (4 + 8 bytes each, two all-to-alls)
					const double tx = now_ms(), bx0 = xbytes(p);
					std::vector<uint64_t> all, so(R), sb(R), ro(R), rb(R);
					if ((rc = gather_host(p, cnt.data(), R, all))) return rc;
					uint64_t a = 0, b = 0;
					for (int q = 0; q < R; ++q) { so[q] = a; sb[q] = cnt[q]; a += cnt[q]; ro[q] = b; rb[q] = all[(size_t)q * R + me]; b += rb[q]; }
					if (!keyB.reserve(std::max<uint64_t>(b, a) + 1) || !slotB.reserve(std::max<uint64_t>(b, a) + 1)) return p->fail(MCOM_E_NOMEM, "contig index");
					auto scaled = [&](const std::vector<uint64_t> &v, uint64_t m) { std::vector<uint64_t> o(v); for (auto &x : o) x *= m; return o; };
					if ((rc = alltoallv_dev(p, keyA.p, scaled(so, 4).data(), scaled(sb, 4).data(), keyB.p, scaled(ro, 4).data(), scaled(rb, 4).data())) ||
					    (rc = alltoallv_dev(p, slotA.p, scaled(so, 8).data(), scaled(sb, 8).data(), slotB.p, scaled(ro, 8).data(), scaled(rb, 8).data(), false))) return rc;
					p->stat["t_x_cindex"] += now_ms() - tx; p->stat["b_x_cindex"] += xbytes(p) - bx0; p->stat["x_cindex_entries"] += (double)(a - cnt[me]);
					if (!keyA.reserve(b + 1) || !slotA.reserve(b + 1)) return p->fail(MCOM_E_NOMEM, "contig index");
					n_ent = b;
					rc = mcom_cindex_place(p->ctx, keyB.p, slotB.p, n_ent, 0, keyA.p, slotA.p, p->L, p->numdict, p->cix_geom, p->d_cix_keys.p, nwords);
				} else if (use_join) {
					// one GPU (round 5): the entries are sorted by partition and kept; the first pass joins them with the singletons' keys and no
					// table is placed (the pass builds it from these arrays after all when the join does not take the input)
					if (!keyB.reserve(n_ent + 1) || !slotB.reserve(n_ent + 1) || !p->jn_pstart.reserve((size_t)(p->cix_geom & 0xFFFFu) + 2)) return p->fail(MCOM_E_NOMEM, "contig index");
					rc = mcom_cindex_partition(p->ctx, keyA.p, slotA.p, n_ent, 1, keyB.p, slotB.p, p->L, p->numdict, p->cix_geom, p->jn_pstart.p, &p->jn_ek, &p->jn_es);
					if (!rc) { p->jn_keyA.swap(keyA); p->jn_keyB.swap(keyB); p->jn_slotA.swap(slotA); p->jn_slotB.swap(slotB); p->jn_entries = true; p->jn_nwords = nwords; }
				} else {
					if (!keyB.reserve(n_ent + 1) || !slotB.reserve(n_ent + 1)) return p->fail(MCOM_E_NOMEM, "contig index");
					rc = mcom_cindex_place(p->ctx, keyA.p, slotA.p, n_ent, 1, keyB.p, slotB.p, p->L, p->numdict, p->cix_geom, p->d_cix_keys.p, nwords);
				}
				// (every rank must come to the same decision: a share that needs more room makes all of them build again)
				uint64_t again = rc == MCOM_E_OVERFLOW ? 1 : 0;
				if (R > 1 && (rc == MCOM_OK || rc == MCOM_E_OVERFLOW)) { int rc2 = allreduce_host(p, &again, 1, 2); if (rc2) return rc2; }
				if (rc != MCOM_OK && rc != MCOM_E_OVERFLOW) return p->gpu(rc);
				p->stat["cix_entries"] += (double)n_ent;
				if (!again) break;
				if (attempt == 3) return p->fail(MCOM_E_OVERFLOW, "contig index: the extension area stays too small");
				nwords += std::max<uint64_t>(nwords / 4, 8 * (share / 7 + 1024) / (attempt < 2 ? 4 : 1));
				p->stat["cix_rebuilds"] += 1;
			}
			if (!p->jn_entries) p->stat["cix_slots"] += (double)nwords;
		}
		p->stage2_uploaded = true;
	}
	p->stat["t_ra_setup"] += busy_now(p) - t0;
	p->stat["passes"] += 1;
	p->stat["windows"] += (double)p->n_windows;
	// every contig is re-sorted at the start of its scan (:318), but the scan itself never looks at the members: the sorts
	// of all passes are folded into materialize()
	if (n_sg) {
		const double tg = busy_now(p);
		DevBuf<uint32_t> d_sg; DevBuf<uint64_t> d_sgbits, d_claim; DevBuf<uint8_t> d_flag;
		if (!d_sg.reserve(n_sg) || !d_sgbits.reserve(n_sg * p->W) || !d_claim.reserve(n_sg) || !d_flag.reserve(n_sg)) return p->fail(MCOM_E_NOMEM, "singleton buffers");
		const bool early = p->early.on && p->early.n_sg == n_sg;
		if (early) {                                                                         // gathered on the copy stream beside the index build
			d_sg.swap(p->early.sg); d_sgbits.swap(p->early.sgbits);
			if ((rc = p->hipc(hipStreamWaitEvent(p->stream, p->ev_early, 0), "wait"))) return rc;
		} else if (p->sg_live_valid && p->n_sg_live == n_sg) {                                // left by the pass before, or sent up beside the set-up
			d_sg.swap(p->d_sg_live); p->sg_live_valid = false;
			if ((rc = p->hipc(hipStreamWaitEvent(p->stream, p->ev_sg, 0), "wait"))) return rc;
		} else if ((rc = p->h2d(d_sg.p, p->sg.data(), n_sg, "upload singletons"))) return rc;
		if (!early && (rc = p->gpu(mcom_gather_rows(p->ctx, p->d_packed.p, d_sg.p, n_sg, p->L, d_sgbits.p)))) return rc;   // singleRead2bitset
		if ((rc = p->gpu(mcom_poly_filter(p->ctx, d_sgbits.p, p->d_nmask.p, d_sg.p, n_sg, p->L, thr, d_flag.p)))) return rc;
		PinVec<uint8_t> &pf = p->raw_flags;
		p->raw_flags_valid = false;
		if (!pf.resize(n_sg)) return p->fail(MCOM_E_NOMEM, "flags");
		DevBuf<uint32_t> d_np; PinVec<uint32_t> h_np;                                        // near-poly singletons of this pass: {index, read, flag}
		const uint32_t np_cap = 1u << 16;
		if (!d_np.reserve(3 * (size_t)np_cap + 1) || !h_np.resize(3 * (size_t)np_cap + 1)) return p->fail(MCOM_E_NOMEM, "flag list");
		// constructdictionary_realign: the read-driven pass needs the dictionaries only where a bin is cut at maxsearch;
		// a screen with hashed counters proves (nearly always) that none is
		mcom_dicts *dicts = nullptr;
		bool big = false;
		int may_exceed = 1;
		DevBuf<uint64_t> d_st;
		if (!d_st.reserve(4)) return p->fail(MCOM_E_NOMEM, "pass counters");
		bool joined = false;                                                                 // this pass's claims are made (join or deferred tuples)
		if (p->jn_entries) {
			// the first pass of a Stage 2 on one GPU: partition-local join of the sorted index entries with the singletons' keys
			const uint64_t dcap = std::max<uint64_t>((uint64_t)1 << 20, n_sg / 2);
			int status = 1; uint64_t nd_ = 0;
			if (!p->jn_defer.reserve(2 * dcap + 2)) return p->fail(MCOM_E_NOMEM, "deferred candidates");
			if ((rc = p->gpu(mcom_realign_join(p->ctx, p->cix_geom, p->jn_ek, p->jn_es, p->jn_pstart.p, d_sgbits.p, d_flag.p, d_sg.p, n_sg, p->d_cbits.p, p->d_coff_words.p, p->d_woff.p,
			                                   (uint32_t)nc, p->L, p->numdict, thr, p->maxthr, p->maxsearch, d_claim.p, d_st.p, p->jn_defer.p, dcap, &nd_, &status)))) return rc;
			if (!status) {
				joined = true; p->jn_deferred = true; p->jn_ndefer = nd_; p->screen_clear = true;
				p->stat["join_passes"] += 1; p->stat["join_deferred"] += (double)nd_;
			} else {
				// not for the join (a dictionary bin may exceed maxsearch, a partition too full for LDS ...): the table from the same entries
				p->stat["join_fallbacks"] += 1;
				uint64_t nwords = p->jn_nwords;
				for (int attempt = 0;; ++attempt) {
					if (!p->d_cix_keys.reserve(nwords)) return p->fail(MCOM_E_NOMEM, "contig index");
					rc = mcom_cindex_assemble(p->ctx, p->jn_ek, p->jn_es, p->jn_pstart.p, p->L, p->numdict, p->cix_geom, p->d_cix_keys.p, nwords);
					if (rc != MCOM_E_OVERFLOW) break;
					if (attempt == 3) return p->fail(MCOM_E_OVERFLOW, "contig index: the extension area stays too small");
					nwords += std::max<uint64_t>(nwords / 4, 1 << 16);
					p->stat["cix_rebuilds"] += 1;
				}
				if (rc) return p->gpu(rc);
				p->stat["cix_slots"] += (double)nwords;
			}
			p->jn_entries = false;                                                            // either way the sorted entries have done their work
			{ DevBuf<uint32_t> a, b; DevBuf<uint64_t> c, d; a.swap(p->jn_keyA); b.swap(p->jn_keyB); c.swap(p->jn_slotA); d.swap(p->jn_slotB); }
			p->jn_ek = nullptr; p->jn_es = nullptr;
		} else if (p->jn_deferred) {
			// a later pass: every candidate that can pass at this threshold was kept by the first pass
			if (!p->jn_map.reserve(p->n + 1)) return p->fail(MCOM_E_NOMEM, "read map");
			if ((rc = p->gpu(mcom_realign_deferred(p->ctx, p->jn_defer.p, p->jn_ndefer, d_sg.p, d_flag.p, n_sg, p->jn_map.p, p->n, thr, d_claim.p, d_st.p)))) return rc;
			joined = true;
			p->stat["join_passes"] += 1;
		}
		if (joined) may_exceed = 0;
		// (singletons only leave between the passes of one Stage 2, so a bin never grows: once the screen has proved that none
		// exceeds maxsearch, it holds for the later passes too)
		if (joined) {
			if (early) { p->early.on = false; int dummy = 0; (void)mcom_dicts_screen_end(p->ctx2, &dummy); }
		} else if (early) {
			p->early.on = false;
			if ((rc = mcom_dicts_screen_end(p->ctx2, &may_exceed))) return p->fail(rc, "%s", mcom_last_error(p->ctx2));
		} else if (p->screen_clear) may_exceed = 0;
		else if (!p->window_scan && ((rc = p->gpu(mcom_dicts_screen_begin_shared(p->ctx, d_sgbits.p, n_sg, p->L, p->numdict, p->maxsearch, p->comm ? p->world : 1, p->comm ? p->rank : 0))) ||
		                              (rc = p->gpu(mcom_dicts_screen_end(p->ctx, &may_exceed))))) return rc;
		if (p->comm && !p->window_scan && !p->screen_clear) {                                  // a rank screened the keys of its share: any of them may say "look"
			uint64_t any = (uint64_t)may_exceed;
			if ((rc = allreduce_host(p, &any, 1, 2))) return rc;
			may_exceed = any ? 1 : 0;
		}
		if (!may_exceed && !p->window_scan) p->screen_clear = true;
		if (may_exceed) {
			if ((rc = p->gpu(mcom_dicts_build(p->ctx, d_sgbits.p, n_sg, p->L, p->numdict, &dicts)))) return rc;
			int nd = 0; uint32_t nk[16], mb[16];
			mcom_dicts_info(dicts, &nd, nk, mb);
			for (int j = 0; j < nd; ++j) if (mb[j] > (uint32_t)p->maxsearch) { p->stat["big_bins"] += 1; big = true; }
			p->stat["dict_builds"] += 1;
		}
		if (p->window_scan)
			rc = p->gpu(mcom_realign_pass(p->ctx, dicts, d_sgbits.p, d_flag.p, p->d_cbits.p, p->d_coff_words.p, p->d_woff.p, (uint32_t)nc,
			                              p->n_windows, thr, p->maxsearch, d_claim.p, nullptr));
		else {
			rc = MCOM_OK;
			if (joined) {}
			else if (!rc && !big)
				rc = p->gpu(mcom_realign_pass_reads(p->ctx, p->d_cix_keys.p, p->cix_geom, d_sgbits.p, d_flag.p, nullptr,
				                                    n_sg, p->d_cbits.p, p->d_coff_words.p, p->d_woff.p, (uint32_t)nc, p->L, p->numdict, thr, d_claim.p, d_st.p));
			else if (!rc)
				rc = realign_big_bins(p, dicts, d_sgbits.p, d_flag.p, n_sg, nc, thr, d_claim.p, d_st.p);
			if (!rc && !big && p->comm) rc = dist_min_claims(p, d_claim.p, n_sg);
			PinVec<uint64_t> hst; hst.resize(3);
			if (!rc) rc = p->d2h(hst.data(), d_st.p, 3, "copy pass counters");
			if (!rc) rc = p->sync("realign pass");
			if (!rc) { p->stat["ra_lookups"] += (double)hst[0]; p->stat["ra_verified"] += (double)hst[1]; p->stat["ra_passing"] += (double)hst[2]; }
			p->stat["ra_singletons"] += (double)n_sg;
		}
		mcom_dicts_free(p->ctx, dicts);
		if (rc) return rc;
		// the appends in the order of the sequential scan (claim key ascending, singleton index descending, :388),
		// resolved on the device; they stay pending until somebody needs the member lists
		P::Appended app;
		if (!app.contig.reserve(n_sg) || !app.member.reserve(n_sg)) return p->fail(MCOM_E_NOMEM, "claim buffers");
		uint64_t nwon = 0;
		if ((rc = p->gpu(mcom_claims_resolve(p->ctx, d_claim.p, d_sg.p, n_sg, (uint32_t)nc, d_flag.p, app.contig.p, app.member.p, &nwon)))) return rc;
		app.n = (size_t)nwon;
		if ((rc = p->gpu(mcom_list_flagged(p->ctx, d_sg.p, d_flag.p, n_sg, d_np.p, np_cap, d_np.p + 3 * (size_t)np_cap)))) return rc;
		if ((rc = p->d2h(h_np.data(), d_np.p, 3 * (size_t)np_cap + 1, "copy flag list"))) return rc;
		if ((rc = p->d2h(pf.data(), d_flag.p, n_sg, "copy flags"))) return rc;
		{                                                                                    // updateSingle for the next pass, on the device
			uint64_t n_next = 0;
			if (!p->d_sg_live.reserve(n_sg)) return p->fail(MCOM_E_NOMEM, "singleton ids");
			if ((rc = p->gpu(mcom_compact_live(p->ctx, d_sg.p, d_flag.p, n_sg, p->d_sg_live.p, &n_next)))) return rc;
			p->sg_next.resize((size_t)n_next);
			if ((rc = p->d2h(p->sg_next.data(), p->d_sg_live.p, (size_t)n_next, "copy live singletons"))) return rc;
			p->n_sg_live = (size_t)n_next; p->sg_live_valid = true; p->sg_next_valid = true;
		}
		if ((rc = p->sync("realign pass"))) return rc;
		p->stat["t_gpu"] += busy_now(p) - tg;
		p->stat["t_ra_gpu"] += busy_now(p) - tg;
		const double tw0 = busy_now(p);
		{                                                                                    // bbhashdict.c:177-216, singleton order
			// sg_flag is made from the raw flags when somebody looks (ensure_sg_flag); the near-poly-A / -T reads (flag 1 / 2) are rare
			// and come as a list from the device, which only has to be put back into singleton order
			p->raw_flags_valid = true;
			const uint32_t nnp = h_np[3 * (size_t)np_cap];
			if (nnp <= np_cap) {
				std::vector<std::pair<uint32_t, std::pair<uint32_t, uint32_t>>> v(nnp);     // index -> (read, flag)
				for (uint32_t q = 0; q < nnp; ++q) v[q] = std::make_pair(h_np[3 * (size_t)q], std::make_pair(h_np[3 * (size_t)q + 1], h_np[3 * (size_t)q + 2]));
				std::sort(v.begin(), v.end());
				for (const auto &e : v) { if (e.second.second == 1) p->fpA.push_back(e.second.first); else p->fpT.push_back(e.second.first); }
			} else {
				const uint8_t *f = pf.data();
				for (size_t q = 0; q < n_sg; ++q) { if (f[q] == 1) p->fpA.push_back(p->sg[q]); else if (f[q] == 2) p->fpT.push_back(p->sg[q]); }
			}
		}
		// a pass that appends nothing still counts: its scan sorts every contig first (:318), the appends of the pass before
		// it included, and materialize() takes "the last pass" from the number of entries here
		p->n_pending += nwon; p->pend.push_back(std::move(app));
		p->stat["t_ra_append"] += busy_now(p) - tw0;
	} else p->pend.emplace_back();
	if (cluster_reads) *cluster_reads = (long)(p->dC.members + p->n_pending);
	p->stat["t_realign"] += busy_now(p) - t0;
	return MCOM_OK;
}

// ----------------------------------------------------------------------------------------------------
// the reference's loop control around the stages                             preprocess.c:141-233
// ----------------------------------------------------------------------------------------------------
static int run_stage2(P *p, FILE *f);

extern "C" int mcomh_stage2(mcomh_pipeline *p) { return p ? stage_exit(p, run_stage2(p, nullptr)) : MCOM_E_ARG; }

extern "C" int mcomh_pre_process(mcomh_pipeline *p)
{
	if (!p) return MCOM_E_ARG;
	int rc;
	if ((rc = mcomh_kt_for_reads(p))) return rc;
	if ((rc = mcomh_kt_for_bucket(p))) return rc;
	if ((rc = mcomh_combine_cluster(p))) return rc;
	return run_stage2(p, nullptr);
}

// ---- dump in the text format of the golden-fixture stage dumps (tests/golden/) ---------------------------------------------------
template <class V> static void dump_list(FILE *f, const char *name, const V &v)
{
	fprintf(f, "LIST %s %zu", name, v.size());
	for (uint32_t x : v) fprintf(f, " %u", x);
	fprintf(f, "\n");
}
static void dump_contigs(FILE *f, const char *stage, const ContigSet &C)
{
	fprintf(f, "CLUSTERS %s %zu\n", stage, C.n());
	for (size_t c = 0; c < C.n(); ++c) {
		fprintf(f, "C %zu %.*s", C.msize(c), (int)C.rsize(c), C.ref.data() + C.roff[c]);
		for (uint64_t q = C.moff[c]; q < C.moff[c + 1]; ++q) fprintf(f, " %" PRIu64, C.mem[q]);
		fprintf(f, "\n");
	}
}
static void dump_buckets(FILE *f, const char *name, const std::vector<mcom_mm128> &recs)
{
	std::vector<std::vector<mcom_mm128>> B(1 << NB_BITS);
	size_t tot = 0; int ne = 0;
	for (const mcom_mm128 &r : recs) { if (r.x == U64MAX && r.y == U64MAX) continue; auto &b = B[r.x & ((1 << NB_BITS) - 1)]; if (b.empty()) ++ne; b.push_back(r); ++tot; }
	fprintf(f, "BUCKETS %s %d %zu\n", name, ne, tot);
	for (size_t i = 0; i < B.size(); ++i) {
		if (B[i].empty()) continue;
		fprintf(f, "B %zu %zu", i, B[i].size());
		for (const mcom_mm128 &r : B[i]) fprintf(f, " %" PRIu64 " %" PRIu64, r.x, r.y);
		fprintf(f, "\n");
	}
}

static int run_stage2(P *p, FILE *f)
{
	p->join_sg();
	if (p->cls_failed) return p->fail(MCOM_E_HIP, "the read classes did not arrive from the device: the class lists are incomplete");
	long pre = 0; int pass = 0;
	for (int thr = p->e;; thr += p->step) {                                                 // preprocess.c:197-232
		if (thr > p->maxthr) break;
		std::vector<uint32_t> before;
		if (f) { ensure_sg_flag(p); for (size_t i = 0; i < p->sg.size(); ++i) if (!p->sg_flag[i]) before.push_back(p->sg[i]); }
		long cr = 0;
		int rc = mcomh_realign_hash(p, thr, &cr);
		if (rc) return rc;
		if (f) {
			fprintf(f, "STAGE realign %d thr %d\n", pass, thr);
			dump_list(f, "sg_in", before);
			ensure_sg_flag(p);
			fprintf(f, "SGFLAG %zu", p->sg.size());
			for (uint8_t v : p->sg_flag) fprintf(f, " %d", v ? 1 : 0);
			fprintf(f, "\n");
			dump_list(f, "fpA", p->fpA); dump_list(f, "fpT", p->fpT);
			if ((rc = materialize(p)) || (rc = ensure_host_contigs(p))) return rc;
			dump_contigs(f, "realign", p->C);
		}
		const long lim = (p->sg.size() > 1000000 && p->L >= 68) ? 10000 : 1000;
		++pass;
		if (cr - pre < lim) break;
		pre = cr;
	}
	const int rc = materialize(p);
	return rc ? rc : mcomh_update_single(p);
}

extern "C" int mcomh_dump_stages(mcomh_pipeline *p, const char *path)
{
	if (!p || !path) return MCOM_E_ARG;
	if (p->h_ascii.empty() && p->n) return p->fail(MCOM_E_ARG, "dump needs the reads on the host");
	FILE *f = fopen(path, "w");
	if (!f) return p->fail(MCOM_E_ARG, "cannot write %s", path);
	int rc = mcomh_kt_for_reads(p);
	if (rc) { fclose(f); return rc; }
	p->join_cls();
	if (!p->h_cls_valid && p->n) {
		if (!p->h_cls.resize(p->n + 8) || hipMemcpy(p->h_cls.data(), p->d_cls.p, p->n, hipMemcpyDeviceToHost) != hipSuccess) { fclose(f); return p->fail(MCOM_E_HIP, "copy classes"); }
		p->h_cls_valid = true;
	}
	const int L = p->L, W = p->W;
	fprintf(f, "PARAMS L %d k %d b %d rw %d e %d cbthr %d m %d n %zu\n", L, p->k, NB_BITS, p->rw, p->e, p->cbthr, p->m, p->n);
	fprintf(f, "STAGE reads\nREADS %zu\n", p->n);
	std::vector<uint64_t> h_packed(p->n * (size_t)W);
	if (p->n && (rc = p->hipc(hipMemcpy(h_packed.data(), p->d_packed.p, p->n * (size_t)W * 8, hipMemcpyDeviceToHost), "copy packed reads"))) { fclose(f); return rc; }
	std::string line((size_t)L, 'A');
	size_t nn = 0;
	for (size_t r = 0; r < p->n; ++r) {
		const uint8_t *src = p->h_ascii.data() + r * (size_t)L;
		if (p->h_cls[r] == MCOM_CLS_SKETCH) for (int i = 0; i < L; ++i) line[i] = ACGT[(h_packed[r * W + (i >> 5)] >> (2 * (i & 31))) & 3];
		else line.assign((const char*)src, (size_t)L);
		fprintf(f, "%s\n", line.c_str());
		if (memchr(src, 'N', (size_t)L)) ++nn;
	}
	fprintf(f, "NPOS %zu\n", nn);
	for (size_t r = 0; r < p->n; ++r) {
		const uint8_t *src = p->h_ascii.data() + r * (size_t)L;
		if (!memchr(src, 'N', (size_t)L)) continue;
		size_t c = 0; for (int i = 0; i < L; ++i) c += src[i] == 'N';
		fprintf(f, "N %zu %zu", r, c);
		for (int i = 0; i < L; ++i) if (src[i] == 'N') fprintf(f, " %d", i);
		fprintf(f, "\n");
	}
	dump_list(f, "allA", p->allA); dump_list(f, "allT", p->allT); dump_list(f, "allN", p->allN);
	dump_list(f, "fpA", p->fpA); dump_list(f, "fpT", p->fpT); dump_list(f, "fpN", p->fpN); dump_list(f, "Nfile", p->Nfile);
	{
		std::vector<mcom_mm128> rec(p->n);
		if (p->n && (rc = p->hipc(hipMemcpy(rec.data(), p->d_rec.p, p->n * sizeof(mcom_mm128), hipMemcpyDeviceToHost), "copy records"))) { fclose(f); return rc; }
		dump_buckets(f, "B0", rec);
	}
	if ((rc = mcomh_kt_for_bucket(p)) || (rc = ensure_host_contigs(p))) { fclose(f); return rc; }
	fprintf(f, "STAGE bucket\n");
	dump_contigs(f, "bucket", p->C);
	p->join_sg();
	dump_list(f, "sg", p->sg);
	// the first-m minimizers the reference pushed into mi[0] while building the contigs (:458-474)
	{
		std::vector<mcom_mm128> rec;
		if (p->C.n()) {
			DevBuf<uint32_t> mo, mo2; DevBuf<mcom_mm128> mr, mr2; uint64_t ta = 0, tm = 0;
			if ((rc = upload_contigs(p, p->C)) || (rc = sketch_contigs(p, p->C.n(), p->C.ref.size(), mo, mr, ta))) { fclose(f); return rc; }
			if (!mo2.reserve(p->C.n() + 2) || !mr2.reserve(p->C.n() * (size_t)p->m + 16)) { fclose(f); return p->fail(MCOM_E_NOMEM, "dump buffers"); }
			if ((rc = p->gpu(mcom_minimizer_prefix(p->ctx, mo.p, mr.p, p->C.n(), (uint32_t)p->m, mo2.p, mr2.p, &tm)))) { fclose(f); return rc; }
			rec.resize(tm);
			if (tm) (void)hipMemcpy(rec.data(), mr2.p, tm * sizeof(mcom_mm128), hipMemcpyDeviceToHost);
			// the reference's id is (index << 8) + tid (kthread_bucket.c:458); ours is the index
			for (mcom_mm128 &r : rec) r.y = ((r.y >> 32) << (32 + MCOM_REF_CONTIG_ID_SHIFT)) | (r.y & 0xFFFFFFFFull);
		}
		dump_buckets(f, "MI0", rec);
	}
	if ((rc = mcomh_combine_cluster(p)) || (rc = ensure_host_contigs(p))) { fclose(f); return rc; }
	fprintf(f, "STAGE combine\n");
	dump_contigs(f, "combine", p->C);
	if ((rc = run_stage2(p, f))) { fclose(f); return rc; }
	fprintf(f, "STAGE final\n");
	dump_list(f, "sg", p->sg);
	fprintf(f, "END\n");
	fclose(f);
	return MCOM_OK;
}

// ---- results -----------------------------------------------------------------------------------------------
extern "C" size_t mcomh_n_contigs(const mcomh_pipeline *p) { if (!p) return 0; (void)ensure_host_contigs(const_cast<mcomh_pipeline*>(p)); return p->C.n(); }
extern "C" const char *mcomh_contig_ref(const mcomh_pipeline *p, size_t i, size_t *len)
{
	(void)ensure_host_contigs(const_cast<mcomh_pipeline*>(p));
	if (len) *len = p->C.rsize(i);
	return p->C.ref.data() + p->C.roff[i];                 // NOT NUL-terminated: use *len
}
static int host_members(P *p) { const int rc = materialize(p); return rc ? rc : ensure_host_contigs(p); }
extern "C" size_t mcomh_contig_n(const mcomh_pipeline *p, size_t i) { if (host_members(const_cast<mcomh_pipeline*>(p))) return 0; return p->C.msize(i); }
extern "C" const uint64_t *mcomh_contig_members(const mcomh_pipeline *p, size_t i) { if (host_members(const_cast<mcomh_pipeline*>(p))) return nullptr; return p->C.mem.data() + p->C.moff[i]; }
extern "C" const uint32_t *mcomh_list(const mcomh_pipeline *p, const char *name, size_t *n)
{
	const_cast<mcomh_pipeline*>(p)->join_sg();
	if (p->cls_failed) { const_cast<mcomh_pipeline*>(p)->fail(MCOM_E_HIP, "the read classes did not arrive from the device"); if (n) *n = 0; return nullptr; }
	const std::vector<uint32_t> *v = nullptr;
	if (!strcmp(name, "allA")) v = &p->allA; else if (!strcmp(name, "allT")) v = &p->allT; else if (!strcmp(name, "allN")) v = &p->allN;
	else if (!strcmp(name, "fpA")) v = &p->fpA; else if (!strcmp(name, "fpT")) v = &p->fpT; else if (!strcmp(name, "fpN")) v = &p->fpN;
	else if (!strcmp(name, "Nfile")) v = &p->Nfile;
	else if (!strcmp(name, "sg")) { if (n) *n = p->sg.size(); return p->sg.data(); }
	if (!v) { if (n) *n = 0; return nullptr; }
	if (n) *n = v->size();
	return v->data();
}
extern "C" int mcomh_contig_set(const mcomh_pipeline *cp, size_t *n_contigs, const char **ref, const uint64_t **ref_off, const uint64_t **mem, const uint64_t **mem_off)
{
	mcomh_pipeline *p = const_cast<mcomh_pipeline*>(cp);
	if (!p) return MCOM_E_ARG;
	const int rc = host_members(p);
	if (rc) return rc;
	if (n_contigs) *n_contigs = p->C.n();
	if (ref) *ref = p->C.ref.data();
	if (ref_off) *ref_off = p->C.roff.data();
	if (mem) *mem = p->C.mem.data();
	if (mem_off) *mem_off = p->C.moff.data();
	return MCOM_OK;
}

extern "C" int mcomh_result_digest(mcomh_pipeline *p, uint64_t out[8])
{
	if (!p || !out) return MCOM_E_ARG;
	for (int i = 0; i < 8; ++i) out[i] = 0;
	p->join_sg();
	int rc = materialize(p);
	if (rc) return rc;
	if (!p->dC_valid) return p->fail(MCOM_E_ARG, "no contig set on the device");
	const DevSet &D = p->dC;
	auto mix = [](const uint64_t sx[2]) { return sx[0] ^ ((sx[1] << 23) | (sx[1] >> 41)); };
	uint64_t sx[2], so[2];
	out[0] = D.n; out[1] = D.chars; out[2] = D.members;
	if ((rc = p->gpu(mcom_digest(p->ctx, D.seq.p, D.chars, sx)))) return rc;
	out[4] = mix(sx);
	if ((rc = p->gpu(mcom_digest(p->ctx, D.mem.p, D.members * 8, sx)))) return rc;
	out[5] = mix(sx);
	if ((rc = p->gpu(mcom_digest(p->ctx, D.soff.p, D.n ? (D.n + 1) * 8 : 0, sx))) || (rc = p->gpu(mcom_digest(p->ctx, D.moff.p, D.n ? (D.n + 1) * 8 : 0, so)))) return rc;
	out[6] = mix(sx) + 3 * mix(so);
	// the lists live on the host
	uint64_t h = 0, nsg = 0;
	ensure_sg_flag(p);
	for (size_t i = 0; i < p->sg.size(); ++i) if (p->sg_flag.size() != p->sg.size() || !p->sg_flag[i]) { h = h * 0x9E3779B97F4A7C15ull + p->sg[i] + 1; ++nsg; }
	out[3] = nsg;
	for (const std::vector<uint32_t> *v : {&p->allA, &p->allT, &p->allN, &p->fpA, &p->fpT, &p->fpN, &p->Nfile}) { h = h * 0xD6E8FEB86659FD93ull + v->size(); for (uint32_t x : *v) h = h * 0x9E3779B97F4A7C15ull + x + 1; }
	out[7] = h;
	return MCOM_OK;
}

extern "C" int mcomh_prof_enable(mcomh_pipeline *p, int on)
{
	if (!p) return MCOM_E_ARG;
	p->prof_on = on != 0;
	if (p->ctx2) (void)mcom_prof_enable(p->ctx2, on);
	return mcom_prof_enable(p->ctx, on);
}
extern "C" int mcomh_prof_read(mcomh_pipeline *p, const char *name, double *total_ms, uint64_t *launches)
{
	if (!p) return MCOM_E_ARG;
	int rc = p->gpu(mcom_prof_read(p->ctx, name, total_ms, launches));
	if (!rc && p->ctx2) {                                                    // what ran on the copy stream's context counts too
		double ms2 = 0; uint64_t l2 = 0;
		if (mcom_prof_read(p->ctx2, name, &ms2, &l2) == MCOM_OK) { if (total_ms) *total_ms += ms2; if (launches) *launches += l2; }
	}
	return rc;
}
extern "C" int mcomh_prof_kernels(mcomh_pipeline *p, const char *name, char *buf, size_t cap, size_t *need)
{
	if (!p) return MCOM_E_ARG;
	// both contexts' tallies, merged line by line (the copy stream's context launches a few of the same kernels)
	std::map<std::string, uint64_t> all;
	mcom_ctx *cs[2] = { p->ctx, p->ctx2 };
	for (mcom_ctx *c : cs) {
		if (!c) continue;
		size_t nb = 0;
		int rc = p->gpu(mcom_prof_kernels(c, name, nullptr, 0, &nb));
		if (rc) return rc;
		std::string t(nb, '\0');
		(void)mcom_prof_kernels(c, name, &t[0], nb, nullptr);
		size_t a = 0;
		while (a < t.size()) {
			size_t e = t.find('\n', a); if (e == std::string::npos) break;
			size_t tab = t.find('\t', a);
			if (tab != std::string::npos && tab < e) all[t.substr(a, tab - a)] += strtoull(t.c_str() + tab + 1, nullptr, 10);
			a = e + 1;
		}
	}
	std::string txt;
	for (const auto &kv : all) { txt += kv.first; txt += '\t'; txt += std::to_string(kv.second); txt += '\n'; }
	if (need) *need = txt.size() + 1;
	if (buf && cap) { size_t m = txt.size() < cap - 1 ? txt.size() : cap - 1; memcpy(buf, txt.data(), m); buf[m] = 0; }
	return MCOM_OK;
}
extern "C" double mcomh_stat(const mcomh_pipeline *p, const char *name)
{
	if (!p) return 0;
	if (!strcmp(name, "k")) return p->k;
	if (!strcmp(name, "n")) return (double)p->n;
	if (!strcmp(name, "L")) return p->L;
	if (!strcmp(name, "e")) return p->e;
	if (!strcmp(name, "step")) return p->step;
	if (!strcmp(name, "maxthr")) return p->maxthr;
	if (!strcmp(name, "rw")) return p->rw;
	if (!strcmp(name, "maxsearch")) return p->maxsearch;
	if (!strcmp(name, "sketch_strings")) return (double)mcom_counter(p->ctx, "sketch_strings");
	if (!strcmp(name, "sort_overflow_segments")) return (double)mcom_counter(p->ctx, "sort_overflow_segments");
	auto it = p->stat.find(name);
	return it == p->stat.end() ? 0.0 : it->second;
}

// ----------------------------------------------------------------------------------------------------
// cluster_dump at one thread: the pre-bsc stream files      kthread_dump.c:364-678
//   ref.bin.0 beg_pos.bin.0 dir.bin.0 dif_char.txt.0 (print_encode, :142-236) info.txt single.seq single_N.seq
//   AA.txt TT.txt NN.txt.  Single-end, not order-preserving.  Host serialisation, outside the timed region.
// ----------------------------------------------------------------------------------------------------
namespace {
struct BitWriter {                          // DNA_push / bit_push (breads.h:232-248)
	FILE *f; unsigned acc = 0; int n = 0; int per;
	BitWriter(FILE *f_, int bits) : f(f_), per(bits) {}
	void push(unsigned x) { acc += x << (per * n); if (++n == 8 / per) { fputc((int)acc, f); acc = 0; n = 0; } }
	void flush() { if (n > 0) fputc((int)acc, f); acc = 0; n = 0; }
};
// run-length text against a constant base (:579-596)
std::string const_base_text(const char *s, int L, char base)
{
	std::string out; int eq = 0;
	for (int t = 0; t < L; ++t) {
		if (s[t] != base) { if (eq > 0) { out += std::to_string(eq); eq = 0; } out.push_back(s[t]); }
		else ++eq;
	}
	if (out.empty()) out = "0";
	return out;
}
}

// cmpcluster3 (kthread_cb.c:72-84): offset, then read id -- the member order of the order-preserving mode
static bool less_cluster3(uint64_t a, uint64_t b)
{
	const int pa = (int)((uint32_t)a >> 1), pb = (int)((uint32_t)b >> 1);
	if (pa != pb) return pa < pb;
	return (int)(a >> 32) < (int)(b >> 32);
}
// one sorted id list as the reference writes it: the first id, then differences (kthread_dump.c:438-520)
static bool write_ids(const std::string &path, const std::vector<uint32_t> &ids)
{
	FILE *f = fopen(path.c_str(), "wb");
	if (!f) return false;
	for (size_t i = 0; i < ids.size(); ++i) { const uint32_t v = i ? ids[i] - ids[i - 1] : ids[i]; fwrite(&v, 4, 1, f); }
	fclose(f);
	return true;
}

// The default mode with the streams made on the device (csrc/streams.hip): the member lists are put into dump order there
// (cmpcluster2, kthread_dump.c:143), one thread per member writes its mismatch text, position delta and direction bit, the
// bit-packed files are one thread per output byte; what crosses PCIe are the finished file images (about 10 bytes per read
// instead of the 64 of packed rows + N masks), and the host writes them with one fwrite each.
static bool write_file(const std::string &path, const void *data, size_t bytes)
{
	FILE *f = fopen(path.c_str(), "wb");
	if (!f) return false;
	const bool ok = bytes == 0 || fwrite(data, 1, bytes, f) == bytes;
	return fclose(f) == 0 && ok;
}
// the reads of a (short) list as strings with their N put back (kthread_dump.c:178-186): rows and masks gathered on the device
static int fetch_read_strings(P *p, const std::vector<uint32_t> &rids, std::vector<char> &out)
{
	const int L = p->L, W = p->W, NW = p->NW;
	const size_t n = rids.size();
	out.assign(n * ((size_t)L + 1), 0);
	if (!n) return MCOM_OK;
	DevBuf<uint32_t> d_r; DevBuf<uint64_t> d_rows, d_nm;
	if (!d_r.reserve(n) || !d_rows.reserve(n * W) || !d_nm.reserve(n * NW)) return p->fail(MCOM_E_NOMEM, "list rows");
	std::vector<uint64_t> rows(n * (size_t)W), nm(n * (size_t)NW);
	int rc;
	if ((rc = p->h2d(d_r.p, rids.data(), n, "upload list")) || (rc = p->gpu(mcom_gather_rows(p->ctx, p->d_packed.p, d_r.p, n, L, d_rows.p))) ||
	    (rc = p->gpu(mcom_gather_rows(p->ctx, p->d_nmask.p, d_r.p, n, 32 * NW, d_nm.p))) ||               // (a mask row = NW words = the packed row of 32 NW bases)
	    (rc = p->d2h(rows.data(), d_rows.p, n * W, "copy rows")) || (rc = p->d2h(nm.data(), d_nm.p, n * NW, "copy masks")) || (rc = p->sync("list rows"))) return rc;
	for (size_t r = 0; r < n; ++r) {
		char *o = out.data() + r * ((size_t)L + 1);
		for (int i = 0; i < L; ++i) o[i] = ((nm[r * NW + (i >> 6)] >> (i & 63)) & 1) ? 'N' : ACGT[(rows[r * W + (i >> 5)] >> (2 * (i & 31))) & 3];
	}
	return MCOM_OK;
}
// mode 0: default; 1: order-preserving (-p, ORDER); 2: paired end (_PE: reads [0, n/2) are the first file, the rest their mates).
// Round 4: all three modes are made here; the host loop below (cluster_dump_impl) is what mcomh_params.host_dump asks for and what
// the tests compare this with.
static int cluster_dump_device(mcomh_pipeline *p, const char *folder, int mode)
{
	const bool order = mode == 1, pe = mode == 2, sorted = mode != 0;
	const uint32_t half = (uint32_t)(p->n / 2);
	if (pe && (p->n & 1)) return p->fail(MCOM_E_ARG, "paired-end mode needs as many reads in the second file as in the first");   // preprocess.c:70
	p->join_sg();
	ensure_sg_flag(p);
	if (p->cls_failed) return p->fail(MCOM_E_HIP, "the read classes did not arrive from the device: the class lists are incomplete");
	int rc = materialize(p);
	if (rc) return rc;
	if (!p->dC_valid) return p->fail(MCOM_E_ARG, "no contig set on the device");
	const double t0 = now_ms();
	DevSet &D = p->dC;
	const int L = p->L;
	const std::string dir(folder);
	PinVec<uint8_t> h_pos, h_dir, h_text, h_ref, h_single, h_ids;
	uint64_t text_bytes = 0, ids_bytes = 0;
	const size_t pos_bytes = 4 * D.n + 2 * (size_t)D.members;
	DevBuf<uint64_t> mem2, moff2;                                                 // the members in dump order (kept for the pairing streams)
	const uint64_t *d_mem_dump = nullptr;
	// Stream sets (the reference: one per thread, kthread_dump.c:370-379): set t holds contigs [sc[t], sc[t + 1]) -- runs of about equal member
	// counts.  The member streams are encoded ONCE for all contigs; a set's beg_pos / dif_char / ids file is a slice of the image (cut at a contig
	// boundary), its bit-packed files (ref.bin, dir.bin, file.bin) are packed per set, from bit 0, as the reference's per-thread writers do.
	const int T = D.n ? (int)std::min<size_t>((size_t)p->stream_sets, D.n) : 1;
	std::vector<uint64_t> sc(T + 1, 0), sm(T + 1, 0), schars(T + 1, 0), stext(T + 1, 0), sids(T + 1, 0), ssecond(T + 1, 0), sdir(T + 1, 0), sref(T + 1, 0);
	if (D.n) {
		if ((rc = ensure_packed_contigs(p))) return rc;
		if (!mem2.reserve(D.members + 1) || !moff2.reserve(D.n + 2)) return p->fail(MCOM_E_NOMEM, "member lists");
		int kb = 2; while ((1ull << kb) < 4 * std::max<uint64_t>(p->maxlen, 2 * (uint64_t)L) + 4) ++kb;
		const uint64_t *d_moff_dump = D.moff.p;
		if (!sorted) {
			// members in dump order: stable by (offset, direction) inside every contig -- the fold of mcom_members_finalize with one empty pass
			const uint32_t *ac[1] = {nullptr}; const uint64_t *am[1] = {nullptr}; const uint64_t an[1] = {0};
			if ((rc = p->gpu(mcom_members_finalize(p->ctx, D.mem.p, D.moff.p, D.n, D.members, ac, am, an, 1, kb, mem2.p, moff2.p)))) return rc;
			d_moff_dump = moff2.p;
		} else if ((rc = p->gpu(mcom_members_order3(p->ctx, D.mem.p, D.moff.p, D.n, D.members, kb, mem2.p)))) return rc;   // cmpcluster3 (kthread_dump.c:34): the pipeline keeps its own order
		d_mem_dump = mem2.p;
		// where the sets are cut
		sc[T] = D.n; sm[T] = D.members; schars[T] = D.chars;
		if (T > 1) {
			std::vector<uint64_t> hm(D.n + 1), hs(D.n + 1);
			if ((rc = p->d2h(hm.data(), d_moff_dump, D.n + 1, "copy member offsets")) || (rc = p->d2h(hs.data(), D.soff.p, D.n + 1, "copy string offsets")) || (rc = p->sync("copy offsets"))) return rc;
			for (int t = 1; t < T; ++t) {
				size_t c = (size_t)(std::lower_bound(hm.begin(), hm.begin() + D.n, (uint64_t)D.members * (uint64_t)t / (uint64_t)T) - hm.begin());
				c = std::max<size_t>(c, (size_t)sc[t - 1] + 1); c = std::min<size_t>(c, D.n - (size_t)(T - t));   // every set holds a contig
				sc[t] = c; sm[t] = hm[c]; schars[t] = hs[c];
			}
		}
		for (int t = 0; t <= T; ++t) { sdir[t] = t ? sdir[t - 1] + (sm[t] - sm[t - 1] + 7) / 8 : 0; sref[t] = t ? sref[t - 1] + (schars[t] - schars[t - 1] + 3) / 4 : 0; }
		const size_t dir_bytes = (size_t)sdir[T], ref_bytes = (size_t)sref[T];
		DevBuf<uint8_t> d_pos, d_dir, d_text, d_ref, d_ids;
		if (!d_pos.reserve(pos_bytes + 16) || !d_dir.reserve(std::max(dir_bytes, ((size_t)D.members + 7) / 8) + 16) || !d_ref.reserve(ref_bytes + 16)) return p->fail(MCOM_E_NOMEM, "stream buffers");
		if ((rc = p->gpu(mcom_dump_members(p->ctx, p->d_packed.p, p->d_nmask.p, L, p->d_cbits.p, p->d_coff_words.p, mem2.p, d_moff_dump, D.n, D.members, d_pos.p, d_dir.p, nullptr, 0, &text_bytes)))) return rc;
		if (!d_text.reserve(text_bytes + 16)) return p->fail(MCOM_E_NOMEM, "stream buffers");
		if ((rc = p->gpu(mcom_dump_members_at(p->ctx, p->d_packed.p, p->d_nmask.p, L, p->d_cbits.p, p->d_coff_words.p, mem2.p, d_moff_dump, D.n, D.members, d_pos.p, d_dir.p, d_text.p,
		                                      text_bytes + 16, &text_bytes, sm.data(), T + 1, stext.data())))) return rc;
		// the bit-packed files, set by set (one set: dir.bin is what mcom_dump_members made already)
		for (int t = 0; t < T; ++t) {
			if (T > 1 && (rc = p->gpu(mcom_dump_member_bits(p->ctx, mem2.p + sm[t], sm[t + 1] - sm[t], 0, 0, d_dir.p + sdir[t])))) return rc;
			if ((rc = p->gpu(mcom_dump_refbin(p->ctx, D.seq.p + schars[t], schars[t + 1] - schars[t], d_ref.p + sref[t])))) return rc;
		}
		if (order) {                                                               // ids.bin.T (kthread_dump.c:116-127)
			ids_bytes = 4 * (uint64_t)D.members;
			if (!d_ids.reserve(ids_bytes + 16)) return p->fail(MCOM_E_NOMEM, "stream buffers");
			if ((rc = p->gpu(mcom_dump_ids_order(p->ctx, mem2.p, D.moff.p, D.n, D.members, (uint32_t*)d_ids.p)))) return rc;
			for (int t = 0; t <= T; ++t) sids[t] = 4 * sm[t];
		} else if (pe) {                                                           // ids.txt.T (kthread_dump_pe.c:70-74)
			if ((rc = p->gpu(mcom_dump_ids_text(p->ctx, mem2.p, D.members, half, nullptr, 0, &ids_bytes)))) return rc;
			if (!d_ids.reserve(ids_bytes + 16)) return p->fail(MCOM_E_NOMEM, "stream buffers");
			if ((rc = p->gpu(mcom_dump_ids_text_at(p->ctx, mem2.p, D.members, half, d_ids.p, ids_bytes + 16, &ids_bytes, sm.data(), T + 1, sids.data())))) return rc;
		}
		if (!h_pos.resize(pos_bytes) || !h_dir.resize(dir_bytes) || !h_text.resize(text_bytes) || !h_ref.resize(ref_bytes) || !h_ids.resize(ids_bytes)) return p->fail(MCOM_E_NOMEM, "stream images");
		if ((rc = p->d2h(h_pos.data(), d_pos.p, pos_bytes, "copy streams")) || (rc = p->d2h(h_dir.data(), d_dir.p, dir_bytes, "copy streams")) ||
		    (rc = p->d2h(h_text.data(), d_text.p, (size_t)text_bytes, "copy streams")) || (rc = p->d2h(h_ref.data(), d_ref.p, ref_bytes, "copy streams")) ||
		    (ids_bytes && (rc = p->d2h(h_ids.data(), d_ids.p, (size_t)ids_bytes, "copy streams"))) || (rc = p->sync("copy streams"))) return rc;
	}
	p->stat["t_dump_members"] += now_ms() - t0;
	// unclustered reads: those with an N join the N file as text, the others are packed four per byte in singleton order (:390-417);
	// the two other modes write every list in read-id order (:420-427) and the singles after their sorted ids
	std::vector<uint32_t> live, single_ids, nfile = p->Nfile;
	live.reserve(p->sg.size());
	for (size_t i = 0; i < p->sg.size(); ++i) if (!p->sg_flag[i]) live.push_back(p->sg[i]);
	if (!live.empty()) {
		DevBuf<uint32_t> d_ids; DevBuf<uint8_t> d_f;
		std::vector<uint8_t> hasn(live.size());
		if (!d_ids.reserve(live.size()) || !d_f.reserve(live.size())) return p->fail(MCOM_E_NOMEM, "singleton buffers");
		if ((rc = p->h2d(d_ids.p, live.data(), live.size(), "upload singletons")) || (rc = p->gpu(mcom_rows_have_n(p->ctx, p->d_nmask.p, d_ids.p, live.size(), L, d_f.p))) ||
		    (rc = p->d2h(hasn.data(), d_f.p, live.size(), "copy flags")) || (rc = p->sync("singleton flags"))) return rc;
		single_ids.reserve(live.size());
		for (size_t i = 0; i < live.size(); ++i) (hasn[i] ? nfile : single_ids).push_back(live[i]);
	}
	std::vector<uint32_t> fpA = p->fpA, fpT = p->fpT, fpN = p->fpN, allA = p->allA, allT = p->allT, allN = p->allN;
	if (sorted) for (std::vector<uint32_t> *v : {&fpA, &fpT, &fpN, &nfile, &single_ids, &allA, &allT, &allN}) std::sort(v->begin(), v->end());
	if (!single_ids.empty()) {
		DevBuf<uint32_t> d_ids; DevBuf<uint8_t> d_single;
		const size_t sbytes = (single_ids.size() * (size_t)L + 3) / 4;
		if (!d_ids.reserve(single_ids.size()) || !d_single.reserve(sbytes + 16) || !h_single.resize(sbytes)) return p->fail(MCOM_E_NOMEM, "singleton buffers");
		if ((rc = p->h2d(d_ids.p, single_ids.data(), single_ids.size(), "upload singletons")) || (rc = p->gpu(mcom_dump_singles(p->ctx, p->d_packed.p, d_ids.p, single_ids.size(), L, d_single.p))) ||
		    (rc = p->d2h(h_single.data(), d_single.p, sbytes, "copy singletons")) || (rc = p->sync("singleton stream"))) return rc;
	}
	// paired end: the pairing streams (kthread_dump_pe.c:270-470, :583-612) over the eight lists and over the members
	PinVec<uint8_t> h_pe_sp, h_pe_0, h_fb_sp, h_fb_0;
	if (pe) {
		std::vector<uint32_t> lists;
		for (const std::vector<uint32_t> *v : {&allA, &allT, &allN, &fpA, &fpT, &fpN, &nfile, &single_ids}) lists.insert(lists.end(), v->begin(), v->end());
		const size_t nl = lists.size(), nm = D.n ? (size_t)D.members : 0;
		DevBuf<uint32_t> d_lists, d_isp, d_i0; DevBuf<uint8_t> d_fsp, d_f0;
		if (!d_lists.reserve(nl + 1) || !d_isp.reserve(nl + 1) || !d_i0.reserve(nm + 1) || !d_fsp.reserve((nl + 7) / 8 + 16) || !d_f0.reserve((nm + 7) / 8 + 16)) return p->fail(MCOM_E_NOMEM, "pairing buffers");
		uint64_t cnt[2] = {0, 0};
		if ((nl && (rc = p->h2d(d_lists.p, lists.data(), nl, "upload lists"))) ||
		    (rc = p->gpu(mcom_dump_pairing_at(p->ctx, d_lists.p, nl, d_mem_dump, nm, half, d_isp.p, d_fsp.p, d_i0.p, d_f0.p, cnt, sm.data(), T + 1, ssecond.data())))) return rc;
		// file.bin.T: a set's file bits packed from bit 0 (the byte ranges of dir.bin.T: one bit per member either way)
		if (T > 1) { if (!d_f0.reserve((size_t)sdir[T] + 16)) return p->fail(MCOM_E_NOMEM, "pairing buffers");
			for (int t = 0; t < T; ++t) if ((rc = p->gpu(mcom_dump_member_bits(p->ctx, d_mem_dump + sm[t], sm[t + 1] - sm[t], 1, half, d_f0.p + sdir[t])))) return rc; }
		const size_t f0_bytes = T > 1 ? (size_t)sdir[T] : (nm + 7) / 8;
		if (!h_pe_sp.resize(4 * cnt[0]) || !h_pe_0.resize(4 * cnt[1]) || !h_fb_sp.resize((nl + 7) / 8) || !h_fb_0.resize(f0_bytes)) return p->fail(MCOM_E_NOMEM, "pairing images");
		if ((cnt[0] && (rc = p->d2h(h_pe_sp.data(), (const uint8_t*)d_isp.p, 4 * cnt[0], "copy pairing"))) || (cnt[1] && (rc = p->d2h(h_pe_0.data(), (const uint8_t*)d_i0.p, 4 * cnt[1], "copy pairing"))) ||
		    (nl && (rc = p->d2h(h_fb_sp.data(), d_fsp.p, (nl + 7) / 8, "copy pairing"))) || (f0_bytes && (rc = p->d2h(h_fb_0.data(), d_f0.p, f0_bytes, "copy pairing"))) || (rc = p->sync("pairing streams"))) return rc;
	}
	p->stat["t_dump_gpu"] += now_ms() - t0;
	const double tw = now_ms();
	// the file images are written side by side (round 5: one fwrite after the other was 0.046 s of the paired-end set's 0.08 at 20 M reads -- the page
	// cache takes what one core copies into it)
	struct Job { std::string path; const void *data; size_t bytes; };
	std::vector<Job> jobs;
	for (int t = 0; t < T; ++t) {
		const std::string sfx = "." + std::to_string(t);
		const size_t pos_a = D.n ? 4 * (size_t)sc[t] + 2 * (size_t)sm[t] : 0, pos_b = D.n ? 4 * (size_t)sc[t + 1] + 2 * (size_t)sm[t + 1] : 0;
		jobs.push_back(Job{dir + "/ref.bin" + sfx, h_ref.data() + sref[t], D.n ? (size_t)(sref[t + 1] - sref[t]) : 0});
		jobs.push_back(Job{dir + "/beg_pos.bin" + sfx, h_pos.data() + pos_a, pos_b - pos_a});
		jobs.push_back(Job{dir + "/dir.bin" + sfx, h_dir.data() + sdir[t], D.n ? (size_t)(sdir[t + 1] - sdir[t]) : 0});
		jobs.push_back(Job{dir + "/dif_char.txt" + sfx, h_text.data() + stext[t], D.n ? (size_t)(stext[t + 1] - stext[t]) : 0});
		if (order) jobs.push_back(Job{dir + "/ids.bin" + sfx, h_ids.data() + sids[t], D.n ? (size_t)(sids[t + 1] - sids[t]) : 0});   // kthread_dump.c:266-269
		if (pe) {
			jobs.push_back(Job{dir + "/ids.txt" + sfx, h_ids.data() + sids[t], D.n ? (size_t)(sids[t + 1] - sids[t]) : 0});
			jobs.push_back(Job{dir + "/peids.bin" + sfx, h_pe_0.data() + 4 * ssecond[t], D.n ? (size_t)(4 * (ssecond[t + 1] - ssecond[t])) : 0});
			jobs.push_back(Job{dir + "/file.bin" + sfx, h_fb_0.data() + sdir[t], D.n ? (size_t)(sdir[t + 1] - sdir[t]) : h_fb_0.size()});
		}
	}
	jobs.push_back(Job{dir + "/single.seq", h_single.data(), h_single.size()});
	if (pe) { jobs.push_back(Job{dir + "/peids.bin.sp", h_pe_sp.data(), h_pe_sp.size()}); jobs.push_back(Job{dir + "/file.bin.sp", h_fb_sp.data(), h_fb_sp.size()}); }
	{
		std::sort(jobs.begin(), jobs.end(), [](const Job &x, const Job &y) { return x.bytes > y.bytes; });   // the large ones first
		std::atomic<size_t> next{0}; std::atomic<bool> failed{false};
		auto writer = [&]() { for (;;) { const size_t i = next.fetch_add(1); if (i >= jobs.size()) return; if (!write_file(jobs[i].path, jobs[i].data, jobs[i].bytes)) failed = true; } };
		const size_t nthreads = std::min<size_t>(std::min<size_t>(8, (size_t)std::max(1, p->host_threads)), jobs.size());
		std::vector<std::thread> wt;
		for (size_t q = 1; q < nthreads; ++q) wt.emplace_back(writer);
		writer();
		for (auto &x : wt) x.join();
		if (failed) return p->fail(MCOM_E_ARG, "cannot write into %s", folder);
	}
	if (order && (!write_ids(dir + "/allA.ids.bin", allA) || !write_ids(dir + "/allT.ids.bin", allT) || !write_ids(dir + "/allN.ids.bin", allN) ||
	              !write_ids(dir + "/AA.ids.bin", fpA) || !write_ids(dir + "/TT.ids.bin", fpT) || !write_ids(dir + "/NN.ids.bin", fpN) ||
	              !write_ids(dir + "/Nfile.ids.bin", nfile) || !write_ids(dir + "/singleFile.ids.bin", single_ids))) return p->fail(MCOM_E_ARG, "cannot write id streams");
	FILE *finfo = fopen((dir + "/info.txt").c_str(), "w");
	if (!finfo) return p->fail(MCOM_E_ARG, "cannot write info.txt");
	if (pe) fprintf(finfo, "%d %d\n%u\n%zu %zu %zu\n", L, T, half, p->allA.size(), p->allT.size(), p->allN.size());   // kthread_dump_pe.c:222-234
	else fprintf(finfo, "%d %d\n%zu %zu %zu\n", L, T, p->allA.size(), p->allT.size(), p->allN.size());   // :375-376
	if (order) fprintf(finfo, "%u\n", (unsigned)p->n);                                               // :377-379
	fclose(finfo);
	// the short lists as text (:566-671)
	struct TextList { const char *name; const std::vector<uint32_t> *ids; char base; };
	const TextList lists[4] = {{"AA.txt", &fpA, 'A'}, {"TT.txt", &fpT, 'T'}, {"NN.txt", &fpN, 'N'}, {"single_N.seq", &nfile, 0}};
	for (const TextList &tl : lists) {
		std::vector<char> strs;
		if ((rc = fetch_read_strings(p, *tl.ids, strs))) return rc;
		FILE *f = fopen((dir + "/" + tl.name).c_str(), "w");
		if (!f) return p->fail(MCOM_E_ARG, "cannot write text streams");
		for (size_t r = 0; r < tl.ids->size(); ++r) {
			const char *t = strs.data() + r * ((size_t)L + 1);
			if (tl.base) fprintf(f, "%s\n", const_base_text(t, L, tl.base).c_str()); else fprintf(f, "%s\n", t);
		}
		fclose(f);
	}
	p->stat["t_dump_write"] += now_ms() - tw;
	p->stat["dump_bytes"] += (double)(pos_bytes + h_dir.size() + h_ref.size() + text_bytes + ids_bytes + h_single.size());
	return MCOM_OK;
}

// mode 0: default; 1: order-preserving (-p, ORDER); 2: paired end (_PE: reads [0, n/2) are the first file, the rest their mates)
static int cluster_dump_impl(mcomh_pipeline *p, const char *folder, int mode)
{
	if (!p || !folder) return MCOM_E_ARG;
	if (!p->host_dump) return cluster_dump_device(p, folder, mode);
	const bool order = mode == 1, pe = mode == 2, sorted = mode != 0;
	const uint32_t half = (uint32_t)(p->n / 2);
	if (pe && (p->n & 1)) return p->fail(MCOM_E_ARG, "paired-end mode needs as many reads in the second file as in the first");   // preprocess.c:70
	p->join_sg();
	ensure_sg_flag(p);
	if (p->cls_failed) return p->fail(MCOM_E_HIP, "the read classes did not arrive from the device: the class lists are incomplete");
	{ int rcm = materialize(p); if (!rcm) rcm = ensure_host_contigs(p); if (rcm) return rcm; }
	const int L = p->L, W = p->W, NW = p->NW;
	const size_t n = p->n;
	int rc;
	std::vector<uint64_t> packed(n * (size_t)W), nmask(n * (size_t)NW);
	if (n && ((rc = p->hipc(hipMemcpy(packed.data(), p->d_packed.p, n * (size_t)W * 8, hipMemcpyDeviceToHost), "copy packed reads")) ||
	          (rc = p->hipc(hipMemcpy(nmask.data(), p->d_nmask.p, n * (size_t)NW * 8, hipMemcpyDeviceToHost), "copy N masks")))) return rc;
	auto has_n = [&](uint32_t rid) { for (int q = 0; q < NW; ++q) if (nmask[(size_t)rid * NW + q]) return true; return false; };
	// the read as stored by the reference with its N put back (:178-186)
	auto read_str = [&](uint32_t rid, char *out) {
		for (int i = 0; i < L; ++i) {
			const bool isn = (nmask[(size_t)rid * NW + (i >> 6)] >> (i & 63)) & 1;
			out[i] = isn ? 'N' : ACGT[(packed[(size_t)rid * W + (i >> 5)] >> (2 * (i & 31))) & 3];
		}
		out[L] = 0;
	};
	auto open = [&](const char *name, const char *mode) { std::string path = std::string(folder) + "/" + name; return fopen(path.c_str(), mode); };
	FILE *fref = open("ref.bin.0", "wb"), *fpos = open("beg_pos.bin.0", "wb"), *fdir = open("dir.bin.0", "wb"), *fdif = open("dif_char.txt.0", "w");
	FILE *fids = order ? open("ids.bin.0", "wb") : pe ? open("ids.txt.0", "w") : nullptr;   // kthread_dump.c:266-269, kthread_dump_pe.c:150
	if (!fref || !fpos || !fdir || !fdif || (sorted && !fids)) return p->fail(MCOM_E_ARG, "cannot write into %s", folder);
	std::vector<uint32_t> pe_members;                                             // paired end: the members in stream order
	BitWriter refbin(fref, 2), dirbin(fdir, 1);
	ContigSet &C = p->C;
	std::vector<char> t((size_t)L + 1), en;
	std::vector<uint64_t> omem;
	static const char RCT[256] = {0};
	for (size_t c = 0; c < C.n(); ++c) {
		const uint64_t *mm = C.mem.data() + C.moff[c];
		if (sorted) {                                                          // a copy: the pipeline keeps the order of the default mode
			omem.assign(mm, mm + C.msize(c));
			std::stable_sort(omem.begin(), omem.end(), less_cluster3);           // :34
			mm = omem.data();
		} else std::stable_sort(C.mem.data() + C.moff[c], C.mem.data() + C.moff[c + 1], less_cluster2);         // :143
		uint32_t pre_rid = 0;
		const char *ref = C.ref.data() + C.roff[c];
		for (size_t i = 0; i < C.rsize(c); ++i) refbin.push(ref[i] == 'A' ? 0u : ref[i] == 'C' ? 1u : ref[i] == 'G' ? 2u : 3u);   // :158-160
		const uint32_t num = (uint32_t)C.msize(c);
		fwrite(&num, 4, 1, fpos);
		int pre_pos = 0;
		for (uint64_t q = 0; q < C.msize(c); ++q) {
			const uint64_t y = mm[q];
			const uint32_t rid = (uint32_t)(y >> 32); const int pos = (int)((uint32_t)y >> 1), dir = (int)(y & 1);
			if (pe) { fprintf(fids, "%d %u\n", rid < half ? 0 : 1, rid); pe_members.push_back(rid); }   // kthread_dump_pe.c:70-74
			read_str(rid, t.data());
			if (dir) {                                                         // reverse_complement, N stays N (preprocess.c:22-37)
				for (int i = 0, j = L - 1; i < j; ++i, --j) std::swap(t[i], t[j]);
				for (int i = 0; i < L; ++i) t[i] = t[i] == 'A' ? 'T' : t[i] == 'T' ? 'A' : t[i] == 'C' ? 'G' : t[i] == 'G' ? 'C' : 'N';
			}
			en.clear();
			int eq = 0;
			for (int tj = 0; tj < L; ++tj) {                                   // :200-216
				if (ref[pos + tj] != t[tj]) {
					if (eq > 1) { const std::string d = std::to_string(eq); en.insert(en.end(), d.begin(), d.end()); }
					else for (int i = tj - eq; i < tj; ++i) en.push_back(t[i]);
					eq = 0;
					en.push_back(t[tj]);
				} else ++eq;
			}
			if (en.empty()) en.push_back('0');
			fwrite(en.data(), 1, en.size(), fdif); fputc('\n', fdif);
			const uint16_t posbin = (uint16_t)(pos - pre_pos);
			fwrite(&posbin, 2, 1, fpos);
			if (order) {                                                       // :116-127: the id, or its difference at an equal position
				const uint32_t v = (q == 0 || posbin > 0) ? rid : rid - pre_rid;
				fwrite(&v, 4, 1, fids);
				pre_rid = rid;
			}
			dirbin.push((unsigned)dir);
			pre_pos = pos;
		}
	}
	(void)RCT;
	dirbin.flush(); refbin.flush();                                               // :303-309
	fclose(fref); fclose(fpos); fclose(fdir); fclose(fdif);
	if (fids) fclose(fids);
	FILE *finfo = open("info.txt", "w");
	if (!finfo) return p->fail(MCOM_E_ARG, "cannot write info.txt");
	if (pe) fprintf(finfo, "%d %d\n%u\n%zu %zu %zu\n", L, 1, half, p->allA.size(), p->allT.size(), p->allN.size());   // kthread_dump_pe.c:222-234
	else fprintf(finfo, "%d %d\n%zu %zu %zu\n", L, 1, p->allA.size(), p->allT.size(), p->allN.size());   // :375-376
	if (order) fprintf(finfo, "%u\n", (unsigned)p->n);                                               // :377-379
	fclose(finfo);
	// singletons: those with N join the N file, the others are packed 4 per byte (:390-417, :545-548)
	FILE *fsingle = open("single.seq", "wb");
	if (!fsingle) return p->fail(MCOM_E_ARG, "cannot write single.seq");
	BitWriter sb(fsingle, 2);
	std::vector<uint32_t> nfile = p->Nfile, single_ids;
	auto push_single = [&](uint32_t rid) { for (int j = 0; j < L; ++j) sb.push((unsigned)((packed[(size_t)rid * W + (j >> 5)] >> (2 * (j & 31))) & 3)); };
	for (size_t i = 0; i < p->sg.size(); ++i) {
		if (p->sg_flag[i]) continue;
		const uint32_t rid = p->sg[i];
		if (has_n(rid)) nfile.push_back(rid);
		else if (sorted) single_ids.push_back(rid);                               // :409-411
		else push_single(rid);
	}
	std::vector<uint32_t> fpA = p->fpA, fpT = p->fpT, fpN = p->fpN;
	if (sorted) {
		// every list sorted by read id (:420-427); the singles follow their sorted ids
		std::vector<uint32_t> allA = p->allA, allT = p->allT, allN = p->allN;
		for (std::vector<uint32_t> *v : {&fpA, &fpT, &fpN, &nfile, &single_ids, &allA, &allT, &allN}) std::sort(v->begin(), v->end());
		const std::string d(folder);
		if (pe) {
			// Pairing (kthread_dump_pe.c:270-470, :583-612): reads of the first file are numbered in the order the decoder
			// will write them (the eight lists, then the contig members); a read of the second file carries the number of
			// its mate, a file bit per read says which kind it is.
			std::vector<uint32_t> mpv(half, 0); uint32_t mpvid = 0;
			const std::vector<uint32_t> *lists[8] = {&allA, &allT, &allN, &fpA, &fpT, &fpN, &nfile, &single_ids};
			for (const std::vector<uint32_t> *v : lists) for (uint32_t rid : *v) if (rid < half) mpv[rid] = mpvid++;
			for (uint32_t rid : pe_members) if (rid < half) mpv[rid] = mpvid++;
			auto write_pairing = [&](const char *ids_name, const char *file_name, auto &&each) -> bool {
				FILE *fi = open(ids_name, "wb"), *ff = open(file_name, "wb");
				if (!fi || !ff) return false;
				BitWriter fb(ff, 1);
				each([&](uint32_t rid) { if (rid < half) fb.push(0); else { fb.push(1); const uint32_t v = mpv[rid - half]; fwrite(&v, 4, 1, fi); } });
				fb.flush(); fclose(fi); fclose(ff);
				return true;
			};
			if (!write_pairing("peids.bin.sp", "file.bin.sp", [&](auto &&f) { for (const std::vector<uint32_t> *v : lists) for (uint32_t rid : *v) f(rid); }) ||
			    !write_pairing("peids.bin.0", "file.bin.0", [&](auto &&f) { for (uint32_t rid : pe_members) f(rid); })) return p->fail(MCOM_E_ARG, "cannot write pairing streams");
		} else if (!write_ids(d + "/allA.ids.bin", allA) || !write_ids(d + "/allT.ids.bin", allT) || !write_ids(d + "/allN.ids.bin", allN) ||
		    !write_ids(d + "/AA.ids.bin", fpA) || !write_ids(d + "/TT.ids.bin", fpT) || !write_ids(d + "/NN.ids.bin", fpN) ||
		    !write_ids(d + "/Nfile.ids.bin", nfile) || !write_ids(d + "/singleFile.ids.bin", single_ids)) return p->fail(MCOM_E_ARG, "cannot write id streams");
		for (uint32_t rid : single_ids) push_single(rid);
	}
	sb.flush(); fclose(fsingle);
	FILE *fa = open("AA.txt", "w"), *ft = open("TT.txt", "w"), *fn = open("NN.txt", "w"), *fnf = open("single_N.seq", "w");
	if (!fa || !ft || !fn || !fnf) return p->fail(MCOM_E_ARG, "cannot write text streams");
	for (uint32_t rid : fpA) { read_str(rid, t.data()); fprintf(fa, "%s\n", const_base_text(t.data(), L, 'A').c_str()); }   // :566-597
	for (uint32_t rid : fpT) { read_str(rid, t.data()); fprintf(ft, "%s\n", const_base_text(t.data(), L, 'T').c_str()); }   // :599-627
	for (uint32_t rid : fpN) { read_str(rid, t.data()); fprintf(fn, "%s\n", const_base_text(t.data(), L, 'N').c_str()); }   // :629-657
	for (uint32_t rid : nfile) { read_str(rid, t.data()); fprintf(fnf, "%s\n", t.data()); }                                     // :659-671
	fclose(fa); fclose(ft); fclose(fn); fclose(fnf);
	return MCOM_OK;
}

extern "C" int mcomh_cluster_dump(mcomh_pipeline *p, const char *folder) { return cluster_dump_impl(p, folder, 0); }
// the order-preserving mode (minicom -p = the reference compiled with ORDER): id streams beside every stream
extern "C" int mcomh_cluster_dump_order(mcomh_pipeline *p, const char *folder) { return cluster_dump_impl(p, folder, 1); }
// paired end (minicompe = the reference compiled with _PE): reads [0, n/2) come from the first file, read n/2 + i is the mate of read i
extern "C" int mcomh_cluster_dump_pe(mcomh_pipeline *p, const char *folder) { return cluster_dump_impl(p, folder, 2); }
