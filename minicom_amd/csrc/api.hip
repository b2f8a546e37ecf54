// minicom_amd/csrc/api.hip -- context, error text and workspace of libmcom_hip.so
#include "mcom_dev.hpp"
#include "../../include/mcom_test.h"
#include <stdarg.h>
#include <stdlib.h>
#include <cxxabi.h>
#include <stdio.h>
#include <string.h>

int mcom_fail(mcom_ctx *ctx, int code, const char *fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	if (ctx) { ctx->err = buf; ctx->pin_wait.clear(); ctx->pin_off = 0; }
	return code;
}

#define MCOM_PIN_BYTES 4096
hipError_t mcom_d2h_async(mcom_ctx *ctx, void *dst, const void *src, size_t bytes)
{
	for (uint32_t q = 0; q < 8; ++q) {                                             // the total of a scan that is on its way: already in pinned memory (newest entry first)
		const mcom_ctx::ScanTotal &t = ctx->scan_last[(ctx->scan_last_at + 7u - q) % 8u];
		if (t.last && t.last == src && t.bytes == bytes && t.gen == ctx->launch_gen) { mcom_ctx::PinWait w{dst, 0, bytes}; w.from = ctx->scan_tot + t.slot; ctx->pin_wait.push_back(w); return hipSuccess; }
	}
	if (bytes <= 256) {
		if (!ctx->pin && hipHostMalloc((void**)&ctx->pin, MCOM_PIN_BYTES, hipHostMallocDefault) != hipSuccess) { ctx->pin = nullptr; (void)hipGetLastError(); }
		const size_t need = (bytes + 7) & ~(size_t)7;
		if (ctx->pin && ctx->pin_off + need <= MCOM_PIN_BYTES) {
			const hipError_t e = hipMemcpyAsync(ctx->pin + ctx->pin_off, src, bytes, hipMemcpyDeviceToHost, ctx->stream);
			if (e == hipSuccess) { ctx->pin_wait.push_back(mcom_ctx::PinWait{dst, ctx->pin_off, bytes}); ctx->pin_off += need; }
			return e;
		}
	}
	return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream);
}
hipError_t mcom_stream_sync(mcom_ctx *ctx) { return mcom_stream_sync_poison(ctx, nullptr); }
// poisoned != nullptr: a tripped poison flag is REPORTED there instead of failing the call -- for a caller that launched the kernel
// whose wait may run out and has another way to the same result (mcom_claim_pairs: the launch-per-round loop)
hipError_t mcom_stream_sync_poison(mcom_ctx *ctx, bool *poisoned)
{
	hipError_t e = hipStreamSynchronize(ctx->stream);
	++ctx->launch_gen;                                                             // (what a scan left in pinned memory is consumed below)
	if (poisoned) *poisoned = false;
	if (e == hipSuccess && ctx->poison && *ctx->poison) {                        // a kernel's bounded wait ran out: its results are wrong
		*ctx->poison = 0;
		if (poisoned) *poisoned = true;
		else {
			ctx->err = "a device-side wait ran out: results of this stream are invalid";
			e = hipErrorLaunchFailure;
		}
	}
	if (e == hipSuccess) for (const mcom_ctx::PinWait &w : ctx->pin_wait) memcpy(w.dst, w.from ? w.from : (const void*)(ctx->pin + w.off), w.bytes);
	ctx->pin_wait.clear(); ctx->pin_off = 0;
	if (e == hipSuccess && !ctx->free_later.empty()) { for (void *q : ctx->free_later) mcom_dfree(q); ctx->free_later.clear(); }
	return e;
}

// One spare workspace per device outlives its context: a job creates a context per run and the workspace reaches
// gigabytes, whose hipMalloc / hipFree cost tens of milliseconds each time.
#include <mutex>
static std::mutex g_ws_mu;
static struct { void *p; size_t bytes; } g_ws_spare[16] = {};

// Device blocks of the library's own objects (index, dictionaries, tables, scratch) are recycled: a job builds and
// drops dozens of multi-gigabyte objects per run, each a different size, and the driver's page-table work for a fresh
// allocation is tens of milliseconds per gigabyte.  Blocks are rounded up so that the next round's slightly different
// request still fits, and are only given back to the driver when an allocation fails.
#include <map>
static std::mutex g_blk_mu;
static std::multimap<std::pair<int, size_t>, void*> g_blk_free;       // (device, bytes) -> block
static std::map<void*, std::pair<int, size_t>> g_blk_live;

static void (*g_oom_hook)(void) = nullptr;
// the free blocks of the pool go back to the runtime (a caller that keeps a pool of its own asks for this when IT runs out of memory)
extern "C" void mcom_pool_trim(void)
{
	std::vector<void*> drop;
	{ std::lock_guard<std::mutex> g(g_blk_mu); for (auto &kv : g_blk_free) drop.push_back(kv.second); g_blk_free.clear(); }
	for (void *q : drop) (void)hipFree(q);
	(void)hipGetLastError();
}
// ... and the other way round: called when a block cannot be had even after the pool's own free blocks are gone
extern "C" void mcom_set_oom_hook(void (*hook)(void)) { g_oom_hook = hook; }

hipError_t mcom_dmalloc(void **out, size_t bytes)
{
	int dev = 0; (void)hipGetDevice(&dev);
	const size_t unit = (size_t)8 << 20;
	size_t cap = ((bytes + bytes / 4 + unit - 1) / unit) * unit;
	{
		std::lock_guard<std::mutex> g(g_blk_mu);
		auto it = g_blk_free.lower_bound(std::make_pair(dev, bytes));
		if (it != g_blk_free.end() && it->first.first == dev && it->first.second <= 2 * cap) {
			*out = it->second; g_blk_live[*out] = it->first; g_blk_free.erase(it); return hipSuccess;
		}
	}
	void *p = nullptr;
	hipError_t e = hipMalloc(&p, cap);
	if (e != hipSuccess) {
		std::vector<void*> drop;
		{ std::lock_guard<std::mutex> g(g_blk_mu); for (auto it = g_blk_free.begin(); it != g_blk_free.end();) { if (it->first.first == dev) { drop.push_back(it->second); it = g_blk_free.erase(it); } else ++it; } }
		for (void *q : drop) (void)hipFree(q);
		(void)hipGetLastError();
		e = hipMalloc(&p, cap);
		if (e != hipSuccess && g_oom_hook) { (void)hipGetLastError(); g_oom_hook(); e = hipMalloc(&p, cap); }
		if (e != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return e; }
	}
	std::lock_guard<std::mutex> g(g_blk_mu);
	g_blk_live[p] = std::make_pair(dev, cap);
	*out = p;
	return hipSuccess;
}
void mcom_dfree(void *p)
{
	if (!p) return;
	std::lock_guard<std::mutex> g(g_blk_mu);
	auto it = g_blk_live.find(p);
	if (it == g_blk_live.end()) { (void)hipFree(p); return; }
	g_blk_free.emplace(it->second, p);
	g_blk_live.erase(it);
}

void mcom_dfree_later(mcom_ctx *ctx, void *p) { if (p) ctx->free_later.push_back(p); }

void *mcom_zeroed(mcom_ctx *ctx, void *fallback, size_t bytes)
{
	const size_t need = (bytes + 7) & ~(size_t)7;
	if (need && need <= 4096) {
		const size_t half = mcom_ctx::ZPOOL_BYTES / 2;
		if (!ctx->zpool) {
			if (hipMalloc((void**)&ctx->zpool, mcom_ctx::ZPOOL_BYTES) != hipSuccess) { ctx->zpool = nullptr; (void)hipGetLastError(); }
			ctx->zpool_half = 1; ctx->zpool_used = half;                               // (the first request clears and takes half 0)
		}
		if (ctx->zpool) {
			// two halves: when one is used up the OTHER one is cleared and taken, so what was handed out stays as its kernel left it for
			// at least half a pool of further requests (a caller may read its counter back a few launches later, not thousands)
			if (ctx->zpool_used + need > half) {
				const int nh = ctx->zpool_half ^ 1;
				if (hipMemsetAsync(ctx->zpool + (size_t)nh * half, 0, half, ctx->stream) != hipSuccess) { (void)hipGetLastError(); goto plain; }
				ctx->zpool_half = nh; ctx->zpool_used = 0;
			}
			void *p = ctx->zpool + (size_t)ctx->zpool_half * half + ctx->zpool_used;
			ctx->zpool_used += need;
			return p;
		}
	}
plain:
	if (!fallback || hipMemsetAsync(fallback, 0, bytes, ctx->stream) != hipSuccess) { ctx->err = "clearing a counter failed"; return nullptr; }
	return fallback;
}

int mcom_ws_reserve(mcom_ctx *ctx, size_t bytes)
{
	if (bytes <= ctx->ws_bytes) return MCOM_OK;
	if (ctx->ws) { MCOM_HIP(ctx, hipStreamSynchronize(ctx->stream)); MCOM_HIP(ctx, hipFree(ctx->ws)); ctx->ws = nullptr; ctx->ws_bytes = 0; }
	if (ctx->device >= 0 && ctx->device < 16) {
		std::lock_guard<std::mutex> g(g_ws_mu);
		if (g_ws_spare[ctx->device].p) {
			void *sp = g_ws_spare[ctx->device].p; const size_t sb = g_ws_spare[ctx->device].bytes;
			g_ws_spare[ctx->device].p = nullptr; g_ws_spare[ctx->device].bytes = 0;
			if (sb >= bytes) { ctx->ws = sp; ctx->ws_bytes = sb; return MCOM_OK; }
			(void)hipFree(sp);
		}
	}
	size_t want = bytes + bytes / 8 + (1 << 20);
	hipError_t e = hipMalloc(&ctx->ws, want);
	if (e != hipSuccess) { (void)hipGetLastError(); mcom_pool_trim(); if (g_oom_hook) g_oom_hook(); e = hipMalloc(&ctx->ws, want); }
	if (e != hipSuccess) { (void)hipGetLastError(); ctx->ws = nullptr; return mcom_fail(ctx, MCOM_E_NOMEM, "workspace of %zu bytes: %s", want, hipGetErrorString(e)); }
	ctx->ws_bytes = want;
	return MCOM_OK;
}

extern "C" const char *mcom_version(void) { return "mcom-hip 0.1 (gfx950)"; }

extern "C" int mcom_create(mcom_ctx **out, int device, void *hip_stream)
{
	if (!out) return MCOM_E_ARG;
	*out = nullptr;
	mcom_ctx *ctx = new mcom_ctx();
	ctx->device = device; ctx->stream = (hipStream_t)hip_stream; ctx->ws = nullptr; ctx->ws_bytes = 0; ctx->n_cu = 256;
	hipError_t e = hipSetDevice(device);
	if (e != hipSuccess) {
		// no usable GPU: the product path must fail loudly, there is no CPU fallback
		fprintf(stderr, "mcom_create: hipSetDevice(%d) failed: %s\n", device, hipGetErrorString(e));
		delete ctx;
		return MCOM_E_HIP;
	}
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ctx->n_cu = prop.multiProcessorCount;
	*out = ctx;
	return MCOM_OK;
}

extern "C" void mcom_destroy(mcom_ctx *ctx)
{
	if (!ctx) return;
	if (!ctx->free_later.empty()) { (void)hipStreamSynchronize(ctx->stream); for (void *q : ctx->free_later) mcom_dfree(q); ctx->free_later.clear(); }
	if (ctx->ws) {
		(void)hipStreamSynchronize(ctx->stream);
		void *drop = ctx->ws;
		if (ctx->device >= 0 && ctx->device < 16) {
			std::lock_guard<std::mutex> g(g_ws_mu);
			if (ctx->ws_bytes > g_ws_spare[ctx->device].bytes) { drop = g_ws_spare[ctx->device].p; g_ws_spare[ctx->device].p = ctx->ws; g_ws_spare[ctx->device].bytes = ctx->ws_bytes; }
		}
		if (drop) (void)hipFree(drop);
	}
	if (ctx->pin) (void)hipHostFree(ctx->pin);
	if (ctx->scan_parts) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(ctx->scan_parts); }
	if (ctx->zpool) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(ctx->zpool); }
	if (ctx->poison) (void)hipHostFree((void*)ctx->poison);
	delete ctx;
}

extern "C" int mcom_set_stream(mcom_ctx *ctx, void *hip_stream)
{
	if (!ctx) return MCOM_E_ARG;
	if (!ctx->free_later.empty() && ctx->stream != (hipStream_t)hip_stream) { (void)hipStreamSynchronize(ctx->stream); for (void *q : ctx->free_later) mcom_dfree(q); ctx->free_later.clear(); }
	if (ctx->zpool && ctx->stream != (hipStream_t)hip_stream) { (void)hipStreamSynchronize(ctx->stream); ctx->zpool_used = mcom_ctx::ZPOOL_BYTES; }   // (the pool's order is the stream's: the next request clears a half on the new one)
	ctx->stream = (hipStream_t)hip_stream;
	return MCOM_OK;
}

extern "C" int mcom_sync(mcom_ctx *ctx)
{
	if (!ctx) return MCOM_E_ARG;
	MCOM_HIP(ctx, mcom_stream_sync(ctx));
	return MCOM_OK;
}

extern "C" const char *mcom_last_error(const mcom_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

// ---- optional kernel timing with HIP events on the context's stream ------------------------------------
static const char *PROF_NAMES[PROF_COUNT] = { "classify_pack", "sketch_reads", "radix_pass", "sketch_contigs", "find_next",
                                              "dict_build", "realign_windows", "consensus", "cindex_build", "realign_reads" };

static std::mutex g_ev_mu;
static std::vector<hipEvent_t> g_ev_free;
hipEvent_t mcom_prof_event_get()
{
	{ std::lock_guard<std::mutex> g(g_ev_mu); if (!g_ev_free.empty()) { hipEvent_t e = g_ev_free.back(); g_ev_free.pop_back(); return e; } }
	hipEvent_t e = nullptr;
	if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
	return e;
}
void mcom_prof_event_put(hipEvent_t e) { if (e) { std::lock_guard<std::mutex> g(g_ev_mu); g_ev_free.push_back(e); } }

static void prof_collect(mcom_ctx *ctx)
{
	if (ctx->prof_open.empty()) return;
	(void)hipStreamSynchronize(ctx->stream);
	for (McomProfSpan &s : ctx->prof_open) {
		float ms = 0;
		if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) { ctx->prof_ms[s.id] += ms; ctx->prof_calls[s.id] += 1; }
		mcom_prof_event_put(s.a); mcom_prof_event_put(s.b);
	}
	ctx->prof_open.clear();
}

extern "C" int mcom_prof_enable(mcom_ctx *ctx, int on)
{
	if (!ctx) return MCOM_E_ARG;
	prof_collect(ctx);
	ctx->prof_on = on != 0;
	return MCOM_OK;
}

extern "C" int mcom_prof_reset(mcom_ctx *ctx)
{
	if (!ctx) return MCOM_E_ARG;
	prof_collect(ctx);
	for (int i = 0; i < PROF_COUNT; ++i) { ctx->prof_ms[i] = 0; ctx->prof_calls[i] = 0; }
	ctx->prof_kernels.clear();
	return MCOM_OK;
}

// host address of a kernel -> "k_x<5, true>": the runtime's (mangled) name, demangled, without return type and argument list
static std::string prof_kernel_text(mcom_ctx *ctx, const void *fn)
{
	const char *m = hipKernelNameRefByPtr(fn, ctx->stream);
	if (!m) { (void)hipGetLastError(); char b[32]; snprintf(b, sizeof b, "kernel@%p", fn); return b; }
	int st = 0;
	char *d = abi::__cxa_demangle(m, nullptr, nullptr, &st);
	std::string t = (st == 0 && d) ? d : m;
	free(d);
	if (t.compare(0, 5, "void ") == 0) t.erase(0, 5);
	const char *anon = "(anonymous namespace)::";
	for (size_t q; (q = t.find(anon)) != std::string::npos; ) t.erase(q, strlen(anon));
	int depth = 0;
	for (size_t i = 0; i < t.size(); ++i) {                                     // cut at the argument list: the first '(' outside template brackets
		if (t[i] == '<') ++depth; else if (t[i] == '>') --depth;
		else if (t[i] == '(' && depth == 0) { t.erase(i); break; }
	}
	return t;
}

extern "C" int mcom_prof_kernels(mcom_ctx *ctx, const char *name, char *buf, size_t cap, size_t *need)
{
	if (!ctx || !name) return MCOM_E_ARG;
	int cls = -1;
	if (!strcmp(name, "*")) cls = -2;
	else if (!strcmp(name, "-")) cls = PROF_COUNT;
	else for (int i = 0; i < PROF_COUNT; ++i) if (!strcmp(name, PROF_NAMES[i])) cls = i;
	if (cls == -1) return mcom_fail(ctx, MCOM_E_ARG, "unknown profiler name %s", name);
	std::map<std::string, uint64_t> out;
	for (const auto &kv : ctx->prof_kernels) if (cls == -2 || kv.first.first == cls) out[prof_kernel_text(ctx, kv.first.second)] += kv.second;
	std::string txt;
	for (const auto &kv : out) { txt += kv.first; txt += '\t'; txt += std::to_string(kv.second); txt += '\n'; }
	if (need) *need = txt.size() + 1;
	if (buf && cap) { size_t m = txt.size() < cap - 1 ? txt.size() : cap - 1; memcpy(buf, txt.data(), m); buf[m] = 0; }
	return MCOM_OK;
}

extern "C" int mcom_prof_read(mcom_ctx *ctx, const char *name, double *total_ms, uint64_t *launches)
{
	if (!ctx || !name) return MCOM_E_ARG;
	prof_collect(ctx);
	for (int i = 0; i < PROF_COUNT; ++i) if (!strcmp(name, PROF_NAMES[i])) {
		if (total_ms) *total_ms = ctx->prof_ms[i];
		if (launches) *launches = ctx->prof_calls[i];
		return MCOM_OK;
	}
	return mcom_fail(ctx, MCOM_E_ARG, "unknown profiler name %s", name);
}

// ---- diagnostics -----------------------------------------------------------------------------------------------
extern "C" uint64_t mcom_counter(const mcom_ctx *ctx, const char *name)
{
	if (!ctx || !name) return 0;
	if (!strcmp(name, "sort_overflow_segments")) return ctx->sort_overflow_segments;
	if (!strcmp(name, "sketch_strings")) return ctx->sketch_strings;
	return 0;
}
extern "C" int mcom_set_segment_capacity(mcom_ctx *ctx, uint32_t records)
{
	if (!ctx || records > 4096) return MCOM_E_ARG;
	ctx->seg_cap = records;
	return MCOM_OK;
}
extern "C" int mcom_set_consensus_capacity(mcom_ctx *ctx, uint32_t members)
{
	if (!ctx) return MCOM_E_ARG;
	ctx->bs_cap = members;
	return MCOM_OK;
}
extern "C" int mcom_set_sketch_kernel(mcom_ctx *ctx, int wave_per_string)
{
	if (!ctx) return MCOM_E_ARG;
	ctx->sketch_wave_only = wave_per_string == 1;
	ctx->sketch_ring64 = wave_per_string == 2;
	ctx->sketch_ring32_only = wave_per_string == 3;
	ctx->sketch_lane_always = wave_per_string >= 2;
	return MCOM_OK;
}
extern "C" int mcom_set_sketch_prefix_bits(mcom_ctx *ctx, int bits)
{
	if (!ctx || bits < 1 || bits > 30) return MCOM_E_ARG;
	ctx->sketch_prefix_bits = bits;
	return MCOM_OK;
}
extern "C" int mcom_set_claim_route(mcom_ctx *ctx, int route)
{
	if (!ctx || route < 0 || route > 3) return MCOM_E_ARG;
	ctx->claim_route = route;
	return MCOM_OK;
}
extern "C" int mcom_claim_fallbacks(const mcom_ctx *ctx) { return ctx ? (int)ctx->claim_fallbacks : 0; }
extern "C" int mcom_set_screen_route(mcom_ctx *ctx, int route)
{
	if (!ctx || route < 0 || route > 2) return MCOM_E_ARG;
	ctx->screen_route = route;
	return MCOM_OK;
}
extern "C" int mcom_screen_fallbacks(const mcom_ctx *ctx) { return ctx ? (int)ctx->screen_fallbacks : 0; }
extern "C" int mcom_set_lookup_route(mcom_ctx *ctx, int route)
{
	if (!ctx || route < 0 || route > 1) return MCOM_E_ARG;
	ctx->lookup_route = route;
	return MCOM_OK;
}
extern "C" int mcom_set_index_capacity(mcom_ctx *ctx, int entries)
{
	if (!ctx) return MCOM_E_ARG;
	ctx->cix_cap_set = entries >= 0; ctx->cix_cap = entries >= 0 ? (uint32_t)entries : 0u;
	return MCOM_OK;
}
