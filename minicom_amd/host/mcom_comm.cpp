// minicom_amd/host/mcom_comm.cpp -- the communicator of the multi-GPU path (include/mcom_host.h, "multi-GPU").
//
// The reference is a shared-memory program (pthreads, kthread_*.c): nothing to mirror here.  One primitive, a byte-wise
// all-to-all with per-peer offsets and sizes, over RCCL (ncclSend / ncclRecv in a group: on MI355X every peer pair has
// its own xGMI link, so a grouped point-to-point exchange drives all seven links of a GPU at once; no ring, no
// all-reduce of payload) or over a caller-supplied host transport (MPI, torch.distributed / gloo: the tests).  The
// all-gather and the small reductions of the pipeline are built on it.
#include "../../include/mcom.h"
#include "../../include/mcom_host.h"
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstdarg>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace {

// librccl.so.1 is loaded on first use: a single-GPU job never needs it, and in a process that has PyTorch loaded the
// soname resolves to the copy PyTorch brought, so that one RCCL serves the process.
struct Rccl {
	void *dl = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
	std::string err;
	bool load() {
		if (dl) return true;
		dl = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
		if (!dl) dl = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
		if (!dl) { err = std::string("cannot load librccl.so.1: ") + dlerror(); return false; }
		auto sym = [&](const char *n) { void *p = dlsym(dl, n); if (!p) err = std::string("librccl lacks ") + n; return p; };
		GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
		CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
		CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
		GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
		GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
		Send = (decltype(Send))sym("ncclSend");
		Recv = (decltype(Recv))sym("ncclRecv");
		GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
		if (!GetUniqueId || !CommInitRank || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !GetErrorString) { dlclose(dl); dl = nullptr; return false; }
		return true;
	}
};
Rccl &rccl() { static Rccl *r = new Rccl(); return *r; }
std::mutex g_rccl_mu;

constexpr uint64_t MAX_MESSAGE = 256ull << 20;    // one ncclSend / ncclRecv never carries more

}  // namespace

struct mcomh_comm {
	int rank = 0, world = 1;
	bool is_rccl = false;
	int device = 0;
	ncclComm_t nc = nullptr;
	mcomh_comm_ops ops{}; void *user = nullptr;
	std::string err;
	uint64_t bytes_sent = 0, calls = 0;
	double seconds = 0;                        // wall time inside all-to-all calls (staging copies and waiting for the peers included)
	// staging: pinned host blocks (callbacks transport with device data), device blocks (RCCL with host data)
	void *h_send = nullptr, *h_recv = nullptr; size_t h_send_cap = 0, h_recv_cap = 0;
	void *d_send = nullptr, *d_recv = nullptr; size_t d_send_cap = 0, d_recv_cap = 0;
	int fail(int code, const char *fmt, ...) {
		char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
		err = buf; return code;
	}
	bool host_stage(void *&p, size_t &cap, size_t need) {
		if (need <= cap) return true;
		if (p) (void)hipHostFree(p);
		p = nullptr; cap = 0;
		const size_t want = need + need / 4 + 4096;
		if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; return false; }
		cap = want; return true;
	}
	bool dev_stage(void *&p, size_t &cap, size_t need) {
		if (need <= cap) return true;
		if (p) (void)hipFree(p);
		p = nullptr; cap = 0;
		const size_t want = need + need / 4 + 4096;
		if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return false; }
		cap = want; return true;
	}
};

extern "C" int mcomh_comm_unique_id(void *id128)
{
	if (!id128) return MCOM_E_ARG;
	std::lock_guard<std::mutex> g(g_rccl_mu);
	if (!rccl().load()) { fprintf(stderr, "mcomh_comm_unique_id: %s\n", rccl().err.c_str()); return MCOM_E_HIP; }
	ncclUniqueId id;
	if (rccl().GetUniqueId(&id) != ncclSuccess) return MCOM_E_HIP;
	static_assert(sizeof id == MCOMH_UNIQUE_ID_BYTES, "unique id size");
	memcpy(id128, &id, sizeof id);
	return MCOM_OK;
}

extern "C" int mcomh_comm_create_rccl(mcomh_comm **out, int rank, int world, const void *id128, int device)
{
	if (!out) return MCOM_E_ARG;
	*out = nullptr;
	if (!id128 || world < 1 || rank < 0 || rank >= world) return MCOM_E_ARG;
	{
		std::lock_guard<std::mutex> g(g_rccl_mu);
		if (!rccl().load()) { fprintf(stderr, "mcomh_comm_create_rccl: %s\n", rccl().err.c_str()); return MCOM_E_HIP; }
	}
	if (hipSetDevice(device) != hipSuccess) { fprintf(stderr, "mcomh_comm_create_rccl: no usable GPU %d\n", device); return MCOM_E_HIP; }
	mcomh_comm *c = new mcomh_comm();
	c->rank = rank; c->world = world; c->is_rccl = true; c->device = device;
	ncclUniqueId id; memcpy(&id, id128, sizeof id);
	const ncclResult_t r = rccl().CommInitRank(&c->nc, world, id, rank);
	if (r != ncclSuccess) { fprintf(stderr, "mcomh_comm_create_rccl: ncclCommInitRank: %s\n", rccl().GetErrorString(r)); delete c; return MCOM_E_HIP; }
	*out = c;
	return MCOM_OK;
}

extern "C" int mcomh_comm_create_ops(mcomh_comm **out, int rank, int world, const mcomh_comm_ops *ops, void *user)
{
	if (!out) return MCOM_E_ARG;
	*out = nullptr;
	if (!ops || !ops->alltoallv || world < 1 || rank < 0 || rank >= world) return MCOM_E_ARG;
	mcomh_comm *c = new mcomh_comm();
	c->rank = rank; c->world = world; c->ops = *ops; c->user = user;
	*out = c;
	return MCOM_OK;
}

extern "C" void mcomh_comm_destroy(mcomh_comm *c)
{
	if (!c) return;
	if (c->nc) (void)rccl().CommDestroy(c->nc);
	if (c->h_send) (void)hipHostFree(c->h_send);
	if (c->h_recv) (void)hipHostFree(c->h_recv);
	if (c->d_send) (void)hipFree(c->d_send);
	if (c->d_recv) (void)hipFree(c->d_recv);
	delete c;
}

extern "C" int mcomh_comm_rank(const mcomh_comm *c) { return c ? c->rank : -1; }
extern "C" int mcomh_comm_world(const mcomh_comm *c) { return c ? c->world : 0; }
extern "C" const char *mcomh_comm_last_error(const mcomh_comm *c) { return c ? c->err.c_str() : "null communicator"; }
extern "C" void mcomh_comm_stats(const mcomh_comm *c, uint64_t *bytes_sent, uint64_t *calls)
{
	if (bytes_sent) *bytes_sent = c ? c->bytes_sent : 0;
	if (calls) *calls = c ? c->calls : 0;
}

// the grouped point-to-point exchange on device buffers; messages above MAX_MESSAGE travel in pieces (both sides of a
// pair know the size, so they cut it the same way)
static int rccl_alltoallv(mcomh_comm *c, const char *send, const uint64_t *so, const uint64_t *sb, char *recv, const uint64_t *ro, const uint64_t *rb, hipStream_t st)
{
	const int R = c->world;
	uint64_t biggest = 0;
	for (int q = 0; q < R; ++q) { if (sb[q] > biggest) biggest = sb[q]; if (rb[q] > biggest) biggest = rb[q]; }
	const uint64_t pieces = biggest ? (biggest + MAX_MESSAGE - 1) / MAX_MESSAGE : 0;
	Rccl &N = rccl();
	for (uint64_t t = 0; t < pieces; ++t) {
		ncclResult_t r = N.GroupStart();
		if (r != ncclSuccess) return c->fail(MCOM_E_HIP, "ncclGroupStart: %s", N.GetErrorString(r));
		for (int d = 0; d < R && r == ncclSuccess; ++d) {
			// peers in a rotated order: rank r talks to r+1, r+2, ... so that no GPU is everybody's first target
			const int q = (c->rank + d) % R, qr = (c->rank - d + R) % R;
			const uint64_t lo = t * MAX_MESSAGE;
			if (sb[q] > lo) { const uint64_t len = sb[q] - lo < MAX_MESSAGE ? sb[q] - lo : MAX_MESSAGE; r = N.Send(send + so[q] + lo, (size_t)len, ncclUint8, q, c->nc, st); }
			if (r == ncclSuccess && rb[qr] > lo) { const uint64_t len = rb[qr] - lo < MAX_MESSAGE ? rb[qr] - lo : MAX_MESSAGE; r = N.Recv(recv + ro[qr] + lo, (size_t)len, ncclUint8, qr, c->nc, st); }
		}
		const ncclResult_t r2 = N.GroupEnd();
		if (r != ncclSuccess) return c->fail(MCOM_E_HIP, "ncclSend/ncclRecv: %s", N.GetErrorString(r));
		if (r2 != ncclSuccess) return c->fail(MCOM_E_HIP, "ncclGroupEnd: %s", N.GetErrorString(r2));
	}
	if (hipStreamSynchronize(st) != hipSuccess) return c->fail(MCOM_E_HIP, "all-to-all: the stream failed");
	return MCOM_OK;
}

static int alltoallv_impl(mcomh_comm *c, const void *send, const uint64_t *so, const uint64_t *sb, void *recv, const uint64_t *ro, const uint64_t *rb,
                          int on_device, void *hip_stream);
extern "C" int mcomh_comm_alltoallv(mcomh_comm *c, const void *send, const uint64_t *so, const uint64_t *sb, void *recv, const uint64_t *ro, const uint64_t *rb,
                                    int on_device, void *hip_stream)
{
	if (!c || !so || !sb || !ro || !rb) return MCOM_E_ARG;
	// the kernels that made the data are waited for BEFORE the clock starts: seconds = the exchange, not the work in front of it
	if (on_device && hipStreamSynchronize((hipStream_t)hip_stream) != hipSuccess) return c->fail(MCOM_E_HIP, "all-to-all: the stream failed");
	const auto t0 = std::chrono::steady_clock::now();
	const int rc = alltoallv_impl(c, send, so, sb, recv, ro, rb, on_device, hip_stream);
	c->seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	return rc;
}
extern "C" double mcomh_comm_seconds(const mcomh_comm *c) { return c ? c->seconds : 0.0; }

static int alltoallv_impl(mcomh_comm *c, const void *send, const uint64_t *so, const uint64_t *sb, void *recv, const uint64_t *ro, const uint64_t *rb,
                          int on_device, void *hip_stream)
{
	const int R = c->world;
	hipStream_t st = (hipStream_t)hip_stream;
	uint64_t ts = 0, tr = 0;
	for (int q = 0; q < R; ++q) { ts += sb[q]; tr += rb[q]; if (q != c->rank) c->bytes_sent += sb[q]; }
	++c->calls;
	if ((ts && !send) || (tr && !recv)) return c->fail(MCOM_E_ARG, "all-to-all: null buffer");
	if (c->is_rccl) {
		if (on_device) return rccl_alltoallv(c, (const char*)send, so, sb, (char*)recv, ro, rb, st);
		// host data (a few counters): packed into device staging blocks, exchanged, unpacked
		if (!c->dev_stage(c->d_send, c->d_send_cap, (size_t)ts + 16) || !c->dev_stage(c->d_recv, c->d_recv_cap, (size_t)tr + 16)) return c->fail(MCOM_E_NOMEM, "all-to-all: staging");
		std::vector<uint64_t> pso(R), pro(R);
		uint64_t a = 0, b = 0;
		for (int q = 0; q < R; ++q) { pso[q] = a; a += sb[q]; pro[q] = b; b += rb[q]; }
		for (int q = 0; q < R; ++q)
			if (sb[q] && hipMemcpyAsync((char*)c->d_send + pso[q], (const char*)send + so[q], sb[q], hipMemcpyHostToDevice, st) != hipSuccess) return c->fail(MCOM_E_HIP, "all-to-all: upload");
		int rc = rccl_alltoallv(c, (const char*)c->d_send, pso.data(), sb, (char*)c->d_recv, pro.data(), rb, st);
		if (rc) return rc;
		for (int q = 0; q < R; ++q)
			if (rb[q] && hipMemcpyAsync((char*)recv + ro[q], (const char*)c->d_recv + pro[q], rb[q], hipMemcpyDeviceToHost, st) != hipSuccess) return c->fail(MCOM_E_HIP, "all-to-all: download");
		if (hipStreamSynchronize(st) != hipSuccess) return c->fail(MCOM_E_HIP, "all-to-all: the stream failed");
		return MCOM_OK;
	}
	// caller-supplied host transport
	if (!on_device) {
		const int rc = c->ops.alltoallv(c->user, send, so, sb, recv, ro, rb);
		return rc ? c->fail(MCOM_E_HIP, "all-to-all: the transport callback returned %d", rc) : MCOM_OK;
	}
	if (!c->host_stage(c->h_send, c->h_send_cap, (size_t)ts + 16) || !c->host_stage(c->h_recv, c->h_recv_cap, (size_t)tr + 16)) return c->fail(MCOM_E_NOMEM, "all-to-all: pinned staging");
	std::vector<uint64_t> pso(R), pro(R);
	uint64_t a = 0, b = 0;
	for (int q = 0; q < R; ++q) { pso[q] = a; a += sb[q]; pro[q] = b; b += rb[q]; }
	for (int q = 0; q < R; ++q)
		if (sb[q] && hipMemcpyAsync((char*)c->h_send + pso[q], (const char*)send + so[q], sb[q], hipMemcpyDeviceToHost, st) != hipSuccess) return c->fail(MCOM_E_HIP, "all-to-all: download");
	if (hipStreamSynchronize(st) != hipSuccess) return c->fail(MCOM_E_HIP, "all-to-all: the stream failed");
	const int rc = c->ops.alltoallv(c->user, c->h_send, pso.data(), sb, c->h_recv, pro.data(), rb);
	if (rc) return c->fail(MCOM_E_HIP, "all-to-all: the transport callback returned %d", rc);
	for (int q = 0; q < R; ++q)
		if (rb[q] && hipMemcpyAsync((char*)recv + ro[q], (const char*)c->h_recv + pro[q], rb[q], hipMemcpyHostToDevice, st) != hipSuccess) return c->fail(MCOM_E_HIP, "all-to-all: upload");
	if (hipStreamSynchronize(st) != hipSuccess) return c->fail(MCOM_E_HIP, "all-to-all: the stream failed");
	return MCOM_OK;
}

extern "C" int mcomh_comm_allgatherv(mcomh_comm *c, const void *send, void *buf, const uint64_t *off, const uint64_t *bytes, int on_device, void *hip_stream)
{
	if (!c || !off || !bytes) return MCOM_E_ARG;
	const int R = c->world, me = c->rank;
	const char *src = send ? (const char*)send : (const char*)buf + off[me];
	// this rank's part into its own place (no transport involved), the same part to every other rank
	if (send && bytes[me] && (const char*)send != (const char*)buf + off[me]) {
		if (on_device) {
			if (hipMemcpyAsync((char*)buf + off[me], send, bytes[me], hipMemcpyDeviceToDevice, (hipStream_t)hip_stream) != hipSuccess) return c->fail(MCOM_E_HIP, "all-gather: copy");
		} else memmove((char*)buf + off[me], send, bytes[me]);
	}
	std::vector<uint64_t> so(R, 0), sb(R), rb(R);
	for (int q = 0; q < R; ++q) { sb[q] = q == me ? 0 : bytes[me]; rb[q] = q == me ? 0 : bytes[q]; }
	return mcomh_comm_alltoallv(c, src, so.data(), sb.data(), buf, off, rb.data(), on_device, hip_stream);
}

extern "C" int mcomh_comm_allreduce_u64(mcomh_comm *c, uint64_t *vals, size_t n, int op)
{
	if (!c || (n && !vals) || op < 0 || op > 2) return MCOM_E_ARG;
	const int R = c->world;
	if (R == 1 || n == 0) return MCOM_OK;
	std::vector<uint64_t> all((size_t)R * n), off(R), bytes(R);
	for (int q = 0; q < R; ++q) { off[q] = (uint64_t)q * n * 8; bytes[q] = n * 8; }
	memcpy(all.data() + (size_t)c->rank * n, vals, n * 8);
	const int rc = mcomh_comm_allgatherv(c, nullptr, all.data(), off.data(), bytes.data(), 0, nullptr);
	if (rc) return rc;
	for (size_t i = 0; i < n; ++i) {
		uint64_t v = all[i];
		for (int q = 1; q < R; ++q) {
			const uint64_t w = all[(size_t)q * n + i];
			v = op == 0 ? v + w : op == 1 ? (w < v ? w : v) : (w > v ? w : v);
		}
		vals[i] = v;
	}
	return MCOM_OK;
}
