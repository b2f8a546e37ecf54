// minicom_amd/host/mcom_pipeline.cpp -- host driver of the MI355X-native minicom hot path.
//
// Restates the control flow of the reference's pre_process (preprocess.c:39-241) and of its stage drivers,
// with every hot loop replaced by a call into libmcom_hip.so (include/mcom.h).  What stays on the host is
// what the reference keeps sequential by design: contig consensus, first-come pair claiming and the
// resolution of Stage-2 claims.  Citations are file:line into yuansliu/minicom src/.
#include "../../include/mcom.h"
#include "../../include/mcom_host.h"
#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <chrono>
#include <cinttypes>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int NB_BITS = 14;                 // MM_IDX_DEF_B, minicommain.c:175
constexpr uint64_t U64MAX = ~0ull;

struct Contig {
	std::vector<uint64_t> a;                // rid<<32 | offset<<1 | dir   (breads.h:49-58)
	std::string ref;
};

template <class T> struct DevBuf {
	T *p = nullptr; size_t cap = 0;
	~DevBuf() { if (p) (void)hipFree(p); }
	bool reserve(size_t n) {
		if (n <= cap) return true;
		if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
		size_t want = n + n / 8 + 64;
		if (hipMalloc(&p, want * sizeof(T)) != hipSuccess) { p = nullptr; return false; }
		cap = want; return true;
	}
};

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

} // namespace

struct mcomh_pipeline {
	mcom_ctx *ctx = nullptr;
	hipStream_t stream = nullptr;
	std::string err;
	size_t n = 0; int L = 0, W = 0, NW = 0;
	int k = 0, e = 0, m = 0, rw = 0, cbthr = 0, max_rounds = 0, step = 0, maxthr = 0, numdict = 0, maxsearch = 500;
	int host_threads = 1;
	// device
	DevBuf<uint8_t> d_ascii_own; const uint8_t *d_ascii = nullptr; size_t pitch = 0;
	const uint64_t *ext_packed = nullptr;    // packed-row input (mcomh_create_packed): no classification stage
	DevBuf<uint64_t> d_packed, d_nmask; DevBuf<uint8_t> d_cls; DevBuf<uint16_t> d_ncnt; DevBuf<mcom_mm128> d_rec;
	// host
	std::vector<uint8_t> h_ascii;            // only when the reads came from the host (needed for the N dump)
	std::vector<uint64_t> h_packed; std::vector<uint8_t> h_cls;
	std::vector<uint32_t> allA, allT, allN, fpA, fpT, fpN, Nfile, sg;
	std::vector<uint8_t> sg_flag;
	std::vector<Contig> C[2]; int idxv = 0;
	std::vector<uint8_t> unsorted;           // Stage 2: contigs whose member list changed since it was last sorted
	std::vector<mcom_mm128> mi0;             // first-m minimizers of the Stage-1 contigs, contig order
	bool stage2_uploaded = false;
	DevBuf<uint8_t> d_cseq; DevBuf<uint64_t> d_coff_chars, d_coff_words, d_cbits, d_woff; DevBuf<uint32_t> d_clen;
	uint64_t n_windows = 0;
	std::map<std::string, double> stat;

	int fail(int code, const char *fmt, ...) {
		char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
		err = buf; return code;
	}
	int gpu(int rc) { if (rc) err = std::string("libmcom_hip: ") + mcom_last_error(ctx); return rc; }
	int hipc(hipError_t e_, const char *what) { if (e_ != hipSuccess) return fail(MCOM_E_HIP, "%s: %s", what, hipGetErrorString(e_)); return 0; }

	inline int base(uint32_t rid, int i) const { return (int)((h_packed[(size_t)rid * W + (i >> 5)] >> (2 * (i & 31))) & 3); }
	// base i of the read as laid on the contig: reverse complement when dir = 1 (reverse_complement, preprocess.c:22)
	inline int obase(uint32_t rid, int dir, int i) const { return dir ? 3 - base(rid, L - 1 - i) : base(rid, i); }
};

using P = mcomh_pipeline;
static const char ACGT[] = "ACGT";

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
	p->host_threads = pp->host_threads > 0 ? pp->host_threads : 1;
	if (p->k > 31 || p->k < 11 || p->rw < 1 || p->rw > 128) { delete p; return MCOM_E_ARG; }
	if (host_reads) {
		p->pitch = (size_t)L;
		p->h_ascii.assign(host_reads, host_reads + n * (size_t)L);
		if (!p->d_ascii_own.reserve(n * (size_t)L + 16)) { mcom_destroy(p->ctx); delete p; return MCOM_E_NOMEM; }
		if (n && hipMemcpyAsync(p->d_ascii_own.p, host_reads, n * (size_t)L, hipMemcpyHostToDevice, p->stream) != hipSuccess) { mcom_destroy(p->ctx); delete p; return MCOM_E_HIP; }
		p->d_ascii = p->d_ascii_own.p;
	} else { p->d_ascii = d_reads; p->pitch = pitch; }
	*out = p;
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

extern "C" void mcomh_destroy(mcomh_pipeline *p)
{
	if (!p) return;
	(void)hipStreamSynchronize(p->stream);
	if (p->ctx) mcom_destroy(p->ctx);
	delete p;
}

extern "C" const char *mcomh_last_error(const mcomh_pipeline *p) { return p ? p->err.c_str() : "null pipeline"; }

// ----------------------------------------------------------------------------------------------------
// kt_for_reads                                                             kthread_reads.c:247, :40-230
// ----------------------------------------------------------------------------------------------------
extern "C" int mcomh_kt_for_reads(mcomh_pipeline *p)
{
	if (!p) return MCOM_E_ARG;
	const double t0 = now_ms();
	const size_t n = p->n;
	if (!p->d_packed.reserve(n * p->W + 1) || !p->d_nmask.reserve(n * p->NW + 1) || !p->d_cls.reserve(n + 1) ||
	    !p->d_ncnt.reserve(n + 1) || !p->d_rec.reserve(n + 1)) return p->fail(MCOM_E_NOMEM, "read buffers");
	int rc;
	if (p->ext_packed) {
		// packed rows handed over by the caller: every read is a kept read (class 0) without N
		if (n && ((rc = p->hipc(hipMemcpyAsync(p->d_packed.p, p->ext_packed, n * (size_t)p->W * 8, hipMemcpyDeviceToDevice, p->stream), "copy packed rows")) ||
		          (rc = p->hipc(hipMemsetAsync(p->d_cls.p, 0, n, p->stream), "clear")) ||
		          (rc = p->hipc(hipMemsetAsync(p->d_nmask.p, 0, n * (size_t)p->NW * 8, p->stream), "clear")))) return rc;
		rc = p->gpu(mcom_sketch_reads(p->ctx, p->d_packed.p, nullptr, n, p->L, p->k, 0, p->d_rec.p));
	} else {
		rc = p->gpu(mcom_process_reads(p->ctx, p->d_ascii, p->pitch, n, p->L, p->k, p->e, 0, p->d_packed.p, p->d_cls.p, p->d_ncnt.p, p->d_nmask.p, p->d_rec.p));
	}
	if (rc) return rc;
	p->h_cls.resize(n); p->h_packed.resize(n * (size_t)p->W);
	if (n) {
		if ((rc = p->hipc(hipMemcpyAsync(p->h_cls.data(), p->d_cls.p, n, hipMemcpyDeviceToHost, p->stream), "copy classes"))) return rc;
		if ((rc = p->hipc(hipMemcpyAsync(p->h_packed.data(), p->d_packed.p, n * (size_t)p->W * 8, hipMemcpyDeviceToHost, p->stream), "copy packed reads"))) return rc;
	}
	if ((rc = p->hipc(hipStreamSynchronize(p->stream), "kt_for_reads"))) return rc;
	for (size_t r = 0; r < n; ++r) {                                              // one thread: rid order
		switch (p->h_cls[r]) {
		case MCOM_CLS_ALLA: p->allA.push_back((uint32_t)r); break;
		case MCOM_CLS_ALLT: p->allT.push_back((uint32_t)r); break;
		case MCOM_CLS_ALLN: p->allN.push_back((uint32_t)r); break;
		case MCOM_CLS_NEARA: p->fpA.push_back((uint32_t)r); break;
		case MCOM_CLS_NEART: p->fpT.push_back((uint32_t)r); break;
		case MCOM_CLS_NEARN: p->fpN.push_back((uint32_t)r); break;
		case MCOM_CLS_NHEAVY: p->Nfile.push_back((uint32_t)r); break;
		default: break;
		}
	}
	p->stat["t_reads"] += now_ms() - t0;
	return MCOM_OK;
}

// ----------------------------------------------------------------------------------------------------
// construct_ref: consensus of one minimizer group                           kthread_bucket.c:69-377
//   members: y values in cmpcluster order.  Keeps the members within e mismatches of the first consensus,
//   rebuilds the consensus from them, returns the rejected rids.
// ----------------------------------------------------------------------------------------------------
struct GroupOut { Contig c; std::vector<uint32_t> rejected; };

static void construct_ref(const P *p, const uint64_t *y, size_t n, GroupOut &out, std::vector<uint32_t> &cnt)
{
	const int L = p->L;
	const int tlen = L << 2;                        // "readlen<<1 + 1" parses as readlen << 2 (:72)
	cnt.assign((size_t)4 * tlen, 0);
	std::vector<uint64_t> &a = out.c.a;
	a.assign(y, y + n);
	int pos0 = L;
	for (size_t q = 0; q < n; ++q) {
		const uint64_t v = a[q];
		const uint32_t rid = (uint32_t)(v >> 32); int pos = (int)((uint32_t)v >> 1); const int dir = (int)(v & 1);
		if (dir) pos = L - pos + p->k - 2;                                          // :93 (the run's first k, always)
		if (q == 0) pos0 = pos;
		const int off = pos0 - pos;
		for (int s = 0; s < L; ++s) ++cnt[(size_t)p->obase(rid, dir, s) * tlen + off + s];
		a[q] = (v >> 32 << 32) | ((uint64_t)off << 1) | (uint64_t)dir;
	}
	std::string ref; ref.reserve(2 * L + 64);
	for (int s = 0; s < tlen; ++s) {
		uint32_t mx = cnt[s]; int b = 0;
		for (int q = 1; q < 4; ++q) if (cnt[(size_t)q * tlen + s] > mx) { mx = cnt[(size_t)q * tlen + s]; b = q; }
		if (mx == 0) break;
		ref.push_back(ACGT[b]);
	}
	const int ref_len = (int)ref.size();
	size_t kept = 0;
	for (size_t q = 0; q < n; ++q) {
		const uint64_t v = a[q];
		const uint32_t rid = (uint32_t)(v >> 32); const int pos = (int)((uint32_t)v >> 1), dir = (int)(v & 1);
		int dif = 0;
		for (int s = 0; s < L; ++s) if (ref[pos + s] != ACGT[p->obase(rid, dir, s)]) ++dif;
		if (dif <= p->e) a[kept++] = v; else out.rejected.push_back(rid);           // :189-213
	}
	a.resize(kept);
	out.c.ref = ref;
	if (kept > 0) {                                                                 // :244-352
		std::fill(cnt.begin(), cnt.end(), 0u);
		int rend = 0;
		for (size_t q = 0; q < kept; ++q) {
			const uint64_t v = a[q];
			const uint32_t rid = (uint32_t)(v >> 32); const int pos = (int)((uint32_t)v >> 1), dir = (int)(v & 1);
			for (int s = 0; s < L; ++s) ++cnt[(size_t)p->obase(rid, dir, s) * tlen + pos + s];
			if (pos + L > rend) rend = pos + L;
		}
		int s = 0;
		for (; s < ref_len; ++s) {
			uint32_t mx = cnt[s];
			for (int q = 1; q < 4; ++q) if (cnt[(size_t)q * tlen + s] > mx) mx = cnt[(size_t)q * tlen + s];
			if (mx != 0) break;
		}
		const int sv = s;
		std::string r2; r2.reserve((size_t)(rend - sv));
		for (; s < rend; ++s) {
			uint32_t mx = cnt[s]; int b = 0;
			for (int q = 1; q < 4; ++q) if (cnt[(size_t)q * tlen + s] > mx) { mx = cnt[(size_t)q * tlen + s]; b = q; }
			r2.push_back(ACGT[b]);
		}
		out.c.ref.swap(r2);
		for (size_t q = 0; q < kept; ++q) {
			const uint64_t v = a[q];
			a[q] = (v >> 32 << 32) | ((uint64_t)((int)((uint32_t)v >> 1) - sv) << 1) | (v & 1);
		}
	}
}

// cmpcluster2: offset ascending, then direction (kthread_cb.c:54-69); the reference's qsort is glibc's merge sort
static bool less_cluster2(uint64_t a, uint64_t b)
{
	const int pa = (int)((uint32_t)a >> 1), pb = (int)((uint32_t)b >> 1);
	if (pa != pb) return pa < pb;
	return (int)(a & 1) < (int)(b & 1);
}

// construct_ref2: consensus of a merged contig                                 kthread_cb.c:105-218
static void construct_ref2(const P *p, Contig &c, std::vector<uint32_t> &cnt)
{
	const int L = p->L;
	std::stable_sort(c.a.begin(), c.a.end(), less_cluster2);
	const int tlen = (int)((uint32_t)c.a.back() >> 1) + (L << 1) + 1;
	cnt.assign((size_t)4 * tlen, 0);
	int rend = 0;
	for (uint64_t v : c.a) {
		const uint32_t rid = (uint32_t)(v >> 32); const int pos = (int)((uint32_t)v >> 1), dir = (int)(v & 1);
		for (int s = 0; s < L; ++s) ++cnt[(size_t)p->obase(rid, dir, s) * tlen + pos + s];
		if (pos + L > rend) rend = pos + L;
	}
	c.ref.assign((size_t)rend, 'A');
	for (int s = 0; s < rend; ++s) {
		uint32_t mx = cnt[s]; int b = 0;
		for (int q = 1; q < 4; ++q) if (cnt[(size_t)q * tlen + s] > mx) { mx = cnt[(size_t)q * tlen + s]; b = q; }
		c.ref[s] = ACGT[b];
	}
}

// ----------------------------------------------------------------------------------------------------
// kt_for_bucket: Stage-1 rounds                                               kthread_bucket.c:562-629
// ----------------------------------------------------------------------------------------------------
extern "C" int mcomh_kt_for_bucket(mcomh_pipeline *p)
{
	if (!p) return MCOM_E_ARG;
	const double t0 = now_ms();
	const int L = p->L;
	size_t n_cur = p->n;
	DevBuf<mcom_mm128> d_cur, d_sorted; DevBuf<uint32_t> d_singles, d_sord, d_goff, d_rids; DevBuf<uint64_t> d_members;
	const mcom_mm128 *cur = p->d_rec.p;                      // round 1 works on the records of kt_for_reads
	std::vector<uint32_t> h_singles, h_sord, h_goff, resk;
	std::vector<uint64_t> h_members;
	std::vector<Contig> &C0 = p->C[0];
	int last_rounds = 0; long pre = 0;
	for (int r = 1;; ++r) {
		if (p->k - r <= 9) ++last_rounds;                                           // :584-585
		if (r == p->max_rounds - 1) ++last_rounds;
		const bool last = last_rounds != 0;
		const int kmer_in = p->k - (r - 1);                  // k the incoming records were sketched with
		const int kmer_next = p->k - r;                       // k for the rejects of this round (:592)
		resk.clear();
		if (n_cur) {
			if (!d_sorted.reserve(n_cur) || !d_singles.reserve(n_cur) || !d_sord.reserve(n_cur) || !d_members.reserve(n_cur) || !d_goff.reserve(n_cur / 2 + 2))
				return p->fail(MCOM_E_NOMEM, "round buffers");
			uint64_t cnts[4];
			const double tg = now_ms();
			p->stat["sort_records"] += (double)n_cur * (double)((2 * kmer_in + 10 + 7) / 8);   // records x LSD passes
			int rc = p->gpu(mcom_sort_group(p->ctx, cur, n_cur, L, p->k, kmer_in, NB_BITS, d_sorted.p, d_singles.p, d_sord.p, d_members.p, d_goff.p, cnts));
			if (rc) return rc;
			const size_t ns = cnts[1], ng = cnts[2], nm = cnts[3];
			h_singles.resize(ns); h_sord.resize(ns); h_members.resize(nm); h_goff.resize(ng + 1);
			if (ns) { (void)hipMemcpyAsync(h_singles.data(), d_singles.p, ns * 4, hipMemcpyDeviceToHost, p->stream); (void)hipMemcpyAsync(h_sord.data(), d_sord.p, ns * 4, hipMemcpyDeviceToHost, p->stream); }
			if (nm) (void)hipMemcpyAsync(h_members.data(), d_members.p, nm * 8, hipMemcpyDeviceToHost, p->stream);
			(void)hipMemcpyAsync(h_goff.data(), d_goff.p, (ng + 1) * 4, hipMemcpyDeviceToHost, p->stream);
			if ((rc = p->hipc(hipStreamSynchronize(p->stream), "round copy"))) return rc;
			p->stat["t_gpu"] += now_ms() - tg;
			p->stat["t_bk_gpu"] += now_ms() - tg;
			const double tb1 = now_ms();
			// consensus of every group; groups are independent, only the appends below are ordered
			std::vector<GroupOut> outs(ng);
			const int nt = std::max(1, std::min<int>(p->host_threads, (int)std::max<size_t>(1, ng / 64)));
			auto work = [&](int tid) {
				std::vector<uint32_t> cnt;
				for (size_t g = (size_t)tid; g < ng; g += (size_t)nt)
					construct_ref(p, h_members.data() + h_goff[g], h_goff[g + 1] - h_goff[g], outs[g], cnt);
			};
			if (nt == 1) work(0);
			else { std::vector<std::thread> th; for (int t = 0; t < nt; ++t) th.emplace_back(work, t); for (auto &t : th) t.join(); }
			const double tb2 = now_ms();
			p->stat["t_bk_cons"] += tb2 - tb1;
			// replay in the reference's visiting order (process_bucket, :398-505)
			size_t si = 0;
			auto reject = [&](uint32_t rid) { if (last) p->sg.push_back(rid); else resk.push_back(rid); };
			for (size_t g = 0; g <= ng; ++g) {
				while (si < ns && h_sord[si] == g) p->sg.push_back(h_singles[si++]);      // groups of one (:402-413)
				if (g == ng) break;
				GroupOut &o = outs[g];
				for (uint32_t rid : o.rejected) reject(rid);                                  // :194-213
				if (o.c.a.size() > 1) C0.emplace_back(std::move(o.c));                        // :451-475
				else if (o.c.a.size() == 1) reject((uint32_t)(o.c.a[0] >> 32));              // :477-498
			}
			p->stat["t_bk_replay"] += now_ms() - tb2;
		}
		p->stat["rounds"] += 1;
		if (last_rounds) ++last_rounds;                                             // :594
		long cr = 0;
		for (const Contig &c : C0) cr += (long)c.a.size();
		if (cr - pre < 100) ++last_rounds;                                          // :614-618
		pre = cr;
		if (last_rounds > 1) break;
		// rejected reads are sketched again with a shorter k (:205, :489); ascending rid keeps the sort's tie rule
		std::sort(resk.begin(), resk.end());
		n_cur = resk.size();
		p->stat["resketch"] += (double)n_cur;
		if (n_cur) {
			if (!d_rids.reserve(n_cur) || !d_cur.reserve(n_cur)) return p->fail(MCOM_E_NOMEM, "re-sketch buffers");
			int rc = p->hipc(hipMemcpyAsync(d_rids.p, resk.data(), n_cur * 4, hipMemcpyHostToDevice, p->stream), "upload rids");
			if (rc) return rc;
			if ((rc = p->gpu(mcom_sketch_reads(p->ctx, p->d_packed.p, d_rids.p, n_cur, L, kmer_next, 0, d_cur.p)))) return rc;
			cur = d_cur.p;
		}
	}
	if (p->sg.size() <= 5000000) p->maxsearch = 2000;                               // preprocess.c:169-172
	p->stat["n_sg0"] = (double)p->sg.size();
	p->stat["t_bucket"] += now_ms() - t0;
	return MCOM_OK;
}

// ----------------------------------------------------------------------------------------------------
// contigs on the device
// ----------------------------------------------------------------------------------------------------
struct DevContigs {
	DevBuf<uint8_t> seq; DevBuf<uint64_t> off, coff, cbits; DevBuf<uint32_t> clen;
	std::vector<uint64_t> h_off, h_coff; std::vector<uint32_t> h_len;
	uint64_t total_words = 0; size_t n = 0;
};

static int upload_contigs(P *p, const std::vector<Contig> &cs, DevContigs &d, bool pack)
{
	const size_t n = cs.size();
	d.n = n;
	d.h_off.assign(n + 1, 0); d.h_coff.assign(n + 1, 0); d.h_len.assign(n, 0);
	for (size_t i = 0; i < n; ++i) {
		d.h_len[i] = (uint32_t)cs[i].ref.size();
		d.h_off[i + 1] = d.h_off[i] + cs[i].ref.size();
		d.h_coff[i + 1] = d.h_coff[i] + (2 * cs[i].ref.size() + 63) / 64 + 1;
	}
	d.total_words = d.h_coff[n];
	std::string cat; cat.reserve(d.h_off[n] + 1);
	for (const Contig &c : cs) cat += c.ref;
	if (!d.seq.reserve(cat.size() + 16) || !d.off.reserve(n + 1) || !d.coff.reserve(n + 1) || !d.clen.reserve(n + 1) || (pack && !d.cbits.reserve(d.total_words + 2)))
		return p->fail(MCOM_E_NOMEM, "contig buffers");
	int rc;
	if (cat.size() && (rc = p->hipc(hipMemcpyAsync(d.seq.p, cat.data(), cat.size(), hipMemcpyHostToDevice, p->stream), "upload contigs"))) return rc;
	if ((rc = p->hipc(hipMemcpyAsync(d.off.p, d.h_off.data(), (n + 1) * 8, hipMemcpyHostToDevice, p->stream), "upload offsets"))) return rc;
	if ((rc = p->hipc(hipMemcpyAsync(d.coff.p, d.h_coff.data(), (n + 1) * 8, hipMemcpyHostToDevice, p->stream), "upload offsets"))) return rc;
	if (n && (rc = p->hipc(hipMemcpyAsync(d.clen.p, d.h_len.data(), n * 4, hipMemcpyHostToDevice, p->stream), "upload lengths"))) return rc;
	if (pack && n) {
		if ((rc = p->hipc(hipMemsetAsync(d.cbits.p, 0, (d.total_words + 2) * 8, p->stream), "clear"))) return rc;
		if ((rc = p->gpu(mcom_pack_contigs(p->ctx, d.seq.p, d.off.p, d.coff.p, (uint32_t)n, d.total_words, d.cbits.p)))) return rc;
	}
	// the host vectors must outlive the async copies
	return p->hipc(hipStreamSynchronize(p->stream), "upload contigs");
}

// mm_sketch_lh_ori of every contig; max_per = 0 keeps all minimizers
static int sketch_contigs(P *p, const DevContigs &d, uint32_t max_per, DevBuf<uint32_t> &moff, DevBuf<mcom_mm128> &out, uint64_t &total)
{
	total = 0;
	p->stat["sketch_bases"] += 2.0 * (double)d.h_off[d.n];          // one count launch + one emit launch
	if (!moff.reserve(d.n + 2)) return p->fail(MCOM_E_NOMEM, "minimizer offsets");
	size_t cap = std::max<size_t>(1024, max_per ? d.n * max_per : d.h_off[d.n] / 8 + d.n);
	for (int attempt = 0; attempt < 2; ++attempt) {
		if (!out.reserve(cap)) return p->fail(MCOM_E_NOMEM, "minimizer records");
		int rc = mcom_sketch_contigs(p->ctx, d.seq.p, d.off.p, nullptr, d.n, p->rw, p->k, max_per, moff.p, out.p, out.cap, &total);
		if (rc == MCOM_E_OVERFLOW) { cap = total; continue; }
		return p->gpu(rc);
	}
	return p->fail(MCOM_E_OVERFLOW, "minimizer buffer");
}

// ----------------------------------------------------------------------------------------------------
// combine_cluster: merge rounds                                                kthread_cb.c:570-630
// ----------------------------------------------------------------------------------------------------
extern "C" int mcomh_combine_cluster(mcomh_pipeline *p)
{
	if (!p) return MCOM_E_ARG;
	const double t0 = now_ms();
	int index = 0; long pre = 0;
	DevContigs dc; DevBuf<uint32_t> moff_m, moff_all; DevBuf<mcom_mm128> rec_m, rec_all, d_pairs;
	std::vector<mcom_mm128> pairs; std::vector<uint32_t> cnt;
	for (;;) {
		std::vector<Contig> &src = p->C[index], &dst = p->C[index ^ 1];
		dst.clear();
		const size_t n = src.size();
		uint64_t n_pass = 0;
		if (n) {
			const double tg = now_ms();
			double tl = tg;
			auto lap = [&](const char *nm) { (void)hipStreamSynchronize(p->stream); const double t = now_ms(); p->stat[nm] += t - tl; tl = t; };
			int rc = upload_contigs(p, src, dc, true);
			if (rc) return rc;
			lap("t_cb_upload");
			uint64_t tm = 0, ta = 0;
			if ((rc = sketch_contigs(p, dc, (uint32_t)p->m, moff_m, rec_m, tm))) return rc;
			lap("t_cb_sketch");       // what the builders pushed to mi[index] (:370-380, :423-432)
			if (index == 0 && p->mi0.empty() && tm) { p->mi0.resize(tm); (void)hipMemcpy(p->mi0.data(), rec_m.p, tm * sizeof(mcom_mm128), hipMemcpyDeviceToHost); }
			mcom_idx *mi = nullptr;
			if ((rc = p->gpu(mcom_idx_build(p->ctx, rec_m.p, tm, p->k, &mi)))) return rc;          // mm_idx_generation (:580)
			lap("t_cb_idx");
			if ((rc = sketch_contigs(p, dc, 0, moff_all, rec_all, ta))) { mcom_idx_destroy(p->ctx, mi); return rc; }   // find_next's own sketch (:234)
			lap("t_cb_sketch");
			uint64_t hc[2] = {0, 0};
			size_t cap = std::max<size_t>(1024, ta);
			for (int attempt = 0; attempt < 2; ++attempt) {
				if (!d_pairs.reserve(cap)) { mcom_idx_destroy(p->ctx, mi); return p->fail(MCOM_E_NOMEM, "candidate pairs"); }
				rc = mcom_find_next_candidates(p->ctx, mi, rec_all.p, ta, dc.cbits.p, dc.coff.p, dc.clen.p, p->cbthr, d_pairs.p, d_pairs.cap, hc);
				if (rc == MCOM_E_OVERFLOW) { cap = hc[1]; continue; }
				break;
			}
			mcom_idx_destroy(p->ctx, mi);
			if (rc) return p->gpu(rc);
			lap("t_cb_findnext");
			n_pass = hc[1];
			pairs.resize(n_pass);
			if (n_pass && (rc = p->hipc(hipMemcpy(pairs.data(), d_pairs.p, n_pass * sizeof(mcom_mm128), hipMemcpyDeviceToHost), "copy candidates"))) return rc;
			lap("t_cb_d2h");
			p->stat["t_gpu"] += now_ms() - tg;
			p->stat["cand_pairs"] += (double)hc[0];
		}
		// first-come claiming in contig order (find_next :267-343 at one thread)
		const double tc0 = now_ms();
		std::vector<uint8_t> flag(n, 0);
		struct Job { uint32_t ci, cj, pos_ori, pos; };
		std::vector<Job> jobs;
		for (size_t q = 0; q < n_pass;) {
			const uint32_t ci = (uint32_t)(pairs[q].x >> 32) >> 8;
			size_t qe = q;
			while (qe < n_pass && ((uint32_t)(pairs[qe].x >> 32) >> 8) == ci) ++qe;
			if (!flag[ci]) {
				for (size_t u = q; u < qe; ++u) {
					const uint32_t cj = (uint32_t)(pairs[u].y >> 32) >> 8;
					if (flag[cj]) continue;
					jobs.push_back(Job{ci, cj, (uint32_t)pairs[u].x >> 1, (uint32_t)pairs[u].y >> 1});
					flag[ci] = flag[cj] = 1;                                                // :339-343
					break;
				}
			}
			q = qe;
		}
		// the merged contigs themselves are independent of each other: build them on all host threads
		const double tc1 = now_ms();
		p->stat["t_claim"] += tc1 - tc0;
		dst.resize(jobs.size());
		{
			const int nt = std::max(1, std::min<int>(p->host_threads, (int)std::max<size_t>(1, jobs.size() / 16)));
			auto work = [&](int tid) {
				std::vector<uint32_t> lcnt;
				for (size_t j = (size_t)tid; j < jobs.size(); j += (size_t)nt) {
					const Job &jb = jobs[j];
					const Contig &a = src[jb.ci], &b = src[jb.cj];
					Contig &t = dst[j];
					t.a.reserve(a.a.size() + b.a.size());
					if (jb.pos_ori >= jb.pos) {                                             // :302-315
						t.a.insert(t.a.end(), a.a.begin(), a.a.end());
						for (uint64_t y : b.a) t.a.push_back((y >> 32 << 32) | (((uint64_t)((uint32_t)y >> 1) + (uint64_t)(jb.pos_ori - jb.pos)) << 1) | (y & 1));
					} else {                                                                // :316-325
						t.a.insert(t.a.end(), b.a.begin(), b.a.end());
						for (uint64_t y : a.a) t.a.push_back((y >> 32 << 32) | (((uint64_t)((uint32_t)y >> 1) + (uint64_t)(jb.pos - jb.pos_ori)) << 1) | (y & 1));
					}
					construct_ref2(p, t, lcnt);
				}
			};
			if (nt == 1) work(0);
			else { std::vector<std::thread> th; for (int t = 0; t < nt; ++t) th.emplace_back(work, t); for (auto &t : th) t.join(); }
		}
		const double tc2 = now_ms();
		p->stat["t_merge_cons"] += tc2 - tc1;
		for (size_t i = 0; i < n; ++i) if (!flag[i]) dst.emplace_back(std::move(src[i]));   // cp_cluster (:397-434)
		const double tc3 = now_ms();
		p->stat["t_cb_copy"] += tc3 - tc2;
		src.clear();
		p->stat["t_cb_free"] += now_ms() - tc3;
		p->stat["merge_rounds"] += 1;
		index ^= 1;
		const long tot = (long)p->C[index].size();
		if (std::labs(pre - tot) < 100) break;                                              // :625
		pre = tot;
	}
	p->idxv = index;
	p->unsorted.assign(p->C[index].size(), 1);
	p->sg_flag.assign(p->sg.size(), 0);                                                     // preprocess.c:182
	p->stage2_uploaded = false;
	p->stat["t_combine"] += now_ms() - t0;
	return MCOM_OK;
}

// ----------------------------------------------------------------------------------------------------
// updateSingle                                                                 preprocess.c:243-255
// ----------------------------------------------------------------------------------------------------
extern "C" int mcomh_update_single(mcomh_pipeline *p)
{
	if (!p) return MCOM_E_ARG;
	size_t nn = 0;
	for (size_t i = 0; i < p->sg.size(); ++i) if (!p->sg_flag[i]) p->sg[nn++] = p->sg[i];
	p->sg.resize(nn);
	p->sg_flag.assign(nn, 0);
	return MCOM_OK;
}

// ----------------------------------------------------------------------------------------------------
// realign_hash: one Stage-2 pass                                    kthread_hash_realign.c:569-594
// ----------------------------------------------------------------------------------------------------
extern "C" int mcomh_realign_hash(mcomh_pipeline *p, int thr, long *cluster_reads)
{
	if (!p) return MCOM_E_ARG;
	const double t0 = now_ms();
	mcomh_update_single(p);                                                                 // preprocess.c:203
	std::vector<Contig> &cs = p->C[p->idxv];
	const size_t n_sg = p->sg.size();
	int rc;
	if (!p->stage2_uploaded) {                      // contig consensus strings do not change during Stage 2
		DevContigs dc;
		// reuse the pipeline-owned buffers: move them in and out of the helper struct
		if ((rc = upload_contigs(p, cs, dc, true))) return rc;
		std::vector<uint64_t> woff(cs.size() + 1, 0);
		for (size_t i = 0; i < cs.size(); ++i) woff[i + 1] = woff[i] + (dc.h_len[i] >= (uint32_t)p->L ? dc.h_len[i] - p->L + 1 : 0);
		p->n_windows = woff[cs.size()];
		if (!p->d_woff.reserve(cs.size() + 1)) return p->fail(MCOM_E_NOMEM, "window offsets");
		if ((rc = p->hipc(hipMemcpy(p->d_woff.p, woff.data(), (cs.size() + 1) * 8, hipMemcpyHostToDevice), "upload window offsets"))) return rc;
		std::swap(p->d_cbits.p, dc.cbits.p); std::swap(p->d_cbits.cap, dc.cbits.cap);
		std::swap(p->d_coff_words.p, dc.coff.p); std::swap(p->d_coff_words.cap, dc.coff.cap);
		p->stage2_uploaded = true;
	}
	p->stat["passes"] += 1;
	p->stat["windows"] += (double)p->n_windows;
	// every contig is re-sorted at the start of its scan (:318)
	const double tr0 = now_ms();
	{   // a stable sort of an already sorted list is the identity: only contigs that changed are sorted again
		if (p->unsorted.size() != cs.size()) p->unsorted.assign(cs.size(), 1);
		const int nt = std::max(1, std::min<int>(p->host_threads, (int)std::max<size_t>(1, cs.size() / 4096)));
		auto work = [&](int tid) {
			for (size_t c = (size_t)tid; c < cs.size(); c += (size_t)nt)
				if (p->unsorted[c]) { std::stable_sort(cs[c].a.begin(), cs[c].a.end(), less_cluster2); p->unsorted[c] = 0; }
		};
		if (nt == 1) work(0);
		else { std::vector<std::thread> th; for (int t = 0; t < nt; ++t) th.emplace_back(work, t); for (auto &t : th) t.join(); }
	}
	p->stat["t_ra_sort"] += now_ms() - tr0;
	if (n_sg) {
		const double tg = now_ms();
		DevBuf<uint32_t> d_sg; DevBuf<uint64_t> d_sgbits, d_claim; DevBuf<uint8_t> d_flag;
		if (!d_sg.reserve(n_sg) || !d_sgbits.reserve(n_sg * p->W) || !d_claim.reserve(n_sg) || !d_flag.reserve(n_sg)) return p->fail(MCOM_E_NOMEM, "singleton buffers");
		if ((rc = p->hipc(hipMemcpyAsync(d_sg.p, p->sg.data(), n_sg * 4, hipMemcpyHostToDevice, p->stream), "upload singletons"))) return rc;
		if ((rc = p->gpu(mcom_gather_rows(p->ctx, p->d_packed.p, d_sg.p, n_sg, p->L, d_sgbits.p)))) return rc;           // singleRead2bitset
		if ((rc = p->gpu(mcom_poly_filter(p->ctx, d_sgbits.p, p->d_nmask.p, d_sg.p, n_sg, p->L, thr, d_flag.p)))) return rc;
		std::vector<uint8_t> pf(n_sg);
		if ((rc = p->hipc(hipMemcpyAsync(pf.data(), d_flag.p, n_sg, hipMemcpyDeviceToHost, p->stream), "copy flags"))) return rc;
		mcom_dicts *dicts = nullptr;
		if ((rc = p->gpu(mcom_dicts_build(p->ctx, d_sgbits.p, n_sg, p->L, p->numdict, &dicts)))) return rc;               // constructdictionary_realign
		{
			int nd = 0; uint32_t nk[16], mb[16];
			mcom_dicts_info(dicts, &nd, nk, mb);
			for (int j = 0; j < nd; ++j) if (mb[j] > (uint32_t)p->maxsearch) p->stat["big_bins"] += 1;
		}
		for (size_t i = 0; i < n_sg; ++i) {                                                  // bbhashdict.c:177-216, singleton order
			if (pf[i] == 1) { p->sg_flag[i] = 1; p->fpA.push_back(p->sg[i]); }
			else if (pf[i] == 2) { p->sg_flag[i] = 1; p->fpT.push_back(p->sg[i]); }
		}
		rc = p->gpu(mcom_realign_pass(p->ctx, dicts, d_sgbits.p, d_flag.p, p->d_cbits.p, p->d_coff_words.p, p->d_woff.p, (uint32_t)cs.size(),
		                              p->n_windows, thr, p->maxsearch, d_claim.p, nullptr));
		std::vector<uint64_t> claim(n_sg);
		if (!rc) rc = p->hipc(hipMemcpyAsync(claim.data(), d_claim.p, n_sg * 8, hipMemcpyDeviceToHost, p->stream), "copy claims");
		if (!rc) rc = p->hipc(hipStreamSynchronize(p->stream), "realign pass");
		mcom_dicts_free(p->ctx, dicts);
		if (rc) return rc;
		p->stat["t_gpu"] += now_ms() - tg;
		p->stat["t_ra_gpu"] += now_ms() - tg;
		// append in the order of the sequential scan: claim key ascending, singleton index descending (:388)
		std::vector<std::pair<uint64_t, uint32_t>> won;
		for (size_t i = 0; i < n_sg; ++i) if (claim[i] != U64MAX) won.emplace_back(claim[i], (uint32_t)i);
		std::sort(won.begin(), won.end(), [](const std::pair<uint64_t, uint32_t> &a, const std::pair<uint64_t, uint32_t> &b) {
			return a.first != b.first ? a.first < b.first : a.second > b.second; });
		for (const auto &w : won) {
			const uint64_t ck = w.first;
			const size_t c = (size_t)(ck >> 33); const uint64_t jj = (ck >> 5) & ((1ull << 28) - 1), dir = (ck >> 4) & 1;
			cs[c].a.push_back((uint64_t)p->sg[w.second] << 32 | (jj << 1) | dir);              // :408-409, :474-475
			p->unsorted[c] = 1;
			p->sg_flag[w.second] = 1;
		}
	}
	long cr = 0;
	for (const Contig &c : cs) cr += (long)c.a.size();
	if (cluster_reads) *cluster_reads = cr;
	p->stat["t_realign"] += now_ms() - t0;
	return MCOM_OK;
}

// ----------------------------------------------------------------------------------------------------
// the reference's loop control around the stages                             preprocess.c:141-233
// ----------------------------------------------------------------------------------------------------
static int run_stage2(P *p, FILE *f);

extern "C" int mcomh_pre_process(mcomh_pipeline *p)
{
	if (!p) return MCOM_E_ARG;
	int rc;
	if ((rc = mcomh_kt_for_reads(p))) return rc;
	if ((rc = mcomh_kt_for_bucket(p))) return rc;
	if ((rc = mcomh_combine_cluster(p))) return rc;
	return run_stage2(p, nullptr);
}

// ---- dump in the text format of oracle/refdump.cpp ---------------------------------------------------
static void dump_list(FILE *f, const char *name, const std::vector<uint32_t> &v)
{
	fprintf(f, "LIST %s %zu", name, v.size());
	for (uint32_t x : v) fprintf(f, " %u", x);
	fprintf(f, "\n");
}
static void dump_contigs(FILE *f, const char *stage, const std::vector<Contig> &cs)
{
	fprintf(f, "CLUSTERS %s %zu\n", stage, cs.size());
	for (const Contig &c : cs) {
		fprintf(f, "C %zu %s", c.a.size(), c.ref.c_str());
		for (uint64_t y : c.a) fprintf(f, " %" PRIu64, y);
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
	long pre = 0; int pass = 0;
	for (int thr = p->e;; thr += p->step) {                                                 // preprocess.c:197-232
		if (thr > p->maxthr) break;
		std::vector<uint32_t> before;
		if (f) for (size_t i = 0; i < p->sg.size(); ++i) if (!p->sg_flag[i]) before.push_back(p->sg[i]);
		long cr = 0;
		int rc = mcomh_realign_hash(p, thr, &cr);
		if (rc) return rc;
		if (f) {
			fprintf(f, "STAGE realign %d thr %d\n", pass, thr);
			dump_list(f, "sg_in", before);
			fprintf(f, "SGFLAG %zu", p->sg.size());
			for (uint8_t v : p->sg_flag) fprintf(f, " %d", v ? 1 : 0);
			fprintf(f, "\n");
			dump_list(f, "fpA", p->fpA); dump_list(f, "fpT", p->fpT);
			dump_contigs(f, "realign", p->C[p->idxv]);
		}
		const long lim = (p->sg.size() > 1000000 && p->L >= 68) ? 10000 : 1000;
		++pass;
		if (cr - pre < lim) break;
		pre = cr;
	}
	return mcomh_update_single(p);
}

extern "C" int mcomh_dump_stages(mcomh_pipeline *p, const char *path)
{
	if (!p || !path) return MCOM_E_ARG;
	if (p->h_ascii.empty() && p->n) return p->fail(MCOM_E_ARG, "dump needs the reads on the host");
	FILE *f = fopen(path, "w");
	if (!f) return p->fail(MCOM_E_ARG, "cannot write %s", path);
	int rc = mcomh_kt_for_reads(p);
	if (rc) { fclose(f); return rc; }
	const int L = p->L;
	fprintf(f, "PARAMS L %d k %d b %d rw %d e %d cbthr %d m %d n %zu\n", L, p->k, NB_BITS, p->rw, p->e, p->cbthr, p->m, p->n);
	fprintf(f, "STAGE reads\nREADS %zu\n", p->n);
	std::string line((size_t)L, 'A');
	size_t nn = 0;
	for (size_t r = 0; r < p->n; ++r) {
		const uint8_t *src = p->h_ascii.data() + r * (size_t)L;
		if (p->h_cls[r] == MCOM_CLS_SKETCH) for (int i = 0; i < L; ++i) line[i] = ACGT[p->base((uint32_t)r, i)];
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
	if ((rc = mcomh_kt_for_bucket(p))) { fclose(f); return rc; }
	fprintf(f, "STAGE bucket\n");
	dump_contigs(f, "bucket", p->C[0]);
	dump_list(f, "sg", p->sg);
	const std::vector<Contig> stage1 = p->C[0];
	(void)stage1;
	// the first-m minimizers the reference pushed into mi[0] while building the contigs (:458-474)
	{
		DevContigs dc; DevBuf<uint32_t> mo; DevBuf<mcom_mm128> mr; uint64_t tm = 0;
		std::vector<mcom_mm128> rec;
		if (!p->C[0].empty()) {
			if ((rc = upload_contigs(p, p->C[0], dc, false)) || (rc = sketch_contigs(p, dc, (uint32_t)p->m, mo, mr, tm))) { fclose(f); return rc; }
			rec.resize(tm);
			if (tm) (void)hipMemcpy(rec.data(), mr.p, tm * sizeof(mcom_mm128), hipMemcpyDeviceToHost);
		}
		dump_buckets(f, "MI0", rec);
	}
	if ((rc = mcomh_combine_cluster(p))) { fclose(f); return rc; }
	fprintf(f, "STAGE combine\n");
	dump_contigs(f, "combine", p->C[p->idxv]);
	if ((rc = run_stage2(p, f))) { fclose(f); return rc; }
	fprintf(f, "STAGE final\n");
	dump_list(f, "sg", p->sg);
	fprintf(f, "END\n");
	fclose(f);
	return MCOM_OK;
}

// ---- results -----------------------------------------------------------------------------------------------
extern "C" size_t mcomh_n_contigs(const mcomh_pipeline *p) { return p ? p->C[p->idxv].size() : 0; }
extern "C" const char *mcomh_contig_ref(const mcomh_pipeline *p, size_t i) { return p->C[p->idxv][i].ref.c_str(); }
extern "C" size_t mcomh_contig_n(const mcomh_pipeline *p, size_t i) { return p->C[p->idxv][i].a.size(); }
extern "C" const uint64_t *mcomh_contig_members(const mcomh_pipeline *p, size_t i) { return p->C[p->idxv][i].a.data(); }
extern "C" const uint32_t *mcomh_list(const mcomh_pipeline *p, const char *name, size_t *n)
{
	const std::vector<uint32_t> *v = nullptr;
	if (!strcmp(name, "allA")) v = &p->allA; else if (!strcmp(name, "allT")) v = &p->allT; else if (!strcmp(name, "allN")) v = &p->allN;
	else if (!strcmp(name, "fpA")) v = &p->fpA; else if (!strcmp(name, "fpT")) v = &p->fpT; else if (!strcmp(name, "fpN")) v = &p->fpN;
	else if (!strcmp(name, "Nfile")) v = &p->Nfile; else if (!strcmp(name, "sg")) v = &p->sg;
	if (!v) { if (n) *n = 0; return nullptr; }
	if (n) *n = v->size();
	return v->data();
}
extern "C" int mcomh_prof_enable(mcomh_pipeline *p, int on) { return p ? mcom_prof_enable(p->ctx, on) : MCOM_E_ARG; }
extern "C" int mcomh_prof_read(mcomh_pipeline *p, const char *name, double *total_ms, uint64_t *launches)
{
	return p ? p->gpu(mcom_prof_read(p->ctx, name, total_ms, launches)) : MCOM_E_ARG;
}
extern "C" double mcomh_stat(const mcomh_pipeline *p, const char *name)
{
	if (!p) return 0;
	if (!strcmp(name, "k")) return p->k;
	if (!strcmp(name, "rw")) return p->rw;
	if (!strcmp(name, "maxsearch")) return p->maxsearch;
	auto it = p->stat.find(name);
	return it == p->stat.end() ? 0.0 : it->second;
}
