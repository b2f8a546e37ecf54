// minicom_amd/host/mcom_fastq.cpp -- FASTQ/FASTA ingest for the MI355X-native minicom hot path.
//
// Replaces bseq_open / bseq_read (reference bseq.c:19-66, kseq.h): the reference strdup()s every read into its own
// heap block and copies them again into reads->seq; here the sequence lines are parsed straight into one of two pinned
// host chunks, and while the parser fills one chunk the other one travels to HBM (hipMemcpyAsync on a copy stream), so
// that the reads are resident on the device as an [n][L] character matrix when the file has been read -- the input
// mcomh_create takes.  Plain or gzip-compressed files (zlib), FASTA or FASTQ, sequences over several lines.
// As in the reference every read must have the same length (bseq.c:54-57): anything else is an error, not exit(1).
#include "../../include/mcom.h"
#include "../../include/mcom_host.h"
#include "mcom_fastq.hpp"
#include <hip/hip_runtime_api.h>
#include <zlib.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sched.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

// mcom_fastq_gz.cpp: a gzip file of several members, inflated and parsed by all cores
int mcom_fastq_gz_members_to_device(const char *path, int device, int *L_io, uint8_t **d_reads, size_t *n_out, size_t *n_members_out, std::string *err);
int mcom_fastq_gz_members(const char *path, int device, int *L_io, uint8_t **d_reads, uint8_t *host_out, size_t host_cap, size_t *n_out, size_t *n_members_out, std::string *err);

namespace {

// kseq-like record reader over a gz stream: yields one sequence at a time, appended to `out`
class SeqReader {
	gzFile f_;
	std::vector<unsigned char> buf_;
	size_t pos_ = 0, end_ = 0;
	bool eof_ = false, io_error_ = false;
	int last_ = 0;                         // the '>' or '@' that starts the next record, once seen
	bool fill() {
		if (eof_) return false;
		const int n = gzread(f_, buf_.data(), (unsigned)buf_.size());
		if (n <= 0) {
			// a clean end of file has n == 0 and gzeof() set; anything else is a read error or a truncated / corrupt gzip stream
			int zerr = 0; (void)gzerror(f_, &zerr);
			if (n < 0 || !gzeof(f_) || (zerr != Z_OK && zerr != Z_STREAM_END)) io_error_ = true;
			eof_ = true; pos_ = end_ = 0; return false;
		}
		pos_ = 0; end_ = (size_t)n; return true;
	}
	int getc() { if (pos_ >= end_ && !fill()) return -1; return buf_[pos_++]; }
	// appends the rest of the current line (without the newline) to out (if non-null); returns false at end of file
	bool line(std::string *out) {
		bool any = false;
		for (;;) {
			if (pos_ >= end_ && !fill()) return any;
			any = true;
			const unsigned char *b = buf_.data() + pos_;
			const unsigned char *nl = (const unsigned char*)memchr(b, '\n', end_ - pos_);
			const size_t len = nl ? (size_t)(nl - b) : end_ - pos_;
			if (out) out->append((const char*)b, len);
			pos_ += len;
			if (nl) { ++pos_; if (out && !out->empty() && out->back() == '\r') out->pop_back(); return true; }
		}
	}
public:
	explicit SeqReader(gzFile f) : f_(f), buf_((size_t)4 << 20) {}
	bool io_error() const { return io_error_; }
	// 1 = a record was read into seq, 0 = end of file, -1 = malformed (quality string shorter than the sequence, read error,
	// truncated gzip stream), -2 = a sequence character outside ACGTN (the archive could not give it back: the packed rows
	// hold two bits per base plus an N mask, so lower-case and IUPAC codes are refused instead of silently changed)
	int next(std::string &seq) {
		const int st = next_record(seq);
		if (io_error_) return -1;
		if (st == 1) for (const char ch : seq) if (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T' && ch != 'N') return -2;
		return st;
	}
private:
	int next_record(std::string &seq) {
		seq.clear();
		int c = last_;
		if (!c) { while ((c = getc()) != -1 && c != '>' && c != '@') {} if (c == -1) return 0; }
		last_ = 0;
		line(nullptr);                                                     // name and comment
		while ((c = getc()) != -1 && c != '>' && c != '+' && c != '@') {   // sequence lines
			if (c == '\n') continue;
			seq.push_back((char)c);
			line(&seq);
		}
		if (c == '>' || c == '@') last_ = c;
		if (c != '+') return 1;                                            // FASTA record, or the last one
		line(nullptr);                                                     // the '+' line
		size_t q = 0;
		std::string ql;
		while (q < seq.size()) { ql.clear(); if (!line(&ql)) return -1; q += ql.size(); }
		return q == seq.size() ? 1 : -1;
	}
};

struct Chunk { unsigned char *p = nullptr; hipEvent_t done = nullptr; bool busy = false; };

}  // namespace

// Host-only form (tests, tools): the reads of a file into a caller buffer of cap_reads rows of L characters.
// *L == 0 on entry: taken from the first read.  Returns 0, MCOM_E_ARG for a malformed file or a read of another
// length, MCOM_E_OVERFLOW when the file holds more than cap_reads reads (*n = that many were stored).
extern "C" int mcomh_fastq_read(const char *path, int *L, uint8_t *out, size_t cap_reads, size_t *n)
{
	if (!path || !L || !n) return MCOM_E_ARG;
	*n = 0;
	if (out) {                                                                // a gzip file of several members: all cores (mcom_fastq_gz.cpp); anything it does not take falls through
		int L2 = *L; size_t n2 = 0, members = 0; std::string msg;
		const int gz = mcom_fastq_gz_members(path, -1, &L2, nullptr, out, cap_reads, &n2, &members, &msg);
		if (gz == 1) { *L = L2; *n = n2; return MCOM_OK; }
	}
	gzFile f = gzopen(path, "rb");
	if (!f) return MCOM_E_ARG;
	gzbuffer(f, 1 << 20);
	SeqReader rd(f);
	std::string seq;
	int rc = MCOM_OK, st;
	while ((st = rd.next(seq)) == 1) {
		if (*L == 0) *L = (int)seq.size();
		if ((int)seq.size() != *L || *L < 1 || *L > 256) { rc = MCOM_E_ARG; break; }  // bseq.c:54-57 exits here
		if (*n >= cap_reads) { rc = MCOM_E_OVERFLOW; break; }
		if (out) memcpy(out + *n * (size_t)*L, seq.data(), (size_t)*L);
		++*n;
	}
	if (st < 0) rc = MCOM_E_ARG;
	gzclose(f);
	return rc;
}

static int fastq_to_device_mapped(const char *path, int device, int *L, uint8_t **d_reads, size_t *n);

// The reads of a file to HBM: *d_reads = [n][L] characters (hipMalloc'ed, the caller hipFree()s it), ready for
// mcomh_create(..., d_reads, pitch = L, ...).  Two pinned chunks of chunk_reads rows alternate between the parser
// and the copy engine.
extern "C" int mcomh_fastq_to_device(const char *path, int device, int *L, size_t chunk_reads, uint8_t **d_reads, size_t *n, char *err, size_t err_cap)
{
	auto fail = [&](int code, const char *msg) { if (err && err_cap) snprintf(err, err_cap, "%s", msg); return code; };
	if (!path || !L || !d_reads || !n) return MCOM_E_ARG;
	*d_reads = nullptr; *n = 0;
	if (hipSetDevice(device) != hipSuccess) return fail(MCOM_E_HIP, "no usable GPU");
	{                                                                         // a plain four-line FASTQ file: parsed by all cores at once
		const int fast = fastq_to_device_mapped(path, device, L, d_reads, n);
		if (fast == 1) return MCOM_OK;
		if (fast < 0) return fail(fast, "upload failed");
	}
	{                                                                         // a gzip file of several members (bgzip, concatenated gzip): inflated and parsed by all cores (mcom_fastq_gz.cpp)
		std::string msg; size_t members = 0;
		const int gz = mcom_fastq_gz_members_to_device(path, device, L, d_reads, n, &members, &msg);
		if (gz == 1) return MCOM_OK;
		if (gz < 0) return fail(gz, msg.empty() ? "upload failed" : msg.c_str());
	}
	gzFile f = gzopen(path, "rb");
	if (!f) return fail(MCOM_E_ARG, "cannot open the input file");
	gzbuffer(f, 1 << 20);
	SeqReader rd(f);
	std::string seq;
	if (chunk_reads == 0) chunk_reads = (size_t)1 << 20;
	hipStream_t cs = nullptr;
	Chunk ch[2];
	uint8_t *dev = nullptr; size_t dev_cap = 0;                               // rows
	int rc = MCOM_OK, st = 0, cur = 0;
	size_t in_chunk = 0, total = 0;
	auto cleanup = [&]() {
		if (cs) { (void)hipStreamSynchronize(cs); (void)hipStreamDestroy(cs); }
		for (Chunk &c : ch) { if (c.p) (void)hipHostFree(c.p); if (c.done) (void)hipEventDestroy(c.done); }
		gzclose(f);
	};
	auto flush = [&](bool last) -> int {                                     // chunk `cur` -> device, asynchronously
		if (!in_chunk) return MCOM_OK;
		if (total + in_chunk > dev_cap) {                                     // grow, keeping what has arrived
			size_t want = dev_cap ? dev_cap * 2 : (last ? in_chunk : chunk_reads * 8);
			while (want < total + in_chunk) want *= 2;
			uint8_t *nd = nullptr;
			if (hipMalloc(&nd, want * (size_t)*L + 16) != hipSuccess) return MCOM_E_NOMEM;
			if (hipStreamSynchronize(cs) != hipSuccess) return MCOM_E_HIP;
			if (dev && total && hipMemcpy(nd, dev, total * (size_t)*L, hipMemcpyDeviceToDevice) != hipSuccess) return MCOM_E_HIP;
			if (dev) (void)hipFree(dev);
			dev = nd; dev_cap = want;
		}
		if (hipMemcpyAsync(dev + total * (size_t)*L, ch[cur].p, in_chunk * (size_t)*L, hipMemcpyHostToDevice, cs) != hipSuccess) return MCOM_E_HIP;
		if (hipEventRecord(ch[cur].done, cs) != hipSuccess) return MCOM_E_HIP;
		ch[cur].busy = true;
		total += in_chunk; in_chunk = 0; cur ^= 1;
		if (ch[cur].busy) { if (hipEventSynchronize(ch[cur].done) != hipSuccess) return MCOM_E_HIP; ch[cur].busy = false; }   // the other chunk must have left
		return MCOM_OK;
	};
	while ((st = rd.next(seq)) == 1) {
		if (*L == 0) *L = (int)seq.size();
		if ((int)seq.size() != *L || *L < 1 || *L > 256) { rc = fail(MCOM_E_ARG, "Length of reads are different. The program can not compress it."); break; }   // bseq.c:54-57
		if (!cs) {
			if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) { rc = fail(MCOM_E_HIP, "stream"); break; }
			for (Chunk &c : ch)
				if (hipHostMalloc((void**)&c.p, chunk_reads * (size_t)*L, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&c.done, hipEventDisableTiming) != hipSuccess) { rc = fail(MCOM_E_NOMEM, "pinned chunks"); break; }
			if (rc) break;
		}
		memcpy(ch[cur].p + in_chunk * (size_t)*L, seq.data(), (size_t)*L);
		if (++in_chunk == chunk_reads && (rc = flush(false))) { fail(rc, "upload failed"); break; }
	}
	if (!rc && st == -2) rc = fail(MCOM_E_ARG, "a sequence holds a character outside ACGTN (lower-case and IUPAC codes are not representable)");
	if (!rc && st < 0) rc = fail(MCOM_E_ARG, rd.io_error() ? "read error or truncated / corrupt gzip stream" : "malformed record (quality string shorter than the sequence)");
	if (!rc && (rc = flush(true))) fail(rc, "upload failed");
	if (!rc && cs && hipStreamSynchronize(cs) != hipSuccess) rc = fail(MCOM_E_HIP, "upload failed");
	cleanup();
	if (rc) { if (dev) (void)hipFree(dev); return rc; }
	*d_reads = dev; *n = total;
	return MCOM_OK;
}

extern "C" void mcomh_device_free(void *d_ptr) { if (d_ptr) (void)hipFree(d_ptr); }

// ---- the common case in parallel: a plain FASTQ file of four-line records ------------------------------------------------
// One parser thread reads ~0.4 GB/s of FASTQ; a 100 M-read file is 32 GB.  The file is mapped, cut into one piece per thread at
// record boundaries, and every thread validates and counts its records (pass 1), then -- the row of its first record being the
// sum of the counts before it -- copies its sequence lines into two page-locked blocks of its own that alternate between the
// thread and the copy engine (pass 2), straight to the rows' place in HBM.  A record boundary inside a piece is found by its
// shape: a line that starts with '@', followed by a line of L bases, a line that starts with '+' and a line of L characters (a
// quality line may start with '@' too, but then the line two further on is a sequence, not a '+' line).  The one-pass form below also
// takes FASTA, with sequences over one or more lines (fasta_record_at).  Anything else -- gzip of one member, FASTQ with sequences over
// several lines, carriage returns, a read of another length -- is left to the sequential reader
// above, which also words the error messages.  Returns 1 = done, 0 = not this layout (nothing touched), < 0 = error.
namespace {
struct Piece { size_t begin = 0, end = 0, records = 0, first_row = 0; bool ok = true; };
inline const char *next_line(const char *p, const char *e) { const char *nl = (const char*)memchr(p, '\n', (size_t)(e - p)); return nl ? nl + 1 : e; }
// a record of the expected shape at p: its end, or nullptr
inline const char *record_at(const char *p, const char *e, int L)
{
	if (p >= e || *p != '@') return nullptr;
	const char *s = next_line(p, e);
	if (s + L >= e || s[L] != '\n') return nullptr;
	const char *plus = s + L + 1;
	if (plus >= e || *plus != '+') return nullptr;
	const char *q = next_line(plus, e);
	if (q + L > e) return nullptr;
	if (q + L == e) return e;                                                  // the last line of the file may lack its newline
	return q[L] == '\n' ? q + L + 1 : nullptr;
}
// FASTA (round 5): a record of the expected shape at p -- a line that starts with '>', then the sequence over ONE OR MORE lines, L characters in
// all, up to a line that starts with '>' or the end of the file (kseq.h reads a sequence line by line until a line starts with '>', '+' or '@';
// a '+' or '@' line, a blank line or a sequence of another length is "not this shape": the sequential reader takes the file and words the
// message).  Returns the record's end, or nullptr; *more = the text ends before the record's end can be seen and it is not the file's end.
inline const char *fasta_record_at(const char *p, const char *e, int L, bool at_eof, bool *more)
{
	*more = false;
	if (p >= e || *p != '>') return nullptr;
	const char *s = (const char*)memchr(p, '\n', (size_t)(e - p));
	if (!s) { *more = !at_eof; return nullptr; }
	++s;
	int got = 0;
	for (;;) {
		if (s >= e) { if (!at_eof) { *more = true; return nullptr; } return got == L ? e : nullptr; }
		if (*s == '>') return got == L ? s : nullptr;
		if (*s == '+' || *s == '@' || *s == '\n') return nullptr;
		const char *nl = (const char*)memchr(s, '\n', (size_t)(e - s));
		const size_t ll = (size_t)((nl ? nl : e) - s);
		if (got + ll > (size_t)L) return nullptr;
		got += (int)ll;
		if (!nl) { if (!at_eof) { *more = true; return nullptr; } return got == L ? e : nullptr; }
		s = nl + 1;
	}
}
}
// ---- pass 1: the file mapped and cut into pieces at record boundaries, every piece validated and counted ---------------------
struct McomFastqIndex {
	const char *base = nullptr; size_t size = 0; int L = 0; std::vector<Piece> pc; size_t total = 0;
	~McomFastqIndex() { if (base) munmap((void*)base, size); }
};
// the CPUs this process may really use: the affinity mask, cut by the cgroup's CPU quota when there is one (a container on a 256-core
// host with "cpu.max = 1600000 100000" runs 16 threads at a time however many it starts)
static size_t usable_cpus()
{
	size_t n = std::max(1u, std::thread::hardware_concurrency());
	cpu_set_t set;
	if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::max<size_t>(1, (size_t)CPU_COUNT(&set));
	if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                          // cgroup v2
		char quota[32] = ""; long period = 0;
		if (fscanf(f, "%31s %ld", quota, &period) == 2 && period > 0 && strcmp(quota, "max") != 0) { const long q = atol(quota); if (q > 0) n = std::min<size_t>(n, (size_t)((q + period - 1) / period)); }
		fclose(f);
	} else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {      // cgroup v1
		long q = -1, period = 0;
		if (fscanf(g, "%ld", &q) != 1) q = -1;
		fclose(g);
		if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%ld", &period) != 1) period = 0; fclose(h); }
		if (q > 0 && period > 0) n = std::min<size_t>(n, (size_t)((q + period - 1) / period));
	}
	return n;
}
size_t mcom_usable_cpus() { return usable_cpus(); }                            // (mcom_fastq_gz.cpp)
static int parser_threads(size_t size)
{
	// round 4: as many parser threads as the process has CPUs for, up to 64 (round 3 stopped at 16 whatever the host)
	size_t nt = std::min<size_t>(64, usable_cpus());
	nt = std::min(nt, std::max<size_t>(1, size >> 24));                          // ... but not less than 16 MB of file per thread
	return (int)nt;
}
// 1 = indexed (*out), 0 = not this layout (nothing kept), < 0 = error
int mcom_fastq_index(const char *path, int want_L, McomFastqIndex **out)
{
	*out = nullptr;
	const int fd = open(path, O_RDONLY);
	if (fd < 0) return 0;
	struct stat st;
	if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 64) { close(fd); return 0; }
	const size_t size = (size_t)st.st_size;
	const char *base = (const char*)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
	close(fd);
	if (base == MAP_FAILED) return 0;
	std::unique_ptr<McomFastqIndex> ix(new McomFastqIndex());
	ix->base = base; ix->size = size;
	(void)madvise((void*)base, size, MADV_SEQUENTIAL);
	const char *end = base + size;
	if ((unsigned char)base[0] == 0x1f && (unsigned char)base[1] == 0x8b) return 0;       // gzip
	if (base[0] != '@') return 0;
	// the first record gives the read length
	const char *s0 = next_line(base, end);
	const char *s1 = (const char*)memchr(s0, '\n', (size_t)(end - s0));
	if (!s1) return 0;
	const int len = (int)(s1 - s0);
	if (len < 1 || len > 256 || (want_L && want_L != len) || !record_at(base, end, len)) return 0;
	const int nt = parser_threads(size);
	std::vector<Piece> &pc = ix->pc;
	pc.assign((size_t)nt, Piece());
	// every piece from the first record boundary at or behind its nominal start to the first at or behind its nominal end
	auto find_start = [&](size_t at) -> size_t {
		if (at == 0) return 0;
		const char *p = next_line(base + at - 1, end);                            // (a boundary exactly at `at` counts)
		for (int tries = 0; tries < 8 && p < end; ++tries) { if (record_at(p, end, len)) return (size_t)(p - base); p = next_line(p, end); }
		return p >= end ? size : (size_t)-1;
	};
	std::atomic<int> bad{0};
	{
		std::vector<std::thread> th;
		for (int t = 0; t < nt; ++t) th.emplace_back([&, t]() {
			Piece &P = pc[(size_t)t];
			P.begin = find_start(size * (size_t)t / (size_t)nt);
			const size_t stop = size * (size_t)(t + 1) / (size_t)nt;
			if (P.begin == (size_t)-1) { bad = 1; return; }
			const char *p = base + P.begin;
			size_t cnt = 0;
			while (p < end && (size_t)(p - base) < stop) {
				const char *q = record_at(p, end, len);
				if (!q) { bad = 1; return; }
				++cnt; p = q;
			}
			P.end = (size_t)(p - base); P.records = cnt;
		});
		for (auto &x : th) x.join();
	}
	if (bad) return 0;
	size_t total = 0;
	for (int t = 0; t < nt; ++t) {
		if (t + 1 < nt && pc[(size_t)t].end != pc[(size_t)t + 1].begin) return 0;   // the pieces must tile the file: every byte belongs to a record of the shape
		pc[(size_t)t].first_row = total; total += pc[(size_t)t].records;
	}
	if (pc[(size_t)nt - 1].end != size || total == 0) return 0;
	ix->L = len; ix->total = total;
	*out = ix.release();
	return 1;
}
void mcom_fastq_index_free(McomFastqIndex *ix) { delete ix; }
size_t mcom_fastq_index_reads(const McomFastqIndex *ix) { return ix ? ix->total : 0; }
int mcom_fastq_index_len(const McomFastqIndex *ix) { return ix ? ix->L : 0; }

// ---- pass 2: the sequence lines to HBM, every parser thread through two page-locked blocks of its own that alternate between the
// thread and the copy engine.  PACKED = false: the characters, [n][L] (what mcomh_create takes).  PACKED = true (round 4): the thread
// packs -- 2 bits per base into W words, one N flag per base into NW words, the row formats of the device -- and 8 (W + NW) bytes per
// read cross PCIe instead of L (64 instead of 150 at L = 150); classes, counts and the majority-base substitution stay on the device
// (mcom_process_reads_packed).  0 = done, MCOM_E_ARG = a character outside ACGTN, other < 0 = error.
namespace {
struct PackLut { uint8_t v[256]; PackLut() { memset(v, 0xFF, sizeof v); v[(unsigned char)'A'] = 0; v[(unsigned char)'C'] = 1; v[(unsigned char)'G'] = 2; v[(unsigned char)'T'] = 3; v[(unsigned char)'N'] = 4; } };
const PackLut g_pack_lut;
}
// where pass 2 spends its time (the slowest thread of the last call): setting up stream and page-locked blocks, parsing / packing,
// waiting for a block to leave
double g_fastq_ms[3] = {0, 0, 0};
static inline double fq_now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int fastq_upload(const McomFastqIndex *ix, int device, void *copy_stream, uint8_t *d_ascii, size_t row0)
{
	const int len = ix->L;
	const char *base = ix->base, *end = ix->base + ix->size;
	const size_t row_bytes = (size_t)len;
	const size_t CH = std::max<size_t>(1024, ((size_t)1 << 20) / row_bytes);      // rows per staging block: about 1 MB (one stream moves 31 GB/s in such blocks, 15 in blocks of 256 KB)
	const size_t nt = ix->pc.size();
	// ONE page-locked arena for all parser threads, two blocks each, and ONE uploader thread that sends what the parsers have filled on
	// ONE stream -- the caller's, when it has one.  Measured (tools/ubench/h2d_setup.cpp): the first streams of a process cost 26 ms EACH
	// to create, and 64 threads copying on 64 streams move 3.5 GB/s where one thread on one stream moves 31.  (The first form of this pass
	// let every parser lock two blocks of its own and create a stream and two events: 0.18 s of a 0.25 s pass at 64 threads.)
	const size_t blk_bytes = (CH * row_bytes + 255) & ~(size_t)255;
	unsigned char *arena = nullptr;
	if (hipHostMalloc((void**)&arena, 2 * nt * blk_bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return MCOM_E_NOMEM; }
	struct Free { unsigned char *p; ~Free() { (void)hipHostFree(p); } } free_arena{arena};
	struct Job { const unsigned char *src; size_t row, take; std::atomic<int> *full; };
	std::mutex mu; std::condition_variable cv; std::deque<Job> jobs; bool closing = false;
	std::unique_ptr<std::atomic<int>[]> full(new std::atomic<int>[2 * nt]);
	for (size_t q = 0; q < 2 * nt; ++q) full[q] = 0;
	std::atomic<int> err{0};
	const int n_up = 1;
	std::vector<std::thread> up;
	for (int u = 0; u < n_up; ++u) up.emplace_back([&]() {
		hipStream_t cs = (hipStream_t)copy_stream;
		const bool own = cs == nullptr;
		if (hipSetDevice(device) != hipSuccess || (own && hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess)) { err = MCOM_E_HIP; cs = nullptr; }
		std::vector<Job> batch;
		for (;;) {
			batch.clear();
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&]() { return closing || !jobs.empty(); });
				while (!jobs.empty() && batch.size() < 16) { batch.push_back(jobs.front()); jobs.pop_front(); }
				if (batch.empty() && closing) break;
			}
			bool ok = cs != nullptr;
			for (const Job &j : batch) {
				if (!ok) break;
				ok = hipMemcpyAsync(d_ascii + j.row * (size_t)len, j.src, j.take * (size_t)len, hipMemcpyHostToDevice, cs) == hipSuccess;
			}
			if (ok) ok = hipStreamSynchronize(cs) == hipSuccess;
			if (!ok && !err) err = MCOM_E_HIP;
			for (const Job &j : batch) j.full->store(0, std::memory_order_release);       // (on an error too: nobody must wait for ever)
		}
		if (cs && own) (void)hipStreamDestroy(cs);
	});
	std::vector<std::thread> th;
	std::vector<double> tm(3 * nt, 0.0);
	for (size_t t = 0; t < nt; ++t) th.emplace_back([&, t]() {
		const Piece &P = ix->pc[t];
		if (!P.records) return;
		const char *p = base + P.begin;
		size_t row = row0 + P.first_row, left = P.records;
		unsigned bad_char = 0;
		int cur = 0;
		while (left && !err) {
			const size_t take = std::min(CH, left);
			const double t_w = fq_now();
			while (full[2 * t + cur].load(std::memory_order_acquire)) std::this_thread::yield();   // the block has not left yet
			const double t_p = fq_now();
			tm[3 * t + 2] += t_p - t_w;
			unsigned char *blk = arena + (2 * t + cur) * blk_bytes;
			unsigned char *o = blk;
			for (size_t r = 0; r < take; ++r) {
				const char *s = next_line(p, end);
				memcpy(o, s, (size_t)len);
				for (int i = 0; i < len; ++i) bad_char |= g_pack_lut.v[o[i]] >> 3;
				o += len;
				const char *q = next_line(s + len + 1, end);
				p = q + len + 1 <= end ? q + len + 1 : end;
			}
			if (bad_char) { err = MCOM_E_ARG; break; }
			tm[3 * t + 1] += fq_now() - t_p;
			full[2 * t + cur].store(1, std::memory_order_release);
			{ std::lock_guard<std::mutex> lk(mu); jobs.push_back(Job{blk, row, take, &full[2 * t + cur]}); }
			cv.notify_one();
			cur ^= 1; row += take; left -= take;
		}
	});
	for (auto &x : th) x.join();
	{ std::lock_guard<std::mutex> lk(mu); closing = true; }
	cv.notify_all();
	for (auto &x : up) x.join();
	for (int q = 0; q < 3; ++q) { g_fastq_ms[q] = 0; for (size_t t = 0; t < nt; ++t) g_fastq_ms[q] = std::max(g_fastq_ms[q], tm[3 * t + q]); }
	return (int)err;
}
// ---- ONE pass (round 4, the packed route): read, check, pack and send in a single sweep over the file ---------------------------
// The two-pass form above maps the file and walks it twice; at 16 usable CPUs the first walk alone took 0.17 s for 6.3 GB -- most of it
// page faults (1.5 M of them) -- before a single row moved.  Here every parser thread pread()s its piece in chunks of 4 MB into a
// buffer of its own (no mapping, no faults), checks the shape of every record as it goes and packs it.  Where a row goes cannot be
// known before all the pieces in front have been counted, so the rows are sent to PROVISIONAL places -- piece t starts at row
// floor(begin_t / smallest possible record) + t, an upper bound of the rows in front of it -- and the caller closes the gaps with one
// device-to-device copy per piece once the counts are in (mcom_fastq_stream_pieces).
struct McomFastqStream {
	int L = 0; size_t total = 0;
	std::vector<size_t> prov, count;                                             // provisional first row and rows of every piece
};
size_t mcom_fastq_stream_total(const McomFastqStream *st) { return st ? st->total : 0; }
int mcom_fastq_stream_len(const McomFastqStream *st) { return st ? st->L : 0; }
size_t mcom_fastq_stream_pieces(const McomFastqStream *st, size_t *prov, size_t *count)
{
	if (!st) return 0;
	if (prov && count) for (size_t q = 0; q < st->prov.size(); ++q) { prov[q] = st->prov[q]; count[q] = st->count[q]; }
	return st->prov.size();
}
void mcom_fastq_stream_free(McomFastqStream *st) { delete st; }
// rows a file of this size can hold at most (+ one per parser thread), and its read length; 0 = not a plain four-line FASTQ file
size_t mcom_fastq_stream_cap(const char *path, int *L)
{
	const int fd = open(path, O_RDONLY);
	if (fd < 0) return 0;
	struct stat stt;
	char head[4096];
	size_t cap = 0;
	if (fstat(fd, &stt) == 0 && S_ISREG(stt.st_mode) && stt.st_size >= 64) {
		const ssize_t got = pread(fd, head, sizeof head, 0);
		if (got > 8 && head[0] == '>') {
			// FASTA: the first record's sequence lines up to the next '>' line (or the end of a file this short) give the read length
			const char *e = head + got;
			const bool whole = (size_t)got == (size_t)stt.st_size;
			const char *s0 = next_line(head, e);
			int len = 0; bool closed = false;
			for (const char *c = s0; c < e; ) {
				if (*c == '>') { closed = true; break; }
				const char *nl = (const char*)memchr(c, '\n', (size_t)(e - c));
				if (!nl) { if (whole) len += (int)(e - c); else len = -1000000; break; }
				len += (int)(nl - c); c = nl + 1;
				if (c >= e && whole) closed = true;
			}
			if (whole) closed = true;
			bool more = false;
			if (closed && len >= 1 && len <= 256 && (!*L || *L == len) && fasta_record_at(head, e, len, whole, &more) != nullptr) {
				*L = len;
				cap = (size_t)stt.st_size / (size_t)(len + 3) + 64 + 1;
			}
		} else if (got > 8 && head[0] == '@' && !((unsigned char)head[0] == 0x1f && (unsigned char)head[1] == 0x8b)) {
			const char *e = head + got;
			const char *s0 = next_line(head, e);
			const char *s1 = (const char*)memchr(s0, '\n', (size_t)(e - s0));
			if (s1) {
				const int len = (int)(s1 - s0);
				if (len >= 1 && len <= 256 && (!*L || *L == len) && ((size_t)got == (size_t)stt.st_size ? record_at(head, e, len) != nullptr : true)) {
					*L = len;
					cap = (size_t)stt.st_size / (size_t)(2 * len + 6) + 64 + 1;
				}
			}
		}
	}
	close(fd);
	return cap;
}
// 1 = every read is on the device (*out says where), 0 = not this layout after all (the caller takes the sequential reader), < 0 = error
int mcom_fastq_stream_packed(const char *path, int L, int device, void *copy_stream, uint64_t *d_packed, uint64_t *d_nmask, size_t cap_rows, McomFastqStream **out)
{
	*out = nullptr;
	const int fd = open(path, O_RDONLY);
	if (fd < 0) return 0;
	struct stat stt;
	if (fstat(fd, &stt) != 0 || !S_ISREG(stt.st_mode) || stt.st_size < 64) { close(fd); return 0; }
	struct CloseFd { int f; ~CloseFd() { close(f); } } close_fd{fd};
	const size_t size = (size_t)stt.st_size;
	const int len = L, W = (2 * len + 63) / 64, NW = (len + 63) / 64;
	char first_byte = 0;
	if (pread(fd, &first_byte, 1, 0) != 1) return 0;
	const bool fasta = first_byte == '>';                                         // (mcom_fastq_stream_cap has looked at the first record)
	const size_t minrec = fasta ? (size_t)(len + 3) : (size_t)(2 * len + 6);
	const size_t nt = (size_t)parser_threads(size);
	const size_t row_bytes = (size_t)8 * (W + NW);
	const size_t CH = std::max<size_t>(1024, ((size_t)1 << 20) / row_bytes);      // rows per staging block: about 1 MB
	const size_t blk_bytes = (CH * row_bytes + 255) & ~(size_t)255;
	const size_t IO = (size_t)4 << 20, SLACK = 4096;                              // bytes per pread; a record is at most 2 L + a name line
	unsigned char *arena = nullptr;
	if (hipHostMalloc((void**)&arena, 2 * nt * blk_bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return MCOM_E_NOMEM; }
	struct Free { unsigned char *p; ~Free() { (void)hipHostFree(p); } } free_arena{arena};
	struct Job { const unsigned char *src; size_t row, take; std::atomic<int> *full; };
	std::mutex mu; std::condition_variable cv; std::deque<Job> jobs; bool closing = false;
	std::unique_ptr<std::atomic<int>[]> full(new std::atomic<int>[2 * nt]);
	for (size_t q = 0; q < 2 * nt; ++q) full[q] = 0;
	std::atomic<int> err{0}, shape{0};
	std::thread up([&]() {
		hipStream_t cs = (hipStream_t)copy_stream;
		const bool own = cs == nullptr;
		if (hipSetDevice(device) != hipSuccess || (own && hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess)) { err = MCOM_E_HIP; cs = nullptr; }
		std::vector<Job> batch;
		for (;;) {
			batch.clear();
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&]() { return closing || !jobs.empty(); });
				while (!jobs.empty() && batch.size() < 16) { batch.push_back(jobs.front()); jobs.pop_front(); }
				if (batch.empty() && closing) break;
			}
			bool ok = cs != nullptr;
			for (const Job &j : batch) {
				if (!ok) break;
				ok = hipMemcpyAsync(d_packed + j.row * (size_t)W, j.src, j.take * (size_t)W * 8, hipMemcpyHostToDevice, cs) == hipSuccess &&
				     hipMemcpyAsync(d_nmask + j.row * (size_t)NW, j.src + j.take * (size_t)W * 8, j.take * (size_t)NW * 8, hipMemcpyHostToDevice, cs) == hipSuccess;
			}
			if (ok) ok = hipStreamSynchronize(cs) == hipSuccess;
			if (!ok && !err) err = MCOM_E_HIP;
			for (const Job &j : batch) j.full->store(0, std::memory_order_release);       // (on an error too: nobody must wait for ever)
		}
		if (cs && own) (void)hipStreamDestroy(cs);
	});
	std::vector<Piece> pc(nt);
	std::vector<size_t> prov(nt, 0);
	std::vector<double> tm(3 * nt, 0.0);
	std::vector<std::thread> th;
	for (size_t t = 0; t < nt; ++t) th.emplace_back([&, t]() {
		const size_t nominal = size * t / nt, stop = size * (t + 1) / nt;
		std::vector<char> buf(IO + SLACK + 16);
		size_t off = nominal ? nominal - 1 : 0;                                    // (a boundary exactly at the nominal start counts: look from the byte before)
		size_t have = 0;                                                          // bytes of buf in use; buf[0] is file offset `off`
		bool at_eof = false;
		auto refill = [&](size_t keep_from) -> bool {                             // drop buf[0, keep_from), read more behind what is left
			memmove(buf.data(), buf.data() + keep_from, have - keep_from);
			have -= keep_from; off += keep_from;
			while (have < IO && !at_eof) {
				const ssize_t got = pread(fd, buf.data() + have, IO + SLACK - have, (off_t)(off + have));
				if (got < 0) return false;
				if (got == 0) { at_eof = true; break; }
				have += (size_t)got;
			}
			return true;
		};
		if (!refill(0)) { err = MCOM_E_ARG; return; }
		// the first record boundary at or behind the nominal start
		size_t pos = 0;
		if (nominal) {
			const char *e = buf.data() + have;
			const char *p = next_line(buf.data(), e);
			bool found = false;
			bool more = false;
			for (int tries = 0; tries < (fasta ? len + 8 : 8) && p < e; ++tries) { if (fasta ? fasta_record_at(p, e, len, at_eof, &more) != nullptr : record_at(p, e, len) != nullptr) { found = true; break; } p = next_line(p, e); }
			if (!found) { if (p < e || !at_eof) { shape = 1; return; } pos = have; }   // no record starts here: only the tail of the one before (or nothing at all) is left
			else pos = (size_t)(p - buf.data());
		}
		Piece &P = pc[t];
		P.begin = off + pos;
		prov[t] = P.begin / minrec + t;
		size_t row = prov[t], in_blk = 0, cnt = 0;
		int cur = 0;
		unsigned bad_char = 0;
		uint64_t *ow = nullptr, *on = nullptr;
		auto flush = [&]() {
			if (!in_blk) return;
			// rows of codes and rows of flags lie interleaved per row while the block fills; the uploader needs them as two runs: the flags were
			// written behind CH rows of codes, so they are moved down to close the gap when the block is short
			unsigned char *blk = arena + (2 * t + cur) * blk_bytes;
			if (in_blk < CH) memmove(blk + in_blk * (size_t)W * 8, blk + CH * (size_t)W * 8, in_blk * (size_t)NW * 8);
			full[2 * t + cur].store(1, std::memory_order_release);
			{ std::lock_guard<std::mutex> lk(mu); jobs.push_back(Job{blk, row, in_blk, &full[2 * t + cur]}); }
			cv.notify_one();
			row += in_blk; in_blk = 0; cur ^= 1;
		};
		double t_p = fq_now();
		for (;;) {
			if (err || shape) return;
			if (off + pos >= stop) break;                                           // the record that starts here is the next piece's first
			// a whole record in the buffer?
			const char *e = buf.data() + have;
			const char *p = buf.data() + pos;
			if (p >= e) { if (at_eof) break; if (!refill(pos)) { err = MCOM_E_ARG; return; } pos = 0; continue; }
			if ((size_t)(e - p) < (size_t)(2 * len + SLACK / 2) && !at_eof) { if (!refill(pos)) { err = MCOM_E_ARG; return; } pos = 0; continue; }
			bool more = false;
			const char *q = fasta ? fasta_record_at(p, e, len, at_eof, &more) : record_at(p, e, len);
			if (!q) { shape = 1; return; }                                          // (FASTA with `more`: a name line of kilobytes -- the sequential reader's)
			if (!fasta && q == e && !at_eof) { if (!refill(pos)) { err = MCOM_E_ARG; return; } pos = 0; continue; }   // (the buffer's end is not the file's: the newline may follow)
			if (in_blk == 0) {
				const double t_w = fq_now();
				tm[3 * t + 1] += t_w - t_p;
				while (full[2 * t + cur].load(std::memory_order_acquire)) std::this_thread::yield();   // the block has not left yet
				t_p = fq_now();
				tm[3 * t + 2] += t_p - t_w;
				ow = (uint64_t*)(arena + (2 * t + cur) * blk_bytes); on = ow + CH * (size_t)W;
			}
			const unsigned char *sq = (const unsigned char*)next_line(p, e);
			unsigned char joined[264];
			if (fasta) {                                                            // a sequence over several lines: its L characters in a row first
				const char *c = (const char*)sq;
				const char *nl = (const char*)memchr(c, '\n', (size_t)(q - c));
				if ((nl ? nl : q) - c < len) {
					int got = 0;
					while (got < len) {
						nl = (const char*)memchr(c, '\n', (size_t)(q - c));
						const size_t ll = (size_t)((nl ? nl : q) - c);
						memcpy(joined + got, c, ll); got += (int)ll;
						c = nl ? nl + 1 : q;
					}
					sq = joined;
				}
			}
			for (int w = 0; w < NW; ++w) on[w] = 0;
			for (int w = 0; w < W; ++w) ow[w] = 0;
			for (int i0 = 0; i0 < len; i0 += 8) {                                   // (codes by bit tricks on the ASCII ((c >> 1 ^ c >> 2) & 3: A0 C1 G2 T3, N gives 0), N flags by an exact byte-equals-'N', and the bases spelled back from their codes to refuse anything else)
				uint64_t x;
				if (i0 + 8 <= len) memcpy(&x, sq + i0, 8);
				else { x = 0x4141414141414141ull; memcpy(&x, sq + i0, (size_t)(len - i0)); }
				const uint64_t t2 = ((x >> 1) ^ (x >> 2)) & 0x0303030303030303ull;
				const uint64_t z = x ^ 0x4E4E4E4E4E4E4E4Eull;
				const uint64_t nm = ~(((z & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | z) & 0x8080808080808080ull;
				const uint64_t c2 = (t2 >> 1) & 0x0101010101010101ull, c3 = c2 & t2;
				const uint64_t spelled = 0x4141414141414141ull + 2 * t2 + 2 * c2 + 0x0B * c3;
				bad_char |= (unsigned)((((x ^ spelled) & ~((nm >> 7) * 0xFFull)) != 0) ? 8u : 0u);
				uint64_t g = (t2 | (t2 >> 6)) & 0x000F000F000F000Full;
				g = (g | (g >> 12)) & 0x000000FF000000FFull;
				g = (g | (g >> 24)) & 0xFFFFull;
				const uint64_t n8 = ((nm >> 7) * 0x0102040810204080ull) >> 56;
				ow[i0 >> 5] |= g << (2 * (i0 & 31));
				on[i0 >> 6] |= n8 << (i0 & 63);
			}
			ow += W; on += NW; ++cnt;
			pos = (size_t)(q - buf.data());
			if (++in_blk == CH) { if (bad_char) { err = MCOM_E_ARG; return; } flush(); }
		}
		if (bad_char) { err = MCOM_E_ARG; return; }
		flush();
		tm[3 * t + 1] += fq_now() - t_p;
		P.end = off + pos; P.records = cnt;
		if (row > cap_rows) err = MCOM_E_OVERFLOW;
	});
	for (auto &x : th) x.join();
	{ std::lock_guard<std::mutex> lk(mu); closing = true; }
	cv.notify_all();
	up.join();
	for (int q = 0; q < 3; ++q) { g_fastq_ms[q] = 0; for (size_t t = 0; t < nt; ++t) g_fastq_ms[q] = std::max(g_fastq_ms[q], tm[3 * t + q]); }
	if (err) return err == MCOM_E_ARG ? 0 : (int)err;                              // a character outside ACGTN: the sequential reader words it
	if (shape) return 0;
	size_t total = 0;
	for (size_t t = 0; t < nt; ++t) {
		if (t + 1 < nt && pc[t].end != pc[t + 1].begin) return 0;                  // the pieces must tile the file: every byte belongs to a record of the shape
		total += pc[t].records;
	}
	if (pc[nt - 1].end != size || total == 0) return 0;
	std::unique_ptr<McomFastqStream> st(new McomFastqStream());
	st->L = len; st->total = total; st->prov = prov;
	st->count.resize(nt);
	for (size_t t = 0; t < nt; ++t) st->count[t] = pc[t].records;
	*out = st.release();
	return 1;
}

static int fastq_to_device_mapped(const char *path, int device, int *L, uint8_t **d_reads, size_t *n)
{
	McomFastqIndex *raw = nullptr;
	const int st = mcom_fastq_index(path, *L, &raw);
	if (st != 1) return st;
	std::unique_ptr<McomFastqIndex> ix(raw);
	if (hipSetDevice(device) != hipSuccess) return MCOM_E_HIP;
	uint8_t *dev = nullptr;
	if (hipMalloc(&dev, ix->total * (size_t)ix->L + 16) != hipSuccess) { (void)hipGetLastError(); return MCOM_E_NOMEM; }
	const int err = fastq_upload(ix.get(), device, nullptr, dev, 0);
	if (err) { (void)hipFree(dev); return err == MCOM_E_ARG ? 0 : err; }                 // a character outside ACGTN: the sequential reader says so
	*L = ix->L; *d_reads = dev; *n = ix->total;
	return 1;
}

// paired end: bseq_read + bseq_read_second (preprocess.c:52-75) -- the mates of read i of the first file is read n/2 + i
extern "C" int mcomh_fastq_pair_to_device(const char *path1, const char *path2, int device, int *L, size_t chunk_reads, uint8_t **d_reads, size_t *n,
                                          char *err, size_t err_cap)
{
	if (!d_reads || !n || !L) return MCOM_E_ARG;
	*d_reads = nullptr; *n = 0;
	uint8_t *d1 = nullptr, *d2 = nullptr; size_t n1 = 0, n2 = 0;
	int rc = mcomh_fastq_to_device(path1, device, L, chunk_reads, &d1, &n1, err, err_cap);
	if (!rc) rc = mcomh_fastq_to_device(path2, device, L, chunk_reads, &d2, &n2, err, err_cap);
	if (!rc && n1 != n2) { if (err && err_cap) snprintf(err, err_cap, "the two files hold %zu and %zu reads", n1, n2); rc = MCOM_E_ARG; }
	uint8_t *d = nullptr;
	if (!rc && n1) {
		const size_t bytes = n1 * (size_t)*L;
		if (hipMalloc(&d, 2 * bytes + 16) != hipSuccess) rc = MCOM_E_NOMEM;
		else if (hipMemcpy(d, d1, bytes, hipMemcpyDeviceToDevice) != hipSuccess || hipMemcpy(d + bytes, d2, bytes, hipMemcpyDeviceToDevice) != hipSuccess) rc = MCOM_E_HIP;
	}
	if (d1) (void)hipFree(d1);
	if (d2) (void)hipFree(d2);
	if (rc) { if (d) (void)hipFree(d); return rc; }
	*d_reads = d; *n = 2 * n1;
	return MCOM_OK;
}
