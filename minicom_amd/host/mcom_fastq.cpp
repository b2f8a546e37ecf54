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
#include <hip/hip_runtime_api.h>
#include <zlib.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

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
// quality line may start with '@' too, but then the line two further on is a sequence, not a '+' line).  Anything else --
// gzip, FASTA, sequences over several lines, carriage returns, a read of another length -- is left to the sequential reader
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
}
static int fastq_to_device_mapped(const char *path, int device, int *L, uint8_t **d_reads, size_t *n)
{
	const int fd = open(path, O_RDONLY);
	if (fd < 0) return 0;
	struct stat st;
	if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 64) { close(fd); return 0; }
	const size_t size = (size_t)st.st_size;
	const char *base = (const char*)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
	close(fd);
	if (base == MAP_FAILED) return 0;
	(void)madvise((void*)base, size, MADV_SEQUENTIAL);
	const char *end = base + size;
	struct Unmap { const char *b; size_t s; ~Unmap() { munmap((void*)b, s); } } unmap{base, size};
	if ((unsigned char)base[0] == 0x1f && (unsigned char)base[1] == 0x8b) return 0;       // gzip
	if (base[0] != '@') return 0;
	// the first record gives the read length
	const char *s0 = next_line(base, end);
	const char *s1 = (const char*)memchr(s0, '\n', (size_t)(end - s0));
	if (!s1) return 0;
	const int len = (int)(s1 - s0);
	if (len < 1 || len > 256 || (*L && *L != len) || !record_at(base, end, len)) return 0;
	int nt = (int)std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency()));
	if (size < ((size_t)64 << 20)) nt = std::min(nt, 2);
	std::vector<Piece> pc((size_t)nt);
	// pass 1: every piece from the first record boundary at or behind its nominal start to the first at or behind its nominal end
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
	// pass 2: sequences to HBM
	if (hipSetDevice(device) != hipSuccess) return MCOM_E_HIP;
	uint8_t *dev = nullptr;
	if (hipMalloc(&dev, total * (size_t)len + 16) != hipSuccess) { (void)hipGetLastError(); return MCOM_E_NOMEM; }
	const size_t CH = (size_t)1 << 16;                                            // rows per staging block
	std::atomic<int> err{0};
	{
		std::vector<std::thread> th;
		for (int t = 0; t < nt; ++t) th.emplace_back([&, t]() {
			const Piece &P = pc[(size_t)t];
			if (!P.records) return;
			if (hipSetDevice(device) != hipSuccess) { err = MCOM_E_HIP; return; }
			hipStream_t cs = nullptr; unsigned char *blk[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; bool busy[2] = {false, false};
			bool ok = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) == hipSuccess;
			for (int b = 0; b < 2 && ok; ++b) ok = hipHostMalloc((void**)&blk[b], CH * (size_t)len, hipHostMallocDefault) == hipSuccess && hipEventCreateWithFlags(&ev[b], hipEventDisableTiming) == hipSuccess;
			const char *p = base + P.begin;
			size_t row = P.first_row, left = P.records;
			int cur = 0, bad_char = 0;
			while (ok && left) {
				const size_t take = std::min(CH, left);
				if (busy[cur]) { ok = hipEventSynchronize(ev[cur]) == hipSuccess; busy[cur] = false; }
				unsigned char *o = blk[cur];
				for (size_t r = 0; r < take && ok; ++r) {
					const char *s = next_line(p, end);
					memcpy(o, s, (size_t)len);
					unsigned flag = 0;
					for (int i = 0; i < len; ++i) { const unsigned char ch = o[i]; flag |= !(ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T' || ch == 'N'); }
					bad_char |= (int)flag;
					o += len;
					p = next_line(next_line(s + len + 1, end), end);                    // behind the '+' line and the quality line
				}
				if (bad_char) { err = MCOM_E_ARG; ok = false; break; }
				ok = ok && hipMemcpyAsync(dev + row * (size_t)len, blk[cur], take * (size_t)len, hipMemcpyHostToDevice, cs) == hipSuccess && hipEventRecord(ev[cur], cs) == hipSuccess;
				busy[cur] = true; cur ^= 1; row += take; left -= take;
			}
			if (cs) (void)hipStreamSynchronize(cs);
			if (!ok && !err) err = MCOM_E_HIP;
			for (int b = 0; b < 2; ++b) { if (blk[b]) (void)hipHostFree(blk[b]); if (ev[b]) (void)hipEventDestroy(ev[b]); }
			if (cs) (void)hipStreamDestroy(cs);
		});
		for (auto &x : th) x.join();
	}
	if (err) { (void)hipFree(dev); return err == MCOM_E_ARG ? 0 : (int)err; }           // a character outside ACGTN: the sequential reader says so
	*L = len; *d_reads = dev; *n = total;
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
