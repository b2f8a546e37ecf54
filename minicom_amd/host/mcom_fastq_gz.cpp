// minicom_amd/host/mcom_fastq_gz.cpp -- .fastq.gz whose gzip stream has MORE THAN ONE MEMBER, inflated and parsed by all cores (round 5).
//
// The reference reads its input through zlib's gzread (bseq.c:19-36, kseq.h), one thread; real inputs are .fastq.gz.  A gzip file is a
// sequence of members and most of the tools that write sequencing data write many of them: bgzip / htslib (BGZF: members of at most
// 64 KB, their compressed size in an extra field of the header), bcl-convert, pigz -i, or simply `cat a.gz b.gz`.  Members are
// independent deflate streams, so a file of many members can be inflated by many threads -- which one member cannot (a deflate
// stream is only decodable from its start).  This file does that:
//
//   discovery   the member that starts at offset 0 says whether the file is BGZF (then the member starts are walked header by header,
//               exactly) or not (then a start is looked for behind evenly spaced offsets, by all threads: the bytes 1f 8b 08 with
//               legal flag bits, confirmed by inflating the first kilobytes there).  One member only (plain `gzip`): that member cannot
//               be cut, but ONE thread decodes it in pieces of text -- ~1 GB/s against gzread's 0.25 -- and the pieces are the work
//               items of the same workers, which then only parse ("streaming" below).
//   groups      consecutive members are grouped into work items of a few MB of compressed data, numbered in file order
//   workers     a worker takes the next item, decodes its members (host/mcom_inflate.cpp; every member must end exactly where the next
//               begins: a false start found by the search cannot survive that), finds the first record boundary of its text by the
//               records' shape (a '@' line, a line of L bases, a '+' line, a line of L characters; mcom_fastq.cpp does the same for plain
//               files), parses the records that lie whole in its text and leaves their sequence lines as rows of L characters in a
//               page-locked block
//   uploader    one thread takes the items in order, puts the record that starts in an item's text and ends in the next one's together
//               (both texts are there by then), and sends the rows to HBM behind those of the items before (the same growing array as
//               the sequential reader's), straight from the block they were written to
// A window of items bounds the memory (a worker does not start item i before item i - window has been sent).  Anything that is not
// the fixed-length four-line layout -- FASTA, sequences over several lines, carriage returns, reads of several lengths -- and any
// inconsistency (a member that does not end at the next start, a text without a record boundary) makes the route step back (return 0)
// and the sequential reader takes the file, which also words the error messages.  Characters outside ACGTN are refused as everywhere.
#include "../../include/mcom.h"
#include <hip/hip_runtime_api.h>
#include <zlib.h>
#include <emmintrin.h>
#include "mcom_inflate.hpp"
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

size_t mcom_usable_cpus();                                                     // mcom_fastq.cpp
static std::atomic<long> g_last_items{0};
// test hook (include/mcom_test.h): work items of the last file this route read to its end (0: the last file went to the sequential reader)
extern "C" long mcomh_test_gz_items(void) { return g_last_items.load(); }

namespace {
struct Fd { int f; ~Fd() { if (f >= 0) close(f); } };

bool read_at(int fd, void *dst, size_t n, size_t off)
{
	char *d = (char*)dst;
	while (n) { const ssize_t g = pread(fd, d, n, (off_t)off); if (g <= 0) return false; d += g; off += (size_t)g; n -= (size_t)g; }
	return true;
}
// a gzip member header at h (at least 18 bytes readable, `have` bytes in all): 0 = not one; otherwise 1, and *bgzf_size = the member's
// compressed size when the header carries BGZF's BC field (0 otherwise)
int member_header(const unsigned char *h, size_t have, uint32_t *bgzf_size)
{
	*bgzf_size = 0;
	if (have < 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || (h[3] & 0xE0)) return 0;
	if (h[3] & 4) {                                                             // FEXTRA: subfields of { SI1, SI2, LEN, data }
		const size_t xlen = (size_t)h[10] | ((size_t)h[11] << 8);
		size_t p = 12, e = std::min(have, 12 + xlen);
		while (p + 4 <= e) {
			const size_t sl = (size_t)h[p + 2] | ((size_t)h[p + 3] << 8);
			if (h[p] == 'B' && h[p + 1] == 'C' && sl == 2 && p + 6 <= e) { *bgzf_size = ((uint32_t)h[p + 4] | ((uint32_t)h[p + 5] << 8)) + 1u; break; }
			p += 4 + sl;
		}
	}
	return 1;
}
// does a deflate stream that makes sense start here?  (the first bytes inflate without an error and give some output)
bool inflates_here(const unsigned char *p, size_t n)
{
	z_stream z; memset(&z, 0, sizeof z);
	if (inflateInit2(&z, 15 + 16) != Z_OK) return false;
	unsigned char out[4096];
	z.next_in = const_cast<unsigned char*>(p); z.avail_in = (uInt)std::min<size_t>(n, 1 << 16);
	z.next_out = out; z.avail_out = sizeof out;
	const int rc = inflate(&z, Z_SYNC_FLUSH);
	const bool ok = (rc == Z_OK || rc == Z_STREAM_END || rc == Z_BUF_ERROR) && z.total_out > 0;
	bool text = ok;
	for (uLong i = 0; text && i < z.total_out; ++i) text = out[i] == '\n' || (out[i] >= 32 && out[i] < 127);   // (a FASTQ file is printable text)
	inflateEnd(&z);
	return text;
}
// the start of the line behind the one p is in (e: there is none).  Names and '+' lines are short: sixteen bytes at a time in place of a call
inline const char *next_line(const char *p, const char *e)
{
	while (e - p >= 16) {
		const int m = _mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i*)p), _mm_set1_epi8('\n')));
		if (m) return p + __builtin_ctz((unsigned)m) + 1;
		p += 16;
	}
	const char *nl = (const char*)memchr(p, '\n', (size_t)(e - p));
	return nl ? nl + 1 : e;
}
// a record of the shape at p, all of it inside [p, e): its end, *seq = its sequence line; nullptr: not one (or not all of it there)
inline const char *record_in(const char *p, const char *e, int L, const char **seq = nullptr)
{
	if (p >= e || *p != '@') return nullptr;
	const char *s = next_line(p, e);
	if (s + L >= e || s[L] != '\n') return nullptr;
	const char *plus = s + L + 1;
	if (plus >= e || *plus != '+') return nullptr;
	const char *q = plus + 1 < e && plus[1] == '\n' ? plus + 2 : next_line(plus, e);
	if (q + L >= e || q[L] != '\n') return nullptr;
	if (seq) *seq = s;
	return q + L + 1;
}
// does a sequence line hold anything but A C G T N?  (sixteen characters at a time; the last sixteen overlap the ones before)
inline bool not_acgtn(const unsigned char *s, int len)
{
	if (len < 16) { bool bad = false; for (int q = 0; q < len; ++q) { const unsigned char c = s[q]; bad |= (c != 'A' && c != 'C' && c != 'G' && c != 'T' && c != 'N'); } return bad; }
	const __m128i a = _mm_set1_epi8('A'), c = _mm_set1_epi8('C'), g = _mm_set1_epi8('G'), t = _mm_set1_epi8('T'), n = _mm_set1_epi8('N');
	__m128i ok = _mm_set1_epi8((char)0xFF);
	auto step = [&](const unsigned char *p) {
		const __m128i v = _mm_loadu_si128((const __m128i*)p);
		ok = _mm_and_si128(ok, _mm_or_si128(_mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(v, a), _mm_cmpeq_epi8(v, c)), _mm_or_si128(_mm_cmpeq_epi8(v, g), _mm_cmpeq_epi8(v, t))), _mm_cmpeq_epi8(v, n)));
	};
	int q = 0;
	for (; q + 16 <= len; q += 16) step(s + q);
	if (q < len) step(s + len - 16);
	return _mm_movemask_epi8(ok) != 0xFFFF;
}
// Buffers that go round: the texts and the rows of a file are tens of gigabytes that would otherwise be fresh pages every time (a page fault
// per 4 KB, under one lock for all threads); a window of items bounds how many are out at once.  `pinned`: page-locked blocks the uploader's
// copies can leave from directly.
struct Pool {
	std::mutex m; std::vector<std::pair<void*, size_t>> idle; bool pinned = false; size_t made = 0;
	void *get(size_t want, size_t *cap)
	{
		{
			std::lock_guard<std::mutex> g(m);
			size_t best = idle.size();
			for (size_t i = 0; i < idle.size(); ++i) if (idle[i].second >= want && (best == idle.size() || idle[i].second < idle[best].second)) best = i;
			if (best < idle.size()) { void *p = idle[best].first; *cap = idle[best].second; idle[best] = idle.back(); idle.pop_back(); return p; }
			if (idle.size() > 8) { void *p = idle.back().first; idle.pop_back(); release(p); }      // (none fits and plenty lie around: do not hoard)
		}
		const size_t c = want + want / 8 + 4096;
		void *p = nullptr;
		if (pinned) { if (hipHostMalloc(&p, c, hipHostMallocDefault) != hipSuccess) return nullptr; }
		else p = malloc(c);
		if (p) { std::lock_guard<std::mutex> g(m); ++made; }
		*cap = c;
		return p;
	}
	void put(void *p, size_t cap) { if (!p) return; std::lock_guard<std::mutex> g(m); idle.emplace_back(p, cap); }
	void release(void *p) { if (pinned) (void)hipHostFree(p); else free(p); }
	~Pool() { for (auto &b : idle) release(b.first); }
};
struct Item {
	size_t begin = 0, end = 0;                                                  // compressed bytes [begin, end): whole members
	struct Text {                                                              // what they inflate to: the buffer the decoder wrote into, as it is
		char *p = nullptr; size_t cap = 0, n = 0, off = 0;                     // (off: one member in pieces -- the 32 KB before the piece lie in front of it)
		const char *data() const { return p + off; }
		size_t size() const { return n; }
		bool empty() const { return n == 0; }
	} text;
	size_t first = 0;                                                           // offset of the first record that starts in this text (text.size(): none)
	size_t tail = 0;                                                            // offset of the record that starts in this text and ends in a later one (text.size(): none)
	unsigned char *rows = nullptr; size_t rows_cap = 0, n_rows = 0;             // the sequence lines of the records that start in this text
	int state = 0;                                                              // 0 waiting, 1 text there, 2 rows there, 3 sent
};
}  // namespace

// 1 = the reads are on the device (*d_reads [n][L] characters, hipMalloc'ed), 0 = not this route (nothing kept), < 0 = error (err says what).
// host_out != NULL: the rows go to host memory instead (room for host_cap rows; MCOM_E_OVERFLOW when there are more) and no GPU is touched.
int mcom_fastq_gz_members(const char *path, int device, int *L_io, uint8_t **d_reads, uint8_t *host_out, size_t host_cap, size_t *n_out, size_t *n_members_out, std::string *err)
{
	if (d_reads) *d_reads = nullptr;
	*n_out = 0;
	g_last_items = 0;
	if (n_members_out) *n_members_out = 0;
	const auto wall0 = std::chrono::steady_clock::now();
	auto wall = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count(); };
	Fd fd{open(path, O_RDONLY)};
	if (fd.f < 0) return 0;
	struct stat st;
	if (fstat(fd.f, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 64) return 0;
	const size_t size = (size_t)st.st_size;
	unsigned char head[64];
	if (!read_at(fd.f, head, sizeof head, 0)) return 0;
	uint32_t bsz = 0;
	if (!member_header(head, sizeof head, &bsz)) return 0;
	// ---- discovery ------------------------------------------------------------------------------------------------------------------
	const size_t nt = std::max<size_t>(1, std::min<size_t>(64, mcom_usable_cpus()));
	std::vector<size_t> starts;                                                 // member starts that bound the work items (ascending, starts[0] = 0)
	starts.push_back(0);
	size_t n_members = 1;
	// (round 5, with the faster decoder: items of ~1/32 of a thread's share -- a few MB of compressed data -- so that the last items of the threads end
	// together and the first rows reach the uploader early)
	const size_t target = std::max<size_t>((size_t)256 << 10, std::min<size_t>((size_t)4 << 20, size / (32 * nt) + 1));   // compressed bytes per item
	if (bsz) {                                                                   // BGZF: every header says where the next one is
		size_t off = 0, last_cut = 0;
		std::vector<unsigned char> buf((size_t)1 << 20);
		size_t b0 = 0, bn = 0;                                                   // buf holds file bytes [b0, b0 + bn)
		for (;;) {
			if (off + 18 > size) return 0;                                       // (a BGZF file ends with an empty member: 28 bytes)
			if (off < b0 || off + 32 > b0 + bn) { b0 = off; bn = std::min(buf.size(), size - off); if (!read_at(fd.f, buf.data(), bn, b0)) return 0; }
			uint32_t s = 0;
			if (!member_header(buf.data() + (off - b0), bn - (off - b0), &s) || !s) return 0;
			const size_t nxt = off + s;
			if (nxt > size) return 0;
			if (nxt == size) break;
			++n_members;
			if (nxt - last_cut >= target) { starts.push_back(nxt); last_cut = nxt; }
			off = nxt;
		}
	} else {                                                                     // any gzip: look for a member start behind evenly spaced offsets
		const size_t step = target;
		const size_t np = size > 64 + step ? (size - 64 - 1) / step : 0;         // probes at step, 2 step, ...
		std::vector<size_t> found(np, 0);
		std::atomic<size_t> next_probe{0};
		std::atomic<bool> io_error{false};
		auto prober = [&]() {
			std::vector<unsigned char> buf((size_t)1 << 20);
			for (;;) {
				const size_t k = next_probe.fetch_add(1);
				if (k >= np) return;
				const size_t want = (k + 1) * step;
				// (a start further away than the next probe's offset is that probe's to find)
				for (size_t base = want; base + 64 < size && base < want + step && !found[k]; base += buf.size() - 64) {
					const size_t bn = std::min(buf.size(), size - base);
					if (!read_at(fd.f, buf.data(), bn, base)) { io_error = true; return; }
					const unsigned char *p = buf.data(), *e = buf.data() + bn - 32;
					while (p < e && !found[k]) {
						p = (const unsigned char*)memchr(p, 0x1f, (size_t)(e - p));
						if (!p) break;
						uint32_t s = 0;
						if (base + (size_t)(p - buf.data()) < want + step && member_header(p, (size_t)(buf.data() + bn - p), &s) && inflates_here(p, (size_t)(buf.data() + bn - p)))
							found[k] = base + (size_t)(p - buf.data());
						++p;
					}
				}
			}
		};
		{
			std::vector<std::thread> pt;
			for (size_t t = 0; t < std::min(nt, std::max<size_t>(np, 1)); ++t) pt.emplace_back(prober);
			for (auto &x : pt) x.join();
		}
		if (io_error) return 0;
		for (size_t k = 0; k < np; ++k) if (found[k] && found[k] > starts.back()) { starts.push_back(found[k]); ++n_members; }
	}
	// One member (or too few to share out): the member cannot be cut, but its decoding and the parsing can be two stages -- ONE thread
	// decodes (host/mcom_inflate.cpp with its output in pieces, ~1 GB/s of text against gzread's 0.25) and hands pieces of text to the
	// same workers, which only parse.  The pieces are the items; how many there will be is known when the member ends.
	const bool streaming = starts.size() < 2;
	if (n_members_out) *n_members_out = n_members;
	const size_t piece = std::max<size_t>((size_t)8 << 20, size / 60 + 1);      // bytes of text per piece (DEFLATE expands at most 1032 x: at most ~62 000 pieces)
	const size_t ni_cap = streaming ? (size_t)65536 : starts.size();
	std::vector<Item> items(ni_cap);
	std::atomic<size_t> ni{streaming ? (size_t)-1 : starts.size()};             // the number of items ((size_t)-1: not known yet)
	size_t produced = streaming ? 0 : starts.size();                            // items that exist (guarded by mu)
	if (!streaming) for (size_t i = 0; i < starts.size(); ++i) { items[i].begin = starts[i]; items[i].end = i + 1 < starts.size() ? starts[i + 1] : size; }
	// ---- the read length: from the head of the first item's text -------------------------------------------------------------------
	int L = *L_io;
	// ---- workers + uploader --------------------------------------------------------------------------------------------------------
	std::mutex mu; std::condition_variable cv;
	std::atomic<size_t> next_item{0};
	size_t sent = 0;                                                            // items the uploader is done with (guarded by mu)
	std::atomic<int> fail{0};                                                   // 1 = not this route after all, < 0 = error
	std::string fail_msg;
	const size_t window = streaming ? nt + 4 : 3 * nt + 2;
	std::atomic<size_t> est_items{0};                                           // (streaming: the decoder's guess after its first piece, for the size of the device array)
	int L_shared = L;                                                           // (guarded by mu; item 0 settles it when *L_io is 0)
	auto give_up = [&](int code, const char *msg) { { std::lock_guard<std::mutex> g(mu); if (!fail) { fail = code; fail_msg = msg; } } cv.notify_all(); };
	Pool texts, rowbufs;
	rowbufs.pinned = host_out == nullptr;                                      // (towards the device the rows leave from page-locked blocks)
	if (rowbufs.pinned && hipSetDevice(device) != hipSuccess) return MCOM_E_HIP;
	const bool trace = getenv("MCOM_GZ_TRACE") != nullptr;
	const double w_discovery = wall();
	double w_workers = 0, w_first_sent = 0;
	std::atomic<long long> t_read{0}, t_inflate{0}, t_parse{0}, t_wait{0}, t_up_wait{0}, t_up_copy{0};
	auto now_ns = []() { return (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	auto worker = [&]() {
		std::vector<unsigned char> in;
		if (rowbufs.pinned) (void)hipSetDevice(device);                           // (this thread may be the one that page-locks a new block)
		for (;;) {
			const size_t i = next_item.fetch_add(1);
			if (i >= ni || fail) return;
			long long tq = now_ns();
			{ std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return fail || i >= ni || (i < produced && i < sent + window); }); if (fail || i >= ni) return; }
			t_wait += now_ns() - tq; tq = now_ns();
			Item &it = items[i];
			if (!streaming) {
			// inflate the item's members
			in.resize(it.end - it.begin);
			if (!read_at(fd.f, in.data(), in.size(), it.begin)) { give_up(MCOM_E_ARG, "read error"); return; }
			t_read += now_ns() - tq; tq = now_ns();
			// every member of the item, one behind the other, by the decoder of mcom_inflate.cpp (header, deflate stream, CRC-32 and ISIZE checked;
			// a member must end where the next begins).  The text buffer grows by doubling when a member does not fit; the member is then decoded again.
			size_t tcap = 0, have = 0;
			char *tbuf = (char*)texts.get(std::max<size_t>(in.size() * 6, (size_t)1 << 20), &tcap);
			if (!tbuf) { give_up(MCOM_E_NOMEM, "out of memory"); return; }
			bool bad = false;
			for (size_t at = 0; at < in.size();) {
				size_t used = 0, got = 0;
				const int rc = mcom_gunzip_member(in.data() + at, in.size() - at, (uint8_t*)tbuf + have, tcap - 16 - have, &used, &got);
				if (rc == MCOM_INFLATE_ROOM) {
					size_t ncap = 0;
					char *nb = (char*)texts.get(2 * tcap, &ncap);
					if (!nb) { texts.put(tbuf, tcap); give_up(MCOM_E_NOMEM, "out of memory"); return; }
					memcpy(nb, tbuf, have); texts.put(tbuf, tcap); tbuf = nb; tcap = ncap;
					continue;
				}
				if (rc != MCOM_INFLATE_OK) { bad = true; break; }
				at += used; have += got;
			}
			if (bad) { texts.put(tbuf, tcap); give_up(1, "the members do not tile the file"); return; }
			it.text.p = tbuf; it.text.cap = tcap; it.text.n = have;
			}
			t_inflate += now_ns() - tq; tq = now_ns();
			// the read length (item 0, from its first record) and this text's first record boundary
			const char *tb = it.text.data(), *te = tb + it.text.size();
			int len;
			if (i == 0) {
				if (it.text.empty() || tb[0] != '@') { give_up(1, "not FASTQ"); return; }
				const char *s0 = next_line(tb, te);
				const char *s1 = (const char*)memchr(s0, '\n', (size_t)(te - s0));
				if (!s1) { give_up(1, "no sequence line"); return; }
				len = (int)(s1 - s0);
				{ std::lock_guard<std::mutex> g(mu); if (L_shared == 0) L_shared = len; len = L_shared; }
				if (len < 1 || len > 256 || (int)(s1 - s0) != len) { give_up(1, "read length"); return; }
				it.first = 0;
				cv.notify_all();
			} else {
				{ std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return fail || L_shared != 0; }); if (fail) return; len = L_shared; }
				// candidates: the text's own first byte (a line start iff the text before ended with a newline, which nobody knows yet: the shape
				// decides, and a wrong guess makes the record across the two texts fail its shape test -- the route steps back), then the line starts
				const char *p = tb == te ? te : next_line(tb, te);
				const char *found = nullptr;
				if (record_in(tb, te, len)) found = tb;
				for (int tries = 0; !found && tries < 8 && p < te; ++tries) { if (record_in(p, te, len)) found = p; else p = next_line(p, te); }
				it.first = found ? (size_t)(found - tb) : it.text.size();
				if (!found && it.text.size() > (size_t)(8 * (2 * len + 512))) { give_up(1, "no record boundary"); return; }
			}
			{ std::lock_guard<std::mutex> g(mu); it.state = 1; }
			cv.notify_all();
			// parse: the records that start in this text; the last one may end in the following texts
			it.n_rows = 0;
			const size_t row_cap = it.text.size() / (size_t)(2 * len + 5) + 2;       // (a record is at least "@\n" + L + "\n+\n" + L + "\n")
			it.rows = (unsigned char*)rowbufs.get(row_cap * (size_t)len + 16, &it.rows_cap);
			if (!it.rows) { give_up(MCOM_E_NOMEM, "out of (page-locked) memory"); return; }
			size_t pos = it.first;
			unsigned bad_char = 0;
			auto take = [&](const char *seq) {
				if (it.n_rows >= row_cap) { bad_char |= 2u; return; }
				unsigned char *dst = it.rows + it.n_rows * (size_t)len;
				memcpy(dst, seq, (size_t)len);
				bad_char |= not_acgtn(dst, len) ? 1u : 0u;
				++it.n_rows;
			};
			while (pos < it.text.size()) {
				const char *sq = nullptr;
				const char *q = record_in(tb + pos, te, len, &sq);
				if (!q) break;
				take(sq);
				pos = (size_t)(q - tb);
			}
			// a record that starts at pos and does not end in this text is put together by the uploader, which passes the items in order and
			// has the following texts at hand by then (a worker that waited for its neighbour's text here stood still for a fifth of its time)
			it.tail = pos;
			t_parse += now_ns() - tq;
			if (bad_char & 2u) { give_up(1, "more records than the text has room for"); return; }
			if (bad_char) { give_up(MCOM_E_ARG, "a sequence holds a character outside ACGTN (lower-case and IUPAC codes are not representable)"); return; }
			{ std::lock_guard<std::mutex> g(mu); it.state = 2; }
			cv.notify_all();
		}
	};
	uint8_t *dev = nullptr; size_t dev_cap = 0, total = 0;
	int up_rc = 0;
	std::thread up([&]() {
		hipStream_t cs = nullptr;
		// copies in flight: the block a copy leaves from goes back to the pool when the event behind the copy has happened
		struct Flight { void *p; size_t cap; hipEvent_t ev; };
		std::vector<Flight> flying; std::vector<hipEvent_t> spare;
		auto land = [&](size_t keep) {                                          // wait until at most `keep` copies are in flight
			while (flying.size() > keep) {
				if (hipEventSynchronize(flying.front().ev) != hipSuccess) return false;
				rowbufs.put(flying.front().p, flying.front().cap); spare.push_back(flying.front().ev);
				flying.erase(flying.begin());
			}
			return true;
		};
		auto bail = [&](int code, const char *msg) { up_rc = code; give_up(code, msg); };
		const bool to_host = host_out != nullptr;
		if (!to_host && (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess)) { bail(MCOM_E_HIP, "no usable GPU"); cs = nullptr; }
		for (size_t i = 0; i < ni && (cs || to_host); ++i) {
			long long tq = now_ns();
			{ std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return fail || i >= ni || items[i].state >= 2; }); if (fail || i >= ni) break; }
			t_up_wait += now_ns() - tq; tq = now_ns();
			Item &it = items[i];
			int len; { std::lock_guard<std::mutex> g(mu); len = L_shared; }
			if (it.tail < it.text.size()) {
				// the record across the end of this text: its start here, then what lies in front of the following texts' first records
				std::string rec(it.text.data() + it.tail, it.text.size() - it.tail);
				bool done = false, bad_shape = false;
				for (size_t j = i + 1; !done && !bad_shape; ++j) {
					{ std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return fail || j >= ni || items[j].state >= 1; }); if (fail) break; }
					if (j >= ni) { rec.push_back('\n'); done = true; break; }              // the file ends here: the last line may lack its newline
					const Item &nx = items[j];
					const size_t headn = std::min(nx.first, nx.text.size());
					rec.append(nx.text.data(), headn);
					if (headn < nx.text.size() || j + 1 == ni) {
						if (headn == nx.text.size() && j + 1 == ni && (rec.empty() || rec.back() != '\n')) rec.push_back('\n');
						done = true;
					}
				}
				if (fail) break;
				const char *sq = nullptr;
				const char *q = done ? record_in(rec.data(), rec.data() + rec.size(), len, &sq) : nullptr;
				if (!q || q != rec.data() + rec.size()) { bail(1, "a record across two members is not of the shape"); break; }
				if ((it.n_rows + 1) * (size_t)len + 16 > it.rows_cap) { bail(1, "more records than the text has room for"); break; }
				memcpy(it.rows + it.n_rows * (size_t)len, sq, (size_t)len);
				if (not_acgtn(it.rows + it.n_rows * (size_t)len, len)) { bail(MCOM_E_ARG, "a sequence holds a character outside ACGTN (lower-case and IUPAC codes are not representable)"); break; }
				++it.n_rows;
			}
			const size_t bytes = it.n_rows * (size_t)len;
			bool kept = false;                                                      // the block stays out (a copy is leaving from it)
			if (bytes && to_host) {
				if (total + it.n_rows > host_cap) { total += it.n_rows; bail(MCOM_E_OVERFLOW, "more reads than the caller has room for"); break; }
				memcpy(host_out + total * (size_t)len, it.rows, bytes);
				total += it.n_rows;
			} else if (bytes) {
				if (total + it.n_rows > dev_cap) {
					const size_t items_in_all = ni != (size_t)-1 ? (size_t)ni : std::max<size_t>(est_items.load(), i + 1);
					size_t want = dev_cap ? dev_cap * 2 : std::max<size_t>(it.n_rows * (items_in_all + 1) + it.n_rows / 4, (size_t)1 << 20);
					while (want < total + it.n_rows) want *= 2;
					uint8_t *nd = nullptr;
					if (hipMalloc(&nd, want * (size_t)len + 16) != hipSuccess) { bail(MCOM_E_NOMEM, "out of device memory"); break; }
					if (hipStreamSynchronize(cs) != hipSuccess || (dev && total && hipMemcpy(nd, dev, total * (size_t)len, hipMemcpyDeviceToDevice) != hipSuccess)) { (void)hipFree(nd); bail(MCOM_E_HIP, "upload failed"); break; }
					if (dev) (void)hipFree(dev);
					dev = nd; dev_cap = want;
				}
				if (!land(6)) { bail(MCOM_E_HIP, "upload failed"); break; }
				hipEvent_t ev = nullptr;
				if (!spare.empty()) { ev = spare.back(); spare.pop_back(); }
				else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { bail(MCOM_E_HIP, "event"); break; }
				if (hipMemcpyAsync(dev + total * (size_t)len, it.rows, bytes, hipMemcpyHostToDevice, cs) != hipSuccess || hipEventRecord(ev, cs) != hipSuccess) { spare.push_back(ev); bail(MCOM_E_HIP, "upload failed"); break; }
				flying.push_back(Flight{it.rows, it.rows_cap, ev}); kept = true;
				total += it.n_rows;
			}
			if (!kept) rowbufs.put(it.rows, it.rows_cap);
			it.rows = nullptr;
			t_up_copy += now_ns() - tq;
			if (i) { texts.put(items[i - 1].text.p, items[i - 1].text.cap); items[i - 1].text.p = nullptr; }   // (item i is through: nobody reads the text before it any more)
			if (i == 0) w_first_sent = wall();
			{ std::lock_guard<std::mutex> g(mu); it.state = 3; sent = i + 1; }
			cv.notify_all();
		}
		if (cs) { if (hipStreamSynchronize(cs) != hipSuccess && !up_rc) up_rc = MCOM_E_HIP; }
		(void)land(0);
		for (auto &f : flying) { rowbufs.put(f.p, f.cap); spare.push_back(f.ev); }     // (only after a failed wait)
		for (hipEvent_t e : spare) (void)hipEventDestroy(e);
		if (cs) (void)hipStreamDestroy(cs);
	});
	// ---- streaming: the one thread that decodes -------------------------------------------------------------------------------------
	auto producer = [&]() {
		void *map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd.f, 0);
		if (map == MAP_FAILED) { give_up(1, "mmap"); return; }
		(void)madvise(map, size, MADV_SEQUENTIAL);
		const uint8_t *base = (const uint8_t*)map;
		const size_t H = 32768;
		std::vector<char> hist(H);
		size_t at = 0, k = 0;
		bool file_done = false;
		while (!file_done && !fail) {
			// the member's header (RFC 1952)
			uint32_t b2 = 0;
			if (size - at < 18 || !member_header(base + at, size - at, &b2)) { give_up(1, "not a gzip member"); break; }
			const unsigned flg = base[at + 3];
			size_t hp = at + 10;
			if (flg & 4) { hp += 2 + ((size_t)base[at + 10] | ((size_t)base[at + 11] << 8)); }
			if (flg & 8) { while (hp < size && base[hp]) ++hp; ++hp; }
			if (flg & 16) { while (hp < size && base[hp]) ++hp; ++hp; }
			if (flg & 2) hp += 2;
			if (hp + 8 > size) { give_up(1, "truncated gzip file"); break; }
			mcom_inflate_stream st;
			mcom_inflate_begin(&st, base + hp, size - hp);
			uint32_t crc = 0; uint64_t isize = 0; size_t hist_n = 0;
			bool member_done = false;
			while (!member_done) {
				if (k >= ni_cap) { give_up(1, "too many pieces"); break; }
				{ std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return fail || produced < sent + window; }); if (fail) break; }
				size_t cap = 0;
				char *buf = (char*)texts.get(H + piece + 64, &cap);
				if (!buf) { give_up(MCOM_E_NOMEM, "out of memory"); break; }
				memcpy(buf + H - hist_n, hist.data() + H - hist_n, hist_n);       // (the history sits at the END of hist)
				size_t n = 0;
				const long long tq = now_ns();
				const int rc = mcom_inflate_run(&st, (uint8_t*)buf + H, piece, hist_n, &n);
				if (rc != MCOM_INFLATE_OK && rc != MCOM_INFLATE_ROOM) { texts.put(buf, cap); give_up(1, "not a valid deflate stream"); break; }
				crc = mcom_crc32(crc, (const uint8_t*)buf + H, n); isize += n;
				t_inflate += now_ns() - tq;
				if (n >= H) { memcpy(hist.data(), buf + H + n - H, H); hist_n = H; }
				else { memmove(hist.data(), hist.data() + n, H - n); memcpy(hist.data() + H - n, buf + H, n); hist_n = std::min(H, hist_n + n); }
				member_done = rc == MCOM_INFLATE_OK;
				if (member_done) {
					const size_t tr = (size_t)(st.in - base);
					if (tr + 8 > size) { texts.put(buf, cap); give_up(1, "truncated gzip file"); break; }
					const uint32_t c = (uint32_t)base[tr] | ((uint32_t)base[tr + 1] << 8) | ((uint32_t)base[tr + 2] << 16) | ((uint32_t)base[tr + 3] << 24);
					const uint32_t z = (uint32_t)base[tr + 4] | ((uint32_t)base[tr + 5] << 8) | ((uint32_t)base[tr + 6] << 16) | ((uint32_t)base[tr + 7] << 24);
					if (c != crc || z != (uint32_t)isize) { texts.put(buf, cap); give_up(1, "gzip member: CRC-32 or length differ"); break; }
					at = tr + 8;
					file_done = at >= size;
				}
				if (k == 0) { const size_t used = std::max<size_t>((size_t)(st.in - base), 1); est_items = size / used + size / used / 8 + 3; }
				{
					std::lock_guard<std::mutex> g(mu);
					Item &it = items[k];
					it.text.p = buf; it.text.cap = cap; it.text.n = n; it.text.off = H;
					produced = ++k;
					if (file_done) ni = k;
				}
				cv.notify_all();
			}
			mcom_inflate_end(&st);
			if (!member_done) break;
		}
		munmap(map, size);
	};
	std::thread prod;
	if (streaming) prod = std::thread(producer);
	std::vector<std::thread> th;
	for (size_t t = 0; t < std::max<size_t>(2, std::min<size_t>(nt, ni)); ++t) th.emplace_back(worker);   // (two at least: the worker of item i waits for the text of item i + 1, which somebody else must make)
	for (auto &x : th) x.join();
	if (prod.joinable()) prod.join();
	w_workers = wall();
	{ std::lock_guard<std::mutex> g(mu); if (!fail && next_item < ni) fail = 1; }
	cv.notify_all();
	up.join();
	for (Item &it : items) { if (it.text.p) { texts.put(it.text.p, it.text.cap); it.text.p = nullptr; } if (it.rows) { rowbufs.put(it.rows, it.rows_cap); it.rows = nullptr; } }
	if (trace) fprintf(stderr, "mcom_fastq_gz: wall: discovery %.3f, first item sent %.3f, workers joined %.3f, uploader joined %.3f\n", w_discovery, w_first_sent, w_workers, wall());
	if (trace) fprintf(stderr, "mcom_fastq_gz: %zu items, %zu threads; worker seconds (sum over threads): wait %.3f read %.3f inflate %.3f parse %.3f; uploader: wait %.3f copy %.3f\n", (size_t)ni, th.size(),
	                   t_wait / 1e9, t_read / 1e9, t_inflate / 1e9, t_parse / 1e9, t_up_wait / 1e9, t_up_copy / 1e9);
	if (fail || up_rc) {
		if (dev) (void)hipFree(dev);
		const int code = fail ? (int)fail : up_rc;
		if (code < 0 && err) *err = fail_msg;
		return code < 0 ? code : 0;
	}
	if (!total) { if (dev) (void)hipFree(dev); return 0; }
	*L_io = L_shared; *n_out = total;
	if (d_reads) *d_reads = dev;
	g_last_items = (long)ni;
	return 1;
}
int mcom_fastq_gz_members_to_device(const char *path, int device, int *L_io, uint8_t **d_reads, size_t *n_out, size_t *n_members_out, std::string *err)
{
	return mcom_fastq_gz_members(path, device, L_io, d_reads, nullptr, 0, n_out, n_members_out, err);
}
