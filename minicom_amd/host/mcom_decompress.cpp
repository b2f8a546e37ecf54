// minicom_amd/host/mcom_decompress.cpp -- inverse of the stream files written by mcomh_cluster_dump
// (= the reference's cluster_dump at one thread, single-end, not order-preserving).
//
// Restates the reference decoder for that mode (decompress.c: decomp_AATTNN_nonorder :646, decomp_single_nonorder
// :612, decompress_nonorder :495, helpers :40-100) so that losslessness (parity level P2) can be
// proven on a machine where the reference does not exist.  Plain host code, no GPU.
#include "../../include/mcom_host.h"
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace {

bool slurp(const std::string &path, std::vector<uint8_t> &out)
{
	FILE *f = fopen(path.c_str(), "rb");
	if (!f) return false;
	fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
	out.resize((size_t)n);
	size_t got = n ? fread(out.data(), 1, (size_t)n, f) : 0;
	fclose(f);
	return got == (size_t)n;
}

struct DnaReader {                      // getDNAcode (decompress.c:56-74): 4 bases per byte, low bits first
	const std::vector<uint8_t> &b; size_t pos = 0; int k = 4; unsigned cur = 0;
	explicit DnaReader(const std::vector<uint8_t> &b_) : b(b_) {}
	int next() { if (k >= 4) { if (pos >= b.size()) return -1; cur = b[pos++]; k = 0; } int c = (int)(cur & 3); cur >>= 2; ++k; return c; }
};
struct BitReader {                      // getDir (decompress.c:40-54): 8 flags per byte, low bit first
	const std::vector<uint8_t> &b; size_t pos = 0; int k = 8; unsigned cur = 0;
	explicit BitReader(const std::vector<uint8_t> &b_) : b(b_) {}
	int next() { if (k >= 8) { cur = pos < b.size() ? b[pos++] : 0; k = 0; } int c = (int)(cur & 1); cur >>= 1; ++k; return c; }
};

// text of a read against a reference string: letters are literal bases, digits a run of matching bases
// (decompress.c:573-590); the missing tail matches
// The archive is untrusted input: a digit run that would read past the L bases of the reference string, a character
// that is neither a base letter nor a digit, or a line that decodes to more than L bases is an error, not a crash.
bool decode_line(const char *txt, size_t n, const char *ref, int L, std::string &seq)
{
	seq.clear();
	long eq = 0;
	for (size_t i = 0; i < n; ++i) {
		const char ch = txt[i];
		if (ch >= 'A' && ch <= 'Z') {
			if ((long)seq.size() + eq + 1 > (long)L) return false;
			for (long j = 0; j < eq; ++j) seq.push_back(ref[seq.size()]);
			eq = 0;
			seq.push_back(ch);
		} else if (ch >= '0' && ch <= '9') {
			eq = eq * 10 + (ch - '0');
			if (eq > (long)L) return false;
		} else return false;
	}
	while ((int)seq.size() < L) seq.push_back(ref[seq.size()]);
	return (int)seq.size() == L;
}

size_t file_bytes(const std::string &path)
{
	FILE *f = fopen(path.c_str(), "rb");
	if (!f) return 0;
	fseek(f, 0, SEEK_END); const long n = ftell(f); fclose(f);
	return n > 0 ? (size_t)n : 0;
}

void revcomp(std::string &s)
{
	const size_t n = s.size();
	for (size_t i = 0, j = n ? n - 1 : 0; i < j; ++i, --j) std::swap(s[i], s[j]);
	for (char &c : s) c = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'N';
}

} // namespace

static int decompress_impl(const char *folder, const char *out_path, uint64_t *n_reads)
{
	if (!folder || !out_path) return -1;
	const std::string dir(folder);
	FILE *fi = fopen((dir + "/info.txt").c_str(), "r");
	if (!fi) return -1;
	int L = 0, nth = 0; long na = 0, nt = 0, nn = 0;
	if (fscanf(fi, "%d %d %ld %ld %ld", &L, &nth, &na, &nt, &nn) != 5) { fclose(fi); return -1; }
	fclose(fi);
	if (L < 1 || L > 256 || nth < 1 || nth > 4096 || na < 0 || nt < 0 || nn < 0) return -1;
	FILE *out = fopen(out_path, "w");
	if (!out) return -1;
	uint64_t total = 0;
	auto line = [&](const std::string &s) { fwrite(s.data(), 1, s.size(), out); fputc('\n', out); ++total; };
	// all-A / all-T / all-N reads are only counted (decompress.c:660-688)
	for (long i = 0; i < na; ++i) line(std::string((size_t)L, 'A'));
	for (long i = 0; i < nt; ++i) line(std::string((size_t)L, 'T'));
	for (long i = 0; i < nn; ++i) line(std::string((size_t)L, 'N'));
	// near-constant reads: text against a constant base (:690-760)
	std::vector<uint8_t> buf;
	std::string seq;
	const char bases[3] = {'A', 'T', 'N'}; const char *names[3] = {"AA.txt", "TT.txt", "NN.txt"};
	for (int q = 0; q < 3; ++q) {
		if (!slurp(dir + "/" + names[q], buf)) { fclose(out); return -1; }
		const std::string cref((size_t)L, bases[q]);
		size_t s = 0;
		for (size_t i = 0; i < buf.size(); ++i) if (buf[i] == '\n') { if (!decode_line((const char*)buf.data() + s, i - s, cref.c_str(), L, seq)) { fclose(out); return -1; } line(seq); s = i + 1; }
	}
	// reads kept as text because they contain N
	if (!slurp(dir + "/single_N.seq", buf)) { fclose(out); return -1; }
	{ size_t s = 0; for (size_t i = 0; i < buf.size(); ++i) if (buf[i] == '\n') { if (i - s != (size_t)L) { fclose(out); return -1; } line(std::string((const char*)buf.data() + s, i - s)); s = i + 1; } }
	// unclustered reads, 2 bits per base (:612-644); a trailing partial byte carries no complete read
	if (!slurp(dir + "/single.seq", buf)) { fclose(out); return -1; }
	{
		DnaReader r(buf);
		for (;;) {
			seq.clear();
			int c = 0;
			for (int i = 0; i < L; ++i) { c = r.next(); if (c < 0) break; seq.push_back("ACGT"[c]); }
			if (c < 0 || (int)seq.size() < L) break;
			line(seq);
		}
	}
	// contigs and their reads, one stream set per writer thread (:495-610)
	for (int th = 0; th < nth; ++th) {
		std::vector<uint8_t> bref, bpos, bdir, bdif;
		const std::string sfx = "." + std::to_string(th);
		if (!slurp(dir + "/ref.bin" + sfx, bref) || !slurp(dir + "/beg_pos.bin" + sfx, bpos) || !slurp(dir + "/dir.bin" + sfx, bdir) ||
		    !slurp(dir + "/dif_char.txt" + sfx, bdif)) { fclose(out); return -1; }
		DnaReader rr(bref); BitReader dr(bdir);
		size_t pp = 0, dp = 0;
		std::string ref;
		while (pp + 4 <= bpos.size()) {
			uint32_t num; memcpy(&num, bpos.data() + pp, 4); pp += 4;
			ref.clear();
			int pre = 0;
			for (uint32_t q = 0; q < num; ++q) {
				if (pp + 2 > bpos.size()) { fclose(out); return -1; }
				uint16_t d; memcpy(&d, bpos.data() + pp, 2); pp += 2;
				const int pos = pre + d; pre = pos;
				while ((int)ref.size() < pos + L) { const int c = rr.next(); if (c < 0) { fclose(out); return -1; } ref.push_back("ACGT"[c]); }   // getRef (:92-100)
				const int rev = dr.next();
				size_t e = dp; while (e < bdif.size() && bdif[e] != '\n') ++e;
				if (e >= bdif.size() || !decode_line((const char*)bdif.data() + dp, e - dp, ref.c_str() + pos, L, seq)) { fclose(out); return -1; }
				dp = e + 1;
				if (rev) revcomp(seq);
				line(seq);
			}
		}
	}
	fclose(out);
	if (n_reads) *n_reads = total;
	return 0;
}

// ---- order-preserving mode (-p): every stream has a companion stream of read ids ------------------------------------
// Restates decomp_AATTNN_order (decompress.c:299-493), decomp_single_order (:238-297) and decompress_order (:109-236):
// the reads are rebuilt into a table indexed by their original position and written out in that order.  Id streams
// are uint32, delta coded inside a list (lists are sorted by id); in ids.bin.T a member carries its id when it is the
// first of its contig or starts at a new position, else the difference to the previous member's id (:164-167).
static int decompress_order_impl(const char *folder, const char *out_path, uint64_t *n_reads)
{
	if (!folder || !out_path) return -1;
	const std::string dir(folder);
	FILE *fi = fopen((dir + "/info.txt").c_str(), "r");
	if (!fi) return -1;
	int L = 0, nth = 0; long na = 0, nt = 0, nn = 0; unsigned long n_seq = 0;
	if (fscanf(fi, "%d %d %ld %ld %ld %lu", &L, &nth, &na, &nt, &nn, &n_seq) != 6) { fclose(fi); return -1; }
	fclose(fi);
	if (L < 1 || L > 256 || nth < 1 || nth > 4096 || na < 0 || nt < 0 || nn < 0) return -1;
	{   // every read of this mode has a 4-byte entry in one of the id streams: n_seq cannot exceed what they hold
		size_t idb = 0;
		for (const char *nm : {"allA.ids.bin", "allT.ids.bin", "allN.ids.bin", "AA.ids.bin", "TT.ids.bin", "NN.ids.bin", "singleFile.ids.bin", "Nfile.ids.bin"}) idb += file_bytes(dir + "/" + nm);
		for (int th = 0; th < nth; ++th) idb += file_bytes(dir + "/ids.bin." + std::to_string(th));
		if ((size_t)n_seq > idb / 4) return -1;
	}
	std::vector<char> table((size_t)n_seq * (size_t)L, 0);
	std::vector<uint8_t> seen((size_t)n_seq, 0);
	bool bad = false;
	auto put = [&](uint64_t idx, const std::string &s) {
		if (idx >= n_seq || (int)s.size() != L || seen[idx]) { bad = true; return; }
		memcpy(table.data() + idx * (size_t)L, s.data(), (size_t)L); seen[idx] = 1;
	};
	struct Ids {                                                        // one delta-coded id list
		std::vector<uint8_t> b; size_t p = 0; uint64_t pre = 0;
		bool next(uint64_t &idx) { if (p + 4 > b.size()) return false; uint32_t v; memcpy(&v, b.data() + p, 4); p += 4; pre += v; idx = pre; return true; }
	};
	std::vector<uint8_t> buf;
	std::string seq;
	uint64_t idx = 0;
	// all-A / all-T / all-N (:320-372)
	{
		const char bases[3] = {'A', 'T', 'N'}; const char *names[3] = {"allA.ids.bin", "allT.ids.bin", "allN.ids.bin"}; const long cnt[3] = {na, nt, nn};
		for (int q = 0; q < 3; ++q) {
			Ids ids; if (!slurp(dir + "/" + names[q], ids.b)) return -1;
			for (long i = 0; i < cnt[q]; ++i) { if (!ids.next(idx)) return -1; put(idx, std::string((size_t)L, bases[q])); }
		}
	}
	// near-constant reads (:374-493)
	{
		const char bases[3] = {'A', 'T', 'N'}; const char *names[3] = {"AA", "TT", "NN"};
		for (int q = 0; q < 3; ++q) {
			Ids ids;
			if (!slurp(dir + "/" + names[q] + ".txt", buf) || !slurp(dir + "/" + names[q] + ".ids.bin", ids.b)) return -1;
			const std::string cref((size_t)L, bases[q]);
			size_t s = 0;
			for (size_t i = 0; i < buf.size(); ++i) if (buf[i] == '\n') {
				if (!decode_line((const char*)buf.data() + s, i - s, cref.c_str(), L, seq)) return -1;
				s = i + 1;
				if (!ids.next(idx)) return -1;
				put(idx, seq);
			}
		}
	}
	// unclustered reads (:238-280) and reads kept as text (:282-296)
	{
		Ids ids;
		if (!slurp(dir + "/single.seq", buf) || !slurp(dir + "/singleFile.ids.bin", ids.b)) return -1;
		DnaReader r(buf);
		for (;;) {
			seq.clear();
			int c = 0;
			for (int i = 0; i < L; ++i) { c = r.next(); if (c < 0) break; seq.push_back("ACGT"[c]); }
			if (c < 0 || (int)seq.size() < L) break;
			if (!ids.next(idx)) break;                                   // padding of the last byte can look like one more read of A
			put(idx, seq);
		}
		Ids nids;
		if (!slurp(dir + "/single_N.seq", buf) || !slurp(dir + "/Nfile.ids.bin", nids.b)) return -1;
		size_t s = 0;
		for (size_t i = 0; i < buf.size(); ++i) if (buf[i] == '\n') { seq.assign((const char*)buf.data() + s, i - s); s = i + 1; if (!nids.next(idx)) return -1; put(idx, seq); }
	}
	// contigs (:109-236)
	for (int th = 0; th < nth; ++th) {
		std::vector<uint8_t> bref, bpos, bdir, bdif, bids;
		const std::string sfx = "." + std::to_string(th);
		if (!slurp(dir + "/ref.bin" + sfx, bref) || !slurp(dir + "/beg_pos.bin" + sfx, bpos) || !slurp(dir + "/dir.bin" + sfx, bdir) ||
		    !slurp(dir + "/dif_char.txt" + sfx, bdif) || !slurp(dir + "/ids.bin" + sfx, bids)) return -1;
		DnaReader rr(bref); BitReader dr(bdir);
		size_t pp = 0, dp = 0, ip = 0;
		std::string ref;
		while (pp + 4 <= bpos.size()) {
			uint32_t num; memcpy(&num, bpos.data() + pp, 4); pp += 4;
			ref.clear();
			int pre = 0; uint64_t pre_idx = 0;
			for (uint32_t q = 0; q < num; ++q) {
				if (pp + 2 > bpos.size() || ip + 4 > bids.size()) return -1;
				uint16_t d; memcpy(&d, bpos.data() + pp, 2); pp += 2;
				uint32_t v; memcpy(&v, bids.data() + ip, 4); ip += 4;
				uint64_t id = v;
				if (d == 0) id += pre_idx;                                 // same begin position: a difference (:178-181)
				pre_idx = id;
				const int pos = pre + d; pre = pos;
				while ((int)ref.size() < pos + L) { const int c = rr.next(); if (c < 0) return -1; ref.push_back("ACGT"[c]); }
				const int rev = dr.next();
				size_t e = dp; while (e < bdif.size() && bdif[e] != '\n') ++e;
				if (e >= bdif.size() || !decode_line((const char*)bdif.data() + dp, e - dp, ref.c_str() + pos, L, seq)) return -1;
				dp = e + 1;
				if (rev) revcomp(seq);
				put(id & 0xFFFFFFFFull, seq);
			}
		}
	}
	if (bad) return -1;
	for (uint8_t s : seen) if (!s) return -1;                              // every position filled exactly once
	FILE *out = fopen(out_path, "w");
	if (!out) return -1;
	for (size_t i = 0; i < n_seq; ++i) { fwrite(table.data() + i * (size_t)L, 1, (size_t)L, out); fputc('\n', out); }
	fclose(out);
	if (n_reads) *n_reads = n_seq;
	return 0;
}

// ---- paired end (_PE): decomp_AATTNN_pe (decompress.c:963-1212) + decompress_pe (:780-917) ----------------------------
// Reads of the first file are written to out_path1 in stream order (the eight lists, then the contig members); a read
// of the second file carries the line number of its mate (peids streams, one file bit per read says which kind a
// read is) and goes to that line of out_path2: line i of the two outputs is a pair.
static int decompress_pe_impl(const char *folder, const char *out_path1, const char *out_path2, uint64_t *n_pairs)
{
	if (!folder || !out_path1 || !out_path2) return -1;
	const std::string dir(folder);
	FILE *fi = fopen((dir + "/info.txt").c_str(), "r");
	if (!fi) return -1;
	int L = 0, nth = 0; long half = 0, na = 0, nt = 0, nn = 0;
	if (fscanf(fi, "%d %d %ld %ld %ld %ld", &L, &nth, &half, &na, &nt, &nn) != 6) { fclose(fi); return -1; }
	fclose(fi);
	if (L < 1 || L > 256 || nth < 1 || nth > 4096 || half < 0 || na < 0 || nt < 0 || nn < 0) return -1;
	{   // every read has one bit in a file.bin stream: the number of pairs cannot exceed what they hold
		size_t fbb = file_bytes(dir + "/file.bin.sp");
		for (int th = 0; th < nth; ++th) fbb += file_bytes(dir + "/file.bin." + std::to_string(th));
		if ((size_t)half > fbb * 8) return -1;
	}
	std::vector<char> table((size_t)half * (size_t)L, 0);
	std::vector<uint8_t> seen((size_t)half, 0);
	FILE *out = fopen(out_path1, "w");
	if (!out) return -1;
	uint64_t left = 0; bool bad = false;
	struct Pairing {                                                      // file bits + mate numbers of one stream set
		std::vector<uint8_t> fb, ids; size_t ip = 0; size_t bpos = 0; int k = 8; unsigned cur = 0;
		int bit() { if (k >= 8) { cur = bpos < fb.size() ? fb[bpos++] : 0; k = 0; } const int c = (int)(cur & 1); cur >>= 1; ++k; return c; }
		bool id(uint32_t &v) { if (ip + 4 > ids.size()) return false; memcpy(&v, ids.data() + ip, 4); ip += 4; return true; }
	};
	auto place = [&](Pairing &pr, const std::string &s) {
		if ((int)s.size() != L) { bad = true; return; }
		if (pr.bit()) {                                                    // a read of the second file: to its mate's line
			uint32_t v; if (!pr.id(v) || v >= (uint64_t)half || seen[v]) { bad = true; return; }
			memcpy(table.data() + (size_t)v * L, s.data(), (size_t)L); seen[v] = 1;
		} else { fwrite(s.data(), 1, s.size(), out); fputc('\n', out); ++left; }
	};
	std::vector<uint8_t> buf;
	std::string seq;
	Pairing sp;
	if (!slurp(dir + "/file.bin.sp", sp.fb) || !slurp(dir + "/peids.bin.sp", sp.ids)) { fclose(out); return -1; }
	for (long i = 0; i < na; ++i) place(sp, std::string((size_t)L, 'A'));
	for (long i = 0; i < nt; ++i) place(sp, std::string((size_t)L, 'T'));
	for (long i = 0; i < nn; ++i) place(sp, std::string((size_t)L, 'N'));
	{
		const char bases[3] = {'A', 'T', 'N'}; const char *names[3] = {"AA.txt", "TT.txt", "NN.txt"};
		for (int q = 0; q < 3; ++q) {
			if (!slurp(dir + "/" + names[q], buf)) { fclose(out); return -1; }
			const std::string cref((size_t)L, bases[q]);
			size_t s = 0;
			for (size_t i = 0; i < buf.size(); ++i) if (buf[i] == '\n') { if (!decode_line((const char*)buf.data() + s, i - s, cref.c_str(), L, seq)) { fclose(out); return -1; } s = i + 1; place(sp, seq); }
		}
	}
	if (!slurp(dir + "/single_N.seq", buf)) { fclose(out); return -1; }
	{ size_t s = 0; for (size_t i = 0; i < buf.size(); ++i) if (buf[i] == '\n') { seq.assign((const char*)buf.data() + s, i - s); s = i + 1; place(sp, seq); } }
	if (!slurp(dir + "/single.seq", buf)) { fclose(out); return -1; }
	{
		// the last byte may be padded with up to three A: reads are taken while whole reads remain (4 bases per byte)
		const size_t whole = buf.size() * 4 / (size_t)L;
		DnaReader r(buf);
		for (size_t i = 0; i < whole; ++i) {
			seq.clear();
			for (int j = 0; j < L; ++j) seq.push_back("ACGT"[r.next() & 3]);
			place(sp, seq);
		}
	}
	for (int th = 0; th < nth; ++th) {
		std::vector<uint8_t> bref, bpos, bdir, bdif;
		Pairing pr;
		const std::string sfx = "." + std::to_string(th);
		if (!slurp(dir + "/ref.bin" + sfx, bref) || !slurp(dir + "/beg_pos.bin" + sfx, bpos) || !slurp(dir + "/dir.bin" + sfx, bdir) ||
		    !slurp(dir + "/dif_char.txt" + sfx, bdif) || !slurp(dir + "/file.bin" + sfx, pr.fb) || !slurp(dir + "/peids.bin" + sfx, pr.ids)) { fclose(out); return -1; }
		DnaReader rr(bref); BitReader dr(bdir);
		size_t pp = 0, dp = 0;
		std::string ref;
		while (pp + 4 <= bpos.size()) {
			uint32_t num; memcpy(&num, bpos.data() + pp, 4); pp += 4;
			ref.clear();
			int pre = 0;
			for (uint32_t q = 0; q < num; ++q) {
				if (pp + 2 > bpos.size()) { fclose(out); return -1; }
				uint16_t d; memcpy(&d, bpos.data() + pp, 2); pp += 2;
				const int pos = pre + d; pre = pos;
				while ((int)ref.size() < pos + L) { const int c = rr.next(); if (c < 0) { fclose(out); return -1; } ref.push_back("ACGT"[c]); }
				const int rev = dr.next();
				size_t e = dp; while (e < bdif.size() && bdif[e] != '\n') ++e;
				if (e >= bdif.size() || !decode_line((const char*)bdif.data() + dp, e - dp, ref.c_str() + pos, L, seq)) { fclose(out); return -1; }
				dp = e + 1;
				if (rev) revcomp(seq);
				place(pr, seq);
			}
		}
	}
	fclose(out);
	if (bad || left != (uint64_t)half) return -1;
	for (uint8_t s : seen) if (!s) return -1;
	FILE *out2 = fopen(out_path2, "w");
	if (!out2) return -1;
	for (long i = 0; i < half; ++i) { fwrite(table.data() + (size_t)i * L, 1, (size_t)L, out2); fputc('\n', out2); }
	fclose(out2);
	if (n_pairs) *n_pairs = (uint64_t)half;
	return 0;
}

// no C++ exception crosses the C boundary (a corrupt info.txt can ask for more memory than there is)
extern "C" int mcomh_decompress(const char *folder, const char *out_path, uint64_t *n_reads)
{
	try { return decompress_impl(folder, out_path, n_reads); } catch (...) { return -1; }
}

// no C++ exception crosses the C boundary (a corrupt info.txt can ask for more memory than there is)
extern "C" int mcomh_decompress_order(const char *folder, const char *out_path, uint64_t *n_reads)
{
	try { return decompress_order_impl(folder, out_path, n_reads); } catch (...) { return -1; }
}

// no C++ exception crosses the C boundary (a corrupt info.txt can ask for more memory than there is)
extern "C" int mcomh_decompress_pe(const char *folder, const char *out_path1, const char *out_path2, uint64_t *n_pairs)
{
	try { return decompress_pe_impl(folder, out_path1, out_path2, n_pairs); } catch (...) { return -1; }
}
