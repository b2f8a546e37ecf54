// minicom_amd/host/mcom_inflate.cpp -- DEFLATE (RFC 1951) / gzip member (RFC 1952) decoder for the parallel .fastq.gz ingest (round 5).
//
// The reference reads its input through zlib's gzread (bseq.c:19-36, kseq.h).  With the members of a file shared out over all cores
// (mcom_fastq_gz.cpp) zlib's inflate itself became the whole of the ingest time (~230 MB/s of text per core), so the member-parallel
// route decodes with this file instead: a decoder written for whole-buffer work -- the complete member in memory, the complete
// output buffer in front of it -- which is what lets it drop what zlib's streaming interface pays for:
//   * a 64-bit bit buffer refilled by one unaligned 8-byte load (no per-byte loop), valid for up to three literals, or a length
//     with its extra bits, between refills;
//   * one table look-up per symbol -- or per TWO literals when both codes fit the 11 index bits, as the two-to-four-bit codes of bases and
//     binned qualities do: 11 bits of the stream index the literal/length table (8 for distances), an entry carries the symbol's value or
//     base, its extra-bit count and its code length; longer codes go through a second-level table;
//   * matches copied eight bytes at a time (distance 1 -- a run of one quality character -- as a fill), literals stored as they
//     are decoded, no sliding window: the output buffer is the window;
//   * bounds are checked per loop iteration against margins, the last bytes of either buffer go through a careful loop.
// CRC-32 and ISIZE of the trailer are verified (carry-less multiplication where the CPU has it, slicing by 8 otherwise).  Any malformed input gives an error code, never a fault: every table
// index is masked, every distance checked against the bytes written, every copy against the end of the buffer.
// Tests: tests/test_fastq.py (python's zlib as the checker: all levels, stored / fixed / dynamic blocks, random and corrupt input).
#include "mcom_inflate.hpp"
#include <algorithm>
#include <cstring>
#include <new>
#include <immintrin.h>

namespace {
typedef uint8_t u8; typedef uint16_t u16; typedef uint32_t u32; typedef uint64_t u64;

inline u64 load64(const u8 *p) { u64 v; memcpy(&v, p, 8); return v; }                // (little-endian host: x86-64)
inline void store64(u8 *p, u64 v) { memcpy(p, &v, 8); }

// ---- table entries ---------------------------------------------------------------------------------------------------------------------
// bits 0-7 code length (bits to take from the stream for the code itself), bits 8-10 kind, bits 12-15 extra bits (kind LEN / DIST) or
// second-level index bits (kind SUB), bits 16-31 literal / base value / offset of the second-level table
// Literal entries of the literal/length table carry bit 11 (LITF) and may hold TWO literals (bit 8 set: the second one's code follows the
// first one's inside the 11 index bits -- bases and binned qualities have codes of two to four bits): value = first | second << 8, code
// length = both codes together.
enum { K_LIT = 0, K_LEN = 1, K_EOB = 2, K_SUB = 3, K_BAD = 4 };
const uint32_t LITF = 0x800u, LIT2 = 0x100u;
inline u32 mk(u32 bits, u32 kind, u32 extra, u32 val) { return bits | (kind << 8) | (extra << 12) | (val << 16); }
inline u32 e_bits(u32 e) { return e & 255u; }
inline u32 e_kind(u32 e) { return (e >> 8) & 7u; }
inline u32 e_extra(u32 e) { return (e >> 12) & 15u; }
inline u32 e_val(u32 e) { return e >> 16; }

const int LL_BITS = 11, D_BITS = 8;
const int LL_SIZE = (1 << LL_BITS) + 288 * 16, D_SIZE = (1 << D_BITS) + 32 * 128;
const u16 len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const u8 len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const u16 dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const u8 dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline u32 ll_entry(int sym, u32 bits)
{
	if (sym < 256) return mk(bits, K_LIT, 0, (u32)sym) | LITF;
	if (sym == 256) return mk(bits, K_EOB, 0, 0);
	if (sym > 285) return mk(bits, K_BAD, 0, 0);
	return mk(bits, K_LEN, len_extra[sym - 257], len_base[sym - 257]);
}
inline u32 d_entry(int sym, u32 bits) { return sym < 30 ? mk(bits, K_LEN, dist_extra[sym], dist_base[sym]) : mk(bits, K_BAD, 0, 0); }

inline u32 rev_bits(u32 c, int n) { u32 r = 0; for (int i = 0; i < n; ++i) { r = (r << 1) | (c & 1); c >>= 1; } return r; }

// the decoding table of a canonical code given by its lengths; false: over-subscribed.  Incomplete codes are accepted (a distance code of
// one symbol is legal and zlib writes it); their unused patterns decode to K_BAD.
template <class ENTRY>
bool build_table(const u8 *lens, int n, u32 *table, int P, int table_size, ENTRY entry)
{
	int count[16] = {0};
	for (int i = 0; i < n; ++i) ++count[lens[i]];
	count[0] = 0;
	u32 left = 1;                                                              // Kraft
	for (int l = 1; l <= 15; ++l) { left <<= 1; if ((u32)count[l] > left) return false; left -= (u32)count[l]; }
	u32 next[16]; u32 code = 0;
	for (int l = 1; l <= 15; ++l) { code = (code + (u32)count[l - 1]) << 1; next[l] = code; }
	const u32 bad = mk(1, K_BAD, 0, 0);
	for (int i = 0; i < (1 << P); ++i) table[i] = bad;
	// codes of at most P bits fill the first level; for the longer ones: the longest code behind every P-bit prefix sizes its second level
	u8 sub_bits[1 << LL_BITS];
	memset(sub_bits, 0, (size_t)1 << P);
	u32 nx[16]; memcpy(nx, next, sizeof nx);
	for (int s = 0; s < n; ++s) {
		const int l = lens[s];
		if (l <= P) { if (l) ++nx[l]; continue; }
		const u32 c = nx[l]++;
		const u32 pre = rev_bits(c >> (l - P), P);                             // the first P bits of the code, as the stream shows them
		if ((int)sub_bits[pre] < l - P) sub_bits[pre] = (u8)(l - P);
	}
	int used = 1 << P;
	u32 sub_at[1 << LL_BITS];
	for (int pre = 0; pre < (1 << P); ++pre) {
		if (!sub_bits[pre]) continue;
		const int sz = 1 << sub_bits[pre];
		if (used + sz > table_size) return false;
		sub_at[pre] = (u32)used;
		for (int i = 0; i < sz; ++i) table[used + i] = bad;
		table[pre] = mk((u32)P, K_SUB, sub_bits[pre], (u32)used);
		used += sz;
	}
	for (int s = 0; s < n; ++s) {
		const int l = lens[s];
		if (!l) continue;
		const u32 c = next[l]++;
		const u32 r = rev_bits(c, l);
		const u32 e = entry(s, (u32)l);
		if (l <= P) { for (u32 i = r; i < (1u << P); i += 1u << l) table[i] = e; }
		else {
			const u32 pre = r & ((1u << P) - 1), sb = sub_bits[pre];
			for (u32 i = r >> P; i < (1u << sb); i += 1u << (l - P)) table[sub_at[pre] + i] = e;
		}
	}
	return true;
}

struct Tables { u32 ll[LL_SIZE]; u32 d[D_SIZE]; };

// first-level literal entries whose remaining index bits hold a second whole literal become entries of two (from the top down: the entry
// looked at for the second literal has a smaller index and is still single)
void pair_literals(u32 *t)
{
	for (int i = (1 << LL_BITS) - 1; i >= 0; --i) {
		const u32 e = t[i];
		if (!(e & LITF)) continue;
		const u32 l1 = e_bits(e);
		if (l1 >= (u32)LL_BITS) continue;
		const u32 e2 = t[(u32)i >> l1];
		if (!(e2 & LITF) || e_bits(e2) > (u32)LL_BITS - l1) continue;
		t[i] = (l1 + e_bits(e2)) | LITF | LIT2 | ((e_val(e) | (e_val(e2) << 8)) << 16);
	}
}

struct Fixed { Tables t; Fixed() {
	u8 l[288 + 32];
	for (int i = 0; i < 144; ++i) l[i] = 8;
	for (int i = 144; i < 256; ++i) l[i] = 9;
	for (int i = 256; i < 280; ++i) l[i] = 7;
	for (int i = 280; i < 288; ++i) l[i] = 8;
	for (int i = 0; i < 32; ++i) l[288 + i] = 5;
	build_table(l, 288, t.ll, LL_BITS, LL_SIZE, ll_entry);
	pair_literals(t.ll);
	build_table(l + 288, 32, t.d, D_BITS, D_SIZE, d_entry);
} };
const Tables &fixed_tables() { static const Fixed f; return f.t; }

// ---- CRC-32 (slicing by 8) -------------------------------------------------------------------------------------------------------------
struct Crc { u32 t[8][256]; Crc() {
	for (u32 i = 0; i < 256; ++i) { u32 c = i; for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u))); t[0][i] = c; }
	for (u32 i = 0; i < 256; ++i) for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 255u];
} };
const Crc &crc_tables() { static const Crc c; return c; }
}  // namespace

// the same by carry-less multiplication (Gopal et al., "Fast CRC computation for generic polynomials using PCLMULQDQ", Intel 2009): four
// 128-bit lanes folded over 64 bytes a step, then 4 -> 1, 128 -> 64 bits and a Barrett reduction; constants of the reflected CRC-32
// polynomial.  Works on the register value (already inverted), on a multiple of 16 bytes, at least 64.
__attribute__((target("pclmul,sse4.1"))) static u32 crc32_clmul(u32 c, const u8 *p, size_t n)
{
	const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll), k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll),
	              k5 = _mm_set_epi64x(0, 0x0163cd6124ll), poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
	__m128i x1 = _mm_loadu_si128((const __m128i*)p), x2 = _mm_loadu_si128((const __m128i*)(p + 16)),
	        x3 = _mm_loadu_si128((const __m128i*)(p + 32)), x4 = _mm_loadu_si128((const __m128i*)(p + 48)), x5, x6, x7, x8;
	x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)c));
	p += 64; n -= 64;
	while (n >= 64) {
		x5 = _mm_clmulepi64_si128(x1, k1k2, 0x00); x6 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
		x7 = _mm_clmulepi64_si128(x3, k1k2, 0x00); x8 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
		x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11); x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
		x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11); x4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
		x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), _mm_loadu_si128((const __m128i*)p));
		x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), _mm_loadu_si128((const __m128i*)(p + 16)));
		x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), _mm_loadu_si128((const __m128i*)(p + 32)));
		x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), _mm_loadu_si128((const __m128i*)(p + 48)));
		p += 64; n -= 64;
	}
	x5 = _mm_clmulepi64_si128(x1, k3k4, 0x00); x1 = _mm_clmulepi64_si128(x1, k3k4, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
	x5 = _mm_clmulepi64_si128(x1, k3k4, 0x00); x1 = _mm_clmulepi64_si128(x1, k3k4, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
	x5 = _mm_clmulepi64_si128(x1, k3k4, 0x00); x1 = _mm_clmulepi64_si128(x1, k3k4, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
	while (n >= 16) {
		x2 = _mm_loadu_si128((const __m128i*)p);
		x5 = _mm_clmulepi64_si128(x1, k3k4, 0x00); x1 = _mm_clmulepi64_si128(x1, k3k4, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
		p += 16; n -= 16;
	}
	const __m128i m32 = _mm_setr_epi32(~0, 0, ~0, 0);
	x2 = _mm_clmulepi64_si128(x1, k3k4, 0x10);
	x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), x2);
	x2 = _mm_srli_si128(x1, 4);
	x1 = _mm_clmulepi64_si128(_mm_and_si128(x1, m32), k5, 0x00);
	x1 = _mm_xor_si128(x1, x2);
	x2 = _mm_clmulepi64_si128(_mm_and_si128(x1, m32), poly, 0x10);
	x2 = _mm_clmulepi64_si128(_mm_and_si128(x2, m32), poly, 0x00);
	x1 = _mm_xor_si128(x1, x2);
	return (u32)_mm_extract_epi32(x1, 1);
}

int mcom_crc32_tables_only = 0;                                                // (tests: the table form alone)
uint32_t mcom_crc32(uint32_t crc, const uint8_t *p, size_t n)
{
	const Crc &T = crc_tables();
	u32 c = ~crc;
	static const bool clmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
	if (clmul && n >= 64 && !mcom_crc32_tables_only) { const size_t k = n & ~(size_t)15; c = crc32_clmul(c, p, k); p += k; n -= k; }
	while (n && ((uintptr_t)p & 7)) { c = (c >> 8) ^ T.t[0][(c ^ *p++) & 255u]; --n; }
	while (n >= 8) {
		const u64 v = load64(p) ^ c;
		c = T.t[7][v & 255u] ^ T.t[6][(v >> 8) & 255u] ^ T.t[5][(v >> 16) & 255u] ^ T.t[4][(v >> 24) & 255u] ^
		    T.t[3][(v >> 32) & 255u] ^ T.t[2][(v >> 40) & 255u] ^ T.t[1][(v >> 48) & 255u] ^ T.t[0][v >> 56];
		p += 8; n -= 8;
	}
	while (n--) c = (c >> 8) ^ T.t[0][(c ^ *p++) & 255u];
	return ~c;
}

// The decoder proper, resumable at the output side: the whole input is there (a mapped file, a member in memory), the output comes in
// pieces.  mcom_inflate_run decodes into out[0 .. out_cap) until the stream ends (MCOM_INFLATE_OK) or the piece is full (MCOM_INFLATE_ROOM:
// call again with the next piece); `hist` bytes in front of out are earlier output of the same stream that matches may reach back into
// (a caller that changes buffers copies the last 32 KB in front of the new one).  A piece may end anywhere, also inside a match's room:
// a symbol that does not fit is not started.
void mcom_inflate_begin(mcom_inflate_stream *s, const uint8_t *in, size_t in_n)
{
	memset(s, 0, sizeof *s);
	s->in = in; s->in_end = in + in_n;
}
void mcom_inflate_end(mcom_inflate_stream *s) { delete (Tables*)s->dyn; s->dyn = nullptr; }

// (compiled twice: with BMI2 -- shifts by a register without the detour through CL, masks in one instruction -- for the CPUs that have it, which
// is every x86-64 a GPU server is built around; the loader picks: +5-8 % on FASTQ)
__attribute__((target_clones("bmi2", "default")))
int mcom_inflate_run(mcom_inflate_stream *S, uint8_t *out0, size_t out_cap, size_t hist, size_t *out_n)
{
	const u8 *in = S->in, *const in_end = S->in_end;
	u8 *out = out0, *const out_end = out0 + out_cap;
	u64 bb = S->bb; u32 bl = S->bl;                                               // bit buffer: bl valid bits at the bottom of bb, nothing above them
	Tables *dyn = (Tables*)S->dyn;
	int last = S->last;
	int phase = S->phase;                                                         // 0 in front of a block header, 1 inside a Huffman block, 2 inside a stored block
	const Tables *T = (const Tables*)S->tables;
	u32 stored_left = S->stored_left;
	// (errors leave the stream where it is: it is not continued.  A macro, not a lambda: a closure over in / bb / bl by reference costs the loops their registers)
#define MCOM_LEAVE(code) do { S->in = in; S->bb = bb; S->bl = bl; S->dyn = dyn; S->last = last; S->phase = phase; S->tables = T; S->stored_left = stored_left; \
                              *out_n = (size_t)(out - out0); return (code); } while (0)
#define fill() do { while (bl < 56 && in < in_end) { bb |= (u64)*in++ << bl; bl += 8; } } while (0)   /* (bl stays below 64) */
	for (;;) {
		if (phase == 2) {                                                          // the bytes of a stored block
			const size_t k = std::min<size_t>(stored_left, (size_t)(out_end - out));
			memcpy(out, in, k); in += k; out += k; stored_left -= (u32)k;
			if (stored_left) MCOM_LEAVE(MCOM_INFLATE_ROOM);
			phase = 0;
		}
		if (phase == 0) {
		if (last) break;
		fill();
		if (bl < 3) return MCOM_INFLATE_TRUNCATED;
		last = (int)(bb & 1); const u32 type = (u32)(bb >> 1) & 3u;
		bb >>= 3; bl -= 3;
		if (type == 0) {                                                           // stored: to the byte boundary, LEN, ~LEN, the bytes
			in -= bl >> 3; bb = 0; bl = 0;                                         // (whole bytes in the buffer go back; the rest of the current byte is dropped)
			if (in_end - in < 4) return MCOM_INFLATE_TRUNCATED;
			const u32 len = (u32)in[0] | ((u32)in[1] << 8), nlen = (u32)in[2] | ((u32)in[3] << 8);
			if ((len ^ nlen) != 0xFFFFu) return MCOM_INFLATE_CORRUPT;
			in += 4;
			if ((size_t)(in_end - in) < len) return MCOM_INFLATE_TRUNCATED;
			stored_left = len; phase = 2;
			continue;
		}
		if (type == 3) return MCOM_INFLATE_CORRUPT;
		if (type == 1) T = &fixed_tables();
		else {
			fill();
			if (bl < 14) return MCOM_INFLATE_TRUNCATED;
			const int hlit = (int)(bb & 31) + 257, hdist = (int)((bb >> 5) & 31) + 1, hclen = (int)((bb >> 10) & 15) + 4;
			bb >>= 14; bl -= 14;
			if (hlit > 286 || hdist > 30) return MCOM_INFLATE_CORRUPT;
			static const u8 order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
			u8 cl[19] = {0};
			for (int i = 0; i < hclen; ++i) { fill(); if (bl < 3) return MCOM_INFLATE_TRUNCATED; cl[order[i]] = (u8)(bb & 7); bb >>= 3; bl -= 3; }
			u32 ct[128];
			auto cl_entry = [](int sym, u32 bits) { return mk(bits, K_LIT, 0, (u32)sym); };
			if (!build_table(cl, 19, ct, 7, 128, cl_entry)) return MCOM_INFLATE_CORRUPT;
			u8 lens[286 + 30 + 140];
			int at = 0; const int total = hlit + hdist;
			while (at < total) {
				fill();
				const u32 e = ct[bb & 127u];
				if (e_kind(e) != K_LIT || e_bits(e) > bl) return bl < 7 && in == in_end ? MCOM_INFLATE_TRUNCATED : MCOM_INFLATE_CORRUPT;
				bb >>= e_bits(e); bl -= e_bits(e);
				const u32 s = e_val(e);
				if (s < 16) { lens[at++] = (u8)s; continue; }
				u32 rep, xb; u8 v = 0;
				if (s == 16) { if (!at) return MCOM_INFLATE_CORRUPT; v = lens[at - 1]; xb = 2; rep = 3; }
				else if (s == 17) { xb = 3; rep = 3; }
				else { xb = 7; rep = 11; }
				if (bl < xb) return MCOM_INFLATE_TRUNCATED;
				rep += (u32)bb & ((1u << xb) - 1); bb >>= xb; bl -= xb;
				if (at + (int)rep > total) return MCOM_INFLATE_CORRUPT;
				memset(lens + at, v, rep); at += (int)rep;
			}
			if (lens[256] == 0) return MCOM_INFLATE_CORRUPT;                       // no end-of-block code
			if (!dyn) { dyn = new (std::nothrow) Tables; S->dyn = dyn; }          // (the stream owns it from here: mcom_inflate_end)
			if (!dyn) return MCOM_INFLATE_NOMEM;
			if (!build_table(lens, hlit, dyn->ll, LL_BITS, LL_SIZE, ll_entry) || !build_table(lens + hlit, hdist, dyn->d, D_BITS, D_SIZE, d_entry)) return MCOM_INFLATE_CORRUPT;
			pair_literals(dyn->ll);
			T = dyn;
		}
		phase = 1;
		}
		const u32 *const lt = T->ll, *const dt = T->d;
		const u32 LM = (1u << LL_BITS) - 1, DM = (1u << D_BITS) - 1;
		bool eob = false;
		// ---- the fast loop: at least 32 bytes of input and 3 + 258 + 8 bytes of room in front of it ------------------------------------
		if (in_end - in >= 32 && out_end - out >= 300) {
			const u8 *const in_fast = in_end - 32; u8 *const out_fast = out_end - 300;
#define MCOM_REFILL() do { bb |= load64(in) << bl; in += (63 - bl) >> 3; bl |= 56; } while (0)
			// (a refill may leave bits of the next byte above bl: they are that byte's own low bits and the next refill, or the careful
			// loop's byte-wise one, puts the same bits in the same places)
			// The entry of the NEXT symbol is looked up as soon as the bits in front of it are known -- behind the last literal, behind a
			// match's distance and in front of its copy -- so that the table's latency passes while bytes are stored.
#define MCOM_LITS() do { bb >>= e_bits(e); bl -= e_bits(e); const u16 two = (u16)(e >> 16); memcpy(out, &two, 2); out += 1 + ((e >> 8) & 1u); } while (0)
			MCOM_REFILL();
			u32 e = lt[bb & LM];
			while (in <= in_fast && out <= out_fast) {                               // (here: at least 56 bits in the buffer, e belongs to them)
				if (e & LITF) {                                                      // up to three look-ups of one or two literals (<= 11 bits each) on one refill
					MCOM_LITS();
					e = lt[bb & LM];
					if (e & LITF) {
						MCOM_LITS();
						e = lt[bb & LM];
						if (e & LITF) { MCOM_LITS(); MCOM_REFILL(); e = lt[bb & LM]; continue; }
					}
				}
				if (e_kind(e) == K_SUB) {
					e = lt[e_val(e) + ((u32)(bb >> LL_BITS) & ((1u << e_extra(e)) - 1))];
					if (e & LITF) { MCOM_LITS(); MCOM_REFILL(); e = lt[bb & LM]; continue; }
				}
				if (e_kind(e) != K_LEN) {
					if (e_kind(e) == K_EOB) { bb >>= e_bits(e); bl -= e_bits(e); eob = true; break; }
					return MCOM_INFLATE_CORRUPT;
				}
				bb >>= e_bits(e); bl -= e_bits(e);                                   // (22 + 15 + 5 bits at most since the refill)
				const u32 xb = e_extra(e);
				const u32 length = e_val(e) + ((u32)bb & ((1u << xb) - 1));
				bb >>= xb; bl -= xb;
				if (bl < 32) MCOM_REFILL();                                          // (a match right behind a match has bits enough left for its distance: 15 + 13)
				u32 d = dt[bb & DM];
				if (e_kind(d) == K_SUB) d = dt[e_val(d) + ((u32)(bb >> D_BITS) & ((1u << e_extra(d)) - 1))];
				if (e_kind(d) != K_LEN) return MCOM_INFLATE_CORRUPT;
				bb >>= e_bits(d); bl -= e_bits(d);
				const u32 dxb = e_extra(d);
				const u32 dist = e_val(d) + ((u32)bb & ((1u << dxb) - 1));
				bb >>= dxb; bl -= dxb;
				MCOM_REFILL();
				e = lt[bb & LM];
				if (dist > hist + (size_t)(out - out0)) return MCOM_INFLATE_CORRUPT;
				const u8 *src = out - dist; u8 *const end = out + length;
				if (dist >= 8) {
					store64(out, load64(src)); store64(out + 8, load64(src + 8));
					if (length > 16) { out += 16; src += 16; do { store64(out, load64(src)); out += 8; src += 8; } while (out < end); }
				} else if (dist == 1) {
					const u64 v = 0x0101010101010101ull * *src;
					do { store64(out, v); out += 8; } while (out < end);
				} else {
					do { *out++ = *src++; } while (out < end);                          // (a pattern of 2 - 7 bytes: byte by byte)
				}
				out = end;
			}
#undef MCOM_LITS
#undef MCOM_REFILL
			bb &= bl < 64 ? (((u64)1 << bl) - 1) : ~(u64)0;                          // (drop the bits a refill left above bl)
		}
		// ---- the careful loop: the same decoding with every step checked ----------------------------------------------------------------
		while (!eob) {
			fill();
			const u8 *const in_s = in; const u64 bb_s = bb; const u32 bl_s = bl;  // (a symbol that does not fit the piece is taken back whole)
			u32 e = lt[bb & LM];
			if (e_kind(e) == K_SUB) e = lt[e_val(e) + ((u32)(bb >> LL_BITS) & ((1u << e_extra(e)) - 1))];
			if (!(e & LITF) && e_kind(e) == K_BAD) return bl < 15 && in == in_end ? MCOM_INFLATE_TRUNCATED : MCOM_INFLATE_CORRUPT;
			if (e_bits(e) > bl) return MCOM_INFLATE_TRUNCATED;
			bb >>= e_bits(e); bl -= e_bits(e);
			if (e & LITF) {
				const size_t k = 1 + ((e >> 8) & 1u);
				if ((size_t)(out_end - out) < k) { in = in_s; bb = bb_s; bl = bl_s; MCOM_LEAVE(MCOM_INFLATE_ROOM); }
				*out++ = (u8)e_val(e); if (k == 2) *out++ = (u8)(e >> 24);
				continue;
			}
			if (e_kind(e) == K_EOB) { eob = true; break; }
			const u32 xb = e_extra(e);
			if (bl < xb) return MCOM_INFLATE_TRUNCATED;
			const u32 length = e_val(e) + ((u32)bb & ((1u << xb) - 1));
			bb >>= xb; bl -= xb;
			fill();
			u32 d = dt[bb & DM];
			if (e_kind(d) == K_SUB) d = dt[e_val(d) + ((u32)(bb >> D_BITS) & ((1u << e_extra(d)) - 1))];
			if (e_kind(d) != K_LEN) return bl < 15 && in == in_end ? MCOM_INFLATE_TRUNCATED : MCOM_INFLATE_CORRUPT;
			if (e_bits(d) + e_extra(d) > bl) return MCOM_INFLATE_TRUNCATED;
			bb >>= e_bits(d); bl -= e_bits(d);
			const u32 dxb = e_extra(d);
			const u32 dist = e_val(d) + ((u32)bb & ((1u << dxb) - 1));
			bb >>= dxb; bl -= dxb;
			if (dist > hist + (size_t)(out - out0)) return MCOM_INFLATE_CORRUPT;
			if ((size_t)(out_end - out) < length) { in = in_s; bb = bb_s; bl = bl_s; MCOM_LEAVE(MCOM_INFLATE_ROOM); }
			const u8 *src = out - dist;
			for (u32 i = 0; i < length; ++i) out[i] = src[i];
			out += length;
		}
		phase = 0;                                                                 // the block's end code: the next header, or the end
	}
	in -= bl >> 3; bb = 0; bl = 0;                                                 // whole bytes still in the buffer were never part of the stream
	MCOM_LEAVE(MCOM_INFLATE_OK);
#undef MCOM_LEAVE
#undef fill
}

// One raw deflate stream at in[0 .. in_n) into out[0 .. out_cap), all at once.  MCOM_INFLATE_OK: *in_used bytes were the stream (it ends on
// a byte boundary behind its last block), *out_n bytes came out.  MCOM_INFLATE_ROOM: out is too small (nothing is kept).
int mcom_inflate_raw(const uint8_t *in0, size_t in_n, uint8_t *out0, size_t out_cap, size_t *in_used, size_t *out_n)
{
	mcom_inflate_stream st;
	mcom_inflate_begin(&st, in0, in_n);
	const int rc = mcom_inflate_run(&st, out0, out_cap, 0, out_n);
	*in_used = (size_t)(st.in - in0);
	mcom_inflate_end(&st);
	return rc;
}

// One gzip member (header, deflate stream, CRC-32 + ISIZE) at in[0 .. in_n).
int mcom_gunzip_member(const uint8_t *in, size_t in_n, uint8_t *out, size_t out_cap, size_t *in_used, size_t *out_n)
{
	if (in_n < 18) return MCOM_INFLATE_TRUNCATED;
	if (in[0] != 0x1f || in[1] != 0x8b || in[2] != 8 || (in[3] & 0xE0)) return MCOM_INFLATE_CORRUPT;
	const u32 flg = in[3];
	size_t p = 10;
	if (flg & 4) { if (p + 2 > in_n) return MCOM_INFLATE_TRUNCATED; const size_t xlen = (size_t)in[p] | ((size_t)in[p + 1] << 8); p += 2 + xlen; }
	if (flg & 8) { while (p < in_n && in[p]) ++p; ++p; }
	if (flg & 16) { while (p < in_n && in[p]) ++p; ++p; }
	if (flg & 2) p += 2;
	if (p >= in_n) return MCOM_INFLATE_TRUNCATED;
	size_t used = 0, n = 0;
	const int rc = mcom_inflate_raw(in + p, in_n - p, out, out_cap, &used, &n);
	if (rc) return rc;
	p += used;
	if (p + 8 > in_n) return MCOM_INFLATE_TRUNCATED;
	const u32 crc = (u32)in[p] | ((u32)in[p + 1] << 8) | ((u32)in[p + 2] << 16) | ((u32)in[p + 3] << 24);
	const u32 isz = (u32)in[p + 4] | ((u32)in[p + 5] << 8) | ((u32)in[p + 6] << 16) | ((u32)in[p + 7] << 24);
	if (isz != (u32)n || crc != mcom_crc32(0, out, n)) return MCOM_INFLATE_CORRUPT;
	*in_used = p + 8; *out_n = n;
	return MCOM_INFLATE_OK;
}

// test hook (include/mcom_test.h)
extern "C" int mcomh_test_gunzip(const uint8_t *in, size_t in_n, uint8_t *out, size_t out_cap, size_t *in_used, size_t *out_n)
{
	size_t u = 0, n = 0;
	const int rc = mcom_gunzip_member(in, in_n, out, out_cap, &u, &n);
	if (in_used) *in_used = u;
	if (out_n) *out_n = n;
	return rc;
}
