// minicom_amd/host/mcom_inflate.hpp -- whole-buffer DEFLATE / gzip-member decoder of the parallel .fastq.gz ingest (mcom_inflate.cpp)
#pragma once
#include <stddef.h>
#include <stdint.h>

enum {
	MCOM_INFLATE_OK = 0,
	MCOM_INFLATE_ROOM = 1,            // the output buffer is too small: call again with a larger one (nothing is kept)
	MCOM_INFLATE_TRUNCATED = -1,      // the input ends inside the stream
	MCOM_INFLATE_CORRUPT = -2,        // not a deflate stream / not a gzip member / CRC-32 or ISIZE differ
	MCOM_INFLATE_NOMEM = -3
};
// the decoder with its output in pieces (mcom_inflate.cpp): begin, run until it says OK, end
struct mcom_inflate_stream {
	const uint8_t *in, *in_end;        // what is left of the input
	uint64_t bb; uint32_t bl;          // bit buffer
	int last, phase; uint32_t stored_left;
	const void *tables; void *dyn;     // the current block's decoding tables; the stream's own (dynamic blocks)
};
void mcom_inflate_begin(mcom_inflate_stream *s, const uint8_t *in, size_t in_n);
int  mcom_inflate_run(mcom_inflate_stream *s, uint8_t *out, size_t out_cap, size_t hist, size_t *out_n);
void mcom_inflate_end(mcom_inflate_stream *s);
int mcom_inflate_raw(const uint8_t *in, size_t in_n, uint8_t *out, size_t out_cap, size_t *in_used, size_t *out_n);
int mcom_gunzip_member(const uint8_t *in, size_t in_n, uint8_t *out, size_t out_cap, size_t *in_used, size_t *out_n);
uint32_t mcom_crc32(uint32_t crc, const uint8_t *p, size_t n);
extern int mcom_crc32_tables_only;
