// minicom_amd/host/mcom_fastq.hpp -- the single-pass FASTQ parser, shared between mcom_fastq.cpp and the pipeline driver.
#pragma once
#include <stddef.h>
#include <stdint.h>
// ---- the single-pass route (round 4): pread in chunks, check, pack and send in one sweep; rows go to provisional places ----
struct McomFastqStream;
// rows a plain four-line FASTQ file can hold at most (room for the provisional places), *L = its read length (in: 0 or the length it
// must have); 0 = not that layout
size_t mcom_fastq_stream_cap(const char *path, int *L);
// 1 = done: d_packed / d_nmask (room for cap_rows rows each) hold every read at its piece's provisional place; 0 = not this layout after
// all; < 0 = error
int mcom_fastq_stream_packed(const char *path, int L, int device, void *copy_stream, uint64_t *d_packed, uint64_t *d_nmask, size_t cap_rows, McomFastqStream **out);
size_t mcom_fastq_stream_total(const McomFastqStream *st);
int mcom_fastq_stream_len(const McomFastqStream *st);
// the pieces: provisional first row and number of rows of each (arrays of the returned size; NULL: just the size)
size_t mcom_fastq_stream_pieces(const McomFastqStream *st, size_t *prov, size_t *count);
void mcom_fastq_stream_free(McomFastqStream *st);
