/* mcom_test.h -- TEST HOOKS of libmcom_hip.so.  Not part of the drop-in boundary (include/mcom.h): nothing a caller of the
 * reference's functions needs is declared here.  Each hook forces a code path that real inputs reach only at sizes a unit test
 * cannot afford (a sort segment beyond the LDS, an index partition beyond the sorted placement, a consensus unit beyond the
 * bit-sliced counters) or selects between kernels that compute the same result.  Results never depend on any of them.  */
#ifndef MCOM_TEST_H
#define MCOM_TEST_H
#include "mcom.h"
#ifdef __cplusplus
extern "C" {
#endif

/* mcom_sort_group: the size above which a segment takes the nine-pass route instead of the in-LDS sort (0 = default, at most 4096). */
int mcom_set_segment_capacity(mcom_ctx *ctx, uint32_t records);
/* mcom_cindex_build: partitions with more than `entries` entries are placed by the scattered kernel instead of
 * the sorted one (0 = all of them; negative = default).  Same index either way.                                       */
int mcom_set_index_capacity(mcom_ctx *ctx, int entries);
/* Merge consensus: a unit of 32 columns that more than `members` members reach sends its tile to the
 * wave-per-tile kernel (0 = default, the 127 the bit-sliced counters hold).  Same consensus either way.              */
int mcom_set_consensus_capacity(mcom_ctx *ctx, uint32_t members);
/* mcom_sketch_contigs has three kernels: one lane per string (windows up to 64 entries, strings below 32768 characters) with a ring of
 * 32-bit hash prefixes (k odd: ties are settled by recomputing the hashes from the string) or of 64-bit hashes, and one wave per
 * string (also the choice for 8192 strings or fewer); wave_per_string = 1 forces the last, 2 the 64-bit ring, 4 the lane per string
 * with its default ring whatever the number of strings, 0 = the default choice.  Same sketch every way.
 * mcom_set_sketch_prefix_bits (1..30; default 14: prefixes of up to 14 bits live in 16-bit ring words, wider ones in 32-bit words)
 * sets the prefix width of the first: a few bits make ties the rule (tests); wave_per_string = 3 keeps 32-bit words at any width.    */
int mcom_set_sketch_kernel(mcom_ctx *ctx, int wave_per_string);
int mcom_set_sketch_prefix_bits(mcom_ctx *ctx, int bits);
/* mcom_claim_pairs settles all rounds inside ONE launch whose workgroups meet at a grid barrier, which only works while every workgroup
 * is resident; a barrier wait that runs out raises the context's poison flag and the launch-per-round loop (two launches and a host
 * round trip per round: needs no co-residency) redoes the claiming.  route 0 = that default, 1 = the loop at once, 2 = the first barrier
 * of the one-launch kernel gives up at once (the flag trips, the loop takes over), 3 = the one-launch kernel without its tail (every round
 * by the whole grid, also when a few edges are left; by default one workgroup finishes a list of at most 16384 live edges on its own).
 * Same jobs and flags every way.
 * mcom_claim_fallbacks: how often the loop has run in this context.                                                               */
int mcom_set_claim_route(mcom_ctx *ctx, int route);
int mcom_claim_fallbacks(const mcom_ctx *ctx);
/* mcom_dicts_screen counts the singletons' (dictionary, key) pairs in hashed counters.  route 0 = the default: the counter numbers are
 * binned by counter range (a region per workgroup and bin, no global atomics) and every bin is counted in LDS; a set whose keys pile up in
 * one region (copies of one read) overflows it and goes through the global-atomics kernel instead.  1 = that kernel at once, 2 = regions of
 * a few keys, so that any input overflows (the fall-back runs).  Same answer every way.  mcom_screen_fallbacks: how often it ran.       */
int mcom_set_screen_route(mcom_ctx *ctx, int route);
int mcom_screen_fallbacks(const mcom_ctx *ctx);
/* mcom_realign_pass_reads / _tuples over a SHARE of the keys (several GPUs: geom names more than one share).  route 0 = the default: a
 * workgroup lists the (singleton, pair) tasks whose keys this share owns on a stack in LDS and a thread takes one task (k_realign_owned);
 * 1 = the kernel of the whole index with several pairs per lane, a lane dropping the pairs it does not own.  Same claims either way.  */
int mcom_set_lookup_route(mcom_ctx *ctx, int route);


/* ---- libmcom_host.so ---- */
struct mcomh_pipeline;
/* Multi-GPU failure protocol (minicom_amd/host/mcom_pipeline.cpp, "Failure protocol"): rank-local errors travel with a flag exchange in
 * front of every collective so that no rank is left waiting for one that has returned.  mcomh_test_inject_failure makes this rank fail
 * right before its k-th flag exchange (k = 1 ...), as if the local work in front of it had failed; mcomh_test_flag_exchanges says how
 * many a finished run made.  tests/test_gpu_distributed.py lets one rank of three fail at points spread over every stage: all three must
 * return an error, none may hang.                                                                                                   */
int  mcomh_test_inject_failure(struct mcomh_pipeline *p, long k);
long mcomh_test_flag_exchanges(const struct mcomh_pipeline *p);
/* Reads of another class than 0 reach the host as a list made on the device (mcom_special_reads): `first` entries travel with the
 * count, up to `cap` in a second copy, and beyond `cap` the class array itself is copied.  Defaults 4096 and 2^20; small values let
 * a test walk all three paths with a few dozen reads.  Before mcomh_kt_for_reads.                                              */
int  mcomh_test_special_capacity(struct mcomh_pipeline *p, uint32_t first, uint32_t cap);
/* A gzip file of several members is inflated and parsed by all cores (host/mcom_fastq_gz.cpp); whatever that route does not take goes to
 * the sequential reader, with the same rows.  Work items (groups of members) of the last file the parallel route read to its end, 0 when
 * the last file went to the sequential reader: lets a test tell which of the two it has checked.                                    */
long mcomh_test_gz_items(void);
/* The member-parallel route decodes with its own DEFLATE decoder (host/mcom_inflate.cpp), not zlib: one gzip member at in[0 .. in_n) into
 * out[0 .. out_cap).  0 = decoded (*in_used bytes were the member, *out_n bytes came out, CRC-32 and ISIZE checked), 1 = out is too
 * small, < 0 = truncated (-1) / not a valid member (-2).  For tests that hold it against zlib.                                        */
int  mcomh_test_gunzip(const uint8_t *in, size_t in_n, uint8_t *out, size_t out_cap, size_t *in_used, size_t *out_n);

#ifdef __cplusplus
}
#endif
#endif
