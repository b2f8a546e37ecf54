/* include/mcom.h -- C ABI of the MI355X-native minicom hot path (libmcom_hip.so).
 *
 * Drop-in boundary: these entry points are what the reference's L2 stage workers (kthread_*.c)
 * would call in place of their per-item L1 functions; each one cites the reference interface it
 * replaces (file:line into yuansliu/minicom src/).  The reference-side stubs are in INTEGRATION.md.
 *
 * Conventions
 *   - plain C, no C++/torch types.  All d_* pointers are DEVICE (HBM) pointers owned by the caller.
 *   - every call is asynchronous on the context's HIP stream unless it returns host values;
 *     mcom_sync() waits.  One context per host thread / per GPU.
 *   - return 0 on success, a negative mcom_status otherwise; mcom_last_error() gives the text.
 *     (the reference asserts / exit(1)s instead: sketch.c:122,248, bseq.c:54-57)
 *   - records are the reference's mm128_t: { x = hash, y = id<<32 | pos<<1 | strand } (minicom.h:17-19)
 *   - contig ids: the reference packs (index in the thread's list << 8) + tid into the 32-bit id of a contig's records
 *     (kthread_bucket.c:458, kthread_cb.c:232), which ends at 2^24 contigs per list; here the id IS the contig index
 *     (32 bits: a 500 M-read job makes ~40 M first-round contigs).  A stage dump shifts by MCOM_REF_CONTIG_ID_SHIFT
 *     when it prints records for comparison with the reference.
 *
 * Packed read format ("packed rows"): W = ceil(2L/64) little-endian 64-bit words per read, base i in
 * bits [2i, 2i+1], A=0 C=1 G=2 T=3 (sketch.c:8-25), unused high bits zero.  This is also the byte
 * layout of the reference's single.seq / ref.bin streams (breads.h:232-239).  The reference's Stage-2
 * bitset (bbhashdict.c:69-74) has the two bits of every base swapped (C=2, G=1); XOR-popcounts and
 * key equality are invariant under that swap, see DESIGN.md.
 */
#ifndef MCOM_H
#define MCOM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t x, y; } mcom_mm128;
#define MCOM_REF_CONTIG_ID_SHIFT 8
typedef struct mcom_ctx mcom_ctx;

enum mcom_status {
	MCOM_OK = 0,
	MCOM_E_ARG = -1,      /* bad argument (k out of 1..31, L out of 1..256, null pointer ...) */
	MCOM_E_HIP = -2,      /* HIP runtime error, text in mcom_last_error */
	MCOM_E_NOMEM = -3,
	MCOM_E_OVERFLOW = -4  /* a caller-provided output capacity was too small; nothing partial is valid */
};

/* read classes of process_reads (kthread_reads.c:84-225) */
enum mcom_read_class {
	MCOM_CLS_SKETCH = 0,  /* kept: N substituted, sketched, goes to a bucket   :182-218 */
	MCOM_CLS_ALLA = 1, MCOM_CLS_ALLT = 2, MCOM_CLS_ALLN = 3,                  /* :84-111  */
	MCOM_CLS_NEARA = 4, MCOM_CLS_NEART = 5, MCOM_CLS_NEARN = 6,                /* :113-126 */
	MCOM_CLS_NHEAVY = 7                                                        /* :219-224 */
};

/* ---- context ---------------------------------------------------------------------------------- */
/* stream: a hipStream_t (may be NULL for the default stream).  The context never owns the stream. */
int  mcom_create(mcom_ctx **out, int device, void *hip_stream);
void mcom_destroy(mcom_ctx *ctx);
int  mcom_set_stream(mcom_ctx *ctx, void *hip_stream);
int  mcom_sync(mcom_ctx *ctx);
const char *mcom_last_error(const mcom_ctx *ctx);
const char *mcom_version(void);

/* Optional kernel timing: when enabled every hot kernel launch is bracketed by HIP events recorded on the
 * context's stream.  Names: classify_pack sketch_reads radix_pass sketch_contigs find_next dict_build
 * realign_windows.  mcom_prof_read synchronises and returns the accumulated device time and launch count. */
int mcom_prof_enable(mcom_ctx *ctx, int on);
int mcom_prof_reset(mcom_ctx *ctx);
int mcom_prof_read(mcom_ctx *ctx, const char *name, double *total_ms, uint64_t *launches);
/* The kernels launched while the profiler was on, as text: one line "kernel name<TAB>launches" per kernel instantiation (the
 * compiler's spelling, which is the name rocprofv3 prints), for the class `name`; "*" = every kernel of the library, inside a
 * timed class or not; "-" = those outside every class.  A caller that attaches counter figures to a class's time (bench.py) can
 * so ask which kernels that time belongs to.  *need = bytes of the full text with its NUL; the text is cut at cap.        */
int mcom_prof_kernels(mcom_ctx *ctx, const char *name, char *buf, size_t cap, size_t *need);

/* The library keeps freed device blocks of its own objects for reuse.  mcom_pool_trim gives the free ones back to the runtime;
 * mcom_set_oom_hook names a function the library calls when it cannot get a block even so (a caller with a pool of its own frees
 * it there), before it gives up with MCOM_E_NOMEM.                                                                          */
void mcom_pool_trim(void);
void mcom_set_oom_hook(void (*hook)(void));

/* Diagnostics.  mcom_counter: "sort_overflow_segments" = segments of mcom_sort_group that did not fit its in-LDS sort and
 * went through the nine-pass sort instead (a minimizer shared by thousands of reads); "sketch_strings" = strings sketched by the
 * lane-per-string kernel of mcom_sketch_contigs so far (64 per wave).  (The knobs that force the rare code paths on small
 * inputs are test hooks and live in include/mcom_test.h, not here.)                                                    */
uint64_t mcom_counter(const mcom_ctx *ctx, const char *name);

/* ---- a4 + a2: reads --------------------------------------------------------------------------- */
/* Replaces kt_for_reads / process_reads (kthread_reads.c:247, :40-230) for a batch of n reads:
 * classify, substitute N by the majority base (tie order A,T,G,C), 2-bit pack, and sketch the kept
 * reads with mm_sketch_two (sketch.c:238).  d_ascii is [n][pitch] bytes, upper-case ACGTN.
 * Outputs (all [n]): d_packed [n][W] (N packs as the substituted base for class 0, as A otherwise),
 * d_cls, d_ncnt (number of N), d_rec (x=y=UINT64_MAX unless class 0; rid = rid0+i).
 * d_nmask (optional, may be NULL): [n][ceil(L/64)] bit i set when base i was 'N'.              */
int mcom_process_reads(mcom_ctx *ctx, const uint8_t *d_ascii, size_t pitch, size_t n, int L, int k, int e,
                       uint32_t rid0, uint64_t *d_packed, uint8_t *d_cls, uint16_t *d_ncnt,
                       uint64_t *d_nmask, mcom_mm128 *d_rec);
/* The two halves of mcom_process_reads as calls of their own (round 5): process_reads' counts, classes, N substitution and packing
 * (kthread_reads.c:55-205) -- bound by HBM -- and mm_sketch_two over the packed rows with the records of the other classes blanked
 * (kthread_reads.c:206-230, sketch.c:238) -- bound by the VALU.  A caller that cuts its reads into batches runs the sketch of one
 * batch on one stream beside the classification of the next on another (mcomh_kt_for_reads does).  Same outputs as the one call.    */
int mcom_classify_reads(mcom_ctx *ctx, const uint8_t *d_ascii, size_t pitch, size_t n, int L, int e,
                        uint64_t *d_packed, uint8_t *d_cls, uint16_t *d_ncnt, uint64_t *d_nmask);
int mcom_sketch_classified(mcom_ctx *ctx, const uint64_t *d_packed, const uint8_t *d_cls, size_t n, int L, int k, uint32_t rid0, mcom_mm128 *d_rec);

/* The same for reads the caller has packed already (round 4: a FASTQ parser that packs on the host sends 2 bits per base and one N
 * flag per base over PCIe instead of a byte per base): d_in_packed [n][W] codes A0 C1 G2 T3 with 0 at an N, d_in_nmask [n][ceil(L/64)].
 * Same outputs; d_packed / d_nmask may be the input arrays themselves (d_nmask NULL: the masks are not kept).                        */
int mcom_process_reads_packed(mcom_ctx *ctx, const uint64_t *d_in_packed, const uint64_t *d_in_nmask, size_t n, int L, int k, int e,
                              uint32_t rid0, uint64_t *d_packed, uint8_t *d_cls, uint16_t *d_ncnt, uint64_t *d_nmask, mcom_mm128 *d_rec);

/* The reads that are not class 0 (all-A / all-T / all-N, near-poly, N-heavy: the reference's special files, preprocess.c:84-126,
 * :219-224), as a list: d_list[i] = rid << 8 | class for i < min(*d_count, cap), in NO particular order (the caller sorts: rid
 * order is file order); *d_count = how many there are (it may exceed cap: take the class array itself then).  A handful per
 * million reads on real data: the host learns them from a few bytes instead of from one byte per read.  Asynchronous on the
 * context's stream; d_count is cleared by the call.                                                                        */
int mcom_special_reads(mcom_ctx *ctx, const uint8_t *d_cls, size_t n, uint64_t *d_list, uint32_t cap, uint32_t *d_count);

/* Batched mm_sketch_two (sketch.c:238-289) on packed rows.  d_rids (optional): sketch rows
 * d_rids[i] of d_packed and stamp that rid (the re-sketch of rejected reads with k-1, k-2, ...,
 * kthread_bucket.c:205, :489); NULL: rows rid0+i... i.e. row i with rid rid0+i.                 */
int mcom_sketch_reads(mcom_ctx *ctx, const uint64_t *d_packed, const uint32_t *d_rids, size_t n, int L,
                      int k, uint32_t rid0, mcom_mm128 *d_rec);
/* The records of reads whose minimizers are known already (they were sketched with the same k on another rank and
 * came through the bucket exchange): d_rec[i] = { d_x[i], (rid0+i)<<32 | d_ylow[i] } with d_ylow = position<<1 | strand
 * as mm_sketch_two left it (sketch.c:271); x = UINT64_MAX (no minimizer) gives the all-ones record.                */
int mcom_records_assemble(mcom_ctx *ctx, const uint64_t *d_x, const uint32_t *d_ylow, size_t n, uint32_t rid0, mcom_mm128 *d_rec);

/* hash64 (sketch.c:27-37) of n k-mers (2k-bit values), batched: the invertible mix every sketch kernel applies.            */
int mcom_hash64_batch(mcom_ctx *ctx, const uint64_t *d_kmer, size_t n, int k, uint64_t *d_hash);

/* ---- a5 + a6: sort and group ------------------------------------------------------------------- */
/* radix_sort_128x (misc.c:22, ksort.h:153): sorts n records in place by x ascending.  Stable: equal keys
 * keep their input order (the reference is stable only up to 64 elements, ksort.h:155).             */
int mcom_radix_sort_128x(mcom_ctx *ctx, mcom_mm128 *d_a, size_t n);

/* One Stage-1 round's front half for all 2^b buckets: the per-bucket radix_sort_128x, the run-length
 * grouping of equal hashes and the cmpcluster ordering of process_bucket (kthread_bucket.c:391-446,
 * :44-62).  d_rec: n records in ascending rid order, sketched with k = kmer (k_orig = the run's first k,
 * which cmpcluster keeps using in later rounds); records with x = UINT64_MAX (no minimizer) are ignored.
 * Outputs, in the order the reference visits them (bucket ascending, hash ascending):
 *   d_sorted  [n]      all records, sorted (ignored records last)
 *   d_singles [<=n]    rid of every group of one                     (:402-413, pushed to reads->sg)
 *   d_single_ord[<=n]  (optional) number of groups >= 2 visited before that single: lets the caller
 *                      interleave singles and groups in the reference's visiting order
 *   d_members [<=n]    y of every member of a group >= 2, each group in cmpcluster order (:438-442)
 *   d_group_off[<=n/2+1]  start of each such group in d_members, plus the end sentinel
 *   h_counts[4] (HOST) = { n_valid, n_singles, n_groups, n_members }.  Synchronous.                  */
int mcom_sort_group(mcom_ctx *ctx, const mcom_mm128 *d_rec, size_t n, int L, int k_orig, int kmer, int b,
                    mcom_mm128 *d_sorted, uint32_t *d_singles, uint32_t *d_single_ord, uint64_t *d_members,
                    uint32_t *d_group_off, uint64_t *h_counts);

/* ---- a3, a7..a9: contigs ------------------------------------------------------------------------- */
/* Batched mm_sketch_lh_ori (sketch.c:116-165): the (w,k)-minimizers of n contigs.  Contig c is the ASCII
 * string d_seq[d_off[c] .. d_off[c+1]) (ACGT, anything else is an ambiguous base that resets the run);
 * its record id is d_ids[c] (NULL: c, the contig index -- see "contig ids" above).
 * max_per_contig > 0 keeps only the first that many (callers index the first m, kthread_bucket.c:463).
 * Out: d_moff[n+1] = start of each contig's minimizers in d_out (position order), *h_total = their
 * number.  MCOM_E_OVERFLOW (with *h_total = a capacity that is enough) when cap is too small.
 * 1 <= w <= 128.  Synchronous.                                                                       */
int mcom_sketch_contigs(mcom_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_off, const uint32_t *d_ids, size_t n,
                        int w, int k, uint32_t max_per_contig, uint32_t *d_moff, mcom_mm128 *d_out, size_t cap,
                        uint64_t *h_total);

/* ASCII contigs -> the packed layout of mcom_realign_pass / mcom_match_pro: contig c occupies words
 * [d_coff[c], d_coff[c] + ceil(2*len/64)] of d_cbits, the last one being the mandatory padding word
 * (so d_coff[c+1] - d_coff[c] = ceil(2*len/64) + 1; total_words = their sum).  Non-ACGT packs as A.   */
int mcom_pack_contigs(mcom_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_off, const uint64_t *d_coff, uint32_t n,
                      uint64_t total_words, uint64_t *d_cbits);
/* The same for the words [w_lo, w_hi) of the layout only (the other words of d_cbits are not touched): on several GPUs a rank packs its
 * share of the words -- shares need not end where contigs do -- and the shares are all-gathered (round 5; before, every rank packed all). */
int mcom_pack_contigs_words(mcom_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_off, const uint64_t *d_coff, uint32_t n,
                            uint64_t total_words, uint64_t *d_cbits, uint64_t w_lo, uint64_t w_hi);
/* The way back: the strings of n contigs (upper-case ACGT: what the consensus kernels write) from their packed words.  d_coff[c] = first
 * word of contig c in d_cbits (n entries), d_off[c] = its first character in d_seq (n + 1 entries; d_seq 16-byte aligned); byte_lo / byte_hi
 * = d_off[0] / d_off[n] on the host.  Bytes of d_seq outside [byte_lo, byte_hi) are not touched.  Several GPUs send a new contig once, as
 * packed words, and every rank makes the strings it did not build itself (no counterpart in the reference, a shared-memory program). */
int mcom_unpack_contigs(mcom_ctx *ctx, const uint64_t *d_cbits, const uint64_t *d_coff, const uint64_t *d_off, uint32_t n,
                        uint64_t byte_lo, uint64_t byte_hi, uint8_t *d_seq);
/* The same for the set after a merge round (cp_cluster, kthread_cb.c:397-434: the merged contigs first, then the untouched ones
 * in their order): contigs [0, n_first) are packed from their strings, contig n_first + u takes the packed words of contig
 * d_keepidx[u] of the set before the round (d_cbits_old / d_coff_old).  Same words as mcom_pack_contigs; d_cbits has room for
 * total_words + 2 words and the two behind the set are cleared (a window at the very end of the last contig reads past it).      */
int mcom_pack_contigs_merged(mcom_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_off, const uint64_t *d_coff, uint32_t n,
                             uint64_t total_words, uint32_t n_first, const uint64_t *d_cbits_old, const uint64_t *d_coff_old,
                             const uint32_t *d_keepidx, uint64_t *d_cbits);

/* mm_idx_init + mm_idx_generation (kthread_idx.c:77, :116-170): index over n minimizer records sketched
 * with k, given in the order the reference pushes them (contig order, minimizer order).
 *   b > 0: the reference's layout and ORDER: 2^b buckets by x & (2^b-1), each sorted by radix_sort_128x with
 *          its exact element order (stable up to 64 entries per bucket, a cycle-leader permutation of equal
 *          keys above, ksort.h:132-144) -- what find_next iterates over (kthread_idx.c:154-155).
 *   b = 0: one stable sort by x (equal minimizers keep the order they were given in).
 * mcom_idx_get = mm_idx_get (:84-101) for n minimizers: start/count into the index's sorted record array
 * (count 0 = absent), which mcom_idx_records copies out.                                              */
typedef struct mcom_idx mcom_idx;
int  mcom_idx_build(mcom_ctx *ctx, const mcom_mm128 *d_rec, size_t n, int k, int b, mcom_idx **out);
/* The same index in steps, for a caller that builds it by bucket range on several GPUs and all-gathers the parts (buckets are
 * independent: kthread_idx.c:116-168 is a loop over them):
 *   mcom_idx_create      an empty index for n records in all
 *   mcom_idx_sort_part   the records of some buckets (in pushing order) sorted into their place rec[base_rec ..): bucket by bucket,
 *                        each in radix_sort_128x's element order; *h_max_bucket = the fullest of them
 *   mcom_idx_table_part  the table regions of buckets [bucket0, bucket1) -- those of the part -- sized for the fullest bucket of
 *                        the WHOLE index (every builder passes the same max_bucket_all); MCOM_E_OVERFLOW: a bucket too large for a
 *                        region (LDS) -- put all sorted records together and call mcom_idx_table_global instead
 *   mcom_idx_buffers     where the parts of other builders are received: sorted records [n], table slots (16 bytes each, *region
 *                        per bucket, bucket v at slot v * region)
 * mcom_idx_build = create + sort_part(everything) + table_part(all buckets).                                              */
int  mcom_idx_create(mcom_ctx *ctx, size_t n, int k, int b, mcom_idx **out);
int  mcom_idx_sort_part(mcom_ctx *ctx, mcom_idx *mi, const mcom_mm128 *d_rec, size_t n, size_t base_rec, uint32_t *h_max_bucket);
int  mcom_idx_table_part(mcom_ctx *ctx, mcom_idx *mi, uint32_t max_bucket_all, uint32_t bucket0, uint32_t bucket1);
/* the regions of all buckets from the sorted records of all parts, once they have been received (instead of receiving the regions too) */
int  mcom_idx_table_all(mcom_ctx *ctx, mcom_idx *mi, uint32_t max_bucket_all);
int  mcom_idx_table_global(mcom_ctx *ctx, mcom_idx *mi);
int  mcom_idx_buffers(mcom_idx *mi, mcom_mm128 **d_rec, uint64_t **d_slots, uint32_t *region);
/* radix_sort_128x with the reference's exact element order (sequential emulation, see mcom_radix_sort_128x
 * for the fast stable sort).                                                                          */
int  mcom_radix_sort_128x_ref_order(mcom_ctx *ctx, mcom_mm128 *d_a, size_t n);
void mcom_idx_destroy(mcom_ctx *ctx, mcom_idx *mi);
int  mcom_idx_get(mcom_ctx *ctx, const mcom_idx *mi, const uint64_t *d_x, size_t n, uint32_t *d_start, uint32_t *d_count);
int  mcom_idx_records(mcom_ctx *ctx, const mcom_idx *mi, mcom_mm128 *d_out, size_t *n);

/* Batched match_pro (kthread_cb.c:36-52): mismatches over the whole overlap of contigs a[i], b[i] when
 * base pos_a[i] of a is laid on base pos_b[i] of b.                                                   */
int mcom_match_pro(mcom_ctx *ctx, const uint64_t *d_cbits, const uint64_t *d_coff, const uint32_t *d_clen,
                   const uint32_t *d_a, const uint32_t *d_pos_a, const uint32_t *d_b, const uint32_t *d_pos_b, size_t n,
                   uint32_t *d_mismatch);

/* Lookup part of find_next (kthread_cb.c:267-291) for all contigs at once.  d_query: every minimizer of
 * every contig (mcom_sketch_contigs with max_per_contig 0), contig after contig.  For each query, in
 * order, and each index hit, in index order, the pair passes when the hit belongs to another contig,
 * has the same strand bit and match_pro <= cbthr.  d_out receives the passing pairs in that order as
 * { x = query y (contig index in the id, pos_ori, dir), y = hit y (other contig, pos, dir) }.  The merge
 * flags (:286, :339) change while merging and stay with the caller.  h_counts = { pairs, passing }.
 * MCOM_E_OVERFLOW when cap is too small.  The counts are back when the call returns; d_out is complete when
 * the context's stream reaches that point (round 5: no wait for the last kernel).                         */
int mcom_find_next_candidates(mcom_ctx *ctx, const mcom_idx *mi, const mcom_mm128 *d_query, size_t n_query,
                              const uint64_t *d_cbits, const uint64_t *d_coff, const uint32_t *d_clen, int cbthr,
                              mcom_mm128 *d_out, size_t cap, uint64_t *h_counts);
/* The same for a merge round after the first (kthread_cb.c:570-627): contigs [0, n_new) are the ones the round before made (they
 * head the new list, cp_cluster order :397-434), the others came through it unmerged.  A pair of two of those others was a candidate
 * in the round before, with the same strings at the same positions, and did not pass then (a passing pair of two contigs that both
 * stay unclaimed does not exist: the first of the two to be visited takes the other, :286-343) -- so it is not evaluated again.
 * Queries of such a contig probe the index only through keys that a new contig has there (a bit map of those keys), and their other
 * pairs are not listed: h_counts[0] counts the pairs listed.  Same output as mcom_find_next_candidates; n_new = 0 lists and
 * evaluates every pair.                                                                                                          */
int mcom_find_next_candidates_new(mcom_ctx *ctx, const mcom_idx *mi, const mcom_mm128 *d_query, size_t n_query,
                                  const uint64_t *d_cbits, const uint64_t *d_coff, const uint32_t *d_clen, int cbthr, uint32_t n_new,
                                  mcom_mm128 *d_out, size_t cap, uint64_t *h_counts);

/* The same over a contig set that the merge rounds leave where it is (cp_cluster, kthread_cb.c:397-434, without its copies: the set is
 * an append-only store, a contig's index -- and the id in its records -- never changes, and the list of a round is d_ord, the
 * contigs in visiting order; NULL: the store's own order).  d_rec / d_roff: all minimizer records of the store and their offsets per
 * contig; the queries are the records of contigs d_ord[0 .. n_contigs) in that order.  The contigs the round before made are the
 * ones with index >= first_new (n_new of them; n_new = 0: the first round, every pair is evaluated).  Same pairs, same order.      */
int mcom_find_next_candidates_ord(mcom_ctx *ctx, const mcom_idx *mi, const mcom_mm128 *d_rec, const uint32_t *d_roff, const uint32_t *d_ord, size_t n_contigs,
                                  const uint64_t *d_cbits, const uint64_t *d_coff, const uint32_t *d_clen, int cbthr, uint32_t first_new, uint32_t n_new,
                                  mcom_mm128 *d_out, size_t cap, uint64_t *h_counts);

/* ---- contig consensus on the device ------------------------------------- */
/* construct_ref (kthread_bucket.c:69-377) for all n_groups groups of mcom_sort_group's output at once.
 * In : d_members / d_group_off as mcom_sort_group returns them (sketch records y in cmpcluster order).
 * Out: d_members[q] = rid<<32 | off<<1 | dir with off = column of the member in the FIRST consensus (:101);
 *      d_keep[q] = 1 when the member has <= e mismatches against it (:189), else it is a reject (:194);
 *      per group: d_nkept, d_sv (first covered column, :305-317; final offsets are off - sv, :349),
 *      d_reflen and the second consensus (:319-341) as ASCII at d_refs + g*ref_stride (ref_stride >= 2L). */
int mcom_group_consensus(mcom_ctx *ctx, const uint64_t *d_packed, uint64_t *d_members, const uint32_t *d_group_off,
                         uint32_t n_groups, int L, int k_orig, int e, uint8_t *d_keep, uint32_t *d_nkept,
                         uint16_t *d_sv, uint16_t *d_reflen, uint8_t *d_refs, int ref_stride);

/* construct_ref2 (kthread_cb.c:105-218) for a batch of merged contigs ("jobs").  Members of job j are
 * d_members[d_job_off[j] .. d_job_off[j+1]), already sorted by cmpcluster2 (offset, then direction); its
 * consensus, of length max(offset)+L = d_ref_off[j+1]-d_ref_off[j], is written as ASCII at d_refs+d_ref_off[j].
 * The work list is (d_tile_job[t], d_tile_idx[t]): tile d_tile_idx[t] (512 columns) of job d_tile_job[t].  */
int mcom_merge_consensus(mcom_ctx *ctx, const uint64_t *d_packed, const uint64_t *d_members, const uint64_t *d_job_off,
                         const uint64_t *d_ref_off, const uint32_t *d_tile_job, const uint32_t *d_tile_idx,
                         uint32_t n_tiles, int L, uint8_t *d_refs);

/* First m minimizers of every contig out of a full mcom_sketch_contigs result (what the reference's contig
 * builders push into the index, kthread_bucket.c:463, kthread_cb.c:370, :423).  h_total may be NULL.      */
int mcom_minimizer_prefix(mcom_ctx *ctx, const uint32_t *d_moff, const mcom_mm128 *d_rec, size_t n, uint32_t m,
                          uint32_t *d_out_moff, mcom_mm128 *d_out, uint64_t *h_total);

/* ... for a set whose contigs are not stored in the order they are visited in: contig d_ord[c] is the c-th (NULL: the set's own order) */
int mcom_minimizer_prefix_ord(mcom_ctx *ctx, const uint32_t *d_moff, const mcom_mm128 *d_rec, const uint32_t *d_ord, size_t n, uint32_t m,
                              uint32_t *d_out_moff, mcom_mm128 *d_out, uint64_t *h_total);

/* ---- a10..a15: Stage-2 realignment --------------------------------------------------------------- */
/* setglobalarrays_realign (kthread_hash_realign.c:153-206): first/last base of every dictionary key.
 * Host only.  Returns numdict_s (>= 1) or a negative status; start/end need room for 16 entries.    */
int mcom_dict_layout(int L, int ininumdict, int *start, int *end);

/* rows d_rids[i] of d_packed -> d_out[i] (the singleton pool of singleRead2bitset, bbhashdict.c:146) */
int mcom_gather_rows(mcom_ctx *ctx, const uint64_t *d_packed, const uint32_t *d_rids, size_t n, int L, uint64_t *d_out);

/* Near-poly-A / poly-T filter of singleRead2bitset (bbhashdict.c:157-222).  d_flag[i] = 1 (A list),
 * 2 (T list) or 0.  d_nmask (optional, with d_rids): N mask rows [n_reads][ceil(L/64)] indexed by rid.  */
int mcom_poly_filter(mcom_ctx *ctx, const uint64_t *d_sgbits, const uint64_t *d_nmask, const uint32_t *d_rids,
                     size_t n_sg, int L, int thr, uint8_t *d_flag);

/* constructdictionary_realign + boomphf::mphf (kthread_hash_realign.c:3-140, BooPHF.h:891-1009): the
 * numdict_s dictionaries over n_sg packed singletons, device resident.  A bin lists its singleton
 * indices ascending (read_id[]); key -> bin is an exact hash table instead of an MPHF (every reference
 * hit is re-verified against the bin key, :385-386, so results do not depend on the map).  Synchronous. */
typedef struct mcom_dicts mcom_dicts;
int  mcom_dicts_build(mcom_ctx *ctx, const uint64_t *d_sgbits, size_t n_sg, int L, int ininumdict, mcom_dicts **out);
void mcom_dicts_free(mcom_ctx *ctx, mcom_dicts *d);
int  mcom_dicts_info(const mcom_dicts *d, int *nd, uint32_t *numkeys, uint32_t *maxbin);
/* bphf->lookup + findpos (bbhashdict.c:33-43) for n keys of dictionary `dict`: bin start/size into the
 * dictionary's id array (size 0 = key absent).  mcom_dicts_ids copies that id array (n_sg entries).   */
int  mcom_dicts_lookup(mcom_ctx *ctx, const mcom_dicts *d, int dict, const uint64_t *d_keys, size_t n,
                       uint32_t *d_start, uint32_t *d_count);
int  mcom_dicts_ids(mcom_ctx *ctx, const mcom_dicts *d, int dict, uint32_t *d_ids_out);

/* realign_hash_search over every window of every contig (kthread_hash_realign.c:316-508).
 *   d_cbits : 2-bit packed contigs; contig c starts at word d_coff[c], holds ceil(2*len/64) words and is
 *             followed by at least ONE padding word;  d_woff[c] = number of windows of contigs < c
 *             (a contig of length len has max(0, len-L+1) windows);  n_windows = their total.
 *   d_sgflag: [n_sg] nonzero = not claimable (flagged by mcom_poly_filter)
 *   d_claim : [n_sg] out.  UINT64_MAX = unclaimed, else c<<33 | window<<5 | dir<<4 | dict: the FIRST
 *             (contig, window, dir, dict) in the reference's visiting order whose test the read passes,
 *             i.e. exactly where the sequential reference claims it.  Appending order inside one tuple
 *             is descending singleton index (the bin is walked from its end, :388).
 *   d_stats : optional [3] = { lookups, bin hits, candidates tested } (diagnostics, slows the kernel)   */
int mcom_realign_pass(mcom_ctx *ctx, const mcom_dicts *d, const uint64_t *d_sgbits, const uint8_t *d_sgflag,
                      const uint64_t *d_cbits, const uint64_t *d_coff, const uint64_t *d_woff, uint32_t n_contigs,
                      uint64_t n_windows, int thr, int maxsearch, uint64_t *d_claim, uint64_t *d_stats);

/* The same pass driven from the singletons (results identical to mcom_realign_pass; it is the production path).
 * The klen-mers of the Stage-2 contigs (which do not change between passes, preprocess.c:197-232) are indexed ONCE.
 * The index (csrc/cindex.hip) is a multi-map of 64-byte lines -- word 0: entries in the line (| 0x100 when entries were
 * pushed past it), words 1-7: entries = 12-bit tag of the key | contig | position (52 bits between them: 24 + 28 up to 2^24 contigs, more
 * contigs take bits from the position; the build refuses a contig too long for its field) -- cut into
 * partitions of equal size; a key hashes to a partition and a home line, its entries lie in the home line and the lines
 * behind it; a key with more copies than a few lines hold (a repeat) keeps them in a run of lines of its own in the
 * extension area behind the partitions.  It is BUILT BY RADIX PARTITIONING: two streaming passes split
 * the entries by partition, then one workgroup per partition places its entries and writes its lines once (no scattered
 * insert, no memset, no atomics in HBM).
 *   mcom_cindex_plan   sizes it for a contig set: *geom (partitions | lines per partition << 16 | shares | owner, handed to the build
 *                      and the lookups), *n_words = uint64 words of d_keys (header, partitions, extension area)
 *   mcom_cindex_build  fills d_keys from the packed contigs (d_woff must hold n_contigs + 1 entries); temporaries of
 *                      24 bytes per entry come from the library's block pool.  MCOM_E_OVERFLOW: the extension area is
 *                      too small for this set's repeats -- build again with a larger n_words.  Synchronous.
 * mcom_realign_pass_reads then looks up, for every unflagged singleton, the key of each dictionary l at contig position
 * window + ds[l] and the reverse complement of that key at window + L - ds[l] - klen (the two probes of
 * kthread_hash_realign.c:380 and :446 seen from the read), verifies every hit like :390-393 / :458-461 -- including the
 * exact key comparison of :385-386, which the tags leave open -- and keeps the minimum claim key per singleton.
 *   d_elig : NULL, or [n_sg] bit l set = the singleton is within the last `maxsearch` entries of its bin of
 *            dictionary l as built (mcom_dicts_eligible).  This static cut equals mcom_realign_pass with its
 *            maxsearch argument; both equal the reference only while no bin exceeds maxsearch -- for longer
 *            bins use mcom_dicts_bigbins / mcom_realign_pass_tuples below
 *   d_stats: optional [3] = { lookups, windows verified, tuples passing }                                */
int mcom_cindex_plan(uint64_t n_windows, uint32_t n_contigs, int L, int ininumdict, uint64_t *n_entries, uint64_t *geom, uint64_t *n_words);
int mcom_cindex_build(mcom_ctx *ctx, const uint64_t *d_cbits, const uint64_t *d_coff, const uint64_t *d_woff, uint32_t n_contigs,
                      uint64_t n_windows, int L, int ininumdict, uint64_t geom, uint64_t *d_keys, uint64_t n_words);
/* Multi-GPU: the ONE index over all contigs shared out BY KEY.  The hash range is cut into `ranks` equal shares; rank q builds and
 * holds share q (a table of its own, 1 / ranks of the entries) and looks up only the keys that hash into it -- every rank sees every
 * singleton (rows replicated), does 1 / ranks of the lookups and verifications, and the claim keys are MIN-reduced.  The build is
 * two calls with the caller's exchange between them:
 *   mcom_cindex_plan_shared  geometry of every share (the same on all ranks but for the owner field): *n_entries = entries of the
 *                            whole index (upper bound), *n_share = room a share needs for the entries it receives, *geom, *n_words
 *   mcom_cindex_entries      the entries of contigs [c0, c1) of the set (a rank takes a range of the replicated set): d_key / d_slot
 *                            [cap >= their positions], grouped by owning share in share order, h_counts[ranks] (HOST) = entries per
 *                            share.  With one share they come grouped by the high byte of their partition (place: grouped = 1).
 *                            Entries carry the global contig index.  MCOM_E_OVERFLOW when cap is too small.  Synchronous.
 *   mcom_cindex_place        this share's table from the n_ent entries it received, in any order (grouped = 0): two radix passes by
 *                            partition (high byte, then low byte inside each high-byte region), then one workgroup per partition writes its lines.  d_key / d_slot are overwritten,
 *                            d_key_tmp / d_slot_tmp are scratch of n_ent entries.  MCOM_E_OVERFLOW as for mcom_cindex_build.
 * mcom_cindex_build = mcom_cindex_entries + mcom_cindex_place for one share.                                                */
int mcom_cindex_plan_shared(uint64_t n_windows, uint32_t n_contigs, int L, int ininumdict, int ranks, int rank, uint64_t *n_entries,
                            uint64_t *n_share, uint64_t *geom, uint64_t *n_words);
int mcom_cindex_entries(mcom_ctx *ctx, const uint64_t *d_cbits, const uint64_t *d_coff, const uint64_t *d_woff, uint32_t n_contigs,
                        uint32_t c0, uint32_t c1, int L, int ininumdict, uint64_t geom, uint32_t *d_key, uint64_t *d_slot, uint64_t cap,
                        uint64_t *h_counts);
int mcom_cindex_place(mcom_ctx *ctx, uint32_t *d_key, uint64_t *d_slot, uint64_t n_ent, int grouped, uint32_t *d_key_tmp, uint64_t *d_slot_tmp,
                      int L, int ininumdict, uint64_t geom, uint64_t *d_keys, uint64_t n_words);
/* mcom_cindex_place in its two steps (round 5): 2a sorts the entries by partition -- the sorted arrays are one of the two pairs handed in
 * (*d_key_sorted / *d_slot_sorted say which), the entries of partition v are [d_pstart[v], d_pstart[v + 1]) (n_parts + 1 device words) --
 * and 2b makes the table from them.  The Stage-2 join (mcom_realign_join) works on the result of 2a alone.                          */
int mcom_cindex_partition(mcom_ctx *ctx, uint32_t *d_key, uint64_t *d_slot, uint64_t n_ent, int grouped, uint32_t *d_key_tmp, uint64_t *d_slot_tmp,
                          int L, int ininumdict, uint64_t geom, uint32_t *d_pstart, const uint32_t **d_key_sorted, const uint64_t **d_slot_sorted);
int mcom_cindex_assemble(mcom_ctx *ctx, const uint32_t *d_key_sorted, const uint64_t *d_slot_sorted, const uint32_t *d_pstart, int L, int ininumdict,
                         uint64_t geom, uint64_t *d_keys, uint64_t n_words);
int mcom_dicts_eligible(mcom_ctx *ctx, const mcom_dicts *d, const uint64_t *d_sgbits, int maxsearch, uint32_t *d_elig);
/* Screen before mcom_dicts_build: *h_may_exceed = 0 proves that no bin of any dictionary over these singletons
 * holds more than maxsearch reads (hashed counters, an upper bound of every bin), so the read-driven pass needs
 * neither the dictionaries nor d_elig; 1 = build them and look.  Synchronous.                                */
int mcom_dicts_screen(mcom_ctx *ctx, const uint64_t *d_sgbits, size_t n_sg, int L, int ininumdict, int maxsearch, int *h_may_exceed);
/* the same in two steps: _begin puts the counting on the context's stream and returns, _end waits and answers (so that a caller
 * can run the screen on a second context / stream beside other work)                                                     */
int mcom_dicts_screen_begin(mcom_ctx *ctx, const uint64_t *d_sgbits, size_t n_sg, int L, int ininumdict, int maxsearch);
/* multi-GPU: the keys shared out as the contig index is (mcom_cindex_plan_shared): this call counts the keys of share `share` of
 * n_shares only -- a bin is counted whole by one rank -- and the caller ORs the answers of all ranks                          */
int mcom_dicts_screen_begin_shared(mcom_ctx *ctx, const uint64_t *d_sgbits, size_t n_sg, int L, int ininumdict, int maxsearch, int n_shares, int share);
int mcom_dicts_screen_end(mcom_ctx *ctx, int *h_may_exceed);
int mcom_realign_pass_reads(mcom_ctx *ctx, const uint64_t *d_keys, uint64_t geom,
                            const uint64_t *d_sgbits, const uint8_t *d_sgflag, const uint32_t *d_elig, size_t n_sg,
                            const uint64_t *d_cbits, const uint64_t *d_coff, const uint64_t *d_woff, uint32_t n_contigs,
                            int L, int ininumdict, int thr, uint64_t *d_claim, uint64_t *d_stats);

/* Stage 2 without the table (round 5; csrc/realign.hip, "PARTITION-LOCAL JOIN"): the singletons' keys are sorted by index partition like
 * the entries (mcom_cindex_partition leaves those as d_ekey / d_eslot / d_epstart) and one workgroup per partition joins the two lists in
 * LDS; every match is verified as mcom_realign_pass_reads verifies it.  d_claim [n_sg] as there (UINT64_MAX = unclaimed).  A candidate that
 * fails at `thr` for its distance alone (<= maxthr) is kept as a deferred tuple (16 bytes: claim key; read id | distance << 32 | the two
 * encode_byte answers) in d_defer [defer_cap]: mcom_realign_deferred is the whole pass at a later threshold -- d_rids / d_sgflag / n_sg: that
 * pass's singleton list and flags, d_map: scratch of n_reads words.  *h_status = 1: not applicable (a dictionary bin may exceed maxsearch,
 * a partition holds more queries than LDS takes, more than 2^27 singletons, more deferred tuples than defer_cap) -- nothing of the
 * output is valid then and the caller builds the table (mcom_cindex_assemble) and runs mcom_realign_pass_reads.  One share only.
 * Both synchronous.                                                                                                                */
int mcom_realign_join(mcom_ctx *ctx, uint64_t geom, const uint32_t *d_ekey, const uint64_t *d_eslot, const uint32_t *d_epstart,
                      const uint64_t *d_sgbits, const uint8_t *d_sgflag, const uint32_t *d_rids, size_t n_sg, const uint64_t *d_cbits,
                      const uint64_t *d_coff, const uint64_t *d_woff, uint32_t n_contigs, int L, int ininumdict, int thr, int maxthr, int maxsearch,
                      uint64_t *d_claim, uint64_t *d_stats, uint64_t *d_defer, uint64_t defer_cap, uint64_t *h_n_defer, int *h_status);
int mcom_realign_deferred(mcom_ctx *ctx, const uint64_t *d_defer, uint64_t n_defer, const uint32_t *d_rids, const uint8_t *d_sgflag, size_t n_sg,
                          uint32_t *d_map, size_t n_reads, int thr, uint64_t *d_claim, uint64_t *d_stats);
/* encode_byte (kthread_hash_realign.c:283-314), batched: d_ok[i] = 1 when the run-length mismatch text of read row i
 * (d_rows [n][W], packed) against the L bases of contig d_contig[i] from base d_pos[i] on -- their reverse complement when
 * d_dir[i] != 0, as :446-461 compare a reverse-strand read -- is at most 0.4 L characters long (the match-run counter is not
 * reset after a short run, as in the reference).  d_cbits / d_coff: packed contigs as for mcom_realign_pass.                */
int mcom_encode_byte(mcom_ctx *ctx, const uint64_t *d_rows, const uint64_t *d_cbits, const uint64_t *d_coff, const uint32_t *d_contig,
                     const uint32_t *d_pos, const uint8_t *d_dir, size_t n, int L, uint8_t *d_ok);

/* Bins longer than maxsearch, exactly.  The reference walks the LIVE part of a bin from its end for at most
 * maxsearch entries (kthread_hash_realign.c:388, findpos at bbhashdict.c:33-49) and takes claimed reads out of every
 * bin after the visit that claimed them (:420-435, remove at bbhashdict.c:51-67), so a read deep in a long bin turns
 * visible once enough of the reads above it are gone.  Only the members of such bins depend on that order:
 *   mcom_dicts_bigbins       d_binstart [nd][n_sg]: start of the singleton's bin in dictionary l when that bin
 *                            holds more than maxsearch reads, else 0xFFFFFFFF;  d_mark [n_sg]: 1 = member of
 *                            at least one such bin
 *   mcom_realign_pass_tuples mcom_realign_pass_reads for the unmarked singletons; for a marked one d_claim stays
 *                            all-ones and EVERY tuple it passes is appended to d_tuples as {claim key, singleton
 *                            index} (2 x uint64 each, at most `cap`; *h_ntuples is the number found, larger
 *                            than cap = run again with a larger buffer).  Synchronous.
 *   mcom_claims_patch        d_claim[d_idx[i]] = d_val[i]: the result of replaying the marked singletons
 * The replay itself (a few thousand reads, in visiting order) is host work: minicom_amd/host/mcom_pipeline.cpp.  */
int mcom_dicts_bigbins(mcom_ctx *ctx, const mcom_dicts *d, const uint64_t *d_sgbits, int maxsearch, uint32_t *d_binstart, uint8_t *d_mark);
int mcom_realign_pass_tuples(mcom_ctx *ctx, const uint64_t *d_keys, uint64_t geom,
                             const uint64_t *d_sgbits, const uint8_t *d_sgflag, const uint8_t *d_mark, size_t n_sg,
                             const uint64_t *d_cbits, const uint64_t *d_coff, const uint64_t *d_woff, uint32_t n_contigs,
                             int L, int ininumdict, int thr, uint64_t *d_claim, uint64_t *d_stats,
                             uint64_t *d_tuples, uint64_t cap, uint64_t *h_ntuples);
int mcom_claims_patch(mcom_ctx *ctx, uint64_t *d_claim, const uint32_t *d_idx, const uint64_t *d_val, size_t n);

/* The member lists after Stage 2 (csrc/finalize.hip).  The reference sorts a contig's members at the start of every
 * scan (cmpcluster2, kthread_hash_realign.c:318) and appends behind them, so after m passes contig c holds
 * stable_sort(C(c) + P_1(c) + ... + P_{m-1}(c)) + P_m(c); Stage 2 never reads the lists, so they are assembled once.
 *   d_mem / d_moff [n_contigs + 1]: the lists before Stage 2;  d_app_contig[i] / d_app_member[i] (HOST arrays of device
 *   pointers) and h_app_n[i]: what pass i appended, as mcom_claims_resolve returned it (a pass that appended nothing
 *   still counts: n_passes is the number of scans);  key_bits: 2^key_bits - 1 exceeds every offset<<1|dir.
 *   out: d_mem2 [n_members + sum h_app_n], d_moff2 [n_contigs + 1].  Synchronous.                                  */
int mcom_members_finalize(mcom_ctx *ctx, const uint64_t *d_mem, const uint64_t *d_moff, size_t n_contigs, uint64_t n_members,
                          const uint32_t *const *d_app_contig, const uint64_t *const *d_app_member, const uint64_t *h_app_n,
                          int n_passes, int key_bits, uint64_t *d_mem2, uint64_t *d_moff2);

/* The members one pass appends, in the order the sequential scan appends them (claim key ascending, singleton
 * index descending: the bin is walked from its end, kthread_hash_realign.c:388, :408-409, :474-475).
 *   d_claim [n_sg] from mcom_realign_pass(_reads);  d_rids [n_sg] read id of every singleton
 *   d_flag  [n_sg] in/out: the entry of every claimed singleton is set to 3
 *   d_app_contig / d_app_member [<= n_sg] out: contig index and member word rid<<32 | window<<1 | dir of the
 *   *h_nwon appended members.  Synchronous.                                                                 */
int mcom_claims_resolve(mcom_ctx *ctx, const uint64_t *d_claim, const uint32_t *d_rids, size_t n_sg, uint32_t n_contigs,
                        uint8_t *d_flag, uint32_t *d_app_contig, uint64_t *d_app_member, uint64_t *h_nwon);

/* ---- the contig set of combine_cluster, resident in HBM between merge rounds (kthread_cb.c:570-630) ---- */
/* A contig set on the device: ASCII consensus strings d_seq with offsets d_soff[n+1], member words d_mem
 * (rid<<32 | offset<<1 | dir, breads.h:49-58) with offsets d_moff[n+1], all minimizers d_rec with offsets
 * d_roff[n+1] (uint32).  The claiming of find_next (:267-343) stays with the caller; these entry points do the
 * rest of a round without the data leaving the device.  All of them are synchronous.                       */

/* What process_bucket does with the outcome of construct_ref (kthread_bucket.c:446-505), for the ng groups of a
 * round after mcom_group_consensus: a group that keeps more than one member (:451) is appended to the contig set
 * (which already holds n_have contigs, chars_have chars, members_have members; capacities seq_cap / mem_cap /
 * off_cap entries), its members re-based to the first covered column (:349).  Every other member of a group is
 * a reject: d_rej_rid / d_rej_group [<= rej_cap] list them in the reference's order (group ascending; inside a
 * group the members construct_ref dropped, :194-213, then the single kept one of a dissolved group, :477-498)
 * with the index of their group.  h_counts[4] = { new contigs, new chars, new members, rejects }.
 * MCOM_E_OVERFLOW (counts set, nothing written) when a capacity is too small.  Synchronous.                   */
int mcom_groups_to_contigs(mcom_ctx *ctx, const uint64_t *d_members, const uint32_t *d_goff, size_t ng, const uint8_t *d_keep,
                           const uint32_t *d_nkept, const uint16_t *d_sv, const uint16_t *d_reflen, const uint8_t *d_refs,
                           int ref_stride, uint64_t n_have, uint64_t chars_have, uint64_t members_have, uint8_t *d_seq,
                           uint64_t seq_cap, uint64_t *d_soff, uint64_t *d_mem, uint64_t mem_cap, uint64_t *d_moff,
                           uint64_t off_cap, uint32_t *d_rej_rid, uint32_t *d_rej_group, uint64_t rej_cap, uint64_t *h_counts);

/* The first-come claiming of find_next (:267-343, :339-343) for a whole round: d_pairs = the passing candidate
 * pairs in the reference's visiting order (as mcom_find_next_candidates emits them).  The reference takes a pair
 * iff neither contig has been taken by an earlier pair, which is the greedy matching over the list; it is settled
 * in rounds (a pair that is the earliest live pair at both of its contigs is taken).  Out: d_jobs = *h_nj x
 * { ci, cj, pos_ori, pos } in claiming order (room for n_contigs / 2 + 1 jobs), d_flag[n_contigs] = 1 for every claimed contig.
 * MCOM_E_OVERFLOW when the list has not settled after max_rounds (the caller then claims sequentially).      */
int mcom_claim_pairs(mcom_ctx *ctx, const mcom_mm128 *d_pairs, size_t n_pairs, size_t n_contigs, int max_rounds, uint32_t *d_jobs,
                     uint8_t *d_flag, uint64_t *h_nj, int *h_rounds);

/* exclusive 64-bit prefix sums, in place allowed */
int mcom_scan_u64(mcom_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, size_t n);
/* packed layout of a set for mcom_pack_contigs: d_coff_words[n+1], d_clen[n], *h_total_words                 */
int mcom_contig_layout(mcom_ctx *ctx, const uint64_t *d_soff, size_t n, uint64_t *d_coff_words, uint32_t *d_clen,
                       uint64_t *h_total_words);
/* Window offsets of a contig set for Stage 2: d_woff[c] = windows of L bases before contig c (a contig shorter than L
 * has none, kthread_hash_realign.c:320), d_woff[n] = *h_n_windows; *h_maxlen (optional) = the longest contig.
 * Synchronous.                                                                                                      */
int mcom_window_layout(mcom_ctx *ctx, const uint64_t *d_soff, size_t n, int L, uint64_t *d_woff, uint64_t *h_n_windows, uint64_t *h_maxlen);
/* updateSingle (preprocess.c:243-255) on the device: d_out = the ids whose flag is zero, in order; *h_n_out their number.
 * Synchronous.                                                                                                      */
int mcom_compact_live(mcom_ctx *ctx, const uint32_t *d_ids, const uint8_t *d_flag, size_t n, uint32_t *d_out, uint64_t *h_n_out);
/* The entries whose flag is 1 or 2 (the near-poly-A / -T singletons mcom_poly_filter found, bbhashdict.c:177-216), in any
 * order: d_out[3 i] = index, d_out[3 i + 1] = d_ids[index], d_out[3 i + 2] = flag for the first `cap` of them; *d_count
 * (DEVICE) = how many there are (more than cap: look at the flags themselves).  Asynchronous.                        */
int mcom_list_flagged(mcom_ctx *ctx, const uint32_t *d_ids, const uint8_t *d_flag, size_t n, uint32_t *d_out, uint32_t cap, uint32_t *d_count);
/* Member lists of the nj claimed pairs (find_next :297-325): d_jobs = nj x {ci, cj, pos_ori, pos} (uint32);
 * the list of the contig whose anchor lies further right first, the other one shifted behind it, then in
 * cmpcluster2 order (stable, as construct_ref2's sort :107 with glibc's merge sort).  key_bits: every
 * offset<<1|dir of the merged lists is below 2^key_bits (<= 29).  Out: d_jm with offsets d_jmoff[nj+1],
 * d_jroff[nj+1] = offsets of the merged consensus strings (length = last offset + L, :112-113),
 * h_totals[3] = { members, consensus chars, longest merged contig }.                                         */
int mcom_merge_members(mcom_ctx *ctx, const uint64_t *d_mem, const uint64_t *d_moff, const uint32_t *d_jobs, size_t nj, int L,
                       int key_bits, uint64_t *d_jm, uint64_t *d_jmoff, uint64_t *d_jroff, uint64_t *h_totals);
/* mcom_merge_consensus with the tile list made on the device from d_jroff.  With the claimed pairs (d_jobs, as
 * given to mcom_merge_members) and the set they came from (d_seq, d_soff) only the columns where the two parents
 * overlap are counted; elsewhere a column sees the members of one parent only and keeps that parent's character.
 * All three NULL: every column is counted.                                                                   */
int mcom_merge_consensus_jobs(mcom_ctx *ctx, const uint64_t *d_packed, const uint64_t *d_jm, const uint64_t *d_jmoff,
                              const uint64_t *d_jroff, size_t nj, uint64_t total_chars, int L, uint8_t *d_refs,
                              const uint32_t *d_jobs, const uint8_t *d_seq, const uint64_t *d_soff);
/* cp_cluster (:397-434): the new set holds the nj merged contigs first (their data and offset entries [0..nj]
 * already in d_seq2/d_soff2/d_mem2/d_moff2), then the nkeep contigs of the old set with d_flag[i] == 0, in
 * their order.  Out: the rest of the new arrays, d_keepidx[nkeep] = old index of every carried contig,
 * h_totals[2] = { chars, members } of the new set.                                                           */
int mcom_contigs_carry(mcom_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_soff, const uint64_t *d_mem, const uint64_t *d_moff,
                       size_t n, const uint8_t *d_flag, size_t nj, size_t nkeep, uint8_t *d_seq2, uint64_t *d_soff2,
                       uint64_t *d_mem2, uint64_t *d_moff2, uint32_t *d_keepidx, uint64_t *h_totals);
/* The minimizers of a carried contig do not change, only its index does: records of old contig d_keepidx[u]
 * are appended at d_rec2[base ...] with id first_id+u, d_roff2[first_id .. first_id+nkeep] are written;
 * *h_total = records in d_rec2 afterwards.  MCOM_E_OVERFLOW when cap2 is too small.                           */
int mcom_records_carry(mcom_ctx *ctx, const mcom_mm128 *d_rec, const uint32_t *d_roff, const uint32_t *d_keepidx, size_t nkeep,
                       uint32_t first_id, uint32_t base, mcom_mm128 *d_rec2, size_t cap2, uint32_t *d_roff2, uint64_t *h_total);

/* The minimizers of the merged contigs of one round without sketching them whole (csrc/resketch.hip): a merged contig
 * differs from its parents only inside their overlap, and whether a k-mer becomes a record is decided by the w
 * entries either side of it (sketch.c:138-161), so the parents' records outside that reach are taken over and only a
 * segment around the overlap is sketched.  Result identical to mcom_sketch_contigs on the merged contigs (k odd).
 *   d_jobs [nj][4] claimed pairs as in mcom_merge_members;  d_soff / d_rec / d_roff: string offsets, records and
 *   record offsets of the PARENT set;  d_seq2 / d_soff2: the merged contigs (jobs first, mcom_merge_consensus_jobs);
 *   out: d_roff2 [nj + 1], d_rec2 [<= cap2] with ids j; *h_total records, *h_sketched_chars bases actually sketched.
 * MCOM_E_OVERFLOW with *h_total = the room needed.  Synchronous.                                                  */
int mcom_resketch_merged(mcom_ctx *ctx, const uint32_t *d_jobs, size_t nj, const uint64_t *d_soff, const mcom_mm128 *d_rec,
                         const uint32_t *d_roff, const uint8_t *d_seq2, const uint64_t *d_soff2, uint64_t merged_chars, int w, int k,
                         uint32_t *d_roff2, mcom_mm128 *d_rec2, size_t cap2, uint64_t *h_total, uint64_t *h_sketched_chars);

/* ... the merged contigs get the ids id_base + j (a store that the merged contigs are appended to: their index there)             */
int mcom_resketch_merged_at(mcom_ctx *ctx, const uint32_t *d_jobs, size_t nj, const uint64_t *d_soff, const mcom_mm128 *d_rec,
                            const uint32_t *d_roff, const uint8_t *d_seq2, const uint64_t *d_soff2, uint64_t merged_chars, int w, int k, uint32_t id_base,
                            uint32_t *d_roff2, mcom_mm128 *d_rec2, size_t cap2, uint64_t *h_total, uint64_t *h_sketched_chars);
/* mcom_merge_members with the room at d_jm stated: MCOM_E_OVERFLOW, h_totals[0] = the members to come and nothing written, when
 * jm_cap (elements) is too small.                                                                                               */
int mcom_merge_members_cap(mcom_ctx *ctx, const uint64_t *d_mem, const uint64_t *d_moff, const uint32_t *d_jobs, size_t nj, int L,
                           int key_bits, uint64_t *d_jm, uint64_t jm_cap, uint64_t *d_jmoff, uint64_t *d_jroff, uint64_t *h_totals);
/* Merge rounds without cp_cluster's copies (kthread_cb.c:397-434).  The contig set is an append-only store: strings, member lists,
 * minimizer records and packed words with offset arrays that gain one entry per contig ever made; a round appends its merged contigs
 * behind everything (mcom_merge_members / _consensus_jobs / mcom_resketch_merged_at / mcom_pack_contigs write behind the store's end with
 * offsets from 0, mcom_offsets_append turns those into entries of the store's arrays: d_dst[j] = base + d_rel[j], j = 0 .. n) and
 * mcom_order_next makes the list of the next round: the nj new contigs (indices first_new ...) in claiming order, then the contigs of
 * d_ord[0 .. n) (NULL: 0 .. n-1) whose d_flag[index] is 0, in their order -- cp_cluster's order.  *h_nkeep = how many those are.
 * mcom_contigs_gather copies the contigs d_idx[0 .. n_idx) into a set of their own, in that order (offset arrays from 0;
 * h_totals = { chars, members }): the store becomes an ordinary set again when the rounds are over.                                */
int mcom_offsets_append(mcom_ctx *ctx, const uint64_t *d_rel, size_t n, uint64_t base, uint64_t *d_dst);
int mcom_offsets_append_u32(mcom_ctx *ctx, const uint32_t *d_rel, size_t n, uint32_t base, uint32_t *d_dst);
int mcom_order_next(mcom_ctx *ctx, const uint32_t *d_ord, size_t n, const uint8_t *d_flag, uint32_t first_new, size_t nj, uint32_t *d_ord2, uint64_t *h_nkeep);
int mcom_contigs_gather(mcom_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_soff, const uint64_t *d_mem, const uint64_t *d_moff,
                        const uint32_t *d_idx, size_t n_idx, uint8_t *d_seq2, uint64_t *d_soff2, uint64_t *d_mem2, uint64_t *d_moff2, uint64_t *h_totals);

/* ---- the stream files of cluster_dump, made where the data is (kthread_dump.c:142-236, :364-417) ---- */
/* print_encode for every member of every contig: d_mem / d_moff = the member lists in dump order (cmpcluster2 inside a contig,
 * kthread_dump.c:143: mcom_members_finalize with one empty pass sorts them so), d_cbits / d_coff the packed contigs
 * (mcom_pack_contigs), d_packed / d_nmask (may be NULL: no read holds an N) the reads.  Out, as file images:
 *   d_pos  [4 n_contigs + 2 n_members] beg_pos.bin: per contig its member count (u32), then the 16-bit position deltas (:167, :224)
 *   d_dir  [(n_members + 7) / 8]       dir.bin: one direction bit per member, least significant first (breads.h:241-248)
 *   d_text [*h_text_bytes]             dif_char.txt: per member the run-length mismatch text of :198-221 and a newline
 * MCOM_E_OVERFLOW (with *h_text_bytes = the room needed; d_pos / d_dir are complete) when text_cap is too small.  Synchronous. */
int mcom_dump_members(mcom_ctx *ctx, const uint64_t *d_packed, const uint64_t *d_nmask, int L, const uint64_t *d_cbits, const uint64_t *d_coff,
                      const uint64_t *d_mem, const uint64_t *d_moff, size_t n_contigs, uint64_t n_members, uint8_t *d_pos, uint8_t *d_dir,
                      uint8_t *d_text, uint64_t text_cap, uint64_t *h_text_bytes);
/* ref.bin: the contig strings back to back, four bases per byte (breads.h:232-239): d_out [(chars + 3) / 4]                      */
int mcom_dump_refbin(mcom_ctx *ctx, const uint8_t *d_seq, uint64_t chars, uint8_t *d_out);
/* single.seq: reads d_rids[0 .. n) back to back, four bases per byte (kthread_dump.c:390-417): d_out [(n L + 3) / 4]            */
int mcom_dump_singles(mcom_ctx *ctx, const uint64_t *d_packed, const uint32_t *d_rids, uint64_t n, int L, uint8_t *d_out);
/* The order-preserving (`minicom -p`, ORDER) and paired-end (_PE) file sets differ from the default one in the member order inside a contig
 * and in their id streams (kthread_dump.c:33-138, kthread_dump_pe.c:35-120, :218-619):
 *   mcom_members_order3  the member lists in cmpcluster3 order (offset, then read id: kthread_cb.c:72-84, the qsort of kthread_dump.c:34):
 *                        d_mem2 [n_members]; key_bits as for mcom_members_finalize.  mcom_dump_members then runs on them unchanged
 *   mcom_dump_ids_order  ids.bin: per member its read id, or the difference to the id before it when the 16-bit position delta is 0 (:116-127)
 *   mcom_dump_ids_text   ids.txt of the paired-end mode: "%d %u\n" = file of the read (id >= half: 1), read id (kthread_dump_pe.c:70-74);
 *                        d_text NULL and text_cap 0: a sizing call (*h_text_bytes)
 *   mcom_dump_pairing    peids.bin.sp / file.bin.sp over the reads of the eight lists (d_lists: their ids in the decoder's order) and
 *                        peids.bin.0 / file.bin.0 over the members: a first-file read is numbered in the decoder's order among first-file
 *                        reads, a second-file read (id >= half) writes the number of its mate id - half; file.bin: one bit per read
 *                        (kthread_dump_pe.c:270-470, :583-612).  d_ids_sp [<= n_list], d_file_sp [(n_list + 7) / 8], d_ids_0 [<= n_members],
 *                        d_file_0 [(n_members + 7) / 8]; h_counts = { entries of d_ids_sp, of d_ids_0 }.  All synchronous.                 */
int mcom_members_order3(mcom_ctx *ctx, const uint64_t *d_mem, const uint64_t *d_moff, size_t n_contigs, uint64_t n_members, int key_bits, uint64_t *d_mem2);
int mcom_dump_ids_order(mcom_ctx *ctx, const uint64_t *d_mem, const uint64_t *d_moff, size_t n_contigs, uint64_t n_members, uint32_t *d_ids);
int mcom_dump_ids_text(mcom_ctx *ctx, const uint64_t *d_mem, uint64_t n_members, uint32_t half, uint8_t *d_text, uint64_t text_cap, uint64_t *h_text_bytes);
int mcom_dump_pairing(mcom_ctx *ctx, const uint32_t *d_lists, uint64_t n_list, const uint64_t *d_mem, uint64_t n_members, uint32_t half,
                      uint32_t *d_ids_sp, uint8_t *d_file_sp, uint32_t *d_ids_0, uint8_t *d_file_0, uint64_t *h_counts);
/* One stream set per thread (kthread_dump.c:370-379, minicom:110-146: the number of stream files is part of the format, info.txt says it):
 * the sets are cut at contig boundaries out of the images above.  The _at forms also say where in a text image the lines of given members
 * start (h_text_at[q] for member index h_at_members[q]) and how many second-file reads lie among the members in front of a given one
 * (h_second_at: where a set's peids.bin starts); mcom_dump_member_bits packs dir.bin (which = 0) or file.bin (which = 1: read id >= half)
 * of the members of one set from bit 0, as the reference's per-thread bit writer does.                                                  */
int mcom_dump_members_at(mcom_ctx *ctx, const uint64_t *d_packed, const uint64_t *d_nmask, int L, const uint64_t *d_cbits, const uint64_t *d_coff,
                         const uint64_t *d_mem, const uint64_t *d_moff, size_t n_contigs, uint64_t n_members, uint8_t *d_pos, uint8_t *d_dir,
                         uint8_t *d_text, uint64_t text_cap, uint64_t *h_text_bytes, const uint64_t *h_at_members, int n_at, uint64_t *h_text_at);
int mcom_dump_ids_text_at(mcom_ctx *ctx, const uint64_t *d_mem, uint64_t n_members, uint32_t half, uint8_t *d_text, uint64_t text_cap, uint64_t *h_text_bytes,
                          const uint64_t *h_at_members, int n_at, uint64_t *h_text_at);
int mcom_dump_pairing_at(mcom_ctx *ctx, const uint32_t *d_lists, uint64_t n_list, const uint64_t *d_mem, uint64_t n_members, uint32_t half,
                         uint32_t *d_ids_sp, uint8_t *d_file_sp, uint32_t *d_ids_0, uint8_t *d_file_0, uint64_t *h_counts,
                         const uint64_t *h_at_members, int n_at, uint64_t *h_second_at);
int mcom_dump_member_bits(mcom_ctx *ctx, const uint64_t *d_mem, uint64_t n_members, int which, uint32_t half, uint8_t *d_out);
/* d_flag[i] = 1 when read d_rids[i] holds an N (such unclustered reads go to single_N.seq as text, kthread_dump.c:400-407)       */
int mcom_rows_have_n(mcom_ctx *ctx, const uint64_t *d_nmask, const uint32_t *d_rids, size_t n, int L, uint8_t *d_flag);

/* ---- multi-GPU helpers (no reference counterpart: the reference is a shared-memory program) ---- */
/* Stable partition of n records by the rank that owns their minimizer bucket: bucket beta = x & (2^b - 1) belongs to
 * rank (beta * ranks) >> b -- contiguous bucket ranges in rank order, so that rank-major order is the reference's
 * bucket-ascending visiting order (kthread_bucket.c:531-560).  Records without a minimizer (x = UINT64_MAX) are dropped.
 * d_out: the records grouped by owner, each group in input order; h_counts[ranks] (HOST).  Synchronous.           */
int mcom_partition_by_owner(mcom_ctx *ctx, const mcom_mm128 *d_rec, size_t n, int b, int ranks, mcom_mm128 *d_out, uint64_t *h_counts);
/* stable sort by read id (y >> 32): the parts received from several senders, each in rid order, become the rid-ordered
 * list mcom_sort_group expects                                                                                    */
int mcom_sort_by_rid(mcom_ctx *ctx, mcom_mm128 *d_a, size_t n);
/* d_out[i] = min over q < n_parts of d_parts[q * stride + i]: folds the ranks' shares of the Stage-2 claim keys      */
int mcom_min_fold_u64(mcom_ctx *ctx, const uint64_t *d_parts, int n_parts, size_t stride, size_t n, uint64_t *d_out);
/* *h_max = largest element (0 for n = 0).  Synchronous.                                                             */
int mcom_max_u16(mcom_ctx *ctx, const uint16_t *d_v, size_t n, uint32_t *h_max);
/* records sketched for contigs [first_contig, ...) of a set as if they were contigs 0, 1, ...: ids += first_contig;
 * their n_off record offsets += first_record                                                                        */
int mcom_records_rebase(mcom_ctx *ctx, mcom_mm128 *d_rec, size_t n_rec, uint32_t first_contig, uint32_t *d_roff, size_t n_off, uint32_t first_record);

/* d_off[i] += delta for n 64-bit offsets: a rank's share of a round's merged contigs moves to its place in the set            */
int mcom_offsets_rebase(mcom_ctx *ctx, uint64_t *d_off, size_t n, uint64_t delta);

/* Digest of a device array: h_sum_xor[0] = wrapping sum of its little-endian 64-bit words, each weighted by an odd
 * function of its index, h_sum_xor[1] = their xor (a tail of fewer than 8 bytes is zero-extended).  d_data 8-byte
 * aligned.  Synchronous.                                                                                            */
int mcom_digest(mcom_ctx *ctx, const void *d_data, size_t bytes, uint64_t *h_sum_xor);

/* ---- synthetic input (bench / tests): same generator as minicom_amd/synth.py ------------------ */
int mcom_synth_reads(mcom_ctx *ctx, uint64_t seed, uint64_t n_reads, int L, int coverage, double sub_rate,
                     uint64_t first, uint64_t count, uint8_t *d_ascii, size_t pitch);
/* genome_kind 0: the uniform genome of mcom_synth_reads; 1: a repeat-rich genome (blocks of 25 kb: a 2 kb segment out of families of
 * forty copies in every block, tandem repeats, poly-A and (AT)n stretches -- csrc/reads.hip), the hard case for buckets, index runs
 * and Stage-2 bins.  Device generator only (bench.py --genome repeats; no host twin).                                          */
int mcom_synth_reads_genome(mcom_ctx *ctx, uint64_t seed, uint64_t n_reads, int L, int coverage, double sub_rate, int genome_kind,
                            uint64_t first, uint64_t count, uint8_t *d_ascii, size_t pitch);

#ifdef __cplusplus
}
#endif
#endif
