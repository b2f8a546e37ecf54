/* include/mcom_host.h -- host side of the MI355X-native minicom hot path (libmcom_host.so).
 *
 * Mirrors the reference's pipeline driver for this path: the stage functions have the reference's names
 * and meaning (pre_process preprocess.c:39, kt_for_reads kthread_reads.c:247, kt_for_bucket
 * kthread_bucket.c:562, combine_cluster kthread_cb.c:570, realign_hash kthread_hash_realign.c:569,
 * updateSingle preprocess.c:243) but run their hot loops as HIP kernels through include/mcom.h.
 * The contig set lives on the device from kt_for_bucket on and is still there, complete (Stage 2's appends folded into
 * the member lists, in the reference's order), when pre_process returns; accessors, stage dumps and the stream writer
 * copy it to the host on demand.  The host keeps the loop control of the stages and orders the singleton list.
 * No CPU fallback: every stage needs the GPU.
 * Results equal the reference at one thread (-t 1, its only deterministic mode).
 */
#ifndef MCOM_HOST_H
#define MCOM_HOST_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
	int k;          /* -k  0 = 31 (17 when L < 80)      minicommain.c:92-114 */
	int e;          /* -e  0 = 4                         minicommain.c:60     */
	int m;          /* -m  0 = 6                         minicommain.c:63     */
	int w;          /* -w  0 = L/2-k (3 when L < 70)     preprocess.c:89-107  */
	int cbthr;      /* -g  0 = 2e                        minicommain.c:122    */
	int max_rounds; /* -R  0 = 35                        minicommain.c:64     */
	int step;       /* -S  0 = e (5 when e > 10)         minicommain.c:130    */
	int maxthr;     /* -E  0 = L/2                       minicommain.c:140    */
	int numdict;    /* -s  0 = L/17 (L/11 when L <= 80)  kthread_hash_realign.c:153 */
	int host_threads; /* host worker threads for the consensus loops (0 = 1); results do not depend on it */
	/* The library reads NO environment variable: what changes results or the kernels used is said here.            */
	int maxsearch;      /* 0 = the reference's 500 / 2000 (preprocess.c:169-172); > 0 forces the bin cut of
	                       kthread_hash_realign.c:388 (test hook: lets a small input exercise bins above the cut)   */
	int window_scan;    /* 1 = Stage 2 with the window-driven kernel (realign_hash_search as written) instead of the
	                       read-driven one; same claims, ~50x slower: the cross-check of tests/test_gpu_realign.py   */
	int full_consensus; /* 1 = count every column of a merged contig (construct_ref2 as written) instead of only the
	                       parents' overlap; same strings (A/B switch for measurements)                             */
	int full_sketch;    /* 1 = sketch merged contigs whole instead of around the overlap; same records (A/B switch) */
	int overlap_screen; /* 1 = the first Stage-2 pass gathers the singletons' rows and screens the dictionaries on the copy stream, beside
	                       the contig index build, instead of behind it in the main stream; same result.  Measured on MI355X (100 M
	                       reads): the two kernels slow each other down by what the overlap saves (243.7 vs 245-249 ms per step), so
	                       it is off by default                                                                                 */
	int host_dump;      /* 1 = mcomh_cluster_dump, _order and _pe write their streams with the host loop (every base of every member, as the
	                       reference's print_encode does) instead of the device encoders (csrc/streams.hip); same files (A/B switch, the
	                       cross-check of tests/test_streams.py)                                                                      */
	int stream_sets;    /* number of stream sets mcomh_cluster_dump* writes (0 or 1: one).  The reference writes one set per thread -- ref.bin.T,
	                       beg_pos.bin.T, dir.bin.T, dif_char.txt.T [, ids.bin.T | ids.txt.T, peids.bin.T, file.bin.T], info.txt = "L n_threads"
	                       (kthread_dump.c:370-379) -- and its decoder takes them in parallel (decompress.c:1248-1300); here the contigs are cut
	                       into that many runs of about equal member counts.  The `minicom -t N` command line passes N (device encoders only)   */
	int stage2_join;    /* 1 = Stage 2 on one GPU as the partition-local join of round 5 (mcom_realign_join once, mcom_realign_deferred in the later
	                       passes: no index table, no dictionary screen pass) instead of the table (mcom_cindex_place + mcom_realign_pass_reads in every
	                       pass); same claims.  Measured at 100 M x 150 bp: the two take the same time, so the table stays the
	                       default; the join needs 17 GB less.  Inputs the join does not take (a dictionary bin that may exceed maxsearch, ...) go
	                       through the table by themselves                                                                                  */
	int read_batches;   /* 1 = mcomh_kt_for_reads in four batches over two streams for reads resident in HBM: the classification of a batch (bound by
	                       HBM) beside the sketch of the batch before (bound by the VALU), instead of one launch each over all reads.  Same arrays.
	                       Measured at 100 M x 150 bp, steps alternating on one box: 178.7 against 179.5 ms -- off by default                      */
} mcomh_params;

typedef struct mcomh_pipeline mcomh_pipeline;

/* reads: n rows of L upper-case ACGTN characters.  host_reads [n][L] lives in host memory and is uploaded;
 * alternatively d_reads [n][pitch] is already resident in HBM (exactly one of the two is non-NULL).      */
int  mcomh_create(mcomh_pipeline **out, int device, void *hip_stream, const uint8_t *host_reads,
                  const uint8_t *d_reads, size_t pitch, size_t n, int L, const mcomh_params *p);
/* Host-to-host form: host_reads [n][L] stays with the caller (valid until mcomh_kt_for_reads has returned) and is NOT
 * copied on the host; kt_for_reads uploads it in chunks through two device staging blocks, the copy of chunk c+1 running
 * beside classify / pack / sketch of chunk c.  Page-locked memory (hipHostMalloc, hipHostRegister) makes that overlap
 * real; pageable memory works, serialised by the runtime.  No stage dump for such a pipeline (the reads are not kept).  */
int  mcomh_create_streamed(mcomh_pipeline **out, int device, void *hip_stream, const uint8_t *host_reads, size_t n, int L,
                           const mcomh_params *p);
/* Same, from 2-bit packed rows already in HBM (include/mcom.h format, ACGT only, every read kept): the
 * entry used after the multi-GPU minimizer-bucket exchange, where a rank receives its partition packed.  */
int  mcomh_create_packed(mcomh_pipeline **out, int device, void *hip_stream, const uint64_t *d_packed, size_t n, int L,
                         const mcomh_params *p);
/* Optional, between mcomh_create_packed and mcomh_kt_for_reads: the minimizers of the packed rows are known (sketched
 * with the pipeline's k by the rank that sent them); d_x [n] hashes, d_ylow [n] position<<1 | strand.  kt_for_reads then
 * assembles the records instead of sketching the rows again.  The arrays must stay valid until kt_for_reads returns. */
/* File(s) -> pipeline (round 4; replaces bseq_open / bseq_read / bseq_read_second, bseq.c:19-66, preprocess.c:52-75).  A plain
 * four-line FASTQ file is parsed by up to 64 threads, packed by them (2 bits per base + one N flag per base: 64 bytes per read over
 * PCIe at L = 150 instead of 150) and sent straight into the pipeline's row arrays; classes, N counts and the majority-base
 * substitution are made on the device.  gzip, FASTA and multi-line records go through the sequential reader.  path2 (may be NULL):
 * the mates' file, whose reads follow those of the first.  err [err_cap]: the message of a failure.                              */
int  mcomh_create_from_fastq(mcomh_pipeline **out, int device, void *hip_stream, const char *path1, const char *path2, const mcomh_params *pp,
                             char *err, size_t err_cap);
int  mcomh_set_records(mcomh_pipeline *p, const uint64_t *d_x, const uint32_t *d_ylow);
void mcomh_destroy(mcomh_pipeline *p);
/* Device blocks of destroyed pipelines are kept for the next one of the process (a steady-state step allocates nothing); this gives
 * the free ones -- the driver's and the library's -- back to the runtime.  Both pools do so by themselves when memory runs out.    */
void mcomh_pool_trim(void);
const char *mcomh_last_error(const mcomh_pipeline *p);

int mcomh_kt_for_reads(mcomh_pipeline *p);
int mcomh_kt_for_bucket(mcomh_pipeline *p);
int mcomh_combine_cluster(mcomh_pipeline *p);
int mcomh_update_single(mcomh_pipeline *p);
/* updateSingle + realign_hash at threshold thr; *cluster_reads = reads held by contigs afterwards */
int mcomh_realign_hash(mcomh_pipeline *p, int thr, long *cluster_reads);
/* the Stage-2 loop alone (preprocess.c:197-232): passes at thr = e, e + S, ... until one adds too few reads        */
int mcomh_stage2(mcomh_pipeline *p);
/* the whole timed region of the reference: Stage 1 + Stage 2 with its loop control (preprocess.c:141-233) */
int mcomh_pre_process(mcomh_pipeline *p);
/* runs everything and writes the state after every stage in the text format of oracle/refdump.cpp */
int mcomh_dump_stages(mcomh_pipeline *p, const char *path);

/* the pre-bsc stream files of cluster_dump at one thread (kthread_dump.c:364-678),
 * single-end, not order-preserving: ref.bin.0 beg_pos.bin.0 dir.bin.0 dif_char.txt.0 info.txt single.seq
 * single_N.seq AA.txt TT.txt NN.txt, written into the existing directory `folder`.  Call after mcomh_pre_process. */
int mcomh_cluster_dump(mcomh_pipeline *p, const char *folder);
/* Inverse of those files (the reference's decompress for that mode, decompress.c:495-760): one read per line into
 * out_path, order = all-A/T/N, near-constant reads, N reads, unclustered reads, contig reads.  No GPU needed. */
int mcomh_decompress(const char *folder, const char *out_path, uint64_t *n_reads);
/* SURVEY section 8f rank 4, single-end part: the order-preserving mode (minicom -p = the reference compiled with
 * ORDER): members ordered by cmpcluster3 (kthread_cb.c:72), ids.bin.0 (kthread_dump.c:116-127), every list sorted by
 * read id with a delta-coded *.ids.bin beside it and the read count in info.txt (:377-379, :420-543); and its inverse
 * (decompress.c:109-493), which writes the reads in their original order.                                        */
int mcomh_cluster_dump_order(mcomh_pipeline *p, const char *folder);
int mcomh_decompress_order(const char *folder, const char *out_path, uint64_t *n_reads);
/* ... and the paired-end part (minicompe = the reference compiled with _PE, kthread_dump_pe.c:218-619): the pipeline
 * holds the reads of the first file as rows [0, n/2) and their mates as rows [n/2, n) (mcomh_fastq_pair_to_device);
 * besides the contig / list streams the file set carries one file bit per read and, for every read of the second
 * file, the output line of its mate (peids.bin.*, file.bin.*).  The inverse (decompress.c:780-1212) writes the first
 * file's reads to out_path1 and every mate to the same line of out_path2.                                          */
int mcomh_cluster_dump_pe(mcomh_pipeline *p, const char *folder);
int mcomh_decompress_pe(const char *folder, const char *out_path1, const char *out_path2, uint64_t *n_pairs);
/* both files of a pair into one device matrix, second file behind the first; MCOM_E_ARG when the counts differ      */
int mcomh_fastq_pair_to_device(const char *path1, const char *path2, int device, int *L, size_t chunk_reads, uint8_t **d_reads, size_t *n,
                               char *err, size_t err_cap);

/* FASTQ / FASTA ingest (bseq_open + bseq_read, bseq.c:19-66; kseq.h), plain or gzip.
 * Every read must have the same length (bseq.c:54-57 exits otherwise; here MCOM_E_ARG).  *L == 0: taken from
 * the first read.
 *   mcomh_fastq_read      : host only, the reads into out[cap_reads][L] (MCOM_E_OVERFLOW when there are more)
 *   mcomh_fastq_to_device : the reads into HBM as [n][L] characters, parsed into two pinned chunks of chunk_reads
 *                           rows (0 = 2^20) that alternate between the parser and the copy engine; *d_reads is
 *                           what mcomh_create takes as d_reads (pitch = L); release it with mcomh_device_free.   */
int mcomh_fastq_read(const char *path, int *L, uint8_t *out, size_t cap_reads, size_t *n);
int mcomh_fastq_to_device(const char *path, int device, int *L, size_t chunk_reads, uint8_t **d_reads, size_t *n, char *err, size_t err_cap);
void mcomh_device_free(void *d_ptr);

/* ---- multi-GPU: one process per GPU, reads sharded, results identical to the single-process run ----
 * The reference is a shared-memory program (pthreads, kthread_*.c); it has no communication layer to mirror, so this
 * is new interface.  A communicator carries ONE primitive, a byte-wise all-to-all with per-peer offsets and sizes
 * (everything else -- all-gather, the sums / minima of a few counters, the MIN-reduction of the Stage-2 claim keys --
 * is built on it), over one of two transports:
 *   RCCL      ncclSend / ncclRecv between ncclGroupStart / ncclGroupEnd on the pipeline's HIP stream: every peer pair
 *             talks over its own xGMI link, no ring.  Messages are cut at 256 MiB.  librccl.so.1 is loaded on first use.
 *   callbacks the caller supplies the all-to-all for HOST buffers (MPI_Alltoallv, torch.distributed over gloo, ...);
 *             device data is staged through pinned memory.  This is what the multi-process tests use on one GPU.
 * All collective calls are synchronous and must be made by every rank in the same order.                            */
typedef struct mcomh_comm mcomh_comm;
#define MCOMH_UNIQUE_ID_BYTES 128
typedef struct {
	/* send + send_off[q] .. + send_bytes[q] goes to rank q; what rank q sent to this rank arrives at recv + recv_off[q]
	 * (recv_bytes[q] bytes; sizes agree pairwise).  Host memory.  Returns 0 on success.                              */
	int (*alltoallv)(void *user, const void *send, const uint64_t *send_off, const uint64_t *send_bytes,
	                 void *recv, const uint64_t *recv_off, const uint64_t *recv_bytes);
} mcomh_comm_ops;
/* rank 0 makes the id (ncclGetUniqueId) and hands its 128 bytes to the other ranks by any means it has */
int  mcomh_comm_unique_id(void *id128);
int  mcomh_comm_create_rccl(mcomh_comm **out, int rank, int world, const void *id128, int device);
int  mcomh_comm_create_ops(mcomh_comm **out, int rank, int world, const mcomh_comm_ops *ops, void *user);
void mcomh_comm_destroy(mcomh_comm *c);
int  mcomh_comm_rank(const mcomh_comm *c);
int  mcomh_comm_world(const mcomh_comm *c);
const char *mcomh_comm_last_error(const mcomh_comm *c);
/* the primitive and what is built on it; on_device: the buffers are HBM pointers (ordered behind hip_stream's work)  */
int  mcomh_comm_alltoallv(mcomh_comm *c, const void *send, const uint64_t *send_off, const uint64_t *send_bytes,
                          void *recv, const uint64_t *recv_off, const uint64_t *recv_bytes, int on_device, void *hip_stream);
/* rank q's part lies at buf + off[q] (bytes[q] bytes) on every rank afterwards; send = NULL: this rank's part is
 * already in place                                                                                                  */
int  mcomh_comm_allgatherv(mcomh_comm *c, const void *send, void *buf, const uint64_t *off, const uint64_t *bytes,
                           int on_device, void *hip_stream);
/* element-wise over n host values: op 0 = sum, 1 = min, 2 = max                                                     */
int  mcomh_comm_allreduce_u64(mcomh_comm *c, uint64_t *vals, size_t n, int op);
/* bytes this rank has sent to OTHER ranks so far, and the number of all-to-alls                                     */
void mcomh_comm_stats(const mcomh_comm *c, uint64_t *bytes_sent, uint64_t *calls);
/* wall seconds this rank has spent inside all-to-all calls so far (staging copies and waiting for the peers included) */
double mcomh_comm_seconds(const mcomh_comm *c);

/* The distributed pipeline: this rank holds reads [rid0, rid0 + n_local) of n_total (shards are contiguous, in rank
 * order, and cover [0, n_total); paired end: the second file's reads follow the first file's, as in mcomh_create).
 * Every stage function below works as for one GPU and must be called by all ranks; afterwards EVERY rank holds the
 * complete result (contig set, lists), identical to what mcomh_create + the same calls give on one GPU over all reads:
 *   kt_for_reads     shard-local classify / pack / sketch; packed rows, classes and N masks all-gathered
 *   kt_for_bucket    per round the minimizer records go to the owner of their bucket (bucket ranges in rank order, so
 *                    that rank-major = the reference's visiting order; rejects re-sketched with k-r move again,
 *                    kthread_bucket.c:205-212, :489-496); new contigs are all-gathered into the replicated set
 *   combine_cluster  replicated set; contig sketching and the candidate evaluation of find_next sharded by contig
 *   realign_hash     contig 17-mer index sharded by contig range, every rank probes all singletons against its part,
 *                    claim keys MIN-reduced (the claim is a minimum)
 * comm is borrowed: it must outlive the pipeline.                                                                    */
int  mcomh_create_dist(mcomh_pipeline **out, int device, void *hip_stream, mcomh_comm *comm, const uint8_t *host_reads,
                       const uint8_t *d_reads, size_t pitch, size_t n_local, uint64_t rid0, uint64_t n_total, int L,
                       const mcomh_params *p);

/* results */
size_t mcomh_n_contigs(const mcomh_pipeline *p);
const char *mcomh_contig_ref(const mcomh_pipeline *p, size_t i, size_t *len);   /* consensus, NOT NUL-terminated */
size_t mcomh_contig_n(const mcomh_pipeline *p, size_t i);
const uint64_t *mcomh_contig_members(const mcomh_pipeline *p, size_t i);   /* rid<<32 | offset<<1 | dir */
const uint32_t *mcomh_list(const mcomh_pipeline *p, const char *name, size_t *n); /* allA allT allN fpA fpT fpN Nfile sg */
/* The whole contig set at once, as the flat host arrays the accessors above index into: consensus strings back to back
 * (ref, string i = [ref_off[i], ref_off[i+1])), member words back to back (mem, list i = [mem_off[i], mem_off[i+1])).
 * Any of the out pointers may be NULL.  The arrays belong to the pipeline.                                          */
int mcomh_contig_set(const mcomh_pipeline *p, size_t *n_contigs, const char **ref, const uint64_t **ref_off,
                     const uint64_t **mem, const uint64_t **mem_off);
/* Digest of the result, computed where the result lives (the contig set in HBM, the lists on the host); two runs over the
 * same reads must give the same eight numbers: out = { contigs, consensus characters, members, unclustered reads (sg),
 * digest of the strings, digest of the member words, digest of the string + member offsets, digest of sg and the class
 * lists } (digests: mcom_digest sum ^ rotated xor).  Call after mcomh_pre_process.                                  */
int mcomh_result_digest(mcomh_pipeline *p, uint64_t out[8]);
/* counters: rounds merge_rounds passes windows resketch n_sg0 big_bins; timers (ms): t_reads t_bucket
 * t_combine t_realign t_gpu t_host */
double mcomh_stat(const mcomh_pipeline *p, const char *name);
/* kernel timing of the pipeline's own mcom_ctx (mcom_prof_enable / mcom_prof_read of include/mcom.h) */
int mcomh_prof_enable(mcomh_pipeline *p, int on);
int mcomh_prof_read(mcomh_pipeline *p, const char *name, double *total_ms, uint64_t *launches);
/* mcom_prof_kernels of include/mcom.h over the pipeline's contexts: "kernel<TAB>launches" lines of class `name` ("*" = all) */
int mcomh_prof_kernels(mcomh_pipeline *p, const char *name, char *buf, size_t cap, size_t *need);

#ifdef __cplusplus
}
#endif
#endif
