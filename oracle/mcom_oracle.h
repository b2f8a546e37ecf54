/* oracle/mcom_oracle.h -- CPU restatement of minicom's sketch + index + overlap hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this.  The product (minicom_amd/, include/mcom.h) never does.
 *
 * Every function restates one reference function at ONE thread (the only deterministic mode of the
 * reference) with a run-time read length; the reference file:line it follows is
 * cited at its definition in mcom_oracle.c.  Parity is PINNED: tests/test_oracle_golden.py checks
 * this file against tests/golden/ (outputs of the compiled reference, made by
 * tests/golden/make_golden.py through oracle/refdump.cpp).
 */
#ifndef MCOM_ORACLE_H
#define MCOM_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t x, y; } mcomo_mm128;

/* read classes assigned by process_reads (kthread_reads.c:84-225) */
enum { MCOMO_CLS_SKETCH = 0, MCOMO_CLS_ALLA = 1, MCOMO_CLS_ALLT = 2, MCOMO_CLS_ALLN = 3,
       MCOMO_CLS_NEARA = 4, MCOMO_CLS_NEART = 5, MCOMO_CLS_NEARN = 6, MCOMO_CLS_NHEAVY = 7 };

/* ---- a1..a3 ---------------------------------------------------------------------------------- */
uint64_t mcomo_hash64(uint64_t key, uint64_t mask);
void     mcomo_sketch_two(const char *s, int len, int k, uint32_t rid, mcomo_mm128 *out);
/* returns the number of minimizers; writes at most cap of them */
size_t   mcomo_sketch_lh_ori(const char *s, int len, int w, int k, uint32_t rid, mcomo_mm128 *out, size_t cap);
/* batch forms: reads are rows of a dense [n][L] ASCII matrix */
void     mcomo_sketch_two_batch(const char *reads, size_t n, int L, int k, uint32_t rid0, mcomo_mm128 *out);

/* ---- a4 -------------------------------------------------------------------------------------- */
/* classify one read, substitute N in place when it is kept, sketch it.  n_pos receives the positions
 * of 'N' (capacity L); returns the class.  rec is written only for MCOMO_CLS_SKETCH. */
int mcomo_process_read(char *seq, int L, int k, int e, uint32_t rid, mcomo_mm128 *rec,
                       uint32_t *n_pos, int *n_npos);
void mcomo_process_reads_batch(char *reads, size_t n, int L, int k, int e, uint32_t rid0,
                               uint8_t *cls, mcomo_mm128 *rec, uint16_t *n_cnt);

/* ---- a5 -------------------------------------------------------------------------------------- */
void mcomo_radix_sort_128x(mcomo_mm128 *beg, mcomo_mm128 *end);

/* ---- a9 / a14 ------------------------------------------------------------------------------- */
int mcomo_match_pro(const char *s0, const char *s1, int i, int j);
int mcomo_encode_byte(const char *seq, const char *ref, int pos, int dir, int L);

/* ---- a10 / a15 ------------------------------------------------------------------------------- */
/* reference bit layout: base i at bits 2i (G, T set it) and 2i+1 (C, T set it) */
void mcomo_string_to_bits(const char *s, int L, uint64_t *w);
int  mcomo_dict_layout(int L, int ininumdict, int *start, int *end); /* returns numdict_s */

/* ---- whole path, staged ---------------------------------------------------------------------- */
typedef struct mcomo_ctx mcomo_ctx;
typedef struct {
	int k;          /* 0 = default (31, or 17 when L < 80)   minicommain.c:92-114 */
	int e;          /* 0 = default 4                          minicommain.c:60 */
	int m;          /* 0 = default 6                          minicommain.c:63 */
	int w;          /* 0 = default L/2-k (3 when L < 70)      preprocess.c:89-107 */
	int cbthr;      /* 0 = default 2e                         minicommain.c:122-126 */
	int max_rounds; /* 0 = default 35                         minicommain.c:64 */
	int step;       /* 0 = default e (5 when e > 10)          minicommain.c:130-137 */
	int maxthr;     /* 0 = default L/2                        minicommain.c:140-143 */
	int numdict;    /* 0 = default                            kthread_hash_realign.c:153-171 */
} mcomo_params;

mcomo_ctx *mcomo_new(const char *reads, size_t n, int L, const mcomo_params *p);
void mcomo_free(mcomo_ctx *c);
void mcomo_force_maxsearch(mcomo_ctx *c, int v);   /* test hook, see mcom_oracle.c */
void mcomo_stage_reads(mcomo_ctx *c);     /* kt_for_reads      kthread_reads.c:247 */
void mcomo_stage_bucket(mcomo_ctx *c);    /* kt_for_bucket     kthread_bucket.c:562 */
void mcomo_stage_combine(mcomo_ctx *c);   /* combine_cluster   kthread_cb.c:570 */
/* one Stage-2 pass: updateSingle + realign_hash (preprocess.c:203-204); returns reads in contigs */
long mcomo_stage_realign_pass(mcomo_ctx *c, int thr);
void mcomo_update_single(mcomo_ctx *c);   /* preprocess.c:243 */
void mcomo_stage_realign_all(mcomo_ctx *c);  /* every Stage-2 pass + the closing updateSingle   preprocess.c:197-232 */
void mcomo_set_log(FILE *f);              /* progress lines of a long run, one per Stage-2 pass */
/* runs everything and writes the same text as "refdump stages" */
int  mcomo_dump_stages(mcomo_ctx *c, const char *path);
/* runs everything without dumping (CPU baseline); fills counters */
void mcomo_run_all(mcomo_ctx *c);

/* accessors used by the GPU parity tests */
size_t mcomo_n_reads(const mcomo_ctx *c);
const char *mcomo_seq(const mcomo_ctx *c);            /* [n][L+1], N substituted */
const uint8_t *mcomo_cls(const mcomo_ctx *c);
const mcomo_mm128 *mcomo_rec0(const mcomo_ctx *c);    /* round-0 records, by rid */
size_t mcomo_n_sg(const mcomo_ctx *c);
const uint32_t *mcomo_sg(const mcomo_ctx *c);
const uint8_t *mcomo_sg_flag(const mcomo_ctx *c);
size_t mcomo_n_contigs(const mcomo_ctx *c);
const char *mcomo_contig_ref(const mcomo_ctx *c, size_t i);
size_t mcomo_contig_n(const mcomo_ctx *c, size_t i);
const uint64_t *mcomo_contig_members(const mcomo_ctx *c, size_t i);
size_t mcomo_counter(const mcomo_ctx *c, const char *name);
const uint32_t *mcomo_list(const mcomo_ctx *c, const char *name, size_t *n);

/* construct_ref2 (kthread_cb.c:105-218) on its own: members [m] are sorted in place (cmpcluster2, :107), reads [n][L]; returns the
 * consensus' length (ref NUL-terminated, cap bytes) or -1 */
long mcomo_construct_ref2(const char *reads, size_t n, int L, uint64_t *members, size_t m, char *ref, size_t cap);

/* the product's mcomh_result_digest restated over the oracle's contig set (include/mcom_host.h): equal digests = equal
 * contig strings, member lists, offsets and id lists */
void mcomo_result_digest(const mcomo_ctx *c, uint64_t out[8]);

/* ---- synthetic reads (same generator as minicom_amd/synth.py, plumbing=False) ------------------ */
void mcomo_synth_reads(uint64_t seed, uint64_t n_reads, int L, int coverage, double sub_rate,
                       uint64_t first, uint64_t count, char *out /* [count][L] */);
mcomo_ctx *mcomo_new_synth(uint64_t seed, size_t n, int L, int coverage, double sub_rate, const mcomo_params *p);

#ifdef __cplusplus
}
#endif
#endif
