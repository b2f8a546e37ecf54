/* oracle/mcom_oracle.c -- CPU restatement of minicom's hot path at one thread.
 * TEST INFRASTRUCTURE ONLY (see mcom_oracle.h).  Pinned against tests/golden/ (reference outputs).
 * All "file:line" citations are into /root/reference/src.
 */
#include "mcom_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <inttypes.h>

#define U64MAX UINT64_MAX

/* ------------------------------------------------------------------------------------------------
 * small containers
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint64_t *a; size_t n, m; } v64;
typedef struct { uint32_t *a; size_t n, m; } v32;
typedef struct { mcomo_mm128 *a; size_t n, m; } v128;
typedef struct { uint64_t *a; size_t n, m; char *ref; } contig_t; /* a[] = rid<<32 | offset<<1 | dir */
typedef struct { contig_t *a; size_t n, m; } vcontig;

#define VPUSH(T, v, val) do { if ((v).n == (v).m) { (v).m = (v).m ? (v).m * 2 : 4; \
	(v).a = (T*)realloc((v).a, (v).m * sizeof(T)); } (v).a[(v).n++] = (val); } while (0)

static void contig_free(contig_t *c) { free(c->a); free(c->ref); c->a = 0; c->ref = 0; c->n = c->m = 0; }

/* A=0 C=1 G=2 T=3 (either case), everything else 4: sketch.c:8-25 */
static uint8_t NT4[256];
static char RC[256];
static int tables_ready = 0;
static void init_tables(void)
{
	if (tables_ready) return;
	memset(NT4, 4, sizeof NT4);
	NT4[0] = 0; NT4[1] = 1; NT4[2] = 2; NT4[3] = 3; /* sketch.c:9 maps raw 0..3 too */
	NT4['A'] = NT4['a'] = 0; NT4['C'] = NT4['c'] = 1; NT4['G'] = NT4['g'] = 2; NT4['T'] = NT4['t'] = 3;
	memset(RC, 0, sizeof RC);
	RC['A'] = 'T'; RC['T'] = 'A'; RC['G'] = 'C'; RC['C'] = 'G'; RC['N'] = 'N'; /* preprocess.c:25-29 */
	tables_ready = 1;
}

/* ------------------------------------------------------------------------------------------------
 * a1  hash64                                                              sketch.c:27-37
 * ---------------------------------------------------------------------------------------------- */
uint64_t mcomo_hash64(uint64_t key, uint64_t mask)
{
	key = (~key + (key << 21)) & mask;
	key ^= key >> 24;
	key = (key + (key << 3) + (key << 8)) & mask;
	key ^= key >> 14;
	key = (key + (key << 2) + (key << 4)) & mask;
	key ^= key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}

/* ------------------------------------------------------------------------------------------------
 * a2  mm_sketch_two: one minimizer per read                               sketch.c:238-289
 *   - no ambiguous-base branch (callers substituted N already)
 *   - a k-mer equal to its reverse complement is skipped without touching the run counter
 *   - strict '<' keeps the leftmost minimum
 * ---------------------------------------------------------------------------------------------- */
void mcomo_sketch_two(const char *s, int len, int k, uint32_t rid, mcomo_mm128 *out)
{
	init_tables();
	const uint64_t shift1 = 2 * (uint64_t)(k - 1), mask = (1ULL << (2 * k)) - 1;
	uint64_t fwd = 0, rev = 0;
	mcomo_mm128 best = { U64MAX, U64MAX };
	int run = 0;
	for (int i = 0; i < len; ++i) {
		uint64_t c = NT4[(uint8_t)s[i]];
		fwd = (fwd << 2 | c) & mask;
		rev = (rev >> 2) | (3ULL ^ c) << shift1;
		if (fwd == rev) continue;
		int z = fwd < rev ? 0 : 1;
		if (++run >= k) {
			uint64_t h = mcomo_hash64(z ? rev : fwd, mask);
			if (h < best.x) { best.x = h; best.y = (uint64_t)rid << 32 | (uint32_t)i << 1 | (uint64_t)z; }
		}
	}
	*out = best;
}

void mcomo_sketch_two_batch(const char *reads, size_t n, int L, int k, uint32_t rid0, mcomo_mm128 *out)
{
	for (size_t i = 0; i < n; ++i) mcomo_sketch_two(reads + i * (size_t)L, L, k, rid0 + (uint32_t)i, &out[i]);
}

/* ------------------------------------------------------------------------------------------------
 * a3  mm_sketch_lh_ori: (w,k)-minimizers of a contig                      sketch.c:116-165
 * ---------------------------------------------------------------------------------------------- */
size_t mcomo_sketch_lh_ori(const char *s, int len, int w, int k, uint32_t rid, mcomo_mm128 *out, size_t cap)
{
	init_tables();
	const uint64_t shift1 = 2 * (uint64_t)(k - 1), mask = (1ULL << (2 * k)) - 1;
	uint64_t fwd = 0, rev = 0;
	mcomo_mm128 *ring = (mcomo_mm128*)malloc((size_t)w * sizeof *ring);
	memset(ring, 0xff, (size_t)w * sizeof *ring);
	mcomo_mm128 best = { U64MAX, U64MAX };
	int run = 0, slot = 0, best_slot = 0;
	size_t n = 0;
#define EMIT(v) do { if (n < cap) out[n] = (v); ++n; } while (0)
	for (int i = 0; i < len; ++i) {
		int c = NT4[(uint8_t)s[i]];
		mcomo_mm128 cur = { U64MAX, U64MAX };
		if (c < 4) {
			fwd = (fwd << 2 | (uint64_t)c) & mask;
			rev = (rev >> 2) | (3ULL ^ (uint64_t)c) << shift1;
			if (fwd == rev) continue;            /* before the ring store and before slot advances */
			int z = fwd < rev ? 0 : 1;
			if (++run >= k) {
				cur.x = mcomo_hash64(z ? rev : fwd, mask);
				cur.y = (uint64_t)rid << 32 | (uint32_t)i << 1 | (uint64_t)z;
			}
		} else run = 0;
		ring[slot] = cur;
		if (run == w + k - 1) {                   /* first full window: earlier copies of the minimum */
			for (int j = slot + 1; j < w; ++j) if (best.x == ring[j].x && ring[j].y != best.y) EMIT(ring[j]);
			for (int j = 0; j < slot; ++j)     if (best.x == ring[j].x && ring[j].y != best.y) EMIT(ring[j]);
		}
		if (cur.x <= best.x) {                    /* '<=': the rightmost of equal hashes wins */
			if (run >= w + k) EMIT(best);
			best = cur; best_slot = slot;
		} else if (slot == best_slot) {           /* the minimum has just left the window */
			if (run >= w + k - 1) EMIT(best);
			best.x = U64MAX;
			for (int j = slot + 1; j < w; ++j) if (best.x >= ring[j].x) { best = ring[j]; best_slot = j; }
			for (int j = 0; j <= slot; ++j)    if (best.x >= ring[j].x) { best = ring[j]; best_slot = j; }
			if (run >= w + k - 1) {
				for (int j = slot + 1; j < w; ++j) if (best.x == ring[j].x && best.y != ring[j].y) EMIT(ring[j]);
				for (int j = 0; j <= slot; ++j)    if (best.x == ring[j].x && best.y != ring[j].y) EMIT(ring[j]);
			}
		}
		if (++slot == w) slot = 0;
	}
	if (best.x != U64MAX) EMIT(best);
#undef EMIT
	free(ring);
	return n;
}

/* ------------------------------------------------------------------------------------------------
 * a4  process_reads: classify, substitute N, sketch                        kthread_reads.c:40-230
 * ---------------------------------------------------------------------------------------------- */
int mcomo_process_read(char *seq, int L, int k, int e, uint32_t rid, mcomo_mm128 *rec,
                       uint32_t *n_pos, int *n_npos)
{
	int cA = 0, cT = 0, cG = 0, cC = 0, cN = 0;
	for (int i = 0; i < L; ++i) {
		switch (seq[i]) {
		case 'A': ++cA; break;
		case 'T': ++cT; break;
		case 'G': ++cG; break;
		case 'C': ++cC; break;
		case 'N': if (n_pos) n_pos[cN] = (uint32_t)i; ++cN; break;
		default: break;
		}
	}
	if (n_npos) *n_npos = cN;
	if (cA == L) return MCOMO_CLS_ALLA;                                   /* :84  */
	if (cT == L) return MCOMO_CLS_ALLT;                                   /* :95  */
	if (cN == L) return MCOMO_CLS_ALLN;                                   /* :106 */
	if (cT + cG + cC + cN <= e) return MCOMO_CLS_NEARA;                   /* :113 */
	if (cA + cG + cC + cN <= e) return MCOMO_CLS_NEART;                   /* :118 */
	if (cA + cT + cG + cC <= e) return MCOMO_CLS_NEARN;                   /* :123 */
	if (!((double)cN <= 0.4 * (double)L)) return MCOMO_CLS_NHEAVY;        /* :182, :219 */
	if (cN > 0) {                                                         /* :183-205, tie order A,T,G,C */
		int mx = cA; if (cT > mx) mx = cT; if (cG > mx) mx = cG; if (cC > mx) mx = cC;
		char rep = 'A';
		if (mx == cA) rep = 'A'; else if (mx == cT) rep = 'T'; else if (mx == cG) rep = 'G'; else rep = 'C';
		for (int i = 0; i < L; ++i) if (seq[i] == 'N') seq[i] = rep;
	}
	mcomo_sketch_two(seq, L, k, rid, rec);                                /* :208 */
	return MCOMO_CLS_SKETCH;
}

void mcomo_process_reads_batch(char *reads, size_t n, int L, int k, int e, uint32_t rid0,
                               uint8_t *cls, mcomo_mm128 *rec, uint16_t *n_cnt)
{
	for (size_t i = 0; i < n; ++i) {
		int nn = 0;
		mcomo_mm128 r = { U64MAX, U64MAX };
		cls[i] = (uint8_t)mcomo_process_read(reads + i * (size_t)L, L, k, e, rid0 + (uint32_t)i, &r, 0, &nn);
		rec[i] = r;
		if (n_cnt) n_cnt[i] = (uint16_t)nn;
	}
}

/* ------------------------------------------------------------------------------------------------
 * a5  radix_sort_128x: American-flag MSD radix sort on .x, 8 bits a pass, insertion sort for
 *     ranges of <= 64 (stable), unstable above                            ksort.h:106-157, misc.c:22
 *     The exact permutation matters: mm_idx consumes equal keys in this order (kthread_idx.c:154).
 * ---------------------------------------------------------------------------------------------- */
static void rs_insertion(mcomo_mm128 *beg, mcomo_mm128 *end)
{
	for (mcomo_mm128 *i = beg + 1; i < end; ++i) {
		if (i->x < (i - 1)->x) {
			mcomo_mm128 t = *i, *j = i;
			while (j > beg && t.x < (j - 1)->x) { *j = *(j - 1); --j; }
			*j = t;
		}
	}
}

static void rs_flag(mcomo_mm128 *beg, mcomo_mm128 *end, int shift)
{
	struct { mcomo_mm128 *b, *e; } bin[256];
	for (int q = 0; q < 256; ++q) bin[q].b = bin[q].e = beg;
	for (mcomo_mm128 *i = beg; i != end; ++i) ++bin[(i->x >> shift) & 255].e;
	for (int q = 1; q < 256; ++q) { bin[q].e += bin[q - 1].e - beg; bin[q].b = bin[q - 1].e; }
	for (int q = 0; q < 256;) {
		if (bin[q].b != bin[q].e) {
			int l = (int)((bin[q].b->x >> shift) & 255);
			if (l != q) {                          /* follow the displacement cycle back to q */
				mcomo_mm128 hold = *bin[q].b, moved;
				do {
					moved = hold; hold = *bin[l].b; *bin[l].b++ = moved;
					l = (int)((hold.x >> shift) & 255);
				} while (l != q);
				*bin[q].b++ = hold;
			} else ++bin[q].b;
		} else ++q;
	}
	bin[0].b = beg;
	for (int q = 1; q < 256; ++q) bin[q].b = bin[q - 1].e;
	if (shift) {
		int next = shift > 8 ? shift - 8 : 0;
		for (int q = 0; q < 256; ++q) {
			ptrdiff_t cnt = bin[q].e - bin[q].b;
			if (cnt > 64) rs_flag(bin[q].b, bin[q].e, next);
			else if (cnt > 1) rs_insertion(bin[q].b, bin[q].e);
		}
	}
}

void mcomo_radix_sort_128x(mcomo_mm128 *beg, mcomo_mm128 *end)
{
	if (end - beg <= 64) rs_insertion(beg, end);
	else rs_flag(beg, end, 56);
}

/* ------------------------------------------------------------------------------------------------
 * stable merge sort on uint64 (glibc 2.35 qsort, which the pinned reference build uses, is a merge
 * sort and therefore stable; cmpcluster2 has ties)
 * ---------------------------------------------------------------------------------------------- */
typedef int (*cmp64_t)(uint64_t a, uint64_t b, const void *ctx);
static void msort64(uint64_t *a, size_t n, cmp64_t cmp, const void *ctx)
{
	if (n < 2) return;
	uint64_t *tmp = (uint64_t*)malloc(n * sizeof *tmp);
	for (size_t wdt = 1; wdt < n; wdt *= 2) {
		for (size_t lo = 0; lo < n; lo += 2 * wdt) {
			size_t mid = lo + wdt < n ? lo + wdt : n, hi = lo + 2 * wdt < n ? lo + 2 * wdt : n;
			size_t i = lo, j = mid, o = lo;
			while (i < mid && j < hi) tmp[o++] = cmp(a[j], a[i], ctx) < 0 ? a[j++] : a[i++];
			while (i < mid) tmp[o++] = a[i++];
			while (j < hi) tmp[o++] = a[j++];
		}
		memcpy(a, tmp, n * sizeof *a);
	}
	free(tmp);
}

/* ------------------------------------------------------------------------------------------------
 * a9  match_pro: mismatches over the whole overlap of two contigs anchored at (i,j)  kthread_cb.c:36-52
 * ---------------------------------------------------------------------------------------------- */
int mcomo_match_pro(const char *s0, const char *s1, int i_, int j_)
{
	int tot = 0, match = 0;
	for (int i = i_, j = j_; s0[i] != '\0' && s1[j] != '\0'; ++i, ++j) { ++tot; if (s0[i] == s1[j]) ++match; }
	for (int i = i_ - 1, j = j_ - 1; i >= 0 && j >= 0; --i, --j) { ++tot; if (s0[i] == s1[j]) ++match; }
	return tot - match;
}

/* ------------------------------------------------------------------------------------------------
 * a14 encode_byte: length estimate of the mismatch text; the match-run counter is NOT reset after a
 *     short run is written literally (kthread_hash_realign.c:301-305)     kthread_hash_realign.c:283-314
 * ---------------------------------------------------------------------------------------------- */
static int ndigits(int v) { int d = 1; while (v >= 10) { v /= 10; ++d; } return d; }

int mcomo_encode_byte(const char *seq, const char *ref, int pos, int dir, int L)
{
	init_tables();
	int len = 0, eq = 0;
	for (int t = 0; t < L; ++t) {
		char c = dir ? RC[(uint8_t)seq[L - 1 - t]] : seq[t];
		if (ref[pos + t] != c) {
			if (eq > 1) { len += ndigits(eq); eq = 0; }
			else len += eq;
			++len;
		} else ++eq;
	}
	if (len == 0) len = 1;
	return (double)len <= (double)L * 0.4;
}

/* ------------------------------------------------------------------------------------------------
 * a10 stringtobitset: base i -> bit 2i set for G,T; bit 2i+1 set for C,T
 *                                          bbhashdict.c:69-74, kthread_hash_realign.c:249-258
 * ---------------------------------------------------------------------------------------------- */
void mcomo_string_to_bits(const char *s, int L, uint64_t *w)
{
	int W = (2 * L + 63) / 64;
	memset(w, 0, (size_t)W * 8);
	for (int i = 0; i < L; ++i) {
		uint64_t v;
		switch (s[i]) { case 'A': v = 0; break; case 'G': v = 1; break; case 'C': v = 2; break; case 'T': v = 3; break; default: v = 0; }
		w[(2 * i) >> 6] |= v << ((2 * i) & 63);
	}
}

/* a15 dictionary layout                                               kthread_hash_realign.c:153-206 */
int mcomo_dict_layout(int L, int ininumdict, int *start, int *end)
{
	int len_t = L <= 80 ? 11 : 17;
	int nd = L / len_t;
	if (ininumdict > 1 && ininumdict < nd) nd = ininumdict;
	start[0] = (ininumdict > 0 && ininumdict < nd) ? L / 2 - (len_t * nd) / 2 : 0;
	end[0] = start[0] + len_t - 1;
	for (int i = 1; i < nd; ++i) { start[i] = end[i - 1] + 1; end[i] = start[i] + len_t - 1; }
	return nd;
}

/* ================================================================================================
 * whole path, staged
 * ============================================================================================== */
#define NB_BITS 14
#define NBUCKET (1 << NB_BITS)

struct mcomo_ctx {
	size_t n; int L, W;
	int k, e, m, rw, cbthr, max_rounds, step, maxthr, numdict_param;
	int maxsearch, maxsearch_forced;
	char *seq;              /* [n][L+1] */
	uint8_t *cls;
	mcomo_mm128 *rec0;
	v32 *npos;              /* per read, positions of N */
	v32 allA, allT, allN, fpA, fpT, fpN, Nfile, sg;
	uint8_t *sg_flag;
	v128 *B[2];             /* read buckets, double buffered          preprocess.c:114-119 */
	v128 *MI[2];            /* contig-minimizer buckets (mm_idx_t.B[].a) */
	vcontig C[2];
	int idxv;
	/* counters for the measurement report */
	size_t cnt_rounds, cnt_windows, cnt_lookups, cnt_passes, cnt_merge_rounds, cnt_resketch, cnt_cand;
};

static char *rd(const mcomo_ctx *c, uint32_t rid) { return c->seq + (size_t)rid * (size_t)(c->L + 1); }

mcomo_ctx *mcomo_new(const char *reads, size_t n, int L, const mcomo_params *p)
{
	init_tables();
	mcomo_params z; memset(&z, 0, sizeof z);
	if (!p) p = &z;
	mcomo_ctx *c = (mcomo_ctx*)calloc(1, sizeof *c);
	c->n = n; c->L = L; c->W = (2 * L + 63) / 64;
	c->k = p->k > 0 ? p->k : (L < 80 ? 17 : 31);
	c->e = p->e > 0 ? p->e : 4;
	c->m = p->m > 0 ? p->m : 6;
	c->cbthr = p->cbthr > 0 ? p->cbthr : 2 * c->e;
	c->max_rounds = (p->max_rounds > 0 && p->max_rounds < 35) ? p->max_rounds : 35;
	c->step = p->step > 0 ? p->step : (c->e > 10 ? 5 : c->e);
	c->maxthr = p->maxthr > 0 ? p->maxthr : L / 2;
	c->rw = L >= 70 ? L / 2 - c->k : 3;
	if (p->w > 0) c->rw = p->w;
	c->numdict_param = p->numdict;
	c->maxsearch = 500;                                                   /* minicommain.c:77 */
	c->seq = (char*)malloc(n * (size_t)(L + 1));
	for (size_t i = 0; i < n; ++i) { memcpy(c->seq + i * (size_t)(L + 1), reads + i * (size_t)L, (size_t)L); c->seq[i * (size_t)(L + 1) + L] = 0; }
	c->cls = (uint8_t*)calloc(n ? n : 1, 1);
	c->rec0 = (mcomo_mm128*)calloc(n ? n : 1, sizeof *c->rec0);
	c->npos = (v32*)calloc(n ? n : 1, sizeof *c->npos);
	for (int i = 0; i < 2; ++i) {
		c->B[i] = (v128*)calloc(NBUCKET, sizeof(v128));
		c->MI[i] = (v128*)calloc(NBUCKET, sizeof(v128));
	}
	return c;
}

static void buckets_clear(v128 *B) { for (int i = 0; i < NBUCKET; ++i) { free(B[i].a); B[i].a = 0; B[i].n = B[i].m = 0; } }
static void contigs_clear(vcontig *v) { for (size_t i = 0; i < v->n; ++i) contig_free(&v->a[i]); v->n = 0; }

void mcomo_free(mcomo_ctx *c)
{
	if (!c) return;
	for (size_t i = 0; i < c->n; ++i) free(c->npos[i].a);
	for (int i = 0; i < 2; ++i) { buckets_clear(c->B[i]); buckets_clear(c->MI[i]); free(c->B[i]); free(c->MI[i]); contigs_clear(&c->C[i]); free(c->C[i].a); }
	free(c->allA.a); free(c->allT.a); free(c->allN.a); free(c->fpA.a); free(c->fpT.a); free(c->fpN.a); free(c->Nfile.a); free(c->sg.a);
	free(c->sg_flag); free(c->npos); free(c->rec0); free(c->cls); free(c->seq); free(c);
}

/* ---- kt_for_reads at one thread: reads in rid order                  kthread_reads.c:247, :40-230 */
void mcomo_stage_reads(mcomo_ctx *c)
{
	uint32_t *tmp = (uint32_t*)malloc((size_t)c->L * sizeof *tmp);
	for (size_t r = 0; r < c->n; ++r) {
		int nn = 0;
		mcomo_mm128 rec = { U64MAX, U64MAX };
		int cl = mcomo_process_read(rd(c, (uint32_t)r), c->L, c->k, c->e, (uint32_t)r, &rec, tmp, &nn);
		c->cls[r] = (uint8_t)cl; c->rec0[r] = rec;
		for (int i = 0; i < nn; ++i) VPUSH(uint32_t, c->npos[r], tmp[i]);
		switch (cl) {
		case MCOMO_CLS_ALLA: VPUSH(uint32_t, c->allA, (uint32_t)r); break;
		case MCOMO_CLS_ALLT: VPUSH(uint32_t, c->allT, (uint32_t)r); break;
		case MCOMO_CLS_ALLN: VPUSH(uint32_t, c->allN, (uint32_t)r); break;
		case MCOMO_CLS_NEARA: VPUSH(uint32_t, c->fpA, (uint32_t)r); break;
		case MCOMO_CLS_NEART: VPUSH(uint32_t, c->fpT, (uint32_t)r); break;
		case MCOMO_CLS_NEARN: VPUSH(uint32_t, c->fpN, (uint32_t)r); break;
		case MCOMO_CLS_NHEAVY: VPUSH(uint32_t, c->Nfile, (uint32_t)r); break;
		default: VPUSH(mcomo_mm128, c->B[0][rec.x & (NBUCKET - 1)], rec); break;
		}
	}
	free(tmp);
}

/* ---- cmpcluster: aligned start position descending, then rid ascending   kthread_bucket.c:44-62
 *      (uses the ORIGINAL k of the run even in later rounds) */
typedef struct { int L, k; } cmpc_ctx;
static int cmp_cluster(uint64_t a, uint64_t b, const void *ctx_)
{
	const cmpc_ctx *x = (const cmpc_ctx*)ctx_;
	int ra = (int)(a >> 32), rb = (int)(b >> 32);
	int pa = (int)((uint32_t)a >> 1), pb = (int)((uint32_t)b >> 1);
	if (a & 1) pa = x->L - pa + x->k - 2;
	if (b & 1) pb = x->L - pb + x->k - 2;
	if (pa == pb) return ra - rb;
	return pb - pa;
}
/* cmpcluster2: offset ascending, then direction                           kthread_cb.c:54-69 */
static int cmp_cluster2(uint64_t a, uint64_t b, const void *ctx_)
{
	(void)ctx_;
	int pa = (int)((uint32_t)a >> 1), pb = (int)((uint32_t)b >> 1);
	if (pa == pb) return (int)(a & 1) - (int)(b & 1);
	return pa - pb;
}

static void oriented(const mcomo_ctx *c, uint32_t rid, int dir, char *out)
{
	const char *s = rd(c, rid);
	int L = c->L;
	if (!dir) memcpy(out, s, (size_t)L);
	else for (int i = 0; i < L; ++i) out[i] = RC[(uint8_t)s[L - 1 - i]];
	out[L] = 0;
}

static void push_first_minimizers(mcomo_ctx *c, v128 *MI, const char *ref, int w, uint32_t id)
{
	size_t len = strlen(ref);
	size_t cap = len + 8;
	mcomo_mm128 *mz = (mcomo_mm128*)malloc(cap * sizeof *mz);
	size_t nm = mcomo_sketch_lh_ori(ref, (int)len, w, c->k, id, mz, cap);
	if (nm > cap) nm = cap;
	for (size_t q = 0; q < nm && q < (size_t)c->m; ++q) VPUSH(mcomo_mm128, MI[mz[q].x & (NBUCKET - 1)], mz[q]);
	free(mz);
}

static void resketch_or_single(mcomo_ctx *c, uint32_t rid, int index, int kmer, int last)
{
	if (last) { VPUSH(uint32_t, c->sg, rid); return; }
	mcomo_mm128 r;
	mcomo_sketch_two(rd(c, rid), c->L, kmer, rid, &r);
	++c->cnt_resketch;
	VPUSH(mcomo_mm128, c->B[index ^ 1][r.x & (NBUCKET - 1)], r);
}

/* ---- construct_ref: consensus of one minimizer group, reject far members   kthread_bucket.c:69-377 */
static void construct_ref(mcomo_ctx *c, contig_t *p, size_t n, int index, int kmer, int last)
{
	const int L = c->L;
	const int tlen = L << 2;                    /* "readlen<<1 + 1" parses as readlen << 2   :72 */
	uint32_t *cnt = (uint32_t*)calloc((size_t)4 * tlen, sizeof *cnt);
	char *t = (char*)malloc((size_t)L + 1);
	int pos0 = L;
	for (size_t q = 0; q < n; ++q) {
		uint64_t y = p->a[q];
		uint32_t rid = (uint32_t)(y >> 32); int pos = (int)((uint32_t)y >> 1), dir = (int)(y & 1);
		oriented(c, rid, dir, t);
		if (dir) pos = L - pos + c->k - 2;
		if (q == 0) pos0 = pos;
		for (int s = 0; s < L; ++s) ++cnt[NT4[(uint8_t)t[s]] * tlen + pos0 - pos + s];
		p->a[q] = (y >> 32 << 32) | ((uint64_t)(pos0 - pos) << 1) | (uint64_t)dir;
	}
	char *ref = (char*)calloc((size_t)tlen + 1, 1);
	for (int s = 0; s < tlen; ++s) {
		uint32_t mx = cnt[s]; ref[s] = 'A';
		for (int b = 1; b < 4; ++b) if (cnt[b * tlen + s] > mx) { mx = cnt[b * tlen + s]; ref[s] = "ACGT"[b]; }
		if (mx == 0) { ref[s] = 0; break; }
	}
	int ref_len = (int)strlen(ref);
	p->ref = (char*)calloc((size_t)ref_len + 1, 1);
	strcpy(p->ref, ref);
	p->n = 0;
	for (size_t q = 0; q < n; ++q) {
		uint64_t y = p->a[q];
		uint32_t rid = (uint32_t)(y >> 32); int pos = (int)((uint32_t)y >> 1), dir = (int)(y & 1);
		oriented(c, rid, dir, t);
		int dif = 0;
		for (int s = 0; s < L; ++s) if (ref[pos + s] != t[s]) ++dif;
		if (dif <= c->e) p->a[p->n++] = y;                               /* :189 */
		else resketch_or_single(c, rid, index, kmer, last);             /* :194-213 */
	}
	if (p->n > 0) {                                                      /* :244-352 */
		memset(cnt, 0, (size_t)4 * tlen * sizeof *cnt);
		int rend = 0;
		for (size_t q = 0; q < p->n; ++q) {
			uint64_t y = p->a[q];
			uint32_t rid = (uint32_t)(y >> 32); int pos = (int)((uint32_t)y >> 1), dir = (int)(y & 1);
			oriented(c, rid, dir, t);
			for (int s = 0; s < L; ++s) ++cnt[NT4[(uint8_t)t[s]] * tlen + pos + s];
			if (pos + L > rend) rend = pos + L;
		}
		int s = 0;
		for (; s < ref_len; ++s) {
			uint32_t mx = cnt[s];
			for (int b = 1; b < 4; ++b) if (cnt[b * tlen + s] > mx) mx = cnt[b * tlen + s];
			if (mx != 0) break;
		}
		int sv = s, r = 0;
		for (; s < rend; ++s, ++r) {
			uint32_t mx = cnt[s]; p->ref[r] = 'A';
			for (int b = 1; b < 4; ++b) if (cnt[b * tlen + s] > mx) { mx = cnt[b * tlen + s]; p->ref[r] = "ACGT"[b]; }
		}
		p->ref[r] = 0;
		for (size_t q = 0; q < p->n; ++q) {
			uint64_t y = p->a[q];
			int pos = (int)((uint32_t)y >> 1);
			p->a[q] = (y >> 32 << 32) | ((uint64_t)(pos - sv) << 1) | (y & 1);
		}
	}
	free(t); free(ref); free(cnt);
}

/* ---- process_bucket                                                      kthread_bucket.c:381-509 */
static void process_bucket(mcomo_ctx *c, int i, int index, int kmer, int last)
{
	v128 *b = &c->B[index][i];
	if (b->n == 0) return;
	mcomo_radix_sort_128x(b->a, b->a + b->n);
	cmpc_ctx cc = { c->L, c->k };
	size_t start = 0;
	for (size_t j = 1; j <= b->n; ++j) {
		if (j < b->n && b->a[j].x == b->a[j - 1].x) continue;
		size_t n = j - start;
		if (n < 2) {
			VPUSH(uint32_t, c->sg, (uint32_t)(b->a[start].y >> 32));      /* :402-413 */
		} else {
			vcontig *cv = &c->C[0];
			contig_t fresh; memset(&fresh, 0, sizeof fresh);
			VPUSH(contig_t, *cv, fresh);
			contig_t *p = &cv->a[cv->n - 1];
			p->a = (uint64_t*)malloc(n * sizeof(uint64_t)); p->m = n; p->n = n;
			for (size_t q = 0; q < n; ++q) p->a[q] = b->a[start + q].y;
			msort64(p->a, n, cmp_cluster, &cc);                           /* :442 */
			construct_ref(c, p, n, index, kmer, last);                    /* :446 */
			if (p->n > 1) {
				push_first_minimizers(c, c->MI[0], p->ref, c->rw, (uint32_t)(((cv->n - 1) << 8) + 0)); /* :458-474 */
			} else {
				if (p->n == 1) resketch_or_single(c, (uint32_t)(p->a[0] >> 32), index, kmer, last); /* :477-498 */
				contig_free(p);
				--cv->n;                                                  /* :499-500 */
			}
		}
		start = j;
	}
	free(b->a); b->a = 0; b->n = b->m = 0;
}

/* ---- kt_for_bucket: Stage-1 rounds                                        kthread_bucket.c:562-629 */
void mcomo_stage_bucket(mcomo_ctx *c)
{
	int index = 0, last_rounds = 0;
	long pre = 0;
	for (int r = 1;; ++r) {
		buckets_clear(c->B[index ^ 1]);
		if (c->k - r <= 9) ++last_rounds;
		if (r == c->max_rounds - 1) ++last_rounds;
		for (int i = 0; i < NBUCKET; ++i) process_bucket(c, i, index, c->k - r, last_rounds != 0);
		++c->cnt_rounds;
		if (last_rounds) ++last_rounds;
		buckets_clear(c->B[index]);
		index ^= 1;
		long cr = 0;
		for (size_t i = 0; i < c->C[0].n; ++i) cr += (long)c->C[0].a[i].n;
		if (cr - pre < 100) ++last_rounds;
		pre = cr;
		if (last_rounds > 1) break;
	}
	buckets_clear(c->B[index]);
	if (c->sg.n <= 5000000) c->maxsearch = 2000;                          /* preprocess.c:169-172 */
	if (c->maxsearch_forced > 0) c->maxsearch = c->maxsearch_forced;
}

/* ---- mm_idx: sort each bucket; equal keys keep the order radix_sort_128x leaves them in
 *                                                                          kthread_idx.c:116-168 */
static void idx_generate(v128 *MI)
{
	for (int i = 0; i < NBUCKET; ++i) if (MI[i].n) mcomo_radix_sort_128x(MI[i].a, MI[i].a + MI[i].n);
}
/* mm_idx_get: all y stored for minimizer x, in index order                   kthread_idx.c:84-101 */
static const mcomo_mm128 *idx_get(const v128 *MI, uint64_t x, int *n)
{
	const v128 *b = &MI[x & (NBUCKET - 1)];
	*n = 0;
	size_t lo = 0, hi = b->n;
	while (lo < hi) { size_t mid = (lo + hi) / 2; if (b->a[mid].x < x) lo = mid + 1; else hi = mid; }
	size_t e = lo;
	while (e < b->n && b->a[e].x == x) ++e;
	*n = (int)(e - lo);
	return *n ? &b->a[lo] : 0;
}

/* ---- construct_ref2: consensus of a merged contig                          kthread_cb.c:105-218 */
static void construct_ref2(mcomo_ctx *c, contig_t *p)
{
	const int L = c->L;
	msort64(p->a, p->n, cmp_cluster2, 0);
	int tot_len = (int)((uint32_t)p->a[p->n - 1] >> 1) + (L << 1);
	int tlen = tot_len + 1;
	uint32_t *cnt = (uint32_t*)calloc((size_t)4 * tlen, sizeof *cnt);
	char *t = (char*)malloc((size_t)L + 1);
	int rend = 0;
	for (size_t q = 0; q < p->n; ++q) {
		uint64_t y = p->a[q];
		uint32_t rid = (uint32_t)(y >> 32); int pos = (int)((uint32_t)y >> 1), dir = (int)(y & 1);
		oriented(c, rid, dir, t);
		for (int s = 0; s < L; ++s) ++cnt[NT4[(uint8_t)t[s]] * tlen + pos + s];
		if (pos + L > rend) rend = pos + L;
	}
	p->ref = (char*)calloc((size_t)rend + 1, 1);
	for (int s = 0; s < rend; ++s) {
		uint32_t mx = cnt[s]; p->ref[s] = 'A';
		for (int b = 1; b < 4; ++b) if (cnt[b * tlen + s] > mx) { mx = cnt[b * tlen + s]; p->ref[s] = "ACGT"[b]; }
	}
	free(t); free(cnt);
}

/* construct_ref2 alone, for a test that has members and reads but no pipeline around them: members [m] (rid << 32 | offset << 1 | dir, sorted
 * in place by cmpcluster2 as :107 does), reads [n][L]; the consensus goes to ref (NUL-terminated; cap bytes); returns its length or -1 */
long mcomo_construct_ref2(const char *reads, size_t n, int L, uint64_t *members, size_t m, char *ref, size_t cap)
{
	if (!m) return -1;
	init_tables();
	mcomo_ctx c; memset(&c, 0, sizeof c);
	c.n = n; c.L = L;
	c.seq = (char*)malloc(n * (size_t)(L + 1));
	for (size_t i = 0; i < n; ++i) { memcpy(c.seq + i * (size_t)(L + 1), reads + i * (size_t)L, (size_t)L); c.seq[i * (size_t)(L + 1) + L] = 0; }
	contig_t p; memset(&p, 0, sizeof p);
	p.a = members; p.n = p.m = m;
	construct_ref2(&c, &p);
	const size_t len = strlen(p.ref);
	long out = -1;
	if (len + 1 <= cap) { memcpy(ref, p.ref, len + 1); out = (long)len; }
	free(p.ref); free(c.seq);
	return out;
}

/* ---- one merge round: find_next for every unflagged contig, then copy the rest
 *                                                     kthread_cb.c:220-395, :397-434, :502-568 */
static void merge_round(mcomo_ctx *c, int index)
{
	vcontig *src = &c->C[index], *dst = &c->C[index ^ 1];
	v128 *mi = c->MI[index], *mo = c->MI[index ^ 1];
	uint8_t *flag = (uint8_t*)calloc(src->n ? src->n : 1, 1);
	for (size_t i = 0; i < src->n; ++i) {
		if (flag[i]) continue;
		contig_t *p = &src->a[i];
		size_t len = strlen(p->ref), cap = len + 8;
		mcomo_mm128 *mz = (mcomo_mm128*)malloc(cap * sizeof *mz);
		size_t nm = mcomo_sketch_lh_ori(p->ref, (int)len, c->rw, c->k, (uint32_t)(i << 8), mz, cap);
		uint32_t rid_ori = (uint32_t)(i << 8);
		int merged = 0;
		for (size_t j = 0; j < nm && !merged; ++j) {
			int nh; const mcomo_mm128 *h = idx_get(mi, mz[j].x, &nh);
			uint32_t pos_ori = (uint32_t)mz[j].y >> 1, dir_ori = (uint32_t)(mz[j].y & 1);
			for (int q = 0; q < nh && !merged; ++q) {
				uint32_t rid = (uint32_t)(h[q].y >> 32);
				if (rid == rid_ori) continue;
				size_t cid = rid >> 8;
				uint32_t pos = (uint32_t)h[q].y >> 1, dir = (uint32_t)(h[q].y & 1);
				++c->cnt_cand;
				if (dir != dir_ori || flag[i] || flag[cid]) continue;
				contig_t *o = &src->a[cid];
				if (mcomo_match_pro(p->ref, o->ref, (int)pos_ori, (int)pos) > c->cbthr) continue;
				contig_t t; memset(&t, 0, sizeof t);
				t.m = p->n + o->n; t.a = (uint64_t*)malloc(t.m * sizeof(uint64_t));
				if (pos_ori >= pos) {                                      /* :302-315 */
					for (size_t u = 0; u < p->n; ++u) t.a[t.n++] = p->a[u];
					for (size_t u = 0; u < o->n; ++u) { uint64_t y = o->a[u];
						t.a[t.n++] = (y >> 32 << 32) | (((uint64_t)((uint32_t)y >> 1) + (uint64_t)(pos_ori - pos)) << 1) | (y & 1); }
				} else {                                                   /* :316-325 */
					for (size_t u = 0; u < o->n; ++u) t.a[t.n++] = o->a[u];
					for (size_t u = 0; u < p->n; ++u) { uint64_t y = p->a[u];
						t.a[t.n++] = (y >> 32 << 32) | (((uint64_t)((uint32_t)y >> 1) + (uint64_t)(pos - pos_ori)) << 1) | (y & 1); }
				}
				construct_ref2(c, &t);
				flag[i] = flag[cid] = 1; merged = 1;
				VPUSH(contig_t, *dst, t);
				push_first_minimizers(c, mo, t.ref, c->rw, (uint32_t)((dst->n - 1) << 8));   /* :365-380 */
			}
		}
		free(mz);
	}
	for (size_t i = 0; i < src->n; ++i) {                                   /* cp_cluster  :397-434 */
		if (flag[i]) continue;
		contig_t *p = &src->a[i], t; memset(&t, 0, sizeof t);
		t.n = t.m = p->n; t.a = (uint64_t*)malloc((p->n ? p->n : 1) * sizeof(uint64_t));
		memcpy(t.a, p->a, p->n * sizeof(uint64_t));
		t.ref = (char*)malloc(strlen(p->ref) + 1); strcpy(t.ref, p->ref);
		VPUSH(contig_t, *dst, t);
		push_first_minimizers(c, mo, t.ref, c->rw, (uint32_t)((dst->n - 1) << 8));
	}
	free(flag);
}

/* ---- combine_cluster                                                       kthread_cb.c:570-630 */
void mcomo_stage_combine(mcomo_ctx *c)
{
	int index = 0;
	long pre = 0;
	for (;;) {
		idx_generate(c->MI[index]);
		contigs_clear(&c->C[index ^ 1]);
		buckets_clear(c->MI[index ^ 1]);
		merge_round(c, index);
		++c->cnt_merge_rounds;
		buckets_clear(c->MI[index]);
		contigs_clear(&c->C[index]);
		index ^= 1;
		long tot = (long)c->C[index].n;
		if (labs(pre - tot) < 100) break;
		pre = tot;
	}
	buckets_clear(c->MI[index]);
	c->idxv = index;
	free(c->sg_flag);
	c->sg_flag = (uint8_t*)calloc(c->sg.n ? c->sg.n : 1, 1);               /* preprocess.c:182 */
}

/* ---- updateSingle                                                          preprocess.c:243-255 */
void mcomo_update_single(mcomo_ctx *c)
{
	size_t nn = 0;
	for (size_t i = 0; i < c->sg.n; ++i) if (!c->sg_flag[i]) c->sg.a[nn++] = c->sg.a[i];
	c->sg.n = nn;
	free(c->sg_flag);
	c->sg_flag = (uint8_t*)calloc(nn ? nn : 1, 1);
}

/* ---- Stage-2 dictionaries                      kthread_hash_realign.c:3-140, bbhashdict.c:33-67
 * The reference maps key -> bin with a BBHash MPHF; every hit is re-verified against the bin's key
 * (:385-386), so any exact key -> dense id map gives the same results.  Here: rank in sorted keys. */
typedef struct {
	uint64_t *keys; uint32_t numkeys;
	uint32_t *startpos, *read_id; uint8_t *empty_bin;
} dict_t;

static uint64_t bits_key(const uint64_t *w, int start, int len)
{
	int bo = 2 * start, wi = bo >> 6, sh = bo & 63;
	uint64_t v = w[wi] >> sh;
	if (sh && sh + 2 * len > 64) v |= w[wi + 1] << (64 - sh);
	return v & ((1ULL << (2 * len)) - 1);
}
static int cmp_u64(const void *a, const void *b) { uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return x < y ? -1 : x > y; }

static int64_t dict_lookup(const dict_t *d, uint64_t key)
{
	size_t lo = 0, hi = d->numkeys;
	while (lo < hi) { size_t mid = (lo + hi) / 2; if (d->keys[mid] < key) lo = mid + 1; else hi = mid; }
	return (lo < d->numkeys && d->keys[lo] == key) ? (int64_t)lo : -1;
}
/* findpos                                                                    bbhashdict.c:33-43 */
static void dict_findpos(const dict_t *d, uint32_t numreads, uint32_t bin, int64_t *lo, int64_t *hi)
{
	*lo = d->startpos[bin];
	uint32_t end = d->startpos[bin + 1];
	if (d->read_id[end - 1] == numreads) *hi = end - 1;
	else if (d->read_id[end - 1] == numreads + 1) *hi = *lo + d->read_id[end - 2];
	else *hi = end;
}
/* remove                                                                     bbhashdict.c:45-67 */
static void dict_remove(dict_t *d, uint32_t numreads, uint32_t bin, int64_t lo, int64_t hi, uint32_t cur)
{
	int64_t size = hi - lo;
	if (size == 1) { d->empty_bin[bin] = 1; return; }
	int64_t a = lo, b = hi;
	while (a < b) { int64_t mid = (a + b) / 2; if (d->read_id[mid] < cur) a = mid + 1; else b = mid; }
	memmove(d->read_id + a, d->read_id + a + 1, (size_t)(hi - a - 1) * sizeof(uint32_t));
	uint32_t end = d->startpos[bin + 1];
	if (hi == end) d->read_id[end - 1] = numreads;
	else if (d->read_id[end - 1] == numreads) { d->read_id[end - 1] = numreads + 1; d->read_id[end - 2] = (uint32_t)(size - 1); }
	else d->read_id[end - 2]--;
}

static int popc_xor(const uint64_t *a, const uint64_t *b, int W)
{
	int s = 0;
	for (int i = 0; i < W; ++i) s += __builtin_popcountll(a[i] ^ b[i]);
	return s;
}

/* run-length text length of a read against a constant base            bbhashdict.c:158-176, :192-210 */
static int const_base_len(const char *s, int L, char base)
{
	int len = 0, eq = 0;
	for (int t = 0; t < L; ++t) {
		if (s[t] != base) { if (eq > 0) { len += ndigits(eq); eq = 0; } ++len; }
		else ++eq;
	}
	return len ? len : 1;
}

/* test hook: the reference fixes maxsearch at 500 / 2000 (minicommain.c:77, preprocess.c:169-172); a small value lets a
 * small input exercise the cut of long bins (kthread_hash_realign.c:388) */
void mcomo_force_maxsearch(mcomo_ctx *c, int v) { c->maxsearch_forced = v; if (v > 0) c->maxsearch = v; }

/* ---- realign_hash                                                 kthread_hash_realign.c:569-594 */
long mcomo_stage_realign_pass(mcomo_ctx *c, int thr)
{
	mcomo_update_single(c);                                                /* preprocess.c:203 */
	const int L = c->L, W = c->W;
	const uint32_t numreads = (uint32_t)c->sg.n;
	int ds[64], de[64];
	const int nd = mcomo_dict_layout(L, c->numdict_param, ds, de);
	++c->cnt_passes;
	/* singleRead2bitset                                                     bbhashdict.c:127-227 */
	uint64_t *bits = (uint64_t*)calloc((size_t)(numreads ? numreads : 1) * W, 8);
	uint64_t *allA = (uint64_t*)calloc((size_t)W, 8), *allT = (uint64_t*)calloc((size_t)W, 8);
	char *t = (char*)malloc((size_t)L + 1);
	memset(t, 'A', (size_t)L); mcomo_string_to_bits(t, L, allA);
	memset(t, 'T', (size_t)L); mcomo_string_to_bits(t, L, allT);
	for (uint32_t i = 0; i < numreads; ++i) {
		uint32_t rid = c->sg.a[i];
		uint64_t *b = bits + (size_t)i * W;
		mcomo_string_to_bits(rd(c, rid), L, b);
		int nearA = popc_xor(b, allA, W) <= thr, nearT = !nearA && popc_xor(b, allT, W) <= thr;
		if (nearA || nearT) {
			memcpy(t, rd(c, rid), (size_t)L);
			for (size_t q = 0; q < c->npos[rid].n; ++q) t[c->npos[rid].a[q]] = 'N';
			if ((double)const_base_len(t, L, nearA ? 'A' : 'T') <= (double)L * 0.4) {
				c->sg_flag[i] = 1;
				if (nearA) VPUSH(uint32_t, c->fpA, rid); else VPUSH(uint32_t, c->fpT, rid);
			}
		}
	}
	/* constructdictionary_realign                                     kthread_hash_realign.c:3-140 */
	dict_t *D = (dict_t*)calloc((size_t)nd, sizeof *D);
	for (int j = 0; j < nd; ++j) {
		dict_t *d = &D[j];
		int kl = de[j] - ds[j] + 1;
		uint64_t *all = (uint64_t*)malloc((size_t)(numreads ? numreads : 1) * 8);
		for (uint32_t i = 0; i < numreads; ++i) all[i] = bits_key(bits + (size_t)i * W, ds[j], kl);
		d->keys = (uint64_t*)malloc((size_t)(numreads ? numreads : 1) * 8);
		memcpy(d->keys, all, (size_t)numreads * 8);
		qsort(d->keys, numreads, 8, cmp_u64);
		uint32_t u = 0;
		for (uint32_t i = 0; i < numreads; ++i) if (i == 0 || d->keys[i] != d->keys[u - 1]) d->keys[u++] = d->keys[i];
		d->numkeys = u;
		d->startpos = (uint32_t*)calloc((size_t)u + 2, sizeof(uint32_t));
		d->empty_bin = (uint8_t*)calloc((size_t)u + 1, 1);
		d->read_id = (uint32_t*)malloc((size_t)(numreads ? numreads : 1) * sizeof(uint32_t));
		for (uint32_t i = 0; i < numreads; ++i) d->startpos[dict_lookup(d, all[i]) + 1]++;
		for (uint32_t i = 1; i <= u; ++i) d->startpos[i] += d->startpos[i - 1];
		uint32_t *fill = (uint32_t*)malloc(((size_t)u + 1) * sizeof(uint32_t));
		memcpy(fill, d->startpos, ((size_t)u + 1) * sizeof(uint32_t));
		for (uint32_t i = 0; i < numreads; ++i) d->read_id[fill[dict_lookup(d, all[i])]++] = i;
		free(fill); free(all);
	}
	/* kt_realign_hash_for / realign_hash_search                  kthread_hash_realign.c:316-508 */
	vcontig *cv = &c->C[c->idxv];
	uint64_t *win = (uint64_t*)malloc((size_t)W * 8), *rwin = (uint64_t*)malloc((size_t)W * 8);
	v32 del; memset(&del, 0, sizeof del);
	for (size_t ci = 0; ci < cv->n; ++ci) {
		contig_t *p = &cv->a[ci];
		msort64(p->a, p->n, cmp_cluster2, 0);                              /* :318 */
		int nwin = (int)strlen(p->ref) - L + 1;
		for (int jj = 0; jj < nwin; ++jj) {
			++c->cnt_windows;
			mcomo_string_to_bits(p->ref + jj, L, win);
			for (int i = 0; i < L; ++i) t[i] = RC[(uint8_t)p->ref[jj + L - 1 - i]];
			mcomo_string_to_bits(t, L, rwin);
			for (int dir = 0; dir < 2; ++dir) {
				const uint64_t *q = dir ? rwin : win;
				for (int l = 0; l < nd; ++l) {
					if (dir && ds[l] <= 0) continue;                       /* :440 */
					if (!dir && de[l] >= L) continue;                      /* :363 */
					++c->cnt_lookups;
					uint64_t key = bits_key(q, ds[l], de[l] - ds[l] + 1);
					int64_t bin = dict_lookup(&D[l], key);
					if (bin < 0) continue;
					int64_t lo, hi;
					dict_findpos(&D[l], numreads, (uint32_t)bin, &lo, &hi);
					if (D[l].empty_bin[bin]) continue;
					/* the bin's key is re-checked against its first live read (:385): always equal here */
					del.n = 0;
					for (int64_t i = hi - 1; i >= lo && i >= hi - c->maxsearch; --i) {
						uint32_t sg_id = D[l].read_id[i], rid = c->sg.a[sg_id];
						if (popc_xor(q, bits + (size_t)sg_id * W, W) > thr) continue;
						if (!dir) { if (!mcomo_encode_byte(rd(c, rid), p->ref, jj, 0, L)) continue; }      /* :393 */
						else if (thr > 24 && !mcomo_encode_byte(rd(c, rid), p->ref, jj, 1, L)) continue;   /* :461 */
						if (c->sg_flag[sg_id]) continue;
						c->sg_flag[sg_id] = 1;
						uint64_t y = (uint64_t)rid << 32 | ((uint64_t)jj << 1) | (uint64_t)dir;
						VPUSH(uint64_t, *p, y);
						VPUSH(uint32_t, del, sg_id);
					}
					for (int l1 = 0; l1 < nd; ++l1) {                      /* :420-435 */
						for (size_t u = 0; u < del.n; ++u) {
							uint32_t sg_id = del.a[u];
							uint64_t k1 = bits_key(bits + (size_t)sg_id * W, ds[l1], de[l1] - ds[l1] + 1);
							int64_t b1 = dict_lookup(&D[l1], k1);
							int64_t lo1, hi1;
							dict_findpos(&D[l1], numreads, (uint32_t)b1, &lo1, &hi1);
							dict_remove(&D[l1], numreads, (uint32_t)b1, lo1, hi1, sg_id);
						}
					}
				}
			}
		}
	}
	free(del.a); free(win); free(rwin);
	for (int j = 0; j < nd; ++j) { free(D[j].keys); free(D[j].startpos); free(D[j].read_id); free(D[j].empty_bin); }
	free(D); free(t); free(allA); free(allT); free(bits);
	long cr = 0;
	for (size_t i = 0; i < cv->n; ++i) cr += (long)cv->a[i].n;
	return cr;
}

/* ================================================================================================
 * dump in the text format of oracle/refdump.cpp ("refdump stages")
 * ============================================================================================== */
static void dump_list(FILE *f, const char *name, const v32 *v)
{
	fprintf(f, "LIST %s %zu", name, v->n);
	for (size_t i = 0; i < v->n; ++i) fprintf(f, " %u", v->a[i]);
	fprintf(f, "\n");
}
static void dump_buckets(FILE *f, const char *name, const v128 *B)
{
	size_t tot = 0; int ne = 0;
	for (int i = 0; i < NBUCKET; ++i) if (B[i].n) { tot += B[i].n; ++ne; }
	fprintf(f, "BUCKETS %s %d %zu\n", name, ne, tot);
	for (int i = 0; i < NBUCKET; ++i) {
		if (!B[i].n) continue;
		fprintf(f, "B %d %zu", i, B[i].n);
		for (size_t j = 0; j < B[i].n; ++j) fprintf(f, " %" PRIu64 " %" PRIu64, B[i].a[j].x, B[i].a[j].y);
		fprintf(f, "\n");
	}
}
static void dump_contigs(FILE *f, const char *stage, const vcontig *cv)
{
	fprintf(f, "CLUSTERS %s %zu\n", stage, cv->n);
	for (size_t i = 0; i < cv->n; ++i) {
		const contig_t *p = &cv->a[i];
		fprintf(f, "C %zu %s", p->n, p->ref);
		for (size_t j = 0; j < p->n; ++j) fprintf(f, " %" PRIu64, p->a[j]);
		fprintf(f, "\n");
	}
}

static FILE *g_log;                                                       /* progress lines of a long run (digest_main.c) */
void mcomo_set_log(FILE *f) { g_log = f; }
static void run_stage2(mcomo_ctx *c, FILE *f)
{
	long pre = 0; int pass = 0;
	for (int thr = c->e;; thr += c->step) {                                /* preprocess.c:197-232 */
		if (thr > c->maxthr) break;
		/* updateSingle happens inside the pass; record its result for the dump first */
		v32 before; memset(&before, 0, sizeof before);
		if (f) { for (size_t i = 0; i < c->sg.n; ++i) if (!c->sg_flag[i]) VPUSH(uint32_t, before, c->sg.a[i]); }
		long cr = mcomo_stage_realign_pass(c, thr);
		if (f) {
			fprintf(f, "STAGE realign %d thr %d\n", pass, thr);
			dump_list(f, "sg_in", &before);
			fprintf(f, "SGFLAG %zu", c->sg.n);
			for (size_t i = 0; i < c->sg.n; ++i) fprintf(f, " %d", c->sg_flag[i] ? 1 : 0);
			fprintf(f, "\n");
			dump_list(f, "fpA", &c->fpA);
			dump_list(f, "fpT", &c->fpT);
			dump_contigs(f, "realign", &c->C[c->idxv]);
		}
		free(before.a);
		if (g_log) { fprintf(g_log, "realign pass %d thr %d: %ld reads in contigs\n", pass, thr, cr); fflush(g_log); }
		long lim = (c->sg.n > 1000000 && c->L >= 68) ? 10000 : 1000;
		++pass;
		if (cr - pre < lim) break;
		pre = cr;
	}
	mcomo_update_single(c);
}

int mcomo_dump_stages(mcomo_ctx *c, const char *path)
{
	FILE *f = fopen(path, "w");
	if (!f) return -1;
	fprintf(f, "PARAMS L %d k %d b %d rw %d e %d cbthr %d m %d n %zu\n", c->L, c->k, NB_BITS, c->rw, c->e, c->cbthr, c->m, c->n);
	mcomo_stage_reads(c);
	fprintf(f, "STAGE reads\nREADS %zu\n", c->n);
	for (size_t i = 0; i < c->n; ++i) fprintf(f, "%s\n", rd(c, (uint32_t)i));
	size_t nn = 0;
	for (size_t i = 0; i < c->n; ++i) if (c->npos[i].n) ++nn;
	fprintf(f, "NPOS %zu\n", nn);
	for (size_t i = 0; i < c->n; ++i) {
		if (!c->npos[i].n) continue;
		fprintf(f, "N %zu %zu", i, c->npos[i].n);
		for (size_t j = 0; j < c->npos[i].n; ++j) fprintf(f, " %u", c->npos[i].a[j]);
		fprintf(f, "\n");
	}
	dump_list(f, "allA", &c->allA); dump_list(f, "allT", &c->allT); dump_list(f, "allN", &c->allN);
	dump_list(f, "fpA", &c->fpA); dump_list(f, "fpT", &c->fpT); dump_list(f, "fpN", &c->fpN);
	dump_list(f, "Nfile", &c->Nfile);
	dump_buckets(f, "B0", c->B[0]);
	mcomo_stage_bucket(c);
	fprintf(f, "STAGE bucket\n");
	dump_contigs(f, "bucket", &c->C[0]);
	dump_list(f, "sg", &c->sg);
	dump_buckets(f, "MI0", c->MI[0]);
	mcomo_stage_combine(c);
	fprintf(f, "STAGE combine\n");
	dump_contigs(f, "combine", &c->C[c->idxv]);
	run_stage2(c, f);
	fprintf(f, "STAGE final\n");
	dump_list(f, "sg", &c->sg);
	fprintf(f, "END\n");
	fclose(f);
	return 0;
}

void mcomo_stage_realign_all(mcomo_ctx *c) { run_stage2(c, 0); }          /* preprocess.c:197-232 */
void mcomo_run_all(mcomo_ctx *c)
{
	mcomo_stage_reads(c);
	mcomo_stage_bucket(c);
	mcomo_stage_combine(c);
	run_stage2(c, 0);
}

/* ---- accessors -------------------------------------------------------------------------------- */
size_t mcomo_n_reads(const mcomo_ctx *c) { return c->n; }
const char *mcomo_seq(const mcomo_ctx *c) { return c->seq; }
const uint8_t *mcomo_cls(const mcomo_ctx *c) { return c->cls; }
const mcomo_mm128 *mcomo_rec0(const mcomo_ctx *c) { return c->rec0; }
size_t mcomo_n_sg(const mcomo_ctx *c) { return c->sg.n; }
const uint32_t *mcomo_sg(const mcomo_ctx *c) { return c->sg.a; }
const uint8_t *mcomo_sg_flag(const mcomo_ctx *c) { return c->sg_flag; }
size_t mcomo_n_contigs(const mcomo_ctx *c) { return c->C[c->idxv].n; }
const char *mcomo_contig_ref(const mcomo_ctx *c, size_t i) { return c->C[c->idxv].a[i].ref; }
size_t mcomo_contig_n(const mcomo_ctx *c, size_t i) { return c->C[c->idxv].a[i].n; }
const uint64_t *mcomo_contig_members(const mcomo_ctx *c, size_t i) { return c->C[c->idxv].a[i].a; }
size_t mcomo_counter(const mcomo_ctx *c, const char *name)
{
	if (!strcmp(name, "rounds")) return c->cnt_rounds;
	if (!strcmp(name, "windows")) return c->cnt_windows;
	if (!strcmp(name, "lookups")) return c->cnt_lookups;
	if (!strcmp(name, "passes")) return c->cnt_passes;
	if (!strcmp(name, "merge_rounds")) return c->cnt_merge_rounds;
	if (!strcmp(name, "resketch")) return c->cnt_resketch;
	if (!strcmp(name, "cand")) return c->cnt_cand;
	if (!strcmp(name, "k")) return (size_t)c->k;
	if (!strcmp(name, "rw")) return (size_t)c->rw;
	if (!strcmp(name, "maxsearch")) return (size_t)c->maxsearch;
	return 0;
}

/* ---- result digest: the same eight numbers as the product's mcomh_result_digest (include/mcom_host.h) computed from the
 * oracle's own contig set, so that a run too large to compare array by array can still be compared exactly: { contigs,
 * characters, members, unclustered reads, digest of the concatenated strings, of the concatenated member words, of the two
 * offset arrays, of the id lists }.  A digest of a byte array = position-weighted wrapping sum and xor of its little-endian
 * 64-bit words (a tail of fewer than eight bytes zero-extended), folded as sum ^ rotl(xor, 23). */
typedef struct { uint64_t s, x, i, part; int have; } dg_t;
static void dg_word(dg_t *d, uint64_t v) { d->s += v * (2 * (d->i & 0xFFFFF) + 1); d->x ^= v; ++d->i; }
static void dg_bytes(dg_t *d, const void *p_, size_t n)
{
	const uint8_t *p = (const uint8_t*)p_;
	for (size_t q = 0; q < n; ++q) {
		d->part |= (uint64_t)p[q] << (8 * d->have);
		if (++d->have == 8) { dg_word(d, d->part); d->part = 0; d->have = 0; }
	}
}
static uint64_t dg_end(dg_t *d)
{
	if (d->have) { dg_word(d, d->part); d->part = 0; d->have = 0; }
	return d->s ^ ((d->x << 23) | (d->x >> 41));
}
void mcomo_result_digest(const mcomo_ctx *c, uint64_t out[8])
{
	const vcontig *V = &c->C[c->idxv];
	dg_t ds, dm, dso, dmo; memset(&ds, 0, sizeof ds); dm = dso = dmo = ds;
	uint64_t chars = 0, members = 0;
	for (size_t i = 0; i < V->n; ++i) {
		const size_t len = strlen(V->a[i].ref);
		dg_word(&dso, chars); dg_word(&dmo, members);
		dg_bytes(&ds, V->a[i].ref, len);
		for (size_t j = 0; j < V->a[i].n; ++j) dg_word(&dm, V->a[i].a[j]);
		chars += len; members += V->a[i].n;
	}
	if (V->n) { dg_word(&dso, chars); dg_word(&dmo, members); }
	out[0] = V->n; out[1] = chars; out[2] = members;
	out[4] = dg_end(&ds); out[5] = dg_end(&dm); out[6] = dg_end(&dso) + 3 * dg_end(&dmo);
	uint64_t h = 0, nsg = 0;
	for (size_t i = 0; i < c->sg.n; ++i) if (!c->sg_flag || !c->sg_flag[i]) { h = h * 0x9E3779B97F4A7C15ull + c->sg.a[i] + 1; ++nsg; }
	out[3] = nsg;
	const v32 *L7[7] = { &c->allA, &c->allT, &c->allN, &c->fpA, &c->fpT, &c->fpN, &c->Nfile };
	for (int q = 0; q < 7; ++q) { h = h * 0xD6E8FEB86659FD93ull + L7[q]->n; for (size_t i = 0; i < L7[q]->n; ++i) h = h * 0x9E3779B97F4A7C15ull + L7[q]->a[i] + 1; }
	out[7] = h;
}

/* ================================================================================================
 * synthetic reads: same counter-based generator as minicom_amd/synth.py (plumbing=False)
 * ============================================================================================== */
static uint64_t sm64(uint64_t z)
{
	z += 0x9E3779B97F4A7C15ULL;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}
void mcomo_synth_reads(uint64_t seed, uint64_t n_reads, int L, int coverage, double sub_rate,
                       uint64_t first, uint64_t count, char *out)
{
	uint64_t G = n_reads * (uint64_t)L / (uint64_t)coverage;
	if (G < (uint64_t)L + 1) G = (uint64_t)L + 1;
	const uint64_t b0 = sm64(seed + 0), b1 = sm64(seed + 1), b2 = sm64(seed + 2), b3 = sm64(seed + 3);
	const uint64_t thr = (uint64_t)(sub_rate * (double)(1 << 24));
	for (uint64_t q = 0; q < count; ++q) {
		uint64_t r = first + q;
		uint64_t start = sm64(b1 + r) % (G - (uint64_t)L + 1);
		int strand = (int)(sm64(b2 + r) & 1);
		char *o = out + q * (uint64_t)L;
		for (int i = 0; i < L; ++i) {
			unsigned b = (unsigned)(sm64(b0 + start + (uint64_t)i) & 3);
			uint64_t u = sm64(b3 + r * (uint64_t)L + (uint64_t)i);
			if ((u & 0xFFFFFF) < thr) b = (b + 1 + (unsigned)((u >> 24) % 3)) & 3;
			if (strand) o[L - 1 - i] = "ACGT"[3 - b]; else o[i] = "ACGT"[b];
		}
	}
}

/* a context over the synthetic set itself, generated straight into the context's rows (a 100 M-read set has no room for a
 * second copy of its characters) */
mcomo_ctx *mcomo_new_synth(uint64_t seed, size_t n, int L, int coverage, double sub_rate, const mcomo_params *p)
{
	mcomo_ctx *c = mcomo_new(0, 0, L, p);
	free(c->seq); free(c->cls); free(c->rec0); free(c->npos);
	c->n = n;
	c->seq = (char*)malloc(n * (size_t)(L + 1));
	c->cls = (uint8_t*)calloc(n ? n : 1, 1);
	c->rec0 = (mcomo_mm128*)calloc(n ? n : 1, sizeof *c->rec0);
	c->npos = (v32*)calloc(n ? n : 1, sizeof *c->npos);
	char *row = (char*)malloc((size_t)L);
	for (size_t i = 0; i < n; ++i) {
		mcomo_synth_reads(seed, n, L, coverage, sub_rate, i, 1, row);
		memcpy(c->seq + i * (size_t)(L + 1), row, (size_t)L); c->seq[i * (size_t)(L + 1) + L] = 0;
	}
	free(row);
	return c;
}

/* named id lists: allA allT allN fpA fpT fpN Nfile sg */
const uint32_t *mcomo_list(const mcomo_ctx *c, const char *name, size_t *n)
{
	const v32 *v = 0;
	if (!strcmp(name, "allA")) v = &c->allA; else if (!strcmp(name, "allT")) v = &c->allT; else if (!strcmp(name, "allN")) v = &c->allN;
	else if (!strcmp(name, "fpA")) v = &c->fpA; else if (!strcmp(name, "fpT")) v = &c->fpT; else if (!strcmp(name, "fpN")) v = &c->fpN;
	else if (!strcmp(name, "Nfile")) v = &c->Nfile; else if (!strcmp(name, "sg")) v = &c->sg;
	if (!v) { *n = 0; return 0; }
	*n = v->n;
	return v->a;
}
