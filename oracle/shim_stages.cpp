// oracle/shim_stages.cpp -- INTEGRATION.md section A as a program that links.  TEST INFRASTRUCTURE ONLY (oracle/build_ref.sh).
//
// The reference's own, unmodified objects -- minicommain.o (main), preprocess.o (pre_process, updateSingle), kthread_dump.o
// (cluster_dump, the stream writer), bseq.o, misc.o, sketch.o, kthread_idx.o, compiled from /root/reference/src where the sources
// lie -- are linked with THIS file instead of kthread_reads.o, kthread_bucket.o, kthread_cb.o, kthread_hash_realign.o and
// bbhashdict.o.  This file defines the four stage drivers pre_process calls (preprocess.c:141, :166, :178, :204) by calling
// libmcom_host.so (include/mcom_host.h) and hands the results back through the reference's own structures (reads_t, breads.h:75-109),
// so that the reference's loop control (preprocess.c:197-232), its updateSingle and its cluster_dump run on them unchanged.
// No reference source is copied or patched; what is restated here is the meaning of the fields the stages leave behind:
//   kt_for_reads   (kthread_reads.c:40-230): seq[i].n_pos (positions of N, or NULL), the class lists (sp->allX / allX_id, fpX_id);
//                  and Nfile_id (more than 0.4 L N, :219-224)
//   kt_for_bucket  (kthread_bucket.c:562-629): sg (its size decides maxsearch, preprocess.c:169-172)
//   combine_cluster(kthread_cb.c:570-627): clusters[idxv][tid]: the contigs (ref, members rid<<32 | offset<<1 | dir)
//   realign_hash   (kthread_hash_realign.c:569-600): members appended, sg_flag set for every read that left the singleton list,
//                  fpA_id / fpT_id extended by the near-poly reads of bbhashdict.c:177-216
// The N substitution of kept reads (kthread_reads.c:183-205) is not written back into seq[i].seq: the only reader left in this
// link is the stream writer, which puts the N back first (kthread_dump.c:69-75) -- either string gives the same bytes.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "config.h"
#include "kvec.h"
#include "breads.h"
#include "../include/mcom_host.h"

static mcomh_pipeline *mp;

static void die(const char *what)
{
	fprintf(stderr, "shim_stages: %s: %s\n", what, mp ? mcomh_last_error(mp) : "no pipeline");
	exit(1);
}
template <class V> static void take_list(V &v, const char *name)
{
	size_t n = 0;
	const uint32_t *a = mcomh_list(mp, name, &n);
	v.n = 0;
	for (size_t i = 0; i < n; ++i) kv_push(uint32_t, v, a[i]);
}

void kt_for_reads(int n_threads_, reads_t *r, long n)
{
	const int L = readlen;
	uint8_t *dense = (uint8_t*)malloc((size_t)n * L + 1);
	for (long i = 0; i < n; ++i) {
		memcpy(dense + (size_t)i * L, r->seq[i].seq, (size_t)L);
		uint32_v *np = (uint32_v*)calloc(1, sizeof(uint32_v));                 // kthread_reads.c:49, :69-80
		for (int j = 0; j < L; ++j) if (r->seq[i].seq[j] == 'N') kv_push(uint32_t, *np, (uint32_t)j);
		if (np->n) r->seq[i].n_pos = np; else { r->seq[i].n_pos = NULL; free(np); }
	}
	mcomh_params prm; memset(&prm, 0, sizeof prm);
	prm.k = r->k; prm.e = diff_threshold; prm.m = first_mininum; prm.w = rw; prm.cbthr = cbthreshold;
	prm.max_rounds = max_rounds; prm.step = thr_step; prm.maxthr = maxthr; prm.numdict = ininumdict;
	prm.host_threads = n_threads_ > 0 ? n_threads_ : 1;
	if (mcomh_create(&mp, 0, NULL, dense, NULL, 0, (size_t)n, L, &prm)) { fprintf(stderr, "shim_stages: no GPU / library\n"); exit(1); }
	if (mcomh_kt_for_reads(mp)) die("kt_for_reads");
	free(dense);
	take_list(r->sp->allA_id, "allA"); r->sp->allA = (int)r->sp->allA_id.n;
	take_list(r->sp->allT_id, "allT"); r->sp->allT = (int)r->sp->allT_id.n;
	take_list(r->sp->allN_id, "allN"); r->sp->allN = (int)r->sp->allN_id.n;
	take_list(r->fpA_id, "fpA"); take_list(r->fpT_id, "fpT"); take_list(r->fpN_id, "fpN");
	take_list(r->Nfile_id, "Nfile");                                           // more than 0.4 L N: kthread_reads.c:219-224
}

void kt_for_bucket(int, reads_t *r, long)
{
	if (mcomh_kt_for_bucket(mp)) die("kt_for_bucket");
	take_list(r->sg, "sg");
}

static void take_contigs(reads_t *r, int index)
{
	cluster_v &cv = r->clusters[index][0];
	for (size_t i = 0; i < cv.n; ++i) { free(cv.a[i].a); free(cv.a[i].ref); }
	cv.n = 0;
	size_t nc = 0; const char *ref; const uint64_t *roff, *mem, *moff;
	if (mcomh_contig_set(mp, &nc, &ref, &roff, &mem, &moff)) die("contig set");
	for (size_t c = 0; c < nc; ++c) {
		cluster_t *p; kv_pushp(cluster_t, cv, &p);
		memset(p, 0, sizeof *p);
		const size_t rl = (size_t)(roff[c + 1] - roff[c]), m = (size_t)(moff[c + 1] - moff[c]);
		p->ref = (char*)malloc(rl + 1); memcpy(p->ref, ref + roff[c], rl); p->ref[rl] = 0;
		p->a = (uint64_t*)malloc((m ? m : 1) * sizeof(uint64_t)); p->n = p->m = m;
		memcpy(p->a, mem + moff[c], m * sizeof(uint64_t));
	}
}

void combine_cluster(int, reads_t *r, int *index)
{
	if (mcomh_combine_cluster(mp)) die("combine_cluster");
	*index = 0;
	take_contigs(r, 0);
}

// one Stage-2 pass.  pre_process has just run the reference's updateSingle on ITS list; the library does its own inside.  Afterwards the
// reference's loop control counts the members of clusters[][] (preprocess.c:205-227) and its next updateSingle drops the entries of sg
// whose sg_flag is set: both are brought up to date here.
void realign_hash(int, reads_t *r, int index, int threshold)
{
	long cr = 0;
	if (mcomh_update_single(mp) || mcomh_realign_hash(mp, threshold, &cr)) die("realign_hash");
	if (mcomh_update_single(mp)) die("updateSingle");                          // the library's list without the reads this pass took
	size_t n = 0;
	const uint32_t *live = mcomh_list(mp, "sg", &n);
	size_t q = 0;
	for (size_t i = 0; i < r->sg.n; ++i) {                                      // both lists are in the same order
		if (q < n && live[q] == r->sg.a[i]) { r->sg_flag[i] = false; ++q; }
		else r->sg_flag[i] = true;
	}
	if (q != n) { fprintf(stderr, "shim_stages: the singleton lists disagree (%zu of %zu matched)\n", q, n); exit(1); }
	take_list(r->fpA_id, "fpA"); take_list(r->fpT_id, "fpT");
	take_contigs(r, index);
}

// cmpcluster2 lives in kthread_cb.c (not in this link): members by offset, then direction (the writer sorts every list with it,
// kthread_dump.c:143; the lists arrive in that order already)
int cmpcluster2(const void *a_, const void *b_)
{
	const uint64_t a = *(const uint64_t*)a_, b = *(const uint64_t*)b_;
	const int pa = (int)((uint32_t)a >> 1), pb = (int)((uint32_t)b >> 1);
	return pa != pb ? pa - pb : (int)(a & 1) - (int)(b & 1);
}
