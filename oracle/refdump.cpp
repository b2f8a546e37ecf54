// oracle/refdump.cpp -- golden-vector dumper.  TEST INFRASTRUCTURE ONLY.
//
// Our own driver, linked against the *reference's* object files (built by oracle/build_ref.sh from
// the read-only sources under /root/reference/src, outputs in oracle/_ref/).  It calls the reference's
// hot-path functions and prints their results as text so that tests/golden/ fixtures can be made by
// tests/golden/make_golden.py.  No reference source is contained here; only its public prototypes
// (breads.h) are included at build time.
//
//   refdump kat            < commands > results      function-level known answers
//   refdump stages IN.fastq [k]         > dump        state after every hot-path stage at -t 1
//
// kat commands (one per line):
//   S2 k rid SEQ          -> "S2 x y"                       mm_sketch_two       (sketch.c:238)
//   LH w k rid SEQ        -> "LH n x y x y ..."              mm_sketch_lh_ori    (sketch.c:116)
//   RS n x y x y ...      -> "RS n x y ..."                  radix_sort_128x     (misc.c:22, ksort.h:153)
//   MP i j STR0 STR1      -> "MP d"                          match_pro           (kthread_cb.c:36)
//   EB pos dir SEQ REF    -> "EB 0|1"                        encode_byte         (kthread_hash_realign.c:283)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <inttypes.h>
#include <string>
#include <vector>
#include <iostream>
#include <sstream>
#include "kvec.h"
#include "breads.h"
#include "config.h"

void kt_for_bucket(int n_threads, reads_t *reads, long n);
void kt_for_reads(int n_threads, reads_t *reads, long n);

static void do_kat()
{
	std::string line;
	while (std::getline(std::cin, line)) {
		std::istringstream in(line);
		std::string cmd;
		if (!(in >> cmd)) continue;
		if (cmd == "S2") {
			int k; uint32_t rid; std::string s;
			in >> k >> rid >> s;
			mm128_t m;
			mm_sketch_two(s.c_str(), (int)s.size(), k, rid, &m);
			printf("S2 %" PRIu64 " %" PRIu64 "\n", m.x, m.y);
		} else if (cmd == "LH") {
			int w, k; uint32_t rid; std::string s;
			in >> w >> k >> rid >> s;
			mm128_v v; kv_init(v);
			mm_sketch_lh_ori(s.c_str(), (int)s.size(), w, k, rid, &v);
			printf("LH %zu", v.n);
			for (size_t i = 0; i < v.n; ++i) printf(" %" PRIu64 " %" PRIu64, v.a[i].x, v.a[i].y);
			printf("\n");
			kv_destroy(v);
		} else if (cmd == "RS") {
			size_t n; in >> n;
			std::vector<mm128_t> a(n);
			for (size_t i = 0; i < n; ++i) in >> a[i].x >> a[i].y;
			radix_sort_128x(a.data(), a.data() + n);
			printf("RS %zu", n);
			for (size_t i = 0; i < n; ++i) printf(" %" PRIu64 " %" PRIu64, a[i].x, a[i].y);
			printf("\n");
		} else if (cmd == "MP") {
			int i, j; std::string s0, s1;
			in >> i >> j >> s0 >> s1;
			printf("MP %d\n", match_pro((char*)s0.c_str(), (char*)s1.c_str(), i, j));
		} else if (cmd == "EB") {
			int pos, dir; std::string s, r;
			in >> pos >> dir >> s >> r;
			printf("EB %d\n", encode_byte((char*)s.c_str(), (char*)r.c_str(), pos, dir) ? 1 : 0);
		}
	}
}

static void dump_u32v(const char *name, const uint32_v &v)
{
	printf("LIST %s %zu", name, v.n);
	for (size_t i = 0; i < v.n; ++i) printf(" %u", v.a[i]);
	printf("\n");
}

static void dump_clusters(const char *stage, int idx)
{
	cluster_v *cv = &reads->clusters[idx][0];
	printf("CLUSTERS %s %zu\n", stage, cv->n);
	for (size_t i = 0; i < cv->n; ++i) {
		cluster_t *c = &cv->a[i];
		printf("C %zu %s", c->n, c->ref);
		for (size_t j = 0; j < c->n; ++j) printf(" %" PRIu64, c->a[j]);
		printf("\n");
	}
}

static void dump_buckets(const char *name, cluster_bucket_t *B, int nb)
{
	size_t tot = 0; int ne = 0;
	for (int i = 0; i < nb; ++i) if (B[i].n) { tot += B[i].n; ++ne; }
	printf("BUCKETS %s %d %zu\n", name, ne, tot);
	for (int i = 0; i < nb; ++i) {
		if (!B[i].n) continue;
		printf("B %d %zu", i, B[i].n);
		for (size_t j = 0; j < B[i].n; ++j) printf(" %" PRIu64 " %" PRIu64, B[i].a[j].x, B[i].a[j].y);
		printf("\n");
	}
}

// State after each stage of the reference's Stage 1 + Stage 2 at one thread.  The call sequence is
// the one the reference's own driver performs (preprocess.c:141-233); every compute call below is a
// reference function.
/* non-default parameters, set the way the reference's main() sets them from config.h (minicommain.c:112-143);
 * the number of dictionaries (-s) is a compile-time macro of kthread_hash_realign.c:153: it needs its own build variant */
struct StageParams { int k, e, m, w, g, R, S, E; };

static void do_stages(const char *fn, const StageParams &sp)
{
	n_threads = 1;
	int k = readlen < 80 ? 17 : 31;
	if (sp.k > 0) k = sp.k;
	if (sp.e > 0) diff_threshold = sp.e;
	if (sp.m > 0) first_mininum = sp.m;
	cbthreshold = sp.g > 0 ? sp.g : 2 * diff_threshold;
	if (sp.R > 0 && sp.R < max_rounds) max_rounds = sp.R;
	thr_step = diff_threshold;
	if (sp.S > 0) thr_step = sp.S; else if (thr_step > 10) thr_step = 5;
	maxthr = readlen / 2;
	if (sp.E > 0) maxthr = sp.E;
	maxmatch = readlen / 2;
	rw = sp.w;
	int b = 14;

	bseq_file_t *fp = bseq_open(fn);
	if (!fp) { fprintf(stderr, "cannot open %s\n", fn); exit(1); }
	reads = (reads_t*)calloc(1, sizeof(reads_t));
	reads->seq_len = readlen;
	reads->seq = bseq_read(fp, &reads->n_seq, readlen);
	bseq_close(fp);
	reads->f = 0; reads->k = k; reads->b = b;
	reads->rw = readlen >= 70 ? readlen / 2 - k : 3;
	if (sp.w > 0) reads->rw = sp.w;                                      /* preprocess.c:105-107 */
	reads->B = (cluster_bucket_t**)calloc(2, sizeof(cluster_bucket_t*));
	reads->B[0] = (cluster_bucket_t*)calloc(1 << b, sizeof(cluster_bucket_t));
	reads->B[1] = (cluster_bucket_t*)calloc(1 << b, sizeof(cluster_bucket_t));
	reads->sp = (sp_reads_t*)calloc(1, sizeof(sp_reads_t));
	kv_init(reads->sg); kv_init(reads->fpA_id); kv_init(reads->fpT_id); kv_init(reads->fpN_id);
	kv_init(reads->Nfile_id); kv_init(reads->singleFile_id);

	printf("PARAMS L %d k %d b %d rw %d e %d cbthr %d m %d n %d\n", readlen, k, b, reads->rw,
	       diff_threshold, cbthreshold, first_mininum, reads->n_seq);

	kt_for_reads(1, reads, reads->n_seq);
	printf("STAGE reads\n");
	printf("READS %d\n", reads->n_seq);
	for (int i = 0; i < reads->n_seq; ++i) printf("%s\n", reads->seq[i].seq);
	int nn = 0;
	for (int i = 0; i < reads->n_seq; ++i) if (reads->seq[i].n_pos) ++nn;
	printf("NPOS %d\n", nn);
	for (int i = 0; i < reads->n_seq; ++i) {
		uint32_v *np = (uint32_v*)reads->seq[i].n_pos;
		if (!np) continue;
		printf("N %d %zu", i, np->n);
		for (size_t j = 0; j < np->n; ++j) printf(" %u", np->a[j]);
		printf("\n");
	}
	dump_u32v("allA", reads->sp->allA_id);
	dump_u32v("allT", reads->sp->allT_id);
	dump_u32v("allN", reads->sp->allN_id);
	dump_u32v("fpA", reads->fpA_id);
	dump_u32v("fpT", reads->fpT_id);
	dump_u32v("fpN", reads->fpN_id);
	dump_u32v("Nfile", reads->Nfile_id);
	dump_buckets("B0", reads->B[0], 1 << b);

	reads->mi = (mm_idx_t**)calloc(2, sizeof(mm_idx_t*));
	reads->mi[0] = mm_idx_init(reads->b);
	reads->clusters = (cluster_v**)calloc(2, sizeof(cluster_v*));
	for (int i = 0; i < 2; ++i) reads->clusters[i] = (cluster_v*)calloc(1, sizeof(cluster_v));
	kv_init(reads->clusters[0][0]);
	kv_resize(cluster_t, reads->clusters[0][0], 1 << 10);
	reads->single = 0;
	kt_for_bucket(1, reads, 1 << reads->b);
	printf("STAGE bucket\n");
	dump_clusters("bucket", 0);
	dump_u32v("sg", reads->sg);
	{
		size_t tot = 0; int ne = 0;
		for (int i = 0; i < (1 << b); ++i) if (reads->mi[0]->B[i].a.n) { tot += reads->mi[0]->B[i].a.n; ++ne; }
		printf("BUCKETS MI0 %d %zu\n", ne, tot);
		for (int i = 0; i < (1 << b); ++i) {
			mm128_v *a = &reads->mi[0]->B[i].a;
			if (!a->n) continue;
			printf("B %d %zu", i, a->n);
			for (size_t j = 0; j < a->n; ++j) printf(" %" PRIu64 " %" PRIu64, a->a[j].x, a->a[j].y);
			printf("\n");
		}
	}
	if (reads->sg.n <= 5000000) { maxmatch = readlen * 2 / 3; maxsearch = 2000; }

	idxv = 0;
	combine_cluster(1, reads, &idxv);
	printf("STAGE combine\n");
	dump_clusters("combine", idxv);

	reads->sg_flag = (bool*)calloc(reads->sg.n, sizeof(bool));
	int pre = 0; bool go = true; int pass = 0;
	for (int thr = diff_threshold; go; thr += thr_step) {
		if (thr > maxthr) break;
		updateSingle();
		size_t nsg = reads->sg.n;
		std::vector<uint32_t> sg_before(reads->sg.a, reads->sg.a + nsg);
		realign_hash(1, reads, idxv, thr);
		printf("STAGE realign %d thr %d\n", pass, thr);
		{
			uint32_v t; t.n = nsg; t.m = nsg; t.a = sg_before.data();
			dump_u32v("sg_in", t);
		}
		printf("SGFLAG %zu", nsg);
		for (size_t i = 0; i < nsg; ++i) printf(" %d", reads->sg_flag[i] ? 1 : 0);
		printf("\n");
		dump_u32v("fpA", reads->fpA_id);
		dump_u32v("fpT", reads->fpT_id);
		dump_clusters("realign", idxv);
		int cr = 0;
		for (size_t i = 0; i < reads->clusters[idxv][0].n; ++i) cr += reads->clusters[idxv][0].a[i].n;
		int lim = (reads->sg.n > 1000000 && readlen >= 68) ? 10000 : 1000;
		if (cr - pre < lim) go = false;
		pre = cr; ++pass;
	}
	updateSingle();
	printf("STAGE final\n");
	dump_u32v("sg", reads->sg);
	printf("END\n");
}

int main(int argc, char **argv)
{
	if (argc >= 2 && !strcmp(argv[1], "kat")) { do_kat(); return 0; }
	if (argc >= 3 && !strcmp(argv[1], "stages")) {
		StageParams sp = {0, 0, 0, 0, 0, 0, 0, 0};
		for (int i = 3; i < argc; ++i) {
			const char *a = argv[i];
			if (a[0] && a[1] == '=') {
				const int v = atoi(a + 2);
				switch (a[0]) { case 'k': sp.k = v; break; case 'e': sp.e = v; break; case 'm': sp.m = v; break; case 'w': sp.w = v; break;
				                case 'g': sp.g = v; break; case 'R': sp.R = v; break; case 'S': sp.S = v; break; case 'E': sp.E = v; break;
				                default: fprintf(stderr, "unknown parameter %s\n", a); return 2; }
			} else sp.k = atoi(a);
		}
		do_stages(argv[2], sp);
		return 0;
	}
	fprintf(stderr, "usage: refdump kat | refdump stages IN.fastq [k] [e=.. m=.. w=.. g=.. R=.. S=.. E=..]\n");
	return 2;
}
