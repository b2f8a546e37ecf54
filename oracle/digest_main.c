/* oracle/digest_main.c -- TEST INFRASTRUCTURE ONLY.
 * Runs the sequential oracle (mcom_oracle.c, the restatement pinned on the reference's own dumps) over one synthetic read set
 * of the generator shared with minicom_amd/synth.py and the device generator, and prints the result digest the product's
 * mcomh_result_digest must reproduce.  This is how the digests of tests/golden/scale_digests.json were made (hours of one
 * core at 100 M reads; see tests/golden/make_scale_digests.sh).
 *     digest_main SEED N_READS READ_LEN [COVERAGE=30]                                                                    */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "mcom_oracle.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

int main(int argc, char **argv)
{
	if (argc < 4) { fprintf(stderr, "usage: %s seed n_reads read_len [coverage]\n", argv[0]); return 2; }
	const uint64_t seed = strtoull(argv[1], 0, 10);
	const size_t n = (size_t)strtoull(argv[2], 0, 10);
	const int L = atoi(argv[3]);
	const int cov = argc > 4 ? atoi(argv[4]) : 30;
	double t0 = now();
	mcomo_ctx *c = mcomo_new_synth(seed, n, L, cov, 0.005, 0);
	fprintf(stderr, "[%.0f s] %zu reads of %d bases generated\n", now() - t0, n, L);
	mcomo_stage_reads(c);   fprintf(stderr, "[%.0f s] kt_for_reads\n", now() - t0);
	mcomo_stage_bucket(c);  fprintf(stderr, "[%.0f s] kt_for_bucket: %zu contigs, %zu rounds\n", now() - t0, mcomo_n_contigs(c), mcomo_counter(c, "rounds"));
	mcomo_stage_combine(c); fprintf(stderr, "[%.0f s] combine_cluster: %zu contigs, %zu merge rounds\n", now() - t0, mcomo_n_contigs(c), mcomo_counter(c, "merge_rounds"));
	mcomo_set_log(stderr);
	mcomo_stage_realign_all(c);
	fprintf(stderr, "[%.0f s] Stage 2: %zu passes\n", now() - t0, mcomo_counter(c, "passes"));
	uint64_t d[8];
	mcomo_result_digest(c, d);
	printf("{\"seed\": %llu, \"n\": %zu, \"L\": %d, \"coverage\": %d, \"oracle_seconds\": %.0f, \"passes\": %zu, \"merge_rounds\": %zu, \"rounds\": %zu, \"digest\": [",
	       (unsigned long long)seed, n, L, cov, now() - t0, mcomo_counter(c, "passes"), mcomo_counter(c, "merge_rounds"), mcomo_counter(c, "rounds"));
	for (int i = 0; i < 8; ++i) printf("%s%llu", i ? ", " : "", (unsigned long long)d[i]);
	printf("]}\n");
	return 0;
}
