// minicom_amd/host/cli/cli_common.hpp -- shared by the three executables the `minicom` script runs (reference minicom:106,
// :229, :383): minicomsg IN OUTDIR, minicompe IN1 IN2 OUTDIR, decompress DIR OUT pe order nthr [OUT2].
// The reference compiles its parameters in (the script writes src/config.h and runs make on every invocation,
// minicom:56-102); here they are run-time options behind the positional arguments, same letters as the script's flags.
#pragma once
#include "../../../include/mcom.h"
#include "../../../include/mcom_host.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>

struct CliOptions { mcomh_params prm; bool order = false; int device = 0; };

// options after the positional arguments: -k -e -m -w -s -S -E (README.md:39-51), -g (merge threshold), -R (max rounds)
// (minicom:446-447), -t (host threads), -p (order preserving), -D (GPU)
static inline bool cli_parse(int argc, char **argv, int first, CliOptions &o)
{
	memset(&o.prm, 0, sizeof o.prm);
	for (int i = first; i < argc; ++i) {
		const char *a = argv[i];
		if (a[0] != '-' || !a[1] || a[2]) return false;
		if (a[1] == 'p') { o.order = true; continue; }
		if (i + 1 >= argc) return false;
		const int v = atoi(argv[++i]);
		switch (a[1]) {
		case 'k': o.prm.k = v; break;          case 'e': o.prm.e = v; break;       case 'm': o.prm.m = v; break;
		case 'w': o.prm.w = v; break;          case 's': o.prm.numdict = v; break; case 'S': o.prm.step = v; break;
		case 'E': o.prm.maxthr = v; break;     case 'g': o.prm.cbthr = v; break;   case 'R': o.prm.max_rounds = v; break;
		case 't': o.prm.host_threads = v; o.prm.stream_sets = v; break;   // (the reference writes one stream set per thread: the count is part of the format)
		case 'D': o.device = v; break;
		default: return false;
		}
	}
	return true;
}

// pre_process with the reference's progress lines (preprocess.c:53-55, :186, :235)
static inline int cli_run(mcomh_pipeline *mp, size_t n, int L)
{
	fprintf(stdout, "Number of reads: %zu\nLength of reads: %d\n", n, L);
	const int rc = mcomh_pre_process(mp);
	if (rc) return rc;
	fprintf(stdout, "[Stage 1] Real time: %.3f sec\n", (mcomh_stat(mp, "t_reads") + mcomh_stat(mp, "t_bucket") + mcomh_stat(mp, "t_combine")) / 1e3);
	fprintf(stdout, "[Stage 2] Real time: %.3f sec\n", mcomh_stat(mp, "t_realign") / 1e3);
	return 0;
}
