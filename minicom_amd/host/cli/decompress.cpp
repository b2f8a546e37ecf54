// decompress DIR OUT pe order nthr [OUT2] -- the decoder binary of the reference (decompress.c:1225-1317, run by the
// script as `./decompress $decomp $result $pe $preserve_order $num_thr $result0`, minicom:383; pe / order are the
// words true / false).  Host only.  Unlike the reference's, which leaves per-thread part files for the script to
// concatenate (minicom:385-399), this one writes the final file(s) itself.
#include "../../../include/mcom_host.h"
#include <cstdio>
#include <cstring>

int main(int argc, char **argv)
{
	if (argc < 6) { fprintf(stderr, "usage: decompress DIR OUT pe(true|false) order(true|false) nthr [OUT2]\n"); return 1; }
	const bool pe = !strcmp(argv[3], "true") || !strcmp(argv[3], "1"), order = !strcmp(argv[4], "true") || !strcmp(argv[4], "1");
	uint64_t n = 0;
	int rc;
	if (pe) {
		if (argc < 7) { fprintf(stderr, "decompress: paired-end archives need OUT2\n"); return 1; }
		rc = mcomh_decompress_pe(argv[1], argv[2], argv[6], &n);
	} else rc = order ? mcomh_decompress_order(argv[1], argv[2], &n) : mcomh_decompress(argv[1], argv[2], &n);
	if (rc) { fprintf(stderr, "decompress: %s does not hold a complete, consistent set of stream files\n", argv[1]); return 1; }
	fprintf(stdout, "%llu %s\n", (unsigned long long)n, pe ? "pairs" : "reads");
	return 0;
}
