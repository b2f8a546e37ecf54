// minicomsg IN.fastq OUTDIR [options] -- the single-end compressor binary of the reference (minicommain.c:81-216, run by the
// script as `./minicomsg $filename $compfiles`, minicom:106), over libmcom_host.so: FASTQ -> HBM -> Stage 1 + Stage 2 ->
// the stream files of cluster_dump in OUTDIR (which must exist).  -p = the reference built with ORDER (minicom:59-62).
#include "cli_common.hpp"

int main(int argc, char **argv)
{
	CliOptions o;
	if (argc < 3 || !cli_parse(argc, argv, 3, o)) { fprintf(stderr, "usage: minicomsg IN.fastq OUTDIR [-k K -e E -m M -w W -s S -S STEP -E MAXTHR -g CBTHR -R ROUNDS -t THREADS -p -D GPU]\n"); return 1; }
	char err[256] = "";
	mcomh_pipeline *mp = nullptr;
	int rc = mcomh_create_from_fastq(&mp, o.device, nullptr, argv[1], nullptr, &o.prm, err, sizeof err);
	if (rc) { fprintf(stderr, "%s: %s\n", argv[1], err[0] ? err : "cannot read the file, no usable GPU or bad parameters"); return 1; }
	const size_t n = (size_t)mcomh_stat(mp, "n"); const int L = (int)mcomh_stat(mp, "L");
	if ((rc = cli_run(mp, n, L)) || (rc = o.order ? mcomh_cluster_dump_order(mp, argv[2]) : mcomh_cluster_dump(mp, argv[2]))) { fprintf(stderr, "%s\n", mcomh_last_error(mp)); return 1; }
	mcomh_destroy(mp);
	return 0;
}
