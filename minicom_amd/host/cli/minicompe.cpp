// minicompe IN_1.fastq IN_2.fastq OUTDIR [options] -- the paired-end compressor binary of the reference (built with _PE,
// minicom:188, :229), over libmcom_host.so: both files -> HBM (the second file's reads behind the first's,
// preprocess.c:60-68) -> Stage 1 + Stage 2 -> the paired-end stream files of cluster_dump_pe in OUTDIR.
#include "cli_common.hpp"

int main(int argc, char **argv)
{
	CliOptions o;
	if (argc < 4 || !cli_parse(argc, argv, 4, o) || o.order) { fprintf(stderr, "usage: minicompe IN_1.fastq IN_2.fastq OUTDIR [-k K -e E -m M -w W -s S -S STEP -E MAXTHR -g CBTHR -R ROUNDS -t THREADS -D GPU]\n"); return 1; }
	char err[256] = "";
	mcomh_pipeline *mp = nullptr;
	int rc = mcomh_create_from_fastq(&mp, o.device, nullptr, argv[1], argv[2], &o.prm, err, sizeof err);
	if (rc) { fprintf(stderr, "%s / %s: %s\n", argv[1], argv[2], err[0] ? err : "cannot read the files, no usable GPU or bad parameters"); return 1; }
	const size_t n = (size_t)mcomh_stat(mp, "n"); const int L = (int)mcomh_stat(mp, "L");
	if ((rc = cli_run(mp, n, L)) || (rc = mcomh_cluster_dump_pe(mp, argv[3]))) { fprintf(stderr, "%s\n", mcomh_last_error(mp)); return 1; }
	mcomh_destroy(mp);
	return 0;
}
