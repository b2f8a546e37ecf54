"""How many contigs the bucket stage makes per read at a given coverage (sizes the > 2^24-contig test)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minicom_amd
from minicom_amd.pipeline import Pipeline
n, L, cov = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx = minicom_amd.Context(0)
reads = ctx.synth_reads(1005, n, L, coverage=cov); ctx.sync()
t = time.time(); p = Pipeline(reads, L=L, host_threads=8); p.pre_process(); d = p.result_digest()
print("%d x %d at %dx: %.2f s, contigs after buckets %d (%.3f per read), after merging %d, final %d contigs %d members %d unclustered, cix_entries %d" % (
    n, L, cov, time.time() - t, p.stat("contigs_bucket"), p.stat("contigs_bucket") / n, p.stat("contigs_combine"), d[0], d[2], d[3], p.stat("cix_entries")), flush=True)
