"""The hot path on the repeat-rich device genome (mcom_synth_reads_genome kind 1): python tools/repeat_bench.py [reads] [L]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, minicom_amd
from minicom_amd.pipeline import Pipeline
from minicom_amd.check import check_result
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
ctx = minicom_amd.Context(0)
for genome in ("uniform", "repeats"):
    reads = ctx.synth_reads(1002, n, L, genome=genome); ctx.sync()
    ds = []
    for it in range(3):
        p = Pipeline(reads, L=L, host_threads=16)
        torch.cuda.synchronize(); t = time.perf_counter(); p.pre_process(); d = p.result_digest(); torch.cuda.synchronize(); dt = (time.perf_counter() - t) * 1e3
        ds.append(d)
        st = {k: p.stat(k) for k in ("rounds", "merge_rounds", "passes", "big_bins", "big_bin_reads", "big_bin_tuples", "dict_builds", "cix_rebuilds", "claim_rounds", "n_sg0",
                                     "t_bucket", "t_combine", "t_realign", "contigs_bucket", "contigs_combine")}
        st["sort_overflow_segments"] = p.lib.mcomh_stat(p._h, b"sort_overflow_segments")
        if it == 2:
            res = check_result(p, reads, L)
        p.close()
        print(f"{genome} run {it}: {dt:.1f} ms = {n / dt / 1e3:.1f} Mreads/s  {st}", flush=True)
    assert ds[0] == ds[1] == ds[2]
    print(genome, "checked:", {k: res[k] for k in ("n_contigs", "members", "n_sg", "longest_contig", "largest_contig_members", "max_mismatch", "mean_mismatch")}, flush=True)
    del reads; torch.cuda.empty_cache()
