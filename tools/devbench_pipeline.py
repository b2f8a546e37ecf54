"""Developer timing of the whole hot path (host driver + kernels) at several sizes."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import minicom_amd
from minicom_amd.pipeline import Pipeline

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, nargs="+", default=[1_000_000])
ap.add_argument("--L", type=int, default=150)
ap.add_argument("--threads", type=int, default=16)
ap.add_argument("--prof", action="store_true")
a = ap.parse_args()
ctx = minicom_amd.Context(0)
for n in a.n:
    reads = ctx.synth_reads(1002, n, a.L)
    ctx.sync()
    t0 = time.perf_counter()
    p = Pipeline(reads, L=a.L, host_threads=a.threads)
    if a.prof:
        p.prof_enable(True)
    p.pre_process()
    dt = time.perf_counter() - t0
    keys = ["t_reads", "t_bucket", "t_combine", "t_realign", "t_gpu", "rounds", "merge_rounds", "passes", "windows", "resketch", "n_sg0", "big_bins", "cand_pairs", "t_claim", "t_merge_cons", "t_merge_members",
            "t_cb_upload", "t_cb_sketch", "t_cb_idx", "t_cb_findnext", "t_cb_d2h", "t_cb_copy", "t_cb_free",
            "t_bk_gpu", "t_bk_sort", "t_bk_cons", "t_bk_replay", "t_ra_sort", "t_ra_gpu", "t_ra_setup", "t_ra_append", "ra_lookups", "ra_verified", "ra_passing", "ra_singletons", "t_mat_join", "t_mat_count", "t_ra_materialize", "t_cb_pack", "t_cb_download", "claim_rounds", "sketch_bases", "resketch_saved_bases", "sketch_records", "t_ra_update"]
    print(f"n={n} L={a.L}: {dt:.3f} s  {n/dt/1e6:.3f} Mreads/s  contigs={len(p.contigs()) if n <= 2_000_000 else -1} sg={len(p.id_list('sg'))}", flush=True)
    print("   " + " ".join(f"{k}={p.stat(k):.0f}" for k in keys), flush=True)
    if a.prof:
        print("   prof " + " ".join("%s=%.1fms/%d" % ((k,) + p.prof_read(k)) for k in ("classify_pack", "sketch_reads", "radix_pass", "sketch_contigs", "find_next", "dict_build", "realign_windows", "consensus", "cindex_build", "realign_reads")), flush=True)
    p.close()
    del reads
    torch.cuda.empty_cache()
