import sys, time
sys.path.insert(0, '.')
import torch, minicom_amd
from minicom_amd.pipeline import Pipeline
ctx = minicom_amd.Context(0)
reads = ctx.synth_reads(1002, 100_000_000, 150); ctx.sync()
for i in range(3):
    p = Pipeline(reads, L=150, host_threads=32)
    t = time.perf_counter(); p.kt_for_reads(); t1 = time.perf_counter(); p.kt_for_bucket(); t2 = time.perf_counter()
    print("kt_for_reads host %.2f ms, kt_for_bucket %.2f ms; stats:" % ((t1 - t) * 1e3, (t2 - t1) * 1e3), {k: round(p.stat(k), 2) for k in ("t_reads", "t_bk_pre", "t_bk_sort", "t_bk_gpu", "t_bucket")})
    p.close()
