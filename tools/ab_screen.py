import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, minicom_amd
from minicom_amd.pipeline import Pipeline
ctx = minicom_amd.Context(0)
n, L = 100_000_000, 150
reads = ctx.synth_reads(1002, n, L); ctx.sync()
for label, prm in (("t16", {"host_threads": 16}), ("t64", {"host_threads": 64}), ("t4", {"host_threads": 4}), ("t64", {"host_threads": 64})):
    ts = []
    for it in range(3):
        p = Pipeline(reads, L=L, **prm); p.prof_enable(True)
        torch.cuda.synchronize(); t = time.perf_counter(); p.pre_process(); d = p.result_digest(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        info = (p.stat("t_realign"), p.prof_read("cindex_build")[0], p.prof_read("realign_reads")[0], p.prof_read("find_next")[0], p.stat("t_combine"))
        p.close()
    print(label, ["%.1f" % x for x in ts], "t_realign %.1f cindex %.1f realign_reads %.1f find_next %.1f t_combine %.1f" % info, flush=True)
