"""A/B on one box: mcomh_kt_for_reads as one classification + one sketch launch (the default) against four batches over two streams
(read_batches = 1): whole steps at 100 M x 150 bp, alternating, same digest.      python tools/ab_read_batches.py [reads] [rounds]"""
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, minicom_amd
from minicom_amd.pipeline import Pipeline
ctx = minicom_amd.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
L = 150
reads = ctx.synth_reads(1002, n, L); ctx.sync()
threads = max(1, min(64, os.cpu_count() or 8))
res = {0: [], 1: []}; stage = {0: [], 1: []}; dg = {}
for it in range(rounds + 1):
    for rb in (0, 1):
        p = Pipeline(reads, L=L, host_threads=threads, read_batches=rb); p.prof_enable(True)
        torch.cuda.synchronize(); t = time.perf_counter(); p.pre_process(); d = p.result_digest(); torch.cuda.synchronize(); ms = (time.perf_counter() - t) * 1e3
        if it: res[rb].append(ms); stage[rb].append(p.stat("t_reads") + p.stat("t_bucket"))
        dg.setdefault(rb, d); assert d == dg[rb]
        p.close()
assert dg[0] == dg[1], "digests differ"
for rb in (0, 1):
    print("read_batches=%d: steps %s  min %.2f  reads+bucket stage %s" % (rb, ["%.2f" % x for x in res[rb]], min(res[rb]), ["%.1f" % x for x in stage[rb]]), flush=True)
