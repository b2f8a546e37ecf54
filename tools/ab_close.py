import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, minicom_amd
from minicom_amd.pipeline import Pipeline
ctx = minicom_amd.Context(0)
n, L = 100_000_000, 150
reads = ctx.synth_reads(1002, n, L); ctx.sync()
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p = Pipeline(reads, L=L, host_threads=16); p.prof_enable(True)
    t1 = time.perf_counter(); p.pre_process(); t2 = time.perf_counter(); d = p.result_digest(); torch.cuda.synchronize(); t3 = time.perf_counter()
    print("   lists wait %.2f ms" % p.stat("t_bk_lists_wait"), d, flush=True)
    s = [p.stat(k) for k in ("rounds", "passes")]; q = [p.prof_read(k) for k in ("classify_pack", "sketch_reads", "radix_pass", "sketch_contigs", "find_next", "dict_build", "realign_windows", "consensus", "cindex_build", "realign_reads")]
    t4 = time.perf_counter(); p.close(); torch.cuda.synchronize(); t5 = time.perf_counter()
    print("create %.2f pre_process %.2f digest %.2f stats %.2f close %.2f total %.2f ms" % tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0)), flush=True)
