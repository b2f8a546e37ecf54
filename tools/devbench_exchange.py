"""Times the pieces of the N > 1 step of bench.py on one GPU (world size 1, forced exchange)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import minicom_amd
from minicom_amd.pipeline import Pipeline
from minicom_amd.distributed import exchange_by_bucket
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000, 150
ctx = minicom_amd.Context(0)
reads = ctx.synth_reads(1002, n, L); ctx.sync()
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for it in range(3):
    t0 = T()
    out = ctx.process_reads(reads, L, 31, rid0=0); t1 = T()
    keep = (out["cls"] == 0).nonzero().squeeze(1)
    x = out["rec"][:, 0][keep]; t2 = T()
    _, rows = exchange_by_bucket(x, keep, out["packed"][keep]); t3 = T()
    del out
    p = Pipeline(rows, L=L, host_threads=64, packed=True)
    p.pre_process(); t4 = T()
    print(f"process_reads {1e3*(t1-t0):.0f}  select {1e3*(t2-t1):.0f}  exchange {1e3*(t3-t2):.0f}  pipeline {1e3*(t4-t3):.0f} ms "
          f"[reads {p.stat('t_reads'):.0f} bucket {p.stat('t_bucket'):.0f} combine {p.stat('t_combine'):.0f} realign {p.stat('t_realign'):.0f}]  "
          f"torch reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB  bk: sort {p.stat('t_bk_sort'):.0f} gpu {p.stat('t_bk_gpu'):.0f} replay {p.stat('t_bk_replay'):.0f} rounds {p.stat('rounds'):.0f} resketch {p.stat('resketch'):.0f} sg0 {p.stat('n_sg0'):.0f}", flush=True)
    p.close()
    if it == 1:
        torch.cuda.empty_cache()
dist.destroy_process_group()
