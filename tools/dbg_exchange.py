import os, sys
sys.path.insert(0, "/root/repo")
import torch, torch.distributed as dist
import minicom_amd
from minicom_amd.distributed import exchange_by_bucket
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
n, L = int(sys.argv[1]), 150
ctx = minicom_amd.Context(0)
reads = ctx.synth_reads(1002, n, L); ctx.sync()
out = ctx.process_reads(reads, L, 31, rid0=0)
keep = (out["cls"] == 0).nonzero().squeeze(1)
x = out["rec"][:, 0][keep]
pk = out["packed"][keep]
rids, rows = exchange_by_bucket(x, keep, pk)
torch.cuda.synchronize()
print("n", n, "kept", keep.numel(), "rows equal", bool(torch.equal(rows, pk)), "rids equal", bool(torch.equal(rids, keep)))
bad = (rows != pk).any(dim=1).nonzero().squeeze(1)
print("first bad rows", bad[:5].tolist(), "count", bad.numel())
dist.destroy_process_group()
