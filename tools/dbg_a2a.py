"""all_to_all_single to self at growing sizes: how much of the payload arrives?"""
import os, sys
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29546")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for mb in (64, 256, 512, 1024, 1536, 2048, 3072, 4096):
    n = mb * (1 << 20) // 8
    src = torch.arange(n, dtype=torch.int64, device="cuda")
    dst = torch.full((n,), -1, dtype=torch.int64, device="cuda")
    dist.all_to_all_single(dst, src, output_split_sizes=[n], input_split_sizes=[n])
    torch.cuda.synchronize()
    ok = int((dst == src).sum()); print(f"{mb} MiB: {ok}/{n} arrived ({ok/n:.3f})", flush=True)
    del src, dst
dist.destroy_process_group()
