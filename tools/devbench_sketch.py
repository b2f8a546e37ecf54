"""Times mcom_sketch_contigs on synthetic short contigs (the first merge round's shape): devbench_sketch.py [N] [LEN] [W] [K]"""
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from minicom_amd.hip import Context  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
ln = int(sys.argv[2]) if len(sys.argv) > 2 else 200
w = int(sys.argv[3]) if len(sys.argv) > 3 else 44
k = int(sys.argv[4]) if len(sys.argv) > 4 else 31
mode = int(sys.argv[5]) if len(sys.argv) > 5 else 0        # 0 default (prefix ring for odd k), 1 wave per string, 2 the 64-bit ring
ctx = Context(0)
ctx.set_sketch_kernel(mode)
if len(sys.argv) > 6:
    ctx.set_sketch_prefix_bits(int(sys.argv[6]))
g = torch.Generator(device="cuda"); g.manual_seed(5)
lens = torch.randint(ln - 40, ln + 60, (n,), device="cuda", generator=g, dtype=torch.int64)
off = torch.zeros(n + 1, dtype=torch.int64, device="cuda"); off[1:] = torch.cumsum(lens, 0)
total = int(off[-1])
codes = torch.randint(0, 4, (total + 64,), device="cuda", generator=g, dtype=torch.int64)
seq = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")[codes]
import ctypes as C  # noqa: E402
cap = 16 * n
moff = torch.empty(n + 1, dtype=torch.int32, device="cuda")
out = ctx.empty_records(cap)
tot = C.c_uint64(0)
for it in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    rc = ctx.lib.mcom_sketch_contigs(ctx._h, ctx._p(seq), ctx._p(off), None, n, w, k, 0, ctx._p(moff), ctx._p(out), cap, C.byref(tot))
    ctx.sync(); torch.cuda.synchronize()
    dt = (time.time() - t0) * 1e3
    print(f"run {it}: rc {rc} {dt:.2f} ms, {tot.value} minimizers, {total / dt / 1e6:.1f} Gpos/s", flush=True)
