"""Developer micro-benchmark of the individual HIP stages (not the driver's bench.py)."""
import argparse
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import minicom_amd

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=20_000_000)
ap.add_argument("--L", type=int, default=150)
ap.add_argument("--k", type=int, default=31)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
ctx = minicom_amd.Context(0)
n, L, k = a.n, a.L, a.k


def timed(name, fn, bytes_=None):
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(a.reps):
        t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    s = f"{name:28s} {best*1e3:9.3f} ms  {n/best/1e6:9.1f} Mreads/s"
    if bytes_:
        s += f"  {bytes_/best/1e9:8.1f} GB/s"
    print(s, flush=True)
    return r


ascii_ = timed("synth", lambda: ctx.synth_reads(1002, n, L))
out = timed("process_reads (pack+sketch)", lambda: ctx.process_reads(ascii_, L, k), n * (L + 8 * ((2 * L + 63) // 64) + 16))
packed = out["packed"]
timed("sketch_reads only", lambda: ctx.sketch_reads(packed, L, k), n * (8 * ((2 * L + 63) // 64) + 16))
timed("sketch_reads k=16", lambda: ctx.sketch_reads(packed, L, 16), n * (8 * ((2 * L + 63) // 64) + 16))
rec = out["rec"]
g = timed("sort_group", lambda: ctx.sort_group(rec, L, k, k), n * 64)
print("singles", g["singles"].numel(), "groups", g["n_groups"], "members", g["members"].numel())
