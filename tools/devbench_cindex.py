"""Times mcom_cindex_build alone on random packed contigs of the benchmark's shape (2.23 M contigs, 0.6 G windows, L = 150):
    python tools/devbench_cindex.py [n_contigs] [mean_len] [runs]
under `rocprofv3 --kernel-trace --stats` the per-kernel split of the build."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from minicom_amd.hip import Context  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_234_000
mean = int(sys.argv[2]) if len(sys.argv) > 2 else 417
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
L = 150
ctx = Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(7)
lens = torch.randint(L, 2 * mean - L, (n,), device="cuda", generator=g, dtype=torch.int64)
words = (2 * lens + 63) // 64 + 1
coff = torch.zeros(n, dtype=torch.int64, device="cuda"); coff[1:] = torch.cumsum(words, 0)[:-1]
tw = int(words.sum())
cbits = torch.randint(-2**62, 2**62, (tw + 2,), device="cuda", generator=g, dtype=torch.int64)
nwin = lens - L + 1
woff = torch.zeros(n + 1, dtype=torch.int64, device="cuda"); woff[1:] = torch.cumsum(nwin, 0)
nw = int(woff[-1])
print(f"{n} contigs, {nw} windows, {tw * 8 / 1e9:.2f} GB packed", flush=True)
for it in range(runs):
    torch.cuda.synchronize(); t0 = time.time()
    keys, geom = ctx.cindex_build(cbits, coff, woff, nw, L)
    ctx.sync(); torch.cuda.synchronize()
    dt = (time.time() - t0) * 1e3
    print(f"run {it}: {dt:.2f} ms, table {keys.numel() * 8 / 1e9:.2f} GB, parts {geom & 0xFFFF} lines {(geom >> 16) & 0xFFFF}", flush=True)
    main = keys[8:8 + (geom & 0xFFFF) * ((geom >> 16) & 0xFFFF) * 8]                 # (the unused part of the extension area is never written)
    now = (int(main.sum()), int((main[0::8] & 0xFF).sum()))                          # equal homes land in any order: sums are order-free
    if it == 0:
        ref = now
        print("   entries in lines:", now[1], flush=True)
    else:
        print("   same content" if now == ref else f"   CONTENT DIFFERS {now} {ref}", flush=True)
    del keys
