"""What this card's HBM gives a plain stream: 16 GB written (fill), read (sum), copied (read + write): python tools/ubench_bw.py"""
import time
import torch
n = 2_000_000_000
a = torch.empty(n, dtype=torch.int64, device="cuda"); b = torch.empty(n, dtype=torch.int64, device="cuda")
for name, fn, nbytes in (("fill (write)", lambda: a.fill_(7), 8 * n), ("copy (read + write)", lambda: b.copy_(a), 16 * n), ("sum (read)", lambda: a.sum(), 8 * n)):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    dt = (time.time() - t) / 5
    print(f"{name:22s} {nbytes / dt / 1e12:.2f} TB/s ({dt * 1e3:.2f} ms)", flush=True)
