"""Times mcom_pack_contigs / mcom_sketch_contigs alone on synthetic contigs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ctypes as C
import minicom_amd
ctx = minicom_amd.Context(0)
n, ln = 5_000_000, 300
lens = np.full(n, ln, dtype=np.int64); lens[::3] = 170
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
words = (2 * lens + 63) // 64 + 1
coff = np.concatenate([[0], np.cumsum(words)]).astype(np.uint64)
tot = int(off[-1]); tw = int(coff[-1])
seq = torch.from_numpy(np.frombuffer(b"ACGT", dtype=np.uint8)).cuda()[torch.randint(0, 4, (tot + 16,), device="cuda")]
d_off = torch.from_numpy(off.view(np.int64)).cuda(); d_coff = torch.from_numpy(coff.view(np.int64)).cuda()
cbits = torch.zeros(tw + 2, dtype=torch.int64, device="cuda")
lib = ctx.lib
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rc = lib.mcom_pack_contigs(ctx._h, C.c_void_p(seq.data_ptr()), C.c_void_p(d_off.data_ptr()), C.c_void_p(d_coff.data_ptr()), n, tw, C.c_void_p(cbits.data_ptr()))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"pack: rc={rc} {tot/1e9:.2f} G bases in {dt*1e3:.2f} ms = {tot/dt/1e9:.1f} Gbases/s", flush=True)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    moff, rec = ctx.sketch_contigs(seq, d_off, n, 44, 31)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"sketch: {tot/1e9:.2f} G bases, {rec.shape[0]/1e6:.1f} M minimizers in {dt*1e3:.2f} ms = {tot/dt/1e9:.1f} Gbases/s", flush=True)
