"""Times the member-parallel .fastq.gz ingest alone (host/mcom_fastq_gz.cpp) with its phase totals (MCOM_GZ_TRACE=1):
   python tools/gz_ingest_probe.py [reads] [zlib level]      (GPU box; files under /tmp)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import minicom_amd
from minicom_amd.pipeline import Pipeline
from tools.e2e import gzip_members, _append_fastq, host_cores

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L = 150
fq = "/tmp/probe.fastq"
ctx = minicom_amd.Context(0)
with open(fq, "wb") as out:
    for lo in range(0, n, 4_000_000):
        cnt = min(4_000_000, n - lo)
        _append_fastq(out, ctx.synth_reads(1002, n, L, first=lo, count=cnt).cpu().numpy(), lo)
ctx.close()
gzp = "/tmp/probe_l%d.fastq.gz" % level
t = time.time(); m = gzip_members(fq, gzp, level=level, threads=max(1, host_cores()))
print("members", m, "written in %.1f s" % (time.time() - t), "%.2f GB from %.2f GB" % (os.path.getsize(gzp) / 1e9, os.path.getsize(fq) / 1e9), flush=True)
os.remove(fq)
os.environ["MCOM_GZ_TRACE"] = "1"
for rep in range(3):
    torch.cuda.synchronize(); t = time.time()
    p = Pipeline.from_fastq(gzp)
    torch.cuda.synchronize(); dt = time.time() - t
    print("rep %d: %.3f s = %.1f Mreads/s (%d reads)" % (rep, dt, p.n / dt / 1e6, p.n), flush=True)
    p.close()
os.remove(gzp)
