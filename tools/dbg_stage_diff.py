"""First difference between the pipeline's and the oracle's stage dumps for one synthetic read set."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from minicom_amd import synth
from minicom_amd.pipeline import Pipeline

seed, n, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reads = synth.synth_reads(seed, n, L)
o = oracle.Pipeline(reads); o.dump_stages("/tmp/o.txt")
p = Pipeline(reads, host_threads=2); p.dump_stages("/tmp/p.txt")
a, b = open("/tmp/p.txt", "rb").read().split(b"\n"), open("/tmp/o.txt", "rb").read().split(b"\n")
print("lines", len(a), len(b))
stage = b""
for i, (x, y) in enumerate(zip(a, b)):
    if y.startswith(b"STAGE"):
        stage = y
    if x != y:
        print("first difference at line", i, "after", stage)
        print(" got ", x[:300])
        print(" want", y[:300])
        xs, ys = x.split(b" "), y.split(b" ")
        for j, (u, v) in enumerate(zip(xs, ys)):
            if u != v:
                print("  token", j, u[:40], v[:40]); break
        break
else:
    print("equal")
for k in ("rounds", "merge_rounds", "passes", "k", "maxsearch"):
    print(k, p.stat(k), o.counter(k) if k in ("k", "maxsearch", "passes") else "")
