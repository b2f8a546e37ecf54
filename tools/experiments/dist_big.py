import os, sys, tempfile, pathlib
import numpy as np
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_distributed as T
from minicom_amd import synth
reads = synth.synth_reads(31337, 3_000_000, 150)
single, st = T._single(reads)
print("single:", {k: len(v) for k, v in single.items()}, st, flush=True)
with tempfile.TemporaryDirectory() as d:
    outs = T._run_ranks(pathlib.Path(d), reads, 4)
    for r, o in enumerate(outs):
        for k in single:
            assert np.array_equal(single[k], o[k]), (r, k)
print("4 ranks x 750k reads: every rank holds the single-GPU result")
