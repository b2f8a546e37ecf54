"""FASTQ -> .minicom -> reads at a few million reads (all three modes), timing the stages."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from minicom_amd import container, synth

n, L = int(sys.argv[1]), 150
reads = np.concatenate([synth.synth_reads(501, n, L), synth.synth_reads(502, n // 50, L, plumbing=True)])
with tempfile.TemporaryDirectory() as td:
    fq = os.path.join(td, "a.fastq"); synth.write_fastq(fq, reads)
    for mode in ("default", "order", "paired"):
        arc, out, out2 = os.path.join(td, mode + ".minicom"), os.path.join(td, "o1"), os.path.join(td, "o2")
        t = time.time()
        if mode == "paired":
            sizes = container.compress_fastq(fq, arc, path2=fq, codec="xz", threads=16)
        else:
            sizes = container.compress_fastq(fq, arc, order=(mode == "order"), codec="xz", threads=16)
        tc = time.time() - t; t = time.time()
        cnt = container.decompress_file(arc, out, out2 if mode == "paired" else None, threads=16)
        td_ = time.time() - t
        got = np.frombuffer(open(out, "rb").read(), dtype=np.uint8).reshape(-1, L + 1)[:, :L]
        if mode == "order":
            ok = np.array_equal(got, reads)
        elif mode == "default":
            ok = np.array_equal(np.sort(np.ascontiguousarray(got).view("S%d" % L).ravel()), np.sort(np.ascontiguousarray(reads).view("S%d" % L).ravel()))
        else:
            got2 = np.frombuffer(open(out2, "rb").read(), dtype=np.uint8).reshape(-1, L + 1)[:, :L]
            ok = np.array_equal(got, got2) and np.array_equal(np.sort(np.ascontiguousarray(got).view("S%d" % L).ravel()), np.sort(np.ascontiguousarray(reads).view("S%d" % L).ravel()))
        print(f"{mode}: reads {len(reads)} archive {os.path.getsize(arc)} B = {8 * os.path.getsize(arc) / (reads.size * (2 if mode == 'paired' else 1)):.3f} bits/base, compress {tc:.1f} s, decompress {td_:.1f} s, lossless {ok}", flush=True)
        assert ok and cnt == len(reads)
