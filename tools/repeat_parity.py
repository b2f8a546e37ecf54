"""Pipeline against the sequential oracle on a repeat-rich read set: a short genome with a many-copy segment, tandem
repeats and low-complexity stretches at very high coverage -- groups of tens of thousands of reads, long runs of equal
minimizers in the contig index, long Stage-2 bins."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

n, L = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 100
gpu = "--oracle-only" not in sys.argv
sub_rate = float(sys.argv[3]) if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else 0.004
from minicom_amd import synth
reads = synth.repeat_rich_reads(n, L, sub_rate)          # (the generator moved to minicom_amd/synth.py: tests/test_gpu_scale.py runs it too)
t00 = time.time()
def _beat():
    while True:
        time.sleep(60); print("... %d s" % (time.time() - t00), flush=True)
threading.Thread(target=_beat, daemon=True).start()
import oracle
t = time.time(); o = oracle.Pipeline(reads); o.run_all(); print("oracle %.1f s, contigs %d, largest contig %d members, sg %d, passes %d" % (
    time.time() - t, len(o.contigs()), max((len(m) for _, m in o.contigs()), default=0), len(o.id_list("sg")), o.counter("passes")), flush=True)
if gpu:
    from minicom_amd.pipeline import Pipeline
    t = time.time(); p = Pipeline(reads, host_threads=16); p.pre_process(); print("gpu %.2f s  big_bins %d big_bin_reads %d" % (time.time() - t, p.stat("big_bins"), p.stat("big_bin_reads")), flush=True)
    oc, pc = o.contigs(), p.contigs()
    assert len(oc) == len(pc), (len(oc), len(pc))
    bad = sum(1 for (r0, m0), (r1, m1) in zip(oc, pc) if r0 != r1 or not np.array_equal(m0, m1))
    print("contigs differing:", bad)
    for name in ("sg", "fpA", "fpT"):
        print(name, np.array_equal(o.id_list(name), p.id_list(name)))
    assert bad == 0
if "--e2e" in sys.argv:                                      # FASTQ -> .minicom -> reads, order-preserving and default
    import tempfile
    from minicom_amd import container, synth
    with tempfile.TemporaryDirectory() as td:
        fq = os.path.join(td, "a.fastq"); synth.write_fastq(fq, reads)
        for order in (True, False):
            arc, out = os.path.join(td, "a.minicom"), os.path.join(td, "a.reads")
            container.compress_fastq(fq, arc, order=order, codec="gz", threads=16)
            assert container.decompress_file(arc, out, threads=16) == n
            got = np.frombuffer(open(out, "rb").read(), dtype=np.uint8).reshape(n, L + 1)[:, :L]
            ok = np.array_equal(got, reads) if order else np.array_equal(np.sort(np.ascontiguousarray(got).view("S%d" % L).ravel()), np.sort(reads.view("S%d" % L).ravel()))
            print("round trip (order=%s): %s, %.3f bits/base" % (order, ok, 8 * os.path.getsize(arc) / reads.size), flush=True)
            assert ok
