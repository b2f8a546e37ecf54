"""Pipeline against the sequential oracle at a size between the test suite and the benchmark (the oracle needs ~20 s per
million reads; a heartbeat keeps the GPU box from taking the silent wait for a hang)."""
import sys, time, threading
sys.path.insert(0, '.')
import numpy as np
import oracle
from minicom_amd import synth
from minicom_amd.pipeline import Pipeline
n, L = int(sys.argv[1]), 100
reads = synth.synth_reads(4242, n, L)
t00 = time.time()
def _beat():
    while True:
        time.sleep(60); print("... %d s" % (time.time() - t00), flush=True)
threading.Thread(target=_beat, daemon=True).start()
t = time.time(); o = oracle.Pipeline(reads); o.run_all(); print("oracle", time.time() - t, flush=True)
t = time.time(); p = Pipeline(reads, host_threads=32); p.pre_process(); print("gpu", time.time() - t, flush=True)
oc, pc = o.contigs(), p.contigs()
print(len(oc), len(pc))
assert len(oc) == len(pc)
bad = sum(1 for (r0, m0), (r1, m1) in zip(oc, pc) if r0 != r1 or not np.array_equal(m0, m1))
print("contigs differing:", bad)
for name in ("sg", "fpA", "fpT"):
    print(name, np.array_equal(o.id_list(name), p.id_list(name)))
