"""GPU parity on inputs at the edges: one read, identical reads, reads that are all one base or all N, the shortest
and longest read lengths, many exact duplicates, N-heavy reads.  Final contigs, members and every id list must equal
the sequential oracle's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cases():
    from minicom_amd import synth
    yield "one_read", synth.synth_reads(1, 1, 100)
    yield "two_identical", np.repeat(synth.synth_reads(2, 1, 100), 2, axis=0)
    yield "three_reads", synth.synth_reads(3, 3, 150)
    yield "all_A", np.full((50, 100), ord("A"), dtype=np.uint8)
    yield "all_N", np.full((20, 100), ord("N"), dtype=np.uint8)
    yield "L32", synth.synth_reads(4, 3000, 32)
    yield "L40", synth.synth_reads(4, 3000, 40)
    yield "L64", synth.synth_reads(5, 3000, 64)
    yield "L75", synth.synth_reads(10, 3000, 75)                 # 70 <= L < 80: k = 17 with w = L/2 - k
    yield "L101", synth.synth_reads(11, 3000, 101)
    yield "L151", synth.synth_reads(12, 2500, 151)
    yield "L199", synth.synth_reads(13, 2000, 199)
    yield "L255", synth.synth_reads(6, 1500, 255)
    yield "L256", synth.synth_reads(7, 1500, 256)
    yield "duplicates", np.repeat(synth.synth_reads(8, 40, 100), 50, axis=0)
    r = synth.synth_reads(9, 2000, 100, plumbing=True)
    r[::3, 10:60] = ord("N")
    yield "N_heavy", r


@pytest.mark.parametrize("name", [n for n, _ in _cases()])
def test_pipeline_equals_oracle_at_the_edges(name):
    import oracle
    from minicom_amd.pipeline import Pipeline
    reads = dict(_cases())[name]
    o = oracle.Pipeline(reads); o.run_all()
    p = Pipeline(reads, host_threads=2); p.pre_process()
    oc, pc = o.contigs(), p.contigs()
    assert len(oc) == len(pc)
    for c, ((r0, m0), (r1, m1)) in enumerate(zip(oc, pc)):
        assert r0 == r1 and np.array_equal(m0, m1), (name, c)
    for lst in ("sg", "fpA", "fpT", "fpN", "allA", "allT", "allN", "Nfile"):
        assert np.array_equal(o.id_list(lst), p.id_list(lst)), (name, lst)
    p.close(); o.close()


@pytest.mark.parametrize("first,cap", [(4096, 1 << 20), (8, 1 << 20), (8, 16)])
def test_special_reads_reach_the_host_as_a_list_or_as_the_class_array(first, cap):
    """Reads of another class than 0 are listed on the device (mcom_special_reads): a few entries travel with the count, more in a
    second copy, and a list that outgrows its room falls back to the class array.  All three paths must give the oracle's id lists."""
    import ctypes as C
    import oracle
    from minicom_amd import synth
    from minicom_amd.pipeline import Pipeline
    r = synth.synth_reads(21, 3000, 100, plumbing=True)
    r[5:40] = ord("A"); r[100:130] = ord("T"); r[700:712] = ord("N")
    r[1000:1300:7, 10:70] = ord("N")
    r[2000:2030, 3:] = ord("A")                                      # near poly-A
    o = oracle.Pipeline(r); o.run_all()
    p = Pipeline(r, host_threads=2)
    p.lib.mcomh_test_special_capacity.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    assert p.lib.mcomh_test_special_capacity(p._h, first, cap) == 0
    p.pre_process()
    n_special = sum(len(o.id_list(l)) for l in ("fpA", "fpT", "fpN", "allA", "allT", "allN", "Nfile"))
    assert n_special > 100
    for lst in ("sg", "fpA", "fpT", "fpN", "allA", "allT", "allN", "Nfile"):
        assert np.array_equal(o.id_list(lst), p.id_list(lst)), lst
    for (r0, m0), (r1, m1) in zip(o.contigs(), p.contigs()):
        assert r0 == r1 and np.array_equal(m0, m1)
    p.close(); o.close()


def test_pipeline_on_an_empty_read_set():
    """No reads (the reference reads none and goes on, bseq.c:38-66): every stage runs and leaves nothing."""
    from minicom_amd.pipeline import Pipeline
    p = Pipeline(np.zeros((0, 100), dtype=np.uint8))
    p.pre_process()
    assert p.contigs() == []
    for lst in ("sg", "fpA", "fpT", "fpN", "allA", "allT", "allN", "Nfile"):
        assert len(p.id_list(lst)) == 0
    p.close()


def test_last_pass_without_appends_still_sorts_the_appends_before_it():
    """Stage 2 sorts every contig at the start of each scan (kthread_hash_realign.c:318): when the last pass claims
    nothing, the members appended by the pass before it must still end up sorted (seen at L = 32)."""
    import oracle
    from minicom_amd import synth
    from minicom_amd.pipeline import Pipeline
    reads = synth.synth_reads(4, 3000, 32)
    o = oracle.Pipeline(reads); o.run_all()
    p = Pipeline(reads); p.pre_process()
    assert p.stat("passes") == o.counter("passes") == 2
    for (r0, m0), (r1, m1) in zip(o.contigs(), p.contigs()):
        assert r0 == r1 and np.array_equal(m0, m1)
    p.close(); o.close()


@pytest.mark.parametrize("name", [n for n, _ in _cases()])
@pytest.mark.parametrize("order", [False, True])
def test_stream_round_trip_at_the_edges(tmp_path, name, order):
    """cluster_dump + decompress give the reads back (as a multiset, or in input order with -p) for every edge input."""
    from minicom_amd.pipeline import Pipeline, decompress
    reads = dict(_cases())[name]
    n, L = reads.shape
    p = Pipeline(reads, host_threads=2); p.pre_process()
    d = tmp_path / "s"; d.mkdir()
    p.cluster_dump(str(d), order=order)
    p.close()
    out = tmp_path / "reads.txt"
    assert decompress(str(d), str(out), order=order) == n
    got = np.frombuffer(out.read_bytes(), dtype=np.uint8).reshape(n, L + 1)[:, :L]
    if order:
        assert np.array_equal(got, reads)
    else:
        a = np.sort(np.ascontiguousarray(got).view("S%d" % L).ravel()); b = np.sort(np.ascontiguousarray(reads).view("S%d" % L).ravel())
        assert np.array_equal(a, b)
