"""GPU parity of the whole hot path (host driver + HIP kernels) against the REFERENCE's own stage dumps:
state after kt_for_reads, kt_for_bucket, combine_cluster and every realign pass, byte for byte."""
import gzip
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _golden_reads(golden_dir, tag):
    with gzip.open(os.path.join(golden_dir, tag + ".reads.gz"), "rb") as f:
        rows = f.read().split(b"\n")[:-1]
    return np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), len(rows[0])).copy()


def _first_diff(got, want):
    gl, wl = got.split(b"\n"), want.split(b"\n")
    for i, (a, b) in enumerate(zip(gl, wl)):
        if a != b:
            stage = [x for x in wl[:i] if x.startswith(b"STAGE")][-1:]
            return f"first difference at line {i} after {stage}: got {a[:200]!r} want {b[:200]!r}"
    return f"length differs: {len(gl)} vs {len(wl)} lines"


NONDEFAULT = dict(e=6, m=4, w=12, cbthr=9, max_rounds=3, step=5, maxthr=30, numdict=4)   # tests/golden/make_golden.py


@pytest.mark.parametrize("tag,params,threads", [("stages_L100", {}, 1), ("stages_L150", {}, 4), ("stages_L100_k24", dict(k=24), 2), ("stages_L40", {}, 2), ("stages_L75", {}, 2),
                                                ("stages_L100_params", NONDEFAULT, 3)])
def test_pipeline_stage_dumps_equal_reference(golden_dir, tmp_path, tag, params, threads):
    from minicom_amd.pipeline import Pipeline
    reads = _golden_reads(golden_dir, tag)
    with gzip.open(os.path.join(golden_dir, tag + ".dump.gz"), "rb") as f:
        want = f.read()
    p = Pipeline(reads, host_threads=threads, **params)
    out = str(tmp_path / "dump.txt")
    p.dump_stages(out)
    got = open(out, "rb").read()
    assert got == want, _first_diff(got, want)
    assert p.stat("rounds") >= 2 and p.stat("passes") >= 2 and p.stat("big_bins") == 0
    p.close()


def test_pipeline_equals_oracle_on_fresh_synthetic_reads():
    """A read set no fixture covers: final contigs, members and leftover singletons equal the oracle's."""
    import oracle
    from minicom_amd import synth
    from minicom_amd.pipeline import Pipeline
    reads = synth.synth_reads(777, 6000, 150, plumbing=True)
    o = oracle.Pipeline(reads); o.run_all()
    p = Pipeline(reads, host_threads=3); p.pre_process()
    oc, pc = o.contigs(), p.contigs()
    assert len(oc) == len(pc) and len(oc) > 20
    for (r0, m0), (r1, m1) in zip(oc, pc):
        assert r0 == r1 and np.array_equal(m0, m1)
    for name in ("sg", "fpA", "fpT", "fpN", "allA", "allT", "allN", "Nfile"):
        assert np.array_equal(o.id_list(name), p.id_list(name)), name
    p.close(); o.close()


def test_pipeline_equals_oracle_where_index_buckets_exceed_64_entries():
    """1.6 M reads: the contig-minimizer index has ~60 entries per bucket on average, many buckets above the 64 at
    which the reference's radix sort turns unstable (ksort.h:155); merging must still follow the reference's order."""
    import oracle
    from minicom_amd import synth
    from minicom_amd.pipeline import Pipeline
    reads = synth.synth_reads(4242, 1_600_000, 100)
    o = oracle.Pipeline(reads); o.run_all()
    p = Pipeline(reads, host_threads=16); p.pre_process()
    oc, pc = o.contigs(), p.contigs()
    assert len(oc) == len(pc) > 10000
    assert all(r0 == r1 and np.array_equal(m0, m1) for (r0, m0), (r1, m1) in zip(oc, pc))
    for name in ("sg", "fpA", "fpT"):
        assert np.array_equal(o.id_list(name), p.id_list(name)), name
    assert p.stat("merge_rounds") >= 5
    p.close(); o.close()


def test_pipeline_device_resident_input_and_lossless_accounting():
    """Reads generated in HBM (the bench path): every read ends in exactly one place."""
    import torch
    import minicom_amd
    from minicom_amd.pipeline import Pipeline
    n, L = 200000, 150
    ctx = minicom_amd.Context(0)
    a = ctx.synth_reads(1002, n, L)
    ctx.sync()
    p = Pipeline(a, L=L, host_threads=8)
    p.pre_process()
    seen = np.zeros(n, dtype=np.int32)
    for _, mem in p.contigs():
        np.add.at(seen, (mem >> np.uint64(32)).astype(np.int64), 1)
    for name in ("sg", "fpA", "fpT", "fpN", "allA", "allT", "allN", "Nfile"):
        np.add.at(seen, p.id_list(name).astype(np.int64), 1)
    assert int(seen.min()) == 1 and int(seen.max()) == 1
    # every member really lies on its contig within the thresholds the pipeline used
    reads = a.cpu().numpy()
    comp = np.zeros(256, dtype=np.uint8); comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
    bad = 0
    for ref, mem in p.contigs()[:300]:
        r = np.frombuffer(ref, dtype=np.uint8)
        for y in mem.tolist():
            rid, off, d = y >> 32, (y & 0xFFFFFFFF) >> 1, y & 1
            s = reads[rid]
            if d:
                s = comp[s][::-1]
            if int((r[off:off + L] != s).sum()) > L // 2:
                bad += 1
    assert bad == 0
    assert p.stat("t_gpu") > 0
    p.close(); ctx.close()


def _repeat_pileup_reads(seed, L, n_clean, n_noisy, starts=None):
    """A genome with one segment repeated at four loci, clean reads that assemble into contigs and noisy reads
    (5-9 substitutions) piled up on few start positions: they end as singletons that share dictionary keys, inside
    and outside the repeat, so Stage-2 bins grow long and the same long bin is visited from several windows."""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    comp = np.zeros(256, dtype=np.uint8); comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
    g = acgt[rng.integers(0, 4, 8000)]
    rep = acgt[rng.integers(0, 4, 2 * L + 40)]
    for at in (500, 2500, 4500, 6500):
        g[at:at + len(rep)] = rep
    rows = []
    for _ in range(n_clean):
        j = int(rng.integers(0, len(g) - L + 1))
        r = g[j:j + L].copy()
        if rng.random() < 0.3:
            r[rng.integers(0, L)] = acgt[rng.integers(0, 4)]
        rows.append(comp[r][::-1] if rng.random() < 0.5 else r)
    if starts is None:
        starts = np.r_[np.arange(0, len(g) - L, 173), 500 + np.arange(0, L + 40, 7), 2500 + np.arange(0, L + 40, 7)]
    for _ in range(n_noisy):
        j = int(starts[rng.integers(0, len(starts))])
        r = g[j:j + L].copy()
        for q in rng.integers(0, L, int(rng.integers(5, 10))):
            r[q] = acgt[rng.integers(0, 4)]
        rows.append(comp[r][::-1] if rng.random() < 0.5 else r)
    rows = np.stack(rows)
    return rows[rng.permutation(len(rows))].copy()


@pytest.mark.parametrize("L,maxsearch", [(100, 2), (150, 3), (100, 7)])
def test_stage2_bins_longer_than_maxsearch_follow_the_sequential_scan(L, maxsearch):
    """The reference scans the last `maxsearch` LIVE reads of a bin and removes claimed reads from every bin after the
    visit (kthread_hash_realign.c:388, :420-435): which reads a visit sees depends on what was claimed before.  With
    the limit forced low (test hook on both sides) the pipeline must still equal the sequential oracle."""
    import oracle
    from minicom_amd.pipeline import Pipeline
    reads = _repeat_pileup_reads(9000 + L + maxsearch, L, 2400, 2500)
    o = oracle.Pipeline(reads); o.force_maxsearch(maxsearch); o.run_all()
    p = Pipeline(reads, host_threads=4, maxsearch=maxsearch); p.pre_process()
    assert p.stat("maxsearch") == maxsearch == o.counter("maxsearch")
    assert p.stat("big_bins") > 0 and p.stat("big_bin_claims") > 0 and p.stat("big_bin_deferred") > 0
    oc, pc = o.contigs(), p.contigs()
    assert len(oc) == len(pc) > 3
    for c, ((r0, m0), (r1, m1)) in enumerate(zip(oc, pc)):
        assert r0 == r1 and np.array_equal(m0, m1), c
    for name in ("sg", "fpA", "fpT"):
        assert np.array_equal(o.id_list(name), p.id_list(name)), name
    p.close(); o.close()


def test_stage2_long_bins_at_the_reference_limit_of_2000():
    """No test hook: 12 000 noisy copies of two loci inside a four-fold repeat keep their first 17-mer, so the first
    dictionary holds bins of ~6000 reads, three times the reference's maxsearch of 2000 (preprocess.c:169-172), and
    each is visited from four windows."""
    import oracle
    from minicom_amd.pipeline import Pipeline
    L = 100
    rng = np.random.default_rng(4711)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    comp = np.zeros(256, dtype=np.uint8); comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
    g = acgt[rng.integers(0, 4, 8000)]
    rep = acgt[rng.integers(0, 4, 2 * L + 40)]
    for at in (500, 2500, 4500, 6500):
        g[at:at + len(rep)] = rep
    rows = []
    for _ in range(2400):
        j = int(rng.integers(0, len(g) - L + 1))
        r = g[j:j + L]
        rows.append(comp[r][::-1] if rng.random() < 0.5 else r.copy())
    for _ in range(12000):
        r = g[(520, 2533)[int(rng.integers(0, 2))]:][:L].copy()
        for q in rng.integers(17, L, int(rng.integers(10, 15))):
            r[q] = acgt[rng.integers(0, 4)]
        rows.append(r)
    reads = np.stack(rows)[rng.permutation(len(rows))].copy()
    o = oracle.Pipeline(reads); o.run_all()
    p = Pipeline(reads, host_threads=4); p.pre_process()
    assert p.stat("maxsearch") == 2000 and p.stat("big_bins") > 0 and p.stat("big_bin_deferred") > 0
    assert p.stat("big_bin_claims") > 4000
    oc, pc = o.contigs(), p.contigs()
    assert len(oc) == len(pc)
    for c, ((r0, m0), (r1, m1)) in enumerate(zip(oc, pc)):
        assert r0 == r1 and np.array_equal(m0, m1), c
    assert np.array_equal(o.id_list("sg"), p.id_list("sg"))
    p.close(); o.close()


def test_packed_rows_with_forwarded_minimizers_equal_resketching():
    """The multi-GPU entry: packed rows handed over together with their minimizers (mcomh_set_records) must give what
    sketching the rows again gives."""
    import torch
    import minicom_amd
    from minicom_amd import synth
    from minicom_amd.pipeline import Pipeline
    L, n = 150, 60000
    reads = synth.synth_reads(2718, n, L)                                         # no N: every read is kept, as after the exchange
    ctx = minicom_amd.Context(0)
    out = ctx.process_reads(torch.from_numpy(reads).cuda(), L, 31)
    ctx.sync()
    assert int((out["cls"] != 0).sum()) == 0
    rows = out["packed"].contiguous()
    x = out["rec"][:, 0].contiguous()
    ylow = (out["rec"][:, 1] & 0xFFFFFFFF).to(torch.int32).contiguous()
    a = Pipeline(rows, L=L, packed=True, host_threads=4); a.pre_process()
    b = Pipeline(rows, L=L, packed=True, records=(x, ylow), host_threads=4); b.pre_process()
    ca, cb = a.contigs(), b.contigs()
    assert len(ca) == len(cb) > 100
    assert all(r0 == r1 and np.array_equal(m0, m1) for (r0, m0), (r1, m1) in zip(ca, cb))
    assert np.array_equal(a.id_list("sg"), b.id_list("sg"))
    a.close(); b.close(); ctx.close()


@pytest.mark.parametrize("kind", ["uniform_150", "uniform_100", "repeats"])
def test_stage2_join_equals_stage2_through_the_table(kind):
    """Round 5: Stage 2 on one GPU can run as a partition-local join of the index entries with the singletons' keys (stage2_join = 1: no
    table; later passes from the candidates the first pass deferred) -- csrc/realign.hip, mcom_realign_join / mcom_realign_deferred.  The
    default runs realign_hash_search's lookups through the table in every pass (kthread_hash_realign.c:316-508 either way): the
    whole result must be the same, pass for pass (the digest covers strings, member lists in appending order, offsets and lists).
    Repeat-rich reads have dictionary bins above maxsearch: there the join steps back by itself and builds the table."""
    from minicom_amd import synth
    from minicom_amd.pipeline import Pipeline
    if kind == "repeats":
        reads = synth.repeat_rich_reads(400_000, 100, 0.02)
    else:
        L = int(kind.split("_")[1])
        reads = np.concatenate([synth.synth_reads(311, 1_500_000, L, sub_rate=0.02), synth.synth_reads(312, 4000, L, plumbing=True)])
    a = Pipeline(reads, host_threads=4, stage2_join=1); a.pre_process()
    b = Pipeline(reads, host_threads=4); b.pre_process()
    assert a.result_digest() == b.result_digest()
    assert b.stat("join_passes") == 0 and a.stat("passes") == b.stat("passes") >= 2
    if kind == "repeats":
        assert a.stat("join_fallbacks") >= 1                               # (its own count per (28 key bits, lane) is an upper bound of a bin: it steps back where the exact dictionaries may still find none above the limit)
    else:
        assert a.stat("join_fallbacks") == 0 and a.stat("join_passes") == a.stat("passes") and a.stat("join_deferred") > 0
    a.close(); b.close()


def _degenerate_sets(L):
    """Read sets at the edges of the domain: a handful of reads, nothing but copies of one read (one huge group), homopolymers (the special-read
    files of kthread_reads.c:84-126), nothing but N (Nfile, :219-224), reads just below and above the 0.4 L limit of N, a read and its reverse
    complement, and a two-letter genome (every k-mer everywhere: long index runs and bins)."""
    from minicom_amd import synth
    rng = np.random.default_rng(L)
    base = synth.synth_reads(4242 + L, 400, L)
    comp = np.zeros(256, dtype=np.uint8); comp[list(b"ACGTN")] = list(b"TGCAN")
    sets = {}
    for n in (1, 2, 3, 65):
        sets["first_%d" % n] = base[:n].copy()
    sets["copies_of_one"] = np.repeat(base[:1], 300, axis=0)
    sets["homopolymers"] = np.concatenate([np.full((40, L), ord(c), dtype=np.uint8) for c in "ACGT"] + [base[:50]])
    sets["all_N"] = np.full((30, L), ord("N"), dtype=np.uint8)
    withn = base[:200].copy()
    for i in range(200):                                                       # N counts from 0 to just above 0.4 L, anywhere in the read
        k = (i * (int(0.4 * L) + 3)) // 199
        withn[i, rng.choice(L, size=k, replace=False)] = ord("N")
    sets["N_around_the_limit"] = withn
    sets["with_reverse_complements"] = np.concatenate([base[:150], comp[base[:150, ::-1]]])
    two = np.frombuffer(b"AC", dtype=np.uint8)[rng.integers(0, 2, size=(1, 4 * L))][0]
    sets["two_letter_genome"] = np.stack([two[o:o + L] for o in rng.integers(0, 3 * L, size=300)])
    return sets


@pytest.mark.parametrize("L", [40, 100, 150, 256])
def test_degenerate_read_sets_equal_the_oracle_and_survive_the_round_trip(L, tmp_path):
    """The edges of the domain through the whole path: final contigs, member lists and every id list equal the sequential oracle's, and the
    stream files decode to the multiset of reads that went in (N restored).  L = 256 is the longest read the row layout holds."""
    import oracle
    from minicom_amd.pipeline import Pipeline, decompress
    for name, reads in _degenerate_sets(L).items():
        reads = np.ascontiguousarray(reads)
        o = oracle.Pipeline(reads); o.run_all()
        p = Pipeline(reads, host_threads=2); p.pre_process()
        oc, pc = o.contigs(), p.contigs()
        assert len(oc) == len(pc), (name, len(oc), len(pc))
        for (r0, m0), (r1, m1) in zip(oc, pc):
            assert r0 == r1 and np.array_equal(m0, m1), name
        for lst in ("sg", "fpA", "fpT", "fpN", "allA", "allT", "allN", "Nfile"):
            assert np.array_equal(o.id_list(lst), p.id_list(lst)), (name, lst)
        td = str(tmp_path / ("%s_%d" % (name, L))); os.makedirs(td)
        p.cluster_dump(td)
        out = os.path.join(td, "reads.txt")
        assert decompress(td, out) == reads.shape[0], name
        got = np.frombuffer(open(out, "rb").read(), dtype=np.uint8).reshape(reads.shape[0], L + 1)[:, :L]
        a = np.sort(np.ascontiguousarray(got).view("S%d" % L).ravel()); b = np.sort(reads.view("S%d" % L).ravel())
        assert np.array_equal(a, b), name
        p.close(); o.close()


def test_no_reads_at_all(tmp_path):
    """An empty read set is a legal input of the library (the reference's script refuses an empty file before it gets that far): no contigs, a
    digest of zeros, stream files that decode to nothing."""
    from minicom_amd.pipeline import Pipeline, decompress
    for L in (100, 150):
        p = Pipeline(np.zeros((0, L), dtype=np.uint8), L=L, host_threads=2); p.pre_process()
        assert len(p.contigs()) == 0 and list(p.result_digest()) == [0] * 8 and p.id_list("sg").size == 0
        td = str(tmp_path / str(L)); os.makedirs(td)
        p.cluster_dump(td)
        assert decompress(td, os.path.join(td, "o.txt")) == 0 and os.path.getsize(os.path.join(td, "o.txt")) == 0
        p.close()

