"""Correctness at the sizes the benchmark is quoted on (BASELINE.json configs[1], configs[2]) and, against the sequential
oracle, at the largest size the oracle finishes in minutes.  The fixture-sized tests cannot see a 32-bit index that
overflows at 10^8 reads or a buffer that is sized wrong only there; these can.

What can be asserted without an oracle at that size (minicom_amd/check.py, plain torch on the device, no kernel shared
with the library): every read lies in exactly one place, every member lies on its contig and resembles it, and a second
run reproduces the first bit for bit (digest of strings, member words, offsets and lists)."""
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_digest(n, L, seed):
    """The sequential oracle's result digest of this very read set (oracle/digest_main.c, run in the build container: hours of one core), when
    tests/golden/scale_digests.json holds it; None otherwise."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scale_digests.json")
    if not os.path.exists(path):
        return None
    for e in json.load(open(path))["runs"]:
        if (e["n"], e["L"], e["seed"], e.get("coverage", 30)) == (n, L, seed, 30):
            return [int(v) for v in e["digest"]]
    return None


def _run_checked(n, L, seed, coverage=30, stats=None):
    import torch
    import minicom_amd
    from minicom_amd.check import check_result
    from minicom_amd.pipeline import Pipeline
    ctx = minicom_amd.Context(0)
    reads = ctx.synth_reads(seed, n, L, coverage=coverage)
    ctx.sync()
    t = time.time()
    p = Pipeline(reads, L=L, host_threads=8)
    p.pre_process()
    d1 = p.result_digest()
    print(f"\n[{n} x {L}] first run {time.time() - t:.1f} s: {d1[0]} contigs, {d1[2]} members, {d1[3]} unclustered", flush=True)
    t = time.time()
    res = check_result(p, reads, L)
    print(f"[{n} x {L}] checked in {time.time() - t:.1f} s: {res}", flush=True)
    if stats is not None:
        stats.update({k: p.stat(k) for k in ("contigs_bucket", "contigs_combine", "cix_entries", "rounds", "merge_rounds", "passes")})
    p.close()
    p = Pipeline(reads, L=L, host_threads=8)
    p.pre_process()
    d2 = p.result_digest()
    p.close()
    del reads
    torch.cuda.empty_cache()
    return res, d1, d2


def test_config1_100m_reads_of_150_bases():
    """BASELINE configs[1]: 100 M x 150 bp synthetic reads, default parameters, one GPU (the workload bench.py times)."""
    n, L = 100_000_000, 150
    res, d1, d2 = _run_checked(n, L, 1002)
    assert d1 == d2                                                        # deterministic, bit for bit
    want = _oracle_digest(n, L, 1002)
    if want is not None:                                                   # ... and equal to the sequential oracle's result on the same 100 M reads
        assert list(d1) == want
    else:
        print("(tests/golden/scale_digests.json has no oracle digest for this set: properties only)", flush=True)
    assert res["n_reads"] == n and res["members"] + res["n_sg"] + sum(res["n_" + k] for k in ("allA", "allT", "allN", "fpA", "fpT", "fpN", "Nfile")) == n
    assert res["members"] > 0.8 * n and res["n_contigs"] > 1_000_000       # 30x coverage: most reads end up in contigs
    assert res["max_mismatch"] <= L // 2 and res["mean_mismatch"] < 0.02 * L
    assert res["members_checked"] == res["members"]


def test_more_than_2_pow_24_contigs_on_one_card():
    """The reference packs (contig index << 8) + thread into 32 bits (kthread_bucket.c:458): 2^24 contigs per list, which a 500 M-read
    job (BASELINE configs[3], ~39 M first-round contigs) exceeds.  Here the id is the 32-bit index and the Stage-2 index entries take
    the bits they need from the position field: 80 M x 100 bp at 5 x coverage leaves > 2^24 contigs after the bucket stage on one
    card -- every read accounted for, every member on its contig, two runs bit-identical."""
    n, L = 80_000_000, 100
    st = {}
    res, d1, d2 = _run_checked(n, L, 1005, coverage=5, stats=st)
    print(f"contigs after the bucket stage {st['contigs_bucket']:.0f}, after merging {st['contigs_combine']:.0f}, index entries {st['cix_entries']:.0f}", flush=True)
    assert st["contigs_bucket"] > 2 ** 24 and st["merge_rounds"] >= 3 and st["passes"] >= 2
    assert d1 == d2
    assert res["n_reads"] == n and res["members"] + res["n_sg"] + sum(res["n_" + k] for k in ("allA", "allT", "allN", "fpA", "fpT", "fpN", "Nfile")) == n
    assert res["members"] > 0.5 * n and res["members_checked"] == res["members"]
    assert res["max_mismatch"] <= L // 2
    # ... and the multi-GPU path on the same input: 2 and 4 ranks (threads of this process on the one card, Comm.threads) end with
    # the single-GPU digest on every rank
    from minicom_amd.pipeline import pool_trim
    for world in (2, 4):
        pool_trim()                                                        # (the blocks the runs before left in the process's pools)
        assert _thread_ranks_digests(n, L, 1005, 5, world) == [d1] * world, world
    pool_trim()


def _thread_ranks_digests(n, L, seed, coverage, world):
    import threading
    import torch
    import minicom_amd
    from minicom_amd.distributed import Comm, DistPipeline
    ctx = minicom_amd.Context(0)
    reads = ctx.synth_reads(seed, n, L, coverage=coverage)
    ctx.sync()
    comms, hub = Comm.threads(world)
    out, errors = [None] * world, []

    def rank_main(rank):
        try:
            lo, hi = n * rank // world, n * (rank + 1) // world
            p = DistPipeline(reads[lo:hi], lo, n, comms[rank], L=L, device=0, host_threads=4)
            p.pre_process()
            out[rank] = p.result_digest()
            p.close()
        except Exception as e:                                                  # noqa: BLE001
            errors.append((rank, repr(e))); hub.abort()
    t0 = time.time()
    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errors, errors
    print(f"[{n} x {L}] {world} ranks: {time.time() - t0:.1f} s", flush=True)
    for c in comms:
        c.close()
    del reads
    torch.cuda.empty_cache()
    return out


def test_config2_67m_reads_of_100_bases():
    """BASELINE configs[2] in shape (the SRR445718 file itself cannot be fetched): 67 M x 100 bp."""
    n, L = 67_000_000, 100
    res, d1, d2 = _run_checked(n, L, 1003)
    assert d1 == d2
    want = _oracle_digest(n, L, 1003)
    if want is not None:
        assert list(d1) == want
    else:
        print("(tests/golden/scale_digests.json has no oracle digest for this set: properties only)", flush=True)
    assert res["members"] > 0.7 * n and res["n_contigs"] > 500_000
    assert res["max_mismatch"] <= L // 2 and res["mean_mismatch"] < 0.02 * L


def test_pipeline_equals_the_sequential_oracle_on_8m_reads():
    """8 M x 100 bp (configs[0] x 8): contig strings, member lists and lists identical to oracle/mcom_oracle.c, the
    restatement pinned on the reference's own dumps.  The oracle needs about three minutes of one core."""
    import threading
    import oracle
    from minicom_amd import synth
    from minicom_amd.pipeline import Pipeline
    n, L = 8_000_000, 100
    reads = synth.synth_reads(4242, n, L)
    t0 = time.time()
    stop = threading.Event()

    def beat():                                                            # a silent wait of minutes looks like a hang from outside
        while not stop.wait(45):
            print(f"... oracle running, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    th = threading.Thread(target=beat, daemon=True); th.start()
    o = oracle.Pipeline(reads); o.run_all()
    stop.set()
    print(f"\noracle: {time.time() - t0:.0f} s", flush=True)
    p = Pipeline(reads, host_threads=8); p.pre_process()
    ref, roff, mem, moff = p.contig_set()
    oc = o.contigs()
    assert len(oc) == len(roff) - 1 > 100_000
    assert b"".join(r for r, _ in oc) == ref.tobytes()
    assert np.array_equal(np.cumsum([0] + [len(r) for r, _ in oc]).astype(np.uint64), roff)
    assert np.array_equal(np.concatenate([m for _, m in oc]), mem)
    assert np.array_equal(np.cumsum([0] + [len(m) for _, m in oc]).astype(np.uint64), moff)
    for name in ("sg", "fpA", "fpT", "fpN", "allA", "allT", "allN", "Nfile"):
        assert np.array_equal(o.id_list(name), p.id_list(name)), name
    assert o.result_digest() == list(p.result_digest())                   # the digest restated in the oracle: what pins the 100 M-read runs above
    p.close(); o.close()


def _oracle_with_heartbeat(reads, label):
    import threading
    import oracle
    t0 = time.time()
    stop = threading.Event()

    def beat():
        while not stop.wait(45):
            print(f"... oracle ({label}) running, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    threading.Thread(target=beat, daemon=True).start()
    o = oracle.Pipeline(reads); o.run_all()
    stop.set()
    print(f"\noracle ({label}): {time.time() - t0:.0f} s", flush=True)
    return o


def _assert_equal_sets(o, p):
    ref, roff, mem, moff = p.contig_set()
    oc = o.contigs()
    assert len(oc) == len(roff) - 1
    assert b"".join(r for r, _ in oc) == ref.tobytes()
    assert np.array_equal(np.cumsum([0] + [len(r) for r, _ in oc]).astype(np.uint64), roff)
    assert np.array_equal(np.concatenate([m for _, m in oc]) if oc else np.zeros(0, np.uint64), mem)
    assert np.array_equal(np.cumsum([0] + [len(m) for _, m in oc]).astype(np.uint64), moff)
    for name in ("sg", "fpA", "fpT", "fpN", "allA", "allT", "allN", "Nfile"):
        assert np.array_equal(o.id_list(name), p.id_list(name)), name
    assert o.result_digest() == list(p.result_digest())


def test_pipeline_equals_the_sequential_oracle_on_2m_reads_of_150_bases():
    """The headline shape against the oracle above fixture size: 2 M x 150 bp with the plumbing extras (N, poly-A/T, N-heavy
    reads): five-word rows, sixteen bases per lane in the classifier, eight Stage-2 dictionaries, index buckets beyond 64
    entries -- contig strings, member lists and every list identical."""
    from minicom_amd import synth
    from minicom_amd.pipeline import Pipeline
    n, L = 2_000_000, 150
    reads = synth.synth_reads(5151, n, L, plumbing=True)
    o = _oracle_with_heartbeat(reads, "2 M x 150")
    p = Pipeline(reads, host_threads=8, overlap_screen=1); p.pre_process()
    assert len(o.contigs()) > 30_000 and p.stat("merge_rounds") >= 4 and p.stat("passes") >= 2
    _assert_equal_sets(o, p)
    assert p.stat("early_screen") >= 1                              # (the variant with the first pass's singleton gather + screen beside the index build)
    d_overlap = p.result_digest()
    p.close()
    p = Pipeline(reads, host_threads=8); p.pre_process()
    assert p.stat("early_screen") == 0 and p.result_digest() == d_overlap
    p.close(); o.close()


def test_read_stage_in_batches_over_two_streams_gives_the_same_result():
    """mcomh_params.read_batches = 1: classification and sketch of the reads in four batches over two streams (the classification of a
    batch beside the sketch of the one before) instead of one launch each -- an A/B switch, same arrays: records, classes, N lists and
    everything made of them.  6 M x 150 bp with N reads, poly-A/T and N-heavy reads (the switch takes 4 M reads or more on the device)."""
    import torch
    from minicom_amd import synth
    from minicom_amd.pipeline import Pipeline
    n, L = 6_000_000, 150
    reads = torch.from_numpy(synth.synth_reads(6262, n, L, plumbing=True)).cuda()
    digests = []
    for rb in (0, 1):
        p = Pipeline(reads, L=L, host_threads=8, read_batches=rb); p.pre_process()
        assert p.stat("read_batches") == (4 if rb else 0)
        digests.append(p.result_digest())
        p.close()
    assert digests[0] == digests[1]


@pytest.mark.parametrize("n,L,sub_rate", [(3_000_000, 100, 0.004), (2_000_000, 100, 0.04), (1_000_000, 150, 0.02)])
def test_pipeline_equals_the_sequential_oracle_on_repeat_rich_reads(n, L, sub_rate):
    """A 200 kb genome with a 2 kb segment in forty copies, a tandem repeat, poly-A and (AT)n stretches at >= 750 x coverage:
    groups beyond 65 535 members, contigs of 10^5 members, heavy runs of equal minimizers in the index, long Stage-2 bins and
    up to five passes (tools/repeat_parity.py of round 1, now in the suite)."""
    from minicom_amd import synth
    from minicom_amd.pipeline import Pipeline
    reads = synth.repeat_rich_reads(n, L, sub_rate)
    o = _oracle_with_heartbeat(reads, f"repeats {n} x {L} @ {sub_rate}")
    p = Pipeline(reads, host_threads=8); p.pre_process()
    print(f"repeat-rich {n} x {L}: {len(o.contigs())} contigs, largest {max((len(m) for _, m in o.contigs()), default=0)} members, "
          f"big_bins {p.stat('big_bins')}, big_bin_reads {p.stat('big_bin_reads')}, passes {p.stat('passes')}", flush=True)
    _assert_equal_sets(o, p)
    p.close(); o.close()


def test_paired_end_file_set_of_4m_reads_of_150_bases_keeps_every_pair(tmp_path):
    """BASELINE configs[4]'s mode at a size the fixtures do not reach: 2 x 2 M reads of 150 bases, the second half of the set standing in for
    the mates (the pairing is by read id: any second file will do).  The paired-end file set is written by the device encoders in four
    stream sets and decoded by this repo's decoder: every (read, mate) pair comes back, side by side."""
    import minicom_amd
    from minicom_amd.pipeline import Pipeline, decompress_pe
    n, L = 4_000_000, 150
    ctx = minicom_amd.Context(0)
    reads = ctx.synth_reads(1005, n, L)
    ctx.sync()
    host = reads.cpu().numpy()[:, :L]
    half = n // 2
    p = Pipeline(reads, L=L, host_threads=8, stream_sets=4)
    p.pre_process()
    d = tmp_path / "pe"; d.mkdir()
    t = time.time()
    p.cluster_dump(str(d), paired=True)
    print(f"\n[paired end, {n} reads] dumped in {time.time() - t:.2f} s (device {p.stat('t_dump_gpu') / 1e3:.3f} s)", flush=True)
    p.close()
    o1, o2 = tmp_path / "r1.txt", tmp_path / "r2.txt"
    assert decompress_pe(str(d), str(o1), str(o2)) == half
    a = np.frombuffer(o1.read_bytes(), dtype=np.uint8).reshape(half, L + 1)[:, :L]
    b = np.frombuffer(o2.read_bytes(), dtype=np.uint8).reshape(half, L + 1)[:, :L]
    got = np.sort(np.ascontiguousarray(np.concatenate([a, b], axis=1)).view(f"S{2 * L}").ravel())
    want = np.sort(np.ascontiguousarray(np.concatenate([host[:half], host[half:]], axis=1)).view(f"S{2 * L}").ravel())
    assert np.array_equal(got, want)


def test_paired_end_over_eight_ranks_2x16m_reads_of_150_bases(tmp_path):
    """BASELINE configs[4]'s mode and rank count at the size one card holds eight replicas of: paired end, 2 x 16 M reads of 150 bases
    (the reference treats the two files as one read pool, preprocess.c:60-68, and pairs by read id when it dumps,
    kthread_dump_pe.c:218-619), cut into eight shards -- mates live on different ranks -- and run by eight ranks (threads of this
    process on the one card, Comm.threads; every exchange of the pipeline happens, through process memory).  Every rank ends with the
    single-GPU digest; the paired-end file set rank 7 writes is decoded by this repo's decoder and every (read, mate) pair comes back."""
    import threading
    import torch
    import minicom_amd
    from minicom_amd.distributed import Comm, DistPipeline
    from minicom_amd.pipeline import Pipeline, decompress_pe, pool_trim
    n, L, world = 32_000_000, 150, 8
    half = n // 2
    ctx = minicom_amd.Context(0)
    reads = ctx.synth_reads(1005, n, L)                                  # (configs[4]'s seed; rows [0, n/2) = file 1, rows [n/2, n) = the mates)
    ctx.sync()
    p = Pipeline(reads, L=L, host_threads=8)
    p.pre_process()
    want = p.result_digest()
    p.close()
    pool_trim()
    comms, hub = Comm.threads(world)
    out, errors = [None] * world, []
    d = tmp_path / "pe"; d.mkdir()

    def rank_main(rank):
        try:
            lo, hi = n * rank // world, n * (rank + 1) // world
            q = DistPipeline(reads[lo:hi], lo, n, comms[rank], L=L, device=0, host_threads=4, stream_sets=4)
            q.pre_process()
            out[rank] = q.result_digest()
            if rank == world - 1:
                q.cluster_dump(str(d), paired=True)
            q.close()
        except Exception as e:                                                  # noqa: BLE001
            errors.append((rank, repr(e))); hub.abort()
    t0 = time.time()
    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errors, errors
    print(f"\n[paired end, {n} reads, {world} ranks] {time.time() - t0:.1f} s", flush=True)
    for c in comms:
        c.close()
    assert out == [want] * world
    host = reads.cpu().numpy()[:, :L]
    del reads
    torch.cuda.empty_cache(); pool_trim()
    o1, o2 = tmp_path / "r1.txt", tmp_path / "r2.txt"
    assert decompress_pe(str(d), str(o1), str(o2)) == half
    a = np.frombuffer(o1.read_bytes(), dtype=np.uint8).reshape(half, L + 1)[:, :L]
    b = np.frombuffer(o2.read_bytes(), dtype=np.uint8).reshape(half, L + 1)[:, :L]
    # the multiset of (read, mate) pairs, compared through a 64-bit hash per pair (two sorted arrays of 16 M strings of 300 bytes would
    # take 20 GB of host memory; a collision among 16 M hashes of 64 bits has probability 1e-5 and could only hide an error, not make one)
    def pair_hashes(x, y):
        out = np.empty(x.shape[0], dtype=np.uint64)
        wt = (np.arange(1, 2 * L + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) | np.uint64(1)
        with np.errstate(over="ignore"):
            for lo in range(0, x.shape[0], 1 << 20):
                hi = min(x.shape[0], lo + (1 << 20))
                row = np.concatenate([x[lo:hi], y[lo:hi]], axis=1).astype(np.uint64)
                h = (row * wt[None, :]).sum(axis=1, dtype=np.uint64)
                h ^= h >> np.uint64(29); h *= np.uint64(0xBF58476D1CE4E5B9); h ^= h >> np.uint64(32)
                out[lo:hi] = h
        return np.sort(out)
    got = pair_hashes(a, b)
    del a, b
    want_pairs = pair_hashes(host[:half], host[half:])
    assert np.array_equal(got, want_pairs)


def test_120m_reads_of_150_bases_beyond_the_sorted_index_placement():
    """Above ~93 M x 150 bp the partitions of the Stage-2 contig index outgrow the LDS-sorted placement (14 336 entries) and are placed by the
    scattered kernel (same index; profiles/r05_size_sweep.txt: 502 Mreads/s at 120 M, 496 at 150 M, 485 at 200 M against 529 at 100 M -- a
    slope, not a cliff).  The properties of the 100 M-read test at 120 M reads: every read in exactly one place, every member on its contig,
    two runs bit-identical."""
    n, L = 120_000_000, 150
    st = {}
    res, d1, d2 = _run_checked(n, L, 1002, stats=st)
    assert d1 == d2
    assert res["n_reads"] == n and res["members"] + res["n_sg"] + sum(res["n_" + k] for k in ("allA", "allT", "allN", "fpA", "fpT", "fpN", "Nfile")) == n
    assert res["members"] > 0.8 * n and res["members_checked"] == res["members"]
    assert res["max_mismatch"] <= L // 2 and res["mean_mismatch"] < 0.02 * L
    assert st["cix_entries"] > 65535 * 14336                              # (more entries than the sorted placement can take in 65 535 partitions)
