"""GPU parity: record sort + grouping (mcom_radix_sort_128x, mcom_sort_group) against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
MAXU = np.uint64(0xFFFFFFFFFFFFFFFF)


@pytest.fixture(scope="module")
def ctx():
    import minicom_amd
    c = minicom_amd.Context(0)
    yield c
    c.close()


def _to_dev(a):
    import torch
    t = np.stack([a["x"], a["y"]], axis=1).view(np.int64)
    return torch.from_numpy(np.ascontiguousarray(t)).cuda()


def _from_dev(t):
    from minicom_amd.hip import records_to_numpy
    return records_to_numpy(t)


def _expected_groups(rec, L, k_orig, b=14):
    """process_bucket front half at one thread (kthread_bucket.c:381-446) with the oracle's radix sort."""
    import oracle
    mask = np.uint64((1 << b) - 1)
    valid = rec[rec["x"] != MAXU]
    bucket = (valid["x"] & mask).astype(np.int64)
    order = np.argsort(bucket, kind="stable")
    valid, bucket = valid[order], bucket[order]
    singles, members, goff = [], [], [0]
    starts = np.flatnonzero(np.r_[True, bucket[1:] != bucket[:-1]])
    ends = np.r_[starts[1:], len(valid)]
    for s, e in zip(starts, ends):
        srt = oracle.radix_sort_128x(valid[s:e])
        xs = srt["x"]
        gs = np.flatnonzero(np.r_[True, xs[1:] != xs[:-1]])
        ge = np.r_[gs[1:], len(xs)]
        for a, z in zip(gs, ge):
            if z - a == 1:
                singles.append(int(srt["y"][a] >> np.uint64(32)))
            else:
                ys = [int(v) for v in srt["y"][a:z]]
                def key(y):
                    pos = (y & 0xFFFFFFFF) >> 1
                    if y & 1:
                        pos = L - pos + k_orig - 2
                    return (-pos, y >> 32)
                ys.sort(key=key)
                members += ys
                goff.append(len(members))
    return np.array(singles, dtype=np.uint32), np.array(members, dtype=np.uint64), np.array(goff, dtype=np.uint32)


def _check_sort_group(ctx, rec, L, k_orig, kmer, b=14):
    out = ctx.sort_group(_to_dev(rec), L, k_orig, kmer, b=b)
    s, m, g = _expected_groups(rec, L, k_orig, b)
    assert out["n_valid"] == int((rec["x"] != MAXU).sum())
    assert np.array_equal(out["singles"].cpu().numpy().view(np.uint32), s)
    assert np.array_equal(out["members"].cpu().numpy().view(np.uint64), m)
    assert np.array_equal(out["group_off"].cpu().numpy().view(np.uint32), g)
    srt = _from_dev(out["sorted"])
    nv = out["n_valid"]
    bucket = srt["x"][:nv] & np.uint64((1 << b) - 1)
    assert np.all(np.diff(bucket.astype(np.int64)) >= 0)
    assert np.all(srt["x"][nv:] == MAXU)
    return len(s), len(g) - 1


@pytest.mark.parametrize("L,k,n", [(100, 31, 20000), (150, 31, 30000), (150, 24, 9000), (100, 17, 5000), (64, 12, 5000)])
def test_sort_group_on_sketched_reads(ctx, L, k, n):
    import oracle
    from minicom_amd import synth
    reads = synth.synth_reads(2000 + L + k, n, L, plumbing=True)
    _, cls, rec, _ = oracle.process_reads_batch(reads, k)
    rec = rec.copy()
    rec["x"][cls != 0] = MAXU; rec["y"][cls != 0] = MAXU
    ns, ng = _check_sort_group(ctx, rec, L, k, k)
    assert ns > 0 and ng > 100


def test_sort_group_later_round_uses_original_k_for_alignment(ctx):
    """cmpcluster keeps reads->k while the records were sketched with k - r (kthread_bucket.c:52)."""
    import oracle
    from minicom_amd import synth
    L, k0, kr = 100, 31, 27
    reads = synth.synth_reads(4242, 8000, L)
    rec = oracle.sketch_two_batch(reads, kr)
    _check_sort_group(ctx, rec, L, k0, kr)


def test_sort_group_heavy_duplicates_and_big_groups(ctx):
    """Many equal hashes: groups far above the 64-element threshold where the reference sort turns unstable."""
    rng = np.random.default_rng(5)
    n = 50000
    keys = rng.integers(0, 1 << 62, 40, dtype=np.uint64)
    rec = np.zeros(n, dtype=[("x", "<u8"), ("y", "<u8")])
    rec["x"] = keys[rng.integers(0, 40, n)]
    rec["x"][::97] = rng.integers(0, 1 << 62, len(rec["x"][::97]), dtype=np.uint64)      # sprinkled singles
    rid = np.arange(n, dtype=np.uint64)
    pos = rng.integers(30, 100, n).astype(np.uint64)
    rec["y"] = (rid << np.uint64(32)) | (pos << np.uint64(1)) | rng.integers(0, 2, n).astype(np.uint64)
    rec["x"][5] = MAXU; rec["y"][5] = MAXU
    _check_sort_group(ctx, rec, 100, 31, 31)


def test_sort_group_segments_beyond_the_lds_sort(ctx):
    """One minimizer shared by 20 000 reads cannot be split by any key prefix: such segments leave the in-LDS segment sort
    for the nine-pass sort (sort.hip).  Three of them, beside ordinary records; then the same with the capacity lowered so
    that nearly every segment takes that route."""
    rng = np.random.default_rng(11)
    n = 90000
    big = rng.integers(0, 1 << 62, 3, dtype=np.uint64)
    rec = np.zeros(n, dtype=[("x", "<u8"), ("y", "<u8")])
    rec["x"] = rng.integers(0, 1 << 62, n, dtype=np.uint64)
    rec["x"][:60000] = big[rng.integers(0, 3, 60000)]
    rec["x"][60000:70000] = rec["x"][70000:80000]                                        # pairs
    perm = rng.permutation(n)
    rec["x"] = rec["x"][perm]
    pos = rng.integers(30, 100, n).astype(np.uint64)
    rec["y"] = (np.arange(n, dtype=np.uint64) << np.uint64(32)) | (pos << np.uint64(1)) | rng.integers(0, 2, n).astype(np.uint64)
    rec["x"][17] = MAXU; rec["y"][17] = MAXU
    before = ctx.counter("sort_overflow_segments")
    _check_sort_group(ctx, rec, 100, 31, 31)
    assert ctx.counter("sort_overflow_segments") - before == 3
    ctx.set_segment_capacity(2)
    try:
        before = ctx.counter("sort_overflow_segments")
        _check_sort_group(ctx, rec, 100, 31, 31)
        assert ctx.counter("sort_overflow_segments") - before > 1000
    finally:
        ctx.set_segment_capacity(0)


def test_sort_group_segments_between_the_two_lds_capacities(ctx):
    """The segment sort comes in two sizes (2048 records in half the LDS, 4096): a segment between the two leaves the small form and
    is sorted by the large one from a list; only what exceeds that goes on to the nine-pass sort (sort.hip).  Five minimizers with
    3 000 reads each, two with 20 000, ordinary records around them."""
    rng = np.random.default_rng(12)
    n = 90000
    keys = rng.integers(0, 1 << 62, 7, dtype=np.uint64)
    rec = np.zeros(n, dtype=[("x", "<u8"), ("y", "<u8")])
    rec["x"] = rng.integers(0, 1 << 62, n, dtype=np.uint64)
    at = 0
    for q, cnt in enumerate([3000, 3000, 3000, 3000, 3000, 20000, 20000]):
        rec["x"][at:at + cnt] = keys[q]; at += cnt
    perm = rng.permutation(n)
    rec["x"] = rec["x"][perm]
    pos = rng.integers(30, 100, n).astype(np.uint64)
    rec["y"] = (np.arange(n, dtype=np.uint64) << np.uint64(32)) | (pos << np.uint64(1)) | rng.integers(0, 2, n).astype(np.uint64)
    before = ctx.counter("sort_overflow_segments")
    _check_sort_group(ctx, rec, 100, 31, 31)
    assert ctx.counter("sort_overflow_segments") - before == 2


def test_sort_group_two_million_records_use_wider_msd_keys(ctx):
    """Above 2048 records per bucket the MSD key takes hash bits too (three global passes instead of two at 2^25 records)."""
    import oracle
    from minicom_amd import synth
    n, L, k = 2_400_000, 100, 31
    reads = synth.synth_reads(77, n, L)
    rec = oracle.sketch_two_batch(reads, k)
    _check_sort_group(ctx, rec, L, k, k, b=6)                                           # few buckets: 37 500 records each, t = 5


@pytest.mark.parametrize("n", [0, 1, 2, 3, 255, 4096, 4097, 8193])
def test_sort_group_ragged_sizes(ctx, n):
    rng = np.random.default_rng(n)
    rec = np.zeros(n, dtype=[("x", "<u8"), ("y", "<u8")])
    rec["x"] = rng.integers(0, 1 << 62, n, dtype=np.uint64) >> np.uint64(40) << np.uint64(40) | np.uint64(7)
    rec["y"] = (np.arange(n, dtype=np.uint64) << np.uint64(32)) | np.uint64(2 * 50)
    if n == 0:
        out = ctx.sort_group(_to_dev(rec), 100, 31, 31)
        assert out["n_valid"] == 0 and out["n_groups"] == 0
    else:
        _check_sort_group(ctx, rec, 100, 31, 31)


def test_radix_sort_128x_matches_reference_vectors(ctx, golden_dir):
    """Golden vectors of the reference's radix_sort_128x: identical key sequence always, identical records
    wherever the reference itself is stable (<= 64 elements, or distinct keys)."""
    import gzip, json, os
    with gzip.open(os.path.join(golden_dir, "kat.json.gz"), "rt") as f:
        kat = json.load(f)
    for t in kat["RS"]:
        a = np.array([tuple(p) for p in t["in"]], dtype=[("x", "<u8"), ("y", "<u8")])
        want = np.array([tuple(p) for p in t["out"]], dtype=a.dtype)
        got = _from_dev(ctx.radix_sort_128x(_to_dev(a)))
        assert np.array_equal(got["x"], want["x"])
        assert sorted(zip(got["x"].tolist(), got["y"].tolist())) == sorted(zip(want["x"].tolist(), want["y"].tolist()))
        if len(a) <= 64 or len(set(a["x"].tolist())) == len(a):
            assert np.array_equal(got["y"], want["y"])
        # stable: equal keys keep input order
        idx = np.argsort(a["x"], kind="stable")
        assert np.array_equal(got["y"], a["y"][idx])


def test_full_size_sort_properties(ctx):
    """Size-independent properties on 2^22 records: sortedness, multiset preservation, group accounting."""
    import torch
    n, L, k = 1 << 22, 150, 31
    a = ctx.synth_reads(1002, n, L)
    rec = ctx.process_reads(a, L, k)["rec"]
    out = ctx.sort_group(rec, L, k, k)
    s = out["sorted"]
    nv = out["n_valid"]
    assert nv == n
    x = s[:, 0]
    bucket = x & 0x3FFF
    hi = (x >> 14) & ((1 << 48) - 1)
    key = bucket * (1 << 48) + hi
    assert bool((key[1:] >= key[:-1]).all())
    assert int(s[:, 1].sum().item()) == int(rec[:, 1].sum().item())
    assert int(torch.bitwise_xor(s[:, 0], torch.roll(s[:, 0], 1)).ne(0).sum().item()) > 0
    assert out["singles"].numel() + out["members"].numel() == nv
    sizes = out["group_off"][1:] - out["group_off"][:-1]
    assert int(sizes.min().item()) >= 2 and int(sizes.sum().item()) == out["members"].numel()
