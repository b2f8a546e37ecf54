"""GPU parity: HIP read kernels (through the C ABI) against the CPU oracle, bit-exact."""
import gzip
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import minicom_amd
    c = minicom_amd.Context(0)
    yield c
    c.close()


def _golden_reads(golden_dir, tag):
    with gzip.open(os.path.join(golden_dir, tag + ".reads.gz"), "rb") as f:
        rows = f.read().split(b"\n")[:-1]
    return np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), len(rows[0])).copy()


def _pack_nt4(reads):
    """numpy statement of the packed-row format of include/mcom.h (bases must be ACGT)."""
    n, L = reads.shape
    W = (2 * L + 63) // 64
    code = np.zeros_like(reads)
    code[reads == ord("C")] = 1; code[reads == ord("G")] = 2; code[reads == ord("T")] = 3
    out = np.zeros((n, W), dtype=np.uint64)
    for i in range(L):
        out[:, i // 32] |= code[:, i].astype(np.uint64) << np.uint64(2 * (i % 32))
    return out


def _check_process(ctx, reads, L, k, e=4, pitch=None):
    import torch
    import oracle
    from minicom_amd.hip import records_to_numpy
    n = reads.shape[0]
    pitch = pitch or L
    buf = np.full((n, pitch), ord("#"), dtype=np.uint8); buf[:, :L] = reads
    out = ctx.process_reads(torch.from_numpy(buf).cuda(), L, k, e=e, rid0=7, want_nmask=True)
    ctx.sync()
    sub, cls, rec, ncnt = oracle.process_reads_batch(reads, k, e=e, rid0=7)
    assert np.array_equal(out["cls"].cpu().numpy(), cls)
    assert np.array_equal(out["ncnt"].cpu().numpy().astype(np.uint16), ncnt)
    got = records_to_numpy(out["rec"])
    keep = cls == 0
    assert np.array_equal(got["x"][keep], rec["x"][keep])
    assert np.array_equal(got["y"][keep], rec["y"][keep])
    assert np.all(got["x"][~keep] == np.uint64(0xFFFFFFFFFFFFFFFF))
    # packed rows of kept reads = the oracle's N-substituted sequence, 2-bit packed
    want = _pack_nt4(sub[keep])
    assert np.array_equal(out["packed"].cpu().numpy().view(np.uint64)[keep], want)
    # N mask
    nm = out["nmask"].cpu().numpy().view(np.uint64)
    isn = reads == ord("N")
    for w in range(nm.shape[1]):
        seg = isn[:, 64 * w: 64 * (w + 1)]
        val = (seg.astype(np.uint64) << np.arange(seg.shape[1], dtype=np.uint64)[None, :]).sum(axis=1, dtype=np.uint64)
        assert np.array_equal(nm[:, w], val)
    return int(keep.sum())


@pytest.mark.parametrize("tag,k", [("stages_L100", 31), ("stages_L150", 31), ("stages_L100_k24", 24)])
def test_process_reads_on_reference_fixture_reads(ctx, golden_dir, tag, k):
    reads = _golden_reads(golden_dir, tag)
    assert _check_process(ctx, reads, reads.shape[1], k) > 1000


@pytest.mark.parametrize("L,k,pitch", [(100, 31, 100), (150, 31, 152), (150, 31, 150), (37, 17, 37), (64, 16, 64),
                                       (128, 31, 128), (129, 25, 131), (256, 31, 256), (200, 10, 200), (31, 31, 32)])
def test_process_reads_shapes_and_k(ctx, L, k, pitch):
    from minicom_amd import synth
    reads = synth.synth_reads(500 + L + k, 3000, L, plumbing=True)
    assert _check_process(ctx, reads, L, k, pitch=pitch) > 2000


def test_process_reads_ragged_sizes_and_empty(ctx):
    import torch
    from minicom_amd import synth
    for n in (1, 2, 3, 63, 64, 65, 257):
        reads = synth.synth_reads(900 + n, n, 150, plumbing=False)
        _check_process(ctx, reads, 150, 31)
    out = ctx.process_reads(torch.empty((0, 150), dtype=torch.uint8, device="cuda"), 150, 31)
    assert out["rec"].shape[0] == 0


def test_bad_arguments_raise(ctx):
    import torch
    import minicom_amd
    a = torch.zeros((4, 100), dtype=torch.uint8, device="cuda")
    with pytest.raises(minicom_amd.McomError):
        ctx.process_reads(a, 100, 32)
    with pytest.raises(minicom_amd.McomError):
        ctx.process_reads(a, 300, 31)


@pytest.mark.parametrize("k", [31, 30, 29, 24, 17, 16, 13, 10])
def test_resketch_subset_with_every_round_k(ctx, k):
    """Stage-1 rounds re-sketch rejected reads with k-1, k-2, ... (kthread_bucket.c:205)."""
    import torch
    import oracle
    from minicom_amd import synth
    from minicom_amd.hip import records_to_numpy
    L = 150
    reads = synth.synth_reads(77, 5000, L)
    packed = torch.from_numpy(_pack_nt4(reads).view(np.int64)).cuda()
    rng = np.random.default_rng(k)
    rids = np.sort(rng.choice(5000, 1234, replace=False)).astype(np.int32)
    rec = ctx.sketch_reads(packed, L, k, rids=torch.from_numpy(rids).cuda())
    ctx.sync()
    got = records_to_numpy(rec)
    for i, r in enumerate(rids[:400]):
        x, y = oracle.sketch_two(reads[r].tobytes(), k, int(r))
        assert (int(got["x"][i]), int(got["y"][i])) == (x, y)


def test_palindromic_and_homopolymer_reads(ctx):
    import torch
    import oracle
    from minicom_amd.hip import records_to_numpy
    L = 100
    seqs = [("AT" * 64)[:L], ("ACGT" * 32)[:L], "C" * L, ("GC" * 50), ("AATT" * 25), ("ACGTACGTTGCA" * 9)[:L]]
    reads = np.frombuffer("".join(seqs).encode(), dtype=np.uint8).reshape(len(seqs), L)
    packed = torch.from_numpy(_pack_nt4(reads).view(np.int64)).cuda()
    for k in (31, 30, 24, 16, 12, 11):
        got = records_to_numpy(ctx.sketch_reads(packed, L, k)); ctx.sync()
        for i, s in enumerate(seqs):
            assert (int(got["x"][i]), int(got["y"][i])) == oracle.sketch_two(s.encode(), k, i), (k, s[:12])


def test_synth_kernel_equals_numpy_generator(ctx):
    from minicom_amd import synth
    for (seed, n, L, pitch) in ((5, 2000, 150, 152), (6, 1000, 100, 100)):
        a = ctx.synth_reads(seed, n, L, pitch=pitch); ctx.sync()
        assert np.array_equal(a.cpu().numpy()[:, :L], synth.synth_reads(seed, n, L))
    a = ctx.synth_reads(9, 10**6, 150, first=123456, count=500); ctx.sync()
    assert np.array_equal(a.cpu().numpy(), synth.synth_reads(9, 10**6, 150, first=123456, count=500))


def test_full_size_property_sketch_is_permutation_invariant(ctx):
    """Size-independent property at a large batch: the record of a read depends only on its own row."""
    import torch
    from minicom_amd.hip import records_to_numpy
    L, k, n = 150, 31, 1 << 20
    a = ctx.synth_reads(1002, n, L)
    out = ctx.process_reads(a, L, k)
    perm = torch.randperm(n, device="cuda")
    rec2 = ctx.sketch_reads(out["packed"], L, k, rids=perm.to(torch.int32))
    ctx.sync()
    r1 = out["rec"][perm]
    assert torch.equal(r1[:, 0], rec2[:, 0])
    assert torch.equal(r1[:, 1], rec2[:, 1])
    x = records_to_numpy(out["rec"][:1000])
    assert np.all(x["x"] < (1 << 62))


@pytest.mark.gpu
@pytest.mark.parametrize("L", [37, 64, 100, 150, 200, 256])
def test_reads_packed_on_the_host_give_what_the_character_matrix_gives(L):
    """Round 4: the FASTQ parser packs on the host (2 bits per base, an N as code 0 plus a flag) and mcom_process_reads_packed does what
    is left of process_reads (kthread_reads.c:55-224) on the device: classes, N counts, majority-base substitution, records -- all equal
    to mcom_process_reads on the characters; in place as well."""
    import torch
    import minicom_amd
    from minicom_amd import synth
    reads = np.concatenate([synth.synth_reads(77 + L, 20000, L), synth.synth_reads(78 + L, 20000, L, plumbing=True)])
    n = reads.shape[0]
    code = np.zeros(256, dtype=np.uint64); code[ord("C")] = 1; code[ord("G")] = 2; code[ord("T")] = 3
    W, NW = (2 * L + 63) // 64, (L + 63) // 64
    packed = np.zeros((n, W), dtype=np.uint64); nmask = np.zeros((n, NW), dtype=np.uint64)
    c = code[reads]; isn = (reads == ord("N")).astype(np.uint64)
    for i in range(L):
        packed[:, i // 32] |= c[:, i] << np.uint64(2 * (i % 32))
        nmask[:, i // 64] |= isn[:, i] << np.uint64(i % 64)
    ctx = minicom_amd.Context(0)
    k = 31 if L >= 80 else 17
    want = ctx.process_reads(torch.from_numpy(reads).cuda(), L, k, want_nmask=True)
    dp, dn = torch.from_numpy(packed.view(np.int64)).cuda(), torch.from_numpy(nmask.view(np.int64)).cuda()
    for in_place in (False, True):
        got = ctx.process_reads_packed(dp.clone(), dn.clone(), L, k, in_place=in_place)
        ctx.sync()
        for name in ("packed", "cls", "ncnt", "nmask"):
            assert torch.equal(got[name], want[name]), (name, in_place)
        assert torch.equal(got["rec"], want["rec"]), in_place


@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 1000, 100003])
def test_special_reads_lists_every_read_of_another_class(ctx, n):
    """mcom_special_reads: (rid << 8 | class) of every read whose class is not 0, whatever n modulo the eight class bytes a thread takes;
    a list that is too short still reports the full count."""
    import torch
    rng = np.random.default_rng(n + 3)
    cls = np.zeros(n, dtype=np.uint8)
    if n:
        hot = rng.random(n) < (0.3 if n < 100 else 0.01)
        cls[hot] = rng.integers(1, 8, int(hot.sum()))
        cls[-1] = 5
    d = torch.from_numpy(cls).cuda() if n else torch.zeros(0, dtype=torch.uint8, device="cuda")
    lst, count = ctx.special_reads(d)
    want = np.sort((np.flatnonzero(cls).astype(np.int64) << 8) | cls[cls != 0].astype(np.int64))
    assert count == len(want)
    assert np.array_equal(np.sort(lst.cpu().numpy()), want)
    if count > 2:
        short, c2 = ctx.special_reads(d, cap=2)
        assert c2 == count and len(short) == 2 and set(short.cpu().numpy().tolist()) <= set(want.tolist())
