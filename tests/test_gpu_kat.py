"""GPU parity on the reference's own known-answer vectors (tests/golden/kat.json.gz, printed by the compiled reference):
hash64 (sketch.c:27-37), mm_sketch_two (sketch.c:238-289) and encode_byte (kthread_hash_realign.c:283-314) through the C ABI.
The LH / RS / MP vectors are in test_gpu_contigs.py and test_gpu_sort.py."""
import collections
import gzip
import json
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


@pytest.fixture(scope="module")
def kat(golden_dir):
    with gzip.open(os.path.join(golden_dir, "kat.json.gz"), "rt") as f:
        return json.load(f)


def _rows(seqs):
    from minicom_amd.hip import pack_nt4
    a = np.frombuffer("".join(seqs).encode(), dtype=np.uint8).reshape(len(seqs), len(seqs[0]))
    return pack_nt4(a)


def test_hash64_reference_vectors(ctx, kat):
    import torch
    by_k = collections.defaultdict(list)
    for t in kat["H64"]:
        by_k[t["k"]].append(t)
    assert {16, 31} <= set(by_k)                    # the 32-bit and the 64-bit form (other k: through the S2 vectors)
    for k, ts in by_k.items():
        kmers = torch.from_numpy(np.array([t["kmer"] for t in ts], dtype=np.uint64).view(np.int64)).cuda()
        got = ctx.hash64(kmers, k).cpu().numpy().view(np.uint64)
        assert got.tolist() == [t["hash"] for t in ts], k


def test_sketch_two_reference_vectors(ctx, kat):
    """All 1 770 vectors: (AT)n / (GC)n and other palindrome-rich reads, even and odd k from 10 to 31, L = 37 ... 256, reads
    without any minimizer."""
    import torch
    groups = collections.defaultdict(list)
    for t in kat["S2"]:
        groups[(len(t["seq"]), t["k"])].append(t)
    assert len(groups) >= 30 and {37, 64, 100, 150, 256} <= {L for L, _ in groups}
    none = 0
    for (L, k), ts in sorted(groups.items()):
        packed = torch.from_numpy(_rows([t["seq"] for t in ts]).view(np.int64)).cuda()
        rec = ctx.sketch_reads(packed, L, k, rid0=0).cpu().numpy().view(np.uint64)
        for i, t in enumerate(ts):
            x, y = int(rec[i, 0]), int(rec[i, 1])
            if t["x"] == 2**64 - 1:                                     # no k-mer differs from its reverse complement
                assert (x, y) == (t["x"], t["y"]), (L, k, t["seq"])
                none += 1
            else:
                assert x == t["x"] and (y & 0xFFFFFFFF) == (t["y"] & 0xFFFFFFFF) and (y >> 32) == i, (L, k, t["seq"])
    assert none >= 20


def test_encode_byte_reference_vectors(ctx, kat):
    import torch
    seen = set()
    for L in (100, 150):
        ts = [t for t in kat["EB"] if t["L"] == L]
        assert len(ts) >= 50
        cg = ctx.upload_contigs([t["ref"].encode() for t in ts])
        rows = torch.from_numpy(_rows([t["seq"] for t in ts]).view(np.int64)).cuda()
        contig = torch.arange(len(ts), dtype=torch.int32, device="cuda")
        pos = torch.tensor([t["pos"] for t in ts], dtype=torch.int32, device="cuda")
        dirs = torch.tensor([t["dir"] for t in ts], dtype=torch.uint8, device="cuda")
        got = ctx.encode_byte(rows, cg, contig, pos, dirs, L).cpu().numpy().tolist()
        assert got == [t["ok"] for t in ts], L
        seen |= set(got)
    assert seen == {0, 1}
