"""GPU parity: contig kernels (mcom_sketch_contigs, mcom_pack_contigs, mcom_idx_*, mcom_match_pro,
mcom_find_next_candidates) against the reference's golden vectors and the oracle."""
import gzip
import json
import os

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


@pytest.fixture(scope="module")
def kat(golden_dir):
    with gzip.open(os.path.join(golden_dir, "kat.json.gz"), "rt") as f:
        return json.load(f)


def _golden_reads(golden_dir, tag):
    with gzip.open(os.path.join(golden_dir, tag + ".reads.gz"), "rb") as f:
        rows = f.read().split(b"\n")[:-1]
    return np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), len(rows[0])).copy()


def _recs(t):
    from minicom_amd.hip import records_to_numpy
    return records_to_numpy(t)


@pytest.mark.parametrize("kernel", ["default", "lane"])
def test_sketch_contigs_equals_reference_vectors(ctx, kat, kernel):
    """mm_sketch_lh_ori golden outputs (including N runs, palindromes, w = 1), through the default choice of kernel (a wave per string
    for so few strings) and through the lane-per-string kernel that sketches the millions."""
    import torch
    ctx.set_sketch_kernel(4 if kernel == "lane" else 0)
    try:
        _sketch_reference_vectors(ctx, kat, torch)
    finally:
        ctx.set_sketch_kernel(0)


def _sketch_reference_vectors(ctx, kat, torch):
    by = {}
    for t in kat["LH"]:
        by.setdefault((t["w"], t["k"]), []).append(t)
    assert len(by) >= 6
    for (w, k), items in by.items():
        refs = [t["seq"].encode() for t in items]
        cg = ctx.upload_contigs(refs)
        ids = torch.tensor([t["rid"] for t in items], dtype=torch.int32, device="cuda")
        moff, out = ctx.sketch_contigs(cg["seq"], cg["off"], len(refs), w, k, ids=ids)
        ctx.sync()
        moff = moff.cpu().numpy(); r = _recs(out)
        for i, t in enumerate(items):
            seg = r[moff[i]:moff[i + 1]]
            flat = np.stack([seg["x"], seg["y"]], axis=1).reshape(-1).tolist()
            assert flat == t["out"], (w, k, i)
        # first-m prefix, as the index builders use it (kthread_bucket.c:463)
        moff6, out6 = ctx.sketch_contigs(cg["seq"], cg["off"], len(refs), w, k, max_per_contig=6, ids=ids)
        ctx.sync()
        moff6 = moff6.cpu().numpy(); r6 = _recs(out6)
        for i, t in enumerate(items):
            seg = r6[moff6[i]:moff6[i + 1]]
            assert np.stack([seg["x"], seg["y"]], axis=1).reshape(-1).tolist() == t["out"][:12]


def test_pack_contigs_layout(ctx):
    from minicom_amd import synth
    from minicom_amd.hip import pack_contigs
    refs = [synth.synth_reads(70 + i, 1, ln)[0].tobytes() for i, ln in enumerate((150, 151, 31, 32, 33, 64, 1000, 97, 160))]
    cg = ctx.upload_contigs(refs)
    ctx.sync()
    cbits, coff, clen = pack_contigs(refs)
    assert np.array_equal(cg["coff"].cpu().numpy().view(np.uint64), coff)
    assert np.array_equal(cg["cbits"].cpu().numpy().view(np.uint64)[: len(cbits)], cbits)


@pytest.mark.parametrize("lens_kind,first,start", [("reads", 0, 0), ("reads", 700, 13), ("tiny", 3, 5), ("mixed", 1000, 4099), ("long", 2, 1)])
def test_unpack_contigs_is_the_inverse_of_pack_contigs(ctx, lens_kind, first, start):
    """mcom_unpack_contigs (several GPUs send a new contig once, as packed words, and every rank makes the strings it did not build): the
    strings of contigs [first, n) of a set come back from the packed words of mcom_pack_contigs byte for byte, into a buffer whose other
    bytes are left alone -- contigs of a read's length and more, of a few characters (more of them under one block than its staged
    offsets hold), some of no characters at all, and of tens of thousands of characters; strings that start at any byte."""
    import torch
    rng = np.random.default_rng(len(lens_kind) * 1000 + first)
    if lens_kind == "reads": lens = rng.integers(100, 900, 3000)
    elif lens_kind == "tiny": lens = rng.integers(0, 9, 5000)
    elif lens_kind == "mixed": lens = np.concatenate([rng.integers(0, 12, 1500), rng.integers(150, 3000, 400), rng.integers(1, 5, 900)])
    else: lens = np.array([70_000, 3, 0, 41_234, 150, 20_001])
    if lens_kind == "mixed": rng.shuffle(lens)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    refs = [acgt[rng.integers(0, 4, int(n))].tobytes() for n in lens]
    n = len(refs)
    # the packed words, restated here (include/mcom.h, mcom_pack_contigs: base i of a contig in bits 2 i of its words, A C G T = 0 1 2 3,
    # one padding word behind every contig) -- mcom_pack_contigs itself wants contigs of at least one character
    code = np.zeros(256, dtype=np.uint64); code[[65, 67, 71, 84]] = [0, 1, 2, 3]
    words = (2 * np.asarray(lens, dtype=np.int64) + 63) // 64 + 1
    coff_h = np.concatenate([[0], np.cumsum(words)])
    cb = np.zeros(int(coff_h[-1]) + 1, dtype=np.uint64)
    for c, r in enumerate(refs):
        v = code[np.frombuffer(r, dtype=np.uint8)]
        for w in range(0, len(v), 32):
            part = v[w:w + 32]
            cb[coff_h[c] + w // 32] = np.bitwise_or.reduce(part << (2 * np.arange(len(part), dtype=np.uint64))) if len(part) else 0
    if min(lens) > 0:
        up = ctx.upload_contigs(refs)
        assert np.array_equal(up["cbits"].cpu().numpy().view(np.uint64)[:int(coff_h[-1])], cb[:-1])
    lens_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    set_ = {"off": torch.from_numpy(lens_off).cuda(), "coff": torch.from_numpy(coff_h[:-1].astype(np.int64)).cuda(), "cbits": torch.from_numpy(cb.view(np.int64)).cuda()}
    # the strings of contigs [first, n) go to out[start + (their offsets - the first one's)]; everything else keeps its 0xEE
    off = set_["off"].cpu().numpy()
    rel = off[first:] - off[first] + start
    total = int(rel[-1])
    out = torch.full((total + 64,), 0xEE, dtype=torch.uint8, device="cuda")
    d_rel = torch.from_numpy(rel.astype(np.int64)).cuda()
    coff = set_["coff"][first:].contiguous()
    ctx.unpack_contigs(set_["cbits"], coff, d_rel, n - first, start, total, out)
    ctx.sync()
    got = out.cpu().numpy()
    want = np.full(total + 64, 0xEE, dtype=np.uint8)
    want[start:total] = np.frombuffer(b"".join(refs[first:]), dtype=np.uint8)
    assert np.array_equal(got, want), int((got != want).sum())


def test_match_pro_equals_reference_vectors(ctx, kat):
    import torch
    refs, a, pa, b, pb, want = [], [], [], [], [], []
    for t in kat["MP"]:
        a.append(len(refs)); refs.append(t["s0"].encode())
        b.append(len(refs)); refs.append(t["s1"].encode())
        pa.append(t["i"]); pb.append(t["j"]); want.append(t["d"])
    cg = ctx.upload_contigs(refs)
    f = lambda v: torch.tensor(v, dtype=torch.int32, device="cuda")
    got = ctx.match_pro(cg, f(a), f(pa), f(b), f(pb))
    ctx.sync()
    assert got.cpu().numpy().tolist() == want


def test_idx_build_and_get(ctx):
    import torch
    rng = np.random.default_rng(3)
    n = 20000
    keys = rng.integers(0, 1 << 62, 6000, dtype=np.uint64)
    rec = np.zeros(n, dtype=[("x", "<u8"), ("y", "<u8")])
    rec["x"] = keys[rng.integers(0, len(keys), n)]
    rec["y"] = np.arange(n, dtype=np.uint64) << np.uint64(32) | rng.integers(0, 400, n).astype(np.uint64)
    t = torch.from_numpy(np.stack([rec["x"], rec["y"]], axis=1).view(np.int64)).cuda()
    idx = ctx.idx_build(t, 31, b=0)
    srt = _recs(idx.records())
    order = np.argsort(rec["x"], kind="stable")
    assert np.array_equal(srt["x"], rec["x"][order]) and np.array_equal(srt["y"], rec["y"][order])
    probe = np.concatenate([keys, keys ^ np.uint64(1 << 40), np.array([MAXU], dtype=np.uint64)])
    s, c = idx.get(torch.from_numpy(probe.view(np.int64)).cuda())
    ctx.sync()
    s, c = s.cpu().numpy(), c.cpu().numpy()
    sx = srt["x"]
    for q, kq in enumerate(probe):
        lo, hi = np.searchsorted(sx, kq, "left"), np.searchsorted(sx, kq, "right")
        if kq == MAXU:
            assert c[q] == 0
        else:
            assert c[q] == hi - lo and (hi == lo or s[q] == lo)
    idx.close()
    e = ctx.idx_build(ctx.empty_records(0), 31, b=14)
    s, c = e.get(torch.zeros(3, dtype=torch.int64, device="cuda"))
    assert c.sum().item() == 0
    e.close()


@pytest.mark.parametrize("tag", ["stages_L100", "stages_L150"])
def test_find_next_candidates_on_reference_fixture_contigs(ctx, golden_dir, tag):
    """Contigs after the reference's kt_for_bucket stage: index of first-6 minimizers, all-minimizer queries,
    passing candidates in find_next's visiting order."""
    import torch
    import oracle
    reads = _golden_reads(golden_dir, tag)
    L = reads.shape[1]
    p = oracle.Pipeline(reads)
    p.stage_reads(); p.stage_bucket()
    contigs = p.contigs()
    refs = [r for r, _ in contigs]
    k, rw, cbthr, m = p.counter("k"), p.counter("rw"), 8, 6
    cg = ctx.upload_contigs(refs)
    moff6, first6 = ctx.sketch_contigs(cg["seq"], cg["off"], len(refs), rw, k, max_per_contig=m)
    idx = ctx.idx_build(first6, k)
    moff, allq = ctx.sketch_contigs(cg["seq"], cg["off"], len(refs), rw, k)
    pairs, n_tested = ctx.find_next_candidates(idx, allq, cg, cbthr)
    ctx.sync()
    got = _recs(pairs)
    # expectation with the oracle's functions
    table = {}
    for i, r in enumerate(refs):
        mz = oracle.sketch_lh_ori(r, rw, k, i)[:m]
        for x, y in zip(mz["x"].tolist(), mz["y"].tolist()):
            table.setdefault(x, []).append(y)
    want = []
    tested = 0
    for i, r in enumerate(refs):
        for x, y in zip(*[v.tolist() for v in (lambda a: (a["x"], a["y"]))(oracle.sketch_lh_ori(r, rw, k, i))]):
            for hy in table.get(x, []):
                tested += 1
                rid = hy >> 32
                if rid == i or (hy & 1) != (y & 1):
                    continue
                if oracle.match_pro(r, refs[rid], (y & 0xFFFFFFFF) >> 1, (hy & 0xFFFFFFFF) >> 1) <= cbthr:
                    want.append((y, hy))
    assert n_tested == tested
    assert list(zip(got["x"].tolist(), got["y"].tolist())) == want
    assert len(want) > 10
    idx.close()


@pytest.mark.parametrize("tag", ["stages_L100", "stages_L150"])
def test_find_next_candidates_over_a_list_of_a_store(ctx, golden_dir, tag):
    """The merge rounds leave the contig set where it is (round 5): the contigs of a round are a LIST of indices into a store, visited in
    list order (cp_cluster's order, kthread_cb.c:397-434, without its copies), ids are store indices.  Same fixture contigs as above, visited
    through a shuffled list that leaves some contigs of the store out: index from the first six minimizers in list order, queries in list
    order, candidates as find_next would list them (:267-291); then the same with the upper part of the store marked "made by the round
    before": pairs of two older contigs are not evaluated again."""
    import torch
    import oracle
    reads = _golden_reads(golden_dir, tag)
    p = oracle.Pipeline(reads)
    p.stage_reads(); p.stage_bucket()
    refs = [r for r, _ in p.contigs()]
    k, rw, cbthr, m = p.counter("k"), p.counter("rw"), 8, 6
    n = len(refs)
    rng = np.random.default_rng(len(refs))
    lst = rng.permutation(n)[: n - n // 7].astype(np.int32)                      # a seventh of the store is dead (merged away in earlier rounds)
    cg = ctx.upload_contigs(refs)
    roff, rec = ctx.sketch_contigs(cg["seq"], cg["off"], n, rw, k)              # the store's records: ids = store indices
    d_lst = torch.from_numpy(lst).cuda()
    _, first6 = ctx.minimizer_prefix_ord(roff, rec, d_lst, m)
    idx = ctx.idx_build(first6, k)
    table = {}
    sk = {int(i): oracle.sketch_lh_ori(refs[i], rw, k, int(i)) for i in lst}
    for i in lst:
        mz = sk[int(i)][:m]
        for x, y in zip(mz["x"].tolist(), mz["y"].tolist()):
            table.setdefault(x, []).append(y)
    first_new = n - n // 4

    def expect(new_from):
        want = []
        for i in lst:
            r = refs[i]
            for x, y in zip(sk[int(i)]["x"].tolist(), sk[int(i)]["y"].tolist()):
                for hy in table.get(x, []):
                    rid = hy >> 32
                    if rid == i or (hy & 1) != (y & 1) or (new_from and i < new_from and rid < new_from):
                        continue
                    if oracle.match_pro(r, refs[rid], (y & 0xFFFFFFFF) >> 1, (hy & 0xFFFFFFFF) >> 1) <= cbthr:
                        want.append((y, hy))
        return want
    for new_from, n_new in ((0, 0), (first_new, n - first_new)):
        pairs, _ = ctx.find_next_candidates_ord(idx, rec, roff, d_lst, cg, cbthr, new_from, n_new)
        ctx.sync()
        got = _recs(pairs)
        want = expect(new_from)
        assert list(zip(got["x"].tolist(), got["y"].tolist())) == want and len(want) > 5
    # no list: the store's own order = the plain entry point
    _, first6 = ctx.minimizer_prefix_ord(roff, rec, None, m)
    idx2 = ctx.idx_build(first6, k)
    a, na = ctx.find_next_candidates_ord(idx2, rec, roff, None, cg, cbthr)
    b, nb = ctx.find_next_candidates(idx2, rec, cg, cbthr)
    ctx.sync()
    assert na == nb and torch.equal(a, b)
    idx.close(); idx2.close()


def test_radix_sort_ref_order_reproduces_the_reference_permutation(ctx, kat):
    """Every golden radix_sort_128x vector, including the sizes where the reference sort is unstable."""
    import torch
    for t in kat["RS"]:
        a = np.array([tuple(p) for p in t["in"]], dtype=[("x", "<u8"), ("y", "<u8")])
        d = torch.from_numpy(np.stack([a["x"], a["y"]], axis=1).view(np.int64)).cuda()
        got = _recs(ctx.radix_sort_128x_ref_order(d))
        assert [[int(p["x"]), int(p["y"])] for p in got] == t["out"]


@pytest.mark.parametrize("n,nkeys,nbuckets", [(40000, 300, 7), (200000, 5000, 64), (30000, 30000, 16384), (150000, 400, 2)])
def test_idx_build_keeps_the_reference_order_inside_big_buckets(ctx, n, nkeys, nbuckets):
    """Index buckets far above 64 entries with many equal minimizers: mm_idx's order (kthread_idx.c:126,154) exactly."""
    import torch
    import oracle
    rng = np.random.default_rng(n)
    bucket_ids = rng.choice(16384, nbuckets, replace=False).astype(np.uint64)
    keys = (rng.integers(0, 1 << 48, nkeys, dtype=np.uint64) << np.uint64(14)) | bucket_ids[rng.integers(0, nbuckets, nkeys)]
    rec = np.zeros(n, dtype=[("x", "<u8"), ("y", "<u8")])
    rec["x"] = keys[rng.integers(0, nkeys, n)]
    rec["y"] = (np.arange(n, dtype=np.uint64) << np.uint64(32)) | rng.integers(0, 600, n).astype(np.uint64)
    d = torch.from_numpy(np.stack([rec["x"], rec["y"]], axis=1).view(np.int64)).cuda()
    idx = ctx.idx_build(d, 31, b=14)
    got = _recs(idx.records())
    b = (rec["x"] & np.uint64(0x3FFF)).astype(np.int64)
    order = np.argsort(b, kind="stable")
    srt, sb = rec[order], b[order]
    starts = np.flatnonzero(np.r_[True, sb[1:] != sb[:-1]]); ends = np.r_[starts[1:], n]
    want = np.concatenate([oracle.radix_sort_128x(srt[s:e]) for s, e in zip(starts, ends)])
    assert max(ends - starts) > 64 or nbuckets == 16384
    assert np.array_equal(got["x"], want["x"]) and np.array_equal(got["y"], want["y"])
    s, c = idx.get(torch.from_numpy(keys[:50].view(np.int64).copy()).cuda())
    ctx.sync()
    for q in range(50):
        lo = int(s[q]); cnt = int(c[q])
        assert cnt == int((rec["x"] == keys[q]).sum()) and np.all(got["x"][lo:lo + cnt] == keys[q])
    idx.close()


@pytest.mark.parametrize("kernel", ["default", "lane", "wave", "ring64", "prefix3", "prefix9", "prefix30", "ring32prefix5"])
@pytest.mark.parametrize("w,k", [(44, 31), (19, 31), (3, 17), (10, 16), (128, 21), (5, 8), (1, 9), (7, 31), (4, 24), (64, 31), (20, 13), (33, 25)])
def test_sketch_contigs_with_repeats_and_ties_equals_oracle(ctx, w, k, kernel):
    """Tie-rich strings (short-period repeats, homopolymers, copied blocks, ambiguous bases): equal hashes inside a
    window exercise the duplicate rules of sketch.c:138-161 and the exact fallback of the prefix window scan.
    All kernels: one lane per string with the ring of 32-bit hash prefixes (k odd: the default; a tie between prefixes is settled by
    recomputing the hashes from the string -- with the prefix narrowed to 3 or 9 bits that is most comparisons) or of 64-bit
    hashes (windows up to 64; homopolymers overflow their room and are scanned again), and one wave per string."""
    import oracle
    if kernel != "default" and w == 128:
        pytest.skip("w = 128 takes the wave-per-string kernel anyway")
    if "prefix" in kernel and k % 2 == 0:
        pytest.skip("the prefix ring is for odd k")
    ctx.set_sketch_kernel({"default": 0, "wave": 1, "ring64": 2, "ring32prefix5": 3}.get(kernel, 4))        # 3: 32-bit ring words whatever the prefix width; 4: a lane per string for these 160 strings too (0 gives them a wave each)
    ctx.set_sketch_prefix_bits(5 if kernel == "ring32prefix5" else int(kernel[6:]) if kernel.startswith("prefix") else 14)
    if kernel == "lane" and k % 2 == 0:
        pytest.skip("the same kernel as ring64 for even k")
    from test_gpu_resketch import _string
    rng = np.random.default_rng(100 * w + k)
    refs = []
    for i in range(160):
        n = int(rng.integers(1, 1500))
        s = _string(rng, n, int(rng.integers(0, 3)))
        if i % 7 == 0:
            s[:] = s[0]                                               # a homopolymer
        if i % 5 == 0 and n > 10:
            s[rng.integers(0, n, 3)] = ord("N")
        if i % 11 == 0 and n > 200:                                   # two copies of a block at a distance below w + k
            s[100:140] = s[40:80]
        refs.append(s.tobytes())
    cg = ctx.upload_contigs(refs)
    try:
        moff, out = ctx.sketch_contigs(cg["seq"], cg["off"], len(refs), w, k)
        ctx.sync()
    finally:
        ctx.set_sketch_kernel(0); ctx.set_sketch_prefix_bits(14)
    moff = moff.cpu().numpy(); r = _recs(out)
    total = 0
    for i, ref in enumerate(refs):
        want = oracle.sketch_lh_ori(ref, w, k, i)
        seg = r[moff[i]:moff[i + 1]]
        assert np.array_equal(seg["x"], want["x"]) and np.array_equal(seg["y"], want["y"]), (w, k, i, len(ref))
        total += len(want)
    assert total > 500
