"""GPU parity of the device-resident contig set (include/mcom.h: mcom_group_consensus, mcom_groups_to_contigs,
mcom_merge_members, mcom_merge_consensus_jobs, mcom_contigs_carry, mcom_records_carry, mcom_scan_u64, mcom_contig_layout)
against plain numpy restatements of the reference's statements (construct_ref kthread_bucket.c:69-377 and :446-505,
find_next's member merge kthread_cb.c:297-325, construct_ref2 :105-218, cp_cluster :397-434)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8); COMP[[65, 67, 71, 84]] = [84, 71, 67, 65]


@pytest.fixture(scope="module")
def ctx():
    import minicom_amd
    c = minicom_amd.Context(0)
    yield c
    c.close()


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _oriented(read, d):
    return COMP[read][::-1] if d else read


def _majority(cols):
    """counts [4][n] -> base per column, ties to the smaller code A < C < G < T (strict '>' scan, kthread_bucket.c:129-141)."""
    best = np.zeros(cols.shape[1], dtype=np.int64); mx = cols[0].copy()
    for q in (1, 2, 3):
        m = cols[q] > mx
        best[m] = q; mx[m] = cols[q][m]
    return best, mx


def test_scan_u64_and_contig_layout(ctx):
    rng = np.random.default_rng(1)
    for n in (1, 7, 1000, 300000):
        x = rng.integers(0, 1 << 40, n, dtype=np.int64)
        got = ctx.scan_u64(_dev(x)).cpu().numpy()
        assert np.array_equal(got, np.concatenate([[0], np.cumsum(x)[:-1]]))
    lens = rng.integers(0, 5000, 4000)
    soff = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    coff, clen, tw = ctx.contig_layout(_dev(soff))
    words = (2 * lens + 63) // 64 + 1
    assert np.array_equal(coff.cpu().numpy(), np.concatenate([[0], np.cumsum(words)])) and tw == int(words.sum())
    assert np.array_equal(clen.cpu().numpy(), lens)


def _make_groups(rng, n_groups, L, k, e):
    """Minimizer groups as mcom_sort_group hands them over: members share a k-mer; records carry rid<<32 | pos<<1 | strand."""
    from minicom_amd.hip import pack_nt4
    reads, members, goff = [], [], [0]
    for g in range(n_groups):
        size = int(rng.choice([2, 2, 3, 5, 9, 30, 70]))
        src = ACGT[rng.integers(0, 4, 3 * L)]
        anchor = L + int(rng.integers(0, L))                                   # the shared k-mer ends here on the source
        rows = []
        for _ in range(size):
            start = anchor - int(rng.integers(k - 1, L))                      # the k-mer lies inside the read
            r = src[start:start + L].copy()
            nerr = int(rng.choice([0, 0, 0, 1, 2, 3 * e]))
            for q in rng.integers(0, L, nerr):
                if not (anchor - start - k + 1 <= q <= anchor - start):
                    r[q] = ACGT[rng.integers(0, 4)]
            d = int(rng.integers(0, 2))
            pos = anchor - start                                               # last base of the k-mer in the oriented read
            stored = _oriented(r, d)                                           # what the read file holds
            rec_pos = pos if d == 0 else L - pos + k - 2                       # cmpcluster's strand flip, inverted (kthread_bucket.c:51-56)
            rows.append((rec_pos if d == 0 else L - 1 - (pos - k + 1), d, stored))
        # cmpcluster order: aligned position descending, then rid ascending (:44-62)
        al = [(p if d == 0 else L - p + k - 2) for p, d, _ in rows]
        order = sorted(range(size), key=lambda i: (-al[i], i))
        for i in order:
            p, d, stored = rows[i]
            members.append(((len(reads)) << 32) | (p << 1) | d); reads.append(stored)
        goff.append(len(members))
    reads = np.stack(reads)
    return reads, pack_nt4(reads), np.array(members, dtype=np.uint64), np.array(goff, dtype=np.int32)


def _construct_ref(reads, members, L, k, e):
    """construct_ref for one group (kthread_bucket.c:69-377): returns (keep flags, new member words, sv, consensus bytes)."""
    al, ds, rids = [], [], []
    for y in members.tolist():
        rid, pos, d = y >> 32, (y & 0xFFFFFFFF) >> 1, y & 1
        al.append(L - pos + k - 2 if d else pos); ds.append(d); rids.append(rid)
    offs = [al[0] - a for a in al]
    TL = 2 * L
    c1 = np.zeros((4, TL), dtype=np.int64)
    ors = [(_oriented(reads[r], d) >> 1 ^ _oriented(reads[r], d) >> 2) & 3 for r, d in zip(rids, ds)]
    for o, code in zip(offs, ors):
        c1[code, o + np.arange(L)] += 1
    first, mx = _majority(c1)
    ref_len = int(np.argmax(mx == 0)) if (mx == 0).any() else TL
    keep, c2, rend = [], np.zeros((4, TL), dtype=np.int64), 0
    for o, code in zip(offs, ors):
        cols = o + np.arange(L)
        dif = int(((cols >= ref_len) | (first[np.minimum(cols, TL - 1)] != code)).sum())
        kp = dif <= e                                                          # :189
        keep.append(kp)
        if kp:
            c2[code, cols] += 1; rend = max(rend, o + L)
    nk = sum(keep)
    new = [(r << 32) | (o << 1) | d for r, o, d in zip(rids, offs, ds)]
    if not nk:
        return keep, new, 0, b""
    cov = c2.sum(axis=0)[:ref_len] > 0
    sv = int(np.argmax(cov)) if cov.any() else ref_len
    second, _ = _majority(c2)
    return keep, new, sv, ACGT[second[sv:rend]].tobytes()


@pytest.mark.parametrize("L,k,e", [(100, 31, 4), (150, 31, 4), (64, 17, 2), (250, 31, 4), (40, 17, 2), (33, 11, 1)])
def test_group_consensus_and_groups_to_contigs(ctx, L, k, e):
    rng = np.random.default_rng(L + e)
    reads, packed, members, goff = _make_groups(rng, 400, L, k, e)
    d_members = _dev(members.view(np.int64))
    gc = ctx.group_consensus(_dev(packed.view(np.int64)), d_members, _dev(goff), L, k, e)
    ctx.sync()
    got_keep, got_mem = gc["keep"].cpu().numpy(), d_members.cpu().numpy().view(np.uint64)
    nk_all, contigs, rejects = [], [], []
    for g in range(len(goff) - 1):
        a, b = goff[g], goff[g + 1]
        keep, new, sv, ref = _construct_ref(reads, members[a:b], L, k, e)
        assert got_keep[a:b].tolist() == [int(x) for x in keep], g
        assert got_mem[a:b].tolist() == new, g
        nk = sum(keep); nk_all.append(nk)
        assert int(gc["nkept"][g]) == nk and (nk == 0 or (int(gc["sv"][g]) == sv and int(gc["reflen"][g]) == len(ref)))
        if nk:
            assert gc["refs"][g * gc["stride"]: g * gc["stride"] + len(ref)].cpu().numpy().tobytes() == ref
        # process_bucket :446-505: more than one kept -> contig (members re-based to sv); everybody else is a reject
        if nk > 1:
            contigs.append((ref, [y - (sv << 1) for y, kp in zip(new, keep) if kp]))
        if not (nk == b - a and nk > 1):
            rejects += [(y >> 32, g) for y, kp in zip(new, keep) if not kp]
            if nk == 1:
                rejects += [(y >> 32, g) for y, kp in zip(new, keep) if kp]
    assert sum(1 for n in nk_all if n > 1) > 100 and len(rejects) > 20
    out = ctx.groups_to_contigs(d_members, _dev(goff), gc, cap_chars=2 * L * len(goff), cap_members=len(members), cap_contigs=len(goff))
    nc, nch, nmm, nrj = out["counts"]
    assert nc == len(contigs) and nrj == len(rejects)
    soff, moff = out["soff"].cpu().numpy(), out["moff"].cpu().numpy()
    seq, mem = out["seq"].cpu().numpy().tobytes(), out["mem"].cpu().numpy().view(np.uint64)
    for c, (ref, mm) in enumerate(contigs):
        assert seq[soff[c]:soff[c + 1]] == ref and mem[moff[c]:moff[c + 1]].tolist() == mm, c
    assert list(zip(out["rej_rid"].cpu().numpy().tolist(), out["rej_group"].cpu().numpy().tolist())) == rejects


def _random_set(rng, n, L):
    """A contig set with members that lie on their contigs (reads cut from the consensus, a few substitutions)."""
    from minicom_amd.hip import pack_nt4
    refs, mems, reads = [], [], []
    for c in range(n):
        ln = int(rng.integers(L, 5 * L))
        ref = ACGT[rng.integers(0, 4, ln)]
        mm = []
        for off in sorted(rng.integers(0, ln - L + 1, int(rng.integers(2, 12))).tolist() + [0, ln - L]):
            r = ref[off:off + L].copy()
            for q in rng.integers(0, L, int(rng.integers(0, 3))):
                r[q] = ACGT[rng.integers(0, 4)]
            d = int(rng.integers(0, 2))
            mm.append((len(reads) << 32) | (off << 1) | d); reads.append(_oriented(r, d))
        refs.append(ref); mems.append(sorted(mm, key=lambda y: y & 0xFFFFFFFF))
    reads = np.stack(reads)
    # consensus = majority of the members, so that "outside the overlap keeps the parent's character" holds as in the pipeline
    for c in range(n):
        cnt = np.zeros((4, len(refs[c])), dtype=np.int64)
        for y in mems[c]:
            rid, off, d = y >> 32, (y & 0xFFFFFFFF) >> 1, y & 1
            o = _oriented(reads[rid], d); cnt[(o >> 1 ^ o >> 2) & 3, off + np.arange(L)] += 1
        refs[c] = ACGT[_majority(cnt)[0]]
    return refs, mems, reads, pack_nt4(reads)


@pytest.mark.parametrize("L", [100, 150, 250, 40])
def test_merge_round_pieces_against_numpy(ctx, L):
    """One merge round on a random set: member merge, consensus (every column, and overlap only), carry of the rest."""
    import torch
    rng = np.random.default_rng(77 + L)
    n = 300
    refs, mems, reads, packed = _random_set(rng, n, L)
    soff = np.concatenate([[0], np.cumsum([len(r) for r in refs])]).astype(np.int64)
    moff = np.concatenate([[0], np.cumsum([len(m) for m in mems])]).astype(np.int64)
    seq = np.concatenate(refs); mem = np.array([y for m in mems for y in m], dtype=np.uint64)
    # claimed pairs: (ci, cj, pos_ori, pos) with anchors inside both contigs
    perm = rng.permutation(n)
    jobs = []
    for j in range(90):
        ci, cj = int(perm[2 * j]), int(perm[2 * j + 1])
        jobs.append((ci, cj, int(rng.integers(30, len(refs[ci]))), int(rng.integers(30, min(len(refs[cj]), 60)))))
    flag = np.zeros(n, dtype=np.uint8)
    for ci, cj, _, _ in jobs:
        flag[ci] = flag[cj] = 1
    d_seq, d_soff, d_mem, d_moff = _dev(seq), _dev(soff), _dev(mem.view(np.int64)), _dev(moff)
    d_jobs = _dev(np.array(jobs, dtype=np.int32))
    jm, jmoff, jroff, tot = ctx.merge_members(d_mem, d_moff, d_jobs, L, 14)
    # kthread_cb.c:297-325 + the stable cmpcluster2 sort of construct_ref2 (:107)
    want_m, want_len = [], []
    for ci, cj, po, pp in jobs:
        if po >= pp:
            lst = mems[ci] + [y + ((po - pp) << 1) for y in mems[cj]]
        else:
            lst = mems[cj] + [y + ((pp - po) << 1) for y in mems[ci]]
        lst = sorted(lst, key=lambda y: y & 0xFFFFFFFF)                          # python's sort is stable
        want_m.append(lst); want_len.append(((lst[-1] & 0xFFFFFFFF) >> 1) + L)
    assert jm.cpu().numpy().view(np.uint64).tolist() == [y for lst in want_m for y in lst]
    assert np.array_equal(jmoff.cpu().numpy(), np.concatenate([[0], np.cumsum([len(x) for x in want_m])]))
    assert np.array_equal(jroff.cpu().numpy(), np.concatenate([[0], np.cumsum(want_len)])) and tot == (sum(len(x) for x in want_m), sum(want_len), max(want_len))
    # consensus of every merged list
    want_refs = []
    for lst, ln in zip(want_m, want_len):
        cnt = np.zeros((4, ln), dtype=np.int64)
        for y in lst:
            rid, off, d = y >> 32, (y & 0xFFFFFFFF) >> 1, y & 1
            o = _oriented(reads[rid], d); cnt[(o >> 1 ^ o >> 2) & 3, off + np.arange(L)] += 1
        want_refs.append(ACGT[_majority(cnt)[0]].tobytes())
    d_packed = _dev(packed.view(np.int64))
    full = ctx.merge_consensus_jobs(d_packed, jm, jmoff, jroff, tot[1], L)
    part = ctx.merge_consensus_jobs(d_packed, jm, jmoff, jroff, tot[1], L, jobs=d_jobs, seq=d_seq, soff=d_soff)
    assert full.cpu().numpy().tobytes() == b"".join(want_refs)
    assert torch.equal(full, part)                                               # counting only the overlap changes nothing
    # ... and the oracle's construct_ref2 (oracle/mcom_oracle.c, pinned on the reference's stage dumps) says the same, list by list: the numpy
    # restatement above and the kernels are not just each other's mirror
    import oracle
    for lst, want in list(zip(want_m, want_refs))[::3]:
        assert oracle.construct_ref2(reads, lst) == want
    # units deeper than the bit-sliced counters hold go tile by tile through the wave-per-tile kernel: the same consensus
    ctx.set_consensus_capacity(3)
    try:
        assert torch.equal(ctx.merge_consensus_jobs(d_packed, jm, jmoff, jroff, tot[1], L), full)
        assert torch.equal(ctx.merge_consensus_jobs(d_packed, jm, jmoff, jroff, tot[1], L, jobs=d_jobs, seq=d_seq, soff=d_soff), full)
    finally:
        ctx.set_consensus_capacity(0)
    # cp_cluster: merged first, then the untouched ones in their order
    nj = len(jobs); nkeep = int((flag == 0).sum()); nn = nj + nkeep
    seq2 = torch.zeros(len(seq) + 16, dtype=torch.uint8, device="cuda"); seq2[: tot[1]] = full
    mem2 = torch.zeros(len(mem) + 1, dtype=torch.int64, device="cuda"); mem2[: tot[0]] = jm
    soff2 = torch.zeros(nn + 1, dtype=torch.int64, device="cuda"); soff2[: nj + 1] = jroff
    moff2 = torch.zeros(nn + 1, dtype=torch.int64, device="cuda"); moff2[: nj + 1] = jmoff
    keepidx, totals = ctx.contigs_carry(d_seq, d_soff, d_mem, d_moff, _dev(flag), nj, seq2, soff2, mem2, moff2)
    kept = np.flatnonzero(flag == 0)
    assert np.array_equal(keepidx.cpu().numpy(), kept)
    all_refs = want_refs + [refs[i].tobytes() for i in kept]
    all_mems = want_m + [mems[i] for i in kept]
    assert totals == (sum(len(r) for r in all_refs), len(mem))
    s2, m2, so2, mo2 = seq2.cpu().numpy().tobytes(), mem2.cpu().numpy().view(np.uint64), soff2.cpu().numpy(), moff2.cpu().numpy()
    for c in range(nn):
        assert s2[so2[c]:so2[c + 1]] == all_refs[c] and m2[mo2[c]:mo2[c + 1]].tolist() == all_mems[c], c
    # minimizers of the untouched contigs keep their values and take the new contig index
    moff_r, rec = ctx.sketch_contigs(d_seq, d_soff, n, 19, 31)
    ctx.sync()
    base = 1234
    rec2 = ctx.empty_records(base + int(rec.shape[0]) + 8)
    roff2 = torch.zeros(nn + 1, dtype=torch.int32, device="cuda")
    total = ctx.records_carry(rec, moff_r, keepidx, nj, base, rec2, roff2)
    ctx.sync()
    r, ro, r2, ro2 = rec.cpu().numpy().view(np.uint64), moff_r.cpu().numpy(), rec2.cpu().numpy().view(np.uint64), roff2.cpu().numpy()
    at = base
    for u, i in enumerate(kept.tolist()):
        seg = r[ro[i]:ro[i + 1]].copy()
        seg[:, 1] = (np.uint64(nj + u) << np.uint64(32)) | (seg[:, 1] & np.uint64(0xFFFFFFFF))
        assert ro2[nj + u] == at and np.array_equal(r2[at:at + len(seg)], seg), u
        at += len(seg)
    assert ro2[nn] == at == total


def test_members_finalize_equals_the_sort_and_append_of_every_pass(ctx):
    """mcom_members_finalize against a numpy restatement of what m Stage-2 passes leave in a contig: the reference
    sorts the members at the start of each scan (stable, by offset then direction, kthread_hash_realign.c:318) and
    appends behind them; a pass that appends nothing still sorts."""
    import torch
    rng = np.random.default_rng(11)
    nc, kb = 3000, 12
    sizes = rng.integers(0, 40, nc)
    sizes[:5] = 0
    moff = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    def members(k):
        rid = rng.integers(0, 1 << 30, k).astype(np.uint64)
        pos = rng.integers(0, 1 << (kb - 2), k).astype(np.uint64)
        pos[rng.random(k) < 0.3] = 7                                          # equal keys: the stable order shows
        return (rid << np.uint64(32)) | (pos << np.uint64(1)) | rng.integers(0, 2, k).astype(np.uint64)
    mem = members(int(moff[-1]))
    for n_passes, empty_last in ((1, False), (3, False), (3, True), (2, True)):
        passes = []
        for i in range(n_passes):
            k = 0 if (empty_last and i == n_passes - 1) else int(rng.integers(200, 4000))
            c = np.sort(rng.integers(0, nc, k)).astype(np.int32)                # a pass appends in contig order
            passes.append((c, members(k)))
        lists = [list(mem[int(moff[c]):int(moff[c + 1])]) for c in range(nc)]
        key = lambda y: (int(y) & 0xFFFFFFFF)
        for c_arr, m_arr in passes:
            lists = [sorted(l, key=key) for l in lists]                          # sorted() is stable
            for c, y in zip(c_arr.tolist(), m_arr.tolist()):
                lists[c].append(np.uint64(y))
        want_off = np.concatenate([[0], np.cumsum([len(l) for l in lists])]).astype(np.int64)
        want = np.array([y for l in lists for y in l], dtype=np.uint64)
        d = lambda a, t: torch.from_numpy(a.view(t)).cuda()
        got, got_off = ctx.members_finalize(d(mem, np.int64), d(moff, np.int64), [(d(c, np.int32), d(m, np.int64)) for c, m in passes], kb)
        ctx.sync()
        assert np.array_equal(got_off.cpu().numpy(), want_off), (n_passes, empty_last)
        assert np.array_equal(got.cpu().numpy().view(np.uint64), want), (n_passes, empty_last)


def test_compact_live_and_window_layout(ctx):
    import torch
    rng = np.random.default_rng(12)
    ids = rng.integers(0, 1 << 31, 100001).astype(np.int32)
    flag = (rng.random(len(ids)) < 0.7).astype(np.uint8) * rng.integers(1, 4, len(ids)).astype(np.uint8)
    got = ctx.compact_live(torch.from_numpy(ids).cuda(), torch.from_numpy(flag).cuda())
    assert np.array_equal(got.cpu().numpy(), ids[flag == 0])
    lens = rng.integers(1, 400, 5000); lens[:3] = [99, 100, 101]
    soff = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    woff, nw, ml = ctx.window_layout(torch.from_numpy(soff).cuda(), 100)
    want = np.concatenate([[0], np.cumsum(np.maximum(lens - 99, 0))])
    assert np.array_equal(woff.cpu().numpy(), want) and nw == int(want[-1]) and ml == int(lens.max())


def test_counters_from_the_zeroed_pool_survive_its_turnover(ctx):
    """Kernels that count into one word (maxima, overflow counts, totals) take it from a pool of zeroed words with two halves that are
    cleared and taken in turn (csrc/api.hip, mcom_zeroed): thousands of requests in a row -- several turnovers -- must all start from zero."""
    import torch
    rng = np.random.default_rng(5)
    for it in range(9000):
        n = 3 + it % 5
        lens = rng.integers(10, 4000, n)
        soff = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)).cuda()
        woff, nw, longest = ctx.window_layout(soff, 100)
        assert longest == int(lens.max()), (it, longest, lens)
        assert nw == int(np.maximum(lens - 99, 0).sum())


def test_merge_rounds_without_copies_list_offsets_and_gather(ctx):
    """The pieces of the merge rounds that leave the contig set where it is (include/mcom.h, "Merge rounds without cp_cluster's copies";
    the reference copies every unmerged contig every round, kthread_cb.c:397-434): the next round's list = new contigs in claiming order,
    then the unclaimed ones of this round's list in their order; offsets of appended contigs; the gather that ends the rounds."""
    import torch
    rng = np.random.default_rng(11)
    n_store = 5000
    # a list over a store: a random subset in random order (what several rounds leave), flags on some of them and on contigs outside the list
    lst = rng.permutation(n_store)[:3000].astype(np.int32)
    flag = (rng.random(n_store) < 0.3).astype(np.uint8)
    nj = 400
    got, nk = ctx.order_next(_dev(lst), len(lst), _dev(flag), n_store, nj)
    want = np.concatenate([np.arange(n_store, n_store + nj), lst[flag[lst] == 0]]).astype(np.int32)
    assert nk == int((flag[lst] == 0).sum()) and np.array_equal(got.cpu().numpy(), want)
    got, nk = ctx.order_next(None, n_store, _dev(flag), n_store, 0)                        # no list yet: the store's own order
    assert np.array_equal(got.cpu().numpy(), np.nonzero(flag == 0)[0].astype(np.int32))
    got, nk = ctx.order_next(None, 0, None, 77, 1000)                                      # nothing but new contigs: 77, 78, ...
    assert nk == 0 and np.array_equal(got.cpu().numpy(), np.arange(77, 1077, dtype=np.int32))
    # offsets of appended contigs
    rel = np.concatenate([[0], np.cumsum(rng.integers(1, 900, 300))]).astype(np.int64)
    dst = torch.zeros(1000, dtype=torch.int64, device="cuda")
    ctx.offsets_append(_dev(rel), 123456789012, dst, 650); ctx.sync()
    assert np.array_equal(dst.cpu().numpy()[650:951], rel + 123456789012) and int(dst[:650].abs().sum()) == 0 and int(dst[951:].abs().sum()) == 0
    rel32 = rel.astype(np.int32); dst32 = torch.zeros(1000, dtype=torch.int32, device="cuda")
    ctx.offsets_append(_dev(rel32), 4000, dst32, 10); ctx.sync()
    assert np.array_equal(dst32.cpu().numpy()[10:311], rel32 + 4000)
    # the gather: contigs of the list as a set of their own
    slen = rng.integers(0, 700, n_store); mlen = rng.integers(1, 40, n_store)
    soff = np.concatenate([[0], np.cumsum(slen)]).astype(np.int64); moff = np.concatenate([[0], np.cumsum(mlen)]).astype(np.int64)
    seq = ACGT[rng.integers(0, 4, int(soff[-1]))]; mem = rng.integers(0, 1 << 62, int(moff[-1]), dtype=np.int64)
    s2, so2, m2, mo2 = ctx.contigs_gather(_dev(seq), _dev(soff), _dev(mem), _dev(moff), _dev(lst))
    assert np.array_equal(s2.cpu().numpy(), np.concatenate([seq[soff[c]:soff[c + 1]] for c in lst]))
    assert np.array_equal(m2.cpu().numpy(), np.concatenate([mem[moff[c]:moff[c + 1]] for c in lst]))
    assert np.array_equal(so2.cpu().numpy(), np.concatenate([[0], np.cumsum(slen[lst])])) and np.array_equal(mo2.cpu().numpy(), np.concatenate([[0], np.cumsum(mlen[lst])]))
