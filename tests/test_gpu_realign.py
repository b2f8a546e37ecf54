"""GPU parity: Stage-2 dictionaries + window scan (mcom_dicts_*, mcom_poly_filter, mcom_realign_pass)
against the oracle's sequential realign pass, which is pinned to the reference dumps."""
import gzip
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


def _golden_reads(golden_dir, tag):
    with gzip.open(os.path.join(golden_dir, tag + ".reads.gz"), "rb") as f:
        rows = f.read().split(b"\n")[:-1]
    return np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), len(rows[0])).copy()


def _stage1(reads, **kw):
    import oracle
    p = oracle.Pipeline(reads, **kw)
    p.stage_reads(); p.stage_bucket(); p.stage_combine()
    return p


def _gpu_pass(ctx, p, reads_orig, thr, check_dicts=False):
    """One Stage-2 pass on the GPU from the oracle's pre-pass state; returns (flags, appended members per contig)."""
    import torch
    from minicom_amd.hip import pack_nt4, pack_contigs
    L = p.L
    p.update_single()
    sg = p.sg
    seq = p.seq                                             # N substituted for kept reads
    keep = p.cls == 0
    packed = np.zeros((p.n, (2 * L + 63) // 64), dtype=np.uint64)
    packed[keep] = pack_nt4(seq[keep])
    d_packed = torch.from_numpy(packed.view(np.int64)).cuda()
    nmask = np.zeros((p.n, (L + 63) // 64), dtype=np.uint64)
    isn = reads_orig == ord("N")
    for w in range(nmask.shape[1]):
        seg = isn[:, 64 * w: 64 * (w + 1)]
        nmask[:, w] = (seg.astype(np.uint64) << np.arange(seg.shape[1], dtype=np.uint64)[None, :]).sum(axis=1, dtype=np.uint64)
    d_nmask = torch.from_numpy(nmask.view(np.int64)).cuda()
    d_sg = torch.from_numpy(sg.astype(np.int32)).cuda()
    sgbits = ctx.gather_rows(d_packed, d_sg, L)
    flag = ctx.poly_filter(sgbits, L, thr, nmask=d_nmask, rids=d_sg)
    dicts = ctx.dicts_build(sgbits, L)
    contigs = p.contigs()
    refs = [r for r, _ in contigs]
    cbits, coff, clen = pack_contigs(refs)
    nwin = np.maximum(clen.astype(np.int64) - L + 1, 0)
    woff = np.concatenate([[0], np.cumsum(nwin)]).astype(np.uint64)
    maxsearch = p.counter("maxsearch")
    d_cbits, d_coff, d_woff = (torch.from_numpy(a.view(np.int64)).cuda() for a in (cbits, coff, woff))
    claim, st = ctx.realign_pass(dicts, sgbits, flag, d_cbits, d_coff, d_woff, int(nwin.sum()), thr, maxsearch, stats=True)
    # the read-driven form of the pass (the production path) must give the same claims
    cix = ctx.cindex_build(d_cbits, d_coff, d_woff, int(nwin.sum()), L)
    claim_r, st_r = ctx.realign_pass_reads(cix, sgbits, flag, d_cbits, d_coff, d_woff, L, thr, stats=True)
    ctx.sync()
    assert torch.equal(claim, claim_r)
    # device-side claim resolution = the lexsort below
    f2 = flag.clone()
    ac, am = ctx.claims_resolve(claim_r, d_sg, len(refs), f2)
    o = np.lexsort((-np.arange(len(sg)), claim_r.cpu().numpy().view(np.uint64)))
    o = o[(claim_r.cpu().numpy() != -1)[o]]
    ck = claim_r.cpu().numpy().view(np.uint64)[o]
    assert np.array_equal(ac.cpu().numpy(), (ck >> np.uint64(33)).astype(np.int32))
    assert np.array_equal(am.cpu().numpy().view(np.uint64), (sg[o].astype(np.uint64) << np.uint64(32)) | (((ck >> np.uint64(5)) & np.uint64((1 << 28) - 1)) << np.uint64(1)) | ((ck >> np.uint64(4)) & np.uint64(1)))
    assert np.array_equal(f2.cpu().numpy(), np.where(claim_r.cpu().numpy() != -1, 3, flag.cpu().numpy()))
    assert int(st_r[0]) > 0 and int(st_r[1]) >= int(st_r[2]) >= int((claim_r != -1).sum())
    if check_dicts:
        _check_dicts(ctx, dicts, sgbits.cpu().numpy().view(np.uint64), L)
    claim = claim.cpu().numpy().view(np.uint64)
    flag = flag.cpu().numpy()
    assert max(dicts.maxbin) <= maxsearch, "fixture has a bin above maxsearch: the parallel claim rule is not exact there"
    claimed = claim != MAXU
    # appended members per contig, reference order: claim key ascending, singleton index descending
    order = np.lexsort((-np.arange(len(sg)), claim))
    order = order[claimed[order]]
    app = {}
    for i in order:
        ck = int(claim[i])
        c, jj, d = ck >> 33, (ck >> 5) & ((1 << 28) - 1), (ck >> 4) & 1
        app.setdefault(c, []).append((int(sg[i]) << 32) | (jj << 1) | d)
    dicts.close()
    return flag, claimed, app, [len(m) for _, m in contigs], st.cpu().numpy()


def _check_dicts(ctx, dicts, sgbits, L):
    """mcom_dicts_lookup / ids against a numpy statement of constructdictionary_realign's CSR layout."""
    import torch
    from minicom_amd.hip import dict_layout
    st, en = dict_layout(L)
    assert dicts.nd == len(st)
    for j in range(dicts.nd):
        bo, kl = 2 * st[j], en[j] - st[j] + 1
        wi, sh = bo >> 6, bo & 63
        key = sgbits[:, wi] >> np.uint64(sh)
        if sh and sh + 2 * kl > 64:
            key = key | (sgbits[:, wi + 1] << np.uint64(64 - sh))
        key = key & np.uint64((1 << (2 * kl)) - 1)
        order = np.argsort(key, kind="stable")
        assert np.array_equal(dicts.ids(j).cpu().numpy().view(np.uint32), order.astype(np.uint32))
        uk, first, cnt = np.unique(key, return_index=True, return_counts=True)
        assert dicts.numkeys[j] == len(uk) and dicts.maxbin[j] == cnt.max()
        sk = key[order]
        starts = np.flatnonzero(np.r_[True, sk[1:] != sk[:-1]])
        probe = np.concatenate([uk, uk ^ np.uint64(1), np.array([0, (1 << (2 * kl)) - 1], dtype=np.uint64)])
        s, c = dicts.lookup(j, torch.from_numpy(probe.view(np.int64)).cuda())
        s, c = s.cpu().numpy(), c.cpu().numpy()
        have = dict(zip(uk.tolist(), zip(starts.tolist(), cnt.tolist())))
        for q, kq in enumerate(probe.tolist()):
            if kq in have:
                assert (int(s[q]), int(c[q])) == have[kq]
            else:
                assert int(c[q]) == 0


@pytest.mark.parametrize("tag", ["stages_L100", "stages_L150", "stages_L40"])
def test_every_realign_pass_matches_sequential_reference_semantics(ctx, golden_dir, tag):
    """All passes of Stage 2 on the reference fixture reads: flags and the members appended to every contig,
    in order, equal the sequential scan (itself byte-identical to the reference dump)."""
    reads = _golden_reads(golden_dir, tag)
    L = reads.shape[1]
    p = _stage1(reads)
    thr, step, pre, npass = 4, 4, 0, 0
    while thr <= L // 2:
        flag, claimed, app, nbefore, st = _gpu_pass(ctx, p, reads, thr, check_dicts=(npass == 0))
        fa0, ft0 = len(p.id_list("fpA")), len(p.id_list("fpT"))
        cr = p.realign_pass(thr)                                  # oracle, sequential
        want_flag = p.sg_flag
        assert np.array_equal((flag != 0) | claimed, want_flag != 0)
        sg = p.sg
        assert np.array_equal(sg[flag == 1], p.id_list("fpA")[fa0:])
        assert np.array_equal(sg[flag == 2], p.id_list("fpT")[ft0:])
        after = p.contigs()
        n_app = 0
        for c, (_, mem) in enumerate(after):
            got = app.get(c, [])
            assert [int(v) for v in mem[nbefore[c]:]] == got, (tag, thr, c)
            n_app += len(got)
        assert n_app == int(claimed.sum())
        assert st[0] > 0 and st[1] > 0 and st[2] >= n_app
        npass += 1
        if cr - pre < 1000:
            break
        pre = cr; thr += step
    assert npass >= 2


def test_realign_high_threshold_uses_cost_filter_on_reverse_strand(ctx):
    """thr > 24 switches encode_byte on for reverse-strand candidates (kthread_hash_realign.c:461)."""
    from minicom_amd import synth
    reads = synth.synth_reads(31337, 4000, 100, sub_rate=0.03)
    p = _stage1(reads)
    total, rev_hi = 0, 0
    for thr in (4, 28, 40):
        flag, claimed, app, nbefore, _ = _gpu_pass(ctx, p, reads, thr)
        p.realign_pass(thr)
        assert np.array_equal((flag != 0) | claimed, p.sg_flag != 0)
        for c, (_, mem) in enumerate(p.contigs()):
            assert [int(v) for v in mem[nbefore[c]:]] == app.get(c, []), (thr, c)
        total += int(claimed.sum())
        if thr > 24:
            rev_hi += sum(1 for v in app.values() for y in v if y & 1)
    assert total > 0 and rev_hi >= 0


@pytest.mark.parametrize("L,maxsearch", [(150, 500), (150, 3), (150, 1), (100, 2), (64, 2), (200, 4)])
def test_read_driven_pass_equals_window_scan_also_with_bins_above_maxsearch(ctx, L, maxsearch):
    """Random contigs, singletons cut from them (both strands, substitutions, many duplicates so that bins exceed
    maxsearch): mcom_realign_pass_reads + mcom_dicts_eligible give the claims of the window scan."""
    import torch
    from minicom_amd.hip import pack_nt4, pack_contigs
    rng = np.random.default_rng(L * 1000 + maxsearch)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    comp = np.zeros(256, dtype=np.uint8); comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
    lens = rng.integers(L, 6 * L, 150)
    lens[:3] = [L, L + 1, 2 * L]
    refs = [acgt[rng.integers(0, 4, n)] for n in lens]
    refs[7] = refs[6].copy()                                               # a repeated contig: several windows per key
    reads = []
    for _ in range(6000):
        c = int(rng.integers(0, len(refs)))
        j = int(rng.integers(0, len(refs[c]) - L + 1))
        if rng.random() < 0.3:
            j = min(j, 3)                                                  # pile-ups: many reads with equal keys
        r = refs[c][j:j + L].copy()
        for q in rng.integers(0, L, int(rng.integers(0, 7))):
            r[q] = acgt[rng.integers(0, 4)]
        reads.append(comp[r][::-1] if rng.random() < 0.5 else r)
    reads = np.stack(reads)
    sgbits = torch.from_numpy(pack_nt4(reads).view(np.int64)).cuda()
    flag = torch.from_numpy((rng.random(len(reads)) < 0.05).astype(np.uint8)).cuda()
    cbits, coff, clen = pack_contigs([r.tobytes() for r in refs])
    nwin = np.maximum(clen.astype(np.int64) - L + 1, 0)
    woff = np.concatenate([[0], np.cumsum(nwin)]).astype(np.uint64)
    d_cbits, d_coff, d_woff = (torch.from_numpy(a.view(np.int64)).cuda() for a in (cbits, coff, woff))
    dicts = ctx.dicts_build(sgbits, L)
    cix = ctx.cindex_build(d_cbits, d_coff, d_woff, int(nwin.sum()), L)
    if maxsearch < 500:
        assert max(dicts.maxbin) > maxsearch
    # the screen may only say "no bin exceeds" when that is true, and must say so for a generous limit
    assert ctx.dicts_screen(sgbits, L, maxsearch) or max(dicts.maxbin) <= maxsearch
    assert not ctx.dicts_screen(sgbits, L, 100000)
    for thr in (4, 12, 28):
        want, _ = ctx.realign_pass(dicts, sgbits, flag, d_cbits, d_coff, d_woff, int(nwin.sum()), thr, maxsearch)
        elig = ctx.dicts_eligible(dicts, sgbits, maxsearch) if max(dicts.maxbin) > maxsearch else None
        got, _ = ctx.realign_pass_reads(cix, sgbits, flag, d_cbits, d_coff, d_woff, L, thr, elig=elig)
        ctx.sync()
        assert torch.equal(want, got), (thr, int((want != got).sum()))
        assert int((want != -1).sum()) > 1000
    dicts.close()


@pytest.mark.parametrize("L,n_sg,copies,overflows", [(150, 300_000, 0, False), (150, 300_000, 3000, True), (100, 40_000, 3000, None), (40, 5_000, 0, False),
                                                      (150, 700, 700, False), (150, 4_500_000, 0, False), (256, 70_000, 70_000, None)])
def test_dictionary_screen_binned_in_lds_and_by_global_atomics_give_sound_answers(ctx, L, n_sg, copies, overflows):
    """mcom_dicts_screen's routes (include/mcom_test.h): counter numbers binned by counter range and counted in LDS (default), global
    atomics (the fall-back), and the default made to overflow its regions.  Whatever the route: the answer "no bin exceeds maxsearch" only
    when that is true (checked against the bins of the dictionaries actually built, constructdictionary_realign,
    kthread_hash_realign.c:3-140), and always for a generous limit.  `copies` rows are one and the same read: one key per dictionary,
    `copies` times, side by side in the list -- the pile-up that overflows a region of the binned route when the counters fill several bins."""
    import torch
    rng = np.random.default_rng(L + n_sg)
    W = (2 * L + 63) // 64
    rows = rng.integers(-2**63, 2**63 - 1, size=(n_sg, W), dtype=np.int64)
    if copies:
        at = min(n_sg // 3, n_sg - copies)
        rows[at:at + copies] = rows[0]
    sgbits = torch.from_numpy(rows).cuda()
    dicts = ctx.dicts_build(sgbits, L)
    m = int(max(dicts.maxbin))
    assert m >= max(copies, 1)
    try:
        for route, falls_back in ((0, overflows), (1, False), (2, True)):
            ctx.set_screen_route(route)
            before = ctx.screen_fallbacks()
            if m > 1: assert ctx.dicts_screen(sgbits, L, m - 1)                           # a bin of m reads exceeds m - 1: the screen must see it
            else: ctx.dicts_screen(sgbits, L, 1)
            if falls_back is not None: assert (ctx.screen_fallbacks() > before) == falls_back
            assert not ctx.dicts_screen(sgbits, L, m + 400)                               # (a counter holds its bins + a few dozen colliding keys)
    finally:
        ctx.set_screen_route(0)


def test_index_keeps_heavy_repeats_in_runs_of_their_own(ctx):
    """Keys with hundreds of copies among the contigs (a 1 kb segment in 120 contigs, poly-A and (AC)n stretches): their
    entries leave the partitions for runs in the extension area (csrc/cindex.hip, CIX_HEAVY).  The claims must still be the
    window scan's; the index reports how many extension lines it used."""
    import torch
    from minicom_amd.hip import pack_nt4, pack_contigs
    L = 100
    rng = np.random.default_rng(321)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    comp = np.zeros(256, dtype=np.uint8); comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
    seg = acgt[rng.integers(0, 4, 1000)]
    refs = [np.concatenate([acgt[rng.integers(0, 4, int(rng.integers(50, 300)))], seg, acgt[rng.integers(0, 4, int(rng.integers(50, 300)))]]) for _ in range(120)]
    refs += [acgt[rng.integers(0, 4, int(n))] for n in rng.integers(L, 5 * L, 200)]
    refs.append(np.full(700, ord("A"), dtype=np.uint8))
    refs.append(np.frombuffer(b"AC" * 450, dtype=np.uint8).copy())
    reads = []
    for _ in range(5000):
        c = int(rng.integers(0, len(refs)))
        j = int(rng.integers(0, len(refs[c]) - L + 1))
        r = refs[c][j:j + L].copy()
        for q in rng.integers(0, L, int(rng.integers(0, 5))):
            r[q] = acgt[rng.integers(0, 4)]
        reads.append(comp[r][::-1] if rng.random() < 0.5 else r)
    reads = np.stack(reads)
    sgbits = torch.from_numpy(pack_nt4(reads).view(np.int64)).cuda()
    flag = torch.zeros(len(reads), dtype=torch.uint8, device="cuda")
    cbits, coff, clen = pack_contigs([r.tobytes() for r in refs])
    nwin = np.maximum(clen.astype(np.int64) - L + 1, 0)
    woff = np.concatenate([[0], np.cumsum(nwin)]).astype(np.uint64)
    d_cbits, d_coff, d_woff = (torch.from_numpy(a.view(np.int64)).cuda() for a in (cbits, coff, woff))
    dicts = ctx.dicts_build(sgbits, L)
    cix = ctx.cindex_build(d_cbits, d_coff, d_woff, int(nwin.sum()), L)
    assert int(cix[0][0]) > 100                                            # extension lines in use: the heavy keys went there
    ctx.set_index_capacity(0)                                              # ... and the same index placed by the scattered kernel
    try:
        cix2 = ctx.cindex_build(d_cbits, d_coff, d_woff, int(nwin.sum()), L)
    finally:
        ctx.set_index_capacity(-1)
    for thr in (4, 16):
        want, _ = ctx.realign_pass(dicts, sgbits, flag, d_cbits, d_coff, d_woff, int(nwin.sum()), thr, 100000)
        for index in (cix, cix2):
            got, _ = ctx.realign_pass_reads(index, sgbits, flag, d_cbits, d_coff, d_woff, L, thr)
            ctx.sync()
            assert torch.equal(want, got), (thr, int((want != got).sum()))
        assert int((want != -1).sum()) > 2000
    dicts.close()


@pytest.mark.parametrize("L,ranks,heavy", [(150, 2, False), (150, 3, False), (150, 8, False), (100, 8, True), (100, 5, True), (64, 16, False), (200, 8, False)])
def test_lookups_over_shares_of_the_keys_min_reduced_equal_the_whole_index(ctx, L, ranks, heavy):
    """Several GPUs share the ONE contig index out BY KEY (mcom_cindex_plan_shared; host/mcom_pipeline.cpp "Stage 2"): rank q places the
    entries whose keys hash into share q and looks up, for every singleton, only the keys it owns; the claim keys are MIN-reduced
    (kthread_hash_realign.c:316-508 takes the first dictionary / position that passes: a minimum).  Here all shares are built
    on the one card: the minimum over the shares must be the whole index's claim of every singleton, pass by pass, for both kernels of
    the shared lookups (include/mcom_test.h, mcom_set_lookup_route: a thread per owned task / pairs per lane), with pile-ups behind
    mcom_dicts_eligible and with heavy repeats in the extension area; and the statistics of the shares add up to the whole index's."""
    import torch
    from minicom_amd.hip import pack_nt4, pack_contigs
    rng = np.random.default_rng(L * 100 + ranks)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    comp = np.zeros(256, dtype=np.uint8); comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
    refs = [acgt[rng.integers(0, 4, int(n))] for n in rng.integers(L, 6 * L, 300)]
    if heavy:
        seg = acgt[rng.integers(0, 4, 800)]
        refs += [np.concatenate([acgt[rng.integers(0, 4, int(rng.integers(50, 300)))], seg, acgt[rng.integers(0, 4, int(rng.integers(50, 300)))]]) for _ in range(100)]
        refs.append(np.full(600, ord("A"), dtype=np.uint8))
    reads = []
    for _ in range(21_000):                                                # (more than one workgroup's singletons, and not a multiple of a batch)
        c = int(rng.integers(0, len(refs)))
        if rng.random() < 0.2:
            c %= 8                                                         # pile-ups: bins of more than maxsearch reads at the heads of eight contigs
            j = int(rng.integers(0, 4))
        else:
            j = int(rng.integers(0, len(refs[c]) - L + 1))
        r = refs[c][j:j + L].copy()
        for q in rng.integers(0, L, int(rng.integers(0, 7))):
            r[q] = acgt[rng.integers(0, 4)]
        reads.append(comp[r][::-1] if rng.random() < 0.5 else r)
    reads = np.stack(reads)[:20_987]
    sgbits = torch.from_numpy(pack_nt4(reads).view(np.int64)).cuda()
    flag = torch.from_numpy((rng.random(len(reads)) < 0.1).astype(np.uint8)).cuda()
    cbits, coff, clen = pack_contigs([r.tobytes() for r in refs])
    nwin = np.maximum(clen.astype(np.int64) - L + 1, 0)
    woff = np.concatenate([[0], np.cumsum(nwin)]).astype(np.uint64)
    d_cbits, d_coff, d_woff = (torch.from_numpy(a.view(np.int64)).cuda() for a in (cbits, coff, woff))
    maxsearch = 40
    dicts = ctx.dicts_build(sgbits, L)
    elig = ctx.dicts_eligible(dicts, sgbits, maxsearch) if max(dicts.maxbin) > maxsearch else None
    assert elig is not None
    whole = ctx.cindex_build(d_cbits, d_coff, d_woff, int(nwin.sum()), L)
    shares = ctx.cindex_build_shares(d_cbits, d_coff, d_woff, int(nwin.sum()), L, ranks)
    assert len(shares) == ranks
    if heavy:
        assert sum(int(k[0]) for k, _ in shares) > 50                      # extension lines in use on the shares too
    big = torch.iinfo(torch.int64).max
    try:
        for thr in (4, 12, 28):
            want, st_want = ctx.realign_pass_reads(whole, sgbits, flag, d_cbits, d_coff, d_woff, L, thr, elig=elig, stats=True)
            ctx.sync()
            by_route = []
            for route in (0, 1):
                ctx.set_lookup_route(route)
                low = torch.full_like(want, big)
                sts = []
                for index in shares:
                    got, st = ctx.realign_pass_reads(index, sgbits, flag, d_cbits, d_coff, d_woff, L, thr, elig=elig, stats=True)
                    ctx.sync()
                    low = torch.minimum(low, torch.where(got < 0, torch.full_like(got, big), got))   # (UINT64_MAX = unclaimed)
                    sts.append(st.tolist())
                low = torch.where(low == big, torch.full_like(low, -1), low)
                assert torch.equal(want, low), (thr, route, int((want != low).sum()))
                # lookups and passing candidates add up to the whole index's (the verified ones in between count tag collisions, which are the
                # layout's: equal between the two kernels over the same share, not between a share and the whole index)
                tot = np.sum(np.array(sts, dtype=np.int64), axis=0)
                assert (int(tot[0]), int(tot[2])) == (int(st_want[0]), int(st_want[2])), (thr, route, st_want.tolist(), tot.tolist())
                by_route.append(sts)
            assert by_route[0] == by_route[1], (thr, by_route)
            assert int((want != -1).sum()) > 5000
    finally:
        ctx.set_lookup_route(0)
    dicts.close()


def test_realign_empty_inputs(ctx):
    import torch
    z64 = torch.zeros((0, 5), dtype=torch.int64, device="cuda")
    d = ctx.dicts_build(z64, 150)
    assert d.nd == 8 and d.numkeys == [0] * 8
    claim, _ = ctx.realign_pass(d, z64, torch.zeros(0, dtype=torch.uint8, device="cuda"), torch.zeros(1, dtype=torch.int64, device="cuda"),
                                torch.zeros(0, dtype=torch.int64, device="cuda"), torch.zeros(0, dtype=torch.int64, device="cuda"), 0, 4, 500)
    assert claim.numel() == 0
    d.close()


def test_dict_layout_matches_oracle():
    import oracle
    from minicom_amd.hip import dict_layout
    for L in (37, 64, 80, 81, 100, 101, 150, 151, 256):
        for nd in (0, 1, 2, 3, 5):
            st, en = dict_layout(L, nd)
            os_, oe = oracle.dict_layout(L, nd)
            assert st == os_.tolist() and en == oe.tolist()
