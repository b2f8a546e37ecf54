"""GPU parity: mcom_claim_pairs (rounds of locally-earliest pairs) against the sequential first-come loop of
find_next (kthread_cb.c:267-343) on synthetic candidate lists."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import minicom_amd
    c = minicom_amd.Context(0)
    yield c
    c.close()


def _sequential(pairs, n):
    """The reference's loop: contigs in index order, an unclaimed contig takes its first unclaimed candidate."""
    flag = np.zeros(n, dtype=np.uint8)
    jobs = []
    for ci, cj, po, pp in pairs:
        if flag[ci] or flag[cj]:
            continue
        jobs.append((ci, cj, po, pp)); flag[ci] = flag[cj] = 1
    return jobs, flag


def _records(pairs):
    a = np.array(pairs, dtype=np.uint64).reshape(-1, 4)
    x = (a[:, 0] << np.uint64(32)) | (a[:, 2] << np.uint64(1))            # the id is the contig index (include/mcom.h)
    y = (a[:, 1] << np.uint64(32)) | (a[:, 3] << np.uint64(1)) | np.uint64(1)
    return np.stack([x, y], axis=1)


def _lists(rng, n, deg, near=0):
    """Candidate lists in visiting order: contig ascending, a few candidates each (never itself)."""
    out = []
    for ci in range(n):
        for _ in range(int(rng.integers(0, deg + 1))):
            cj = int(rng.integers(0, n)) if not near else int(np.clip(ci + rng.integers(-near, near + 1), 0, n - 1))
            if cj != ci:
                out.append((ci, cj, int(rng.integers(0, 5000)), int(rng.integers(0, 5000))))
    return out


@pytest.mark.parametrize("n,deg,near", [(50, 3, 0), (5000, 4, 0), (200000, 3, 0), (3000, 2, 1), (20000, 5, 3)])
def test_claim_rounds_equal_the_sequential_loop(ctx, n, deg, near):
    import torch
    rng = np.random.default_rng(n + deg + near)
    pairs = _lists(rng, n, deg, near)
    want_jobs, want_flag = _sequential(pairs, n)
    rec = torch.from_numpy(_records(pairs).view(np.int64)).cuda()
    jobs, flag, rounds = ctx.claim_pairs(rec, n)
    ctx.sync()
    assert jobs.cpu().numpy().tolist() == [list(j) for j in want_jobs]
    assert np.array_equal(flag.cpu().numpy(), want_flag)
    assert rounds >= 1 and len(want_jobs) > 5


def test_claim_chain_needs_many_rounds_and_reports_when_it_does_not_settle(ctx):
    """A path 0-1-2-...: every taken pair frees the next one, one round each (near = 1 lists make such chains)."""
    import torch
    from minicom_amd.hip import McomError
    n = 400
    pairs = [(i, i + 1, 7, 9) for i in range(n - 1)]
    want_jobs, want_flag = _sequential(pairs, n)
    rec = torch.from_numpy(_records(pairs).view(np.int64)).cuda()
    jobs, flag, rounds = ctx.claim_pairs(rec, n)
    assert jobs.cpu().numpy().tolist() == [list(j) for j in want_jobs] and rounds >= n // 2 - 1
    with pytest.raises(McomError):
        ctx.claim_pairs(rec, n, max_rounds=10)


@pytest.mark.parametrize("route", [1, 2])
def test_claim_falls_back_to_the_launch_per_round_loop(ctx, route):
    """The one-launch kernel's grid barrier needs every workgroup resident (another process or thread-rank on the card can prevent
    that).  route 2 makes its first barrier give up: the poison flag trips, mcom_claim_pairs notices, clears it and redoes the claiming
    with two launches and a read-back per round; route 1 takes that loop at once.  Same jobs and flags as the sequential loop, and the
    context stays usable (the next default call runs the one-launch kernel again)."""
    import torch
    before = ctx.claim_fallbacks()
    try:
        ctx.set_claim_route(route)
        for n, deg, near in [(5000, 4, 0), (200000, 3, 0), (3000, 2, 1)]:
            rng = np.random.default_rng(n + deg + near)
            pairs = _lists(rng, n, deg, near)
            want_jobs, want_flag = _sequential(pairs, n)
            rec = torch.from_numpy(_records(pairs).view(np.int64)).cuda()
            jobs, flag, rounds = ctx.claim_pairs(rec, n)
            ctx.sync()
            assert jobs.cpu().numpy().tolist() == [list(j) for j in want_jobs]
            assert np.array_equal(flag.cpu().numpy(), want_flag) and rounds >= 1
        assert ctx.claim_fallbacks() == before + 3
        from minicom_amd.hip import McomError
        chain = [(i, i + 1, 7, 9) for i in range(399)]
        rec = torch.from_numpy(_records(chain).view(np.int64)).cuda()
        with pytest.raises(McomError):
            ctx.claim_pairs(rec, 400, max_rounds=10)
    finally:
        ctx.set_claim_route(0)
    pairs = _lists(np.random.default_rng(5), 5000, 4)
    want_jobs, _ = _sequential(pairs, 5000)
    jobs, _, _ = ctx.claim_pairs(torch.from_numpy(_records(pairs).view(np.int64)).cuda(), 5000)
    assert jobs.cpu().numpy().tolist() == [list(j) for j in want_jobs] and ctx.claim_fallbacks() == before + 4


def test_claim_tail_by_one_workgroup_equals_the_whole_grid(ctx):
    """Once at most 16384 edges are alive the one-launch kernel lists them and ONE workgroup runs the remaining rounds over the list
    (csrc/claim.hip, "the tail"); route 3 keeps every round on the whole grid.  Same jobs and same flags (the number of rounds may
    differ by the rounds that stale bids add: csrc/claim.hip), on lists far above the tail's size, near-diagonal lists (long chains) and a chain that does
    not settle inside the budget."""
    import torch
    from minicom_amd.hip import McomError
    try:
        for n, deg, near in [(1_500_000, 3, 0), (400_000, 4, 2), (30_000, 2, 1), (9_000, 6, 0)]:
            rng = np.random.default_rng(n + deg + near)
            ci = np.repeat(np.arange(n, dtype=np.int64), rng.integers(0, deg + 1, n))
            cj = rng.integers(0, n, len(ci)) if not near else np.clip(ci + rng.integers(-near, near + 1, len(ci)), 0, n - 1)
            ok = ci != cj
            ci, cj = ci[ok], cj[ok]
            pairs = np.stack([ci, cj, rng.integers(0, 5000, len(ci)), rng.integers(0, 5000, len(ci))], axis=1)
            rec = torch.from_numpy(_records(pairs.tolist()).view(np.int64)).cuda()
            got = {}
            for route in (0, 3):
                ctx.set_claim_route(route)
                jobs, flag, rounds = ctx.claim_pairs(rec, n)
                ctx.sync()
                got[route] = (jobs.cpu().numpy(), flag.cpu().numpy(), rounds)
            assert np.array_equal(got[0][0], got[3][0]) and np.array_equal(got[0][1], got[3][1])
            if n <= 30_000:
                want_jobs, want_flag = _sequential(pairs.tolist(), n)
                assert got[0][0].tolist() == [list(j) for j in want_jobs] and np.array_equal(got[0][1], want_flag)
        chain = [(i, i + 1, 7, 9) for i in range(19_999)]
        rec = torch.from_numpy(_records(chain).view(np.int64)).cuda()
        for route in (0, 3):
            ctx.set_claim_route(route)
            with pytest.raises(McomError):
                ctx.claim_pairs(rec, 20_000, max_rounds=300)
            jobs, flag, rounds = ctx.claim_pairs(rec, 20_000, max_rounds=10_001)
            assert jobs.shape[0] == 10_000 and rounds >= 9_999
    finally:
        ctx.set_claim_route(0)


def test_claim_empty(ctx):
    import torch
    jobs, flag, rounds = ctx.claim_pairs(torch.zeros((0, 2), dtype=torch.int64, device="cuda"), 7)
    assert jobs.shape[0] == 0 and int(flag.sum()) == 0
