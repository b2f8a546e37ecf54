"""GPU: mcom_resketch_merged (parents' records + a sketch around the overlap) must equal mcom_sketch_contigs of the
merged contigs -- which tests/test_gpu_contigs.py pins to the oracle's mm_sketch_lh_ori."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


@pytest.fixture(scope="module")
def ctx():
    import minicom_amd
    c = minicom_amd.Context(0)
    yield c
    c.close()


def _string(rng, n, kind):
    if kind == 0:
        return ACGT[rng.integers(0, 4, n)]
    if kind == 1:                                        # short-period repeats: equal hashes inside a window, ties everywhere
        unit = ACGT[rng.integers(0, 4, int(rng.integers(1, 7)))]
        return np.tile(unit, n // len(unit) + 1)[:n].copy()
    s = ACGT[rng.integers(0, 4, n)]                      # random with repeated blocks
    for _ in range(3):
        a, b, ln = int(rng.integers(0, n)), int(rng.integers(0, n)), int(rng.integers(5, 90))
        ln = min(ln, n - a, n - b)
        s[b:b + ln] = s[a:a + ln].copy()
    return s


def _cat(strings):
    off = np.zeros(len(strings) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(s) for s in strings])
    return np.concatenate(strings) if strings else np.zeros(0, np.uint8), off


@pytest.mark.parametrize("w,k,seed", [(44, 31, 1), (19, 31, 2), (3, 17, 3), (10, 15, 4), (44, 31, 5), (128, 21, 6), (1, 9, 7)])
def test_resketch_equals_full_sketch_of_the_merged_contigs(ctx, w, k, seed):
    import torch
    rng = np.random.default_rng(seed)
    parents, jobs, merged = [], [], []
    for j in range(700):
        kind = int(rng.integers(0, 3)) if seed != 5 else 1
        long_ = rng.random() < 0.5
        lf = int(rng.integers(1, 3000 if long_ else 300))
        ls = int(rng.integers(1, 3000 if long_ else 300))
        f = _string(rng, lf, kind)
        r = rng.random()
        if r < 0.15:
            sh = 0
        elif r < 0.3:
            sh = lf                                       # the second parent starts where the first ends: empty overlap
        elif r < 0.45:
            sh = max(0, lf - int(rng.integers(1, 40)))    # short overlaps
        else:
            sh = int(rng.integers(0, lf + 1))
        if rng.random() < 0.2:
            ls = max(1, min(ls, (lf - sh) // 2))          # contained in the first parent
        s = _string(rng, ls, kind)
        lo, hi = min(sh, lf), max(min(sh, lf), min(lf, sh + ls))
        # the parents agree outside the overlap by construction; inside it the merged contig is a third string
        ov = f[lo:hi].copy()
        for q in rng.integers(0, max(hi - lo, 1), int(rng.integers(0, 4))):
            if hi > lo:
                ov[q] = ACGT[rng.integers(0, 4)]
        m = max(lf, sh + ls)
        tail = f[hi:] if hi < lf else s[hi - sh:]
        mg = np.concatenate([f[:lo], ov, tail])
        assert len(mg) == m
        pos = int(rng.integers(0, 50))
        if rng.random() < 0.5:
            ci, cj, po, pp = 2 * j, 2 * j + 1, pos + sh, pos      # first parent is ci
            parents += [f, s]
        else:
            ci, cj, po, pp = 2 * j, 2 * j + 1, pos, pos + sh      # first parent is cj
            parents += [s, f]
            if sh == 0:
                parents[-2:] = [f, s]                              # pos_ori >= pos: ci is taken as the first
        jobs.append((ci, cj, po, pp))
        merged.append(mg)
    pseq, poff = _cat(parents)
    mseq, moff = _cat(merged)
    d = lambda a, t=None: torch.from_numpy(a.view(t) if t is not None else a).cuda()
    d_pseq, d_poff, d_mseq, d_moff = d(pseq), d(poff, np.int64), d(mseq), d(moff, np.int64)
    proff, prec = ctx.sketch_contigs(d_pseq, d_poff, len(parents), w, k)
    want_off, want = ctx.sketch_contigs(d_mseq, d_moff, len(merged), w, k)
    d_jobs = torch.from_numpy(np.array(jobs, dtype=np.uint32).view(np.int32)).cuda()
    got_off, got, sketched = ctx.resketch_merged(d_jobs, d_poff, prec, proff, d_mseq, d_moff, int(moff[-1]), w, k)
    ctx.sync()
    assert torch.equal(got_off, want_off)
    assert torch.equal(got, want)
    assert int(want.shape[0]) > 1000 or w > 100
    assert 0 < sketched <= int(moff[-1])
    if w <= 44:
        assert sketched < 0.8 * int(moff[-1])             # the long contigs are not sketched whole


def test_resketch_rejects_even_k(ctx):
    import torch
    import minicom_amd
    z = torch.zeros(4, dtype=torch.int64, device="cuda")
    with pytest.raises(minicom_amd.McomError):
        ctx.resketch_merged(torch.zeros((1, 4), dtype=torch.int32, device="cuda"), z, ctx.empty_records(1), z.int(), z.to(torch.uint8), z, 10, 5, 24)
