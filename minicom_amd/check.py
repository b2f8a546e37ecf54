"""Size-independent verification of a pipeline result (tests at benchmark size, bench.py's checked run).

NOT part of the product path: an independent check written with plain torch tensor operations on the device (bincount,
gathers, comparisons), so that it shares no kernel with the library it checks.  Properties (the ones that make the
archive lossless, whatever the size):
  * every read id lies in exactly one place: one contig's member list, the unclustered list (sg), or one of the class
    lists (all-A/T/N, near-poly-A/T/N, N-heavy);
  * every member lies on its contig (offset + L <= length of the consensus string), member lists are non-empty;
  * every member resembles its contig there: the read (reverse-complemented when its direction bit is set) against the
    consensus window; positions where the read holds an N are not counted (the pipeline substitutes them).
"""
from __future__ import annotations

import numpy as np

LIST_NAMES = ("sg", "allA", "allT", "allN", "fpA", "fpT", "fpN", "Nfile")


def check_result(p, reads, L: int, rid0: int = 0, chunk: int = 1 << 21) -> dict:
    """p: a Pipeline / DistPipeline after pre_process().  reads: torch uint8 CUDA tensor [m, pitch] holding reads
    rid0 .. rid0 + m - 1 as ASCII (m = all reads on one GPU; a rank's shard in a multi-GPU job: the coverage test always
    runs over all ids, the resemblance test over the members whose reads this rank holds).  Returns the findings; raises
    AssertionError on a violated property."""
    import torch
    dev = reads.device
    n = p.n
    ref, roff, mem, moff = p.contig_set()
    nc = len(roff) - 1
    t_ref = torch.from_numpy(ref).to(dev)
    t_roff = torch.from_numpy(roff.view(np.int64)).to(dev)
    t_mem = torch.from_numpy(mem.view(np.int64)).to(dev)
    t_moff = torch.from_numpy(moff.view(np.int64)).to(dev)
    out = {"n_reads": int(n), "n_contigs": int(nc), "members": int(len(mem)), "chars": int(len(ref))}

    # ---- every read in exactly one place
    rid = (t_mem >> 32) & 0xFFFFFFFF
    count = torch.bincount(rid, minlength=n) if len(mem) else torch.zeros(n, dtype=torch.int64, device=dev)
    assert count.numel() == n, "a member word holds a read id beyond the number of reads"
    for name in LIST_NAMES:
        ids = p.id_list(name)
        out["n_" + name] = int(len(ids))
        if len(ids):
            t = torch.from_numpy(ids.astype(np.int64)).to(dev)
            assert int(t.max()) < n, f"list {name} holds a read id beyond the number of reads"
            count += torch.bincount(t, minlength=n)
    missing, twice = int((count == 0).sum()), int((count > 1).sum())
    assert missing == 0 and twice == 0, f"{missing} reads are in no place, {twice} in more than one"
    del count

    # ---- members lie on their contigs
    sizes = t_moff[1:] - t_moff[:-1]
    assert nc == 0 or int(sizes.min()) >= 1, "a contig without members"
    cid = torch.repeat_interleave(torch.arange(nc, device=dev), sizes)
    off = (t_mem & 0xFFFFFFFF) >> 1
    rev = (t_mem & 1).bool()
    clen = t_roff[1:] - t_roff[:-1]
    over = int(((off + L) > clen[cid]).sum()) if nc else 0
    assert over == 0, f"{over} members reach beyond the end of their contig"
    out["longest_contig"] = int(clen.max()) if nc else 0
    out["largest_contig_members"] = int(sizes.max()) if nc else 0

    # ---- members resemble their contigs (the reads this process holds)
    m = reads.shape[0]
    comp = torch.full((256,), ord("N"), dtype=torch.uint8, device=dev)
    for a, b in zip(b"ACGTN", b"TGCAN"):
        comp[a] = b
    ar = torch.arange(L, device=dev)
    mine = ((rid >= rid0) & (rid < rid0 + m)).nonzero().squeeze(1) if (rid0 or m < n) else None
    total = len(mem) if mine is None else int(mine.numel())
    worst, summ = 0, 0
    hist = torch.zeros(L + 1, dtype=torch.int64, device=dev)
    for s in range(0, total, chunk):
        sel = slice(s, min(total, s + chunk)) if mine is None else mine[s:s + chunk]
        r = reads[rid[sel] - rid0][:, :L]
        rc = comp[r.long()].flip(1)
        r = torch.where(rev[sel][:, None], rc, r)
        win = t_ref[(t_roff[cid[sel]] + off[sel])[:, None] + ar]
        mm = ((r != win) & (r != ord("N"))).sum(1)
        worst = max(worst, int(mm.max()))
        summ += int(mm.sum())
        hist += torch.bincount(mm, minlength=L + 1)
    out["members_checked"] = total
    out["max_mismatch"] = worst
    out["mean_mismatch"] = summ / max(1, total)
    out["mismatch_hist_head"] = hist[:12].tolist()
    return out
