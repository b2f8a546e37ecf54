"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed over RCCL ("nccl") / gloo.

The path has ONE real exchange step: after every rank has sketched its shard of the
reads, the 16-byte minimizer records -- here together with the 2-bit packed rows they belong to -- are
redistributed so that minimizer bucket  beta = x & (2^b - 1)  lives on rank  beta mod R.  All reads that share
a minimizer then sit on one rank, which runs the rest of the path (grouping, contigs, merging, realignment)
on its partition without further data-path collectives.  Every rank therefore produces an independent
archive of its partition: lossless for the union, not byte-identical to a single-process run (contigs never
span partitions).

xGMI is point to point (7 links x ~153 GB/s per GPU): an all-to-all drives all links at once, each peer pair on
its own link, so one large all_to_all_single per tensor is used, never a ring all-reduce of payload.

Everything here is torch-only plumbing (argsort / bincount / all_to_all_single), device agnostic, so the same
code is exercised with gloo on CPU in tests/test_distributed_cpu.py.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

BUCKET_BITS = 14


def bucket_owner(x: torch.Tensor, world: int) -> torch.Tensor:
    """Owner rank of every record: (x & 0x3fff) % world (x: int64 view of the unsigned hash)."""
    return (x & ((1 << BUCKET_BITS) - 1)) % world


# A single all_to_all_single message above about 1 GiB arrives HALF on this stack (RCCL of PyTorch 2.10 / ROCm 7.0,
# measured on MI355X: 1024 MiB intact, 1536 MiB and more only the first half -- tools/dbg_a2a.py), without any error.
# The exchange is therefore cut into slices whose per-peer messages stay far below that, and every slice is verified
# with a checksum that travels beside it.
MAX_MESSAGE_BYTES = 256 << 20


def _checksums(rows_s: torch.Tensor, counts: list) -> torch.Tensor:
    """Wrapping int64 sum of every per-peer segment of the send buffer."""
    out, o = [], 0
    for c in counts:
        out.append(rows_s[o:o + c].sum() if c else rows_s.new_zeros(()))
        o += c
    return torch.stack(out)


def exchange_by_bucket(rec_x: torch.Tensor, rids: torch.Tensor, rows: torch.Tensor, group=None, max_message_bytes: int = MAX_MESSAGE_BYTES,
                       extras=None):
    """All-to-all of the reads of this rank to the owners of their minimizer buckets.

    rec_x : int64 [n]     minimizer hash of every kept read of this rank
    rids  : int64 [n]     global read ids
    rows  : int64 [n, W]  packed rows
    extras: optional list of 1-D tensors [n] that travel with the reads (e.g. the minimizer hash and position of every
            read, so that the receiver need not sketch again); then a third result, the list of received tensors.
    Returns (rids_recv int64 [m], rows_recv int64 [m, W]).  The reads travel in slices of the sender's order (so that no
    message exceeds max_message_bytes); inside a slice they arrive ordered by source rank, then by the sender's order.
    Raises RuntimeError when a slice does not arrive intact."""
    world = dist.get_world_size(group)
    n, W = int(rows.shape[0]), int(rows.shape[1])
    owner_all = bucket_owner(rec_x, world)
    # The cap is per MESSAGE (one peer's share of a slice).  Buckets are spread evenly, so a slice of `world` times the
    # cap (taken at 0.6 of it) sends messages of about 0.6 cap: with eight ranks that is a sixth of the slices -- and of
    # their argsorts, splits and read-backs -- that a cap on the whole slice would need.  The largest message any rank is
    # about to send is agreed on first; a skewed slice is cut further, so the cap holds whatever the data.
    per_slice = max(1, int(0.6 * (max_message_bytes // (8 * W))) * world)
    rounds = torch.tensor([(n + per_slice - 1) // per_slice], dtype=torch.int64, device=rows.device)
    dist.all_reduce(rounds, op=dist.ReduceOp.MAX, group=group)
    rounds = max(1, int(rounds.item()))
    extras = list(extras) if extras is not None else None
    out_rids, out_rows, out_extras = [], [], [[] for _ in (extras or [])]
    for r in range(rounds):
        lo0, hi0 = min(n, r * per_slice), min(n, (r + 1) * per_slice)
        biggest = torch.bincount(owner_all[lo0:hi0], minlength=world).max().reshape(1).to(torch.int64) if hi0 > lo0 else torch.zeros(1, dtype=torch.int64, device=rows.device)
        dist.all_reduce(biggest, op=dist.ReduceOp.MAX, group=group)
        parts = max(1, -(-int(biggest.item()) * 8 * W // max_message_bytes))              # ceil: 1 unless the slice is skewed
        for q in range(parts):
            lo = lo0 + (hi0 - lo0) * q // parts
            hi = lo0 + (hi0 - lo0) * (q + 1) // parts
            owner = owner_all[lo:hi]
            perm = torch.argsort(owner, stable=True)
            send_counts = torch.bincount(owner, minlength=world).to(torch.int64)
            recv_counts = torch.empty_like(send_counts)
            dist.all_to_all_single(recv_counts, send_counts, group=group)
            sc, rc = send_counts.tolist(), recv_counts.tolist()
            m = int(sum(rc))
            rids_s = rids[lo:hi][perm].contiguous()
            rows_s = rows[lo:hi][perm].contiguous()
            rids_r = torch.empty(m, dtype=rids.dtype, device=rids.device)
            rows_r = torch.empty((m, W), dtype=rows.dtype, device=rows.device)
            dist.all_to_all_single(rids_r, rids_s, output_split_sizes=rc, input_split_sizes=sc, group=group)
            dist.all_to_all_single(rows_r.view(-1), rows_s.view(-1), output_split_sizes=[c * W for c in rc],
                                   input_split_sizes=[c * W for c in sc], group=group)
            # what left must be what arrived
            sums_s = _checksums(rows_s, sc) + _checksums(rids_s, sc)
            sums_r = torch.empty_like(sums_s)
            dist.all_to_all_single(sums_r, sums_s, group=group)
            if not torch.equal(sums_r, _checksums(rows_r, rc) + _checksums(rids_r, rc)):
                raise RuntimeError(f"minimizer-bucket exchange: slice {r}.{q} did not arrive intact (collective library fault)")
            out_rids.append(rids_r); out_rows.append(rows_r)
            for j, t in enumerate(extras or []):
                t_s = t[lo:hi][perm].contiguous()
                t_r = torch.empty(m, dtype=t.dtype, device=t.device)
                dist.all_to_all_single(t_r, t_s, output_split_sizes=rc, input_split_sizes=sc, group=group)
                cs_s = _checksums(t_s.to(torch.int64), sc)
                cs_r = torch.empty_like(cs_s)
                dist.all_to_all_single(cs_r, cs_s, group=group)
                if not torch.equal(cs_r, _checksums(t_r.to(torch.int64), rc)):
                    raise RuntimeError(f"minimizer-bucket exchange: slice {r}.{q}, extra {j} did not arrive intact (collective library fault)")
                out_extras[j].append(t_r)
    cat = lambda parts: parts[0] if len(parts) == 1 else torch.cat(parts)
    if extras is not None:
        return cat(out_rids), cat(out_rows), [cat(e) for e in out_extras]
    return cat(out_rids), cat(out_rows)
