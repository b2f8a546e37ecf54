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


def exchange_by_bucket(rec_x: torch.Tensor, rids: torch.Tensor, rows: torch.Tensor, group=None):
    """All-to-all of the reads of this rank to the owners of their minimizer buckets.

    rec_x : int64 [n]     minimizer hash of every kept read of this rank
    rids  : int64 [n]     global read ids
    rows  : int64 [n, W]  packed rows
    Returns (rids_recv int64 [m], rows_recv int64 [m, W]) ordered by source rank, then by the sender's order
    (so ascending global rid when every rank holds a contiguous ascending rid range)."""
    world = dist.get_world_size(group)
    owner = bucket_owner(rec_x, world)
    perm = torch.argsort(owner, stable=True)
    send_counts = torch.bincount(owner, minlength=world).to(torch.int64)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    sc, rc = send_counts.tolist(), recv_counts.tolist()
    m = int(sum(rc))
    W = rows.shape[1]
    rids_s = rids[perm].contiguous()
    rows_s = rows[perm].contiguous()
    rids_r = torch.empty(m, dtype=rids.dtype, device=rids.device)
    rows_r = torch.empty((m, W), dtype=rows.dtype, device=rows.device)
    dist.all_to_all_single(rids_r, rids_s, output_split_sizes=rc, input_split_sizes=sc, group=group)
    dist.all_to_all_single(rows_r.view(-1), rows_s.view(-1), output_split_sizes=[c * W for c in rc],
                           input_split_sizes=[c * W for c in sc], group=group)
    return rids_r, rows_r
