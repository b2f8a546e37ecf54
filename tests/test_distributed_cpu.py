"""N > 1 path on CPU: world_size 2 over gloo.  The minimizer-bucket all-to-all (minicom_amd/distributed.py) is
device agnostic; here its inputs come from the CPU oracle so the exchange logic is covered without a GPU."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, n, L, k, out_dir, max_bytes=None):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from minicom_amd import synth
    from minicom_amd.distributed import exchange_by_bucket, bucket_owner
    from minicom_amd.hip import pack_nt4
    per = n // world
    first = rank * per
    reads = synth.synth_reads(4321, n, L, first=first, count=per)
    rec = oracle.sketch_two_batch(reads, k, rid0=first)
    rows = torch.from_numpy(pack_nt4(reads).view(np.int64))
    x = torch.from_numpy(rec["x"].view(np.int64).copy())
    rids = torch.arange(first, first + per, dtype=torch.int64)
    ylow = torch.from_numpy((rec["y"] & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32).copy())
    kw = {} if max_bytes is None else {"max_message_bytes": max_bytes}
    rids_r, rows_r, (x_r, ylow_r) = exchange_by_bucket(x, rids, rows, extras=[x, ylow], **kw)
    assert rids_r.shape == x_r.shape == ylow_r.shape
    # every received read belongs to a bucket this rank owns; one slice: ascending global rid; several slices (messages
    # capped at max_bytes): slice after slice, ascending inside a (slice, source rank) run
    all_reads = synth.synth_reads(4321, n, L)
    rec_all = oracle.sketch_two_batch(all_reads, k)
    own = bucket_owner(torch.from_numpy(rec_all["x"].view(np.int64).copy()), world).numpy()
    want = np.flatnonzero(own == rank)
    got = rids_r.numpy()
    if max_bytes is None:
        assert np.array_equal(got, want)
    else:
        assert np.array_equal(np.sort(got), want) and not np.array_equal(got, want)
    assert np.array_equal(rows_r.numpy().view(np.uint64), pack_nt4(all_reads[got]))
    # the minimizers that travelled are those of the reads they arrived with
    assert np.array_equal(x_r.numpy().view(np.uint64), rec_all["x"][got])
    assert np.array_equal(ylow_r.numpy().view(np.uint32), (rec_all["y"][got] & np.uint64(0xFFFFFFFF)).astype(np.uint32))
    np.save(os.path.join(out_dir, f"rids_{rank}.npy"), rids_r.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_exchange_world_size_2_gloo(tmp_path):
    n, L, k, world = 4000, 100, 31, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, L, k, str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"rids_{r}.npy") for r in range(world)])
    assert np.array_equal(np.sort(got), np.arange(n))          # a partition of all reads: nothing lost, nothing doubled


def test_bucket_exchange_in_slices_world_size_2_gloo(tmp_path):
    """Messages capped at 4 kB: dozens of verified slices instead of one (the cap exists because a single message
    above ~1 GiB arrives half on the GPU stack, minicom_amd/distributed.py)."""
    n, L, k, world = 4000, 100, 31, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, L, k, str(tmp_path), 4000), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"rids_{r}.npy") for r in range(world)])
    assert np.array_equal(np.sort(got), np.arange(n))


def test_reads_sharing_a_minimizer_land_on_one_rank():
    sys.path.insert(0, ROOT)
    import oracle
    from minicom_amd import synth
    from minicom_amd.distributed import bucket_owner
    reads = synth.synth_reads(99, 3000, 150)
    rec = oracle.sketch_two_batch(reads, 31)
    for world in (2, 4, 8):
        own = bucket_owner(torch.from_numpy(rec["x"].view(np.int64).copy()), world).numpy()
        by_x = {}
        for x, o in zip(rec["x"].tolist(), own.tolist()):
            assert by_x.setdefault(x, o) == o
        assert len(set(own.tolist())) == world


def _skew_worker(rank, world, port, n, W, cap):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from minicom_amd.distributed import exchange_by_bucket
    g = torch.Generator().manual_seed(7 + rank)
    rows = torch.randint(-(1 << 62), 1 << 62, (n, W), dtype=torch.int64, generator=g)
    rids = torch.arange(rank * n, (rank + 1) * n, dtype=torch.int64)
    x = torch.zeros(n, dtype=torch.int64)                                     # every read belongs to bucket 0: rank 0 owns them all
    rids_r, rows_r = exchange_by_bucket(x, rids, rows, max_message_bytes=cap)
    if rank == 0:
        assert rids_r.shape[0] == world * n and torch.equal(torch.sort(rids_r).values, torch.arange(world * n))
        mine = rids_r < n
        assert torch.equal(rows_r[mine][torch.argsort(rids_r[mine])], rows)
    else:
        assert rids_r.shape[0] == 0 and rows_r.shape[0] == 0
    dist.barrier()
    dist.destroy_process_group()


def test_skewed_exchange_splits_its_slices_world_size_2_gloo():
    """All reads owned by one rank: a slice sized for an even spread would send a message of twice the cap; the ranks
    agree on the largest message first and cut the slice."""
    port = _free_port()
    mp.spawn(_skew_worker, args=(2, port, 3000, 4, 8000), nprocs=2, join=True)
