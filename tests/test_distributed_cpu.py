"""N > 1 path on CPU: world_size 2 (and 3) over gloo.  The multi-GPU path of libmcom_host.so moves everything through ONE
primitive, a byte-wise all-to-all (include/mcom_host.h); here that primitive and what the library builds on it
(all-gather of ragged parts, reductions of counters) run between real processes through the callback transport
(torch.distributed / gloo), on host memory -- no GPU involved.  The pipeline stages themselves need the GPU:
tests/test_gpu_distributed.py runs them with several ranks on one card and compares with the single-GPU result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _part(src, dst, world):
    """What rank src sends to rank dst: ragged, empty for some pairs, content a function of (src, dst)."""
    n = (src * 7 + dst * 13 + 5) % 11
    if (src + dst) % 4 == 3:
        n = 0
    return (np.arange(n * 37, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(src * 1000 + dst)).view(np.uint8)


def _worker(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from minicom_amd.distributed import Comm
    c = Comm.torch()
    assert (c.rank, c.world) == (rank, world)
    # the primitive: every pair its own size
    send = [_part(rank, q, world) for q in range(world)]
    want = [_part(q, rank, world) for q in range(world)]
    got = c.alltoallv(np.concatenate(send), [len(s) for s in send], [len(w) for w in want])
    assert np.array_equal(got, np.concatenate(want))
    # all-gather of ragged parts (how new contigs, singleton lists and candidate pairs are replicated)
    mine = _part(rank, rank, world)
    sizes = [len(_part(q, q, world)) for q in range(world)]
    got = c.allgatherv(mine, sizes)
    assert np.array_equal(got, np.concatenate([_part(q, q, world) for q in range(world)]))
    # counters: sums (clustered reads of a round), minima / maxima (agreement checks)
    v = np.array([rank + 1, 10 * rank, 2 ** 40 + rank], dtype=np.uint64)
    assert c.allreduce(v, "sum").tolist() == [world * (world + 1) // 2, 10 * world * (world - 1) // 2, world * 2 ** 40 + world * (world - 1) // 2]
    assert c.allreduce(v, "min").tolist() == [1, 0, 2 ** 40]
    assert c.allreduce(v, "max").tolist() == [world, 10 * (world - 1), 2 ** 40 + world - 1]
    # the MIN-reduction of Stage-2 claim keys as the library does it (reduce-scatter by all-to-all, then all-gather),
    # restated on the host forms of the two collectives
    n = 1001
    rng = np.random.default_rng(77)
    claims = rng.integers(0, 2 ** 62, size=(world, n), dtype=np.uint64)
    claims[rng.random((world, n)) < 0.7] = np.uint64(2 ** 64 - 1)                   # most tuples are not seen by a given rank
    lo = [n * q // world for q in range(world + 1)]
    mine = claims[rank]
    parts = c.alltoallv(mine.view(np.uint8), [8 * (lo[q + 1] - lo[q]) for q in range(world)], [8 * (lo[rank + 1] - lo[rank])] * world)
    folded = parts.view(np.uint64).reshape(world, -1).min(axis=0)
    out = c.allgatherv(folded.view(np.uint8), [8 * (lo[q + 1] - lo[q]) for q in range(world)]).view(np.uint64)
    assert np.array_equal(out, claims.min(axis=0))
    sent, calls = c.stats()
    assert calls >= 6 and sent > 0
    c.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_library_collectives_between_processes_over_gloo(world):
    mp.spawn(_worker, args=(world, _free_port()), nprocs=world, join=True)


def _collective_checks(c, rank, world):
    send = [_part(rank, q, world) for q in range(world)]
    want = [_part(q, rank, world) for q in range(world)]
    got = c.alltoallv(np.concatenate(send), [len(s) for s in send], [len(w) for w in want])
    assert np.array_equal(got, np.concatenate(want))
    got = c.allgatherv(_part(rank, rank, world), [len(_part(q, q, world)) for q in range(world)])
    assert np.array_equal(got, np.concatenate([_part(q, q, world) for q in range(world)]))
    v = np.array([rank + 1, 2 ** 40 + rank], dtype=np.uint64)
    assert c.allreduce(v, "sum").tolist() == [world * (world + 1) // 2, world * 2 ** 40 + world * (world - 1) // 2]
    assert c.allreduce(v, "max").tolist() == [world, 2 ** 40 + world - 1]


@pytest.mark.parametrize("world,serialize", [(4, False), (8, True)])
def test_library_collectives_between_threads_of_one_process(world, serialize):
    """Comm.threads: the transport of tools/dist_work.py (more ranks on one card than processes are allowed there); with
    serialize a rank computes only while it holds the hub's lock and gives it up inside an exchange."""
    import threading
    sys.path.insert(0, ROOT)
    from minicom_amd.distributed import Comm
    comms, hub = Comm.threads(world, serialize=serialize)
    errors = []

    def run(rank):
        hub.enter()
        try:
            _collective_checks(comms[rank], rank, world)
        except Exception as e:                                                  # noqa: BLE001
            errors.append((rank, repr(e))); hub.abort()
        finally:
            hub.leave()
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]; [t.join(timeout=60) for t in ts]
    assert not errors, errors
    want = sum(len(_part(s, d, world)) for s in range(world) for d in range(world) if s != d)
    assert int(hub.bytes.sum()) - int(np.trace(hub.bytes)) >= want
    assert all(c.seconds() > 0 for c in comms)
    for c in comms:
        c.close()


def test_communicator_rejects_bad_arguments():
    """No GPU, no peers: argument checks of the C entry points."""
    import ctypes as C
    from minicom_amd.distributed import _lib
    L = _lib()
    h = C.c_void_p()
    assert L.mcomh_comm_create_ops(C.byref(h), 0, 1, None, None) == -1
    assert L.mcomh_comm_create_rccl(C.byref(h), 2, 2, None, 0) == -1
    assert L.mcomh_comm_rank(None) == -1 and L.mcomh_comm_world(None) == 0
    assert L.mcomh_comm_last_error(None) == b"null communicator"
    assert L.mcomh_create_dist(C.byref(h), 0, None, None, None, None, 0, 0, 0, 0, 100, None) == -1
