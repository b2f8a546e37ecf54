"""ctypes binding of the multi-GPU entry points of libmcom_host.so (include/mcom_host.h, "multi-GPU").

One process per GPU.  The product code -- the exchange of minimizer records to the owners of their buckets every bucket
round, the all-gathers of the replicated contig set, the MIN-reduction of the Stage-2 claim keys -- is C++
(minicom_amd/host/mcom_comm.cpp, mcom_pipeline.cpp) over ONE primitive, a byte-wise all-to-all, with two transports:

  Comm.rccl(...)   ncclSend / ncclRecv groups over xGMI (every peer pair on its own link; no ring): production, bench.py
  Comm.torch(...)  the all-to-all handed to the library as a callback over torch.distributed (gloo): what the tests use to
                   run several ranks on one GPU, or on no GPU at all for the host-side collectives

A DistPipeline gives, on every rank, exactly the result Pipeline gives on one GPU over all reads (tests/test_gpu_distributed.py).
"""
from __future__ import annotations

import ctypes as C
import traceback

import numpy as np

from .hip import McomError
from .pipeline import Params, Pipeline, load_host_library

UNIQUE_ID_BYTES = 128
_u64p = C.POINTER(C.c_uint64)
_ALLTOALLV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, _u64p, _u64p, C.c_void_p, _u64p, _u64p)


class _Ops(C.Structure):
    _fields_ = [("alltoallv", _ALLTOALLV)]


_bound = False


def _lib():
    global _bound
    L = load_host_library()
    if not _bound:
        vp, i32, sz = C.c_void_p, C.c_int, C.c_size_t
        L.mcomh_comm_unique_id.restype = i32; L.mcomh_comm_unique_id.argtypes = [vp]
        L.mcomh_comm_create_rccl.restype = i32; L.mcomh_comm_create_rccl.argtypes = [C.POINTER(vp), i32, i32, vp, i32]
        L.mcomh_comm_create_ops.restype = i32; L.mcomh_comm_create_ops.argtypes = [C.POINTER(vp), i32, i32, C.POINTER(_Ops), vp]
        L.mcomh_comm_destroy.restype = None; L.mcomh_comm_destroy.argtypes = [vp]
        L.mcomh_comm_rank.restype = i32; L.mcomh_comm_rank.argtypes = [vp]
        L.mcomh_comm_world.restype = i32; L.mcomh_comm_world.argtypes = [vp]
        L.mcomh_comm_last_error.restype = C.c_char_p; L.mcomh_comm_last_error.argtypes = [vp]
        L.mcomh_comm_alltoallv.restype = i32; L.mcomh_comm_alltoallv.argtypes = [vp, vp, _u64p, _u64p, vp, _u64p, _u64p, i32, vp]
        L.mcomh_comm_allgatherv.restype = i32; L.mcomh_comm_allgatherv.argtypes = [vp, vp, vp, _u64p, _u64p, i32, vp]
        L.mcomh_comm_allreduce_u64.restype = i32; L.mcomh_comm_allreduce_u64.argtypes = [vp, _u64p, sz, i32]
        L.mcomh_comm_stats.restype = None; L.mcomh_comm_stats.argtypes = [vp, _u64p, _u64p]
        L.mcomh_comm_seconds.restype = C.c_double; L.mcomh_comm_seconds.argtypes = [vp]
        L.mcomh_create_dist.restype = i32
        L.mcomh_create_dist.argtypes = [C.POINTER(vp), i32, vp, vp, vp, vp, sz, sz, C.c_uint64, C.c_uint64, i32, C.POINTER(Params)]
        _bound = True
    return L


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(_u64p)


class _ThreadHub:
    """The meeting point of Comm.threads: every rank publishes where its parts lie, all wait, every rank copies what is meant
    for it, all wait again (the parts are the senders' own buffers)."""

    def __init__(self, world: int, serialize: bool):
        import threading
        self.world = world
        self.barrier = threading.Barrier(world)
        self.lock = threading.Lock() if serialize else None
        self.parts = [None] * world
        self.bytes = np.zeros((world, world), dtype=np.uint64)            # [sender, receiver]

    def enter(self):
        """a rank starts computing (serialize: waits for its turn)"""
        if self.lock:
            self.lock.acquire()

    def leave(self):
        if self.lock:
            self.lock.release()

    def abort(self):
        self.barrier.abort()

    def make_callback(self, rank: int):
        world = self.world

        def alltoallv(_user, send, so, sb, recv, ro, rb):
            try:
                self.leave()
                try:
                    self.parts[rank] = [(send + int(so[q]) if sb[q] else 0, int(sb[q])) for q in range(world)]
                    self.barrier.wait()
                    for q in range(world):
                        addr, nbytes = self.parts[q][rank]
                        if nbytes != int(rb[q]):
                            raise RuntimeError(f"rank {q} sends {nbytes} bytes to rank {rank}, which expects {int(rb[q])}")
                        if nbytes:
                            C.memmove(recv + int(ro[q]), addr, nbytes)
                            self.bytes[q, rank] += nbytes
                    self.barrier.wait()
                finally:
                    self.enter()
                return 0
            except Exception:                                                    # never let an exception cross the C boundary
                traceback.print_exc()
                return 1
        return alltoallv


class Comm:
    """A communicator of the library (mcomh_comm)."""

    def __init__(self, handle, rank, world, keep=None):
        self._h, self.rank, self.world, self._keep = handle, rank, world, keep

    @staticmethod
    def unique_id() -> bytes:
        """ncclGetUniqueId: made by rank 0, handed to the other ranks by whatever means the job has."""
        buf = C.create_string_buffer(UNIQUE_ID_BYTES)
        if _lib().mcomh_comm_unique_id(buf):
            raise McomError("mcomh_comm_unique_id failed: no RCCL")
        return buf.raw

    @classmethod
    def rccl(cls, rank: int, world: int, unique_id: bytes, device: int):
        h = C.c_void_p()
        if _lib().mcomh_comm_create_rccl(C.byref(h), rank, world, C.c_char_p(unique_id), device):
            raise McomError("mcomh_comm_create_rccl failed (see stderr)")
        return cls(h, rank, world)

    @classmethod
    def torch(cls, group=None):
        """The all-to-all over torch.distributed (any backend that moves CPU uint8 tensors: gloo)."""
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)

        def alltoallv(_user, send, so, sb, recv, ro, rb):
            try:
                sbl = [int(sb[q]) for q in range(world)]; rbl = [int(rb[q]) for q in range(world)]
                parts = [np.frombuffer((C.c_uint8 * sbl[q]).from_address(send + int(so[q])), dtype=np.uint8) for q in range(world) if sbl[q]]
                inp = torch.from_numpy(np.concatenate(parts)) if parts else torch.empty(0, dtype=torch.uint8)
                out = torch.empty(sum(rbl), dtype=torch.uint8)
                dist.all_to_all_single(out, inp, output_split_sizes=rbl, input_split_sizes=sbl, group=group)
                o, at = out.numpy(), 0
                for q in range(world):
                    if rbl[q]:
                        C.memmove(recv + int(ro[q]), o[at:at + rbl[q]].ctypes.data, rbl[q])
                        at += rbl[q]
                return 0
            except Exception:                                                    # never let an exception cross the C boundary
                traceback.print_exc()
                return 1

        cb = _ALLTOALLV(alltoallv)
        ops = _Ops(cb)
        h = C.c_void_p()
        if _lib().mcomh_comm_create_ops(C.byref(h), rank, world, C.byref(ops), None):
            raise McomError("mcomh_comm_create_ops failed")
        return cls(h, rank, world, keep=(cb, ops))

    @classmethod
    def threads(cls, world: int, serialize: bool = False):
        """`world` communicators for `world` threads of THIS process (one pipeline per thread): the all-to-all goes through
        process memory.  For measurements and tests that want more ranks on one GPU than processes are allowed there.
        serialize: a lock lets one rank compute at a time (it is given up while a rank waits in an exchange), so that kernel
        and stage times of a rank are not disturbed by the others sharing the card.  Returns (comms, hub)."""
        hub = _ThreadHub(world, serialize)
        comms = []
        for rank in range(world):
            cb = _ALLTOALLV(hub.make_callback(rank))
            ops = _Ops(cb)
            h = C.c_void_p()
            if _lib().mcomh_comm_create_ops(C.byref(h), rank, world, C.byref(ops), None):
                raise McomError("mcomh_comm_create_ops failed")
            comms.append(cls(h, rank, world, keep=(cb, ops, hub)))
        return comms, hub

    def seconds(self) -> float:
        """wall seconds spent inside all-to-all calls so far"""
        return float(_lib().mcomh_comm_seconds(self._h))

    def _check(self, rc):
        if rc:
            raise McomError(f"communicator error {rc}: {_lib().mcomh_comm_last_error(self._h).decode()}")

    # host-side forms of the collectives (tests; numpy uint8 / uint64 arrays)
    def alltoallv(self, send: np.ndarray, send_bytes, recv_bytes) -> np.ndarray:
        send = np.ascontiguousarray(send, dtype=np.uint8)
        sb, sbp = _u64(send_bytes); rb, rbp = _u64(recv_bytes)
        so, sop = _u64(np.concatenate(([0], np.cumsum(sb)[:-1]))); ro, rop = _u64(np.concatenate(([0], np.cumsum(rb)[:-1])))
        recv = np.zeros(int(rb.sum()), dtype=np.uint8)
        self._check(_lib().mcomh_comm_alltoallv(self._h, send.ctypes.data_as(C.c_void_p), sop, sbp, recv.ctypes.data_as(C.c_void_p), rop, rbp, 0, None))
        return recv

    def allgatherv(self, mine: np.ndarray, sizes) -> np.ndarray:
        mine = np.ascontiguousarray(mine, dtype=np.uint8)
        b, bp = _u64(sizes)
        off, offp = _u64(np.concatenate(([0], np.cumsum(b)[:-1])))
        buf = np.zeros(int(b.sum()), dtype=np.uint8)
        self._check(_lib().mcomh_comm_allgatherv(self._h, mine.ctypes.data_as(C.c_void_p), buf.ctypes.data_as(C.c_void_p), offp, bp, 0, None))
        return buf

    def allreduce(self, vals, op: str = "sum") -> np.ndarray:
        v, vp = _u64(np.array(vals, dtype=np.uint64, copy=True))
        self._check(_lib().mcomh_comm_allreduce_u64(self._h, vp, v.size, {"sum": 0, "min": 1, "max": 2}[op]))
        return v

    def stats(self):
        b, c = C.c_uint64(), C.c_uint64()
        _lib().mcomh_comm_stats(self._h, C.byref(b), C.byref(c))
        return int(b.value), int(c.value)

    def close(self):
        if getattr(self, "_h", None):
            _lib().mcomh_comm_destroy(self._h)
            self._h = None


class DistPipeline(Pipeline):
    """Stage 1 + Stage 2 over the GPUs of one node: this rank's shard of the reads in, the complete result on every rank.

    reads: numpy uint8 [n_local, L] (host) or a torch uint8 CUDA tensor [n_local, pitch]: reads rid0 .. rid0 + n_local - 1
    of n_total (shards contiguous, in rank order)."""

    def __init__(self, reads, rid0: int, n_total: int, comm: Comm, L: int | None = None, device: int = 0, stream=None, **params):
        self.lib = _lib()
        p = Params(**{k: int(v) for k, v in params.items()})
        h = C.c_void_p()
        s = C.c_void_p(stream.cuda_stream) if stream is not None else C.c_void_p(0)
        self._keep = (reads, comm)
        if isinstance(reads, np.ndarray):
            reads = np.ascontiguousarray(reads, dtype=np.uint8)
            n, L = reads.shape if reads.ndim == 2 and reads.shape[0] else (0, L)
            assert L is not None
            self._keep = (reads, comm)
            rc = self.lib.mcomh_create_dist(C.byref(h), device, s, comm._h, reads.ctypes.data_as(C.c_void_p) if n else None, None, L, n, rid0, n_total, L, C.byref(p))
        else:
            assert reads.is_cuda and reads.is_contiguous() and L is not None
            n, pitch = reads.shape
            rc = self.lib.mcomh_create_dist(C.byref(h), device, s, comm._h, None, C.c_void_p(reads.data_ptr()), pitch, n, rid0, n_total, L, C.byref(p))
        if rc:
            raise McomError(f"mcomh_create_dist failed ({rc}): no usable GPU, bad arguments or shards that do not tile [0, n_total)")
        self._h = h
        self.n, self.L, self.n_local, self.rid0 = n_total, L, n, rid0
