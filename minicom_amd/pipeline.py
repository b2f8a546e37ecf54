"""ctypes binding of libmcom_host.so (include/mcom_host.h): the C++ host driver of the hot path."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .hip import McomError, load_library

HERE = os.path.dirname(os.path.abspath(__file__))


class Params(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("k", "e", "m", "w", "cbthr", "max_rounds", "step", "maxthr", "numdict", "host_threads", "maxsearch", "window_scan",
                                         "full_consensus", "full_sketch", "overlap_screen", "host_dump", "stream_sets", "stage2_join", "read_batches")]


def host_lib_path() -> str:
    return os.path.join(HERE, "lib", "libmcom_host.so")


_lib = None


def load_host_library():
    global _lib
    if _lib is not None:
        return _lib
    load_library()                      # libmcom_hip.so first: fails loudly when it has not been built
    p = host_lib_path()
    if not os.path.exists(p):
        raise McomError(f"{p} is missing: run __graft_entry__.build()")
    L = C.CDLL(p)
    vp, sz, i32, cp = C.c_void_p, C.c_size_t, C.c_int, C.c_char_p
    L.mcomh_create.restype = i32
    L.mcomh_create.argtypes = [C.POINTER(vp), i32, vp, vp, vp, sz, sz, i32, C.POINTER(Params)]
    L.mcomh_create_streamed.restype = i32
    L.mcomh_create_streamed.argtypes = [C.POINTER(vp), i32, vp, vp, sz, i32, C.POINTER(Params)]
    L.mcomh_set_records.restype = i32; L.mcomh_set_records.argtypes = [vp, vp, vp]
    L.mcomh_create_packed.restype = i32
    L.mcomh_create_packed.argtypes = [C.POINTER(vp), i32, vp, vp, sz, i32, C.POINTER(Params)]
    L.mcomh_destroy.restype = None; L.mcomh_destroy.argtypes = [vp]
    L.mcomh_last_error.restype = cp; L.mcomh_last_error.argtypes = [vp]
    for f in ("mcomh_kt_for_reads", "mcomh_kt_for_bucket", "mcomh_combine_cluster", "mcomh_update_single", "mcomh_pre_process"):
        getattr(L, f).restype = i32; getattr(L, f).argtypes = [vp]
    L.mcomh_realign_hash.restype = i32; L.mcomh_realign_hash.argtypes = [vp, i32, C.POINTER(C.c_long)]
    L.mcomh_dump_stages.restype = i32; L.mcomh_dump_stages.argtypes = [vp, cp]
    L.mcomh_cluster_dump.restype = i32; L.mcomh_cluster_dump.argtypes = [vp, cp]
    L.mcomh_decompress.restype = i32; L.mcomh_decompress.argtypes = [cp, cp, C.POINTER(C.c_uint64)]
    L.mcomh_cluster_dump_order.restype = i32; L.mcomh_cluster_dump_order.argtypes = [vp, cp]
    L.mcomh_decompress_order.restype = i32; L.mcomh_decompress_order.argtypes = [cp, cp, C.POINTER(C.c_uint64)]
    L.mcomh_cluster_dump_pe.restype = i32; L.mcomh_cluster_dump_pe.argtypes = [vp, cp]
    L.mcomh_decompress_pe.restype = i32; L.mcomh_decompress_pe.argtypes = [cp, cp, cp, C.POINTER(C.c_uint64)]
    L.mcomh_fastq_pair_to_device.restype = i32
    L.mcomh_fastq_pair_to_device.argtypes = [cp, cp, i32, C.POINTER(i32), sz, C.POINTER(vp), C.POINTER(sz), C.c_char_p, sz]
    L.mcomh_n_contigs.restype = sz; L.mcomh_n_contigs.argtypes = [vp]
    L.mcomh_contig_ref.restype = vp; L.mcomh_contig_ref.argtypes = [vp, sz, C.POINTER(sz)]
    L.mcomh_contig_n.restype = sz; L.mcomh_contig_n.argtypes = [vp, sz]
    L.mcomh_contig_members.restype = vp; L.mcomh_contig_members.argtypes = [vp, sz]
    L.mcomh_list.restype = vp; L.mcomh_list.argtypes = [vp, cp, C.POINTER(sz)]
    L.mcomh_contig_set.restype = i32
    L.mcomh_contig_set.argtypes = [vp, C.POINTER(sz), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.mcomh_result_digest.restype = i32; L.mcomh_result_digest.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.mcomh_stat.restype = C.c_double; L.mcomh_stat.argtypes = [vp, cp]
    L.mcomh_prof_enable.restype = i32; L.mcomh_prof_enable.argtypes = [vp, i32]
    L.mcomh_prof_read.restype = i32; L.mcomh_prof_read.argtypes = [vp, cp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.mcomh_prof_kernels.restype = i32; L.mcomh_prof_kernels.argtypes = [vp, cp, C.c_char_p, sz, C.POINTER(sz)]
    L.mcomh_create_from_fastq.restype = i32; L.mcomh_create_from_fastq.argtypes = [C.POINTER(vp), i32, vp, cp, cp, C.POINTER(Params), C.c_char_p, sz]
    L.mcomh_fastq_read.restype = i32; L.mcomh_fastq_read.argtypes = [cp, C.POINTER(i32), vp, sz, C.POINTER(sz)]
    L.mcomh_fastq_to_device.restype = i32
    L.mcomh_fastq_to_device.argtypes = [cp, i32, C.POINTER(i32), sz, C.POINTER(vp), C.POINTER(sz), C.c_char_p, sz]
    L.mcomh_device_free.restype = None; L.mcomh_device_free.argtypes = [vp]
    _lib = L
    return L


HOST_ABI_SYMBOLS = ["mcomh_create", "mcomh_create_streamed", "mcomh_create_packed", "mcomh_create_from_fastq", "mcomh_set_records", "mcomh_destroy", "mcomh_last_error", "mcomh_kt_for_reads", "mcomh_kt_for_bucket",
                    "mcomh_combine_cluster", "mcomh_update_single", "mcomh_realign_hash", "mcomh_stage2", "mcomh_pre_process",
                    "mcomh_dump_stages", "mcomh_cluster_dump", "mcomh_decompress", "mcomh_n_contigs", "mcomh_contig_ref", "mcomh_contig_n", "mcomh_contig_members",
                    "mcomh_list", "mcomh_stat", "mcomh_prof_enable", "mcomh_prof_read", "mcomh_prof_kernels", "mcomh_fastq_read", "mcomh_fastq_to_device",
                    "mcomh_device_free", "mcomh_cluster_dump_order", "mcomh_decompress_order",
                    "mcomh_cluster_dump_pe", "mcomh_decompress_pe", "mcomh_fastq_pair_to_device",
                    "mcomh_contig_set", "mcomh_result_digest",
                    # multi-GPU (bound in minicom_amd/distributed.py)
                    "mcomh_comm_unique_id", "mcomh_comm_create_rccl", "mcomh_comm_create_ops", "mcomh_comm_destroy", "mcomh_comm_rank",
                    "mcomh_comm_world", "mcomh_comm_last_error", "mcomh_comm_alltoallv", "mcomh_comm_allgatherv", "mcomh_comm_allreduce_u64",
                    "mcomh_comm_stats", "mcomh_comm_seconds", "mcomh_create_dist", "mcomh_pool_trim"]


def decompress(folder: str, out_path: str, order: bool = False) -> int:
    """mcomh_decompress(_order): stream files -> one read per line (order=True: the -p file set, original order).
    Returns the number of reads.  Host only."""
    n = C.c_uint64()
    lib = load_host_library()
    rc = (lib.mcomh_decompress_order if order else lib.mcomh_decompress)(folder.encode(), out_path.encode(), C.byref(n))
    if rc:
        raise McomError(f"cannot decode the stream files in {folder}")
    return int(n.value)


def pool_trim():
    """mcomh_pool_trim: pooled device blocks of closed pipelines go back to the runtime."""
    load_host_library().mcomh_pool_trim()


def decompress_pe(folder: str, out_path1: str, out_path2: str) -> int:
    """mcomh_decompress_pe: the paired-end file set -> two files, line i of both is a pair.  Returns the number of pairs."""
    n = C.c_uint64()
    rc = load_host_library().mcomh_decompress_pe(folder.encode(), out_path1.encode(), out_path2.encode(), C.byref(n))
    if rc:
        raise McomError(f"cannot decode the paired-end stream files in {folder}")
    return int(n.value)


def read_fastq(path: str, L: int = 0) -> np.ndarray:
    """mcomh_fastq_read: the reads of a FASTQ/FASTA file (plain or .gz) as a numpy uint8 [n, L] array.  Host only."""
    lib = load_host_library()
    Lc, n = C.c_int(L), C.c_size_t()
    rc = lib.mcomh_fastq_read(path.encode(), C.byref(Lc), None, C.c_size_t(-1).value, C.byref(n))      # count + check
    if rc:
        raise McomError(f"{path}: not a FASTQ/FASTA file of equal-length reads ({rc})")
    out = np.empty((n.value, Lc.value), dtype=np.uint8)
    rc = lib.mcomh_fastq_read(path.encode(), C.byref(Lc), out.ctypes.data_as(C.c_void_p), n.value, C.byref(n))
    if rc:
        raise McomError(f"{path}: read failed ({rc})")
    return out


class Pipeline:
    """Stage 1 + Stage 2 of minicom on one GPU.  reads: numpy uint8 [n, L] (host) or a torch uint8 CUDA tensor [n, pitch]."""

    def __init__(self, reads, L: int | None = None, device: int = 0, stream=None, packed: bool = False, records=None, **params):
        """records (packed=True only): (x int64 [n], ylow int32 [n]) CUDA tensors, the minimizers the rows were sketched to with
        this pipeline's k on the rank that sent them: kt_for_reads then assembles the records instead of sketching again."""
        self.lib = load_host_library()
        p = Params(**{k: int(v) for k, v in params.items()})
        h = C.c_void_p()
        s = C.c_void_p(stream.cuda_stream) if stream is not None else C.c_void_p(0)
        if packed:
            # int64 CUDA tensor [n, W] of packed rows (include/mcom.h format)
            assert reads.is_cuda and reads.is_contiguous() and L is not None and reads.shape[1] == (2 * L + 63) // 64
            n = int(reads.shape[0])
            self._keep = reads
            rc = self.lib.mcomh_create_packed(C.byref(h), device, s, C.c_void_p(reads.data_ptr()), n, L, C.byref(p))
        elif isinstance(reads, np.ndarray):
            reads = np.ascontiguousarray(reads, dtype=np.uint8)
            n, L = reads.shape
            self._keep = reads
            rc = self.lib.mcomh_create(C.byref(h), device, s, reads.ctypes.data_as(C.c_void_p), None, L, n, L, C.byref(p))
        else:
            assert reads.is_cuda and reads.is_contiguous() and L is not None
            n, pitch = reads.shape
            self._keep = reads
            rc = self.lib.mcomh_create(C.byref(h), device, s, None, C.c_void_p(reads.data_ptr()), pitch, n, L, C.byref(p))
        if rc:
            raise McomError(f"mcomh_create failed ({rc}): no usable GPU or bad arguments; there is no CPU fallback")
        self._h = h
        self.n, self.L = n, L
        if records is not None:
            assert packed, "records travel with packed rows"
            x, ylow = records
            import torch
            assert x.is_cuda and ylow.is_cuda and x.is_contiguous() and ylow.is_contiguous() and x.dtype == torch.int64 and ylow.dtype == torch.int32
            assert int(x.shape[0]) == n and int(ylow.shape[0]) == n
            self._keep = (self._keep, x, ylow)
            self._check(self.lib.mcomh_set_records(self._h, C.c_void_p(x.data_ptr()), C.c_void_p(ylow.data_ptr())))

    @classmethod
    def from_host_streamed(cls, reads, device: int = 0, **params):
        """mcomh_create_streamed: reads = torch uint8 CPU tensor (ideally pinned) or numpy array [n, L], NOT copied on the
        host; kt_for_reads uploads it chunk by chunk beside the classification kernels.  Keep `reads` alive."""
        lib = load_host_library()
        n, L = int(reads.shape[0]), int(reads.shape[1])
        ptr = reads.ctypes.data if isinstance(reads, np.ndarray) else reads.data_ptr()
        self = cls.__new__(cls)
        self.lib = lib
        p = Params(**{k: int(v) for k, v in params.items()})
        h = C.c_void_p()
        rc = lib.mcomh_create_streamed(C.byref(h), device, C.c_void_p(0), C.c_void_p(ptr), n, L, C.byref(p))
        if rc:
            raise McomError(f"mcomh_create_streamed failed ({rc})")
        self._h, self._keep = h, reads
        self.n, self.L = n, L
        return self

    @classmethod
    def from_fastq(cls, path: str, L: int = 0, device: int = 0, chunk_reads: int = 0, path2: str | None = None, **params):
        """FASTQ/FASTA (plain or .gz) -> HBM through two pinned chunks (mcomh_fastq_to_device) -> pipeline.
        path2: the mates' file (paired end): its reads follow those of the first file."""
        lib = load_host_library()
        err = C.create_string_buffer(256)
        self = cls.__new__(cls)
        self.lib = lib
        p = Params(**{k: int(v) for k, v in params.items()})
        h = C.c_void_p()
        rc = lib.mcomh_create_from_fastq(C.byref(h), device, C.c_void_p(0), path.encode(), path2.encode() if path2 else None, C.byref(p), err, 256)
        if rc:
            raise McomError(f"{path}: {err.value.decode() or rc}")
        self._h, self._dev_reads, self._keep = h, None, None
        self.n, self.L = int(lib.mcomh_stat(h, b"n")), int(lib.mcomh_stat(h, b"L"))
        if L and L != self.L:
            self.close()
            raise McomError(f"{path}: reads of {self.L} bases, {L} expected")
        return self

    def _check(self, rc):
        if rc:
            raise McomError(f"mcom host error {rc}: {self.lib.mcomh_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None):
            self.lib.mcomh_destroy(self._h)
            self._h = None
        if getattr(self, "_dev_reads", None):
            self.lib.mcomh_device_free(self._dev_reads)
            self._dev_reads = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def kt_for_reads(self): self._check(self.lib.mcomh_kt_for_reads(self._h))
    def kt_for_bucket(self): self._check(self.lib.mcomh_kt_for_bucket(self._h))
    def combine_cluster(self): self._check(self.lib.mcomh_combine_cluster(self._h))
    def update_single(self): self._check(self.lib.mcomh_update_single(self._h))
    def pre_process(self): self._check(self.lib.mcomh_pre_process(self._h))
    def stage2(self): self._check(self.lib.mcomh_stage2(self._h))

    def realign_hash(self, thr: int) -> int:
        cr = C.c_long()
        self._check(self.lib.mcomh_realign_hash(self._h, thr, C.byref(cr)))
        return cr.value

    def dump_stages(self, path: str): self._check(self.lib.mcomh_dump_stages(self._h, path.encode()))

    def cluster_dump(self, folder: str, order: bool = False, paired: bool = False):
        """Writes the reference's pre-bsc stream files (cluster_dump at one thread) into an existing directory;
        order=True: the order-preserving file set of minicom -p; paired=True: the paired-end file set (rows [0, n/2) are
        the first file, rows [n/2, n) their mates)."""
        fn = self.lib.mcomh_cluster_dump_pe if paired else self.lib.mcomh_cluster_dump_order if order else self.lib.mcomh_cluster_dump
        self._check(fn(self._h, folder.encode()))

    def prof_enable(self, on: bool = True): self._check(self.lib.mcomh_prof_enable(self._h, 1 if on else 0))

    def prof_read(self, name: str):
        """(device milliseconds, launches) of one kernel class, measured with HIP events on the launch stream."""
        ms = C.c_double(); n = C.c_uint64()
        self._check(self.lib.mcomh_prof_read(self._h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def prof_kernels(self, name: str = "*") -> dict:
        """{kernel name: launches} of one kernel class ("*": every kernel of the library) while the profiler was on -- the
        compiler's spelling of each instantiation, which is the name rocprofv3 prints"""
        need = C.c_size_t()
        self._check(self.lib.mcomh_prof_kernels(self._h, name.encode(), None, 0, C.byref(need)))
        buf = C.create_string_buffer(need.value + 1)
        self._check(self.lib.mcomh_prof_kernels(self._h, name.encode(), buf, need.value + 1, None))
        out = {}
        for ln in buf.value.decode().splitlines():
            k, _, c = ln.rpartition("\t")
            out[k] = int(c)
        return out

    def stat(self, name: str) -> float:
        return float(self.lib.mcomh_stat(self._h, name.encode()))

    def id_list(self, name: str) -> np.ndarray:
        n = C.c_size_t()
        ptr = self.lib.mcomh_list(self._h, name.encode(), C.byref(n))
        if not n.value:
            return np.zeros(0, dtype=np.uint32)
        return np.frombuffer((C.c_char * (4 * n.value)).from_address(ptr), dtype=np.uint32).copy()

    def contig_set(self):
        """(ref uint8 [chars], ref_off uint64 [n + 1], mem uint64 [members], mem_off uint64 [n + 1]): the whole contig set as
        flat numpy arrays (copies), mcomh_contig_set."""
        n = C.c_size_t(); ref, ro, mem, mo = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._check(self.lib.mcomh_contig_set(self._h, C.byref(n), C.byref(ref), C.byref(ro), C.byref(mem), C.byref(mo)))
        nc = n.value

        def arr(ptr, dtype, count):
            if not count:
                return np.zeros(0, dtype=dtype)
            return np.frombuffer((C.c_char * (np.dtype(dtype).itemsize * count)).from_address(ptr.value), dtype=dtype).copy()
        roff = arr(ro, np.uint64, nc + 1) if nc else np.zeros(1, np.uint64)
        moff = arr(mo, np.uint64, nc + 1) if nc else np.zeros(1, np.uint64)
        return arr(ref, np.uint8, int(roff[-1])), roff, arr(mem, np.uint64, int(moff[-1])), moff

    def result_digest(self):
        """mcomh_result_digest: eight numbers that two runs over the same reads must share."""
        out = (C.c_uint64 * 8)()
        self._check(self.lib.mcomh_result_digest(self._h, out))
        return [int(v) for v in out]

    def contigs(self):
        out = []
        for i in range(self.lib.mcomh_n_contigs(self._h)):
            n = self.lib.mcomh_contig_n(self._h, i)
            mem = np.frombuffer((C.c_char * (8 * n)).from_address(self.lib.mcomh_contig_members(self._h, i)), dtype=np.uint64).copy() if n else np.zeros(0, np.uint64)
            ln = C.c_size_t()
            ptr = self.lib.mcomh_contig_ref(self._h, i, C.byref(ln))
            out.append((C.string_at(ptr, ln.value), mem))
        return out
