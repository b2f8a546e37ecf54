"""ctypes binding of oracle/libmcom_oracle.so -- TEST INFRASTRUCTURE ONLY.

Importable only from tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke().
The product package (minicom_amd) never imports this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmcom_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "mcom_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "-B" if force else "-s", "libmcom_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return LIB_PATH


class MM128(C.Structure):
    _fields_ = [("x", C.c_uint64), ("y", C.c_uint64)]


class Params(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("k", "e", "m", "w", "cbthr", "max_rounds", "step", "maxthr", "numdict")]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    u64, u32, i32, sz, vp, cp = C.c_uint64, C.c_uint32, C.c_int, C.c_size_t, C.c_void_p, C.c_char_p
    L.mcomo_hash64.restype = u64; L.mcomo_hash64.argtypes = [u64, u64]
    L.mcomo_sketch_two.restype = None; L.mcomo_sketch_two.argtypes = [cp, i32, i32, u32, C.POINTER(MM128)]
    L.mcomo_sketch_lh_ori.restype = sz; L.mcomo_sketch_lh_ori.argtypes = [cp, i32, i32, i32, u32, vp, sz]
    L.mcomo_sketch_two_batch.restype = None; L.mcomo_sketch_two_batch.argtypes = [vp, sz, i32, i32, u32, vp]
    L.mcomo_process_reads_batch.restype = None
    L.mcomo_process_reads_batch.argtypes = [vp, sz, i32, i32, i32, u32, vp, vp, vp]
    L.mcomo_radix_sort_128x.restype = None; L.mcomo_radix_sort_128x.argtypes = [vp, vp]
    L.mcomo_match_pro.restype = i32; L.mcomo_match_pro.argtypes = [cp, cp, i32, i32]
    L.mcomo_encode_byte.restype = i32; L.mcomo_encode_byte.argtypes = [cp, cp, i32, i32, i32]
    L.mcomo_string_to_bits.restype = None; L.mcomo_string_to_bits.argtypes = [cp, i32, vp]
    L.mcomo_dict_layout.restype = i32; L.mcomo_dict_layout.argtypes = [i32, i32, vp, vp]
    L.mcomo_new.restype = vp; L.mcomo_new.argtypes = [vp, sz, i32, C.POINTER(Params)]
    L.mcomo_free.restype = None; L.mcomo_free.argtypes = [vp]
    L.mcomo_force_maxsearch.restype = None; L.mcomo_force_maxsearch.argtypes = [vp, i32]
    for f in ("mcomo_stage_reads", "mcomo_stage_bucket", "mcomo_stage_combine", "mcomo_update_single", "mcomo_run_all"):
        getattr(L, f).restype = None; getattr(L, f).argtypes = [vp]
    L.mcomo_stage_realign_pass.restype = C.c_long; L.mcomo_stage_realign_pass.argtypes = [vp, i32]
    L.mcomo_dump_stages.restype = i32; L.mcomo_dump_stages.argtypes = [vp, cp]
    for f in ("mcomo_n_reads", "mcomo_n_sg", "mcomo_n_contigs"):
        getattr(L, f).restype = sz; getattr(L, f).argtypes = [vp]
    for f in ("mcomo_seq", "mcomo_cls", "mcomo_rec0", "mcomo_sg", "mcomo_sg_flag"):
        getattr(L, f).restype = vp; getattr(L, f).argtypes = [vp]
    L.mcomo_contig_ref.restype = cp; L.mcomo_contig_ref.argtypes = [vp, sz]
    L.mcomo_contig_n.restype = sz; L.mcomo_contig_n.argtypes = [vp, sz]
    L.mcomo_contig_members.restype = vp; L.mcomo_contig_members.argtypes = [vp, sz]
    L.mcomo_counter.restype = sz; L.mcomo_counter.argtypes = [vp, cp]
    L.mcomo_list.restype = vp; L.mcomo_list.argtypes = [vp, cp, C.POINTER(sz)]
    L.mcomo_construct_ref2.restype = C.c_long; L.mcomo_construct_ref2.argtypes = [vp, sz, i32, vp, sz, vp, sz]
    L.mcomo_result_digest.restype = None; L.mcomo_result_digest.argtypes = [vp, C.POINTER(u64)]
    L.mcomo_synth_reads.restype = None
    L.mcomo_synth_reads.argtypes = [u64, u64, i32, i32, C.c_double, u64, u64, vp]
    _lib = L
    return L


MM_DTYPE = np.dtype([("x", "<u8"), ("y", "<u8")])


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def hash64(key: int, mask: int) -> int:
    return int(lib().mcomo_hash64(key, mask))


def sketch_two(seq: bytes, k: int, rid: int):
    m = MM128()
    lib().mcomo_sketch_two(seq, len(seq), k, rid, C.byref(m))
    return int(m.x), int(m.y)


def sketch_lh_ori(seq: bytes, w: int, k: int, rid: int) -> np.ndarray:
    cap = len(seq) + 8
    out = np.zeros(cap, dtype=MM_DTYPE)
    n = lib().mcomo_sketch_lh_ori(seq, len(seq), w, k, rid, _ptr(out), cap)
    assert n <= cap
    return out[:n]


def sketch_two_batch(reads: np.ndarray, k: int, rid0: int = 0) -> np.ndarray:
    reads = np.ascontiguousarray(reads, dtype=np.uint8)
    n, L = reads.shape
    out = np.zeros(n, dtype=MM_DTYPE)
    lib().mcomo_sketch_two_batch(_ptr(reads), n, L, k, rid0, _ptr(out))
    return out


def process_reads_batch(reads: np.ndarray, k: int, e: int = 4, rid0: int = 0):
    """Returns (substituted reads, cls, rec, n_cnt); the input is not modified."""
    reads = np.array(reads, dtype=np.uint8, order="C", copy=True)
    n, L = reads.shape
    cls = np.zeros(n, dtype=np.uint8)
    rec = np.zeros(n, dtype=MM_DTYPE)
    ncnt = np.zeros(n, dtype=np.uint16)
    lib().mcomo_process_reads_batch(_ptr(reads), n, L, k, e, rid0, _ptr(cls), _ptr(rec), _ptr(ncnt))
    return reads, cls, rec, ncnt


def radix_sort_128x(a: np.ndarray) -> np.ndarray:
    a = np.array(a, dtype=MM_DTYPE, order="C", copy=True)
    base = a.ctypes.data
    lib().mcomo_radix_sort_128x(C.c_void_p(base), C.c_void_p(base + 16 * len(a)))
    return a


def match_pro(s0: bytes, s1: bytes, i: int, j: int) -> int:
    return int(lib().mcomo_match_pro(s0, s1, i, j))


def encode_byte(seq: bytes, ref: bytes, pos: int, d: int, L: int) -> int:
    return int(lib().mcomo_encode_byte(seq, ref, pos, d, L))


def construct_ref2(reads: np.ndarray, members) -> bytes:
    """construct_ref2 (kthread_cb.c:105-218) of one member list over reads [n, L]: the consensus string."""
    reads = np.ascontiguousarray(reads, dtype=np.uint8)
    n, L = reads.shape
    mem = np.array(members, dtype=np.uint64)
    cap = int(((mem & np.uint64(0xFFFFFFFF)) >> np.uint64(1)).max()) + 2 * L + 8
    buf = C.create_string_buffer(cap)
    ln = lib().mcomo_construct_ref2(_ptr(reads), n, L, _ptr(mem), len(mem), buf, cap)
    assert ln >= 0
    return buf.raw[:ln]


def string_to_bits(s: bytes) -> np.ndarray:
    L = len(s)
    w = np.zeros((2 * L + 63) // 64, dtype=np.uint64)
    lib().mcomo_string_to_bits(s, L, _ptr(w))
    return w


def dict_layout(L: int, ininumdict: int = 0):
    st = np.zeros(64, dtype=np.int32); en = np.zeros(64, dtype=np.int32)
    nd = lib().mcomo_dict_layout(L, ininumdict, _ptr(st), _ptr(en))
    return st[:nd].copy(), en[:nd].copy()


def synth_reads(seed: int, n_reads: int, L: int, coverage: int = 30, sub_rate: float = 0.005,
                first: int = 0, count: int | None = None) -> np.ndarray:
    if count is None:
        count = n_reads - first
    out = np.zeros((count, L), dtype=np.uint8)
    lib().mcomo_synth_reads(seed, n_reads, L, coverage, sub_rate, first, count, _ptr(out))
    return out


class Pipeline:
    """Staged CPU run of the whole hot path (Stage 1 + Stage 2 of the reference at one thread)."""

    def __init__(self, reads: np.ndarray, **params):
        reads = np.ascontiguousarray(reads, dtype=np.uint8)
        self.n, self.L = reads.shape
        p = Params(**{k: int(v) for k, v in params.items()})
        self._h = lib().mcomo_new(_ptr(reads), self.n, self.L, C.byref(p))

    def force_maxsearch(self, v: int):
        """Test hook: cut bins at v entries instead of the reference's 500 / 2000."""
        lib().mcomo_force_maxsearch(self._h, int(v))

    def close(self):
        if self._h:
            lib().mcomo_free(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def stage_reads(self): lib().mcomo_stage_reads(self._h)
    def stage_bucket(self): lib().mcomo_stage_bucket(self._h)
    def stage_combine(self): lib().mcomo_stage_combine(self._h)
    def realign_pass(self, thr: int) -> int: return int(lib().mcomo_stage_realign_pass(self._h, thr))
    def update_single(self): lib().mcomo_update_single(self._h)
    def run_all(self): lib().mcomo_run_all(self._h)

    def dump_stages(self, path: str):
        rc = lib().mcomo_dump_stages(self._h, path.encode())
        if rc:
            raise OSError("cannot write " + path)

    def result_digest(self):
        """The product's mcomh_result_digest restated over the oracle's contig set: eight numbers, equal iff strings, member
        lists, offsets and id lists are."""
        out = (C.c_uint64 * 8)()
        lib().mcomo_result_digest(self._h, out)
        return [int(x) for x in out]

    def counter(self, name: str) -> int:
        return int(lib().mcomo_counter(self._h, name.encode()))

    def _arr(self, ptr, dtype, count):
        if count == 0:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_char * (np.dtype(dtype).itemsize * count)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype, count=count).copy()

    @property
    def seq(self) -> np.ndarray:
        a = self._arr(lib().mcomo_seq(self._h), np.uint8, self.n * (self.L + 1))
        return a.reshape(self.n, self.L + 1)[:, : self.L].copy()

    @property
    def cls(self): return self._arr(lib().mcomo_cls(self._h), np.uint8, self.n)
    @property
    def rec0(self): return self._arr(lib().mcomo_rec0(self._h), MM_DTYPE, self.n)
    @property
    def sg(self): return self._arr(lib().mcomo_sg(self._h), np.uint32, lib().mcomo_n_sg(self._h))
    @property
    def sg_flag(self): return self._arr(lib().mcomo_sg_flag(self._h), np.uint8, lib().mcomo_n_sg(self._h))

    def id_list(self, name: str) -> np.ndarray:
        n = C.c_size_t()
        p = lib().mcomo_list(self._h, name.encode(), C.byref(n))
        return self._arr(p, np.uint32, n.value)

    def contigs(self):
        out = []
        for i in range(lib().mcomo_n_contigs(self._h)):
            n = lib().mcomo_contig_n(self._h, i)
            out.append((lib().mcomo_contig_ref(self._h, i), self._arr(lib().mcomo_contig_members(self._h, i), np.uint64, n)))
        return out
