"""ctypes binding of libmcom_hip.so (include/mcom.h) over torch CUDA(=HIP) tensors."""
from __future__ import annotations

import ctypes as C
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MM_DTYPE = np.dtype([("x", "<u8"), ("y", "<u8")])


class McomError(RuntimeError):
    pass


def lib_path() -> str:
    return os.path.join(HERE, "lib", "libmcom_hip.so")


def header_path() -> str:
    return os.path.join(os.path.dirname(HERE), "include", "mcom.h")


def _declared_symbols() -> list[str]:
    """every entry point include/mcom.h (the boundary) and include/mcom_test.h (test hooks) declare"""
    names = set()
    for h in (header_path(), os.path.join(os.path.dirname(header_path()), "mcom_test.h")):
        txt = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(mcom_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


ABI_SYMBOLS = _declared_symbols()

_lib = None


def load_library():
    """Loads the HIP library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise McomError(f"{p} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(there is no CPU fallback for the product path)")
    # PyTorch ships its own libamdhip64.so.7; whichever HIP runtime is mapped first serves the whole process,
    # and torch.cuda stops working when it is not torch's.  So torch is always imported before our library.
    import torch  # noqa: F401
    L = C.CDLL(p)
    vp, sz, i32, u32, u64 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_uint64
    L.mcom_create.restype = i32; L.mcom_create.argtypes = [C.POINTER(vp), i32, vp]
    L.mcom_destroy.restype = None; L.mcom_destroy.argtypes = [vp]
    L.mcom_set_stream.restype = i32; L.mcom_set_stream.argtypes = [vp, vp]
    L.mcom_sync.restype = i32; L.mcom_sync.argtypes = [vp]
    L.mcom_last_error.restype = C.c_char_p; L.mcom_last_error.argtypes = [vp]
    L.mcom_version.restype = C.c_char_p; L.mcom_version.argtypes = []
    L.mcom_prof_enable.restype = i32; L.mcom_prof_enable.argtypes = [vp, i32]
    L.mcom_prof_reset.restype = i32; L.mcom_prof_reset.argtypes = [vp]
    L.mcom_prof_read.restype = i32; L.mcom_prof_read.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(u64)]
    L.mcom_prof_kernels.restype = i32; L.mcom_prof_kernels.argtypes = [vp, C.c_char_p, C.c_char_p, sz, C.POINTER(sz)]
    L.mcom_process_reads.restype = i32
    L.mcom_process_reads.argtypes = [vp, vp, sz, sz, i32, i32, i32, u32, vp, vp, vp, vp, vp]
    L.mcom_special_reads.restype = i32
    L.mcom_special_reads.argtypes = [vp, vp, sz, vp, u32, vp]
    L.mcom_process_reads_packed.restype = i32
    L.mcom_process_reads_packed.argtypes = [vp, vp, vp, sz, i32, i32, i32, u32, vp, vp, vp, vp, vp]
    L.mcom_sketch_reads.restype = i32
    L.mcom_sketch_reads.argtypes = [vp, vp, vp, sz, i32, i32, u32, vp]
    L.mcom_hash64_batch.restype = i32; L.mcom_hash64_batch.argtypes = [vp, vp, sz, i32, vp]
    L.mcom_encode_byte.restype = i32; L.mcom_encode_byte.argtypes = [vp, vp, vp, vp, vp, vp, vp, sz, i32, vp]
    L.mcom_radix_sort_128x.restype = i32; L.mcom_radix_sort_128x.argtypes = [vp, vp, sz]
    L.mcom_sort_group.restype = i32
    L.mcom_sort_group.argtypes = [vp, vp, sz, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]
    L.mcom_sketch_contigs.restype = i32
    L.mcom_sketch_contigs.argtypes = [vp, vp, vp, vp, sz, i32, i32, u32, vp, vp, sz, C.POINTER(u64)]
    L.mcom_pack_contigs.restype = i32; L.mcom_pack_contigs.argtypes = [vp, vp, vp, vp, u32, u64, vp]
    L.mcom_idx_build.restype = i32; L.mcom_idx_build.argtypes = [vp, vp, sz, i32, i32, C.POINTER(vp)]
    L.mcom_radix_sort_128x_ref_order.restype = i32; L.mcom_radix_sort_128x_ref_order.argtypes = [vp, vp, sz]
    L.mcom_idx_destroy.restype = None; L.mcom_idx_destroy.argtypes = [vp, vp]
    L.mcom_idx_get.restype = i32; L.mcom_idx_get.argtypes = [vp, vp, vp, sz, vp, vp]
    L.mcom_idx_records.restype = i32; L.mcom_idx_records.argtypes = [vp, vp, vp, C.POINTER(sz)]
    L.mcom_match_pro.restype = i32; L.mcom_match_pro.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.mcom_find_next_candidates.restype = i32
    L.mcom_find_next_candidates.argtypes = [vp, vp, vp, sz, vp, vp, vp, i32, vp, sz, vp]
    L.mcom_find_next_candidates_new.restype = i32
    L.mcom_find_next_candidates_new.argtypes = [vp, vp, vp, sz, vp, vp, vp, i32, u32, vp, sz, vp]
    L.mcom_dict_layout.restype = i32; L.mcom_dict_layout.argtypes = [i32, i32, vp, vp]
    L.mcom_gather_rows.restype = i32; L.mcom_gather_rows.argtypes = [vp, vp, vp, sz, i32, vp]
    L.mcom_poly_filter.restype = i32; L.mcom_poly_filter.argtypes = [vp, vp, vp, vp, sz, i32, i32, vp]
    L.mcom_dicts_build.restype = i32; L.mcom_dicts_build.argtypes = [vp, vp, sz, i32, i32, C.POINTER(vp)]
    L.mcom_dicts_free.restype = None; L.mcom_dicts_free.argtypes = [vp, vp]
    L.mcom_dicts_info.restype = i32; L.mcom_dicts_info.argtypes = [vp, C.POINTER(i32), vp, vp]
    L.mcom_dicts_lookup.restype = i32; L.mcom_dicts_lookup.argtypes = [vp, vp, i32, vp, sz, vp, vp]
    L.mcom_dicts_ids.restype = i32; L.mcom_dicts_ids.argtypes = [vp, vp, i32, vp]
    L.mcom_realign_pass.restype = i32
    L.mcom_realign_pass.argtypes = [vp, vp, vp, vp, vp, vp, vp, u32, u64, i32, i32, vp, vp]
    L.mcom_cindex_plan.restype = i32
    L.mcom_cindex_plan.argtypes = [u64, u32, i32, i32, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    L.mcom_cindex_build.restype = i32
    L.mcom_cindex_build.argtypes = [vp, vp, vp, vp, u32, u64, i32, i32, u64, vp, u64]
    L.mcom_dicts_eligible.restype = i32
    L.mcom_dicts_eligible.argtypes = [vp, vp, vp, i32, vp]
    L.mcom_realign_pass_reads.restype = i32
    L.mcom_realign_pass_reads.argtypes = [vp, vp, u64, vp, vp, vp, sz, vp, vp, vp, u32, i32, i32, i32, vp, vp]
    L.mcom_dicts_screen.restype = i32
    L.mcom_dicts_screen.argtypes = [vp, vp, sz, i32, i32, i32, C.POINTER(i32)]
    L.mcom_claims_resolve.restype = i32
    L.mcom_claims_resolve.argtypes = [vp, vp, vp, sz, u32, vp, vp, vp, C.POINTER(u64)]
    L.mcom_scan_u64.restype = i32; L.mcom_scan_u64.argtypes = [vp, vp, vp, sz]
    L.mcom_contig_layout.restype = i32; L.mcom_contig_layout.argtypes = [vp, vp, sz, vp, vp, C.POINTER(u64)]
    L.mcom_group_consensus.restype = i32; L.mcom_group_consensus.argtypes = [vp, vp, vp, vp, u32, i32, i32, i32, vp, vp, vp, vp, vp, i32]
    L.mcom_groups_to_contigs.restype = i32
    L.mcom_groups_to_contigs.argtypes = [vp, vp, vp, sz, vp, vp, vp, vp, vp, i32, u64, u64, u64, vp, u64, vp, vp, u64, vp, u64, vp, vp, u64, C.POINTER(u64)]
    L.mcom_merge_members.restype = i32; L.mcom_merge_members.argtypes = [vp, vp, vp, vp, sz, i32, i32, vp, vp, vp, C.POINTER(u64)]
    L.mcom_merge_consensus_jobs.restype = i32; L.mcom_merge_consensus_jobs.argtypes = [vp, vp, vp, vp, vp, sz, u64, i32, vp, vp, vp, vp]
    L.mcom_contigs_carry.restype = i32; L.mcom_contigs_carry.argtypes = [vp, vp, vp, vp, vp, sz, vp, sz, sz, vp, vp, vp, vp, vp, C.POINTER(u64)]
    L.mcom_resketch_merged.restype = i32
    L.mcom_resketch_merged.argtypes = [vp, vp, sz, vp, vp, vp, vp, vp, u64, i32, i32, vp, vp, sz, C.POINTER(u64), C.POINTER(u64)]
    L.mcom_members_finalize.restype = i32
    L.mcom_members_finalize.argtypes = [vp, vp, vp, sz, u64, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64), i32, i32, vp, vp]
    L.mcom_compact_live.restype = i32; L.mcom_compact_live.argtypes = [vp, vp, vp, sz, vp, C.POINTER(u64)]
    L.mcom_window_layout.restype = i32; L.mcom_window_layout.argtypes = [vp, vp, sz, i32, vp, C.POINTER(u64), C.POINTER(u64)]
    L.mcom_records_carry.restype = i32; L.mcom_records_carry.argtypes = [vp, vp, vp, vp, sz, u32, u32, vp, sz, vp, C.POINTER(u64)]
    L.mcom_claim_pairs.restype = i32
    L.mcom_claim_pairs.argtypes = [vp, vp, sz, sz, i32, vp, vp, C.POINTER(u64), C.POINTER(i32)]
    L.mcom_synth_reads.restype = i32
    L.mcom_synth_reads.argtypes = [vp, u64, u64, i32, i32, C.c_double, u64, u64, vp, sz]
    L.mcom_synth_reads_genome.restype = i32
    L.mcom_synth_reads_genome.argtypes = [vp, u64, u64, i32, i32, C.c_double, i32, u64, u64, vp, sz]
    _lib = L
    return L


def words_per_read(L: int) -> int:
    return (2 * L + 63) // 64


def _torch():
    import torch
    return torch


class Context:
    """One context per process / GPU (mcom_ctx).  All tensors must live on this context's device."""

    def __init__(self, device: int = 0, stream=None):
        torch = _torch()
        if not torch.cuda.is_available():
            raise McomError("no GPU visible: the minicom_amd product path has no CPU fallback")
        self.lib = load_library()
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self._stream = stream if stream is not None else torch.cuda.current_stream(self.device)
        h = C.c_void_p()
        rc = self.lib.mcom_create(C.byref(h), device, C.c_void_p(self._stream.cuda_stream))
        if rc:
            raise McomError(f"mcom_create failed ({rc})")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.lib.mcom_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers
    def _check(self, rc: int):
        if rc:
            raise McomError(f"mcom error {rc}: {self.lib.mcom_last_error(self._h).decode()}")

    def _p(self, t, dtype=None):
        if t is None:
            return C.c_void_p(0)
        assert t.is_cuda and t.is_contiguous(), "device, contiguous tensors only"
        if dtype is not None:
            assert t.dtype == dtype, (t.dtype, dtype)
        return C.c_void_p(t.data_ptr())

    def sync(self):
        self._check(self.lib.mcom_sync(self._h))

    def empty_records(self, n: int):
        torch = _torch()
        return torch.empty((n, 2), dtype=torch.int64, device=self.device)  # mm128_t = 2 x u64, viewed as int64

    # -- include/mcom.h entry points
    def process_reads(self, ascii_, L: int, k: int, e: int = 4, rid0: int = 0, want_nmask: bool = False):
        """mcom_process_reads.  ascii_: uint8 [n, pitch] on the device.  Returns dict of device tensors."""
        torch = _torch()
        n, pitch = ascii_.shape
        W = words_per_read(L)
        packed = torch.empty((n, W), dtype=torch.int64, device=self.device)
        cls = torch.empty(n, dtype=torch.uint8, device=self.device)
        ncnt = torch.empty(n, dtype=torch.int16, device=self.device)
        nmask = torch.empty((n, (L + 63) // 64), dtype=torch.int64, device=self.device) if want_nmask else None
        rec = self.empty_records(n)
        self._check(self.lib.mcom_process_reads(self._h, self._p(ascii_, torch.uint8), pitch, n, L, k, e, rid0,
                                                self._p(packed), self._p(cls), self._p(ncnt), self._p(nmask), self._p(rec)))
        return {"packed": packed, "cls": cls, "ncnt": ncnt, "nmask": nmask, "rec": rec}

    def process_reads_packed(self, in_packed, in_nmask, L: int, k: int, e: int = 4, rid0: int = 0, in_place: bool = False):
        """mcom_process_reads_packed: reads as a packing parser sends them (int64 [n, W] codes with 0 at an N, int64 [n, ceil(L/64)] N flags)."""
        torch = _torch()
        n = int(in_packed.shape[0])
        packed = in_packed if in_place else torch.empty_like(in_packed)
        nmask = in_nmask if in_place else torch.empty_like(in_nmask)
        cls = torch.empty(n, dtype=torch.uint8, device=self.device)
        ncnt = torch.empty(n, dtype=torch.int16, device=self.device)
        rec = self.empty_records(n)
        self._check(self.lib.mcom_process_reads_packed(self._h, self._p(in_packed, torch.int64), self._p(in_nmask, torch.int64), n, L, k, e, rid0,
                                                       self._p(packed), self._p(cls), self._p(ncnt), self._p(nmask), self._p(rec)))
        return {"packed": packed, "cls": cls, "ncnt": ncnt, "nmask": nmask, "rec": rec}

    def special_reads(self, cls, cap: int = 1 << 20):
        """mcom_special_reads: the reads of another class than 0 as (rid << 8 | class), in no particular order.  Returns (list int64 [min(count, cap)], count)."""
        torch = _torch()
        n = int(cls.shape[0])
        out = torch.empty(max(cap, 1), dtype=torch.int64, device=self.device)
        cnt = torch.zeros(2, dtype=torch.int32, device=self.device)
        self._check(self.lib.mcom_special_reads(self._h, self._p(cls, torch.uint8), n, self._p(out), cap, self._p(cnt)))
        self.sync()
        c = int(cnt[0].item())
        return out[: min(c, cap)], c

    def sketch_reads(self, packed, L: int, k: int, rids=None, rid0: int = 0, out=None):
        """mcom_sketch_reads.  packed: int64 [N, W]; rids: optional int32 [n] row selector."""
        torch = _torch()
        n = int(rids.shape[0]) if rids is not None else int(packed.shape[0])
        rec = out if out is not None else self.empty_records(n)
        self._check(self.lib.mcom_sketch_reads(self._h, self._p(packed, torch.int64),
                                               self._p(rids, torch.int32) if rids is not None else C.c_void_p(0),
                                               n, L, k, rid0, self._p(rec)))
        return rec

    def hash64(self, kmers, k: int):
        """mcom_hash64_batch: hash64 (sketch.c:27-37) of int64 k-mers."""
        torch = _torch()
        out = torch.empty_like(kmers)
        self._check(self.lib.mcom_hash64_batch(self._h, self._p(kmers, torch.int64), int(kmers.shape[0]), k, self._p(out)))
        return out

    def encode_byte(self, rows, cg, contig, pos, dirs, L: int):
        """mcom_encode_byte: the cost test of kthread_hash_realign.c:283-314 for (read row, contig window) pairs."""
        torch = _torch()
        n = int(rows.shape[0])
        ok = torch.empty(max(n, 1), dtype=torch.uint8, device=self.device)
        self._check(self.lib.mcom_encode_byte(self._h, self._p(rows, torch.int64), self._p(cg["cbits"]), self._p(cg["coff"]), self._p(contig, torch.int32),
                                              self._p(pos, torch.int32), self._p(dirs, torch.uint8), n, L, self._p(ok)))
        return ok[:n]

    def radix_sort_128x(self, rec):
        """mcom_radix_sort_128x: in place, by x ascending, stable."""
        torch = _torch()
        self._check(self.lib.mcom_radix_sort_128x(self._h, self._p(rec, torch.int64), int(rec.shape[0])))
        return rec

    def counter(self, name: str) -> int:
        self.lib.mcom_counter.restype = C.c_uint64; self.lib.mcom_counter.argtypes = [C.c_void_p, C.c_char_p]
        return int(self.lib.mcom_counter(self._h, name.encode()))

    def set_segment_capacity(self, records: int):
        """Test hook of mcom_sort_group: segments above `records` go through the nine-pass fallback (0 = default 4096)."""
        self.lib.mcom_set_segment_capacity.restype = C.c_int; self.lib.mcom_set_segment_capacity.argtypes = [C.c_void_p, C.c_uint32]
        self._check(self.lib.mcom_set_segment_capacity(self._h, records))

    def set_index_capacity(self, entries: int):
        """Test hook of mcom_cindex_build: partitions above `entries` use the scattered placement (negative = default)."""
        self.lib.mcom_set_index_capacity.restype = C.c_int; self.lib.mcom_set_index_capacity.argtypes = [C.c_void_p, C.c_int]
        self._check(self.lib.mcom_set_index_capacity(self._h, entries))

    def set_sketch_kernel(self, wave_per_string):
        """Test hook of mcom_sketch_contigs: 1 / True = the wave-per-string kernel, 2 = the lane-per-string kernel with the 64-bit ring,
        0 / False = the default choice (the ring of 32-bit prefixes for odd k)."""
        self.lib.mcom_set_sketch_kernel.restype = C.c_int; self.lib.mcom_set_sketch_kernel.argtypes = [C.c_void_p, C.c_int]
        self._check(self.lib.mcom_set_sketch_kernel(self._h, int(wave_per_string)))

    def set_sketch_prefix_bits(self, bits: int):
        """Test hook: the prefix width of the prefix ring (14 by default, in 16-bit words; above 14 in 32-bit words); a few bits make prefix ties the rule."""
        self.lib.mcom_set_sketch_prefix_bits.restype = C.c_int; self.lib.mcom_set_sketch_prefix_bits.argtypes = [C.c_void_p, C.c_int]
        self._check(self.lib.mcom_set_sketch_prefix_bits(self._h, bits))

    def set_claim_route(self, route: int):
        """Test hook of mcom_claim_pairs: 0 = default (one launch, the launch-per-round loop behind it), 1 = the loop at once, 2 = the
        one-launch kernel's first barrier gives up (poison flag trips, the loop takes over), 3 = the one-launch kernel without the
        single-workgroup tail."""
        self.lib.mcom_set_claim_route.restype = C.c_int; self.lib.mcom_set_claim_route.argtypes = [C.c_void_p, C.c_int]
        self._check(self.lib.mcom_set_claim_route(self._h, route))

    def claim_fallbacks(self) -> int:
        self.lib.mcom_claim_fallbacks.restype = C.c_int; self.lib.mcom_claim_fallbacks.argtypes = [C.c_void_p]
        return int(self.lib.mcom_claim_fallbacks(self._h))

    def set_screen_route(self, route: int):
        """Test hook of mcom_dicts_screen: 0 = default (keys binned by counter range, counted in LDS; the global-atomics kernel behind it),
        1 = the global-atomics kernel at once, 2 = regions of a few keys (every input overflows: the fall-back runs)."""
        self.lib.mcom_set_screen_route.restype = C.c_int; self.lib.mcom_set_screen_route.argtypes = [C.c_void_p, C.c_int]
        self._check(self.lib.mcom_set_screen_route(self._h, route))

    def screen_fallbacks(self) -> int:
        self.lib.mcom_screen_fallbacks.restype = C.c_int; self.lib.mcom_screen_fallbacks.argtypes = [C.c_void_p]
        return int(self.lib.mcom_screen_fallbacks(self._h))

    def set_consensus_capacity(self, members: int):
        """Test hook of the merge consensus: units that more than `members` members reach use the wave-per-tile kernel (0 = default 127)."""
        self.lib.mcom_set_consensus_capacity.restype = C.c_int; self.lib.mcom_set_consensus_capacity.argtypes = [C.c_void_p, C.c_uint32]
        self._check(self.lib.mcom_set_consensus_capacity(self._h, members))

    def sort_group(self, rec, L: int, k_orig: int, kmer: int, b: int = 14):
        """mcom_sort_group.  Returns dict(sorted, singles, members, group_off) trimmed to their counts."""
        torch = _torch()
        n = int(rec.shape[0])
        srt = self.empty_records(n)
        singles = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        sord = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        members = torch.empty(max(n, 1), dtype=torch.int64, device=self.device)
        goff = torch.empty(n // 2 + 2, dtype=torch.int32, device=self.device)
        cnt = (C.c_uint64 * 4)()
        self._check(self.lib.mcom_sort_group(self._h, self._p(rec, torch.int64), n, L, k_orig, kmer, b, self._p(srt),
                                             self._p(singles), self._p(sord), self._p(members), self._p(goff), cnt))
        nv, ns, ng, nm = (int(x) for x in cnt)
        return {"sorted": srt, "n_valid": nv, "singles": singles[:ns], "single_ord": sord[:ns], "members": members[:nm], "group_off": goff[:ng + 1] if n else goff[:0],
                "n_groups": ng}

    # -- contigs
    def upload_contigs(self, refs):
        """Host list of contig strings (bytes) -> device (seq uint8, off int64 [n+1], coff int64 [n], clen int32 [n], cbits)."""
        torch = _torch()
        lens = np.array([len(r) for r in refs], dtype=np.int64)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        words = (2 * lens + 63) // 64 + 1
        coff = np.concatenate([[0], np.cumsum(words)[:-1]]).astype(np.int64) if len(refs) else np.zeros(0, np.int64)
        total = int(words.sum())
        seq = torch.from_numpy(np.frombuffer(b"".join(refs) + b"\0", dtype=np.uint8).copy()).to(self.device)
        d_off = torch.from_numpy(off).to(self.device)
        d_coff = torch.from_numpy(coff).to(self.device)
        d_clen = torch.from_numpy(lens.astype(np.int32)).to(self.device)
        cbits = torch.zeros(total + 1, dtype=torch.int64, device=self.device)
        self._check(self.lib.mcom_pack_contigs(self._h, self._p(seq), self._p(d_off), self._p(d_coff), len(refs), total, self._p(cbits)))
        return {"seq": seq, "off": d_off, "coff": d_coff, "clen": d_clen, "cbits": cbits, "n": len(refs), "lens": lens}

    def unpack_contigs(self, cbits, coff, off, n: int, byte_lo: int, byte_hi: int, seq):
        """mcom_unpack_contigs: the strings of n contigs from their packed words, into seq (uint8, 16-byte aligned) at off[c]."""
        self.lib.mcom_unpack_contigs.restype = C.c_int
        self.lib.mcom_unpack_contigs.argtypes = [C.c_void_p] * 4 + [C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p]
        self._check(self.lib.mcom_unpack_contigs(self._h, self._p(cbits), self._p(coff), self._p(off), n, byte_lo, byte_hi, self._p(seq)))

    def sketch_contigs(self, seq, off, n: int, w: int, k: int, max_per_contig: int = 0, ids=None):
        """mcom_sketch_contigs.  Returns (moff int32 [n+1], records int64 [total,2])."""
        torch = _torch()
        moff = torch.empty(n + 1, dtype=torch.int32, device=self.device)
        total = C.c_uint64(0)
        cap = max(1024, 2 * n * (max_per_contig if max_per_contig else 4))
        while True:
            out = self.empty_records(cap)
            rc = self.lib.mcom_sketch_contigs(self._h, self._p(seq), self._p(off), self._p(ids), n, w, k, max_per_contig,
                                              self._p(moff), self._p(out), cap, C.byref(total))
            if rc == -4:
                cap = int(total.value)
                continue
            self._check(rc)
            return moff, out[: int(total.value)]

    def idx_build(self, rec, k: int, b: int = 14):
        """mcom_idx_build: b > 0 reproduces the reference's bucket order, b = 0 is one stable sort by x."""
        return Index(self, rec, k, b)

    def radix_sort_128x_ref_order(self, rec):
        """mcom_radix_sort_128x_ref_order: in place, the reference's exact (unstable) element order."""
        torch = _torch()
        self._check(self.lib.mcom_radix_sort_128x_ref_order(self._h, self._p(rec, torch.int64), int(rec.shape[0])))
        return rec

    def match_pro(self, cg, a, pa, b, pb):
        torch = _torch()
        n = int(a.shape[0])
        out = torch.empty(n, dtype=torch.int32, device=self.device)
        self._check(self.lib.mcom_match_pro(self._h, self._p(cg["cbits"]), self._p(cg["coff"]), self._p(cg["clen"]),
                                            self._p(a, torch.int32), self._p(pa, torch.int32), self._p(b, torch.int32), self._p(pb, torch.int32), n, self._p(out)))
        return out

    def find_next_candidates(self, idx, query, cg, cbthr: int):
        """mcom_find_next_candidates.  Returns (pairs int64 [n_pass,2], n_pairs_tested)."""
        cnt = (C.c_uint64 * 2)()
        cap = max(1024, int(query.shape[0]))
        while True:
            out = self.empty_records(cap)
            rc = self.lib.mcom_find_next_candidates(self._h, idx._h, self._p(query), int(query.shape[0]), self._p(cg["cbits"]),
                                                    self._p(cg["coff"]), self._p(cg["clen"]), cbthr, self._p(out), cap, cnt)
            if rc == -4:
                cap = int(cnt[1])
                continue
            self._check(rc)
            return out[: int(cnt[1])], int(cnt[0])

    def minimizer_prefix_ord(self, moff, rec, ord_, m: int):
        """mcom_minimizer_prefix_ord: the first m records of contigs ord_[0 ..] (None: all, in order), concatenated.  Returns (moff2, records)."""
        torch = _torch()
        self.lib.mcom_minimizer_prefix_ord.restype = C.c_int
        self.lib.mcom_minimizer_prefix_ord.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
        n = int(ord_.shape[0]) if ord_ is not None else int(moff.shape[0]) - 1
        moff2 = torch.empty(n + 1, dtype=torch.int32, device=self.device)
        out = self.empty_records(max(n * m, 1))
        tot = C.c_uint64()
        self._check(self.lib.mcom_minimizer_prefix_ord(self._h, self._p(moff, torch.int32), self._p(rec), self._p(ord_, torch.int32) if ord_ is not None else None, n, m,
                                                       self._p(moff2), self._p(out), C.byref(tot)))
        return moff2, out[: int(tot.value)]

    def find_next_candidates_ord(self, idx, rec, roff, ord_, cg, cbthr: int, first_new: int = 0, n_new: int = 0):
        """mcom_find_next_candidates_ord: queries = the records of contigs ord_[0 ..] (None: all) of a store.  Returns (pairs, n_pairs_listed)."""
        self.lib.mcom_find_next_candidates_ord.restype = C.c_int
        self.lib.mcom_find_next_candidates_ord.argtypes = [C.c_void_p] * 5 + [C.c_size_t] + [C.c_void_p] * 3 + [C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p]
        torch = _torch()
        n = int(ord_.shape[0]) if ord_ is not None else int(roff.shape[0]) - 1
        cnt = (C.c_uint64 * 2)()
        cap = 1024
        while True:
            out = self.empty_records(cap)
            rc = self.lib.mcom_find_next_candidates_ord(self._h, idx._h, self._p(rec), self._p(roff, torch.int32), self._p(ord_, torch.int32) if ord_ is not None else None, n,
                                                        self._p(cg["cbits"]), self._p(cg["coff"]), self._p(cg["clen"]), cbthr, first_new, n_new, self._p(out), cap, cnt)
            if rc == -4:
                cap = int(cnt[1])
                continue
            self._check(rc)
            return out[: int(cnt[1])], int(cnt[0])

    # -- Stage 2
    def gather_rows(self, packed, rids, L: int):
        torch = _torch()
        n = int(rids.shape[0])
        out = torch.empty((n, words_per_read(L)), dtype=torch.int64, device=self.device)
        self._check(self.lib.mcom_gather_rows(self._h, self._p(packed, torch.int64), self._p(rids, torch.int32), n, L, self._p(out)))
        return out

    def poly_filter(self, sgbits, L: int, thr: int, nmask=None, rids=None):
        torch = _torch()
        n = int(sgbits.shape[0])
        flag = torch.empty(n, dtype=torch.uint8, device=self.device)
        self._check(self.lib.mcom_poly_filter(self._h, self._p(sgbits, torch.int64), self._p(nmask), self._p(rids), n, L, thr, self._p(flag)))
        return flag

    def dicts_build(self, sgbits, L: int, ininumdict: int = 0):
        return Dicts(self, sgbits, L, ininumdict)

    def realign_pass(self, dicts, sgbits, sgflag, cbits, coff, woff, n_windows: int, thr: int, maxsearch: int, stats: bool = False):
        """mcom_realign_pass.  Returns (claim int64 [n_sg], stats tensor or None)."""
        torch = _torch()
        n_sg = int(sgbits.shape[0])
        claim = torch.empty(max(n_sg, 1), dtype=torch.int64, device=self.device)
        st = torch.zeros(3, dtype=torch.int64, device=self.device) if stats else None
        self._check(self.lib.mcom_realign_pass(self._h, dicts._h, self._p(sgbits, torch.int64), self._p(sgflag, torch.uint8),
                                               self._p(cbits, torch.int64), self._p(coff, torch.int64), self._p(woff, torch.int64),
                                               int(coff.shape[0]), int(n_windows), thr, maxsearch, self._p(claim), self._p(st)))
        return claim[:n_sg], st

    def cindex_build(self, cbits, coff, woff, n_windows: int, L: int, ininumdict: int = 0):
        """mcom_cindex_plan + mcom_cindex_build.  Returns (index words int64, n_parts)."""
        torch = _torch()
        ne, parts, nw = C.c_uint64(), C.c_uint64(), C.c_uint64()
        n_contigs = int(coff.shape[0])
        self._check(self.lib.mcom_cindex_plan(int(n_windows), n_contigs, L, ininumdict, C.byref(ne), C.byref(parts), C.byref(nw)))
        words = int(nw.value)
        for attempt in range(5):                                            # MCOM_E_OVERFLOW: this set's repeats need a larger extension area
            keys = torch.empty(words, dtype=torch.int64, device=self.device)
            rc = self.lib.mcom_cindex_build(self._h, self._p(cbits, torch.int64), self._p(coff, torch.int64), self._p(woff, torch.int64),
                                            n_contigs, int(n_windows), L, ininumdict, parts.value, self._p(keys), words)
            if rc != -4:
                break
            words += max(words, 8 * (int(ne.value) // 7 + 1024))
        self._check(rc)
        return keys, parts.value

    def cindex_build_shares(self, cbits, coff, woff, n_windows: int, L: int, ranks: int, ininumdict: int = 0):
        """The ONE index over all contigs cut by key into `ranks` shares, all of them built on this GPU: mcom_cindex_plan_shared,
        mcom_cindex_entries over the whole set (grouped by owning share) and mcom_cindex_place per share -- what R ranks do with an
        all-to-all of the entries in between (host/mcom_pipeline.cpp).  Returns [(index words int64, geom)] by share."""
        torch = _torch()
        u64 = C.c_uint64
        self.lib.mcom_cindex_plan_shared.restype = C.c_int
        self.lib.mcom_cindex_plan_shared.argtypes = [u64, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int] + [C.POINTER(u64)] * 4
        self.lib.mcom_cindex_entries.restype = C.c_int
        self.lib.mcom_cindex_entries.argtypes = [C.c_void_p] * 4 + [C.c_uint32] * 3 + [C.c_int, C.c_int, u64, C.c_void_p, C.c_void_p, u64, C.POINTER(u64)]
        self.lib.mcom_cindex_place.restype = C.c_int
        self.lib.mcom_cindex_place.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, u64, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, u64, C.c_void_p, u64]
        n_contigs = int(coff.shape[0])
        ne, share, geom0, nw = u64(), u64(), u64(), u64()
        self._check(self.lib.mcom_cindex_plan_shared(int(n_windows), n_contigs, L, ininumdict, ranks, 0, C.byref(ne), C.byref(share), C.byref(geom0), C.byref(nw)))
        cap = int(ne.value) + 1
        key = torch.empty(cap, dtype=torch.int32, device=self.device)
        slot = torch.empty(cap, dtype=torch.int64, device=self.device)
        counts = (u64 * ranks)()
        self._check(self.lib.mcom_cindex_entries(self._h, self._p(cbits, torch.int64), self._p(coff, torch.int64), self._p(woff, torch.int64), n_contigs, 0, n_contigs,
                                                 L, ininumdict, geom0.value, self._p(key), self._p(slot), cap, counts))
        out, at = [], 0
        for q in range(ranks):
            g, nwq = u64(), u64()
            self._check(self.lib.mcom_cindex_plan_shared(int(n_windows), n_contigs, L, ininumdict, ranks, q, C.byref(ne), C.byref(share), C.byref(g), C.byref(nwq)))
            n = int(counts[q])
            words = int(nwq.value)
            for attempt in range(5):                                        # MCOM_E_OVERFLOW: a larger extension area (place overwrites its input: copies)
                k, sl = key[at:at + n].clone(), slot[at:at + n].clone()
                kt, st = torch.empty(n + 1, dtype=torch.int32, device=self.device), torch.empty(n + 1, dtype=torch.int64, device=self.device)
                keys = torch.empty(words, dtype=torch.int64, device=self.device)
                rc = self.lib.mcom_cindex_place(self._h, self._p(k), self._p(sl), n, 0, self._p(kt), self._p(st), L, ininumdict, g.value, self._p(keys), words)
                if rc != -4:
                    break
                words += max(words, 8 * (int(ne.value) // 7 + 1024))
            self._check(rc)
            self.sync()
            out.append((keys, g.value))
            at += n
        return out

    def set_lookup_route(self, route: int):
        """Test hook of the Stage-2 lookups over a share of the keys: 0 = default (a thread per owned task, k_realign_owned), 1 = several
        (direction, dictionary) pairs per lane of the whole-index kernel."""
        self.lib.mcom_set_lookup_route.restype = C.c_int; self.lib.mcom_set_lookup_route.argtypes = [C.c_void_p, C.c_int]
        self._check(self.lib.mcom_set_lookup_route(self._h, route))

    def dicts_eligible(self, dicts, sgbits, maxsearch: int):
        torch = _torch()
        el = torch.zeros(max(int(sgbits.shape[0]), 1), dtype=torch.int32, device=self.device)
        self._check(self.lib.mcom_dicts_eligible(self._h, dicts._h, self._p(sgbits, torch.int64), maxsearch, self._p(el)))
        return el

    def dicts_screen(self, sgbits, L: int, maxsearch: int, ininumdict: int = 0) -> bool:
        """mcom_dicts_screen: False proves that no dictionary bin exceeds maxsearch."""
        torch = _torch()
        out = C.c_int(0)
        self._check(self.lib.mcom_dicts_screen(self._h, self._p(sgbits, torch.int64), int(sgbits.shape[0]), L, ininumdict, maxsearch, C.byref(out)))
        return bool(out.value)

    def realign_pass_reads(self, cindex, sgbits, sgflag, cbits, coff, woff, L: int, thr: int, ininumdict: int = 0, elig=None,
                           stats: bool = False):
        """mcom_realign_pass_reads.  cindex = cindex_build's result.  Returns (claim int64 [n_sg], stats or None)."""
        torch = _torch()
        keys, lg = cindex
        n_sg = int(sgbits.shape[0])
        claim = torch.empty(max(n_sg, 1), dtype=torch.int64, device=self.device)
        st = torch.zeros(3, dtype=torch.int64, device=self.device) if stats else None
        self._check(self.lib.mcom_realign_pass_reads(self._h, self._p(keys), lg, self._p(sgbits, torch.int64),
                                                     self._p(sgflag, torch.uint8), self._p(elig), n_sg, self._p(cbits, torch.int64),
                                                     self._p(coff, torch.int64), self._p(woff, torch.int64), int(coff.shape[0]), L,
                                                     ininumdict, thr, self._p(claim), self._p(st)))
        return claim[:n_sg], st

    # -- the device-resident contig set (include/mcom.h, "the contig set of combine_cluster")
    def scan_u64(self, x):
        torch = _torch()
        out = torch.empty_like(x)
        self._check(self.lib.mcom_scan_u64(self._h, self._p(x, torch.int64), self._p(out), int(x.shape[0])))
        return out

    def contig_layout(self, soff):
        """mcom_contig_layout.  Returns (coff_words int64 [n+1], clen int32 [n], total_words)."""
        torch = _torch()
        n = int(soff.shape[0]) - 1
        coff = torch.empty(n + 1, dtype=torch.int64, device=self.device)
        clen = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        tw = C.c_uint64()
        self._check(self.lib.mcom_contig_layout(self._h, self._p(soff, torch.int64), n, self._p(coff), self._p(clen), C.byref(tw)))
        return coff, clen[:n], int(tw.value)

    def group_consensus(self, packed, members, goff, L: int, k_orig: int, e: int):
        """mcom_group_consensus.  members is rewritten in place.  Returns dict(keep, nkept, sv, reflen, refs, stride)."""
        torch = _torch()
        ng, nm = int(goff.shape[0]) - 1, int(members.shape[0])
        stride = (2 * L + 15) & ~15
        o = dict(keep=torch.empty(max(nm, 1), dtype=torch.uint8, device=self.device), nkept=torch.empty(max(ng, 1), dtype=torch.int32, device=self.device),
                 sv=torch.empty(max(ng, 1), dtype=torch.int16, device=self.device), reflen=torch.empty(max(ng, 1), dtype=torch.int16, device=self.device),
                 refs=torch.zeros(max(ng, 1) * stride + 16, dtype=torch.uint8, device=self.device), stride=stride)
        self._check(self.lib.mcom_group_consensus(self._h, self._p(packed, torch.int64), self._p(members, torch.int64), self._p(goff, torch.int32), ng, L, k_orig, e,
                                                  self._p(o["keep"]), self._p(o["nkept"]), self._p(o["sv"]), self._p(o["reflen"]), self._p(o["refs"]), stride))
        return o

    def groups_to_contigs(self, members, goff, gc, cap_chars: int, cap_members: int, cap_contigs: int):
        """mcom_groups_to_contigs into a fresh set.  Returns dict(seq, soff, mem, moff, rej_rid, rej_group, counts)."""
        torch = _torch()
        ng = int(goff.shape[0]) - 1
        seq = torch.zeros(cap_chars + 16, dtype=torch.uint8, device=self.device)
        soff = torch.zeros(cap_contigs + 2, dtype=torch.int64, device=self.device)
        mem = torch.zeros(cap_members + 1, dtype=torch.int64, device=self.device)
        moff = torch.zeros(cap_contigs + 2, dtype=torch.int64, device=self.device)
        rr = torch.zeros(int(members.shape[0]) + 1, dtype=torch.int32, device=self.device)
        rg = torch.zeros(int(members.shape[0]) + 1, dtype=torch.int32, device=self.device)
        cnt = (C.c_uint64 * 4)()
        self._check(self.lib.mcom_groups_to_contigs(self._h, self._p(members, torch.int64), self._p(goff, torch.int32), ng, self._p(gc["keep"]), self._p(gc["nkept"]),
                                                    self._p(gc["sv"]), self._p(gc["reflen"]), self._p(gc["refs"]), gc["stride"], 0, 0, 0, self._p(seq), cap_chars,
                                                    self._p(soff), self._p(mem), cap_members, self._p(moff), cap_contigs + 1, self._p(rr), self._p(rg),
                                                    int(members.shape[0]), cnt))
        nc, nch, nmm, nrj = (int(v) for v in cnt)
        return dict(seq=seq[:nch], soff=soff[:nc + 1], mem=mem[:nmm], moff=moff[:nc + 1], rej_rid=rr[:nrj], rej_group=rg[:nrj], counts=(nc, nch, nmm, nrj))

    def merge_members(self, mem, moff, jobs, L: int, key_bits: int):
        """mcom_merge_members.  jobs: int32 [nj, 4].  Returns (jm, jmoff, jroff, totals)."""
        torch = _torch()
        nj = int(jobs.shape[0])
        jm = torch.empty(int(mem.shape[0]) + 1, dtype=torch.int64, device=self.device)
        jmoff = torch.empty(nj + 1, dtype=torch.int64, device=self.device)
        jroff = torch.empty(nj + 1, dtype=torch.int64, device=self.device)
        tot = (C.c_uint64 * 3)()
        self._check(self.lib.mcom_merge_members(self._h, self._p(mem, torch.int64), self._p(moff, torch.int64), self._p(jobs, torch.int32), nj, L, key_bits,
                                                self._p(jm), self._p(jmoff), self._p(jroff), tot))
        return jm[: int(tot[0])], jmoff, jroff, tuple(int(v) for v in tot)

    def merge_consensus_jobs(self, packed, jm, jmoff, jroff, total_chars: int, L: int, jobs=None, seq=None, soff=None):
        torch = _torch()
        nj = int(jmoff.shape[0]) - 1
        refs = torch.zeros(total_chars + 16, dtype=torch.uint8, device=self.device)
        self._check(self.lib.mcom_merge_consensus_jobs(self._h, self._p(packed, torch.int64), self._p(jm, torch.int64), self._p(jmoff, torch.int64),
                                                       self._p(jroff, torch.int64), nj, total_chars, L, self._p(refs), self._p(jobs), self._p(seq), self._p(soff)))
        self.sync()
        return refs[:total_chars]

    def contigs_carry(self, seq, soff, mem, moff, flag, nj: int, seq2, soff2, mem2, moff2):
        """mcom_contigs_carry: appends the unflagged contigs behind the nj merged ones already in the *2 arrays."""
        torch = _torch()
        n = int(soff.shape[0]) - 1
        nkeep = int((flag == 0).sum())
        keepidx = torch.empty(max(nkeep, 1), dtype=torch.int32, device=self.device)
        tot = (C.c_uint64 * 2)()
        self._check(self.lib.mcom_contigs_carry(self._h, self._p(seq), self._p(soff, torch.int64), self._p(mem, torch.int64), self._p(moff, torch.int64), n,
                                                self._p(flag, torch.uint8), nj, nkeep, self._p(seq2), self._p(soff2, torch.int64), self._p(mem2, torch.int64),
                                                self._p(moff2, torch.int64), self._p(keepidx), tot))
        return keepidx[:nkeep], (int(tot[0]), int(tot[1]))

    def order_next(self, ord_, n: int, flag, first_new: int, nj: int):
        """mcom_order_next: the list of the next merge round.  ord_ int32 [n] or None (0 .. n-1), flag uint8 indexed by contig."""
        torch = _torch()
        self.lib.mcom_order_next.restype = C.c_int
        self.lib.mcom_order_next.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32, C.c_size_t, C.c_void_p, C.POINTER(C.c_uint64)]
        out = torch.empty(max(n + nj, 1), dtype=torch.int32, device=self.device)
        nk = C.c_uint64()
        self._check(self.lib.mcom_order_next(self._h, self._p(ord_, torch.int32) if ord_ is not None else None, n, self._p(flag, torch.uint8) if flag is not None else None,
                                             first_new, nj, self._p(out), C.byref(nk)))
        return out[: nj + int(nk.value)], int(nk.value)

    def contigs_gather(self, seq, soff, mem, moff, idx):
        """mcom_contigs_gather: contigs idx[...] of a set as a set of their own.  Returns (seq2, soff2, mem2, moff2)."""
        torch = _torch()
        self.lib.mcom_contigs_gather.restype = C.c_int
        self.lib.mcom_contigs_gather.argtypes = [C.c_void_p] * 6 + [C.c_size_t] + [C.c_void_p] * 4 + [C.POINTER(C.c_uint64)]
        k = int(idx.shape[0])
        seq2 = torch.empty(int(seq.shape[0]) + 16, dtype=torch.uint8, device=self.device)
        mem2 = torch.empty(int(mem.shape[0]) + 1, dtype=torch.int64, device=self.device)
        soff2 = torch.empty(k + 1, dtype=torch.int64, device=self.device); moff2 = torch.empty(k + 1, dtype=torch.int64, device=self.device)
        tot = (C.c_uint64 * 2)()
        self._check(self.lib.mcom_contigs_gather(self._h, self._p(seq), self._p(soff, torch.int64), self._p(mem, torch.int64), self._p(moff, torch.int64),
                                                 self._p(idx, torch.int32), k, self._p(seq2), self._p(soff2), self._p(mem2), self._p(moff2), tot))
        return seq2[: int(tot[0])], soff2, mem2[: int(tot[1])], moff2

    def offsets_append(self, rel, base: int, dst, at: int):
        """mcom_offsets_append / _u32: dst[at + j] = base + rel[j] for the n + 1 entries of rel."""
        torch = _torch()
        n = int(rel.shape[0]) - 1
        if rel.dtype == torch.int64:
            self.lib.mcom_offsets_append.restype = C.c_int; self.lib.mcom_offsets_append.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p]
            self._check(self.lib.mcom_offsets_append(self._h, self._p(rel), n, base, C.c_void_p(dst.data_ptr() + 8 * at)))
        else:
            self.lib.mcom_offsets_append_u32.restype = C.c_int; self.lib.mcom_offsets_append_u32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]
            self._check(self.lib.mcom_offsets_append_u32(self._h, self._p(rel, torch.int32), n, base, C.c_void_p(dst.data_ptr() + 4 * at)))

    def members_finalize(self, mem, moff, passes, key_bits: int):
        """mcom_members_finalize.  mem int64 [M], moff int64 [n+1]; passes: list of (contig int32 [k], member int64 [k]).
        Returns (mem2 int64, moff2 int64 [n+1])."""
        torch = _torch()
        n = int(moff.shape[0]) - 1
        m = len(passes)
        total = int(mem.shape[0]) + sum(int(c.shape[0]) for c, _ in passes)
        mem2 = torch.empty(max(total, 1), dtype=torch.int64, device=self.device)
        moff2 = torch.empty(n + 1, dtype=torch.int64, device=self.device)
        ac = (C.c_void_p * max(m, 1))(*[self._p(c, torch.int32).value if c.shape[0] else None for c, _ in passes])
        am = (C.c_void_p * max(m, 1))(*[self._p(v, torch.int64).value if v.shape[0] else None for _, v in passes])
        an = (C.c_uint64 * max(m, 1))(*[int(c.shape[0]) for c, _ in passes])
        self._check(self.lib.mcom_members_finalize(self._h, self._p(mem, torch.int64), self._p(moff, torch.int64), n, int(mem.shape[0]), ac, am, an, m, key_bits,
                                                   self._p(mem2), self._p(moff2)))
        return mem2[:total], moff2

    def compact_live(self, ids, flag):
        """mcom_compact_live: the ids whose flag is zero, in order."""
        torch = _torch()
        n = int(ids.shape[0])
        out = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        cnt = C.c_uint64()
        self._check(self.lib.mcom_compact_live(self._h, self._p(ids, torch.int32), self._p(flag, torch.uint8), n, self._p(out), C.byref(cnt)))
        return out[: int(cnt.value)]

    def window_layout(self, soff, L: int):
        """mcom_window_layout.  Returns (woff int64 [n+1], n_windows, longest contig)."""
        torch = _torch()
        n = int(soff.shape[0]) - 1
        woff = torch.empty(n + 1, dtype=torch.int64, device=self.device)
        nw, ml = C.c_uint64(), C.c_uint64()
        self._check(self.lib.mcom_window_layout(self._h, self._p(soff, torch.int64), n, L, self._p(woff), C.byref(nw), C.byref(ml)))
        return woff, int(nw.value), int(ml.value)

    def records_carry(self, rec, roff, keepidx, first_id: int, base: int, rec2, roff2):
        torch = _torch()
        tot = C.c_uint64()
        self._check(self.lib.mcom_records_carry(self._h, self._p(rec), self._p(roff, torch.int32), self._p(keepidx, torch.int32), int(keepidx.shape[0]), first_id, base,
                                                self._p(rec2), int(rec2.shape[0]), self._p(roff2, torch.int32), C.byref(tot)))
        return int(tot.value)

    def resketch_merged(self, jobs, soff, rec, roff, seq2, soff2, merged_chars: int, w: int, k: int):
        """mcom_resketch_merged.  Returns (roff2 int32 [nj+1], records int64 [total,2], bases sketched)."""
        torch = _torch()
        nj = int(jobs.shape[0])
        roff2 = torch.empty(nj + 1, dtype=torch.int32, device=self.device)
        tot, sk = C.c_uint64(), C.c_uint64()
        cap = max(1024, merged_chars // 8 + nj)
        while True:
            out = self.empty_records(cap)
            rc = self.lib.mcom_resketch_merged(self._h, self._p(jobs, torch.int32), nj, self._p(soff), self._p(rec), self._p(roff, torch.int32),
                                               self._p(seq2), self._p(soff2), merged_chars, w, k, self._p(roff2), self._p(out), cap,
                                               C.byref(tot), C.byref(sk))
            if rc == -4 and int(tot.value) > cap:
                cap = int(tot.value)
                continue
            self._check(rc)
            return roff2, out[: int(tot.value)], int(sk.value)

    def claim_pairs(self, pairs, n_contigs: int, max_rounds: int = 4096):
        """mcom_claim_pairs.  pairs: int64 [n, 2] records in visiting order.  Returns (jobs int32 [nj, 4], flag uint8 [n_contigs], rounds)."""
        torch = _torch()
        n = int(pairs.shape[0])
        jobs = torch.empty((max(n_contigs // 2 + 1, 1), 4), dtype=torch.int32, device=self.device)
        flag = torch.empty(max(n_contigs, 1), dtype=torch.uint8, device=self.device)
        nj, rounds = C.c_uint64(), C.c_int()
        self._check(self.lib.mcom_claim_pairs(self._h, self._p(pairs), n, n_contigs, max_rounds, self._p(jobs), self._p(flag), C.byref(nj), C.byref(rounds)))
        return jobs[: nj.value], flag[:n_contigs], rounds.value

    def claims_resolve(self, claim, rids, n_contigs: int, flag):
        """mcom_claims_resolve.  flag is updated in place; returns (contig int32 [nwon], member int64 [nwon])."""
        torch = _torch()
        n = int(claim.shape[0])
        ac = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        am = torch.empty(max(n, 1), dtype=torch.int64, device=self.device)
        nw = C.c_uint64()
        self._check(self.lib.mcom_claims_resolve(self._h, self._p(claim, torch.int64), self._p(rids, torch.int32), n, n_contigs,
                                                 self._p(flag, torch.uint8), self._p(ac), self._p(am), C.byref(nw)))
        return ac[:nw.value], am[:nw.value]

    def synth_reads(self, seed: int, n_reads: int, L: int, coverage: int = 30, sub_rate: float = 0.005,
                    first: int = 0, count: int | None = None, pitch: int | None = None, genome: str = "uniform"):
        """mcom_synth_reads_genome; genome: "uniform" or "repeats" (device generator only)."""
        torch = _torch()
        if count is None:
            count = n_reads - first
        pitch = pitch or L
        out = torch.empty((count, pitch), dtype=torch.uint8, device=self.device)
        self._check(self.lib.mcom_synth_reads_genome(self._h, seed, n_reads, L, coverage, sub_rate, {"uniform": 0, "repeats": 1}[genome], first, count, self._p(out), pitch))
        return out


class Index:
    """Device-resident contig-minimizer index (mcom_idx)."""

    def __init__(self, ctx: Context, rec, k: int, b: int = 14):
        self.ctx = ctx
        self.n = int(rec.shape[0])
        h = C.c_void_p()
        ctx._check(ctx.lib.mcom_idx_build(ctx._h, ctx._p(rec), self.n, k, b, C.byref(h)))
        self._h = h

    def get(self, x):
        torch = _torch()
        n = int(x.shape[0])
        start = torch.empty(n, dtype=torch.int32, device=self.ctx.device)
        count = torch.empty(n, dtype=torch.int32, device=self.ctx.device)
        self.ctx._check(self.ctx.lib.mcom_idx_get(self.ctx._h, self._h, self.ctx._p(x, torch.int64), n, self.ctx._p(start), self.ctx._p(count)))
        return start, count

    def records(self):
        out = self.ctx.empty_records(max(self.n, 1))
        n = C.c_size_t()
        self.ctx._check(self.ctx.lib.mcom_idx_records(self.ctx._h, self._h, self.ctx._p(out), C.byref(n)))
        return out[: self.n]

    def close(self):
        if getattr(self, "_h", None):
            self.ctx.lib.mcom_idx_destroy(self.ctx._h, self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Dicts:
    """Device-resident Stage-2 dictionaries (mcom_dicts)."""

    def __init__(self, ctx: Context, sgbits, L: int, ininumdict: int = 0):
        torch = _torch()
        self.ctx = ctx
        self.n_sg = int(sgbits.shape[0])
        h = C.c_void_p()
        ctx._check(ctx.lib.mcom_dicts_build(ctx._h, ctx._p(sgbits, torch.int64), self.n_sg, L, ininumdict, C.byref(h)))
        self._h = h
        nd = C.c_int()
        nk = (C.c_uint32 * 16)(); mb = (C.c_uint32 * 16)()
        ctx._check(ctx.lib.mcom_dicts_info(self._h, C.byref(nd), nk, mb))
        self.nd = nd.value
        self.numkeys = [int(nk[i]) for i in range(self.nd)]
        self.maxbin = [int(mb[i]) for i in range(self.nd)]

    def lookup(self, dict_idx: int, keys):
        torch = _torch()
        n = int(keys.shape[0])
        start = torch.empty(n, dtype=torch.int32, device=self.ctx.device)
        count = torch.empty(n, dtype=torch.int32, device=self.ctx.device)
        self.ctx._check(self.ctx.lib.mcom_dicts_lookup(self.ctx._h, self._h, dict_idx, self.ctx._p(keys, torch.int64), n,
                                                       self.ctx._p(start), self.ctx._p(count)))
        return start, count

    def ids(self, dict_idx: int):
        torch = _torch()
        out = torch.empty(max(self.n_sg, 1), dtype=torch.int32, device=self.ctx.device)
        self.ctx._check(self.ctx.lib.mcom_dicts_ids(self.ctx._h, self._h, dict_idx, self.ctx._p(out)))
        return out[: self.n_sg]

    def close(self):
        if getattr(self, "_h", None):
            self.ctx.lib.mcom_dicts_free(self.ctx._h, self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def dict_layout(L: int, ininumdict: int = 0):
    """mcom_dict_layout (host only): (start[], end[]) of the dictionary keys."""
    lib = load_library()
    st = (C.c_int * 16)(); en = (C.c_int * 16)()
    nd = lib.mcom_dict_layout(L, ininumdict, st, en)
    if nd < 0:
        raise McomError("mcom_dict_layout failed")
    return [st[i] for i in range(nd)], [en[i] for i in range(nd)]


def pack_nt4(reads: np.ndarray) -> np.ndarray:
    """Host statement of the packed-row format of include/mcom.h for ACGT-only rows (tests, host driver)."""
    n, L = reads.shape
    W = words_per_read(L)
    code = ((reads >> 1) ^ (reads >> 2)) & 3
    out = np.zeros((n, W), dtype=np.uint64)
    for i in range(L):
        out[:, i // 32] |= code[:, i].astype(np.uint64) << np.uint64(2 * (i % 32))
    return out


def pack_contigs(refs) -> tuple:
    """Packs contig strings (bytes, ACGT) as mcom_realign_pass wants them: returns (cbits u64, coff u64, clen u32)."""
    coff = np.zeros(len(refs), dtype=np.uint64)
    clen = np.array([len(r) for r in refs], dtype=np.uint32)
    words = [(2 * int(l) + 63) // 64 + 1 for l in clen]          # one padding word after every contig
    total = int(sum(words)) + 1
    cbits = np.zeros(total, dtype=np.uint64)
    o = 0
    for i, r in enumerate(refs):
        coff[i] = o
        a = np.frombuffer(r, dtype=np.uint8)
        code = (((a >> 1) ^ (a >> 2)) & 3).astype(np.uint64)
        sh = (2 * (np.arange(len(a)) % 32)).astype(np.uint64)
        np.bitwise_or.at(cbits, o + np.arange(len(a)) // 32, code << sh)
        o += words[i]
    return cbits, coff, clen


def records_to_numpy(rec) -> np.ndarray:
    """Device [n,2] int64 record tensor -> numpy structured (x, y) uint64."""
    a = rec.detach().cpu().numpy().view(np.uint64)
    out = np.empty(a.shape[0], dtype=MM_DTYPE)
    out["x"] = a[:, 0]; out["y"] = a[:, 1]
    return out
