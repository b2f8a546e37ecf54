"""File -> stream files, timed stage by stage (bench.py's `value_file_to_streams`, tools/e2e_bench.py).

The hot path's number (bench.py `value`) starts with the reads resident in HBM and ends with the contig set there.  What a user of
the `minicom` command waits for starts with a FASTQ file on disk and ends with the pre-entropy-coder stream files on disk
(ref.bin / beg_pos.bin / dir.bin / dif_char.txt / single.seq ...: the contents of the .minicom container before bsc / 7z / xz):
    parse + upload   mcomh_fastq_to_device   (mapped file, all cores, page-locked blocks straight to the rows' place in HBM)
    hot path         mcomh_pre_process       (Stage 1 + Stage 2)
    encode + write   mcomh_cluster_dump      (streams made on the device, csrc/streams.hip; file images copied back and written)
The reference binary (oracle/_ref, when present) runs the same way on a prefix of the same file: its whole process, file to
stream files, by the wall clock."""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tools/ -> repo root


def gzip_members(src: str, dst: str, member_bytes: int = 8 << 20, level: int = 1, threads: int = 16) -> int:
    """src -> dst as a gzip file of MANY members (what bgzip, bcl-convert and pigz -i write; `zcat` reads it like any gzip file): members
    of member_bytes of text, compressed by a pool of threads (zlib releases the interpreter lock).  Returns the number of members."""
    import zlib
    from concurrent.futures import ThreadPoolExecutor

    def one(block: bytes) -> bytes:
        c = zlib.compressobj(level, zlib.DEFLATED, 31)
        return c.compress(block) + c.flush()
    n = 0
    with open(src, "rb") as f, open(dst, "wb") as g, ThreadPoolExecutor(threads) as ex:
        while True:
            blocks = [b for b in (f.read(member_bytes) for _ in range(4 * threads)) if b]
            if not blocks:
                break
            for comp in ex.map(one, blocks):
                g.write(comp); n += 1
    return n


def gzip_one_member(src: str, dst: str, chunk_bytes: int = 8 << 20, level: int = 1, threads: int = 16) -> None:
    """src -> dst as a gzip file of ONE member (what plain `gzip` writes), made by many threads the way pigz makes it: every chunk becomes a
    raw deflate stream that ends on a byte boundary without a final block (Z_SYNC_FLUSH), the last one ends the stream, and the pieces
    one behind the other are one valid deflate stream; header and CRC-32 / ISIZE trailer around it."""
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor

    def one(arg):
        block, last = arg
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        return c.compress(block) + (c.flush(zlib.Z_FINISH) if last else c.flush(zlib.Z_SYNC_FLUSH))
    total = os.path.getsize(src)
    crc, done = 0, 0
    with open(src, "rb") as f, open(dst, "wb") as g, ThreadPoolExecutor(threads) as ex:
        g.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03")
        while done < total:
            blocks = [b for b in (f.read(chunk_bytes) for _ in range(4 * threads)) if b]
            args = []
            for b in blocks:
                done += len(b); args.append((b, done >= total))
            for comp in ex.map(one, args):
                g.write(comp)
            for b in blocks:
                crc = zlib.crc32(b, crc)
        g.write(struct.pack("<II", crc, total & 0xFFFFFFFF))


def file_to_streams(n: int, L: int, seed: int, host_threads: int = 16, ref_reads: int = 1_000_000, workdir: str | None = None, keep: bool = False, mode: str = "default",
                    gz: bool = False, gz_one_member: bool = False):
    """mode: "default" (multiset of reads), "order" (minicom -p), "paired" (minicompe: the first n / 2 reads are file 1, the others their
    mates in file 2); gz: the input is a .fastq.gz of many members (8 MB of text each), inflated and parsed by all cores
    (host/mcom_fastq_gz.cpp)"""
    import numpy as np
    import torch
    import minicom_amd
    from minicom_amd import synth
    from minicom_amd.pipeline import Pipeline
    td = tempfile.mkdtemp(prefix="mcom_e2e_", dir=workdir)
    try:
        ctx = minicom_amd.Context(0)
        fq = os.path.join(td, "reads.fastq")
        fq2 = os.path.join(td, "reads_2.fastq") if mode == "paired" else None
        if mode == "paired":
            n -= n & 1
        t = time.perf_counter()
        # the file is written in blocks of 4 M reads generated on the device (the generator of the benchmark)
        block = 4_000_000
        for path, a, b in ((fq, 0, n // 2 if fq2 else n), (fq2, n // 2, n)):
            if path is None:
                continue
            with open(path, "wb") as out:
                for lo in range(a, b, block):
                    cnt = min(block, b - lo)
                    part = ctx.synth_reads(seed, n, L, first=lo, count=cnt).cpu().numpy()
                    _append_fastq(out, part, lo - a)
                    del part
        ctx.close()
        gz_members = 0
        if gz:
            for path in (fq, fq2):
                if path:
                    if gz_one_member:
                        gzip_one_member(path, path + ".gz", threads=max(1, host_cores())); gz_members += 1
                    else:
                        gz_members += gzip_members(path, path + ".gz", threads=max(1, host_cores()))
                    os.remove(path)
            fq, fq2 = fq + ".gz", (fq2 + ".gz" if fq2 else None)
        t_write_input = time.perf_counter() - t
        size = os.path.getsize(fq) + (os.path.getsize(fq2) if fq2 else 0)
        out_dir = os.path.join(td, "streams"); os.makedirs(out_dir)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        p = Pipeline.from_fastq(fq, path2=fq2, host_threads=host_threads)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        p.pre_process()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        p.cluster_dump(out_dir, order=mode == "order", paired=mode == "paired")
        t3 = time.perf_counter()
        assert p.n == n and p.L == L
        stream_bytes = sum(os.path.getsize(os.path.join(out_dir, f)) for f in os.listdir(out_dir))
        res = {"mode": mode, "reads": n, "read_len": L, "fastq_bytes": size, "stream_bytes": stream_bytes,
               "seconds": {"parse_and_upload": round(t1 - t0, 3), "of_which_read_pack_upload": round(p.stat("t_fastq_upload") / 1e3, 3), "of_which_closing_the_gaps": round(p.stat("t_fastq_close_gaps") / 1e3, 3),
                           "pass_2_slowest_thread": {"setup": round(p.stat("t_fastq_setup_max") / 1e3, 3), "pack": round(p.stat("t_fastq_pack_max") / 1e3, 3), "wait_for_copies": round(p.stat("t_fastq_wait_max") / 1e3, 3)}, "hot_path": round(t2 - t1, 3), "encode_copy_write": round(t3 - t2, 3),
                           "of_which_device_encode_and_copy": round(p.stat("t_dump_gpu") / 1e3, 3), "of_which_file_writes": round(p.stat("t_dump_write") / 1e3, 3),
                           "total": round(t3 - t0, 3)},
               "value": round(n / (t3 - t0) / 1e6, 3), "unit": "Mreads/s",
               "fastq_GB_per_s": round(size / (t1 - t0) / 1e9, 2),
               "note": "FASTQ file (page cache) -> parse -> HBM -> Stage 1 + Stage 2 -> stream files written; the entropy coder (bsc / 7z / xz, external) is not part of it",
               "input_written_in_s": round(t_write_input, 1)}
        if gz and gz_one_member:
            res["gzip_members"] = 1
            res["fastq_text_GB_per_s"] = round(n * (2 * L + 6 + len(str(n))) / (t1 - t0) / 1e9, 2)
            res["note"] = ("the same from a .fastq.gz of ONE gzip member (what plain `gzip` writes; zlib level 1): the member cannot be cut, so one thread decodes it in pieces "
                           "of text (host/mcom_inflate.cpp) and the other cores parse the pieces; gzread + one parsing thread, as the reference reads, ran at ~1.5 Mreads/s")
        elif gz:
            res["gzip_members"] = gz_members
            res["fastq_text_GB_per_s"] = round(n * (2 * L + 6 + len(str(n))) / (t1 - t0) / 1e9, 2)
            res["note"] = ("the same from a .fastq.gz of %d gzip members (8 MB of text each, zlib level 1; quality lines are all 'I', so this file inflates "
                           "faster than sequencer output would): members shared out over all cores, decoded by the library's own DEFLATE decoder (host/mcom_inflate.cpp, "
                           "~2 x zlib per core; CRC-32 checked) and parsed there, rows sent as characters from page-locked blocks; a gzip file of ONE member "
                           "is decoded by one thread and parsed by the others: value_file_to_streams_gz_one_member" % gz_members)
        p.close()
        res["reference"] = _reference_on_prefix(fq, td, n, L, ref_reads) if mode == "default" and ref_reads else None
        return res
    finally:
        if not keep:
            shutil.rmtree(td, ignore_errors=True)


def host_cores() -> int:
    """CPUs this process may really use: the affinity mask cut by the cgroup's CPU quota (the GPU box gives a job a share of the host:
    256 cores visible, cpu.max = 16 CPUs)"""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:                                                           # noqa: BLE001
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max" and int(period) > 0:
            n = min(n, max(1, -(-int(quota) // int(period))))
    except Exception:                                                           # noqa: BLE001
        pass
    return n


def reference_binaries(L: int):
    """[(path, variant, threads, march)] of the reference builds to try, best first: the most threads the box has cores for
    (oracle/build_ref.sh compiles the thread count in, as the reference's own script does: minicom:56-91), then the x86-64-v2
    build of the same should the box lack AVX2"""
    cores = host_cores()
    out = []
    for t in (64, 32, 16, 8, 1):
        for suffix in ("", "_v2"):
            variant = (f"L{L}_t{t}" if t > 1 else f"L{L}") + suffix
            d = os.path.join(ROOT, "oracle", "_ref", variant)
            exe = os.path.join(d, "minicom_bin")
            if os.path.exists(exe) and t <= cores:
                march = open(os.path.join(d, ".march")).read().strip() if os.path.exists(os.path.join(d, ".march")) else "x86-64-v2"
                out.append((exe, variant, t, march))
    return out


def reference_binary(L: int):
    """(path, variant, threads) of the first of reference_binaries, or None"""
    c = reference_binaries(L)
    return c[0][:3] if c else None


def _append_fastq(out, reads, first_id):
    """four-line records @r<id>, quality 'I' (synth.write_fastq_fast's layout, appended block by block)"""
    import numpy as np
    m, L = reads.shape
    lo = first_id
    end = first_id + m
    while lo < end:
        digits = len(str(lo))
        hi = min(end, 10 ** digits)
        k = hi - lo
        rec = np.empty((k, 2 + digits + 1 + L + 3 + L + 1), dtype=np.uint8)
        rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
        ids = np.arange(lo, hi, dtype=np.int64)
        for d in range(digits):
            rec[:, 2 + digits - 1 - d] = ord("0") + (ids // 10 ** d) % 10
        o = 2 + digits
        rec[:, o] = 10
        rec[:, o + 1:o + 1 + L] = reads[lo - first_id:hi - first_id]
        rec[:, o + 1 + L] = 10; rec[:, o + 2 + L] = ord("+"); rec[:, o + 3 + L] = 10
        rec[:, o + 4 + L:o + 4 + 2 * L] = ord("I")
        rec[:, o + 4 + 2 * L] = 10
        out.write(rec.tobytes())
        lo = hi


def _reference_on_prefix(fq, td, n, L, ref_reads):
    """the reference's own binary (built by oracle/build_ref.sh in the build container; it travels as a built file), whole process"""
    found = reference_binary(L)
    if not found:
        return None
    exe, variant, threads = found
    m = min(n, ref_reads)
    pre = os.path.join(td, "prefix.fastq")
    with open(fq, "rb") as f, open(pre, "wb") as g:                            # the first m records
        need = 4 * m
        buf = b""
        while need > 0:
            chunk = f.read(64 << 20)
            if not chunk:
                break
            lines = chunk.count(b"\n")
            if lines <= need:
                g.write(chunk); need -= lines
            else:
                pos = -1
                for _ in range(need):
                    pos = chunk.index(b"\n", pos + 1)
                g.write(chunk[:pos + 1]); need = 0
    out = os.path.join(td, "ref_out"); os.makedirs(out)
    cwd = os.path.join(td, "ref_cwd"); os.makedirs(os.path.join(cwd, "output_ref"))
    t = time.perf_counter()
    try:
        pr = subprocess.run([exe, pre, out], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    except subprocess.TimeoutExpired:
        return {"value": None, "note": f"the reference did not finish {m} reads in 900 s"}
    dt = time.perf_counter() - t
    if pr.returncode != 0:
        return {"value": None, "note": f"the reference binary failed ({pr.returncode})"}
    st = [float(x) for x in re.findall(r"\[Stage \d\] Real time: ([\d.]+)", pr.stdout.decode())]
    return {"value": round(m / dt / 1e6, 4), "unit": "Mreads/s", "reads": m, "threads": threads, "seconds_total": round(dt, 2),
            "seconds_stage1_plus_stage2": round(sum(st), 2) if len(st) == 2 else None,
            "note": f"oracle/_ref/{variant}/minicom_bin on the first {m} reads of the same file, whole process (load + Stage 1 + Stage 2 + cluster_dump), wall clock"}
