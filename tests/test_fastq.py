"""FASTQ / FASTA ingest (reference bseq.c:19-66 + kseq.h).  CPU: the parser against files
written here in the shapes kseq accepts.  GPU: file -> pinned chunks -> HBM -> pipeline equals the array path, and the
whole way FASTQ -> stream files reproduces the reference's files byte for byte."""
import gzip
import io
import os
import tarfile

import numpy as np
import pytest


def _write(path, text, gz=False):
    data = text if isinstance(text, bytes) else text.encode()
    if gz:
        with gzip.open(path, "wb") as f:
            f.write(data)
    else:
        with open(path, "wb") as f:
            f.write(data)


def _reads(seed, n, L):
    from minicom_amd import synth
    return synth.synth_reads(seed, n, L, plumbing=True)


def test_plain_and_gzip_fastq_round_trip(tmp_path):
    from minicom_amd import synth
    from minicom_amd.pipeline import read_fastq
    reads = _reads(5, 3000, 100)
    p = str(tmp_path / "a.fastq")
    synth.write_fastq(p, reads)
    assert np.array_equal(read_fastq(p), reads)
    with open(p, "rb") as f, gzip.open(p + ".gz", "wb") as g:
        g.write(f.read())
    assert np.array_equal(read_fastq(p + ".gz"), reads)
    assert np.array_equal(read_fastq(p, L=100), reads)


def test_shapes_kseq_accepts(tmp_path):
    """Multi-line FASTA, multi-line FASTQ, CRLF line ends, '@' and '+' as first quality characters, no final newline."""
    from minicom_amd.pipeline import read_fastq
    s = [b"ACGTACGTAC", b"TTTTGGGGCC", b"NNNNACGTTT"]
    p = str(tmp_path / "x")
    _write(p, b">r1 some comment\nACGTA\nCGTAC\n>r2\nTTTTGGGGCC\n>r3\nNNNN\nACGT\nTT")
    assert read_fastq(p).tobytes() == b"".join(s)
    _write(p, b"@r1\nACGTA\nCGTAC\n+r1\n@@@@@\n+++++\n@r2\r\nTTTTGGGGCC\r\n+\r\n@IIIIIIII+\r\n@r3\nNNNNACGTTT\n+\nIIIIIIIIII")
    assert read_fastq(p).tobytes() == b"".join(s)
    _write(p, b"\n\n@r1\nACGTACGTAC\n+\nIIIIIIIIII\n")
    assert read_fastq(p).tobytes() == s[0]
    _write(p, b"")
    assert read_fastq(p).shape[0] == 0


def test_errors_the_reference_exits_on(tmp_path):
    from minicom_amd.hip import McomError
    from minicom_amd.pipeline import read_fastq
    p = str(tmp_path / "bad.fastq")
    _write(p, b"@r1\nACGTACGTAC\n+\nIIIIIIIIII\n@r2\nACGTACGTA\n+\nIIIIIIIII\n")       # bseq.c:54-57: lengths differ
    with pytest.raises(McomError):
        read_fastq(p)
    _write(p, b"@r1\nACGTACGTAC\n+\nIIII\n")                                          # kseq: truncated quality
    with pytest.raises(McomError):
        read_fastq(p)
    with pytest.raises(McomError):
        read_fastq(str(tmp_path / "missing.fastq"))
    _write(p, b"@r1\nACGTACGTAC\n+\nIIIIIIIIII\n")
    with pytest.raises(McomError):
        read_fastq(p, L=12)


def _gzip_members(path, data, cuts, level=1):
    with open(path, "wb") as f:
        for a, b in zip([0] + cuts, cuts + [len(data)]):
            f.write(gzip.compress(data[a:b], level))


def _bgzf(path, data, block=60000):
    """what bgzip writes: members of at most 64 KB with their compressed size in a BC extra field, an empty member at the end"""
    import struct
    import zlib
    with open(path, "wb") as f:
        for a in range(0, len(data), block):
            blk = data[a:a + block]
            c = zlib.compressobj(1, zlib.DEFLATED, -15)
            comp = c.compress(blk) + c.flush()
            f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(blk), len(blk)))
        f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))


def test_gzip_files_of_many_members_are_read_by_all_cores_and_equal_the_plain_file(tmp_path):
    """Real inputs are .fastq.gz, and most writers of sequencing data write gzip files of many members (bgzip, bcl-convert, pigz -i):
    host/mcom_fastq_gz.cpp inflates the members in parallel and parses every piece of text from the first record boundary it finds by
    shape, a record that straddles two pieces being put together by the piece it starts in (the reference: gzread in one thread,
    bseq.c:19-36).  Members cut at arbitrary bytes (inside names, sequences, quality lines, between a newline and '@', at record ends),
    BGZF blocks, quality lines that start with '@' and '+': the rows equal the array the file was written from.  One member, a
    name that contains '@' at a cut, FASTA: the sequential reader takes over, same rows."""
    import re
    from minicom_amd import synth
    from minicom_amd.pipeline import read_fastq
    L = 100
    reads = np.concatenate([synth.synth_reads(5, 199000, L), synth.synth_reads(6, 1000, L, plumbing=True)])
    fq = str(tmp_path / "a.fastq")
    synth.write_fastq_fast(fq, reads)                                          # (tricky quality lines: every third starts with '@', every fifth with '+')
    data = open(fq, "rb").read()
    n = reads.shape[0]
    out = np.empty((n, L), dtype=np.uint8)

    def read_into(path, cap=n):
        import ctypes as C
        from minicom_amd.pipeline import load_host_library
        lib = load_host_library()
        Lc, cnt = C.c_int(0), C.c_size_t()
        rc = lib.mcomh_fastq_read(path.encode(), C.byref(Lc), out.ctypes.data_as(C.c_void_p), cap, C.byref(cnt))
        lib.mcomh_test_gz_items.restype = C.c_long
        items.append(lib.mcomh_test_gz_items())
        return rc, Lc.value, cnt.value
    items = []
    rng = np.random.default_rng(1)
    recs = [m.start() + 1 for m in re.finditer(b"\n@r", data)]
    for name, cuts in (("random", sorted({int(x) for x in rng.integers(1, len(data) - 1, 120)})),
                       ("record_ends", [recs[i] for i in range(500, len(recs), 1500)]),
                       ("around_newlines", sorted({recs[i] + d for i in range(700, len(recs), 2100) for d in (-1, 0, 1)}))):
        p = str(tmp_path / (name + ".fastq.gz"))
        _gzip_members(p, data, cuts)
        out[:] = 0
        assert read_into(p) == (0, L, n) and np.array_equal(out, reads), name
        assert items[-1] >= 8, (name, items)                                   # (the parallel route read it, in that many work items)
    p = str(tmp_path / "b.fastq.gz")
    _bgzf(p, data)
    out[:] = 0
    assert read_into(p) == (0, L, n) and np.array_equal(out, reads) and items[-1] >= 8
    rc, _, cnt = read_into(p, cap=n - 5)                                       # more reads than the caller has room for
    assert rc != 0
    # ONE member (what plain `gzip` writes) cannot be cut, but decoding and parsing are two stages: one thread decodes the member in pieces of
    # text (mcom_inflate_run), the workers parse the pieces -- the pieces are the work items.  With and without the last newline.
    for name, body in (("one", data), ("one_no_newline", data[:-1])):
        with gzip.open(str(tmp_path / (name + ".fastq.gz")), "wb", compresslevel=1) as g:
            g.write(body)
        out[:] = 0
        assert read_into(str(tmp_path / (name + ".fastq.gz"))) == (0, L, n) and np.array_equal(out, reads), name
        assert items[-1] >= 4, (name, items)
    assert np.array_equal(read_fastq(str(tmp_path / "one.fastq.gz")), reads)
    # a damaged member (one bit of its CRC-32) is an error whichever reader meets it
    dmg = bytearray(open(str(tmp_path / "one.fastq.gz"), "rb").read()); dmg[-6] ^= 1
    open(str(tmp_path / "dmg.fastq.gz"), "wb").write(bytes(dmg))
    assert read_into(str(tmp_path / "dmg.fastq.gz"))[0] != 0
    # a character outside ACGTN in a many-member file is an error, not a silent change
    bad = bytearray(data); bad[recs[70000] + 12] = ord("a")
    _gzip_members(str(tmp_path / "bad.fastq.gz"), bytes(bad), sorted({int(x) for x in rng.integers(1, len(data) - 1, 60)}))
    assert read_into(str(tmp_path / "bad.fastq.gz"))[0] != 0
    # multi-line FASTA in many members: not the four-line layout, the sequential reader reads it
    fa = b"".join(b">s%d\n%s\n%s\n" % (i, reads[i, :60].tobytes(), reads[i, 60:].tobytes()) for i in range(30000))
    _gzip_members(str(tmp_path / "x.fa.gz"), fa, sorted({int(x) for x in rng.integers(1, len(fa) - 1, 30)}))
    assert np.array_equal(read_fastq(str(tmp_path / "x.fa.gz")), reads[:30000])


def test_the_member_decoder_against_zlib():
    """host/mcom_inflate.cpp (the DEFLATE decoder of the member-parallel route: 64-bit bit buffer, one look-up per symbol, matches copied
    by words, CRC-32 by carry-less multiplication) against zlib as the checker: every compression level and strategy (stored, fixed and
    dynamic Huffman blocks, Huffman-only, run-length), empty / tiny / incompressible / highly repetitive / FASTQ-like inputs, output
    buffers that fit exactly and not at all, members followed by other bytes, every kind of truncation; and flipped bits and random bytes
    behind a valid header must give an error code -- the decoder never trusts its input (the reference reads through zlib's gzread,
    bseq.c:19-36, which has the same obligations)."""
    import ctypes as C
    import random
    import zlib
    from minicom_amd.pipeline import load_host_library
    lib = load_host_library()
    lib.mcomh_test_gunzip.restype = C.c_int
    lib.mcomh_test_gunzip.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]

    def gun(data, cap):
        out = C.create_string_buffer(max(cap, 1))
        u, n = C.c_size_t(), C.c_size_t()
        rc = lib.mcomh_test_gunzip(data, len(data), out, cap, C.byref(u), C.byref(n))
        return rc, u.value, out.raw[:n.value]

    def gz(data, level, strategy=zlib.Z_DEFAULT_STRATEGY, header=b""):
        co = zlib.compressobj(level, zlib.DEFLATED, 31, 9, strategy)
        return co.compress(data) + co.flush()
    rng = random.Random(5)
    r = np.random.default_rng(3)
    fq = b"".join(b"@A00:1:HX:1:1101:%d:%d 1:N:0:ACGT\n" % (1000 + i % 3000, i) + np.frombuffer(b"ACGT", dtype=np.uint8)[r.integers(0, 4, 150)].tobytes() + b"\n+\n" +
                  np.frombuffer(b"FFFFFFF:,#", dtype=np.uint8)[r.integers(0, 10, 150)].tobytes() + b"\n" for i in range(3000))
    cases = [b"", b"a", b"abc" * 5, bytes(1000), os.urandom(70000), b"ACGT" * 20000, fq, bytes(range(256)) * 300,
             b"".join(bytes([rng.randrange(4) + 65]) * rng.randrange(1, 400) for _ in range(3000))]
    for ci, data in enumerate(cases):
        for lvl in (0, 1, 4, 6, 9):
            for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                c = gz(data, lvl, strat)
                for cap in (len(data), len(data) + 1000):
                    assert gun(c + b"TRAIL", cap) == (0, len(c), data), (ci, lvl, strat, cap)
                if data:
                    assert gun(c, len(data) - 1)[0] == 1, (ci, lvl, strat)              # no room: said so, nothing written behind the buffer
                for cut in (1, 5, len(c) // 2, len(c) - 9, len(c) - 1):
                    if 0 < cut < len(c):
                        assert gun(c[:cut], len(data) + 100)[0] < 0, (ci, lvl, strat, cut)
    # header fields: extra, name, comment, header CRC
    with io.BytesIO() as b:
        with gzip.GzipFile(filename="reads.fastq", mode="wb", fileobj=b, compresslevel=6) as g:
            g.write(fq)
        named = b.getvalue()
    assert gun(named, len(fq)) == (0, len(named), fq)
    body = gz(fq, 6)[10:]
    fancy = b"\x1f\x8b\x08\x1e\0\0\0\0\0\x03" + b"\x05\x00ab\x01\x00z" + b"name\0" + b"comment\0" + b"\x12\x34" + body
    assert gun(fancy, len(fq)) == (0, len(fancy), fq)
    # damaged members: an error code every time (the CRC catches what still decodes)
    c = bytearray(gz(fq, 6))
    for _ in range(1500):
        d = bytearray(c)
        for _ in range(rng.randrange(1, 4)):
            d[rng.randrange(10, len(d))] ^= 1 << rng.randrange(8)
        assert gun(bytes(d), len(fq) + rng.randrange(0, 2000))[0] != 0
    for _ in range(1500):                                                          # random bytes as a deflate stream: any answer but a fault
        gun(b"\x1f\x8b\x08\x00\0\0\0\0\0\x03" + os.urandom(rng.randrange(1, 3000)), rng.randrange(0, 100000))


def test_the_member_decoder_under_the_sanitizers(tmp_path):
    """tests/fuzz_inflate.cpp: the decoder compiled with AddressSanitizer + UBSan (CPU build: the GPU pool has no sanitizer runs) against exact-size
    heap buffers, pieces of odd sizes, thousands of mutated members and random bytes -- an input is never trusted, no access leaves a buffer."""
    import shutil
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    host = os.path.join(here, "..", "minicom_amd", "host")
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "fuzz_inflate")
    b = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I" + host,
                        os.path.join(here, "fuzz_inflate.cpp"), os.path.join(host, "mcom_inflate.cpp"), "-lz", "-o", exe], capture_output=True, text=True)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("this g++ has no sanitizer runtime")
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert "exact ok 400" in r.stdout and "room 300" in r.stdout


@pytest.mark.gpu
def test_pipeline_from_a_gzip_file_of_many_members(tmp_path):
    """file -> HBM through the parallel gzip route -> pipeline = the pipeline over the array"""
    from minicom_amd import synth
    from minicom_amd.pipeline import Pipeline
    reads = np.concatenate([synth.synth_reads(81, 299000, 150), synth.synth_reads(82, 1000, 150, plumbing=True)])
    fq = str(tmp_path / "r.fastq")
    synth.write_fastq_fast(fq, reads)
    data = open(fq, "rb").read()
    rng = np.random.default_rng(3)
    _gzip_members(fq + ".gz", data, sorted({int(x) for x in rng.integers(1, len(data) - 1, 90)}))
    p = Pipeline.from_fastq(fq + ".gz", host_threads=4)
    assert (p.n, p.L) == reads.shape
    p.pre_process()
    q = Pipeline(reads, host_threads=4); q.pre_process()
    assert p.result_digest() == q.result_digest()
    p.close()
    # ... and through the two-stage route of a file of ONE member (one decoding thread, pieces of text parsed by the others, rows sent from
    # page-locked blocks): the same pipeline
    import ctypes as C
    from minicom_amd.pipeline import load_host_library
    with gzip.open(str(tmp_path / "one.fastq.gz"), "wb", compresslevel=1) as g:
        g.write(data)
    p = Pipeline.from_fastq(str(tmp_path / "one.fastq.gz"), host_threads=4)
    lib = load_host_library(); lib.mcomh_test_gz_items.restype = C.c_long
    assert lib.mcomh_test_gz_items() >= 4                                       # (the pieces: it did not go to the sequential reader)
    assert (p.n, p.L) == reads.shape
    p.pre_process()
    assert p.result_digest() == q.result_digest()
    p.close(); q.close()


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", [0, 700])
def test_pipeline_from_fastq_equals_pipeline_from_array(tmp_path, chunk):
    """chunk = 700 rows: dozens of chunk hand-overs and several growths of the device matrix."""
    from minicom_amd import synth
    from minicom_amd.pipeline import Pipeline
    reads = _reads(77, 30000, 150)
    path = str(tmp_path / "r.fastq.gz")
    buf = io.BytesIO()
    synth.write_fastq(str(tmp_path / "r.fastq"), reads)
    with open(str(tmp_path / "r.fastq"), "rb") as f, gzip.open(path, "wb") as g:
        g.write(f.read())
    a = Pipeline(reads, host_threads=4); a.pre_process()
    b = Pipeline.from_fastq(path, chunk_reads=chunk, host_threads=4); b.pre_process()
    assert (b.n, b.L) == reads.shape
    ca, cb = a.contigs(), b.contigs()
    assert len(ca) == len(cb) > 50
    assert all(r0 == r1 and np.array_equal(m0, m1) for (r0, m0), (r1, m1) in zip(ca, cb))
    for name in ("sg", "fpA", "fpT", "fpN", "allA", "allT", "allN", "Nfile"):
        assert np.array_equal(a.id_list(name), b.id_list(name)), name
    a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["stages_L100", "stages_L150"])
def test_fastq_to_stream_files_equals_the_reference(golden_dir, tmp_path, tag):
    """The reference's own flow, file in -> stream files out, on its fixture."""
    from minicom_amd import synth
    from minicom_amd.pipeline import Pipeline
    with gzip.open(os.path.join(golden_dir, tag + ".reads.gz"), "rb") as f:
        rows = f.read().split(b"\n")[:-1]
    reads = np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), len(rows[0])).copy()
    fq = str(tmp_path / "in.fastq")
    synth.write_fastq(fq, reads)
    p = Pipeline.from_fastq(fq, host_threads=2)
    p.pre_process()
    d = tmp_path / "streams"; d.mkdir()
    p.cluster_dump(str(d))
    p.close()
    with gzip.open(os.path.join(golden_dir, "streams_" + tag + ".tar.gz"), "rb") as g:
        tf = tarfile.open(fileobj=io.BytesIO(g.read()))
        want = {m.name: tf.extractfile(m).read() for m in tf.getmembers()}
    assert sorted(os.listdir(d)) == sorted(want)
    for name, data in want.items():
        assert (d / name).read_bytes() == data, name


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c0_streams_1m_100", "c0_streams_1m_150", "c0_streams_4m_150"])
def test_stream_files_equal_the_reference_run_on_configs0(golden_dir, tmp_path, name):
    """BASELINE configs[0] exactly (1 M x 100 bp, seed 1001), the 150-base shape at that size and at 4 M reads: the reference itself
    (oracle/_ref/L100|L150/minicom_bin, one thread, its whole timed region preprocess.c:137-234 and its writer) ran on these sets
    in the build container and the md5 of every stream file it wrote is the fixture (tests/golden/make_streams.py --c0).  File ->
    stream files here must give the same bytes: no oracle in between."""
    import hashlib
    import json
    from minicom_amd import synth
    from minicom_amd.pipeline import Pipeline
    path = os.path.join(golden_dir, name + ".md5.json")
    if not os.path.exists(path):
        pytest.skip(name + ": fixture not generated")
    doc = json.load(open(path))
    reads = synth.synth_reads(doc["seed"], doc["n"], doc["L"])
    fq = str(tmp_path / "in.fastq")
    synth.write_fastq_fast(fq, reads, tricky_quality=False)
    p = Pipeline.from_fastq(fq, host_threads=8)
    assert (p.n, p.L) == (doc["n"], doc["L"])
    p.pre_process()
    d = tmp_path / "streams"; d.mkdir()
    p.cluster_dump(str(d))
    p.close()
    assert sorted(os.listdir(d)) == sorted(doc["files"])
    for nm, want in doc["files"].items():
        data = (d / nm).read_bytes()
        assert len(data) == want["bytes"], nm
        assert hashlib.md5(data).hexdigest() == want["md5"], nm


@pytest.mark.gpu
def test_from_fastq_reports_unequal_lengths(tmp_path):
    from minicom_amd.hip import McomError
    from minicom_amd.pipeline import Pipeline
    p = str(tmp_path / "bad.fastq")
    _write(p, b"@r1\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n@r2\nACGT\n+\nIIII\n")
    with pytest.raises(McomError, match="Length of reads are different"):
        Pipeline.from_fastq(p)


@pytest.mark.gpu
def test_parallel_parser_of_four_line_fastq_equals_the_sequential_reader(tmp_path):
    """A plain four-line FASTQ file is mapped and parsed by all cores at once (mcom_fastq.cpp: record boundaries found by their
    shape); 400 k reads of 150 bases = 130 MB, record sizes varying with the name, quality lines that start with '@' and '+':
    the rows in HBM equal the array the file was written from, and the sequential reader's result."""
    import torch
    from minicom_amd import synth
    from minicom_amd.pipeline import Pipeline, read_fastq
    reads = np.concatenate([synth.synth_reads(71, 399000, 150), synth.synth_reads(72, 1000, 150, plumbing=True)])
    fq = str(tmp_path / "big.fastq")
    synth.write_fastq_fast(fq, reads)
    assert os.path.getsize(fq) > 100 << 20
    p = Pipeline.from_fastq(fq)
    assert (p.n, p.L) == reads.shape
    p.pre_process()
    q = Pipeline(reads); q.pre_process()
    assert p.result_digest() == q.result_digest()
    p.close(); q.close()
    assert np.array_equal(read_fastq(fq)[::997], reads[::997])
    # not the four-line shape (a blank line in the middle): the sequential reader takes over, same result
    with open(fq, "rb") as f:
        data = f.read()
    cut = data.index(b"\n@r200000\n") + 1
    open(fq, "wb").write(data[:cut] + b"\n" + data[cut:])
    p = Pipeline.from_fastq(fq)
    assert p.n == reads.shape[0]
    p.close()


def _write_fasta(path, reads, width, last_newline=True):
    """FASTA with names >r0, >r1 ... and the sequence in lines of `width` characters (0: one line)."""
    n, L = reads.shape
    w = width or L
    parts = []
    for i in range(n):
        row = reads[i].tobytes()
        parts.append(b">r%d some text\n" % i + b"\n".join(row[a:a + w] for a in range(0, L, w)) + b"\n")
    data = b"".join(parts)
    with open(path, "wb") as f:
        f.write(data if last_newline else data[:-1])


@pytest.mark.gpu
@pytest.mark.parametrize("L,width,last_newline", [(150, 0, True), (150, 60, True), (150, 70, False), (100, 7, True), (40, 1, True), (256, 80, True)])
def test_parallel_parser_takes_fasta_with_sequences_over_several_lines(tmp_path, L, width, last_newline):
    """bseq.c:38-66 reads FASTA through kseq.h as it reads FASTQ: a '>' line, then the sequence over any number of lines.  The one-pass packing
    parser (host/mcom_fastq.cpp) takes that shape too -- every core parses a piece, a sequence over several lines is put in a row first -- and
    gives the rows of the array the file was written from (the digest of the whole pipeline, and the statistics say that it was the parallel
    route: the sequential reader does not set t_fastq_upload); the sequential reader reads the same reads.  A '+' line in the file is not FASTA:
    that file goes to the sequential reader."""
    from minicom_amd import synth
    from minicom_amd.pipeline import Pipeline, read_fastq
    n = 120_000 if width != 1 else 30_000
    reads = np.concatenate([synth.synth_reads(81, n - 500, L), synth.synth_reads(82, 500, L, plumbing=True)])
    fa = str(tmp_path / "reads.fa")
    _write_fasta(fa, reads, width, last_newline)
    p = Pipeline.from_fastq(fa)
    assert (p.n, p.L) == reads.shape
    assert p.stat("t_fastq_upload") > 0                                     # the packing parser, not the sequential reader
    p.pre_process()
    q = Pipeline(reads); q.pre_process()
    assert p.result_digest() == q.result_digest()
    p.close(); q.close()
    assert np.array_equal(read_fastq(fa)[::97], reads[::97])
    if width == 60:
        data = open(fa, "rb").read()
        cut = data.index(b"\n>r60000 ") + 1
        open(fa, "wb").write(data[:cut] + b"@x\nACGT\n+\nIIII\n" + data[cut:])
        with pytest.raises(Exception):
            Pipeline.from_fastq(fa)                                        # (a read of another length: the sequential reader's message)
