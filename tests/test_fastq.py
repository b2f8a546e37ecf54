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
