"""Stream files: P1 = the files our driver writes are byte-identical to the ones the
compiled reference wrote at one thread (tests/golden/streams_*.tar.gz); P2 = decoding them gives the input reads back."""
import gzip
import io
import os
import tarfile

import numpy as np
import pytest


def _golden_reads(golden_dir, tag):
    with gzip.open(os.path.join(golden_dir, tag + ".reads.gz"), "rb") as f:
        return f.read().split(b"\n")[:-1]


def _golden_streams(golden_dir, tag):
    with gzip.open(os.path.join(golden_dir, "streams_" + tag + ".tar.gz"), "rb") as g:
        tf = tarfile.open(fileobj=io.BytesIO(g.read()))
        return {m.name: tf.extractfile(m).read() for m in tf.getmembers()}


@pytest.mark.parametrize("tag", ["stages_L100", "stages_L150", "stages_L40"])
def test_our_decoder_inverts_the_reference_streams(golden_dir, tmp_path, tag):
    """CPU: the decoder (minicom_amd/host/mcom_decompress.cpp) applied to streams written by the reference itself."""
    from minicom_amd.pipeline import decompress
    d = tmp_path / "streams"; d.mkdir()
    for name, data in _golden_streams(golden_dir, tag).items():
        (d / name).write_bytes(data)
    out = tmp_path / "reads.txt"
    n = decompress(str(d), str(out))
    want = _golden_reads(golden_dir, tag)
    got = out.read_bytes().split(b"\n")[:-1]
    assert n == len(want) == len(got)
    assert sorted(got) == sorted(want)                      # default mode keeps the multiset, not the order (README.md:33-37)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["stages_L100", "stages_L150", "stages_L40"])
def test_stream_files_byte_identical_to_reference_and_lossless(golden_dir, tmp_path, tag):
    from minicom_amd.pipeline import Pipeline, decompress
    rows = _golden_reads(golden_dir, tag)
    reads = np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), len(rows[0])).copy()
    p = Pipeline(reads, host_threads=4)
    p.pre_process()
    d = tmp_path / "streams"; d.mkdir()
    p.cluster_dump(str(d))
    p.close()
    want = _golden_streams(golden_dir, tag)
    assert sorted(os.listdir(d)) == sorted(want)
    for name, data in want.items():
        assert (d / name).read_bytes() == data, name                       # P1
    out = tmp_path / "reads.txt"
    assert decompress(str(d), str(out)) == len(rows)
    assert sorted(out.read_bytes().split(b"\n")[:-1]) == sorted(rows)      # P2


@pytest.mark.gpu
def test_round_trip_at_a_size_no_fixture_covers():
    """P2 on 300 k device-generated reads with the plumbing extras mixed in on the host."""
    import tempfile
    from minicom_amd import synth
    from minicom_amd.pipeline import Pipeline, decompress
    reads = synth.synth_reads(2024, 300000, 100)
    extra = synth.synth_reads(2025, 20000, 100, plumbing=True)
    reads = np.concatenate([reads, extra])
    p = Pipeline(reads, host_threads=16)
    p.pre_process()
    with tempfile.TemporaryDirectory() as td:
        p.cluster_dump(td)
        out = os.path.join(td, "reads.txt")
        assert decompress(td, out) == reads.shape[0]
        got = np.frombuffer(open(out, "rb").read(), dtype=np.uint8).reshape(reads.shape[0], 101)[:, :100]
        a = np.sort(got.view("S100").ravel()); b = np.sort(np.ascontiguousarray(reads).view("S100").ravel())
        assert np.array_equal(a, b)
    p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("L", [64, 100, 150, 250])
def test_device_stream_encoder_equals_the_host_loop(tmp_path, L):
    """cluster_dump's default mode is made on the device (csrc/streams.hip: one thread per member writes print_encode's text,
    kthread_dump.c:198-221); the host loop of the -p / paired-end modes writes the same files (host_dump = 1).  200 k reads plus
    20 k with N, poly-A/T and N-heavy reads: every file byte-identical."""
    from minicom_amd import synth
    from minicom_amd.pipeline import Pipeline
    reads = np.concatenate([synth.synth_reads(3030 + L, 200000, L), synth.synth_reads(3031 + L, 20000, L, plumbing=True)])
    files = {}
    for how in (0, 1):
        p = Pipeline(reads, host_threads=8, host_dump=how)
        p.pre_process()
        d = tmp_path / f"streams{how}"; d.mkdir()
        p.cluster_dump(str(d))
        if how == 0:
            assert p.stat("dump_bytes") > 0                              # the device encoder ran
        p.close()
        files[how] = {f: (d / f).read_bytes() for f in sorted(os.listdir(d))}
    assert sorted(files[0]) == sorted(files[1])
    for name in files[1]:
        assert files[0][name] == files[1][name], name
    assert len(files[0]["dif_char.txt.0"]) > 100000 and b"N" in files[0]["dif_char.txt.0"]


# ---- order-preserving mode (minicom -p, the reference compiled with ORDER): SURVEY section 8f rank 4, single-end part
def _golden_order_streams(golden_dir, tag):
    with gzip.open(os.path.join(golden_dir, "streams_order_" + tag + ".tar.gz"), "rb") as g:
        tf = tarfile.open(fileobj=io.BytesIO(g.read()))
        return {m.name: tf.extractfile(m).read() for m in tf.getmembers()}


@pytest.mark.parametrize("tag", ["stages_L100", "stages_L150"])
def test_our_decoder_restores_the_original_order_from_the_reference_order_streams(golden_dir, tmp_path, tag):
    """CPU: mcomh_decompress_order applied to the -p file set written by the reference itself."""
    from minicom_amd.pipeline import decompress
    d = tmp_path / "streams"; d.mkdir()
    for name, data in _golden_order_streams(golden_dir, tag).items():
        (d / name).write_bytes(data)
    out = tmp_path / "reads.txt"
    n = decompress(str(d), str(out), order=True)
    want = _golden_reads(golden_dir, tag)
    assert n == len(want)
    assert out.read_bytes().split(b"\n")[:-1] == want            # the same reads in the same order


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["stages_L100", "stages_L150"])
def test_order_stream_files_byte_identical_to_reference_and_exact(golden_dir, tmp_path, tag):
    from minicom_amd.pipeline import Pipeline, decompress
    rows = _golden_reads(golden_dir, tag)
    reads = np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), len(rows[0])).copy()
    p = Pipeline(reads, host_threads=4)
    p.pre_process()
    d = tmp_path / "ordered"; d.mkdir()
    p.cluster_dump(str(d), order=True)
    want = _golden_order_streams(golden_dir, tag)
    assert sorted(os.listdir(d)) == sorted(want)
    for name, data in want.items():
        assert (d / name).read_bytes() == data, name                       # P1, order mode
    out = tmp_path / "reads.txt"
    assert decompress(str(d), str(out), order=True) == len(rows)
    assert out.read_bytes().split(b"\n")[:-1] == rows                      # P2, exact order
    # the default file set written afterwards from the same pipeline is unaffected by the order-mode dump
    d2 = tmp_path / "default"; d2.mkdir()
    p.cluster_dump(str(d2))
    p.close()
    for name, data in _golden_streams(golden_dir, tag).items():
        assert (d2 / name).read_bytes() == data, name


@pytest.mark.gpu
def test_order_round_trip_at_a_size_no_fixture_covers(tmp_path):
    from minicom_amd import synth
    from minicom_amd.pipeline import Pipeline, decompress
    reads = np.concatenate([synth.synth_reads(31, 200000, 150), synth.synth_reads(32, 15000, 150, plumbing=True)])
    p = Pipeline(reads, host_threads=16)
    p.pre_process()
    d = tmp_path / "s"; d.mkdir()
    p.cluster_dump(str(d), order=True)
    p.close()
    out = tmp_path / "reads.txt"
    assert decompress(str(d), str(out), order=True) == reads.shape[0]
    got = np.frombuffer(out.read_bytes(), dtype=np.uint8).reshape(reads.shape[0], 151)[:, :150]
    assert np.array_equal(got, reads)


# ---- paired end (minicompe, the reference compiled with _PE): SURVEY section 8f rank 4, second part --------------------
def _golden_pe_streams(golden_dir, tag):
    with gzip.open(os.path.join(golden_dir, "streams_pe_" + tag + ".tar.gz"), "rb") as g:
        tf = tarfile.open(fileobj=io.BytesIO(g.read()))
        return {m.name: tf.extractfile(m).read() for m in tf.getmembers()}


@pytest.mark.parametrize("tag", ["stages_L100", "stages_L150"])
def test_our_decoder_pairs_the_mates_from_the_reference_pe_streams(golden_dir, tmp_path, tag):
    """CPU: mcomh_decompress_pe applied to the paired-end file set written by the reference itself (file 1 = the first half
    of the fixture reads, file 2 = the second half)."""
    from minicom_amd.pipeline import decompress_pe
    d = tmp_path / "streams"; d.mkdir()
    for name, data in _golden_pe_streams(golden_dir, tag).items():
        (d / name).write_bytes(data)
    o1, o2 = tmp_path / "r1.txt", tmp_path / "r2.txt"
    rows = _golden_reads(golden_dir, tag)
    half = len(rows) // 2
    assert decompress_pe(str(d), str(o1), str(o2)) == half
    a, b = o1.read_bytes().split(b"\n")[:-1], o2.read_bytes().split(b"\n")[:-1]
    assert sorted(zip(a, b)) == sorted(zip(rows[:half], rows[half:2 * half]))      # every pair, still paired


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["stages_L100", "stages_L150"])
def test_pe_stream_files_byte_identical_to_reference_and_pairs_kept(golden_dir, tmp_path, tag):
    from minicom_amd import synth
    from minicom_amd.pipeline import Pipeline, decompress_pe
    rows = _golden_reads(golden_dir, tag)
    half = len(rows) // 2
    reads = np.frombuffer(b"".join(rows[:2 * half]), dtype=np.uint8).reshape(2 * half, len(rows[0])).copy()
    f1, f2 = str(tmp_path / "a_1.fastq"), str(tmp_path / "a_2.fastq")
    synth.write_fastq(f1, reads[:half]); synth.write_fastq(f2, reads[half:])
    p = Pipeline.from_fastq(f1, path2=f2, host_threads=4)
    assert p.n == 2 * half
    p.pre_process()
    d = tmp_path / "pe"; d.mkdir()
    p.cluster_dump(str(d), paired=True)
    p.close()
    want = _golden_pe_streams(golden_dir, tag)
    assert sorted(os.listdir(d)) == sorted(want)
    for name, data in want.items():
        assert (d / name).read_bytes() == data, name                       # P1, paired end
    o1, o2 = tmp_path / "r1.txt", tmp_path / "r2.txt"
    assert decompress_pe(str(d), str(o1), str(o2)) == half
    a, b = o1.read_bytes().split(b"\n")[:-1], o2.read_bytes().split(b"\n")[:-1]
    assert sorted(zip(a, b)) == sorted(zip(rows[:half], rows[half:2 * half]))   # P2: mate pairs


@pytest.mark.gpu
def test_pe_round_trip_at_a_size_no_fixture_covers(tmp_path):
    """Mates = the two ends of fragments: the second half of a synthetic set stands in for them."""
    from minicom_amd import synth
    from minicom_amd.pipeline import Pipeline, decompress_pe
    reads = np.concatenate([synth.synth_reads(41, 120000, 150), synth.synth_reads(42, 8000, 150, plumbing=True)])
    half = reads.shape[0] // 2
    p = Pipeline(reads, host_threads=16)
    p.pre_process()
    d = tmp_path / "s"; d.mkdir()
    p.cluster_dump(str(d), paired=True)
    p.close()
    o1, o2 = tmp_path / "r1.txt", tmp_path / "r2.txt"
    assert decompress_pe(str(d), str(o1), str(o2)) == half
    a = np.frombuffer(o1.read_bytes(), dtype=np.uint8).reshape(half, 151)[:, :150]
    b = np.frombuffer(o2.read_bytes(), dtype=np.uint8).reshape(half, 151)[:, :150]
    got = np.sort(np.concatenate([a, b], axis=1).view("S300").ravel())
    want = np.sort(np.ascontiguousarray(np.concatenate([reads[:half], reads[half:]], axis=1)).view("S300").ravel())
    assert np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["order", "paired"])
@pytest.mark.parametrize("L", [100, 150])
def test_device_encoders_of_the_order_and_paired_end_modes_equal_the_host_loop(tmp_path, mode, L):
    """Round 4: the -p and paired-end file sets are made on the device too (csrc/streams.hip: cmpcluster3 member order by two
    grouped sorts, ids.bin / ids.txt one thread per member, the pairing streams by scans; kthread_dump.c:33-138,
    kthread_dump_pe.c:35-120, :218-619).  The host loop (host_dump = 1) walks every base of every member as the reference does:
    every file of the two must be byte-identical on 200 k reads + 20 k with N, poly-A/T and N-heavy reads."""
    from minicom_amd import synth
    from minicom_amd.pipeline import Pipeline
    reads = np.concatenate([synth.synth_reads(5050 + L, 200000, L), synth.synth_reads(5051 + L, 20000, L, plumbing=True)])
    files = {}
    for how in (0, 1):
        p = Pipeline(reads, host_threads=8, host_dump=how)
        p.pre_process()
        d = tmp_path / f"streams{how}"; d.mkdir()
        p.cluster_dump(str(d), order=mode == "order", paired=mode == "paired")
        if how == 0:
            assert p.stat("dump_bytes") > 0                              # the device encoder ran
        p.close()
        files[how] = {f: (d / f).read_bytes() for f in sorted(os.listdir(d))}
    assert sorted(files[0]) == sorted(files[1])
    for name in files[1]:
        assert files[0][name] == files[1][name], name
    assert len(files[0]["ids.bin.0" if mode == "order" else "peids.bin.0"]) > 100000


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["default", "order", "paired"])
def test_one_stream_set_per_thread_as_the_reference_writes_them(tmp_path, mode):
    """The number of stream files is part of the format (info.txt = "L n_threads"; kthread_dump.c:370-379, minicom:110-146): the reference
    writes one set per thread and its decoder takes them in parallel (decompress.c:1248-1300).  stream_sets = 5 cuts the contigs into five
    runs: the per-member files of the five sets concatenate to the one-set files, info.txt says 5, our decoder gives the reads back, and so
    does THE REFERENCE'S OWN decoder (oracle/_ref/L150*/decompress, built from /root/reference/src by oracle/build_ref.sh: test
    infrastructure) -- in the reads' original order for -p, mate beside mate for the paired-end mode."""
    import subprocess
    from minicom_amd import synth
    from minicom_amd.pipeline import Pipeline, decompress, decompress_pe
    L = 150
    reads = np.concatenate([synth.synth_reads(909, 120000, L), synth.synth_reads(910, 6000, L, plumbing=True)])
    n = reads.shape[0]
    half = n // 2
    kw = {"order": mode == "order", "paired": mode == "paired"}
    dirs = {}
    for T in (1, 5):
        p = Pipeline(reads, host_threads=8, stream_sets=T)
        p.pre_process()
        d = tmp_path / f"sets{T}"; d.mkdir()
        p.cluster_dump(str(d), **kw)
        p.close()
        dirs[T] = d
    one, five = dirs[1], dirs[5]
    assert (five / "info.txt").read_text().split()[:2] == [str(L), "5"]
    per_member = ["beg_pos.bin", "dif_char.txt"] + (["ids.bin"] if mode == "order" else ["ids.txt", "peids.bin"] if mode == "paired" else [])
    for name in per_member:
        assert b"".join((five / f"{name}.{t}").read_bytes() for t in range(5)) == (one / f"{name}.0").read_bytes(), name
    for name in ("single.seq", "AA.txt", "TT.txt", "NN.txt", "single_N.seq"):
        assert (five / name).read_bytes() == (one / name).read_bytes(), name
    assert all((five / f"ref.bin.{t}").stat().st_size > 0 for t in range(5))

    def check(rows1, rows2=None):
        if mode == "order":
            assert rows1 == [r.tobytes() for r in reads]
        elif mode == "paired":
            assert sorted(zip(rows1, rows2)) == sorted((reads[i].tobytes(), reads[half + i].tobytes()) for i in range(half))
        else:
            assert sorted(rows1) == sorted(r.tobytes() for r in reads)
    # our decoder
    o1, o2 = tmp_path / "o1.txt", tmp_path / "o2.txt"
    if mode == "paired":
        assert decompress_pe(str(five), str(o1), str(o2)) == half
        check(o1.read_bytes().split(b"\n")[:-1], o2.read_bytes().split(b"\n")[:-1])
    else:
        assert decompress(str(five), str(o1), order=mode == "order") == n
        check(o1.read_bytes().split(b"\n")[:-1])
    # the reference's decoder, when its build travelled with the repo
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    variant = {"default": "L150", "order": "L150_order", "paired": "L150_pe"}[mode]
    exe = os.path.join(root, "oracle", "_ref", variant, "decompress")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref is not built here")
    r1, r2 = tmp_path / "r1.txt", tmp_path / "r2.txt"
    subprocess.run([exe, str(five), str(r1), "true" if mode == "paired" else "false", "true" if mode == "order" else "false", "4", str(r2)], check=True, cwd=str(tmp_path),
                   stdout=subprocess.DEVNULL)
    if mode == "order":
        check(r1.read_bytes().split(b"\n")[:-1])
    else:
        # the default and paired-end decoders leave their pieces in the folder; the reference's script concatenates them (minicom:386-396: with single_N.seq in the default mode; decompress.c:1296-1308)
        parts = [five / "aatt.fasta"] + ([five / "single_N.seq"] if mode == "default" else []) + [five / "single_dec.fasta"] + [five / f"result_{t}.seq" for t in range(5)]
        rows = b"".join(q.read_bytes() for q in parts if q.exists()).split(b"\n")[:-1]
        if mode == "paired":
            check(rows, r2.read_bytes().split(b"\n")[:-1])
        else:
            check(rows)


@pytest.mark.gpu
@pytest.mark.parametrize("L", [100, 150])
def test_order_and_paired_end_round_trips_at_the_edges_of_the_domain(tmp_path, L):
    """`minicom -p` and `minicompe` on read sets at the edges: two reads, copies of one read, homopolymers and all-N reads among ordinary ones,
    a pair of identical mates: the order-preserving set decodes to the reads in their order, the paired-end set to every mate pair."""
    from minicom_amd import synth
    from minicom_amd.pipeline import Pipeline, decompress, decompress_pe
    base = synth.synth_reads(900 + L, 300, L)
    sets = {"two": base[:2].copy(), "copies_of_one": np.repeat(base[:1], 64, axis=0),
            "specials": np.concatenate([np.full((6, L), ord("A"), dtype=np.uint8), np.full((4, L), ord("N"), dtype=np.uint8), base[:90], np.full((6, L), ord("T"), dtype=np.uint8), base[90:184]]),
            "identical_mates": np.concatenate([base[:100], base[:100]])}
    for name, reads in sets.items():
        reads = np.ascontiguousarray(reads)
        n = reads.shape[0]
        p = Pipeline(reads, host_threads=2); p.pre_process()
        d = tmp_path / ("o_%s_%d" % (name, L)); d.mkdir()
        p.cluster_dump(str(d), order=True)
        out = tmp_path / ("o_%s_%d.txt" % (name, L))
        assert decompress(str(d), str(out), order=True) == n, name
        got = np.frombuffer(out.read_bytes(), dtype=np.uint8).reshape(n, L + 1)[:, :L]
        assert np.array_equal(got, reads), name
        half = n // 2
        d2 = tmp_path / ("p_%s_%d" % (name, L)); d2.mkdir()
        p.cluster_dump(str(d2), paired=True)
        p.close()
        o1, o2 = tmp_path / ("p1_%s_%d.txt" % (name, L)), tmp_path / ("p2_%s_%d.txt" % (name, L))
        assert decompress_pe(str(d2), str(o1), str(o2)) == half, name
        a = np.frombuffer(o1.read_bytes(), dtype=np.uint8).reshape(half, L + 1)[:, :L]
        b = np.frombuffer(o2.read_bytes(), dtype=np.uint8).reshape(half, L + 1)[:, :L]
        got = np.sort(np.concatenate([a, b], axis=1).view("S%d" % (2 * L)).ravel())
        want = np.sort(np.ascontiguousarray(np.concatenate([reads[:half], reads[half:2 * half]], axis=1)).view("S%d" % (2 * L)).ravel())
        assert np.array_equal(got, want), name

