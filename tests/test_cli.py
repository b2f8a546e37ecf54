"""The command-line drop-in surface: `bin/minicom` with the reference script's flags and output
names, over the three executables the script runs (`minicomsg IN OUTDIR`, `minicompe IN1 IN2 OUTDIR`,
`decompress DIR OUT pe order nthr [OUT2]`; reference minicom:106, :229, :383)."""
import gzip
import io
import os
import subprocess
import tarfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bin")


def _golden_reads(golden_dir, tag):
    with gzip.open(os.path.join(golden_dir, tag + ".reads.gz"), "rb") as f:
        return f.read().split(b"\n")[:-1]


def _golden_streams(golden_dir, tag):
    with gzip.open(os.path.join(golden_dir, "streams_" + tag + ".tar.gz"), "rb") as g:
        tf = tarfile.open(fileobj=io.BytesIO(g.read()))
        return {m.name: tf.extractfile(m).read() for m in tf.getmembers()}


def _write_fastq(path, rows):
    with open(path, "wb") as f:
        for i, r in enumerate(rows):
            f.write(b"@r%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)))


def _run(cmd, cwd):
    p = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-3000:]
    return p.stdout.decode(errors="replace")


# ---- CPU: the decoder side needs no GPU ----------------------------------------------------------------------------
@pytest.mark.parametrize("tag,pe,order", [("stages_L100", "false", "false"), ("order_stages_L100", "false", "true"), ("pe_stages_L100", "true", "false")])
def test_decompress_executable_on_reference_streams(golden_dir, tmp_path, tag, pe, order):
    d = tmp_path / "streams"; d.mkdir()
    for name, data in _golden_streams(golden_dir, tag).items():
        (d / name).write_bytes(data)
    rows = _golden_reads(golden_dir, "stages_L100")
    out = _run([os.path.join(BIN, "decompress"), str(d), "out1.txt", pe, order, "4", "out2.txt"], tmp_path)
    a = (tmp_path / "out1.txt").read_bytes().split(b"\n")[:-1]
    if pe == "true":
        half = len(rows) // 2
        b = (tmp_path / "out2.txt").read_bytes().split(b"\n")[:-1]
        assert out.split()[0] == str(half)
        assert sorted(zip(a, b)) == sorted(zip(rows[:half], rows[half:2 * half]))
    elif order == "true":
        assert a == rows                                                   # -p: the original order
    else:
        assert sorted(a) == sorted(rows)


def test_decompress_executable_rejects_a_corrupt_stream_set(golden_dir, tmp_path):
    """a hostile archive must give an error, not an over-read: an endless digit run, a character that is
    neither base nor digit, a truncated stream."""
    base = _golden_streams(golden_dir, "stages_L100")
    for name, mutate in (("dif_char.txt.0", lambda b: b"99999999999999999999\n" + b),
                         ("dif_char.txt.0", lambda b: b"4*7\n" + b),
                         ("AA.txt", lambda b: b"250C\n" + b),
                         ("ref.bin.0", lambda b: b[: len(b) // 2])):
        d = tmp_path / ("bad_" + name.replace(".", "_") + str(len(os.listdir(tmp_path)))); d.mkdir()
        for n, data in base.items():
            (d / n).write_bytes(mutate(data) if n == name else data)
        p = subprocess.run([os.path.join(BIN, "decompress"), str(d), str(d / "out.txt"), "false", "false", "1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert p.returncode == 1 and b"consistent" in p.stdout, name


def test_minicom_d_reads_an_archive_without_a_gpu(golden_dir, tmp_path):
    """`minicom -d X.minicom` -> X_dec.reads in the working directory (reference minicom:316-317)."""
    from minicom_amd import container
    d = tmp_path / "streams"; d.mkdir()
    for name, data in _golden_streams(golden_dir, "stages_L100").items():
        (d / name).write_bytes(data)
    container.pack(str(d), str(tmp_path / "sample_comp.minicom"), codec="xz")
    out = _run(["bash", os.path.join(BIN, "minicom"), "-d", "sample_comp.minicom", "-t", "2"], tmp_path)
    assert "The decompressed file: sample_comp_dec.reads" in out
    got = (tmp_path / "sample_comp_dec.reads").read_bytes().split(b"\n")[:-1]
    assert sorted(got) == sorted(_golden_reads(golden_dir, "stages_L100"))
    assert not (tmp_path / "sample_comp").exists()                         # the working directory is cleaned up (minicom:402)


def test_minicom_usage_and_bad_flags(tmp_path):
    p = subprocess.run(["bash", os.path.join(BIN, "minicom"), "-h"], stdout=subprocess.PIPE)
    assert p.returncode == 0 and b"-p \t\torder-preserving mode" in p.stdout and b"-s \t\tnumber of indexed substring" in p.stdout
    p = subprocess.run(["bash", os.path.join(BIN, "minicom"), "-x", "f"], stdout=subprocess.PIPE)
    assert p.returncode == 1 and b"Error parameters." in p.stdout


# ---- GPU: the compressor side --------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("threads", ["1", "4"])
@pytest.mark.parametrize("mode,tag,suffix", [("r", "stages_L100", "_comp"), ("p", "order_stages_L100", "_comp_order"), ("pe", "pe_stages_L100", "_comp_pe")])
def test_minicom_script_writes_the_reference_streams_and_reads_them_back(golden_dir, tmp_path, mode, tag, suffix, threads):
    """-t 1: the archive holds the stream files of the reference at one thread, byte for byte.  -t 4: four stream sets, as the reference writes
    with four threads (its own output is not reproducible then: the threads race for the contigs) -- info.txt says 4 and the reads come back."""
    from minicom_amd import container
    rows = _golden_reads(golden_dir, "stages_L100")
    half = len(rows) // 2
    if mode == "pe":
        _write_fastq(tmp_path / "s_1.fastq", rows[:half]); _write_fastq(tmp_path / "s_2.fastq", rows[half:2 * half])
        out = _run(["bash", os.path.join(BIN, "minicom"), "-1", "s_1.fastq", "-2", "s_2.fastq", "-t", threads], tmp_path)
        arch = tmp_path / ("s" + suffix + ".minicom")               # file_1.fastq -> file_comp_pe.minicom, as the reference (minicom:179)
    else:
        _write_fastq(tmp_path / "s.fastq", rows)
        out = _run(["bash", os.path.join(BIN, "minicom"), "-r", "s.fastq", "-t", threads] + (["-p"] if mode == "p" else []), tmp_path)
        arch = tmp_path / ("s" + suffix + ".minicom")
    assert "[Stage 1] Real time" in out and "[Stage 2] Real time" in out and "Compressed file:" in out
    assert arch.exists() and not (tmp_path / arch.name[: -len(".minicom")]).exists()
    # the archive holds the reference's stream files, byte for byte (entropy stage: xz here, bsc when it is installed)
    d = tmp_path / "unpacked"
    container.unpack(str(arch), str(d))
    want = _golden_streams(golden_dir, tag)
    want.pop("ids.txt.0", None)                                            # a temporary the script removes (minicom:242)
    if threads == "1":
        assert sorted(os.listdir(d)) == sorted(want)
        for name, data in want.items():
            assert (d / name).read_bytes() == data, name
    else:
        assert (d / "info.txt").read_text().split()[1] == threads and (d / "ref.bin.3").exists() and not (d / "ref.bin.4").exists()
    # and back
    out = _run(["bash", os.path.join(BIN, "minicom"), "-d", arch.name], tmp_path)
    base = arch.name[: -len(".minicom")]
    if mode == "pe":
        a = (tmp_path / (base + "_dec_1.reads")).read_bytes().split(b"\n")[:-1]
        b = (tmp_path / (base + "_dec_2.reads")).read_bytes().split(b"\n")[:-1]
        assert sorted(zip(a, b)) == sorted(zip(rows[:half], rows[half:2 * half]))
    else:
        got = (tmp_path / (base + "_dec.reads")).read_bytes().split(b"\n")[:-1]
        assert got == rows if mode == "p" else sorted(got) == sorted(rows)


@pytest.mark.gpu
def test_minicomsg_takes_the_script_flags(golden_dir, tmp_path):
    """Non-default parameters reach the pipeline: the stream files equal the reference built with the same flags
    (`-e 6 -m 4 -w 12 -g 9 -R 3 -S 5 -E 30 -s 4`, tests/golden/stages_L100_params)."""
    rows = _golden_reads(golden_dir, "stages_L100_params")
    _write_fastq(tmp_path / "s.fastq", rows)
    d = tmp_path / "out"; d.mkdir()
    _run([os.path.join(BIN, "minicomsg"), "s.fastq", "out", "-e", "6", "-m", "4", "-w", "12", "-g", "9", "-R", "3", "-S", "5", "-E", "30", "-s", "4"], tmp_path)
    from minicom_amd.pipeline import Pipeline
    reads = np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), len(rows[0])).copy()
    p = Pipeline(reads, e=6, m=4, w=12, cbthr=9, max_rounds=3, step=5, maxthr=30, numdict=4)
    p.pre_process()
    d2 = tmp_path / "out2"; d2.mkdir()
    p.cluster_dump(str(d2)); p.close()
    assert sorted(os.listdir(d)) == sorted(os.listdir(d2))
    for n in os.listdir(d):
        assert (d / n).read_bytes() == (d2 / n).read_bytes(), n
