"""The .minicom container (minicom_amd/container.py): layout of the reference's script, built-in entropy stage."""
import gzip
import io
import os
import tarfile

import numpy as np
import pytest


def _unpack_golden(golden_dir, name, dst):
    with tarfile.open(os.path.join(golden_dir, name), "r:gz") as t:
        for m in t.getmembers():
            if m.isfile():
                with open(os.path.join(dst, os.path.basename(m.name)), "wb") as f:
                    f.write(t.extractfile(m).read())


@pytest.mark.parametrize("codec", ["xz", "bz2", "gz", "raw"])
@pytest.mark.parametrize("fixture,extra", [("streams_stages_L100.tar.gz", ()), ("streams_order_stages_L100.tar.gz", ("idsbin.tar",)),
                                           ("streams_pe_stages_L100.tar.gz", ("peidsbin.tar", "filebin.tar"))])
def test_pack_unpack_restores_the_reference_stream_files(golden_dir, tmp_path, codec, fixture, extra):
    """The reference's own stream files -> .minicom -> the same files, with the member names of the reference's script."""
    from minicom_amd import container
    src = tmp_path / "src"; src.mkdir()
    _unpack_golden(golden_dir, fixture, str(src))
    before = {n: open(src / n, "rb").read() for n in os.listdir(src)}
    arc = str(tmp_path / "x.minicom")
    sizes = container.pack(str(src), arc, codec=codec, threads=3)
    with tarfile.open(arc) as t:
        names = [m.name for m in t.getmembers()]
    assert names[0] == "info.txt"
    want = {"info.txt"} | {g + "." + codec for g in ("refbin.tar", "dirbin.tar", "begposbin.tar", "dif_char.tar") + tuple(extra)}
    want |= {s + "." + codec for s in container.SINGLES if s in before}
    assert set(names) == want == set(sizes)
    dst = tmp_path / "dst"
    kinds = container.unpack(arc, str(dst))
    assert kinds == {"order": "idsbin.tar" in extra, "paired": "filebin.tar" in extra}
    after = {n: open(dst / n, "rb").read() for n in os.listdir(dst)}
    assert after == {k: v for k, v in before.items() if not k.startswith("ids.txt.")} or after == before
    if codec in ("xz", "bz2"):                                       # (the outer tar's own blocking aside: the fixture is 23 KB)
        assert sum(sizes.values()) < sum(len(v) for v in before.values())


def test_pack_refuses_what_is_not_a_stream_directory(tmp_path):
    from minicom_amd import container
    with pytest.raises(FileNotFoundError):
        container.pack(str(tmp_path), str(tmp_path / "x.minicom"))
    with pytest.raises(ValueError):
        container.pack(str(tmp_path), str(tmp_path / "x.minicom"), codec="zip")


def test_bsc_codec_needs_the_binary(golden_dir, tmp_path):
    import shutil
    from minicom_amd import container
    if shutil.which("bsc"):
        pytest.skip("bsc is installed")
    src = tmp_path / "src"; src.mkdir()
    _unpack_golden(golden_dir, "streams_stages_L100.tar.gz", str(src))
    with pytest.raises(RuntimeError, match="bsc"):
        container.pack(str(src), str(tmp_path / "x.minicom"), codec="bsc")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["default", "order", "paired"])
def test_fastq_to_minicom_and_back(tmp_path, mode):
    """End to end on the GPU box: FASTQ (gz) -> .minicom -> reads; multiset equality, exact order with -p, pairs kept."""
    from minicom_amd import container, synth
    L, n = 100, 30000
    reads = np.concatenate([synth.synth_reads(77, n, L), synth.synth_reads(78, 2000, L, plumbing=True)])
    fq = str(tmp_path / "a.fastq")
    synth.write_fastq(fq, reads)
    arc = str(tmp_path / "a.minicom")
    out = str(tmp_path / "a.reads")
    if mode == "paired":
        mates = np.concatenate([synth.synth_reads(79, n, L), synth.synth_reads(80, 2000, L, plumbing=True)])
        fq2 = str(tmp_path / "b.fastq")
        synth.write_fastq(fq2, mates)
        sizes = container.compress_fastq(fq, arc, path2=fq2, codec="bz2")
        out2 = str(tmp_path / "b.reads")
        assert container.decompress_file(arc, out, out2) == len(reads)
        a = np.frombuffer(open(out, "rb").read(), dtype=np.uint8).reshape(len(reads), L + 1)[:, :L]
        b = np.frombuffer(open(out2, "rb").read(), dtype=np.uint8).reshape(len(reads), L + 1)[:, :L]
        got = sorted(zip(a.view("S%d" % L).ravel().tolist(), b.view("S%d" % L).ravel().tolist()))
        want = sorted(zip(np.ascontiguousarray(reads).view("S%d" % L).ravel().tolist(), np.ascontiguousarray(mates).view("S%d" % L).ravel().tolist()))
        assert got == want
        assert sizes["n_reads"] == 2 * len(reads)
    else:
        sizes = container.compress_fastq(fq, arc, order=(mode == "order"))
        assert container.decompress_file(arc, out) == len(reads)
        got = np.frombuffer(open(out, "rb").read(), dtype=np.uint8).reshape(len(reads), L + 1)[:, :L]
        if mode == "order":
            assert np.array_equal(got, reads)
        else:
            assert np.array_equal(np.sort(np.ascontiguousarray(got).view("S%d" % L).ravel()), np.sort(np.ascontiguousarray(reads).view("S%d" % L).ravel()))
        assert sizes["n_reads"] == len(reads)
    assert os.path.getsize(arc) < 0.45 * reads.size * (2 if mode == "paired" else 1)     # far below 2 bits per base + ids
