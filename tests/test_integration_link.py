"""INTEGRATION.md section A as a program: the reference's own main(), pre_process (its loop control and updateSingle), FASTQ reader and
stream writer -- unmodified objects compiled from the reference's sources (oracle/build_ref.sh) -- linked with oracle/shim_stages.cpp,
which defines kt_for_reads / kt_for_bucket / combine_cluster / realign_hash over libmcom_host.so and hands the results back through
reads_t.  The binary (oracle/_ref/<variant>/minicom_gpu, built in the build container, travels with the repo) must write the
reference's stream files byte for byte."""
import gzip
import io
import os
import subprocess
import tarfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tag,variant", [("stages_L100", "L100"), ("stages_L150", "L150")])
def test_reference_main_linked_against_the_library_writes_the_reference_streams(golden_dir, tmp_path, tag, variant):
    from minicom_amd import synth
    exe = os.path.join(ROOT, "oracle", "_ref", variant, "minicom_gpu")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/%s/minicom_gpu not built (needs the reference's sources: build container only)" % variant)
    with gzip.open(os.path.join(golden_dir, tag + ".reads.gz"), "rb") as f:
        rows = f.read().split(b"\n")[:-1]
    reads = np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), len(rows[0])).copy()
    fq = str(tmp_path / "in.fastq")
    synth.write_fastq(fq, reads)
    out = tmp_path / "out"; out.mkdir()
    cwd = tmp_path / "cwd"; (cwd / "output_ref").mkdir(parents=True)
    r = subprocess.run([exe, fq, str(out)], cwd=str(cwd), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "[Stage 1]" in r.stdout and "[Stage 2]" in r.stdout                  # the reference's own pre_process ran the stages
    with gzip.open(os.path.join(golden_dir, "streams_" + tag + ".tar.gz"), "rb") as g:
        tf = tarfile.open(fileobj=io.BytesIO(g.read()))
        want = {m.name: tf.extractfile(m).read() for m in tf.getmembers()}
    assert sorted(os.listdir(out)) == sorted(want)
    for name, data in want.items():
        assert (out / name).read_bytes() == data, name
