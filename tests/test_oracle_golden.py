"""Pins the CPU oracle (oracle/mcom_oracle.c) against outputs of the compiled reference.

The fixtures under tests/golden/ were printed by the reference's own functions (oracle/refdump.cpp
linked against the reference objects; generator: tests/golden/make_golden.py).
"""
import gzip
import json
import os

import numpy as np
import pytest

import oracle


@pytest.fixture(scope="module")
def kat(golden_dir):
    with gzip.open(os.path.join(golden_dir, "kat.json.gz"), "rt") as f:
        return json.load(f)


def test_hash64_known_answers(kat):
    assert len(kat["H64"]) > 1000
    ks = set()
    for t in kat["H64"]:
        k = t["k"]; ks.add(k)
        assert oracle.hash64(t["kmer"], (1 << (2 * k)) - 1) == t["hash"]
    assert {31, 16} <= ks


def test_sketch_two_matches_reference(kat):
    assert len(kat["S2"]) > 1000
    for t in kat["S2"]:
        x, y = oracle.sketch_two(t["seq"].encode(), t["k"], t["rid"])
        assert (x, y) == (t["x"], t["y"]), t


def test_sketch_lh_ori_matches_reference(kat):
    for t in kat["LH"]:
        out = oracle.sketch_lh_ori(t["seq"].encode(), t["w"], t["k"], t["rid"])
        flat = np.stack([out["x"], out["y"]], axis=1).reshape(-1).tolist()
        assert flat == t["out"], (t["w"], t["k"], t["seq"][:30])


def test_radix_sort_128x_reproduces_reference_permutation(kat):
    for t in kat["RS"]:
        a = np.array([tuple(p) for p in t["in"]], dtype=oracle.MM_DTYPE)
        out = oracle.radix_sort_128x(a)
        assert [[int(p["x"]), int(p["y"])] for p in out] == t["out"]


def test_match_pro_matches_reference(kat):
    for t in kat["MP"]:
        assert oracle.match_pro(t["s0"].encode(), t["s1"].encode(), t["i"], t["j"]) == t["d"]


def test_encode_byte_matches_reference(kat):
    seen = set()
    for t in kat["EB"]:
        got = oracle.encode_byte(t["seq"].encode(), t["ref"].encode(), t["pos"], t["dir"], t["L"])
        assert got == t["ok"], t
        seen.add(got)
    assert seen == {0, 1}


NONDEFAULT = dict(e=6, m=4, w=12, cbthr=9, max_rounds=3, step=5, maxthr=30, numdict=4)   # tests/golden/make_golden.py


@pytest.mark.parametrize("tag,params", [("stages_L100", {}), ("stages_L150", {}), ("stages_L100_k24", dict(k=24)), ("stages_L40", {}), ("stages_L75", {}),
                                        ("stages_L100_params", NONDEFAULT)])
def test_all_stages_match_reference_dump(golden_dir, tmp_path, tag, params):
    """Whole hot path (reads -> buckets -> contigs -> merged contigs -> every realign pass): the
    oracle's state after each stage must equal the reference's, byte for byte."""
    with gzip.open(os.path.join(golden_dir, tag + ".reads.gz"), "rb") as f:
        rows = f.read().split(b"\n")[:-1]
    reads = np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), len(rows[0]))
    with gzip.open(os.path.join(golden_dir, tag + ".dump.gz"), "rb") as f:
        want = f.read()
    p = oracle.Pipeline(reads, **params)
    out = str(tmp_path / "dump.txt")
    p.dump_stages(out)
    p.close()
    got = open(out, "rb").read()
    if got != want:
        gl, wl = got.split(b"\n"), want.split(b"\n")
        for i, (a, b) in enumerate(zip(gl, wl)):
            if a != b:
                stage = [x for x in wl[:i] if x.startswith(b"STAGE")][-1:]
                pytest.fail(f"first difference at line {i} after {stage}: got {a[:160]!r} want {b[:160]!r}")
        pytest.fail(f"length differs: {len(gl)} vs {len(wl)} lines")


def test_synth_generator_c_equals_numpy():
    from minicom_amd import synth
    for (seed, n, L) in ((7, 300, 100), (8, 200, 150), (9, 50, 37)):
        a = synth.synth_reads(seed, n, L)
        b = oracle.synth_reads(seed, n, L)
        assert np.array_equal(a, b)
    a = synth.synth_reads(11, 1000, 100, first=400, count=100)
    b = oracle.synth_reads(11, 1000, 100, first=400, count=100)
    assert np.array_equal(a, b) and np.array_equal(a, synth.synth_reads(11, 1000, 100)[400:500])
