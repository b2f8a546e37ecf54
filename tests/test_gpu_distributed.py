"""Multi-GPU path, exactness: the distributed pipeline -- reads sharded, minimizer records exchanged
to bucket owners every bucket round, contig set replicated by all-gather, Stage-2 claim keys MIN-reduced -- must leave on
EVERY rank exactly what the single-GPU pipeline computes over all reads: contig strings, member lists, singleton and
class lists, hence identical stream files (single-end, -p and paired-end).

Two ways to have several ranks with one GPU: real processes on the one card with gloo carrying the library's all-to-all
(callback transport), and a one-rank RCCL communicator (the production transport; ncclSend / ncclRecv to itself)."""
import gzip
import io
import json
import os
import socket
import subprocess
import sys
import tarfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _golden_reads(golden_dir, tag):
    with gzip.open(os.path.join(golden_dir, tag + ".reads.gz"), "rb") as f:
        rows = f.read().split(b"\n")[:-1]
    return np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), len(rows[0])).copy()


def _golden_streams(golden_dir, tag):
    with gzip.open(os.path.join(golden_dir, "streams_" + tag + ".tar.gz"), "rb") as g:
        tf = tarfile.open(fileobj=io.BytesIO(g.read()))
        return {m.name: tf.extractfile(m).read() for m in tf.getmembers()}


def _single(reads, **params):
    from dist_worker import result_arrays
    from minicom_amd.pipeline import Pipeline
    p = Pipeline(reads, host_threads=2, **params)
    p.pre_process()
    res = result_arrays(p)
    stats = {k: p.stat(k) for k in ("rounds", "merge_rounds", "passes", "big_bins")}
    p.close()
    return res, stats


def _run_ranks(tmp_path, reads, world, bounds=None, params=None, dump="", transport="gloo"):
    np.save(tmp_path / "reads.npy", reads)
    port = _free_port()
    procs = []
    for r in range(world):
        cmd = [sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), "--rank", str(r), "--world", str(world), "--port", str(port),
               "--reads", str(tmp_path / "reads.npy"), "--out", str(tmp_path), "--params", json.dumps(params or {}), "--dump", dump, "--transport", transport]
        if bounds:
            cmd += ["--bounds", ",".join(str(b) for b in bounds)]
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for pr in procs:
        try:
            o, _ = pr.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode(errors="replace"))
    for r, pr in enumerate(procs):
        assert pr.returncode == 0, f"rank {r} failed:\n{outs[r][-3000:]}"
    return [dict(np.load(tmp_path / f"rank{r}.npz")) for r in range(world)]


def _assert_same(want, got, who):
    for k, v in want.items():
        assert np.array_equal(v, got[k]), f"{who}: {k} differs ({len(v)} vs {len(got[k])} entries)"


def _synthetic(n, L, seed=2024):
    from minicom_amd import synth
    return synth.synth_reads(seed, n, L, plumbing=True)


@pytest.mark.parametrize("tag,world", [("stages_L100", 2), ("stages_L150", 3), ("stages_L40", 2)])
def test_ranks_hold_the_single_gpu_result_on_reference_fixtures(golden_dir, tmp_path, tag, world):
    reads = _golden_reads(golden_dir, tag)
    want, st = _single(reads)
    ranks = _run_ranks(tmp_path, reads, world)
    for r, got in enumerate(ranks):
        _assert_same(want, got, f"rank {r} of {world}")
        assert got["stats"][:3].tolist() == [st["rounds"], st["merge_rounds"], st["passes"]]
    assert sum(g["stats"][4] for g in ranks) > 0                              # records did change hands


def test_three_ranks_uneven_shards_200k_reads(tmp_path):
    """Shards of very different sizes, one of them empty; 200 k reads: several bucket rounds, merge rounds and Stage-2 passes."""
    reads = _synthetic(200_000, 150)
    n = len(reads)
    want, st = _single(reads)
    assert st["rounds"] >= 3 and st["merge_rounds"] >= 2 and st["passes"] >= 2 and len(want["ref_len"]) > 1000
    for bounds in ([0, 1000, 1000, n], None):
        sub = tmp_path / ("b" + str(bool(bounds))); sub.mkdir()
        for r, got in enumerate(_run_ranks(sub, reads, 3, bounds=bounds)):
            _assert_same(want, got, f"rank {r} bounds {bounds}")


def test_long_bins_replay_is_the_same_on_every_rank(tmp_path):
    """Bins above maxsearch: the tuples of the marked singletons are found against three ranks' index
    shares, gathered, and replayed identically everywhere."""
    from test_gpu_pipeline import _repeat_pileup_reads
    reads = _repeat_pileup_reads(9107, 100, 2400, 2500)
    want, st = _single(reads, maxsearch=3)
    assert st["big_bins"] > 0
    for r, got in enumerate(_run_ranks(tmp_path, reads, 3, params={"maxsearch": 3})):
        _assert_same(want, got, f"rank {r}")
        assert got["stats"][3] > 0


@pytest.mark.parametrize("mode,tag", [("pe", "pe_stages_L100"), ("order", "order_stages_L100"), ("", "stages_L100")])
def test_stream_files_written_by_a_rank_equal_the_reference(golden_dir, tmp_path, mode, tag):
    """Paired end over several GPUs (BASELINE configs[4] in small): mates live on different ranks (rank 0 holds most of the
    first file, rank 1 the rest and the mates); read ids are global, so the pairing streams a rank writes are the reference's."""
    reads = _golden_reads(golden_dir, "stages_L100")
    if mode == "pe":                                                          # the fixture: file 1 = first half of the set, file 2 = second half
        reads = reads[: 2 * (len(reads) // 2)]
    want = _golden_streams(golden_dir, tag)
    _run_ranks(tmp_path, reads, 2, dump=mode or "se")
    d = tmp_path / "streams"
    assert sorted(os.listdir(d)) == sorted(want)
    for name, data in want.items():
        assert (d / name).read_bytes() == data, name


def test_one_rank_rccl_communicator(golden_dir):
    """The production transport: ncclCommInitRank, grouped ncclSend / ncclRecv (to itself, with one rank), on device buffers."""
    import torch
    from dist_worker import result_arrays
    from minicom_amd.distributed import Comm, DistPipeline
    reads = _golden_reads(golden_dir, "stages_L150")
    want, _ = _single(reads)
    comm = Comm.rccl(0, 1, Comm.unique_id(), 0)
    p = DistPipeline(torch.from_numpy(reads).cuda(), 0, len(reads), comm, L=reads.shape[1], device=0, host_threads=2)
    p.pre_process()
    got = result_arrays(p)
    assert p.stat("x_records") > 0
    sent, calls = comm.stats()
    assert calls > 10 and sent == 0                                           # one rank: everything it "sends" stays with it
    p.close(); comm.close()
    _assert_same(want, got, "rccl, one rank")


def test_a_rank_that_fails_takes_the_others_with_it_instead_of_leaving_them_waiting():
    """Round 3's advisor finding: a rank that returned from a rank-local error before a collective left the others waiting in it.
    Now every collective starts with (or carries) a flag exchange and a failing rank announces itself with one when its stage
    function returns (mcom_pipeline.cpp, "Failure protocol").  Three ranks as threads of this process on 60 k reads; rank 1 is made
    to fail right before its k-th flag exchange (include/mcom_test.h) for k spread over every stage of the run: every rank must
    come back with an error -- the failing one with its own, the others naming it -- and none may hang."""
    import ctypes as C
    import threading
    from minicom_amd.distributed import Comm, DistPipeline
    from minicom_amd.hip import McomError
    reads = _synthetic(60000, 100, seed=77)
    n, L = reads.shape
    world = 3

    def run(inject_at):
        comms, hub = Comm.threads(world)
        res, total = [None] * world, [0] * world

        def rank_main(rank):
            lo, hi = n * rank // world, n * (rank + 1) // world
            p = DistPipeline(reads[lo:hi], lo, n, comms[rank], L=L, device=0, host_threads=2)
            p.lib.mcomh_test_inject_failure.argtypes = [C.c_void_p, C.c_long]
            p.lib.mcomh_test_flag_exchanges.restype = C.c_long; p.lib.mcomh_test_flag_exchanges.argtypes = [C.c_void_p]
            made = p.lib.mcomh_test_flag_exchanges(p._h)                   # (creating the pipeline exchanged the shard bounds already)
            if rank == 1 and inject_at > 0:
                p.lib.mcomh_test_inject_failure(p._h, made + inject_at)
            try:
                p.pre_process()
                res[rank] = "ok"
            except McomError as e:
                res[rank] = str(e)
            total[rank] = p.lib.mcomh_test_flag_exchanges(p._h) - made
            p.close()
        th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=180)
        assert not any(t.is_alive() for t in th), f"a rank hangs after an injected failure at exchange {inject_at}: {res}"
        for c in comms:
            c.close()
        return res, total

    res, total = run(0)
    assert res == ["ok"] * world and total[0] == total[1] == total[2] and total[1] > 40, (res, total)
    points = sorted({1, 2, 3, 5, 8, 13, 21, 34, total[1] // 2, total[1] - 8, total[1] - 1, total[1]})
    for k in points:
        res, _ = run(k)
        assert "injected failure" in res[1], (k, res)
        assert all(r != "ok" and "rank 1 failed" in r for i, r in enumerate(res) if i != 1), (k, res)


def test_two_rccl_ranks_between_real_peers(tmp_path):
    """The production transport between two GPUs: grouped ncclSend / ncclRecv over xGMI (minicom_amd/host/mcom_comm.cpp), one process
    per GPU.  No multi-GPU node has been available to this build in any round, so this test has never run: it is skipped on a box with
    one card and is what the first node that appears exercises -- the result of both ranks must be the single-GPU result."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the build's boxes have one): the RCCL transport has run with one rank only")
    reads = _synthetic(200000, 150, seed=99)
    want, _ = _single(reads)
    got = _run_ranks(tmp_path, reads, 2, transport="rccl")
    for r in range(2):
        _assert_same(want, got[r], f"rccl rank {r}")
