#!/usr/bin/env python3
"""bench.py -- Mreads/s of minicom's sketch + index + overlap hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--genome uniform|repeats]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One step = one full pass of the hot path (the reference's timed region, Stage 1 + Stage 2, preprocess.c:137-234) over one
batch of synthetic reads that are already resident in HBM.  The step ends with the result digest read back
(mcomh_result_digest), which every timed step must share with an untimed run whose result was CHECKED (minicom_amd/check.py:
every read in exactly one place, every member on and like its contig).

Workload.  N = 1: BASELINE.json configs[1], 100 M x 150 bp, k = 31, default parameters.  N > 1: configs[3]'s shape, ONE job of
62.5 M x N reads (500 M at N = 8) sharded over the GPUs -- reads per GPU fixed, "scaling": "weak" -- and, beside it,
`value_strong_100m`: configs[1]'s 100 M-read job sharded over the same N GPUs.  One process per GPU; the distributed pipeline of
libmcom_host.so (mcomh_create_dist): reads sharded, minimizer records exchanged to bucket owners every bucket round over RCCL
(ncclSend / ncclRecv groups), merge-round index built by bucket range, merges by claimed-pair range, the Stage-2 contig index
shared out by key, claims MIN-reduced; every rank ends with the complete result, identical to the single-GPU result over all reads.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 1002            # seed = 1000 + config number (configs[1]); the N > 1 job uses 1004 (configs[3])
CLOCK_HZ = 2.4e9       # MI355X engine clock; 256 CUs x 4 SIMDs; a wave64 VALU instruction issues over 2 cycles (MI355X_MICROARCH.md)
VALU_PEAK = 1024 * CLOCK_HZ / 2
HBM_PEAK = 8000.0      # GB/s (MI355X_MICROARCH.md: 8 TB/s peak, ~6.3 achievable)
READS_PER_GPU_WEAK = 62_500_000


def cpu_baseline(L, sample):
    """The reference itself (oracle/_ref, built from /root/reference in the build container, compiled for the most threads the box
    has cores for: the thread count is a compile-time constant of the reference, minicom:56-91) on the host cores, on a bounded
    sample of the same workload, by its own Stage 1 + Stage 2 timers; falls back to the C restatement (oracle/) when absent."""
    from minicom_amd import synth
    from tools.e2e import host_cores, reference_binaries
    reads = synth.synth_reads(SEED, sample, L)
    tag = f"{sample} reads x {L} bp, same generator (seed {SEED}, 30x coverage, 0.5% substitutions)"
    tried = []
    for exe, variant, threads, march in reference_binaries(L):
        try:
            with tempfile.TemporaryDirectory() as td:
                fq = os.path.join(td, "s.fastq")
                synth.write_fastq_fast(fq, reads, tricky_quality=False)
                out = os.path.join(td, "out"); os.makedirs(out)
                cwd = os.path.join(td, "cwd"); os.makedirs(os.path.join(cwd, "output_ref"))
                t0 = time.perf_counter()
                p = subprocess.run([exe, fq, out], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
                wall = time.perf_counter() - t0
                t = [float(x) for x in re.findall(r"\[Stage \d\] Real time: ([\d.]+)", p.stdout.decode())]
                if p.returncode == 0 and len(t) == 2:
                    return {"value": round(sample / sum(t) / 1e6, 6), "unit": "Mreads/s", "cores": threads, "kind": "reference", "sample_reads": sample,
                            "sample": tag + f"; oracle/_ref/{variant}/minicom_bin (-t {threads}; the box offers {host_cores()} cores; g++ -O3 -march={march} -- the reference's Makefile says "
                                            "-march=native, which cannot travel between boxes), its own Stage 1 + Stage 2 timers, measured in this run",
                            "march": march, "seconds": round(sum(t), 3), "seconds_whole_process": round(wall, 3), "variants_that_failed": tried or None}
                tried.append(f"{variant}: exit {p.returncode}")
        except Exception as e:                                                 # noqa: BLE001
            tried.append(f"{variant}: {type(e).__name__}")
    import oracle
    sample = min(sample, 1_000_000)
    p = oracle.Pipeline(reads[:sample])
    t0 = time.perf_counter(); p.run_all(); dt = time.perf_counter() - t0
    p.close()
    return {"value": round(sample / dt / 1e6, 6), "unit": "Mreads/s", "cores": 1, "kind": "port", "sample_reads": sample,
            "sample": f"{sample} reads x {L} bp, same generator; oracle/mcom_oracle.c, one thread", "seconds": round(dt, 3), "variants_that_failed": tried or None}


CINDEX_NOTE = ("model 'passes': the bytes the radix partitioning moves by design (two streaming passes over the index entries; round 5 on one GPU: no placement, "
               "the singletons' query tuples through the same two passes instead -- cindex_bytes() in this file, DESIGN.md section 3)")
KERNELS = ("classify_pack", "sketch_reads", "radix_pass", "sketch_contigs", "find_next", "dict_build", "realign_windows", "consensus",
           "cindex_build", "realign_reads")


def algorithmic_bytes(name, st, L, nd):
    """Algorithmic bytes of every launch of a timed kernel class over the timed steps."""
    W = (2 * L + 63) // 64
    if name == "realign_windows":      # per (window, dir, dict): 8 key + 8 table word + 8 rank + 8 startpos + 4 id + 8W verify
        return (36 + 8 * W) * (2 * nd - 1) * st["windows"]
    if name == "sketch_reads":         # packed row in, one record out
        return (8 * W + 16) * (st["n"] + st["resketch"])
    if name == "classify_pack":        # ASCII in, packed row + class + N count out
        return (L + 8 * W + 3) * st["n"]
    if name == "realign_reads":
        if st.get("join_passes", 0) and not st.get("join_fallbacks", 0):
            # round 5, the partition-local join (k_rj_queries + k_rj_join): per singleton its row in and its claim out; per query a 12-byte
            # tuple out and, sorted, in again; the index entries streamed once (12 bytes each); per verified candidate the singleton's row, the
            # contig window and the offsets around it
            return (8 * W + 9) * st["ra_singletons"] + 24 * st["ra_lookups"] + 12 * st.get("cix_entries", 0) + (24 + 8 * W + 8 * (W + 1)) * st["ra_verified"]
        # per lookup one 64-B line of keys; per verified window value + offsets + packed window; per singleton row, flag, claim
        return 64 * st["ra_lookups"] + (8 + 24 + 8 * (W + 1)) * st["ra_verified"] + (8 * W + 9) * st["ra_singletons"]
    if name == "cindex_build":
        return cindex_bytes(st)
    if name == "sketch_contigs":       # every contig base (1 byte) in, 16 B per minimizer out, one launch per call
        return (st.get("sketch_bases", 0) + 16 * st.get("sketch_records", 0)) or None
    return None                        # radix_pass / dict_build / find_next: launches of many sizes, no single byte model


def cindex_bytes(st):
    """The contig 17-mer index of Stage 2 (csrc/cindex.hip), model "passes": what THIS build moves by design.  Per entry (12 bytes:
    4 of partition + home bits, 8 of slot): written by pass 1, read twice (histogram: the 4-byte half only) and written by pass 2,
    read by the placement (its second read comes from L2); plus the table, written once."""
    b = (12 + 4 + 12 + 12) * st.get("cix_entries", 0)
    if st.get("cix_slots", 0):
        b += 12 * st.get("cix_entries", 0) + 8 * st.get("cix_slots", 0)      # the placement (the table route: several GPUs, stage2_table, inputs the join does not take)
    if st.get("join_passes", 0) and not st.get("join_fallbacks", 0):
        b += (4 + 12 + 12) * 2 * st.get("ra_lookups", 0)                     # round 5: the singletons' query tuples go through the same two passes (timed in this class)
    return b


# HBM traffic and SQ figures per kernel from the PMC passes committed under profiles/ (separate rocprofv3 runs of this same command,
# tools/profile_round.sh; kernels cannot be counted while bench.py itself is timing them).  A figure is quoted ONLY for the exact
# kernel instantiations this run launched inside the class (the library's own tally, mcom_prof_kernels) and only while the source
# file the kernel was compiled from -- and the shared headers -- are the ones that were profiled (tools/source_sha.py); otherwise
# the figure is null and the line says which kernel was not covered.
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_constants.json")


def pmc_load():
    try:
        with open(PMC_FILE) as f:
            d = json.load(f)
        from tools import source_sha
        d["_by_norm"] = {source_sha.norm(k): v for k, v in d["kernels"].items()}
        d["_shas"], d["_kfiles"] = source_sha.file_shas(), source_sha.kernel_files()
        d.setdefault("_meta", {}).setdefault("commit", "unknown (made before round 4)")
        d["_meta"].setdefault("valu_busy", "")
        return d
    except Exception:                                                          # noqa: BLE001
        return None


def pmc_entry(d, name):
    """(entry, None) of one observed kernel name, or (None, why not)"""
    from tools import source_sha
    e = d["_by_norm"].get(source_sha.norm(name))
    if e is None:
        return None, f"{name}: not in profiles/pmc_constants.json"
    if e.get("source_sha") != source_sha.kernel_sha(name, d["_shas"], d["_kfiles"]):
        return None, f"{name}: {e.get('source_file')} or a shared header changed since commit {d['_meta']['commit'][:8]} was profiled"
    return e, None


def pmc_class(d, observed, field):
    """sum over the kernels observed in a class of field x launches; (value, []) or (None, [reasons])"""
    if not d:
        return None, ["profiles/pmc_constants.json missing"]
    if not observed:
        return None, ["no kernel observed in this class"]
    tot, why = 0.0, []
    for name, cnt in observed.items():
        e, w = pmc_entry(d, name)
        if e is None:
            why.append(w)
        elif field not in e:
            why.append(f"{name}: no {field}")
        else:
            tot += e[field] * cnt
    return (None, brief(why)) if why else (tot, [])


def brief(why, keep=6):
    """a long list of reasons cut to its first few and a count"""
    return why if len(why) <= keep else why[:keep] + [f"... and {len(why) - keep} more"]


def oracle_digest_verdict(n, L, seed, genome, digest):
    """Is the digest the sequential oracle's of this very read set?  tests/golden/scale_digests.json holds the digests oracle/digest_main printed in the
    build container (a fixture: data, hours of one core per line); nothing of oracle/ runs here."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "scale_digests.json")
    try:
        runs = json.load(open(path))["runs"]
    except (OSError, ValueError, KeyError):
        return {"equal": None, "note": "tests/golden/scale_digests.json is missing"}
    for e in runs:
        if genome == "uniform" and (e["n"], e["L"], e["seed"], e.get("coverage", 30)) == (n, L, seed, 30):
            return {"equal": [int(v) for v in e["digest"]] == [int(v) for v in digest],
                    "note": f"tests/golden/scale_digests.json: oracle/digest_main {seed} {n} {L}, {e['oracle_seconds']} s of one core in the build container"}
    return {"equal": None, "note": "the oracle has not been run on this read set (tests/golden/scale_digests.json)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=0, help="reads of the job (sharded over the GPUs); 0 = 100 M on one GPU, 62.5 M per GPU on several")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--genome", choices=("uniform", "repeats"), default="uniform", help="repeats: the repeat-rich device genome (forty-copy segments, tandem repeats, poly-A, (AT)n)")
    ap.add_argument("--host-threads", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=4_000_000)
    ap.add_argument("--e2e-reads", type=int, default=20_000_000, help="reads of the file -> stream files measurement (0 = skip)")
    ap.add_argument("--force-exchange", action="store_true", help="run the distributed pipeline (one-rank RCCL communicator) even with one GPU (testing)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the checked run (the digest comparison between steps stays)")
    ap.add_argument("--no-host-to-host", action="store_true")
    ap.add_argument("--no-event-ab", action="store_true", help="skip the K extra steps without profiler events (the A/B of what the events cost)")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the strong-scaling 100 M-read figure")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    import minicom_amd
    from minicom_amd.hip import McomError
    from minicom_amd.pipeline import Pipeline, pool_trim

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world == 1 and a.gpus > 1:
        raise SystemExit("launch N > 1 with torch.distributed.run (one process per GPU)")
    torch.cuda.set_device(local_rank)
    distributed = world > 1 or a.force_exchange
    comm = None
    if world > 1:
        # torch.distributed (gloo) only bootstraps: it hands the RCCL unique id round and carries the barriers around the timed
        # region; the data path is the library's own communicator (ncclSend / ncclRecv groups over xGMI)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if distributed:
        from minicom_amd.distributed import Comm, DistPipeline
        box = [Comm.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        # RCCL announces itself on STDOUT when its first communicator is made (version, host, library path): this process's stdout is
        # the ONE JSON line, so file descriptor 1 points at stderr while that happens
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            comm = Comm.rccl(rank, world, box[0], local_rank)
        finally:
            sys.stdout.flush()
            os.dup2(keep, 1)
            os.close(keep)
    L = a.read_len
    threads = a.host_threads or max(1, min(64, (os.cpu_count() or 8) // max(1, world)))
    ctx = minicom_amd.Context(local_rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def agree(ok):
        """all ranks must take the same branch"""
        if world == 1:
            return ok
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    STATS = ("windows", "passes", "rounds", "merge_rounds", "claim_rounds", "resketch", "n_sg0", "big_bins", "big_bin_reads", "big_bin_tuples", "dict_builds", "cix_rebuilds",
             "sort_overflow_segments", "sketch_bases", "sort_records", "sketch_strings", "t_reads", "t_bucket", "t_combine", "t_realign", "t_gpu", "ra_lookups", "ra_verified",
             "ra_singletons", "cix_slots", "cix_entries", "sketch_records", "x_records", "x_cindex_entries", "contigs_bucket", "contigs_combine",
             "t_x_reads", "t_x_records", "t_x_contigs", "t_x_sketch", "t_x_index", "t_x_pairs", "t_x_merged", "t_x_cindex", "store_grows", "store_contigs", "store_chars", "join_passes", "join_fallbacks", "join_deferred", "ra_passing")

    def measure(n_total, seed, steps, warmup, check, ab_events=False):
        """K timed steps of one job of n_total reads; returns (seconds, aggregate stats, digests, reads, make)."""
        n_local = n_total // world
        n_total = n_local * world
        # synthetic input, resident in HBM before any timed region: this rank's shard of one n_total-read set.  (The ranks agree that
        # everybody holds its shard before anybody enters the pipeline: from there on a failing rank takes the others with it -- the
        # library's failure protocol -- but an allocation that fails HERE on one rank would leave the others waiting in an exchange.)
        reads, alloc_err = None, None
        try:
            reads = ctx.synth_reads(seed, n_total, L, first=rank * n_local, count=n_local, genome=a.genome)
            ctx.sync()
        except (McomError, RuntimeError) as e:
            alloc_err = e
        if not agree(alloc_err is None):
            raise alloc_err if alloc_err is not None else McomError("another rank could not allocate its shard of the reads")

        def make(**kw):
            if not distributed:
                return Pipeline(reads, L=L, device=local_rank, host_threads=threads, **kw)
            return DistPipeline(reads, rank * n_local, n_total, comm, L=L, device=local_rank, host_threads=threads)
        agg, digests, observed = {}, [], {}

        def step(timed, events=True):
            p = make()
            # HIP events around the hot kernel classes, on the launch stream, inside the timed steps (the contract's live kernel times):
            # about a hundred event pairs per step out of a process-wide pool; what they cost is measured below (`event_overhead`)
            p.prof_enable(events)
            p.pre_process()
            dg = p.result_digest()
            if timed:
                digests.append(dg)
                for k in STATS:
                    agg[k] = agg.get(k, 0.0) + p.stat(k)
                agg["n"] = agg.get("n", 0.0) + (n_local if distributed else p.n)
                for name in KERNELS:
                    ms, calls = p.prof_read(name)
                    agg["ms_" + name] = agg.get("ms_" + name, 0.0) + ms
                    agg["calls_" + name] = agg.get("calls_" + name, 0) + calls
                    o = observed.setdefault(name, {})
                    for kn, c in p.prof_kernels(name).items():                   # the exact kernels the class's time belongs to
                        o[kn] = o.get(kn, 0) + c
                o = observed.setdefault("*", {})
                for kn, c in p.prof_kernels("*").items():
                    o[kn] = o.get(kn, 0) + c
            p.close()
            return dg
        for _ in range(warmup):
            step(False)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(True)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        # A/B of the event profiler: the same K steps again with the events off (untimed for `value`; reported beside it)
        agg["_events_off_s"] = None
        if ab_events:
            barrier()
            t1 = time.perf_counter()
            for _ in range(steps):
                if step(False, events=False) != digests[0]:
                    raise SystemExit("a step without events gave another result")
            barrier()
            agg["_events_off_s"] = time.perf_counter() - t1
        # A/B of Stage 2 as a partition-local join (mcomh_params.stage2_join: no index table): the same K steps, untimed for `value`
        agg["_join_s"] = None
        if ab_events and not distributed:
            try:
                q = make(stage2_join=1); q.pre_process(); q.close()                  # (its buffers are of other sizes than the table route's: one step to have them in the pools)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(steps):
                    q = make(stage2_join=1); q.pre_process()
                    if q.result_digest() != digests[0]:
                        raise SystemExit("a step with the Stage-2 join gave another result")
                    agg["_join_passes"] = q.stat("join_passes"); agg["_join_fallbacks"] = q.stat("join_fallbacks")
                    q.close()
                torch.cuda.synchronize()
                agg["_join_s"] = time.perf_counter() - t1
            except McomError as e:
                agg["_join_err"] = str(e)
        agg["_observed"] = observed
        checked = None
        if check:
            from minicom_amd.check import check_result
            p = make()
            p.pre_process()
            ref_digest = p.result_digest()
            checked = check_result(p, reads, L, rid0=rank * n_local)
            p.close()
            for i, dg in enumerate(digests):
                if dg != ref_digest:
                    raise SystemExit(f"timed step {i} gave another result than the checked run: {dg} vs {ref_digest}")
        else:
            ref_digest = digests[0]
            if any(dg != ref_digest for dg in digests):
                raise SystemExit("the timed steps disagree with each other")
        if world > 1:
            every = [None] * world
            dist.all_gather_object(every, ref_digest)
            if any(dg != every[0] for dg in every):
                raise SystemExit(f"the ranks hold different results: {every}")
        return dt, agg, ref_digest, checked, reads, n_total, n_local

    # ---- the job: configs[1] on one GPU, configs[3]'s shape on several (62.5 M reads per GPU)
    weak = world > 1 and a.reads == 0
    n_job = a.reads or (READS_PER_GPU_WEAK * world if world > 1 else 100_000_000)
    seed = 1004 if weak else SEED
    fell_back = None
    try:
        ok, err = True, None
        try:
            dt, agg, ref_digest, checked, reads, n_total, n_local = measure(n_job, seed, a.steps, a.warmup, not a.no_check, ab_events=not a.no_event_ab and world == 1)
        except (McomError, RuntimeError) as e:                                  # e.g. out of memory at a size no single card could rehearse
            ok, err = False, f"{type(e).__name__}: {e}"
        if not agree(ok):
            if not weak:
                raise SystemExit(f"the job failed: {err}")
            fell_back = f"the {n_job // 1_000_000} M-read job failed on some rank ({err}); this line is the 100 M-read job sharded over the GPUs instead"
            pool_trim(); torch.cuda.empty_cache()
            weak, n_job, seed = False, 100_000_000, SEED
            dt, agg, ref_digest, checked, reads, n_total, n_local = measure(n_job, seed, a.steps, a.warmup, not a.no_check)
    finally:
        pass

    # ---- N > 1: the strong-scaling figure beside the weak one -- configs[1]'s 100 M-read job over the same GPUs
    strong = None
    if world > 1 and weak and not a.no_strong:
        del reads
        pool_trim(); torch.cuda.empty_cache()
        try:
            sdt, sagg, sdig, _, sreads, sn_total, sn_local = measure(100_000_000, SEED, max(1, a.steps // 2), 1, False)
            strong = {"value": round(sn_total * max(1, a.steps // 2) / sdt / 1e6, 3), "unit": "Mreads/s", "ms_per_step": round(sdt / max(1, a.steps // 2) * 1e3, 2),
                      "workload": f"{sn_total // 1_000_000}M x {L}bp (BASELINE configs[1]) as one job sharded over {world} GPUs", "digest": [str(v) for v in sdig]}
            del sreads
        except (McomError, RuntimeError) as e:
            strong = {"value": None, "note": f"not measured: {type(e).__name__}: {e}"}
        reads = None

    # ---- host to host (one GPU): reads in page-locked host memory, uploaded in chunks beside classify / pack / sketch, and the
    # whole result copied back, all inside the timed region -- what a caller pays who hands over host buffers (never `value`)
    h2h = None
    if world == 1 and not distributed and not a.no_host_to_host and rank == 0 and a.genome == "uniform":
        try:
            import psutil
            need = n_local * L
            if psutil.virtual_memory().available < 3 * need + (8 << 30):
                h2h = {"value": None, "note": "not enough host memory for a page-locked copy of the reads"}
            else:
                host = torch.empty((n_local, L), dtype=torch.uint8, pin_memory=True)
                host.copy_(reads[:, :L])
                torch.cuda.synchronize()

                def h_step():
                    p = Pipeline.from_host_streamed(host, device=local_rank, host_threads=threads)
                    p.pre_process()
                    nc = int(p.lib.mcomh_n_contigs(p._h))                        # first accessor: copies strings, members, offsets to the host
                    dg = p.result_digest()
                    p.close()
                    return nc, dg
                h_step()
                hs = max(1, a.steps // 2)
                t1 = time.perf_counter()
                for _ in range(hs):
                    nc, dg = h_step()
                    if dg != ref_digest:
                        raise SystemExit("the host-to-host run gave another result")
                dth = time.perf_counter() - t1
                h2h = {"value": round(n_local * hs / dth / 1e6, 3), "unit": "Mreads/s", "ms_per_step": round(dth / hs * 1e3, 2),
                       "h2d_bytes": need, "d2h": "contig strings + member lists + offsets", "n_contigs": nc,
                       "note": "reads in page-locked host memory, chunked upload overlapped with classify/pack/sketch; results copied to the host inside the timed region"}
                del host
        except SystemExit:
            raise
        except Exception as e:                                                   # noqa: BLE001  never lose the bench line to the extra measurement
            h2h = {"value": None, "note": f"not measured: {type(e).__name__}: {e}"}

    # ---- file -> stream files (one GPU): what the `minicom` command's user waits for, before the external entropy coder
    e2e, e2e_modes = None, {}
    if world == 1 and not distributed and rank == 0 and a.e2e_reads > 0 and a.genome == "uniform":
        reads = None
        pool_trim(); torch.cuda.empty_cache()
        try:
            from tools.e2e import file_to_streams
            e2e = file_to_streams(a.e2e_reads, L, SEED, host_threads=threads, ref_reads=1_000_000)
        except Exception as e:                                                   # noqa: BLE001
            e2e = {"value": None, "note": f"not measured: {type(e).__name__}: {e}"}
        # the same for the two other file sets of the command line: `minicom -p` (order-preserving) and `minicom -1 -2` (paired end)
        for mode in ("order", "paired"):
            try:
                pool_trim(); torch.cuda.empty_cache()
                r = file_to_streams(a.e2e_reads, L, SEED, host_threads=threads, ref_reads=0, mode=mode)
                e2e_modes[mode] = {k: r[k] for k in ("mode", "reads", "value", "unit", "seconds", "stream_bytes")}
            except Exception as e:                                               # noqa: BLE001
                e2e_modes[mode] = {"value": None, "note": f"not measured: {type(e).__name__}: {e}"}
        # ... and from a .fastq.gz of many gzip members (what real inputs are), inflated and parsed by all cores
        try:
            pool_trim(); torch.cuda.empty_cache()
            r = file_to_streams(a.e2e_reads, L, SEED, host_threads=threads, ref_reads=0, gz=True)
            e2e_modes["gz"] = {k: r[k] for k in ("mode", "reads", "value", "unit", "seconds", "stream_bytes", "fastq_bytes", "gzip_members", "fastq_text_GB_per_s", "note")}
        except Exception as e:                                                   # noqa: BLE001
            e2e_modes["gz"] = {"value": None, "note": f"not measured: {type(e).__name__}: {e}"}
        # ... and from a .fastq.gz of ONE member (plain `gzip`): one decoding thread, the others parse (a quarter of the reads: it is the slow road)
        try:
            pool_trim(); torch.cuda.empty_cache()
            r = file_to_streams(max(a.e2e_reads // 4, 1_000_000), L, SEED, host_threads=threads, ref_reads=0, gz=True, gz_one_member=True)
            e2e_modes["gz_one"] = {k: r[k] for k in ("mode", "reads", "value", "unit", "seconds", "stream_bytes", "fastq_bytes", "gzip_members", "fastq_text_GB_per_s", "note")}
        except Exception as e:                                                   # noqa: BLE001
            e2e_modes["gz_one"] = {"value": None, "note": f"not measured: {type(e).__name__}: {e}"}

    if rank == 0:
        nd = len(minicom_amd.hip.dict_layout(L)[0])
        st = dict(agg)
        P = pmc_load()
        default_workload = n_local == 100_000_000 and L == 150 and not distributed and a.genome == "uniform"
        per_step = {q: round(agg.get("ms_" + q, 0.0) / a.steps, 2) for q in KERNELS}

        observed = agg.get("_observed", {})

        def hbm_line(cls, model):
            b = algorithmic_bytes(cls, st, L, nd)
            ms, calls = agg.get("ms_" + cls, 0.0), agg.get("calls_" + cls, 0)
            if not b or ms <= 0 or not calls:
                return None
            ach = (b / calls) / (ms / calls * 1e-3) / 1e9
            line = {"kernel": cls, "bound": "hbm", "model": model, "achieved": round(ach, 2), "peak": HBM_PEAK, "unit": "GB/s", "frac": round(ach / HBM_PEAK, 4),
                    "launches": int(calls), "avg_launch_ms": round(ms / calls, 4), "algorithmic_bytes_per_launch": int(b / calls),
                    "kernels_observed": {k: round(v / a.steps, 1) for k, v in sorted(observed.get(cls, {}).items())}}
            # PMC traffic of exactly these kernels: raw counters, and with the guide's x2 for FETCH_SIZE on the kernels whose loads are wide coalesced streams
            if default_workload:
                raw, why = pmc_class(P, observed.get(cls, {}), "traffic_bytes_per_launch")
                cor, _ = pmc_class(P, observed.get(cls, {}), "traffic_corrected_bytes_per_launch")
            else:
                raw, cor, why = None, None, ["PMC passes were taken at the default workload only"]
            line["traffic"] = int(cor / calls) if cor is not None else None
            line["traffic_raw"] = int(raw / calls) if raw is not None else None
            line["traffic_corrected"] = line["traffic"]
            if why:
                line["traffic_stale"] = why
            return line

        def issue_line(cls):
            """ALU roofline of a kernel class from the SQ pass of its dominant kernel: the share of the SIMDs' cycles in which a VALU instruction was
            executing (SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel time x 2.4 GHz)), with the wave-side split beside it"""
            obs = observed.get(cls, {})
            if not P or not obs:
                return None
            best, why = None, []
            for name in obs:
                e, w = pmc_entry(P, name)
                if e is None:
                    why.append(w)
                elif "valu_busy" in e and (best is None or e.get("sq_pass_ms", 0) > best[1].get("sq_pass_ms", 0)):
                    best = (name, e)
            if why or best is None:
                return {"bound": "valu", "frac": None, "stale": brief(why) or ["no SQ pass for this class"]}
            name, e = best
            return {"bound": "valu", "kernel": name, "frac": e["valu_busy"], "valu_busy": e["valu_busy"], "valu_active_of_wave_cycles": e.get("valu_active_of_wave_cycles"),
                    "wait_any_of_wave_cycles": e.get("wait_any"), "issue_stall_of_wave_cycles": e.get("wait_inst"), "waves_per_simd_avg": e.get("waves_per_simd_avg"),
                    "valu_insts_per_wave": e.get("valu_per_wave"), "source": f"profiles/pmc_constants.json (commit {P['_meta']['commit'][:8]}, SQ pass at {e.get('sq_pass_reads')} reads): " + P["_meta"]["valu_busy"]}

        roof = None
        for cand in sorted(KERNELS, key=lambda q: -agg.get("ms_" + q, 0.0)):      # the dominant class that has a byte model
            roof = hbm_line(cand, "passes" if cand == "cindex_build" else "word")
            if roof:
                break
        if roof:
            roof["traffic_source"] = (f"profiles/pmc_constants.json, commit {P['_meta']['commit'][:8]} (FETCH_SIZE + WRITE_SIZE, separate PMC passes of this command; corrected = FETCH x 2 for the "
                                      "kernels flagged as wide streaming readers, MI355X_MICROARCH.md 'HBM'); quoted for the kernels this run observed, null when one is not covered") if P and default_workload else None
            roof["device_ms_per_step_by_kernel"] = per_step
            if roof["kernel"] == "cindex_build":
                # what the build MUST touch, whatever its passes: the packed contigs in, the table out
                minimal = (2 * st.get("cix_entries", 0) / 8 + 8 * st.get("cix_slots", 0)) / max(1, agg.get("calls_cindex_build", 1))
                roof["algorithmic_minimal"] = {"bytes_per_launch": int(minimal), "achieved": round(minimal / (roof["avg_launch_ms"] * 1e-3) / 1e9, 2), "unit": "GB/s",
                                               "frac": round(minimal / (roof["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK, 4), "what": "packed contigs read once + index table written once"}
                roof["note"] = CINDEX_NOTE
            sk = hbm_line("sketch_reads", "word")
            if sk:
                sk["ms_per_step"] = round(agg["ms_sketch_reads"] / a.steps, 3)
                sk["mreads_per_s"] = round((st["n"] + st["resketch"]) / (agg["ms_sketch_reads"] * 1e-3) / 1e6, 1)
                sk["algorithmic_bytes_per_read"] = 8 * ((2 * L + 63) // 64) + 16
                sk["issue_roofline"] = issue_line("sketch_reads")
                sk["note"] = ("north-star '>= 50 % of HBM bandwidth in the sketch kernel': UNREACHABLE by construction -- the kernel reads 56 bytes per read and executes ~37 "
                              "VALU instructions per BASE (rolling k-mers + hash64 in registers: SURVEY section 7 predicted it); it is VALU-bound at 0.80 of the measured "
                              "issue roofline (issue_roofline below, profiles/pmc_constants.json), which is the figure of merit for it, not the HBM fraction")
                roof["sketch_kernel"] = sk
            sc = hbm_line("sketch_contigs", "word")
            if sc:
                sc["issue_roofline"] = issue_line("sketch_contigs")
                sc["note"] = "one lane per string with an LDS ring per lane: bound by latency and issue at few waves per SIMD, not by bytes"
                if roof["kernel"] != "sketch_contigs":
                    roof["sketch_contigs"] = sc
                else:
                    roof["issue_roofline"] = sc["issue_roofline"]
            if roof["kernel"] != "cindex_build":
                hb = hbm_line("cindex_build", "passes")
                if hb:
                    hb["note"] = CINDEX_NOTE
                    roof["hbm_bound_kernel"] = hb
        whole = None
        launches_live = sum(observed.get("*", {}).values()) / a.steps if observed.get("*") else None
        if P and default_workload and "_whole_step" in P:
            w = P["_whole_step"]
            ms = dt / a.steps * 1e3
            # the whole-step counter total stands only while every kernel this run launched is one that was counted, from unchanged source
            why = [x for x in (pmc_entry(P, k)[1] for k in observed.get("*", {})) if x]
            whole = {"traffic_GB": round(w["traffic_corrected_bytes"] / 1e9, 1) if not why else None, "traffic_raw_GB": round(w["traffic_raw_bytes"] / 1e9, 1) if not why else None,
                     "hbm_frac": round(w["traffic_corrected_bytes"] / 1e9 / (ms * 1e-3) / HBM_PEAK, 4) if not why else None, "launches_in_pmc_pass": w.get("launches"),
                     "kernel_launches_per_step": launches_live, "stale": brief(why) or None,
                     "source": f"profiles/pmc_constants.json (commit {P['_meta']['commit'][:8]}): FETCH_SIZE + WRITE_SIZE summed over every kernel of one step (PMC passes of this command), over this run's "
                               "ms_per_step; kernel_launches_per_step = the library's own tally in the timed steps (memsets and copies not included)"}
        elif launches_live is not None:
            whole = {"kernel_launches_per_step": launches_live}
        ev_ab = None
        if agg.get("_events_off_s"):
            ev_ab = {"ms_per_step_with_events": round(dt / a.steps * 1e3, 2), "ms_per_step_without_events": round(agg["_events_off_s"] / a.steps * 1e3, 2), "steps_each": a.steps,
                     "note": "the timed steps carry the HIP events of the kernel timing (the contract's live measurement); the same K steps repeated without them, same process, same inputs"}
        join_ab = None
        if agg.get("_join_s"):
            join_ab = {"ms_per_step_table": round((agg["_events_off_s"] or dt) / a.steps * 1e3, 2), "ms_per_step_join": round(agg["_join_s"] / a.steps * 1e3, 2), "steps_each": a.steps,
                       "join_passes": agg.get("_join_passes"), "join_fallbacks": agg.get("_join_fallbacks"),
                       "note": "Stage 2 on one GPU as a partition-local join (mcomh_params.stage2_join = 1: index entries sorted by partition joined in LDS with the singletons' keys, no table, later "
                               "passes from deferred candidates; csrc/realign.hip) against the default table route: the same K steps without events after one warm-up step, same digest.  Built and exact; "
                               "kernel for kernel it takes what the table route takes (DESIGN.md section 3.4), so the table stays the default"}
        ppp = a.steps
        res = {
            "metric": "Mreads/s (sketch+index+overlap) on 150bp reads, 1/2/4/8 GPU; bit-exact decompress",
            "value": round(n_total * a.steps / dt / 1e6, 4), "unit": "Mreads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak" if (weak or world == 1) else "strong", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": (f"{n_total // 1_000_000}M x {L}bp synthetic reads, k=31 default params, "
                                    + ("BASELINE configs[1], one GPU" if world == 1 else
                                       (f"BASELINE configs[3]'s shape: one job of {READS_PER_GPU_WEAK / 1e6:g} M reads per GPU" if weak else "BASELINE configs[1] as one job") + f" sharded over {world} GPUs")
                                    + ("" if a.genome == "uniform" else "; REPEAT-RICH genome (forty-copy 2 kb segments, tandem repeats, poly-A, (AT)n)") + "; full Stage 1 + Stage 2 per step"),
                       "genome": a.genome, "reads_total": n_total, "reads_per_gpu": n_local, "read_len": L, "k": 31,
                       "parallelism": "1 GPU" if not distributed else f"{world} GPU(s): reads sharded; per bucket round minimizer records to bucket owners; merge-round index by bucket range, merges by "
                                                                     "claimed-pair range, Stage-2 index shared out by key -- all over RCCL send/recv groups; result replicated, identical to the "
                                                                     "single-GPU result",
                       "scaling_note": ("reads per GPU: 100 M at N = 1 (configs[1]), 62.5 M at N > 1 (configs[3] = 500 M at N = 8); value_strong_100m = configs[1]'s job over the same GPUs. "
                                        "No multi-GPU node was available to this build: the N > 1 path is exact (tests), its scaling unmeasured (profiles/r05_dist_work.json: per-rank work on one card)"),
                       "fell_back": fell_back, "host_threads": threads,
                       "per_step": {q: round(agg.get(q, 0.0) / ppp, 1) for q in ("rounds", "merge_rounds", "claim_rounds", "passes", "windows", "resketch", "n_sg0", "contigs_bucket",
                                                                                  "contigs_combine", "big_bins", "big_bin_reads", "big_bin_tuples", "dict_builds", "cix_rebuilds",
                                                                                  "sort_overflow_segments", "x_records", "x_cindex_entries", "store_grows", "store_contigs", "store_chars", "join_passes", "join_fallbacks", "join_deferred")},
                       "stage_ms_rank0": {q: round(agg.get(q, 0.0) / ppp, 1) for q in ("t_reads", "t_bucket", "t_combine", "t_realign", "t_gpu", "t_x_reads", "t_x_records",
                                                                                        "t_x_contigs", "t_x_sketch", "t_x_index", "t_x_pairs", "t_x_merged", "t_x_cindex")},
                       "kernel_timing": "HIP events around the hot kernel classes on the launch stream, inside the timed steps (pooled events, ~100 pairs per step; cost: event_overhead)",
                       "results": "contig set (strings + member lists) complete in HBM at the end of a step, its digest read back inside the step; host copy on demand"},
            "result": {"digest": [str(v) for v in ref_digest], "digest_fields": "contigs, chars, members, unclustered, strings, member words, offsets, lists",
                       "every_timed_step_equal": True, "checked_run": checked, "sequential_oracle": oracle_digest_verdict(n_total, L, seed, a.genome, ref_digest)},
            "value_host_to_host": h2h,
            "value_file_to_streams": e2e,
            "value_file_to_streams_order_preserving": e2e_modes.get("order"),
            "value_file_to_streams_paired_end": e2e_modes.get("paired"),
            "value_file_to_streams_gz": e2e_modes.get("gz"),
            "value_file_to_streams_gz_one_member": e2e_modes.get("gz_one"),
            "stage2_join_ab": join_ab,
            "value_strong_100m": strong,
            "whole_step": whole,
            "event_overhead": ev_ab,
            "roofline": roof,
        }
        if comm is not None:
            sent, calls = comm.stats()
            res["config"]["rccl_bytes_sent_rank0_total"] = int(sent)
        if not a.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(L, a.cpu_sample)
        elif not a.no_cpu_baseline:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
