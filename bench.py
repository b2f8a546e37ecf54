#!/usr/bin/env python3
"""bench.py -- Mreads/s of minicom's sketch + index + overlap hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One step = one full pass of the hot path (the reference's timed region, Stage 1 + Stage 2, preprocess.c:137-234) over one
batch of synthetic reads that are already resident in HBM: BASELINE.json configs[1], 100 M x 150 bp, k = 31, default
parameters, per GPU.  The step ends with the result digest read back (mcomh_result_digest), which every timed step must
share with an untimed run whose result was CHECKED (minicom_amd/check.py: every read in exactly one place, every member on
and like its contig).
N > 1: one process per GPU; the distributed pipeline of libmcom_host.so (mcomh_create_dist): reads sharded, minimizer
records exchanged to bucket owners every bucket round over RCCL (ncclSend/ncclRecv groups), contig set replicated by
all-gather, Stage-2 claims MIN-reduced; every rank ends with the complete result, identical to the single-GPU result over all
N x reads ("weak" scaling: reads per GPU fixed).  Prints ONE JSON line on rank 0.
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

SEED = 1002            # seed = 1000 + config number
CLOCK_HZ = 2.4e9       # MI355X engine clock; 256 CUs x 4 SIMDs; a wave64 VALU instruction issues over 2 cycles (MI355X_MICROARCH.md)
VALU_PEAK = 1024 * CLOCK_HZ / 2


def cpu_baseline(L, sample):
    """The reference itself (oracle/_ref, built from /root/reference in the build container) on the host cores,
    on a bounded sample of the same workload; falls back to the C restatement (oracle/) when the binary is absent."""
    from minicom_amd import synth
    reads = synth.synth_reads(SEED, sample, L)
    tag = f"{sample} reads x {L} bp, same generator (seed {SEED}, 30x coverage, 0.5% substitutions)"
    for variant, cores in (("L150_t16", 16), ("L150", 1)):
        exe = os.path.join(ROOT, "oracle", "_ref", variant, "minicom_bin")
        if L != 150 or not os.path.exists(exe):
            continue
        try:
            with tempfile.TemporaryDirectory() as td:
                fq = os.path.join(td, "s.fastq")
                synth.write_fastq(fq, reads)
                out = os.path.join(td, "out"); os.makedirs(out)
                cwd = os.path.join(td, "cwd"); os.makedirs(os.path.join(cwd, "output_ref"))
                p = subprocess.run([exe, fq, out], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
                t = [float(x) for x in re.findall(r"\[Stage \d\] Real time: ([\d.]+)", p.stdout.decode())]
                if p.returncode == 0 and len(t) == 2:
                    return {"value": round(sample / sum(t) / 1e6, 6), "unit": "Mreads/s", "cores": cores, "kind": "reference", "sample_reads": sample,
                            "sample": tag + f"; reference's own Stage 1 + Stage 2 timers, -t {cores} (the reference slows down with size: "
                                            "0.056 Mreads/s on 8 M reads, DESIGN.md section 6)", "seconds": round(sum(t), 3)}
        except Exception:
            pass
    import oracle
    p = oracle.Pipeline(reads)
    t0 = time.perf_counter(); p.run_all(); dt = time.perf_counter() - t0
    p.close()
    return {"value": round(sample / dt / 1e6, 6), "unit": "Mreads/s", "cores": 1, "kind": "port", "sample_reads": sample,
            "sample": tag + "; oracle/mcom_oracle.c, one thread", "seconds": round(dt, 3)}


KERNELS = ("classify_pack", "sketch_reads", "radix_pass", "sketch_contigs", "find_next", "dict_build", "realign_windows", "consensus",
           "cindex_build", "realign_reads")


def algorithmic_bytes(name, st, L, nd, model="word"):
    """Algorithmic bytes of every launch of a timed kernel class over the timed steps.
    model "word": the words an entry touches, as section 8d counts them; "sector": the 64-byte lines a random access moves."""
    W = (2 * L + 63) // 64
    if name == "realign_windows":      # per (window, dir, dict): 8 key + 8 table word + 8 rank + 8 startpos + 4 id + 8W verify
        return (36 + 8 * W) * (2 * nd - 1) * st["windows"]
    if name == "sketch_reads":         # packed row in, one record out
        return (8 * W + 16) * (st["n"] + st["resketch"])
    if name == "classify_pack":        # ASCII in, packed row + class + N count out
        return (L + 8 * W + 3) * st["n"]
    if name == "realign_reads":        # per lookup one 64-B line of keys; per verified window value + offsets + packed window; per singleton row, flag, claim
        return 64 * st["ra_lookups"] + (8 + 24 + 8 * (W + 1)) * st["ra_verified"] + (8 * W + 9) * st["ra_singletons"]
    if name == "cindex_build":
        return cindex_bytes(st, model)
    if name == "sketch_contigs":       # every contig base (1 byte) in, 16 B per minimizer out, one launch per call
        return (st.get("sketch_bases", 0) + 16 * st.get("sketch_records", 0)) or None
    return None                        # radix_pass / dict_build / find_next: launches of many sizes, no single byte model


def cindex_bytes(st, model):
    """The contig 17-mer index of Stage 2 (csrc/cindex.hip), per build: radix-partitioned, everything streams, so words and
    sectors coincide.  Per entry (12 bytes: 4 of partition + home bits, 8 of slot): written by pass 1, read twice (histogram: the
    4-byte half only) and written by pass 2, read by the placement (its second read comes from L2); plus the table, written once."""
    return (12 + 4 + 12 + 12 + 12) * st.get("cix_entries", 0) + 8 * st.get("cix_slots", 0)


# HBM traffic and SQ instruction counts per kernel from the PMC passes committed under profiles/ (separate rocprofv3 runs of
# this same command; kernels cannot be counted while bench.py itself is timing them)
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_constants.json")
PMC_KERNELS = {"sketch_contigs": ["k_sketch_scan<true>", "k_sketch_scan<false>", "k_sketch_contigs"], "realign_reads": ["k_realign_reads<5, 16, false>"], "classify_pack": ["k_classify_pack16"],
               "sketch_reads": ["k_sketch_reads<5, true, true>", "k_sketch_reads<5, true>"],
               "cindex_build": ["k_cindex_blocks", "k_cx_hist1", "k_cx_scatter1", "k_cx_hist2", "k_cx_scatter2", "k_cx_bounds", "k_cx_assemble_sorted", "k_cx_assemble"]}


def pmc(cls, field):
    """A PMC figure of a kernel class (summed over the kernels of the class that were counted), or None."""
    try:
        with open(PMC_FILE) as f:
            d = json.load(f)
        vals = [d["kernels"][k][field] for k in PMC_KERNELS[cls] if k in d["kernels"] and field in d["kernels"][k]]
        return sum(vals) if vals else None
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=100_000_000, help="reads of the job (sharded over the GPUs)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--host-threads", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=1_000_000)
    ap.add_argument("--force-exchange", action="store_true", help="run the distributed pipeline (one-rank RCCL communicator) even with one GPU (testing)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the checked run (the digest comparison between steps stays)")
    ap.add_argument("--no-host-to-host", action="store_true")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    import minicom_amd
    from minicom_amd.pipeline import Pipeline

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
        comm = Comm.rccl(rank, world, box[0], local_rank)
    dev = torch.device("cuda", local_rank)
    # One JOB of a.reads reads, sharded: the scaling series is STRONG (total work fixed).  A job is bounded by the format the
    # reference and this implementation share -- contig ids are index << 8 in 32 bits (kthread_bucket.c:458): 2^24 contigs, about
    # 200 M reads of this generator -- so "100 M reads per GPU" cannot exist as one job on 4 or 8 GPUs, for the reference either.
    L = a.read_len
    n_local = a.reads // world
    n_total = n_local * world
    threads = a.host_threads or max(1, min(64, (os.cpu_count() or 8) // max(1, world)))
    ctx = minicom_amd.Context(local_rank)

    # synthetic input, resident in HBM before any timed region: this rank's shard of one n_total-read set
    reads = ctx.synth_reads(SEED, n_total, L, first=rank * n_local, count=n_local)
    ctx.sync()

    def make():
        if not distributed:
            return Pipeline(reads, L=L, device=local_rank, host_threads=threads)
        return DistPipeline(reads, rank * n_local, n_total, comm, L=L, device=local_rank, host_threads=threads)

    agg, digests = {}, []

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    STATS = ("windows", "passes", "rounds", "merge_rounds", "resketch", "n_sg0", "big_bins", "sketch_bases", "sort_records", "sketch_strings", "t_reads", "t_bucket", "t_combine",
             "t_realign", "t_gpu", "ra_lookups", "ra_verified", "ra_singletons", "cix_slots", "cix_entries", "sketch_records", "x_records", "t_x_reads",
             "t_x_records", "t_x_contigs", "t_x_sketch", "t_x_pairs")

    def step(timed):
        p = make()
        p.prof_enable(True)
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
        p.close()

    for _ in range(a.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- the checked run (untimed): the result the timed steps must reproduce
    checked = None
    if not a.no_check:
        from minicom_amd.check import check_result
        p = make()
        p.pre_process()
        ref_digest = p.result_digest()
        checked = check_result(p, reads, L, rid0=rank * n_local)
        p.close()
        for i, dg in enumerate(digests):
            assert dg == ref_digest, f"timed step {i} gave another result than the checked run: {dg} vs {ref_digest}"
    else:
        ref_digest = digests[0]
        assert all(dg == ref_digest for dg in digests), "the timed steps disagree with each other"
    if world > 1:
        every = [None] * world
        dist.all_gather_object(every, ref_digest)
        assert all(dg == every[0] for dg in every), f"the ranks hold different results: {every}"

    # ---- host to host (one GPU): reads in page-locked host memory, uploaded in chunks beside classify / pack / sketch, and the
    # whole result copied back, all inside the timed region -- what a caller pays who hands over host buffers (never `value`)
    h2h = None
    if world == 1 and not distributed and not a.no_host_to_host and rank == 0:
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
                t1 = time.perf_counter()
                for _ in range(a.steps):
                    nc, dg = h_step()
                    assert dg == ref_digest, "the host-to-host run gave another result"
                dth = time.perf_counter() - t1
                h2h = {"value": round(n_local * a.steps / dth / 1e6, 3), "unit": "Mreads/s", "ms_per_step": round(dth / a.steps * 1e3, 2),
                       "h2d_bytes": need, "d2h": "contig strings + member lists + offsets", "n_contigs": nc,
                       "note": "reads in page-locked host memory, chunked upload overlapped with classify/pack/sketch; results copied to the host inside the timed region"}
                del host
        except Exception as e:                                                   # never lose the bench line to the extra measurement
            h2h = {"value": None, "note": f"not measured: {type(e).__name__}: {e}"}

    if rank == 0:
        nd = len(minicom_amd.hip.dict_layout(L)[0])
        st = dict(agg)
        default_workload = n_local == 100_000_000 and L == 150 and not distributed
        per_step = {q: round(agg.get("ms_" + q, 0.0) / a.steps, 2) for q in KERNELS}

        def hbm_line(cls, model="word"):
            b = algorithmic_bytes(cls, st, L, nd, model)
            ms, calls = agg.get("ms_" + cls, 0.0), agg.get("calls_" + cls, 0)
            if not b or ms <= 0 or not calls:
                return None
            ach = (b / calls) / (ms / calls * 1e-3) / 1e9
            return {"kernel": cls, "bound": "hbm", "model": model, "achieved": round(ach, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4),
                    "launches": int(calls), "avg_launch_ms": round(ms / calls, 4), "algorithmic_bytes_per_launch": int(b / calls),
                    "traffic": pmc(cls, "traffic_bytes_per_launch") if default_workload else None}

        def issue_line(cls, waves):
            """Integer-issue roofline of an ALU-bound kernel: VALU wave-instructions per second against 1024 SIMDs x clock / 2."""
            v = pmc(cls, "valu_per_wave")
            ms = agg.get("ms_" + cls, 0.0)
            if not v or ms <= 0:
                return None
            rate = v * waves / (ms * 1e-3)
            return {"bound": "valu_issue", "valu_insts_per_wave": v, "waves": int(waves), "achieved": round(rate / 1e9, 1), "peak": round(VALU_PEAK / 1e9, 1),
                    "unit": "G wave-instructions/s", "frac": round(rate / VALU_PEAK, 4), "source": "profiles/pmc_constants.json (SQ_INSTS_VALU / SQ_WAVES, PMC pass)"}

        roof = None
        for cand in sorted(KERNELS, key=lambda q: -agg.get("ms_" + q, 0.0)):      # the dominant class that has a byte model
            roof = hbm_line(cand)
            if roof:
                break
        if roof:
            roof["traffic_source"] = "profiles/pmc_constants.json (FETCH_SIZE + WRITE_SIZE, separate PMC passes of this command)" if default_workload else None
            roof["device_ms_per_step_by_kernel"] = per_step
            if roof["kernel"] == "sketch_contigs":
                roof["note"] = "integer-issue bound (see issue_roofline): 1 byte in and 0.07 records out per position; the HBM fraction is small by construction"
            # the per-read sketch kernel alone (SURVEY section 8d asks for it): ALU bound, so both rooflines
            sk = hbm_line("sketch_reads")
            if sk:
                sk["ms_per_step"] = round(agg["ms_sketch_reads"] / a.steps, 3)
                sk["mreads_per_s"] = round((st["n"] + st["resketch"]) / (agg["ms_sketch_reads"] * 1e-3) / 1e6, 1)
                sk["algorithmic_bytes_per_read"] = 8 * ((2 * L + 63) // 64) + 16
                sk["issue_roofline"] = issue_line("sketch_reads", (st["n"] + st["resketch"]) / 64)
                sk["note"] = "ALU bound: one thread per read, rolling k-mers + hash64 in registers (48 VALU instructions per base, PMC); the HBM fraction cannot be high"
                roof["sketch_kernel"] = sk
            sc = hbm_line("sketch_contigs")
            if sc:
                # one lane per string: 64 strings per wave; the kernel is neither byte- nor issue-bound (5-6 waves per CU: latency),
                # both fractions are given so that this can be seen
                sc["issue_roofline"] = issue_line("sketch_contigs", st.get("sketch_strings", 0.0) / 64)
                sc["note"] = "one lane per string, LDS rings allow 5-6 waves per CU: bound by latency (PMC: SQ_WAIT_ANY 55 % of wave cycles), neither by bytes nor by issue"
                if roof["kernel"] != "sketch_contigs":
                    roof["sketch_contigs"] = sc
                else:
                    roof["issue_roofline"] = sc["issue_roofline"]
            # the heaviest kernel that IS bound by HBM, both byte models
            if roof["kernel"] != "cindex_build":
                hb = hbm_line("cindex_build")
                if hb:
                    hb["note"] = "radix-partitioned build (two streaming passes + one placement pass per partition): word and sector models coincide"
                    roof["hbm_bound_kernel"] = hb
        res = {
            "metric": "Mreads/s (sketch+index+overlap) on 150bp reads, 1/2/4/8 GPU; bit-exact decompress",
            "value": round(n_total * a.steps / dt / 1e6, 4), "unit": "Mreads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{n_total // 1_000_000}M x {L}bp synthetic reads, k=31 default params (BASELINE configs[1]), one job"
                                   + (f" sharded over {world} GPUs ({n_local // 1_000_000}M reads each)" if world > 1 else "") + "; full Stage 1 + Stage 2 per step",
                       "reads_total": n_total, "reads_per_gpu": n_local, "read_len": L, "k": 31,
                       "parallelism": "1 GPU" if not distributed else f"{world} GPU(s): reads sharded, per-round minimizer-record exchange to bucket owners + all-gathers over RCCL "
                                                                     "send/recv groups, result replicated and identical to the single-GPU result",
                       "host_threads": threads,
                       "per_step": {q: round(agg.get(q, 0.0) / a.steps, 1) for q in ("rounds", "merge_rounds", "passes", "windows", "resketch", "n_sg0", "big_bins", "x_records")},
                       "stage_ms_rank0": {q: round(agg.get(q, 0.0) / a.steps, 1) for q in ("t_reads", "t_bucket", "t_combine", "t_realign", "t_gpu", "t_x_reads", "t_x_records",
                                                                                           "t_x_contigs", "t_x_sketch", "t_x_pairs")},
                       "results": "contig set (strings + member lists) complete in HBM at the end of a step, its digest read back inside the step; host copy on demand"},
            "result": {"digest": [str(v) for v in ref_digest], "digest_fields": "contigs, chars, members, unclustered, strings, member words, offsets, lists",
                       "every_timed_step_equal": True, "checked_run": checked},
            "value_host_to_host": h2h,
            "roofline": roof,
        }
        if comm is not None:
            sent, calls = comm.stats()
            res["config"]["rccl_bytes_sent_rank0_per_step"] = int(sent / (a.steps + a.warmup + (0 if a.no_check else 1)))
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
