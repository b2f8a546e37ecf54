"""Per-rank work of the multi-GPU path at world = 1 / 2 / 4 / 8, measured on ONE card.

No multi-GPU node is available to this build, so the scaling curve itself cannot be measured.  What can be measured is what
bounds it from the compute side: how much device work a rank is left with when the job is cut R ways.  R pipelines run as R
threads of this process (Comm.threads: the all-to-all goes through process memory); a lock lets ONE rank compute at a time
and is given up while a rank waits in an exchange, so a rank's stage times are not disturbed by the others sharing the card.
Per stage and rank:  busy_ms = wall time of the stage - time inside all-to-all calls (staging copies and waiting included),
i.e. kernels + launches + host logic of that rank.  Exchange volume is counted per (sender, receiver) pair.

    python tools/dist_work.py [--reads 32000000] [--read-len 150] [--worlds 1,2,4,8] [--out profiles/r03_dist_work.json]
"""
import argparse
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

STAGES = ("kt_for_reads", "kt_for_bucket", "combine_cluster", "stage2")
PROF = ("classify_pack", "sketch_reads", "radix_pass", "sketch_contigs", "find_next", "consensus", "cindex_build", "realign_reads")


def run_world(reads, n, L, world, single_gpu_path):
    import numpy as np
    import torch
    from minicom_amd.distributed import Comm, DistPipeline
    from minicom_amd.pipeline import Pipeline
    comms, hub = Comm.threads(world, serialize=True)
    res = [None] * world
    errors = []

    def rank_main(rank):
        hub.enter()
        try:
            lo, hi = n * rank // world, n * (rank + 1) // world
            if single_gpu_path:
                p = Pipeline(reads, L=L, host_threads=8)
            else:
                p = DistPipeline(reads[lo:hi], lo, n, comms[rank], L=L, device=0, host_threads=8)
            p.prof_enable(True)
            st = {}
            for name in STAGES:
                t0, c0 = time.perf_counter(), comms[rank].seconds()
                getattr(p, name)()
                torch.cuda.synchronize()                                  # (the stage functions return with kernels still in flight)
                wall, comm = (time.perf_counter() - t0) * 1e3, (comms[rank].seconds() - c0) * 1e3
                st[name] = {"wall_ms": round(wall, 2), "comm_ms": round(comm, 2), "busy_ms": round(wall - comm, 2)}
            digest = p.result_digest()
            prof = {k: round(p.prof_read(k)[0], 2) for k in PROF}
            counters = {k: p.stat(k) for k in ("rounds", "merge_rounds", "passes", "contigs_bucket", "contigs_combine", "cix_entries", "x_records", "x_cindex_entries")}
            laps = {k: round(p.stat(k), 2) for k in ("t_bk_pre", "t_bk_sort", "t_bk_gpu", "t_cb_upload", "t_cb_sketch", "t_cb_pack", "t_cb_idx", "t_cb_findnext", "t_claim", "t_merge_members", "t_merge_cons",
                                                     "t_merge_local", "t_merge_gather", "t_cb_copy", "t_cb_download", "t_cb_join", "t_cb_sg", "t_ra_setup", "t_ra_gpu", "t_ra_update", "t_ra_append", "t_ra_materialize",
                                                     "t_x_reads", "t_x_records", "t_x_contigs", "t_x_sketch", "t_x_index", "t_x_packed", "t_x_pairs", "t_x_merged", "t_x_cindex")}
            laps.update({k: int(p.stat(k)) for k in ("b_x_reads", "b_x_records", "b_x_contigs", "b_x_sketch", "b_x_index", "b_x_packed", "b_x_pairs", "b_x_merged", "b_x_cindex")})
            res[rank] = {"reads": hi - lo, "stages": st, "busy_ms": round(sum(v["busy_ms"] for v in st.values()), 2), "kernel_classes_ms": prof,
                         "digest": digest, "counters": counters, "host_laps_ms": laps}
            p.close()
        except Exception as e:                                                  # noqa: BLE001
            import traceback
            traceback.print_exc()
            errors.append((rank, repr(e))); hub.abort()
        finally:
            hub.leave()
    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    if errors:
        raise SystemExit(f"world {world}: {errors}")
    sent = hub.bytes.copy()
    np.fill_diagonal(sent, 0)
    for c in comms:
        c.close()
    return res, sent


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=32_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--out", default="")
    ap.add_argument("--no-baseline", action="store_true", help="skip the single-GPU code path (for a profiler run of one world)")
    ap.add_argument("--runs", type=int, default=2, help="runs per world; the last one is reported")
    a = ap.parse_args()
    import torch
    import minicom_amd
    n, L = a.reads, a.read_len
    ctx = minicom_amd.Context(0)
    reads = ctx.synth_reads(1004, n, L)                               # configs[3]'s generator (seed 1000 + config number)
    ctx.sync()
    out = {"what": "per-rank busy time (stage wall - time in exchanges) of the distributed pipeline, R ranks as threads on ONE MI355X, one rank computing at a time",
           "workload": f"{n} x {L} bp synthetic reads (configs[3]'s generator), {n // 8} per rank at 8 ranks", "worlds": {}}
    worlds = [int(w) for w in a.worlds.split(",")]
    base = None
    for world in ([] if a.no_baseline else [0]) + worlds:             # 0 = the single-GPU code path (mcomh_create), the baseline
        label = "single_gpu_path" if world == 0 else str(world)
        from minicom_amd.pipeline import pool_trim
        pool_trim()                                                     # (the blocks the world before left in the pools are of other sizes: they only stand in the way)
        for attempt in ["warm-up"] * (a.runs - 1) + ["measured"]:
            t0 = time.time()
            res, sent = run_world(reads, n, L, max(world, 1), world == 0)
            print(f"world {label} ({attempt}): {time.time() - t0:.1f} s", flush=True)
        digests = {tuple(r["digest"]) for r in res}
        if base is None:
            base = res[0]
        entry = {"ranks": res, "all_ranks_same_digest": len(digests) == 1, "digest_equals_single_gpu": digests == {tuple(base["digest"])},
                 "bytes_sent_per_rank_max": int(sent.sum(axis=1).max()), "bytes_per_peer_max": int(sent.max())}
        import statistics
        # median over the ranks beside the maximum: eight pipelines sharing ONE process share its block pools (a rank finds "its" blocks
        # taken by another and pays a hipMalloc of gigabytes: tens of ms, in one or two ranks per run) -- eight processes would not
        for how, f in (("median", statistics.median), ("max", max)):
            entry[how + "_busy_ms_by_stage"] = {s: round(f(r["stages"][s]["busy_ms"] for r in res), 2) for s in STAGES}
            entry[how + "_busy_ms"] = round(f(r["busy_ms"] for r in res), 2)
            entry[how + "_busy_vs_single_gpu_by_stage"] = {s: round(entry[how + "_busy_ms_by_stage"][s] / max(base["stages"][s]["busy_ms"], 1e-9), 3) for s in STAGES}
            entry[how + "_busy_vs_single_gpu"] = round(entry[how + "_busy_ms"] / base["busy_ms"], 3)
        entry["median_kernel_classes_ms"] = {k: round(statistics.median(r["kernel_classes_ms"][k] for r in res), 2) for k in PROF}
        free_b, total_b = torch.cuda.mem_get_info()
        entry["hbm_in_use_GB_after"] = round((total_b - free_b) / 1e9, 1)
        out["worlds"][label] = entry
        print(f"world {label}: median busy {entry['median_busy_ms']:.1f} ms = {entry['median_busy_vs_single_gpu']:.3f} of the single-GPU path (max {entry['max_busy_vs_single_gpu']:.3f}); by stage "
              f"{entry['median_busy_vs_single_gpu_by_stage']}; same digest {entry['digest_equals_single_gpu']}; {entry['bytes_sent_per_rank_max'] / 1e9:.2f} GB sent per rank, "
              f"{entry['hbm_in_use_GB_after']} GB of HBM held", flush=True)
    txt = json.dumps(out, indent=1)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        open(a.out, "w").write(txt + "\n")
    del reads
    torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
