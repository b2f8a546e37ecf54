"""Turns the raw output of tools/profile_round.sh (gpurun_out/prof_TAG/) into the tracked evidence of one state of the code:

    python tools/pmc_constants.py gpurun_out/prof_TAG TAG

    profiles/TAG_bench100m_kernel_stats.csv   rocprofv3 --kernel-trace --stats: top_kernels view (durations in us)
    profiles/TAG_bench100m_timeline.txt       the last timed step: launches, span, kernel time, idle, per kernel
    profiles/TAG_bench100m_trace_step.csv.gz  every launch of that step (name, start, duration)
    profiles/TAG_bench100m_line.json          the bench line of the profiled run
    profiles/TAG_bench100m_default_line.json  `python bench.py` unprofiled, same box, same code
    profiles/TAG_pmc_traffic_100m.json        FETCH_SIZE / WRITE_SIZE per kernel (two passes)
    profiles/TAG_pmc_sq.json                  SQ counters per kernel (two passes, at the smaller size named inside)
    profiles/pmc_constants.json               what bench.py reads: per EXACT kernel name its traffic and SQ figures, the sha of the source
                                              file it was compiled from (tools/source_sha.py), and the commit; bench.py drops a kernel's
                                              figures when the name it observed is not here or its source has changed since
"""
import collections
import csv
import datetime
import gzip
import io
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import source_sha                                                                   # noqa: E402

# Kernels whose loads are wide coalesced streams (every lane of a wave reads consecutive 8- or 16-byte words of arrays walked front
# to back): on gfx950 FETCH_SIZE tallies such 128-byte requests at 64 bytes (MI355X_MICROARCH.md, "HBM"), so their FETCH is doubled
# for `traffic_corrected`.  Calibration in this code base: k_classify_pack16 reads 15.0 GB of ASCII and FETCH_SIZE says 8.1; k_cx_scatter2
# reads 12 bytes x 0.88 G entries = 10.6 GB and FETCH_SIZE says 5.4.  Kernels that gather (table lookups, row gathers, per-object
# kernels) keep their raw count.
STREAMING = ("k_classify_flat", "k_classify_pack16", "k_classify_pack", "k_cx_hist2", "k_cx_scatter2", "k_cx_bounds", "k_cx_assemble_sorted", "k_cx_place", "k_cx_split", "k_radix_hist", "k_radix_scatter",
             "k_scan_tile", "k_scan_add", "k_scan_one", "k_scan64_tile", "k_scan64_add", "k_digest", "k_mask_records", "k_st_refbin", "k_table_heads", "k_bucket_starts", "k_live_flags", "k_prefix_copy",
             "k_min_fold", "k_clear_list")
CLOCK_HZ = 2.4e9
N_SIMD = 1024


def read_counters(path):
    """{kernel: {counter: sum, '_launches': n, '_ns': summed duration}} from a rocprofv3 counter_collection.csv(.gz)"""
    op = gzip.open if path.endswith(".gz") else open
    res = collections.defaultdict(lambda: collections.Counter())
    seen = set()
    with op(path, "rt", newline="") as f:
        for r in csv.DictReader(f):
            k = source_sha.norm(r["Kernel_Name"])
            res[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                res[k]["_launches"] += 1
                res[k]["_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return res


def find(d, suffix):
    for base, _, files in os.walk(d):
        for f in files:
            if f.endswith(suffix) or f.endswith(suffix + ".gz"):
                return os.path.join(base, f)
    raise SystemExit(f"no {suffix} under {d}")


def is_step_kernel(k):
    return not (k.startswith("k_synth") or k.startswith("at::") or "at::native" in k or k.startswith("hip") or k.startswith("__amd"))


def main():
    src, tag = sys.argv[1], sys.argv[2]
    prof = os.path.join(ROOT, "profiles")
    sha_then = json.load(open(os.path.join(src, "source_sha.json")))
    sha_now = {"files": source_sha.file_shas()}
    if sha_then["files"] != sha_now["files"]:
        changed = sorted(f for f in set(sha_then["files"]) | set(sha_now["files"]) if sha_then["files"].get(f) != sha_now["files"].get(f))
        print("NOTE: the tree has changed since the profile was taken:", changed)
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], stdout=subprocess.PIPE).stdout.decode().strip()
    dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "minicom_amd/csrc"], stdout=subprocess.PIPE).stdout.strip())

    shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(prof, f"{tag}_bench100m_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "timeline.txt"), os.path.join(prof, f"{tag}_bench100m_timeline.txt"))
    with open(os.path.join(src, "kernel_trace_step2.csv"), "rb") as f, gzip.open(os.path.join(prof, f"{tag}_bench100m_trace_step.csv.gz"), "wb", 9) as g:
        g.write(f.read())
    for a, b in (("line_profiled.json", "line"), ("line_default.json", "default_line")):
        txt = [ln for ln in open(os.path.join(src, a)).read().splitlines() if ln.startswith("{")]
        if txt:
            with open(os.path.join(prof, f"{tag}_bench100m_{b}.json"), "w") as f:
                f.write(txt[-1] + "\n")

    fetch = read_counters(find(os.path.join(src, "fetch"), "counter_collection.csv"))
    write = read_counters(find(os.path.join(src, "write"), "counter_collection.csv"))
    traffic = {}
    for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch[k]["FETCH_SIZE"] + write[k]["WRITE_SIZE"])):
        traffic[k] = {"launches": int(fetch[k]["_launches"] or write[k]["_launches"]), "fetch_bytes": fetch[k]["FETCH_SIZE"] * 1024.0, "write_bytes": write[k]["WRITE_SIZE"] * 1024.0,
                      "ms_in_fetch_pass": round(fetch[k]["_ns"] / 1e6, 3)}
    json.dump(traffic, open(os.path.join(prof, f"{tag}_pmc_traffic_100m.json"), "w"), indent=1)

    sqa = read_counters(find(os.path.join(src, "sqa"), "counter_collection.csv"))
    sqb = read_counters(find(os.path.join(src, "sqb"), "counter_collection.csv"))
    sq = {}
    for k in sorted(sqa, key=lambda k: -sqa[k]["SQ_WAVE_CYCLES"]):
        d = {c: v for c, v in sqa[k].items() if not c.startswith("_")}
        d.update({c: v for c, v in sqb.get(k, {}).items() if not c.startswith("_") and c != "SQ_WAVES"})
        d["launches"] = int(sqa[k]["_launches"]); d["ms_in_pass"] = round(sqa[k]["_ns"] / 1e6, 3)
        sq[k] = d
    sq_reads = None
    try:
        sq_reads = json.loads([ln for ln in open(os.path.join(src, "sqa.out")).read().splitlines() if ln.startswith("{")][-1])["config"]["reads_total"]
    except Exception:                                                               # noqa: BLE001
        pass
    json.dump({"_reads": sq_reads, "_note": "sums over the launches of ONE step of bench.py at _reads reads; SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)",
               "kernels": sq}, open(os.path.join(prof, f"{tag}_pmc_sq.json"), "w"), indent=1)

    shas, kfiles = source_sha.file_shas(), source_sha.kernel_files()
    res = {"_meta": {"tag": tag, "commit": commit, "commit_dirty_csrc": dirty, "made": datetime.date.today().isoformat(),
                     "source_files_at_profile_time": sha_then["files"],
                     "traffic": f"profiles/{tag}_pmc_traffic_100m.json: FETCH_SIZE and WRITE_SIZE, separate rocprofv3 passes (--kernel-trace --pmc) of `python3 bench.py --steps 1 --warmup 0 "
                                "--no-cpu-baseline --no-host-to-host --no-check --e2e-reads 0` (100 M x 150 bp); raw = counters x 1024; corrected = FETCH x 2 for the wide streaming readers "
                                "listed in tools/pmc_constants.py (MI355X_MICROARCH.md, HBM)",
                     "sq": f"profiles/{tag}_pmc_sq.json: two SQ passes of the same command at --reads {sq_reads}; per-wave figures do not depend on the size",
                     "valu_busy": "rocprofv3's VALUBusy: SQ_ACTIVE_INST_VALU / CU_NUM / GRBM_GUI_ACTIVE (per XCD) -- on gfx950 SQ_ACTIVE_INST_VALU equals SQ_INSTS_VALU (one quad-cycle per "
                                  "instruction), so it is 4 / simd_cycles_per_valu_inst and exceeds 1 when the SIMDs issue cheaper-than-4-cycle instructions back to back: >= 1 means saturated; "
                                  "simd_cycles_per_valu_inst = 1024 SIMDs x GRBM_GUI_ACTIVE / 8 / SQ_INSTS_VALU, at the clock the chip held (clock_ghz)"},
           "kernels": {}}
    whole = {"traffic_raw_bytes": 0, "traffic_corrected_bytes": 0, "launches": 0}
    then_shas = sha_then["files"]

    def sha_at_profile(k):
        f = kfiles.get(k.split("<")[0])
        if not f or f not in then_shas:
            return None
        import hashlib
        return hashlib.sha256("|".join([then_shas[f]] + [then_shas[h] for h in sorted(then_shas) if h.endswith(".hpp")]).encode()).hexdigest()[:16]
    for k, v in traffic.items():
        base = k.split("<")[0]
        mult = 2.0 if base in STREAMING else 1.0
        n = max(1, v["launches"])
        e = res["kernels"].setdefault(k, {})
        e.update({"launches_per_step": v["launches"], "traffic_bytes_per_launch": int((v["fetch_bytes"] + v["write_bytes"]) / n),
                  "traffic_corrected_bytes_per_launch": int((mult * v["fetch_bytes"] + v["write_bytes"]) / n),
                  "fetch_bytes_per_launch": int(v["fetch_bytes"] / n), "write_bytes_per_launch": int(v["write_bytes"] / n), "fetch_x2": mult == 2.0})
        if is_step_kernel(k):
            whole["traffic_raw_bytes"] += int(v["fetch_bytes"] + v["write_bytes"]); whole["traffic_corrected_bytes"] += int(mult * v["fetch_bytes"] + v["write_bytes"])
            whole["launches"] += v["launches"]
    for k, d in sq.items():
        w, cyc = d.get("SQ_WAVES", 0), d.get("SQ_WAVE_CYCLES", 0)
        if not w or not cyc:
            continue
        e = res["kernels"].setdefault(k, {})
        secs = d["ms_in_pass"] * 1e-3
        gui = d.get("GRBM_GUI_ACTIVE", 0)                                            # summed over the 8 XCDs: / 8 = cycles the kernel was on the chip
        e.update({"valu_per_wave": int(d.get("SQ_INSTS_VALU", 0) / w), "salu_per_wave": int(d.get("SQ_INSTS_SALU", 0) / w), "lds_per_wave": int(d.get("SQ_INSTS_LDS", 0) / w),
                  "vmem_rd_per_wave": round(d.get("SQ_INSTS_VMEM_RD", 0) / w, 1), "vmem_wr_per_wave": round(d.get("SQ_INSTS_VMEM_WR", 0) / w, 1),
                  "wait_any": round(d.get("SQ_WAIT_ANY", 0) / cyc, 3), "wait_inst": round(d.get("SQ_WAIT_INST_ANY", 0) / cyc, 3), "active": round(d.get("SQ_ACTIVE_INST_ANY", 0) / cyc, 3),
                  "valu_active_of_wave_cycles": round(d.get("SQ_ACTIVE_INST_VALU", 0) / cyc, 3),
                  "valu_busy": round(d.get("SQ_ACTIVE_INST_VALU", 0) / 256.0 / (gui / 8.0), 4) if gui else None,
                  "clock_ghz": round(gui / 8.0 / secs / 1e9, 3) if gui and secs > 0 else None,
                  "simd_cycles_per_valu_inst": round(N_SIMD * (gui / 8.0) / d["SQ_INSTS_VALU"], 3) if gui and d.get("SQ_INSTS_VALU") else None,
                  "ns_per_valu_inst_per_simd": round(secs * N_SIMD / d["SQ_INSTS_VALU"] * 1e9, 4) if secs > 0 and d.get("SQ_INSTS_VALU") else None,
                  "int64_share_of_valu": round(d.get("SQ_INSTS_VALU_INT64", 0) / d["SQ_INSTS_VALU"], 3) if d.get("SQ_INSTS_VALU") and "SQ_INSTS_VALU_INT64" in d else None,
                  "waves_per_simd_avg": round(cyc * 4 / (N_SIMD * (gui / 8.0)), 2) if gui else None,
                  "sq_busy_cycles": d.get("SQ_BUSY_CYCLES"), "grbm_gui_active": d.get("GRBM_GUI_ACTIVE"), "sq_pass_ms": d["ms_in_pass"], "sq_pass_reads": sq_reads})
    for k, e in res["kernels"].items():
        e["source_file"] = kfiles.get(k.split("<")[0])
        e["source_sha"] = sha_at_profile(k)
    res["_whole_step"] = whole
    json.dump(res, open(os.path.join(prof, "pmc_constants.json"), "w"), indent=1, sort_keys=True)
    print(len(res["kernels"]), "kernels; whole step", round(whole["traffic_corrected_bytes"] / 1e9, 1), "GB corrected,", whole["launches"], "launches; commit", commit[:8], "dirty" if dirty else "clean")


if __name__ == "__main__":
    main()
