"""profiles/pmc_constants.json (what bench.py reads for `roofline.traffic` and the VALU-issue roofline) from the PMC passes:
   pmc_constants.py TRAFFIC.json SQ_RAW.txt OUT.json TAG"""
import json
import re
import sys

traffic, sq, out, tag = sys.argv[1:5]
# Kernels whose loads are wide coalesced streams (every lane of a wave reads consecutive 8- or 16-byte words of arrays walked front
# to back): on gfx950 FETCH_SIZE tallies such 128-byte requests at 64 bytes (MI355X_MICROARCH.md, "HBM"), so their FETCH is doubled
# for `traffic_corrected`.  Calibration in this code base: k_classify_pack16 reads 15.0 GB of ASCII and FETCH_SIZE says 8.1; k_cx_scatter2
# reads 12 bytes x 0.88 G entries = 10.6 GB and FETCH_SIZE says 5.4.  Kernels that gather (table lookups, row gathers, per-object
# kernels) keep their raw count.
STREAMING = ("k_classify_flat", "k_classify_pack16", "k_classify_pack", "k_cx_hist2", "k_cx_scatter2", "k_cx_bounds", "k_cx_assemble_sorted", "k_radix_hist", "k_radix_scatter", "k_scan_tile", "k_scan_add",
             "k_scan64_tile", "k_scan64_add", "k_digest", "k_mask_records", "k_st_refbin", "k_table_heads", "k_bucket_starts", "k_live_flags", "k_prefix_copy", "k_min_fold")
res = {"_source": {"traffic": f"profiles/{tag}_pmc_traffic_100m.json (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 passes of `bench.py --steps 1 --warmup 0`, 100 M x 150 bp; "
                              "raw = counters x 1024; corrected = FETCH x 2 for the wide streaming readers listed in tools/pmc_constants.py)",
                   "sq": f"profiles/{tag}_pmc_sq_32m.txt (SQ_* counters, one pass of tools/devbench_pipeline.py at 32 M reads; per-wave instruction counts do not depend on the size)"},
       "kernels": {}}
d = json.load(open(traffic))
whole = {"traffic_raw_bytes": 0, "traffic_corrected_bytes": 0, "launches": 0}
for k, v in d.items():
    if isinstance(v, dict) and "fetch_bytes" in v:
        base = k.split("<")[0]
        mult = 2.0 if base in STREAMING else 1.0
        L_ = max(1, v["launches"])
        res["kernels"].setdefault(k, {})["traffic_bytes_per_launch"] = int((v["fetch_bytes"] + v["write_bytes"]) / L_)
        res["kernels"][k]["traffic_corrected_bytes_per_launch"] = int((mult * v["fetch_bytes"] + v["write_bytes"]) / L_)
        res["kernels"][k]["fetch_bytes_per_launch"] = int(v["fetch_bytes"] / L_)
        res["kernels"][k]["write_bytes_per_launch"] = int(v["write_bytes"] / L_)
        res["kernels"][k]["fetch_x2"] = mult == 2.0
        if not (base.startswith("k_synth") or base.startswith("at::") or "at::native" in k or base.startswith("void at::")):   # the generator and the checker are not the step
            whole["traffic_raw_bytes"] += int(v["fetch_bytes"] + v["write_bytes"]); whole["traffic_corrected_bytes"] += int(mult * v["fetch_bytes"] + v["write_bytes"]); whole["launches"] += v["launches"]
res["_whole_step"] = whole
lines = open(sq).read().splitlines()
names = lines[0].split()[2:]
for ln in lines[1:]:
    m = re.match(r"(.+?)\s+(\d+)\s+((?:[\d.e+]+\s*)+)$", ln)
    if not m:
        continue
    vals = dict(zip(names, (float(x) for x in m.group(3).split())))
    k = m.group(1).strip().replace("void ", "")
    w, cyc = vals.get("SQ_WAVES", 0), vals.get("SQ_WAVE_CYCLES", 0)
    if not w or not cyc:
        continue
    e = res["kernels"].setdefault(k, {})
    e.update({"valu_per_wave": int(vals["SQ_INSTS_VALU"] / w), "salu_per_wave": int(vals["SQ_INSTS_SALU"] / w), "lds_per_wave": int(vals["SQ_INSTS_LDS"] / w),
              "wait_any": round(vals["SQ_WAIT_ANY"] / cyc, 2), "wait_inst": round(vals["SQ_WAIT_INST_ANY"] / cyc, 2), "active": round(vals["SQ_ACTIVE_INST_ANY"] / cyc, 2)})
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(len(res["kernels"]), "kernels")
