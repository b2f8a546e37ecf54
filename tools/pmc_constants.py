"""profiles/pmc_constants.json (what bench.py reads for `roofline.traffic` and the VALU-issue roofline) from the PMC passes:
   pmc_constants.py TRAFFIC.json SQ_RAW.txt OUT.json TAG"""
import json
import re
import sys

traffic, sq, out, tag = sys.argv[1:5]
res = {"_source": {"traffic": f"profiles/{tag}_pmc_traffic_100m.json (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 passes of `bench.py --steps 1 --warmup 0`, 100 M x 150 bp; "
                              "raw counters x 1024, the guide's x2 for 16-byte-per-lane streaming reads NOT applied)",
                   "sq": f"profiles/{tag}_pmc_sq_32m.txt (SQ_* counters, one pass of tools/devbench_pipeline.py at 32 M reads; per-wave instruction counts do not depend on the size)"},
       "kernels": {}}
d = json.load(open(traffic))
for k, v in d.items():
    if isinstance(v, dict) and "fetch_bytes" in v:
        res["kernels"].setdefault(k, {})["traffic_bytes_per_launch"] = int((v["fetch_bytes"] + v["write_bytes"]) / max(1, v["launches"]))
        res["kernels"][k]["fetch_bytes_per_launch"] = int(v["fetch_bytes"] / max(1, v["launches"]))
        res["kernels"][k]["write_bytes_per_launch"] = int(v["write_bytes"] / max(1, v["launches"]))
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
