#!/bin/bash
# Counters of the kernels whose name matches PATTERN in ONE run of tools/dist_work.py at world R (R thread-ranks on one card, no warm-up):
# an SQ pass (waves, cycles, waiting, VALU) and the FETCH_SIZE / WRITE_SIZE passes, each a run of its own with --kernel-trace only.
#     bash tools/pmc_dist.sh TAG PATTERN [WORLD] [READS]
set -e -o pipefail
TAG=${1:?tag}; PAT=${2:?pattern}; W=${3:-8}; N=${4:-64000000}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
O=gpurun_out/pmcd_$TAG; rm -rf "$O"; mkdir -p "$O"
RUN="tools/dist_work.py --reads $N --worlds $W --no-baseline --runs 1"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$O/sq" -o s -- python3 $RUN --out "$O/work_sq.json" > "$O/sq.out" 2> "$O/sq.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -o f -- python3 $RUN --out "$O/work_f.json" > "$O/fetch.out" 2> "$O/fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/write" -o w -- python3 $RUN --out "$O/work_w.json" > "$O/write.out" 2> "$O/write.err"
python3 - "$O" "$PAT" <<'PY' | tee "$O/summary.txt"
import csv, glob, sys, collections
O, pats = sys.argv[1], sys.argv[2].split(",")
for sub in ("sq", "fetch", "write"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for f in glob.glob(f"{O}/{sub}/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            if any(p in r["Kernel_Name"] for p in pats):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                if (r["Dispatch_Id"], k) not in seen: seen.add((r["Dispatch_Id"], k)); calls[k] += 1
    for k, v in sorted(agg.items()):
        print(sub, k, "launches", calls[k], {a: f"{b:.5g}" for a, b in sorted(v.items())})
PY
find "$O" -name '*kernel_trace.csv' -delete; find "$O" -name '*agent_info.csv' -delete; find "$O" -name '*counter_collection.csv' -delete
