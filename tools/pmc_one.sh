#!/bin/bash
# SQ counters of the kernels whose name matches PATTERN, from one bench step at 32 M reads (two rocprofv3 --pmc passes, kernel-trace only):
#     bash tools/pmc_one.sh TAG PATTERN
set -e -o pipefail
TAG=${1:?tag}; PAT=${2:?pattern}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
O=gpurun_out/pmc_$TAG; rm -rf "$O"; mkdir -p "$O"
COUNTED="--steps 1 --warmup 0 --no-event-ab --no-cpu-baseline --no-host-to-host --no-check --e2e-reads 0 --reads 32000000"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
    --output-format csv -d "$O/a" -o s -- python3 bench.py $COUNTED > "$O/a.out" 2> "$O/a.err"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --output-format csv -d "$O/b" -o s -- python3 bench.py $COUNTED > "$O/b.out" 2> "$O/b.err"
python3 - "$O" "$PAT" <<'PY'
import csv, glob, sys, collections
O, pat = sys.argv[1], sys.argv[2]
for sub in ("a", "b"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"{O}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        print(sub, k, {a: f"{b:.4g}" for a, b in sorted(v.items())})
PY
