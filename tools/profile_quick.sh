#!/bin/bash
# Kernel trace of two bench steps only (the first pass of tools/profile_round.sh): per-kernel totals, the timeline of step 2 and its
# trace, under gpurun_out/quick_TAG/ -- for comparing a change before the whole evidence set is collected again.
#     bash tools/profile_quick.sh TAG
set -e -o pipefail
TAG=${1:?tag}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
O=gpurun_out/quick_$TAG
rm -rf "$O"; mkdir -p "$O"
rocprofv3 --kernel-trace --stats -d "$O/kt" -o kt -- python3 bench.py --steps 2 --warmup 1 --no-event-ab --no-cpu-baseline --no-host-to-host --e2e-reads 0 > "$O/line_profiled.json" 2> "$O/kt.err"
DB=$(find "$O/kt" -name '*.db' | head -1)
python3 tools/prof_export.py top "$DB" "$O/kernel_stats.csv"
python3 tools/prof_timeline.py "$DB" 2 > "$O/timeline.txt"
python3 tools/prof_export.py trace "$DB" "$O/kernel_trace_step2.csv" 2
rm -rf "$O/kt"
head -40 "$O/timeline.txt"
