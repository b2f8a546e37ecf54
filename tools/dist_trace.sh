#!/bin/bash
# rocprofv3 kernel trace of tools/dist_work.py at ONE world size (R thread-ranks of one process on one card): per-kernel totals over all
# ranks, to hold against the same trace at world 1 -- what is replicated shows up as a kernel whose total grows with R.
#     bash tools/dist_trace.sh TAG WORLD [READS]
set -e -o pipefail
TAG=${1:?tag}; W=${2:?world}; N=${3:-64000000}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
O=gpurun_out/dist_$TAG
mkdir -p "$O"; rm -rf "$O/kt_$W"
rocprofv3 --kernel-trace --stats -d "$O/kt_$W" -o kt -- python3 tools/dist_work.py --reads "$N" --worlds "$W" --no-baseline --runs 1 --out "$O/work_$W.json" > "$O/log_$W.txt" 2>&1
DB=$(find "$O/kt_$W" -name '*.db' | head -1)
python3 tools/prof_export.py top "$DB" "$O/kernels_$W.csv"
rm -rf "$O/kt_$W"
head -5 "$O/kernels_$W.csv"
