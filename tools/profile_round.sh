#!/bin/bash
# The whole evidence set of one state of the code, in ONE call on the GPU box (from the repo root):
#     bash tools/profile_round.sh TAG [reads_for_sq_passes]
# writes raw rocprofv3 output under gpurun_out/prof_TAG/; back in the build container
#     python tools/pmc_constants.py gpurun_out/prof_TAG TAG
# turns it into profiles/TAG_* and profiles/pmc_constants.json (stamped with the commit and the source hashes).
# Counter passes are separate runs with --kernel-trace only (FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md).
# The program itself follows `--` (python3 bench.py ...): no env / bash -c hop under the profiler.
set -e -o pipefail
TAG=${1:?tag}
SQ_READS=${2:-32000000}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
O=gpurun_out/prof_$TAG
rm -rf "$O"; mkdir -p "$O"
python3 tools/source_sha.py > "$O/source_sha.json"
PROFILED="--steps 2 --warmup 1 --no-event-ab --no-cpu-baseline --no-host-to-host --e2e-reads 0"
COUNTED="--steps 1 --warmup 0 --no-event-ab --no-cpu-baseline --no-host-to-host --no-check --e2e-reads 0"
echo "[1/6] kernel trace + stats: bench.py $PROFILED"
rocprofv3 --kernel-trace --stats -d "$O/kt" -o kt -- python3 bench.py $PROFILED > "$O/line_profiled.json" 2> "$O/kt.err"
DB=$(find "$O/kt" -name '*.db' | head -1)
python3 tools/prof_export.py top "$DB" "$O/kernel_stats.csv"
python3 tools/prof_timeline.py "$DB" 2 > "$O/timeline.txt"
python3 tools/prof_export.py trace "$DB" "$O/kernel_trace_step2.csv" 2
rm -f "$DB"                                                        # tens of MB; what is kept is the three summaries above
echo "[2/6] FETCH_SIZE: bench.py $COUNTED"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -o f -- python3 bench.py $COUNTED > "$O/fetch.out" 2> "$O/fetch.err"
echo "[3/6] WRITE_SIZE"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/write" -o w -- python3 bench.py $COUNTED > "$O/write.out" 2> "$O/write.err"
echo "[4/6] SQ pass A (issue / busy) at $SQ_READS reads"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$O/sqa" -o s -- python3 bench.py $COUNTED --reads "$SQ_READS" > "$O/sqa.out" 2> "$O/sqa.err"
echo "[5/6] SQ pass B (instruction mix)"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_LDS_BANK_CONFLICT \
    --output-format csv -d "$O/sqb" -o s -- python3 bench.py $COUNTED --reads "$SQ_READS" > "$O/sqb.out" 2> "$O/sqb.err"
echo "[6/6] the unprofiled line: python3 bench.py (defaults)"
python3 bench.py > "$O/line_default.json" 2> "$O/default.err"
# keep what travels back small: the counter CSVs only
find "$O" -name '*kernel_trace.csv' -delete
find "$O" -name "*agent_info.csv" -delete
find "$O" -name "*counter_collection.csv" -exec gzip -9 {} +
du -sh "$O"
tail -c 600 "$O/line_default.json"
