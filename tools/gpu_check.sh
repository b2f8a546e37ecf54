#!/bin/bash
# quick loop on the GPU box: the parity tests that cover the merge rounds and Stage 2, the contig-index devbench under the profiler,
# then a short bench line:   bash tools/gpu_check.sh [pytest args...]
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
mkdir -p gpurun_out
python -m pytest ${@:-tests/test_gpu_realign.py tests/test_gpu_pipeline.py tests/test_gpu_contigs.py tests/test_gpu_merge.py tests/test_gpu_claim.py} -x -q -m gpu 2>&1 | tail -4 || exit 1
rm -rf gpurun_out/cx1
rocprofv3 --kernel-trace --stats -d gpurun_out/cx1 -o cx -- python3 tools/devbench_cindex.py > gpurun_out/cx1.log 2> gpurun_out/cx1.err
tail -3 gpurun_out/cx1.log
python3 tools/prof_export.py top "$(find gpurun_out/cx1 -name '*.db')" gpurun_out/cx1_top.csv
rm -rf gpurun_out/cx1
python3 - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/cx1_top.csv')):
    n = r['name'].replace('void ', '').split('(')[0]
    if n.startswith('k_cx'):
        print('%-28s calls %3s avg %9.1f us' % (n[:28], r['total_calls'], float(r['average'])))
PY
python3 bench.py --steps 3 --no-cpu-baseline --no-host-to-host --e2e-reads 0 --no-event-ab > gpurun_out/bench_short.json 2> gpurun_out/bench_short.err || { tail -5 gpurun_out/bench_short.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_short.json').read().strip().splitlines()[-1])
print(d['value'], 'Mreads/s', d['ms_per_step'], 'ms', d['config']['stage_ms_rank0'])
print(d['roofline']['device_ms_per_step_by_kernel'], 'launches', d['whole_step'])
PY
