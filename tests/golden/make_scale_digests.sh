#!/bin/bash
# The sequential oracle (oracle/mcom_oracle.c, pinned on the reference's own dumps) over the read sets the benchmark is quoted on; its result
# digests go to tests/golden/scale_digests.json, which tests/test_gpu_scale.py compares the GPU pipeline's digests with.  Build container
# only: hours of one core and ~25 GB per run (100 M x 150 bp: Stage 1 17 min, Stage 2 several hours).
#     bash tests/golden/make_scale_digests.sh            (then: python tests/golden/make_scale_digests.py OUT/*.json)
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$HERE/../..
make -C "$ROOT/oracle" digest_main
OUT=${OUT:-/tmp/dg}; mkdir -p "$OUT"
"$ROOT/oracle/digest_main" 1002 100000000 150 > "$OUT/c1_100m_150.json" 2> "$OUT/c1_100m_150.log"     # BASELINE configs[1]
"$ROOT/oracle/digest_main" 1003 67000000 100 > "$OUT/c2_67m_100.json" 2> "$OUT/c2_67m_100.log"       # configs[2]'s shape
