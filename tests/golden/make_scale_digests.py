#!/usr/bin/env python3
"""Collects the lines oracle/digest_main printed (tests/golden/make_scale_digests.sh) into tests/golden/scale_digests.json.
    python tests/golden/make_scale_digests.py /tmp/dg/c1_100m_150.json /tmp/dg/c2_67m_100.json"""
import json
import os
import sys

runs = []
for path in sys.argv[1:]:
    txt = open(path).read().strip()
    if txt:
        runs.append(json.loads(txt))
doc = {"what": "result digests (mcomh_result_digest's eight numbers) of the sequential oracle, oracle/digest_main SEED N L, over the synthetic read sets of minicom_amd/synth.py",
       "runs": runs}
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scale_digests.json")
json.dump(doc, open(out, "w"), indent=1)
open(out, "a").write("\n")
print(out, [(r["n"], r["L"], r["oracle_seconds"]) for r in runs])
