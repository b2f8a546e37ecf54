"""Prints the figures of a bench.py line that a change is judged by.   python tools/show_line.py LOGFILE"""
import json
import sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")][-1]
d = json.loads(l)
print({k: d[k] for k in ("value", "ms_per_step", "steps", "n_gpus")})
print(d["config"]["stage_ms_rank0"])
print(d["config"].get("per_step"))
print("digest", d["result"]["digest"], d["result"].get("every_timed_step_equal"))
for k in ("value_host_to_host", "value_file_to_streams", "value_file_to_streams_gz", "value_file_to_streams_gz_one_member"):
    if k in d:
        print(k, d[k] if not isinstance(d[k], dict) else d[k].get("value"))
