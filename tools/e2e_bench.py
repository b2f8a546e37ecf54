"""File -> stream files at N reads, stage by stage (tools/e2e.py): python tools/e2e_bench.py [reads] [read_len] [ref_reads]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.e2e import file_to_streams, host_cores
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
ref = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
print("host cores:", host_cores(), flush=True)
print(json.dumps(file_to_streams(n, L, 1002, ref_reads=ref), indent=1), flush=True)
