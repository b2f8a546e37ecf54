"""VALU/SALU instruction counts per source line of one kernel, from a `hipcc -gline-tables-only -save-temps` .s file."""
import collections
import re
import sys

path, kernel = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith(kernel) and l.rstrip().split(";")[0].rstrip().endswith(":")][0]
end = [i for i, l in enumerate(lines) if i > start and "s_endpgm" in l][0]
files, cur = {}, None
hv, hs, hl = collections.Counter(), collections.Counter(), collections.Counter()
for l in lines[:start]:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = m.group(3) or m.group(2)
for l in lines[start:end]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)).split("/")[-1], int(m.group(2)))
        continue
    if re.match(r"\s+v_", l):
        hv[cur] += 1
    elif re.match(r"\s+s_", l):
        hs[cur] += 1
    elif re.match(r"\s+ds_", l):
        hl[cur] += 1
print("VALU", sum(hv.values()), "SALU", sum(hs.values()), "LDS", sum(hl.values()))
for k, v in sorted(hv.items(), key=lambda kv: -kv[1])[:top]:
    print(k, "valu", v, "salu", hs[k], "lds", hl[k])
