"""Sum rocprofv3 counter_collection.csv per kernel and counter (kernel name cut at the first parenthesis)."""
import collections
import csv
import sys

tot = collections.defaultdict(lambda: collections.Counter())
calls = collections.Counter()
for path in sys.argv[1:]:
    seen = set()
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"]); calls[k] += 1
names = sorted({c for k in tot for c in tot[k]})
print("kernel".ljust(44), "calls".rjust(6), " ".join(n.rjust(22) for n in names))
key = "SQ_WAVE_CYCLES" if "SQ_WAVE_CYCLES" in names else names[0]
for k in sorted(tot, key=lambda k: -tot[k][key])[:18]:
    print(k[:44].ljust(44), str(calls[k]).rjust(6), " ".join(("%.4g" % tot[k][n]).rjust(22) for n in names))
