"""Turn rocprofv3 output into the files kept under profiles/.

  prof_export.py top  RESULTS.db OUT.csv          the top_kernels view of a rocpd database (durations in us)
  prof_export.py pmc  FETCH.csv WRITE.csv OUT.json  per kernel: launches, FETCH_SIZE and WRITE_SIZE in bytes (KB counters x 1024)
"""
import collections
import csv
import json
import sqlite3
import sys


def top(db, out):
    con = sqlite3.connect(db)
    view = [r[0] for r in con.execute("select name from sqlite_master where type='view' and name like 'top_kernels%'")][0]
    cur = con.execute(f"select * from {view}")
    cols = [d[0] for d in cur.description]
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(cols)
        for row in cur:
            w.writerow(row)


def pmc(fetch_csv, write_csv, out):
    res = collections.defaultdict(lambda: {"launches": 0, "fetch_bytes": 0.0, "write_bytes": 0.0})
    for path, key, counter in ((fetch_csv, "fetch_bytes", "FETCH_SIZE"), (write_csv, "write_bytes", "WRITE_SIZE")):
        seen = set()
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] != counter:
                    continue
                k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
                if k.startswith("void "):
                    k = k[5:]
                res[k][key] += float(r["Counter_Value"]) * 1024.0
                if key == "fetch_bytes" and r["Dispatch_Id"] not in seen:
                    seen.add(r["Dispatch_Id"]); res[k]["launches"] += 1
    order = sorted(res, key=lambda k: -(res[k]["fetch_bytes"] + res[k]["write_bytes"]))
    with open(out, "w") as f:
        json.dump({k: res[k] for k in order}, f, indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "top":
        top(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
