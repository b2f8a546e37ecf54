"""Turn rocprofv3 output into the files kept under profiles/.

  prof_export.py top  RESULTS.db OUT.csv          the top_kernels view of a rocpd database (durations in us)
  prof_export.py pmc  FETCH.csv WRITE.csv OUT.json  per kernel: launches, FETCH_SIZE and WRITE_SIZE in bytes (KB counters x 1024)
  prof_export.py trace RESULTS.db OUT.csv [STEP]   every launch of one bench step (a step starts at a k_classify_* dispatch): name, start (us from the step's start), duration (us)
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


def trace(db, out, step=None):
    con = sqlite3.connect(db)
    have = [d[0] for d in con.execute("select * from kernels limit 0").description]
    extra = [c for c in ("grid_x", "workgroup_x", "lds_size", "vgpr_count", "scratch_size") if c in have]       # launch shape, when the view has it
    rows = list(con.execute("select name, start, end%s from kernels order by start" % "".join(", " + c for c in extra)))
    short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "")
    starts = [i for i, r in enumerate(rows) if short(r[0]).startswith("k_classify_")]
    step = len(starts) - 1 if step is None else step
    a = starts[step]; b = starts[step + 1] if step + 1 < len(starts) else len(rows)
    t0 = rows[a][1]
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "start_us", "duration_us"] + extra)
        for n, s, e, *x in rows[a:b]:
            n = short(n)
            depth = 0
            for i, ch in enumerate(n):
                if ch == "<": depth += 1
                elif ch == ">": depth -= 1
                elif ch == "(" and depth == 0:
                    n = n[:i]; break
            w.writerow([n, round((s - t0) / 1e3, 2), round((e - s) / 1e3, 2)] + list(x))


if __name__ == "__main__":
    if sys.argv[1] == "top":
        top(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "trace":
        trace(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else None)
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
