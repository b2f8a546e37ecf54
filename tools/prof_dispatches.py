"""List the longest dispatches of kernels whose name contains a pattern, from a rocprofv3 rocpd database.
   prof_dispatches.py RESULTS.db PATTERN [N]"""
import sqlite3
import sys

db, pat, top = sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 30
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("kernels")] or [t for t in tabs if "kernel_dispatch" in t]
view = kd[0]
cur = con.execute(f"select * from {view} limit 1")
cols = [d[0] for d in cur.description]
print(view, cols, file=sys.stderr)
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
dur = "duration" if "duration" in cols else None
q = f"select {name_col}, " + (dur if dur else "(end - start)") + " as d, * from " + view + f" where {name_col} like ? order by d desc limit {top}"
for r in con.execute(q, ("%" + pat + "%",)):
    print(r[0][:40], r[1], {c: v for c, v in zip(cols, r[2:]) if c in ("grid_size", "grid_x", "workgroup_size", "start", "grid_size_x")})
