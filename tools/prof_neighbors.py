"""Dispatches of kernels whose name contains PATTERN in the last step of bench.py, longest first, each with the kernels launched
   before and after it (to tell which host call a copy or fill belongs to).  prof_neighbors.py RESULTS.db PATTERN [N]"""
import sqlite3
import sys

db, pat, top = sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 25
con = sqlite3.connect(db)
rows = list(con.execute("select name, start, end, grid_x from kernels order by start"))
starts = [i for i, r in enumerate(rows) if "k_classify_" in r[0]]
rows = rows[starts[-1]:]
short = lambda s: s.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:28]
hits = [(r[2] - r[1], i) for i, r in enumerate(rows) if pat in r[0]]
print("total ms", sum(h[0] for h in hits) / 1e6, "count", len(hits))
for d, i in sorted(hits, reverse=True)[:top]:
    print(f"{d / 1e3:9.1f} us grid {rows[i][3]:>12}  after {short(rows[i - 1][0]) if i else '-':28s} before {short(rows[i + 1][0]) if i + 1 < len(rows) else '-'}")
