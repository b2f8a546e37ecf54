"""Spans between consecutive launches of a marker kernel in the last step of bench.py (e.g. one merge round each):
   prof_rounds.py RESULTS.db MARKER"""
import sqlite3
import sys

db, marker = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
rows = list(con.execute("select name, start, end from kernels order by start"))
starts = [i for i, r in enumerate(rows) if "k_classify_pack" in r[0]]
rows = rows[starts[-1]:]
t0 = rows[0][1]
marks = [(r[1] - t0) / 1e6 for r in rows if marker in r[0]]
print("step span %.1f ms; %s at ms:" % ((max(r[2] for r in rows) - t0) / 1e6, marker), [round(m, 1) for m in marks])
print("gaps:", [round(b - a, 1) for a, b in zip(marks, marks[1:])])
