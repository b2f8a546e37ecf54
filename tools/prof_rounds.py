"""Spans between consecutive launches of a marker kernel in the last step of bench.py (e.g. one merge round each):
   prof_rounds.py RESULTS.db MARKER"""
import sqlite3
import sys

db, marker = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
rows = list(con.execute("select name, start, end from kernels order by start"))
starts = [i for i, r in enumerate(rows) if "k_classify_" in r[0]]
rows = rows[starts[-1]:]
t0 = rows[0][1]
marks = [(r[1] - t0) / 1e6 for r in rows if marker in r[0]]
print("step span %.1f ms; %s at ms:" % ((max(r[2] for r in rows) - t0) / 1e6, marker), [round(m, 1) for m in marks])
print("gaps:", [round(b - a, 1) for a, b in zip(marks, marks[1:])])
if len(sys.argv) > 3:                                   # top kernels inside the gap with that index
    import collections
    gi = int(sys.argv[3])
    a, b = marks[gi] * 1e6 + t0, marks[gi + 1] * 1e6 + t0
    busy = collections.Counter(); calls = collections.Counter()
    last = a; idle = 0
    for name, s, e in rows:
        if s < a or s >= b:
            continue
        k = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        busy[k] += e - s; calls[k] += 1
        if s > last:
            idle += s - last
        last = max(last, e)
    print("gap %d: %.1f ms, kernels %.1f ms, idle %.1f ms, launches %d" % (gi, (b - a) / 1e6, sum(busy.values()) / 1e6, idle / 1e6, sum(calls.values())))
    for k, v in busy.most_common(22):
        print("  %7.2f ms %4d  %s" % (v / 1e6, calls[k], k[:60]))
