"""Timeline of ONE step of bench.py from a rocprofv3 rocpd database: busy time per kernel, idle gaps, launches.
   prof_timeline.py RESULTS.db [STEP]   (a step starts at a k_classify_* dispatch; default: the last one)"""
import collections
import sqlite3
import sys

db = sys.argv[1]
if db.endswith(".csv"):                                                        # rocprofv3 --output-format csv: *_kernel_trace.csv
    import csv
    rows = sorted(((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(db))), key=lambda r: r[1])
else:
    con = sqlite3.connect(db)
    rows = list(con.execute("select name, start, end from kernels order by start"))
starts = [i for i, r in enumerate(rows) if r[0].replace("void ", "").replace("(anonymous namespace)::", "").startswith("k_classify_")]
step = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 1
a = starts[step]; b = starts[step + 1] if step + 1 < len(starts) else len(rows)
rows = rows[a:b]
t0, t1 = rows[0][1], max(r[2] for r in rows)
busy = collections.Counter(); calls = collections.Counter()
gap = 0; last_end = rows[0][1]; gaps = []
for name, s, e in rows:
    k = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    busy[k] += e - s; calls[k] += 1
    if s > last_end:
        gap += s - last_end; gaps.append((s - last_end, k))
    last_end = max(last_end, e)
tot = sum(busy.values())
print(f"step {step}: {len(rows)} launches, span {(t1 - t0) / 1e6:.1f} ms, kernel time {tot / 1e6:.1f} ms, idle {gap / 1e6:.1f} ms")
for k, v in busy.most_common(45):
    print(f"{v / 1e6:8.2f} ms {calls[k]:5d}  {k[:70]}")
print("largest gaps (ms, before kernel):", [(round(g / 1e6, 2), k[:24]) for g, k in sorted(gaps, reverse=True)[:14]])
byk = collections.Counter(); cnt = collections.Counter()
for g, k in gaps:
    byk[k] += g; cnt[k] += 1
print("idle by the kernel that ended it (ms, count):", [(k[:22], round(v / 1e6, 2), cnt[k]) for k, v in byk.most_common(16)])
hist = collections.Counter()
for g, _ in gaps:
    hist[min(6, len(str(g // 1000)))] += g
print("idle by gap size (digits of us):", {k: round(v / 1e6, 1) for k, v in sorted(hist.items())})
big = sorted(((e - s, i) for i, (n, s, e) in enumerate(rows) if "copyBuffer" in n or "fillBuffer" in n), reverse=True)[:16]
short = lambda n: n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:28]
print("largest copies / fills (ms, kind, kernel before, kernel after):", [(round(d / 1e6, 2), "copy" if "copy" in rows[i][0] else "fill", short(rows[i - 1][0]) if i else "", short(rows[i + 1][0]) if i + 1 < len(rows) else "") for d, i in big])
