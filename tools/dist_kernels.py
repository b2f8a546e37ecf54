"""Holds the kernel totals of two rocprofv3 traces of tools/dist_work.py against each other (tools/dist_trace.sh): world 1 and world R on one
card.  A kernel whose total over all ranks exceeds its world-1 time is replicated (or bound by a per-launch floor) by that much.
    python tools/dist_kernels.py kernels_1.csv kernels_8.csv [ranks]"""
import csv, re, sys

def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        n = r["name"].replace("void ", "").replace("(anonymous namespace)::", "")
        n = re.sub(r"\((unsigned|mcom|CixGeom|int|char|uint|float|HIP|const|long|bool|Fs|Rj|Seg|Job)[^)]*\).*$", "", n)[:60]
        t, c = d.get(n, (0.0, 0))
        d[n] = (t + float(r["total_duration"]) / 1000, c + int(r["total_calls"]))
    return d

a, b = load(sys.argv[1]), load(sys.argv[2])
R = int(sys.argv[3]) if len(sys.argv) > 3 else 8
skip = lambda k: k.startswith(("at::", "k_synth", "__amd", "k_digest"))
ta = sum(v[0] for k, v in a.items() if not skip(k)); tb = sum(v[0] for k, v in b.items() if not skip(k))
print("kernel time: world 1 %.1f ms; world %d, all ranks together %.1f ms = %.2f x (1.00 = nothing replicated); per rank %.1f ms against %.1f ideal" % (ta, R, tb, tb / ta, tb / R, ta / R))
rows = sorted(((b.get(k, (0, 0))[0] - a.get(k, (0, 0))[0], k, a.get(k, (0, 0)), b.get(k, (0, 0))) for k in set(a) | set(b) if not skip(k)), reverse=True)
print("%10s  %-58s %16s %16s" % ("excess ms", "kernel", "world 1 ms/calls", "world %d ms/calls" % R))
for ex, k, x, y in rows:
    if abs(ex) >= 0.3:
        print("%10.2f  %-58s %9.2f/%-6d %9.2f/%-6d" % (ex, k, x[0], x[1], y[0], y[1]))
