"""From tools/trace_dump.py-style input (a rocprofv3 kernel trace directory): per kernel name, the time it ran ALONE in the shortest
(replayed) step versus overlapped -- a kernel's speed-up converts 1:1 to wall time only for its exclusive part.
   python tools/exclusive_time.py <trace dir>"""
import csv, glob, os, sys
from collections import defaultdict

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "scale_translate" in r[2]]
a, b = min(zip(starts[:-1], starts[1:]), key=lambda ab: rows[ab[1]][0] - rows[ab[0]][0])
step = rows[a:b]
ev = []
for i, (s, e, n) in enumerate(step):
    ev.append((s, 1, i))
    ev.append((e, 0, i))
ev.sort()
live, last = set(), None
excl, tot, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
conc = defaultdict(float)
for t, kind, i in ev:
    if last is not None and live:
        conc[min(len(live), 4)] += t - last
        if len(live) == 1:
            excl[short(step[next(iter(live))][2]) if False else next(iter(live))] += t - last
    last = t
    if kind:
        live.add(i)
    else:
        live.discard(i)


def short(n):
    s = n.split("(")[0].replace("void ", "").replace("gm3d::", "")
    if s.startswith("Cijk") or s.startswith("Custom"):
        s = "LIB " + n[n.find("MT"):n.find("MT") + 14]
    return s[:70]


ex2 = defaultdict(float)
for i, v in excl.items():
    ex2[short(step[i][2])] += v
for s, e, n in step:
    tot[short(n)] += e - s
    cnt[short(n)] += 1
wall = rows[b][0] - step[0][0]
print("step wall %.3f ms; time with 1 / 2 / 3 / >=4 kernels running: %s ms" % (wall / 1e6, " / ".join("%.3f" % (conc[k] / 1e6) for k in (1, 2, 3, 4))))
print("%-72s %5s %9s %9s" % ("kernel", "calls", "total us", "alone us"))
for n in sorted(tot, key=lambda n: -ex2[n])[:32]:
    print("%-72s %5d %9.1f %9.1f" % (n, cnt[n], tot[n] / 1e3, ex2[n] / 1e3))
