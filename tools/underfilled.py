"""Which launches of a replayed step leave most of the chip idle?  From a rocprofv3 kernel trace: per kernel name, launches, time and
workgroups per launch; kernels under 256 workgroups (one per CU) listed by time.   python tools/underfilled.py <trace dir>"""
import csv, glob, os, sys
from collections import defaultdict

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        wgs = 1
        for ax in "XYZ":
            wgs *= max(int(r["Grid_Size_" + ax]) // max(int(r["Workgroup_Size_" + ax]), 1), 1)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], wgs))
rows.sort()
starts = [i for i, r in enumerate(rows) if "scale_translate" in r[2]]
a, b = min(zip(starts[:-1], starts[1:]), key=lambda ab: rows[ab[1]][0] - rows[ab[0]][0])
step = rows[a:b]


def short(n):
    s = n.split("(")[0].replace("void ", "").replace("gm3d::", "")
    if s.startswith("Cijk") or s.startswith("Custom"):
        s = "LIB " + n[n.find("MT"):n.find("MT") + 14]
    return s[:64]


acc = defaultdict(lambda: [0, 0.0, 0])
for s, e, n, wgs in step:
    k = (short(n), wgs)
    acc[k][0] += 1
    acc[k][1] += (e - s) / 1e3
tot = sum(v[1] for v in acc.values())
under = sum(v[1] for k, v in acc.items() if k[1] < 256)
print("summed kernel time %.0f us; in launches of < 256 workgroups: %.0f us (%.0f %%); < 64 workgroups: %.0f us" %
      (tot, under, 100 * under / tot, sum(v[1] for k, v in acc.items() if k[1] < 64)))
print("%-66s %6s %6s %9s %7s" % ("kernel", "wgs", "calls", "total us", "avg us"))
for (n, wgs), v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if wgs < 256 and v[1] > 15:
        print("%-66s %6d %6d %9.1f %7.1f" % (n, wgs, v[0], v[1], v[1] / v[0]))
