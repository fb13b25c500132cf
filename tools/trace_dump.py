"""Kernel-by-kernel dump of one replayed step from a rocprofv3 kernel trace (see tools/timeline.py for how to record one):
   python tools/trace_dump.py <trace dir> [first] [count]   ->  start offset, duration, gap since the previous end, queue, name"""
import csv, glob, os, sys

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
starts = [i for i, r in enumerate(rows) if "scale_translate" in r[2]]
a, b = min(zip(starts[:-1], starts[1:]), key=lambda ab: rows[ab[1]][0] - rows[ab[0]][0])
step = rows[a:b]
t0 = step[0][0]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else len(step)
prev_end = t0
for i, (s, e, n, q) in enumerate(step):
    if first <= i < first + count:
        short = n.split("(")[0].replace("void ", "").replace("gm3d::", "")[:60]
        if short.startswith("Cijk") or short.startswith("Custom"):
            short = "LIB " + n[n.find("MT"):n.find("MT") + 14]
        print("%4d  +%8.1f us  dur %7.1f  gap %7.1f  q%-3s %s" % (i, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, q, short))
    prev_end = max(prev_end, e)
print("step wall %.3f ms, %d kernels" % ((rows[b][0] - t0) / 1e6, len(step)))
