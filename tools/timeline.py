"""Phase timeline of one replayed step from a rocprofv3 kernel trace:
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline
  python tools/timeline.py gpurun_out/trace
Takes the shortest step (from one gm3d scale_translate launch to the next: a graph replay), prints where the wall time of the step goes
(phases delimited by marker kernels), how much of it has >= 1 kernel running, and the average number of kernels in flight."""
import csv, glob, os, sys

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "scale_translate" in r[2]]
# the shortest step of the trace = a hipGraph replay (the eager probe / re-run steps are host-bound and much longer)
a, b = min(zip(starts[:-1], starts[1:]), key=lambda ab: rows[ab[1]][0] - rows[ab[0]][0])
step = rows[a:b]
t0, t1 = step[0][0], rows[b][0]
print("step wall %.3f ms, %d kernels, summed kernel time %.3f ms" % ((t1 - t0) / 1e6, len(step), sum(e - s for s, e, _ in step) / 1e6))
# busy time (union of intervals)
busy, cur_s, cur_e = 0, None, None
for s, e, _ in step:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("time with >= 1 kernel running %.3f ms (idle %.3f ms)" % (busy / 1e6, (t1 - t0 - busy) / 1e6))
markers = [("fps_kernel", "FPS"), ("knn", "KNN"), ("mask_select", "mask (teacher done)"), ("group_select_maps", "student embed conv4"),
           ("chamfer32_fwd", "losses (student fwd done)"), ("chamfer32_bwd", "backward starts"), ("group_max_bwd", "embed backward starts"),
           ("flat_sumsq", "optimizer"), ("adamw_ema_flat", "adamw")]
seen = set()
for s, e, n in step:
    for key, label in markers:
        if key in n and key not in seen:
            seen.add(key)
            print("  +%7.3f ms  %s" % ((s - t0) / 1e6, label))
print("  +%7.3f ms  end" % ((t1 - t0) / 1e6))
