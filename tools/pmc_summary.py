"""Summarise rocprofv3 --pmc passes into per-kernel, per-grid, per-launch averages (the file bench.py's roofline.traffic reads).

  rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch --output-format csv -- python3 bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write --output-format csv -- python3 bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline
  python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/rNN_pmc_fetch_write_per_launch.json

Counters are KB; on gfx950 FETCH_SIZE counts half of a coalesced stream (MI355X_MICROARCH.md), so traffic = (2*FETCH + WRITE)*1024."""
import collections, csv, glob, json, os, re, sys


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(gm3d::\w+)(<[^>]*>)?", name)
    if m:
        t = m.group(2) or ""
        t = "<bf16>" if "bf16" in t or "__bf16" in t else ("<float>" if "float" in t else "")
        return m.group(1) + t
    m = re.match(r"_ZN4gm3d\d+(\w+?_kernel)I(DF16b|f)", name)
    if m:
        return "gm3d::%s<%s>" % (m.group(1), "bf16" if m.group(2) == "DF16b" else "float")
    return name[:120]


FULL = "--full-names" in sys.argv          # keep the template arguments: one row per kernel INSTANCE and grid (diagnosis; bench.py reads the short form)
if FULL:
    sys.argv.remove("--full-names")
    _short = short
    short = lambda name: re.sub(r"\(.*", "", re.sub(r"^void ", "", name))[:100] if name.startswith(("void gm3d", "gm3d")) else _short(name)
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            key = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
            a = acc[key][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
out = []
for (k, grid), ctr in sorted(acc.items()):
    if not k.startswith("gm3d::"):
        continue
    row = {"kernel": k, "grid_threads": grid, "launches": max(v[0] for v in ctr.values())}
    for c, (n, s) in ctr.items():
        row[c + "_KB_avg"] = round(s / n, 1)
    out.append(row)
json.dump(out, sys.stdout, indent=0)
