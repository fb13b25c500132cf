#!/bin/bash
# same-box A/B of module switches: tools/ab.sh OUTDIR "label1:sw1 sw2" "label2:..." ...   (each variant run twice, interleaved)
out=$1; shift
mkdir -p $out
for rep in 1 2; do
  for spec in "$@"; do
    label=${spec%%:*}; sw=${spec#*:}
    python tools/bench_with.py $sw -- --steps 40 --warmup 10 --no-cpu-baseline --no-secondary > $out/ab_${label}_$rep.json 2> $out/ab_${label}_$rep.err
    python - <<PY
import json
d=json.loads(open("$out/ab_${label}_$rep.json").read().strip().splitlines()[-1])
print("%-28s rep $rep  %8.0f clouds/s  %.3f ms  (min %.3f med %.3f max %.3f)" % ("$label", d["value"], d["ms_per_step"], d["ms_per_step_spread"]["min"], d["ms_per_step_spread"]["median"], d["ms_per_step_spread"]["max"]))
PY
  done
done
