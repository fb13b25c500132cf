#!/bin/bash
# same-box A/B of module switches on the Point-M2AE step: tools/ab_m2ae.sh OUTDIR "label1:--set gemm.WS_BN=False" "label2:" ...  (each twice, interleaved)
out=$1; shift
mkdir -p $out
for rep in 1 2; do
  for spec in "$@"; do
    label=${spec%%:*}; sw=${spec#*:}
    python tools/bench_m2ae.py --steps 20 --warmup 3 $sw > $out/abm_${label}_$rep.json 2> $out/abm_${label}_$rep.err
    python - <<PY
import json
d=json.loads(open("$out/abm_${label}_$rep.json").read().strip().splitlines()[-1])
print("%-28s rep $rep  %8.0f clouds/s  %.3f ms" % ("$label", d["value"], d["ms_per_step"]))
PY
  done
done
