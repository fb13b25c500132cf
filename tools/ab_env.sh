# usage: bash tools/ab_env.sh "VAR=a" "VAR=b" ...   -- bench.py under each environment setting, twice, same box
for rep in 1 2; do
for cfg in "$@"; do
  env $cfg python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$cfg', round(d['value']), round(d['ms_per_step'],3))"
done; done
