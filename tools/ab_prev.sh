# usage: bash tools/ab_prev.sh   -- bench.py of this tree vs the tree checked out under _prev/ (git worktree of an older commit), same box, alternating
for rep in 1 2 3; do
  for d in _prev .; do
    (cd $d && python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$d', round(d['value']), round(d['ms_per_step'],3))")
  done
done
