"""Summarise a rocprofv3 kernel_stats.csv by category.  python tools/prof_summary.py <csv> <steps>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 0
def cat(n):
    if n.startswith('Cijk'): return 'gemm (hipBLASLt)'
    if 'batch_norm' in n: return 'batchnorm'
    if 'layer_norm' in n or 'GammaBeta' in n: return 'layernorm (torch)'
    if 'gm3d::' in n: return 'gm3d:' + n.split('gm3d::')[1].split('(')[0].split('<')[0]
    if n.startswith('_ZN4gm3d'):          # mangled template instantiations: _ZN4gm3d<len><name>I...
        import re
        m = re.match(r'_ZN4gm3d(\d+)', n)
        if m:
            k = int(m.group(1)); st = len(m.group(0))
            return 'gm3d:' + n[st:st + k]
    if 'bfloat16_copy' in n or 'bfloat16tofloat32' in n or 'float32tobfloat16' in n or 'float_copy' in n: return 'dtype-cast'
    if 'reduce_kernel' in n: return 'reduce'
    if 'multi_tensor' in n: return 'optimizer/foreach'
    if 'Gelu' in n: return 'gelu (torch)'
    if 'fillBuffer' in n or 'copyBuffer' in n: return 'memset/copy'
    if 'elementwise' in n: return 'elementwise'
    if 'sort' in n.lower() or 'radix' in n.lower(): return 'sort'
    return 'other'
acc = {}
for r in rows:
    d = acc.setdefault(cat(r['Name']), [0, 0]); d[0] += int(r['TotalDurationNs']); d[1] += int(r['Calls'])
tot = sum(v[0] for v in acc.values())
print('total %.3f ms/step, %.0f launches/step' % (tot / 1e6 / steps, sum(v[1] for v in acc.values()) / steps))
for k, (t, n) in sorted(acc.items(), key=lambda x: -x[1][0]):
    print('%-34s %7.3f ms/step %7.1f launches/step  avg %7.1f us' % (k, t / 1e6 / steps, n / steps, t / 1e3 / n))
for r in rows[:top]:
    print('%6s calls avg %8.1f us  %5.2f%%  %s' % (r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage']), r['Name'][:150]))
