"""print the headline numbers of a bench.py JSON line (file argument)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['unit'], d['ms_per_step'], 'ms')
for k in ('kernel_ms', 'kernels_ms', 'stage_ms'):
    if k in d:
        print({a: round(b, 3) for a, b in d[k].items()})
print(d.get('roofline'))
