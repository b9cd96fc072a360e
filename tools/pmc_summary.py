"""Summarise rocprofv3 --pmc CSVs (gpurun_out/pmc/*): per kernel name, mean counter value per dispatch."""
import csv, glob, os, sys, json
from collections import defaultdict
root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc'
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, '*', '*', '*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, d in sorted(acc.items()):
    out[k] = {c: sum(v) / len(v) for c, v in d.items()}
    out[k]['dispatches'] = max(len(v) for v in d.values())
for k, d in out.items():
    if d['dispatches'] < 3 or 'rocclr' in k: continue
    print(k, d['dispatches'])
    for c, v in sorted(d.items()):
        if c != 'dispatches': print('    %-28s %.4g' % (c, v))
json.dump(out, open(os.path.join(root, 'summary.json'), 'w'), indent=1)
