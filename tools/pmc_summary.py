"""Summarise rocprofv3 --pmc CSVs (gpurun_out/pmc/*): per kernel name, mean counter value per dispatch.  The RESID GEMM kernel serves two
layers with the same grid (proj and fc2 alternate in dispatch order: the even / odd dispatches of a forward are also listed separately
as '<kernel> #even' / '<kernel> #odd').   usage: python tools/pmc_summary.py <dir>"""
import csv, glob, os, sys, json
from collections import defaultdict
root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc'
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, '*', '*', '*counter_collection.csv')):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Dispatch_Id']))
    seen = defaultdict(lambda: defaultdict(int))
    for r in rows:
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        if k.startswith('k_gemm_dma<2,') or k.startswith('k_gemm_dma<6,'):
            i = seen[k][r['Counter_Name']]
            seen[k][r['Counter_Name']] += 1
            acc[k + (' #even' if i % 2 == 0 else ' #odd')][r['Counter_Name']].append(float(r['Counter_Value']))
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
