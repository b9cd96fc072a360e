"""Analyse a rocprofv3 kernel trace (kernel_trace.csv): per hardware queue, the time covered by kernels against the span, and the idle
gaps between consecutive kernels of the queue (in-stream dependency / dispatch gaps).   usage: python tools/trace_gaps.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
cols = rows[0].keys()
qk = 'Queue_Id' if 'Queue_Id' in cols else [c for c in cols if 'queue' in c.lower()][0]
sk, ek = 'Start_Timestamp', 'End_Timestamp'
nk = 'Kernel_Name'
per = collections.defaultdict(list)
for r in rows:
    per[r[qk]].append((int(r[sk]), int(r[ek]), r[nk]))
# steady-state window: the last 40 % of the trace
t_all = sorted(x for v in per.values() for x in v)
t0 = t_all[int(len(t_all) * 0.6)][0]
tot_busy = 0
for q, v in sorted(per.items(), key=lambda kv: -len(kv[1])):
    v = sorted(x for x in v if x[0] >= t0)
    if len(v) < 50:
        continue
    span = v[-1][1] - v[0][0]
    busy = sum(e - s for s, e, _ in v)
    gaps = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
    pos = [g for g in gaps if g > 0]
    print('queue %s: %d kernels, span %.2f ms, kernels cover %.1f %%, mean gap %.2f us, median %.2f us, p90 %.2f us, overlapping successors %d' %
          (q, len(v), span / 1e6, 100.0 * busy / span, sum(pos) / max(1, len(pos)) / 1e3, sorted(pos)[len(pos) // 2] / 1e3 if pos else 0,
           sorted(pos)[int(len(pos) * 0.9)] / 1e3 if pos else 0, sum(1 for g in gaps if g <= 0)))
    by = collections.defaultdict(list)
    for i in range(len(v) - 1):
        by[v[i][2][:40] + ' -> ' + v[i + 1][2][:40]].append(v[i + 1][0] - v[i][1])
    for k, g in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:6]:
        print('     %-85s n=%4d mean gap %.2f us' % (k, len(g), sum(g) / len(g) / 1e3))
