"""Analyse a rocprofv3 kernel trace of a GPU-bound run (tools/overlap_profile.sh): steady-state window = the last 60 % of the trace.
Per hardware queue: time covered by kernels / span; over all queues: union coverage, mean number of kernels in flight; per kernel
name: launches, average duration in the window (i.e. UNDER OVERLAP with the other queues' kernels), share of the summed durations.
usage: python tools/trace_overlap.py <kernel_trace.csv>"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name']) for r in rows)
t_lo = ev[int(len(ev) * 0.4)][0]
ev = [e for e in ev if e[0] >= t_lo]
t_hi = max(e[1] for e in ev)
span = t_hi - t_lo
perq = collections.defaultdict(list)
for s, e, q, n in ev:
    perq[q].append((s, e))
print('steady-state window %.2f ms, %d kernels on %d queues' % (span / 1e6, len(ev), len(perq)))
for q, v in sorted(perq.items(), key=lambda kv: -len(kv[1])):
    v.sort()
    busy, cur_s, cur_e = 0, v[0][0], v[0][1]
    for s, e in v[1:]:
        if s > cur_e:
            busy += cur_e - cur_s; cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    qspan = v[-1][1] - v[0][0]
    print('  queue %s: %5d kernels, covered %.1f %% of its span (%.2f ms), sum of durations %.2f ms' % (q, len(v), 100.0 * busy / max(1, qspan), qspan / 1e6, sum(e - s for s, e in v) / 1e6))
pts = sorted([(s, 1) for s, e, _, _ in ev] + [(e, -1) for s, e, _, _ in ev])
depth, last, hist = 0, t_lo, collections.Counter()
for t, d in pts:
    hist[depth] += t - last
    last, depth = t, depth + d
tot = sum(hist.values())
print('kernels in flight: ' + ', '.join('%d: %.1f %%' % (k, 100.0 * v / tot) for k, v in sorted(hist.items())) + '  (mean %.2f)' % (sum(k * v for k, v in hist.items()) / tot))
by = collections.defaultdict(list)
for s, e, q, n in ev:
    n = re.sub(r'^void ', '', n); n = re.sub(r'\(.*$', '', n)
    by[n].append(e - s)
allsum = sum(sum(v) for v in by.values())
print('%-44s %7s %10s %8s' % ('kernel', 'count', 'avg us', 'share'))
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print('%-44s %7d %10.2f %7.1f %%' % (n[:44], len(v), sum(v) / len(v) / 1e3, 100.0 * sum(v) / allsum))
