#!/usr/bin/env python3
"""Kernel-trace digest: per-kernel duration and the idle gap to the previous kernel on the same queue.
usage: kt_gaps.py <..._kernel_trace.csv> [skip_first_n]"""
import csv
import sys
from collections import defaultdict

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
byq = defaultdict(list)
for r in rows:
    byq[r['Queue_Id']].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-60:]))
for q, v in sorted(byq.items()):
    v.sort()
    v = v[skip:]
    if len(v) < 20:
        continue
    dur, gap = defaultdict(list), defaultdict(list)
    for i in range(1, len(v)):
        dur[v[i][2]].append(v[i][1] - v[i][0])
        gap[v[i][2]].append(v[i][0] - v[i - 1][1])
    span = v[-1][1] - v[0][0]
    busy = sum(e - s for s, e, _ in v)
    print('queue %s: %d kernels, span %.3f ms, busy %.3f ms (%.1f %%)' % (q, len(v), span / 1e6, busy / 1e6, 100.0 * busy / span))
    for k in sorted(dur, key=lambda k: -sum(dur[k])):
        print('  %-62s n=%5d  dur p50 %7.1f us  mean %7.1f  | gap before p50 %6.1f us mean %6.1f' % (
            k, len(dur[k]), np.median(dur[k]) / 1e3, np.mean(dur[k]) / 1e3, np.median(gap[k]) / 1e3, np.mean(gap[k]) / 1e3))

# overlap across queues: how many kernels are in flight, by share of the traced span
ev = []
for q, v in byq.items():
    for s, e, _ in sorted(v)[skip:]:
        ev.append((s, 1)); ev.append((e, -1))
ev.sort()
if ev:
    depth, last, acc = 0, ev[0][0], defaultdict(int)
    for t, d in ev:
        acc[depth] += t - last
        last = t
        depth += d
    tot = sum(acc.values())
    print('kernels in flight over the span: ' + ', '.join('%d: %.1f %%' % (k, 100.0 * acc[k] / tot) for k in sorted(acc)))
