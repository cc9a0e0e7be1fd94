#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace result (rocpd .db or *_kernel_trace.csv) into a per-kernel stats table:
calls, total, average, min, max duration and share of GPU time.  Usage: rocprof_summary.py <results.db|csv> [out.csv]"""
import csv
import sqlite3
import sys


def from_db(path):
    c = sqlite3.connect(path).cursor()
    return c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), grid_x "
                     "from kernels group by name order by 3 desc").fetchall()


def from_csv(path):
    agg = {}
    for r in csv.DictReader(open(path)):
        d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        a = agg.setdefault(r['Kernel_Name'], [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    return sorted([(k, v[0], v[1], v[1] / v[0], v[2], v[3], 0) for k, v in agg.items()], key=lambda r: -r[2])


def main():
    path = sys.argv[1]
    rows = from_db(path) if path.endswith('.db') else from_csv(path)
    tot = sum(r[2] for r in rows)
    out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
    w = csv.writer(out)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'MinNs', 'MaxNs', 'Percentage'])
    for r in rows:
        w.writerow([r[0], r[1], int(r[2]), '%.1f' % r[3], r[4], r[5], '%.2f' % (100.0 * r[2] / tot)])


if __name__ == '__main__':
    main()
