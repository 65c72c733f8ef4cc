#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc run (csv) per kernel: mean counter value per dispatch."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
files = sorted(glob.glob(f'{d}/**/*counter_collection.csv', recursive=True))
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in files:
    for r in csv.DictReader(open(f)):
        name = r.get('Kernel_Name', r.get('Name', '?'))
        e = acc[name][r['Counter_Name']]
        e[0] += 1
        e[1] += float(r['Counter_Value'])
print(f'{"kernel":80s} {"counter":14s} {"dispatches":>10s} {"mean/dispatch":>16s} {"sum":>16s}')
for name, cs in sorted(acc.items(), key=lambda kv: -max(v[1] for v in kv[1].values())):
    for c, (n, s) in cs.items():
        print(f'{name[:80]:80s} {c:14s} {n:10d} {s / n:16.1f} {s:16.1f}')
