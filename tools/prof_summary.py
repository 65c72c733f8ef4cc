#!/usr/bin/env python3
"""Print the per-kernel summary of a rocprofv3 --kernel-trace --stats run (csv output directory)."""
import csv
import glob
import sys

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
f = sorted(glob.glob(f'{d}/**/*kernel_stats.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print(f'{f}: total kernel time {tot / 1e6:.2f} ms')
print(f'{"kernel":90s} {"calls":>7s} {"total_ms":>10s} {"avg_us":>9s} {"%":>6s}')
for r in rows[:top]:
    print(f'{r["Name"][:90]:90s} {r["Calls"]:>7s} {int(r["TotalDurationNs"]) / 1e6:10.3f} {float(r["AverageNs"]) / 1e3:9.2f} {float(r["Percentage"]):6.2f}')
