#!/usr/bin/env python3
"""How busy is the GPU in the timed region?  From a `rocprofv3 --kernel-trace --output-format csv` run of bench.py: the union of all
kernel intervals, the share of wall time with >= 1 / >= 2 / >= 3 kernels in flight, over the densest `--window` seconds of the run.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --no-cpu-baseline --no-roofline
    python tools/trace_coverage.py gpurun_out/trace [--window 3.0]"""
import csv
import glob
import sys

d = sys.argv[1]
window = float(sys.argv[sys.argv.index('--window') + 1]) if '--window' in sys.argv else 3.0
f = sorted(glob.glob(f'{d}/**/*kernel_trace.csv', recursive=True))[-1]
ev = []
names = {}
for r in csv.DictReader(open(f)):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    ev.append((s, 1, r['Kernel_Name']))
    ev.append((e, -1, r['Kernel_Name']))
ev.sort()
t_end = ev[-1][0]
# the timed region of bench.py is followed by the serial pass: take the window that ends 35 % before the end of the trace ... simpler: densest window
lo = t_end - int(12e9)
pts = [(t, k) for t, k, _ in ev if t >= lo]
best = None
import bisect
ts = [t for t, _ in pts]
# sliding windows every 0.25 s
w = int(window * 1e9)
t0 = ts[0]
while t0 + w <= ts[-1]:
    i, j = bisect.bisect_left(ts, t0), bisect.bisect_right(ts, t0 + w)
    n = j - i
    if best is None or n > best[0]:
        best = (n, t0)
    t0 += int(0.25e9)
n, t0 = best
depth = 0
for t, k, _ in ev:
    if t >= t0:
        break
    depth += k
busy = [0, 0, 0, 0]
last = t0
for t, k, _ in ev:
    if t < t0:
        continue
    if t > t0 + w:
        break
    dt = t - last
    for lvl in range(4):
        if depth > lvl:
            busy[lvl] += dt
    last = t
    depth += k
print(f'{f}: window of {window} s with {n // 2} kernels')
print('share of wall time with >= 1 / 2 / 3 / 4 kernels in flight: ' + ' / '.join(f'{100 * b / w:.1f} %' for b in busy))
