#!/usr/bin/env python3
"""MFMA utilisation per kernel from a tools/pmc_summary.py output of a
`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE` pass.

SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of every SIMD's matrix pipe (32 per v_mfma_f32_32x32x16_bf16: checked against
the conv_out launch, 9.44 M MFMAs -> 302 M); SQ_BUSY_CU_CYCLES sums the cycles each CU had work.  MfmaUtil = MFMA_BUSY /
(4 SIMDs x BUSY_CU): the share of the busy CUs' matrix-pipe cycles that issued MFMA work (GRBM_GUI_ACTIVE on this part is summed
over the 8 XCDs, so the gfx94x formula MFMA_BUSY / (GUI_ACTIVE x CUs x 4) must be multiplied by 8; both are printed).
usage: pmc_mfma.py <summary.txt> [out.txt]
"""
import collections
import re
import sys

rows = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    m = re.match(r'^(.*?)\s+(SQ_VALU_MFMA_BUSY_CYCLES|SQ_BUSY_CU_CYCLES|GRBM_GUI_ACTIVE)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$', line)
    if m:
        rows[m.group(1).strip()][m.group(2)] = (int(m.group(3)), float(m.group(4)))
out = []
for k, v in rows.items():
    if v.get('SQ_VALU_MFMA_BUSY_CYCLES', (0, 0.0))[1] > 0:
        n, mf = v['SQ_VALU_MFMA_BUSY_CYCLES']
        gui, bc = v['GRBM_GUI_ACTIVE'][1], v['SQ_BUSY_CU_CYCLES'][1]
        out.append((mf / (4 * bc), 8 * mf / (gui * 256 * 4), n, mf, bc, gui, k))
lines = [f'{"MfmaUtil":>9s} {"(gfx94x x8)":>11s} {"dispatches":>10s} {"MFMA_BUSY/disp":>15s} {"BUSY_CU/disp":>14s} {"GUI_ACTIVE/disp":>15s}  kernel']
for u, u2, n, mf, bc, gui, k in sorted(out, reverse=True):
    lines.append(f'{100 * u:8.1f}% {100 * u2:10.1f}% {n:10d} {mf:15.0f} {bc:14.0f} {gui:15.0f}  {k[:110]}')
text = '\n'.join(lines)
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(__doc__.split('usage:')[0].strip() + '\n\n' + text + '\n')
