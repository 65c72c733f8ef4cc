#!/usr/bin/env python3
"""Audit of the hand-scheduled loop of conv3x3_split_ring_kernel in the compiler's output.

The kernel keeps asm-issued loads in flight across its loop's back edge; that is only safe if (a) the loop body is ONE basic block and
(b) hipcc never touches a destination register of such a load except in the MFMAs / ds_writes that consume it after the counted wait.
This script checks both on the generated ISA:   hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only split_stream_conv.hip -o x.s;
                                                 python hqtransformer_amd/csrc/audit_ring.py x.s [mangled kernel name]
(both hand-scheduled kernels: _Z25conv3x3_split_ring_kernelILi0EEv8GemmArgs, _Z27conv3x3_split_ring16_kernelILi0EEv8GemmArgs)
"""
import re
import sys

src = open(sys.argv[1]).read().split('\n')
name = (sys.argv[2] if len(sys.argv) > 2 else '_Z25conv3x3_split_ring_kernelILi0EEv8GemmArgs') + ':'        # mangled kernel name
i0 = next(i for i, l in enumerate(src) if l.startswith(name))
i1 = next(i for i in range(i0, len(src)) if 's_endpgm' in src[i])
body = src[i0:i1]
hdr = [i for i, l in enumerate(body) if 'Loop Header' in l]
assert hdr, 'no loop found'            # (the first loop is the main loop; the epilogue's shuffle reduction may be a second)
lo = hdr[0]
hi = next(i for i in range(lo, len(body)) if re.search(r's_cbranch_\w+\s+' + re.escape(body[lo].split(':')[0]), body[i]))
loop = body[lo + 1:hi]
labels = [l for l in loop if re.match(r'^\.LBB', l)]
branches = [l for l in loop if re.search(r's_cbranch|s_branch', l)]
print(f'loop: {len(loop)} lines, {sum("v_mfma" in l for l in loop)} MFMAs, labels inside: {len(labels)}, branches inside: {len(branches)}')


def regs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


# flow-sensitive check: walk the prologue once and the loop body three times; a register is "in flight" from the load that writes it
# until a vmcnt wait retires that load (vmcnt retires in issue order); nothing may read or write it in between
start = next(i for i in range(len(body)) if body[i].strip().startswith('buffer_load_dwordx4'))       # the prologue's first patch piece
seq = body[start:lo] + loop * 3
inflight = []            # [(regs)] oldest first
bad = []
for l in seq:
    t = l.strip()
    if not t or t.startswith(';'):
        continue
    m = re.match(r's_waitcnt\s+(.*)', t)
    if m:
        v = re.search(r'vmcnt\((\d+)\)', m.group(1))
        if v:
            n = int(v.group(1))
            inflight = inflight[len(inflight) - n:] if n else []
        continue
    toks = re.findall(r'v\[\d+:\d+\]|v\d+', t)
    used = set().union(*[regs(x) for x in toks]) if toks else set()
    busy = set().union(*inflight) if inflight else set()
    if used & busy:
        bad.append(t)
    if t.startswith(('global_load_dwordx4', 'buffer_load_dwordx4')) and ' lds' not in t:
        inflight.append(regs(t.split()[1].rstrip(',')))
    elif t.startswith(('global_load', 'buffer_load', 'scratch_load')):
        inflight.append(set())      # LDS-DMA piece / other: counts in vmcnt, writes no tracked register
print('instructions touching a register with a load in flight:', len(bad))
for b in bad[:20]:
    print('   ', b)
scratch = [l for l in loop if 'scratch_' in l]
print('scratch accesses in the loop:', len(scratch))
ok = not labels and not branches and not bad and not scratch
print('AUDIT', 'OK' if ok else 'FAILED')
sys.exit(0 if ok else 1)
