#!/usr/bin/env python3
"""Is the SPLIT decode power-limited in a way that leaves room beside it?  Decode alone and the AR pass alone on streams restricted to a
fraction of the CUs (hipExtStreamCreateWithCUMask), then both side by side on disjoint CU sets against the plain two-lane order.
    python tools/cu_mask_probe.py [--rows 640]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hqtransformer_amd.config import load_config  # noqa: E402
from hqtransformer_amd.models import ImageGPT2  # noqa: E402
from hqtransformer_amd.sampling import sampling_ihqgpt  # noqa: E402


def hip():
    torch.cuda.init()
    for line in open('/proc/self/maps'):
        if 'libamdhip64' in line:
            return C.CDLL(line.split()[-1])
    raise RuntimeError('libamdhip64 not mapped')


def masked_stream(lib, keep):
    """keep(i) -> bool for CU i of 256"""
    words = (C.c_uint32 * 8)()
    n = 0
    for i in range(256):
        if keep(i):
            words[i // 32] |= 1 << (i % 32)
            n += 1
    st = C.c_void_p()
    rc = lib.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    if rc != 0:
        raise RuntimeError(f'hipExtStreamCreateWithCUMask -> {rc}')
    return torch.cuda.ExternalStream(st.value), n


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--rows', type=int, default=640)
    p.add_argument('--config', default='configs/imagenet-12l.yaml')
    a = p.parse_args()
    dev = torch.device('cuda:0')
    lib = hip()
    model = ImageGPT2(load_config(a.config), seed=0).to(dev).eval()
    R = a.rows
    cond = torch.arange(R) % 1000
    out = {'rows': R}

    def ar(lane, graph=True):
        return sampling_ihqgpt(model.stage2, num_candidates=R, cond=cond, use_fp16=True, is_tqdm=False, max_seq_len=64, seed=1, lane=lane, use_graph=graph)
    ct, cb = ar(0)
    torch.cuda.synchronize()

    def dec(lane):
        return model.stage1.decode_sequences(ct, cb, precision='split', clamp01=True, lane=lane)

    def timed(fn, st, n=2):
        with torch.cuda.stream(st):
            fn()
        st.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(st):
            for _ in range(n):
                fn()
        st.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    pats = {'256': lambda i: True, '224': lambda i: i % 8 != 7, '192': lambda i: i % 4 != 3, '160': lambda i: i % 8 not in (3, 6, 7), '128': lambda i: i % 2 == 0,
            '64': lambda i: i % 4 == 3}
    for name, k in pats.items():
        st, n = masked_stream(lib, k)
        out[f'decode_ms_cus{name}'] = round(timed(lambda: dec(0), st), 1)
        out[f'ar_graph_ms_cus{name}'] = round(timed(lambda: ar(0, True), st), 1)
        out[f'ar_eager_ms_cus{name}'] = round(timed(lambda: ar(0, False), st), 1)
        print(name, n, out[f'decode_ms_cus{name}'], out[f'ar_graph_ms_cus{name}'], out[f'ar_eager_ms_cus{name}'], flush=True)
    # side by side: lane 0 decodes on the big set while lane 1 samples on the complement; against the same two on plain streams
    for big, small, kb in (('192', '64', lambda i: i % 4 != 3), ('160', '96', lambda i: i % 8 not in (3, 6, 7)), ('224', '32', lambda i: i % 8 != 7)):
        sb, _ = masked_stream(lib, kb)
        ss, _ = masked_stream(lib, lambda i, kb=kb: not kb(i))
        for graph in (True, False):
            def both():
                with torch.cuda.stream(sb):
                    for _ in range(2):
                        dec(0)
                with torch.cuda.stream(ss):
                    for _ in range(2):
                        ar(1, graph)
                sb.synchronize(); ss.synchronize()
            both()
            t0 = time.perf_counter()
            both()
            out[f'side_by_side_decode{big}_ar{small}_graph{int(graph)}_ms_per_pair'] = round((time.perf_counter() - t0) / 2 * 1e3, 1)
    s0, s1 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def plain():
        with torch.cuda.stream(s0):
            for _ in range(2):
                dec(0)
        with torch.cuda.stream(s1):
            for _ in range(2):
                ar(1, True)
        s0.synchronize(); s1.synchronize()
    plain()
    t0 = time.perf_counter()
    plain()
    out['plain_two_streams_ms_per_pair'] = round((time.perf_counter() - t0) / 2 * 1e3, 1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
