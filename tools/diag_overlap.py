#!/usr/bin/env python3
"""Where does the time of the default bench go?  AR passes alone, decodes alone and both, on 1 .. 3 lanes (rows = merge x batch).

    python tools/diag_overlap.py [--rows 256] [--lanes 3] [--passes 9]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hqtransformer_amd.config import load_config  # noqa: E402
from hqtransformer_amd.models import ImageGPT2  # noqa: E402
from hqtransformer_amd.pipeline import InflightSampler  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--config', default='configs/imagenet-12l.yaml')
    p.add_argument('--rows', type=int, default=256)
    p.add_argument('--lanes', type=int, default=3)
    p.add_argument('--passes', type=int, default=9)
    p.add_argument('--decode-precision', default='split')
    p.add_argument('--ar-priority', action='store_true', help='AR loops on streams of the highest priority')
    a = p.parse_args()
    dev = torch.device('cuda:0')
    model = ImageGPT2(load_config(a.config), seed=0).to(dev).eval()
    R = a.rows
    cond = torch.arange(R) % 1000
    out = {}
    for lanes in sorted({1, a.lanes}):
        pipe = InflightSampler(model, lanes=lanes, device=dev, ar_high_priority=a.ar_priority)

        def run(kind, n):
            t0 = None
            res = None
            for i in range(n + lanes):
                if i == lanes:
                    pipe.drain(); torch.cuda.synchronize(); t0 = time.perf_counter()
                if kind == 'decode':
                    lane = pipe.k % pipe.n
                    pipe.k += 1
                    with torch.cuda.stream(pipe.streams[lane]):
                        model.stage1.decode_sequences(codes[0], codes[1], precision=a.decode_precision, clamp01=True, lane=lane)
                else:
                    res = pipe._launch(R, cond, seed=i, max_seq_len=64, use_fp16=True, decode=(kind == 'both'), precision=a.decode_precision,
                                       top_k_top=2048, top_k_bot=2048)
            pipe.drain(); torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3, res
        ms, res = run('ar', a.passes)
        codes = (res[0], res[1])
        out[f'ar_only_lanes{lanes}_ms_per_pass'] = round(ms, 2)
        out[f'decode_only_lanes{lanes}_ms_per_pass'] = round(run('decode', a.passes)[0], 2)
        out[f'both_lanes{lanes}_ms_per_pass'] = round(run('both', a.passes)[0], 2)
    out['rows'] = R
    out['ar_priority'] = bool(a.ar_priority)
    out['images_per_s_both'] = round(R / out[f'both_lanes{a.lanes}_ms_per_pass'] * 1e3, 1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
