#!/usr/bin/env python3
"""One batch-64 step at a time, its AR loop cut into sub-batches that run side by side on lanes.

The reference harness (measure_throughput/__main__.py:84-116) runs ONE batch-64 step at a time.  A 64-row AR loop is a dependent chain
of ~104 launches per position, each latency-bound on a fraction of the chip; the rows of a step are independent chains, so the step can
be cut into P sub-batches (same seed, sample_offset = first row: every row draws what it draws in the uncut call) that run on P lanes
at once.  This probe measures the AR time of one step for P = 1 .. 4 under both kernel policies, and the whole step (AR + one 64-image decode).

    python tools/split_step_probe.py [--batch 64] [--steps 6]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hqtransformer_amd._lib import POLICY_LATENCY, POLICY_THROUGHPUT  # noqa: E402
from hqtransformer_amd.config import load_config  # noqa: E402
from hqtransformer_amd.models import ImageGPT2  # noqa: E402
from hqtransformer_amd.sampling import sampling_ihqgpt  # noqa: E402


def parts_of(B, P):
    q, r = divmod(B, P)
    sizes = [q + (1 if i < r else 0) for i in range(P)]
    offs = [sum(sizes[:i]) for i in range(P)]
    return sizes, offs


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--config', default='configs/imagenet-12l.yaml')
    p.add_argument('--batch', type=int, default=64)
    p.add_argument('--steps', type=int, default=6)
    p.add_argument('--max-parts', type=int, default=4)
    a = p.parse_args()
    dev = torch.device('cuda:0')
    model = ImageGPT2(load_config(a.config), seed=0).to(dev).eval()
    B = a.batch
    cond_all = (torch.arange(B) * 7) % 1000
    out = {'batch': B}
    streams = [torch.cuda.Stream(device=dev) for _ in range(a.max_parts)]
    ref_codes = None
    for P in range(1, a.max_parts + 1):
        sizes, offs = parts_of(B, P)
        for pol_name, pol in (('latency', POLICY_LATENCY), ('throughput', POLICY_THROUGHPUT)):
            for lane, n in enumerate(sizes):
                eng = model.stage2.engine(n, 64, lane)
                if eng.policy != pol:
                    eng.set_policy(pol)

            def one_step(seed, decode):
                cur = torch.cuda.current_stream(dev)
                res = []
                for lane, (n, o) in enumerate(zip(sizes, offs)):
                    st = streams[lane]
                    st.wait_stream(cur)
                    with torch.cuda.stream(st):
                        res.append(sampling_ihqgpt(model.stage2, num_candidates=n, cond=cond_all[o:o + n], use_fp16=True, is_tqdm=False,
                                                   max_seq_len=64, seed=seed, sample_offset=o, lane=lane))
                for lane in range(P):
                    cur.wait_stream(streams[lane])
                ct = torch.cat([r[0] for r in res]) if P > 1 else res[0][0]
                cb = torch.cat([r[1] for r in res]) if P > 1 else res[0][1]
                px = model.stage1.decode_sequences(ct, cb, precision='split', clamp01=True) if decode else None
                return ct, cb, px
            for decode in (False, True):
                for w in range(2):
                    one_step(100 + w, decode)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for k in range(a.steps):
                    ct, cb, px = one_step(1 + k, decode)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / a.steps * 1e3
                out[f'parts{P}_{pol_name}_{"step" if decode else "ar"}_ms'] = round(ms, 2)
            ct, cb, _ = one_step(4242, False)
            torch.cuda.synchronize()
            if ref_codes is None:
                ref_codes = (ct.clone(), cb.clone())
            else:
                out[f'parts{P}_{pol_name}_same_draws_as_uncut'] = round(float(((ct == ref_codes[0]).float().mean() + (cb == ref_codes[1]).float().mean()) / 2), 4)
        print(json.dumps({k: v for k, v in out.items() if k.startswith(f'parts{P}_')}), flush=True)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
