"""AR-loop time of one merged pass (rows = merge x 64) of the ImageNet-12L model, graphed, one lane -- and, un-graphed with the
per-launch timers, where it goes.  For A/B runs of the GEMM planner (HQT_TILE_FORCE, HQT_NO_TILE_GEMM).
    python tools/ar_pass_time.py --rows 512 1280 [--policy 1] [--breakdown]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from hqtransformer_amd.config import load_config
    from hqtransformer_amd.models import ImageGPT2
    from hqtransformer_amd.sampling import sampling_ihqgpt
    p = argparse.ArgumentParser()
    p.add_argument('--rows', type=int, nargs='+', default=[512, 1280])
    p.add_argument('--policy', type=int, default=0)
    p.add_argument('--breakdown', action='store_true')
    p.add_argument('--by-rows', action='store_true', help='one timing slot per (GEMM, row count)')
    p.add_argument('--precision', default='fast', choices=['fast', 'exact', 'split'])
    p.add_argument('--reps', type=int, default=3)
    p.add_argument('--config', default=os.path.join(ROOT, 'configs', 'imagenet-12l.yaml'))
    a = p.parse_args()
    if a.by_rows:
        os.environ['HQT_TIMING_BY_ROWS'] = '1'
    m = ImageGPT2(load_config(a.config), seed=0).to('cuda').eval()
    out = {'env': {k: v for k, v in os.environ.items() if k.startswith('HQT_')}, 'precision': a.precision}
    for rows in a.rows:
        eng = m.stage2.engine(rows, 64)
        eng.set_policy(a.policy)
        cond = torch.randint(0, 1000, (rows,))

        def run(graph=True):
            return sampling_ihqgpt(m.stage2, num_candidates=rows, cond=cond, use_fp16=True, precision=a.precision, is_tqdm=False, max_seq_len=64, seed=1, use_graph=graph)
        run(); run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        rec = {'ar_ms_per_pass': round(ms, 2), 'ar_ms_per_64_images': round(ms * 64 / rows, 3)}
        if a.breakdown:
            eng.timing(True); eng.timing_reset()
            run(False)
            torch.cuda.synchronize()
            rep = eng.timing_report()
            eng.timing(False)
            rec['eager_ms'] = {k: round(v[1], 2) for k, v in sorted(rep.items(), key=lambda kv: -kv[1][1]) if not k.startswith('variant:')}
            rec['variants'] = {k[8:]: v[0] for k, v in rep.items() if k.startswith('variant:')}
        out[f'rows={rows}'] = rec
    print(json.dumps(out))


if __name__ == '__main__':
    main()
