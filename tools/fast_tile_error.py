"""FAST teacher-forced logit error of a 512-row pass at the ImageNet width against the oracle, tile GEMMs vs streaming GEMMs
(HQT_NO_TILE_GEMM=1), per draw: is the LDS-tiled path as accurate as the kernels it replaces?"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from hqtransformer_amd import synth
    from hqtransformer_amd._lib import POLICY_THROUGHPUT, PRECISION_FAST
    from hqtransformer_amd.engine import Engine
    from hqtransformer_amd.spec import Stage2Spec
    from oracle import hqt_oracle as O
    spec = Stage2Spec(embed_dim=1536, n_layers=1, n_heads=24, n_layers_depth=1, vocab_top=8192, vocab_bot=8192, vocab_txt=64,
                      ctx_len_img=64, ctx_len_txt=16, n_classes=1000, cond=1, embedding=0)
    weights = synth.stage2_weights(spec, 31, 'fixture')
    B, n = 512, 3
    noise = synth.exp_noise(32, n, B, spec.vocab_top)
    cond = (np.arange(B) * 7) % spec.n_classes
    cache = '/tmp/fast_tile_error_want.npz'
    if os.path.exists(cache):
        z = np.load(cache)
        want = (z['a'], z['b'], z['c'])
    else:
        want = O.OracleStage2(spec, weights).sample(cond, B, n, noise, return_logits=True)
        np.savez(cache, a=want[0], b=want[1], c=want[2])
    eng = Engine(spec, None, torch.device('cuda:0'), B, spec.ctx_len_img)
    eng.load(stage2=weights)
    eng.finalize()
    eng.set_policy(POLICY_THROUGHPUT)
    _, _, lf = eng.sample(B, torch.from_numpy(cond), n, precision=PRECISION_FAST, noise=torch.from_numpy(noise), force_top=torch.from_numpy(want[0]),
                          force_bot=torch.from_numpy(want[1]), return_logits=True, use_graph=False)
    err = np.abs(lf.cpu().numpy() - want[2])
    print('HQT_NO_TILE_GEMM =', os.environ.get('HQT_NO_TILE_GEMM'), 'logit std', float(want[2].std()), 'max err', float(err.max()), 'mean err', float(err.mean()))
    print('  per (step, draw) max:', np.round(err.max(axis=(2, 3)), 4).tolist())


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        main()
    else:
        for v in ('0', '1'):
            env = dict(os.environ)
            if v == '1':
                env['HQT_NO_TILE_GEMM'] = '1'
            subprocess.run([sys.executable, __file__, 'child'], env=env, check=False)
