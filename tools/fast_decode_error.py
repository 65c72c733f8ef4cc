import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
from hqtransformer_amd import synth
from hqtransformer_amd._lib import PRECISION_EXACT, PRECISION_FAST
from hqtransformer_amd.config import load_config
from hqtransformer_amd.engine import Engine
from hqtransformer_amd.spec import stage1_spec_from_config
s1 = stage1_spec_from_config(load_config('configs/imagenet-12l.yaml'))
for prof in ('bench', 'fixture'):
    e = Engine(None, s1, torch.device('cuda:0'), 8)
    e.load(stage1=synth.stage1_weights(s1, 1, prof)); e.finalize()
    r = np.random.default_rng(0)
    ct = torch.from_numpy(r.integers(0, s1.n_embed, (8, 8, 8))); cb = torch.from_numpy(r.integers(0, s1.n_embed, (8, 16, 16)))
    ex = e.decode(ct, cb, precision=PRECISION_EXACT, clamp01=False).double()
    fa = e.decode(ct, cb, precision=PRECISION_FAST, clamp01=False).double()
    d = (ex - fa).abs()
    rng = float(ex.max() - ex.min())
    mse = float((d ** 2).mean())
    print(prof, 'range', round(rng, 3), 'std', round(float(ex.std()), 3), 'max', round(float(d.max()), 4), 'mean', round(float(d.mean()), 5),
          'rel rms', round((mse ** 0.5) / float(ex.std()), 5), 'PSNR(range) dB', round(10 * np.log10(rng * rng / mse), 1))
    e.close()
