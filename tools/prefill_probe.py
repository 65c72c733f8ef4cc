import sys, os, json, torch, numpy as np
sys.path.insert(0, os.getcwd())
from hqtransformer_amd import synth
from hqtransformer_amd.config import load_config
from hqtransformer_amd.models import ImageGPT2
from hqtransformer_amd.sampling import sampling_ihqgpt
cfg = load_config('configs/cc15m-12l-txt.yaml')
m = ImageGPT2(cfg, seed=0).to('cuda').eval()
s2 = m.stage2.spec
B = 64
txt = torch.from_numpy(synth.text_ids(1, B, s2.ctx_len_txt, s2.vocab_txt)).cuda()
def run(n, graph=True):
    return sampling_ihqgpt(m.stage2, num_candidates=B, cond=txt, top_k_top=None, top_p_top=None, top_k_bot=None, top_p_bot=None,
                           softmax_temperature=[1.0, 1.0], use_fp16=True, is_tqdm=False, max_seq_len=n, seed=3, use_graph=graph)
for n in (1, 2, 64):
    run(n); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(n); b.record(); torch.cuda.synchronize()
    print('n_steps', n, 'ms', round(a.elapsed_time(b), 3))
e = m.stage2.engine(B, 64)
e.timing(True); e.timing_reset()
run(1, graph=False); torch.cuda.synchronize()
print(json.dumps({k: [n, round(t, 3)] for k, (n, t) in sorted(e.timing_report().items(), key=lambda kv: -kv[1][1])}))
