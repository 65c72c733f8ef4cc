"""Diagnostic: two root engines sampling concurrently on two streams (persistent launches ordered by the library), against each engine alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hqtransformer_amd import synth
from hqtransformer_amd._lib import PRECISION_FAST
from hqtransformer_amd.engine import Engine
from hqtransformer_amd.config import load_config
from hqtransformer_amd.spec import stage2_spec_from_config

s2 = stage2_spec_from_config(load_config(os.path.join(ROOT, 'configs', 'imagenet-12l.yaml')))
w = synth.stage2_weights(s2, 0, 'bench')
dev = torch.device('cuda:0')


def mk():
    e = Engine(s2, None, dev, 64, 64)
    e.load(stage2=w)
    e.finalize()
    return e


a, b = mk(), mk()
B, n = 64, 8
cond = torch.from_numpy(synth.class_ids(5, B, s2.n_classes))


def run(tag, persist, graph, other, side=True):
    a.set_persist(persist); b.set_persist(persist)
    al_a = a.sample(B, cond, n, precision=PRECISION_FAST, seed=3, use_graph=graph); torch.cuda.synchronize()
    al_a2 = a.sample(B, cond, n, precision=PRECISION_FAST, seed=3, use_graph=graph); torch.cuda.synchronize()
    al_b = b.sample(B, cond, n, precision=PRECISION_FAST, seed=4, use_graph=graph); torch.cuda.synchronize()
    sa, sb = (torch.cuda.Stream() if side else torch.cuda.current_stream()), torch.cuda.Stream()
    bad_a = bad_b = 0
    x = torch.zeros(1 << 20, device=dev)
    for rep in range(6):
        with torch.cuda.stream(sa):
            ra = a.sample(B, cond, n, precision=PRECISION_FAST, seed=3, use_graph=graph)
        with torch.cuda.stream(sb):
            if other == 'engine':
                rb = b.sample(B, cond, n, precision=PRECISION_FAST, seed=4, use_graph=graph)
            elif other == 'torch':
                for _ in range(200):
                    x.add_(1.0)
                rb = al_b
            else:
                rb = al_b
        torch.cuda.synchronize()
        bad_a += int((ra[0] != al_a[0]).sum()) + int((ra[1] != al_a[1]).sum())
        bad_b += int((rb[0] != al_b[0]).sum()) + int((rb[1] != al_b[1]).sum())
    try:
        a.range_check(); b.range_check(); rc = 'ok'
    except Exception as e:
        rc = str(e)[:80]
    print(f'{tag:55s} alone twice equal: {bool(torch.equal(al_a[0], al_a2[0]) and torch.equal(al_a[1], al_a2[1]))}  mismatching codes a: {bad_a}  b: {bad_b}  range_check: {rc}', flush=True)


run('persistent, graphs, engine a ALONE on a side stream', True, True, 'none')
run('persistent, graphs, engine a on the default stream beside torch kernels', True, True, 'torch', side=False)
run('chain only, graphs, two engines', False, True, 'engine')
run('persistent, graphs, engine a beside torch kernels', True, True, 'torch')
run('persistent, graphs, two engines', True, True, 'engine')
run('persistent, eager, two engines', True, False, 'engine')
