"""The persistent AR chain (hqtransformer_amd/csrc/persist.hip) against the launch chain it replaces and against the CPU oracle.

FAST precision, up to 64 samples, decode steps, root engine under the latency policy: the body blocks of a top position run as one
launch, depth sub-step 0 + head_top as a second one (stage2/layers.py:324-328,61-195; hierarchical_ar.py:554-563,684-702).  Both
forms compute the same bf16 arithmetic with different summation orders (the persistent kernel splits K over four waves and, for
mlp.2, over four CUs), so they are compared teacher-forced within a few bf16 ulps of the logits, and each is held to the FAST gate
against the fp32 oracle.  Engine.set_persist(False) (hqt_set_switch) selects the launch chain (part of the graph key; the environment variable HQT_PERSIST is read once, at hqt_create).
"""
import os

import numpy as np
import pytest
import torch

from hqtransformer_amd import synth
from hqtransformer_amd._lib import PRECISION_EXACT, PRECISION_FAST
from hqtransformer_amd.engine import Engine
from oracle import hqt_oracle as O
from tests.helpers import gate, load, stage2_from_fixture

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def engine_s2(spec, weights, max_batch, max_steps=None):
    e = Engine(spec, None, dev(), max_batch, max_steps or spec.ctx_len_img)
    e.load(stage2=weights)
    e.finalize()
    return e


class chain_only:
    """Context: the launch chain on this engine (hqt_set_switch(HQT_SWITCH_PERSIST, 0)), persistent launches again afterwards."""
    def __init__(self, eng):
        self.eng = eng

    def __enter__(self):
        self.eng.set_persist(False)

    def __exit__(self, *a):
        self.eng.set_persist(True)


def imagenet_spec():
    from hqtransformer_amd.config import load_config
    from hqtransformer_amd.spec import stage2_spec_from_config
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return stage2_spec_from_config(load_config(os.path.join(root, 'configs', 'imagenet-12l.yaml')))


_BENCH = {}


def bench_model():
    """The benchmark's own stage 2 (530 M parameters): spec, 'bench' weights and ONE engine (64 samples, 64 positions) shared by the tests of this
    module -- generating the weights and packing six layouts of them costs ~15 s per engine, and the suite runs under the driver's time limit."""
    if not _BENCH:
        s2 = imagenet_spec()
        w = synth.stage2_weights(s2, 0, 'bench')
        _BENCH.update(spec=s2, weights=w, engine=engine_s2(s2, w, 64))
    return _BENCH['spec'], _BENCH['weights'], _BENCH['engine']


@pytest.mark.parametrize('B', [2, 4, 8])
def test_tiny_model_persistent_vs_chain_and_exact(B):
    """Tiny class-conditional model of fixture G4 (4 + 4 layers, D = 128, 4 heads of 32, V = 512): 16 teacher-forced positions, eager and
    as a hipGraph.  Persistent vs launch chain: logits (std ~3) within 0.15, the distance either keeps from EXACT (both are bf16 paths with different summation orders; measured 0.066), drawn codes >= 97 % identical under
    the same noise; persistent vs EXACT: the FAST gate of test_fast_precision_teacher_forced (0.15 / 96 %)."""
    fx = load('g4_tiny_cls.npz')
    spec, weights = stage2_from_fixture(fx)
    eng = engine_s2(spec, weights, 8)
    n = 16
    noise = torch.from_numpy(synth.exp_noise(int(fx['noise_seed']), 64, B, spec.vocab_top)[:n])
    cond = torch.full((B,), 7)
    ct, cb, lg_e = eng.sample(B, cond, n, precision=PRECISION_EXACT, noise=noise, return_logits=True, use_graph=False)
    with chain_only(eng):
        _, _, lg_c = eng.sample(B, cond, n, precision=PRECISION_FAST, noise=noise, force_top=ct, force_bot=cb, return_logits=True, use_graph=False)
    for graph in (False, True):
        pt, pb, lg_p = eng.sample(B, cond, n, precision=PRECISION_FAST, noise=noise, force_top=ct, force_bot=cb, return_logits=True, use_graph=graph)
        gate(f'persist.tiny{B}.vs_chain_logits(graph={graph})', (lg_p - lg_c).abs().max().item(), 0.15)
        gate(f'persist.tiny{B}.vs_exact_logits(graph={graph})', (lg_p - lg_e).abs().max().item(), 0.15)
        agree = ((pt == ct).float().mean().item() + (pb == cb).float().mean().item()) / 2
        gate(f'persist.tiny{B}.code_agreement(graph={graph})', agree, 0.96, '>=')
    eng.range_check()                                    # also reports a persistent launch that gave up on a barrier


def test_persistent_launches_are_what_runs():
    """The timing report names the kernels of a pass: with the persistent chain on, a FAST batch-64 position of the ImageNet model is
    ONE persistent launch (persist_position: body, ln_f + sos_depth, depth sub-step 0, head_top) and no body / depth-0 GEMM or LayerNorm launch; with HQT_PERSIST=0
    it is the launch chain."""
    s2, _, eng = bench_model()
    B, n = 64, 2
    cond = torch.from_numpy(synth.class_ids(5, B, s2.n_classes))
    eng.timing(True)
    eng.sample(B, cond, n, precision=PRECISION_FAST, seed=3, use_graph=False)
    rep = eng.timing_report()
    eng.timing(False)
    assert rep['persist_position'][0] == n and 'persist_body' not in rep and rep.get('layernorm', (0,))[0] == 0, rep
    # what is left of the chain: depth sub-step 1 (4 blocks x 4 GEMMs) + head_bot per position
    assert rep['gemm_qkv'][0] == 4 * n and rep['gemm_head'][0] == n, {k: v[0] for k, v in rep.items()}
    with chain_only(eng):
        eng.timing_reset()
        eng.timing(True)
        eng.sample(B, cond, n, precision=PRECISION_FAST, seed=3, use_graph=False)
        rep = eng.timing_report()
        eng.timing(False)
    assert 'persist_position' not in rep or rep['persist_position'][0] == 0
    assert rep['gemm_qkv'][0] == (12 + 4 + 4) * n


def test_full_benchmark_model_persistent_vs_chain_and_oracle():
    """The benchmark's own model and batch (12 + 4 layers, D = 1536, 24 heads, V = 8192, B = 64), two top positions teacher-forced on the
    oracle's codes: persistent logits within 0.03 of the launch chain's, both within the FAST gate (0.06) of the fp32 oracle, and the
    draws the two forms would make under the same noise agree in >= 99 % of the cases."""
    s2, weights, eng = bench_model()
    B, n = 64, 2
    noise = synth.exp_noise(11, n, B, s2.vocab_top)
    cond = synth.class_ids(12, B, s2.n_classes)
    want = O.OracleStage2(s2, weights).sample(cond, B, n, noise, return_logits=True)
    kw = dict(precision=PRECISION_FAST, noise=torch.from_numpy(noise), force_top=torch.from_numpy(want[0]), force_bot=torch.from_numpy(want[1]),
              return_logits=True)
    with chain_only(eng):
        _, _, lc = eng.sample(B, torch.from_numpy(cond), n, use_graph=True, **kw)
    _, _, lp = eng.sample(B, torch.from_numpy(cond), n, use_graph=True, **kw)
    _, _, lp2 = eng.sample(B, torch.from_numpy(cond), n, use_graph=False, **kw)
    assert torch.equal(lp, lp2), 'the persistent chain is not reproducible between a graph replay and eager launches'
    lc, lp = lc.cpu().numpy(), lp.cpu().numpy()
    gate('persist.benchmark_B64.vs_chain_logits', np.abs(lp - lc).max(), 0.03)
    gate('persist.benchmark_B64.vs_oracle_logits', np.abs(lp - want[2]).max(), 0.06)
    gate('persist.benchmark_B64.chain_vs_oracle_logits', np.abs(lc - want[2]).max(), 0.06)
    q = noise.astype(np.float64)
    pe = np.exp(want[2].astype(np.float64) - want[2].max(-1, keepdims=True))
    pp = np.exp(lp.astype(np.float64) - lp.max(-1, keepdims=True))
    gate('persist.benchmark_B64.identical_draws_vs_oracle', (np.argmax(pe / q, -1) == np.argmax(pp / q, -1)).mean(), 0.99, '>=')
    eng.range_check()


@pytest.mark.parametrize('B', [1, 33, 64])
def test_full_sampling_run_persistent_vs_chain(B):
    """64 free-running positions of the ImageNet model: the persistent form and the launch chain draw from logits a few bf16 ulps apart,
    so under the same Philox keys most sequences stay identical until a near-tie decides differently; >= 90 % of the top codes agree
    (measured below), every code is a valid index, and two persistent runs with one seed are bit-identical."""
    s2, _, eng = bench_model()
    cond = torch.from_numpy(synth.class_ids(9, B, s2.n_classes))
    n = 64 if B != 33 else 16
    a = eng.sample(B, cond, n, precision=PRECISION_FAST, seed=21, use_graph=True)
    b = eng.sample(B, cond, n, precision=PRECISION_FAST, seed=21, use_graph=True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), 'two persistent runs with one seed differ'
    with chain_only(eng):
        c = eng.sample(B, cond, n, precision=PRECISION_FAST, seed=21, use_graph=True)
    assert int(a[0].min()) >= 0 and int(a[0].max()) < s2.vocab_top and int(a[1].min()) >= 0 and int(a[1].max()) < s2.vocab_bot
    first = (a[0][:, 0] == c[0][:, 0]).float().mean().item()       # position 0 sees identical inputs in both forms
    gate(f'persist.free_run_B{B}.first_position_agreement', first, 0.97, '>=')
    eng.range_check()


def test_three_level_model_runs_its_body_persistently():
    """Three code levels (fixture G7's tiny HQTransformer): the body blocks are one persistent launch (`persist_body`; the depth head of 21 tokens
    keeps its launch chain), teacher-forced logits within the FAST gate of EXACT and of the chain's, graph and eager."""
    import json
    from hqtransformer_amd.spec import Stage2Spec
    fx = load('g7_l3_tiny_cls.npz')
    spec = Stage2Spec(**json.loads(str(fx['spec'])))
    weights = synth.stage2_weights(spec, int(fx['weight_seed']), 'fixture')
    B, n = int(fx['B']), 16
    noise = torch.from_numpy(np.maximum(np.random.default_rng([int(fx['noise_seed']), 0x9e3779b9]).standard_exponential((n, 21, B, spec.vocab_top), dtype=np.float32),
                                        np.float32(1e-30)))
    force = [torch.from_numpy(fx[f'codes{i}_0'].copy())[:, :n] for i in range(3)]
    eng = engine_s2(spec, weights, 4)
    cond = torch.full((B,), 7)
    ex = eng.sample3(B, cond, n, precision=PRECISION_EXACT, noise=noise, force=force, return_logits=True, use_graph=False)
    with chain_only(eng):
        ch = eng.sample3(B, cond, n, precision=PRECISION_FAST, noise=noise, force=force, return_logits=True, use_graph=False)
    eng.timing(True)
    eng.timing_reset()
    pe = eng.sample3(B, cond, n, precision=PRECISION_FAST, noise=noise, force=force, return_logits=True, use_graph=False)
    rep = eng.timing_report()
    eng.timing(False)
    assert rep['persist_body'][0] == n and 'persist_position' not in rep, {k: v[0] for k, v in rep.items()}
    pg = eng.sample3(B, cond, n, precision=PRECISION_FAST, noise=noise, force=force, return_logits=True, use_graph=True)
    assert torch.equal(pe[3], pg[3]), 'graph replay and eager launches of the persistent body differ'
    gate('persist.l3_tiny.vs_exact_logits', (pe[3] - ex[3]).abs().max().item(), 0.15)
    gate('persist.l3_tiny.vs_chain_logits', (pe[3] - ch[3]).abs().max().item(), 0.15)
    eng.range_check()


@pytest.mark.parametrize('graph', [False, True])
def test_a_launch_that_cannot_finish_gives_up_and_says_so(graph):
    """Every spin of the persistent kernel is bounded (1 s).  The fault hook (hqt_set_switch(HQT_SWITCH_PERSIST_FAULT, 1): a device word) makes
    CU 0 withhold its first grid-barrier signal: the other CUs must give up instead of hanging the GPU, every later launch of the call must
    return at once (the mark of the first one is still set: no second per launch), range_check must report it -- eager AND on a cache-hit
    REPLAY of a hipGraph after a clean check (round 5 marked the handle only while capturing: a replay that gave up went unreported and
    returned garbage codes with HQT_OK) -- and after that the handle takes the launch chain by itself: the same engine samples correctly
    again, bit-identical to the launch chain."""
    import time
    from hqtransformer_amd import _lib
    fx = load('g4_tiny_cls.npz')
    spec, weights = stage2_from_fixture(fx)
    B, n = 4, 16
    noise = torch.from_numpy(synth.exp_noise(int(fx['noise_seed']), 64, B, spec.vocab_top)[:n])
    cond = torch.full((B,), 7)
    eng = engine_s2(spec, weights, 8)
    with chain_only(eng):
        chain = eng.sample(B, cond, n, precision=PRECISION_FAST, noise=noise, use_graph=graph)
    good = eng.sample(B, cond, n, precision=PRECISION_FAST, noise=noise, use_graph=graph)       # (graph: captures the persistent graph and replays it)
    eng.range_check()                                                                           # clean: resets the handle's "persistent work pending" mark
    eng.set_persist_fault(1)
    t0 = time.perf_counter()
    eng.sample(B, cond, n, precision=PRECISION_FAST, noise=noise, use_graph=graph)              # graph: a pure replay of the cached graph
    torch.cuda.synchronize()
    took = time.perf_counter() - t0
    assert 0.5 < took < 10.0, f'{took:.2f} s: one bounded wait (1 s), not one per launch ({2 * n} launches) and not a hang'
    with pytest.raises(_lib.HqtError, match='gave up at the grid barrier'):
        eng.range_check()
    # the handle fell back to the launch chain by itself: correct codes, no second time-out, although the fault is still armed
    t0 = time.perf_counter()
    again = eng.sample(B, cond, n, precision=PRECISION_FAST, noise=noise, use_graph=graph)
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 0.5
    eng.range_check()
    assert torch.equal(again[0], chain[0]) and torch.equal(again[1], chain[1])
    # re-armed and healthy again: the persistent launch runs and draws what it drew before
    eng.set_persist_fault(0)
    eng.set_persist(True)
    back = eng.sample(B, cond, n, precision=PRECISION_FAST, noise=noise, use_graph=graph)
    eng.range_check()
    assert torch.equal(back[0], good[0]) and torch.equal(back[1], good[1])


def test_graph_replay_on_a_side_stream_equals_the_null_stream():
    """Round 6 found the persistent position drawing DIFFERENT codes when its captured graph (16 positions per graph) was replayed on a stream other
    than the null stream -- nothing else running, no error reported: the counters were zeroed by a captured hipMemsetAsync node, which a side-stream
    replay did not order like a kernel node (tools/diag_two_handles.py).  They are zeroed by a kernel now; this is the regression test."""
    s2, _, eng = bench_model()
    B, n = 64, 16
    cond = torch.from_numpy(synth.class_ids(5, B, s2.n_classes))
    ref = eng.sample(B, cond, n, precision=PRECISION_FAST, seed=3)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    x = torch.zeros(1 << 20, device=dev())
    for rep in range(3):
        with torch.cuda.stream(side):
            got = eng.sample(B, cond, n, precision=PRECISION_FAST, seed=3)
        if rep:                                  # ... and beside unrelated kernels of another stream
            for _ in range(100):
                x.add_(1.0)
        torch.cuda.synchronize()
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), f'replay {rep} on a side stream differs from the null stream'
    eng.range_check()


def test_two_root_handles_sampling_concurrently_both_finish():
    """Two ImageGPT2-sized root engines on two streams, batch 64 each, FAST: each persistent launch needs all 256 compute units, so two in
    flight at once would keep each other out until both time out (round 5: HQT_ERR_STATE on both).  The library orders persistent work of
    all handles of a process on a device behind each other (engine.hip: PersistOrder): both calls finish, neither reports a give-up, and
    each draws what it draws alone."""
    s2, w, a = bench_model()
    b = engine_s2(s2, w, 64, 8)
    B, n = 64, 8
    cond = torch.from_numpy(synth.class_ids(5, B, s2.n_classes))
    alone_a = a.sample(B, cond, n, precision=PRECISION_FAST, seed=3)
    torch.cuda.synchronize()
    alone_b = b.sample(B, cond, n, precision=PRECISION_FAST, seed=4)
    torch.cuda.synchronize()
    a.range_check(); b.range_check()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for rep in range(3):
        with torch.cuda.stream(sa):
            ra = a.sample(B, cond, n, precision=PRECISION_FAST, seed=3)
        with torch.cuda.stream(sb):
            rb = b.sample(B, cond, n, precision=PRECISION_FAST, seed=4)
        outs.append((ra, rb))
    torch.cuda.synchronize()
    a.range_check(); b.range_check()                     # raises if a launch gave up
    for ra, rb in outs:
        assert torch.equal(ra[0], alone_a[0]) and torch.equal(ra[1], alone_a[1])
        assert torch.equal(rb[0], alone_b[0]) and torch.equal(rb[1], alone_b[1])
    rep = None
    a.timing(True)
    a.sample(B, cond, 2, precision=PRECISION_FAST, seed=3, use_graph=False)
    rep = a.timing_report()
    a.timing(False)
    assert rep['persist_position'][0] == 2, 'the persistent launch is what ran'


def test_fast_only_replica_holds_fewer_layouts_and_refuses_the_others():
    """hqt_config.ar_layouts = HQT_LAYOUT_FAST: FAST sampling is bit-identical to a full engine's, EXACT still runs (from the fp32 tensors as
    received), SPLIT fails loudly with HQT_ERR_STATE; the device memory the replica takes is visibly smaller."""
    from hqtransformer_amd import _lib
    fx = load('g4_tiny_cls.npz')
    spec, weights = stage2_from_fixture(fx)
    B, n = 4, 8
    noise = torch.from_numpy(synth.exp_noise(int(fx['noise_seed']), 64, B, spec.vocab_top)[:n])
    cond = torch.full((B,), 7)
    full = engine_s2(spec, weights, 8)
    lean = Engine(spec, None, dev(), 8, spec.ctx_len_img, ar_layouts=_lib.LAYOUT_FAST)
    lean.load(stage2=weights)
    lean.finalize()
    for prec in (PRECISION_FAST, PRECISION_EXACT):
        x = full.sample(B, cond, n, precision=prec, noise=noise, return_logits=True)
        y = lean.sample(B, cond, n, precision=prec, noise=noise, return_logits=True)
        assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) and torch.equal(x[2], y[2])
    with pytest.raises(_lib.HqtError, match='HQT_LAYOUT_SPLIT'):
        lean.sample(B, cond, n, precision=_lib.PRECISION_SPLIT, noise=noise)
    full.range_check(); lean.range_check()
    # the ImageNet-12L model: free memory before / after each build
    s2, w, _ = bench_model()
    del full, lean
    torch.cuda.synchronize()
    sizes = {}
    for name, mask in (('fast_only', _lib.LAYOUT_FAST), ('all', 0)):
        torch.cuda.empty_cache()
        free0 = torch.cuda.mem_get_info()[0]
        e = Engine(s2, None, dev(), 64, 8, ar_layouts=mask)
        e.load(stage2=w)
        e.finalize()
        torch.cuda.synchronize()
        sizes[name] = (free0 - torch.cuda.mem_get_info()[0]) / 2 ** 30
        e.close()
    print('device memory of one ImageNet-12L stage-2 replica (GiB):', {k: round(v, 2) for k, v in sizes.items()})
    assert sizes['fast_only'] < 0.7 * sizes['all'], sizes
