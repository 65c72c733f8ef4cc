"""Parity of the HIP path (through the C ABI) against the reference-generated golden fixtures and
against the CPU oracle on the same seeded inputs.  Runs on the MI355X box only (-m gpu).

Bars (EXACT precision = what the reference computes on its CPU path):
  * sampled code sequences: bit-exact under fixed noise;
  * fp32 logits: |diff| <= 2e-4 (logits of std ~3 after up to 8 fp32 blocks, summation order differs);
  * decoded pixels: |diff| <= 1e-4 (north_star's tolerance).
FAST precision (bf16 weights/activations, fp32 accumulation) is gated by teacher-forced logits and
pixel tolerances stated in each test, plus the code agreement rate.
"""
import json

import numpy as np
import pytest
import torch

from hqtransformer_amd import synth
from hqtransformer_amd._lib import PRECISION_EXACT, PRECISION_FAST
from hqtransformer_amd.engine import Engine
from hqtransformer_amd.spec import Stage1Spec, Stage2Spec
from oracle import hqt_oracle as O
from tests.helpers import gate, load, oracle_stage1, oracle_stage2, stage1_from_fixture, stage2_from_fixture

pytestmark = pytest.mark.gpu
LOGIT_TOL = 2e-4
PIXEL_TOL = 1e-4


def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def engine_s2(spec, weights, max_batch, max_steps=None):
    e = Engine(spec, None, dev(), max_batch, max_steps or spec.ctx_len_img)
    e.load(stage2=weights)
    e.finalize()
    return e


def engine_s1(spec, weights, max_batch):
    e = Engine(None, spec, dev(), max_batch)
    e.load(stage1=weights)
    e.finalize()
    return e


def np_(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope='module')
def tiny_cls():
    fx = load('g4_tiny_cls.npz')
    spec, weights = stage2_from_fixture(fx)
    return fx, spec, weights, engine_s2(spec, weights, 8)


@pytest.mark.parametrize('si', [0, 1, 2])
@pytest.mark.parametrize('graph', [False, True])
def test_tiny_cls_codes_bit_exact_vs_reference_fixture(tiny_cls, si, graph):
    fx, spec, weights, eng = tiny_cls
    tk, tp, T = json.loads(str(fx['settings']))[si]
    B, n = int(fx['B']), int(fx['n_steps'])
    noise = synth.exp_noise(int(fx['noise_seed']), n, B, spec.vocab_top)
    ct, cb, lg = eng.sample(B, torch.full((B,), 7), n, precision=PRECISION_EXACT, top_k=tk, top_p=tp, temperature=T,
                            noise=torch.from_numpy(noise), return_logits=True, use_graph=graph)
    torch.cuda.synchronize()
    scale = np.array([T[0]] + [T[1]] * 4, np.float32)[None, :, None, None]
    err = np.abs(np_(lg)[fx['keep_steps']] / scale - fx[f'logits_{si}']).max()
    assert err <= LOGIT_TOL, f'logit error {err}'
    assert (np_(ct) == fx[f'codes_top_{si}']).all(), 'top codes differ from the reference'
    assert (np_(cb) == fx[f'codes_bot_{si}']).all(), 'bottom codes differ from the reference'


def test_tiny_cls_given_top_code(tiny_cls):
    fx, spec, weights, eng = tiny_cls
    B = int(fx['B'])
    noise = synth.exp_noise(int(fx['noise_seed']), 64, B, spec.vocab_top)[:8]
    ct, cb, lg = eng.sample(B, torch.full((B,), 3), 8, precision=PRECISION_EXACT, noise=torch.from_numpy(noise.copy()),
                            force_top=torch.from_numpy(fx['given_top']), return_logits=True, use_graph=False)
    assert (np_(cb) == fx['given_codes_bot']).all()
    assert np.abs(np_(lg) - fx['given_logits']).max() <= LOGIT_TOL


def test_tiny_reduce_uncond(tiny_cls):
    fx = load('g3_tiny_reduce_uncond.npz')
    spec, weights = stage2_from_fixture(fx)
    eng = engine_s2(spec, weights, 4)
    B, n = int(fx['B']), int(fx['n_steps'])
    noise = synth.exp_noise(int(fx['noise_seed']), n, B, spec.vocab_top)
    k, p, T = int(fx['top_k']), float(fx['top_p']), [float(t) for t in fx['temps']]
    ct, cb, lg = eng.sample(B, None, n, precision=PRECISION_EXACT, top_k=(k, k), top_p=(p, p), temperature=T,
                            noise=torch.from_numpy(noise), return_logits=True)
    scale = np.array([T[0]] + [T[1]] * 4, np.float32)[None, :, None, None]
    assert np.abs(np_(lg)[fx['keep_steps']] / scale - fx['logits']).max() <= LOGIT_TOL
    assert (np_(ct) == fx['codes_top']).all() and (np_(cb) == fx['codes_bot']).all()


def test_tiny_txt_prefill():
    fx = load('g3_tiny_txt.npz')
    spec, weights = stage2_from_fixture(fx)
    eng = engine_s2(spec, weights, 4)
    B, n = int(fx['B']), int(fx['n_steps'])
    noise = synth.exp_noise(int(fx['noise_seed']), n, B, spec.vocab_top)
    txt = synth.text_ids(int(fx['text_seed']), B, spec.ctx_len_txt, spec.vocab_txt)
    for graph in (False, True):
        ct, cb, lg = eng.sample(B, torch.from_numpy(txt), n, precision=PRECISION_EXACT, noise=torch.from_numpy(noise),
                                return_logits=True, use_graph=graph)
        assert np.abs(np_(lg)[fx['keep_steps']] - fx['logits']).max() <= LOGIT_TOL
        assert (np_(ct) == fx['codes_top']).all() and (np_(cb) == fx['codes_bot']).all()
    # FAST arithmetic through the same prefill (bf16, MFMA): teacher-forced on the reference's codes, logits within the
    # bf16 budget -- on a workspace deliberately filled with NaNs first (a prefill that read a stale packed activation
    # buffer once passed on freshly zeroed memory)
    junk = torch.full((64 << 20,), float('nan'), device=dev())
    del junk
    eng2 = engine_s2(spec, weights, 4)
    ft, fb = torch.from_numpy(fx['codes_top'].copy()), torch.from_numpy(fx['codes_bot'].copy())
    for graph in (False, True):
        _, _, lf = eng2.sample(B, torch.from_numpy(txt), n, precision=PRECISION_FAST, noise=torch.from_numpy(noise), force_top=ft,
                               force_bot=fb, return_logits=True, use_graph=graph)
        err = np.abs(np_(lf)[fx['keep_steps']] - fx['logits']).max()
        gate(f'tiny_txt_prefill.fast_logits(graph={graph})', err, 0.15)


@pytest.mark.parametrize('T', [20, 32, 48, 64])
def test_causal_prefill_on_the_matrix_cores_vs_oracle(T):
    """The causal prompt prefill of FAST precision at head size 64 (attention_prefill_mfma_kernel: one wave per (sample, head),
    S^T = K Q^T and O^T = V^T P^T on v_mfma_f32_32x32x16_bf16, layers.py:107-111) for prompt lengths that fill one tile, straddle
    two and fill both -- rows beyond the prompt are clamped and masked, key tiles above the diagonal skipped.  Two body layers, so that
    the second layer's keys depend on the first layer's attention output at EVERY prompt position (a wrong row anywhere in the
    T x T attention changes the logits); teacher-forced on the oracle's codes, logits inside the bf16 gate, on a NaN-poisoned
    workspace (cache rows beyond the prompt must never be read), batch 5 (the last workgroup has idle waves)."""
    spec = Stage2Spec(embed_dim=128, n_layers=2, n_heads=2, n_layers_depth=1, vocab_top=256, vocab_bot=256, vocab_txt=512,
                      ctx_len_img=64, ctx_len_txt=T, n_classes=0, cond=2, embedding=0)
    weights = synth.stage2_weights(spec, 501 + T, 'fixture')
    B, n = 5, 3
    noise = synth.exp_noise(502, n, B, spec.vocab_top)
    txt = synth.text_ids(503, B, T, spec.vocab_txt)
    want = O.OracleStage2(spec, weights).sample(txt, B, n, noise, return_logits=True)
    import os
    os.environ['HQT_POISON_WORKSPACE'] = '1'
    try:
        eng = engine_s2(spec, weights, B, 8)
    finally:
        del os.environ['HQT_POISON_WORKSPACE']
    ct, cb, lg = eng.sample(B, torch.from_numpy(txt), n, precision=PRECISION_EXACT, noise=torch.from_numpy(noise), return_logits=True)
    assert np.abs(np_(lg) - want[2]).max() <= LOGIT_TOL and (np_(ct) == want[0]).all() and (np_(cb) == want[1]).all()
    ft, fb = torch.from_numpy(want[0]), torch.from_numpy(want[1])
    for graph in (False, True):
        _, _, lf = eng.sample(B, torch.from_numpy(txt), n, precision=PRECISION_FAST, noise=torch.from_numpy(noise), force_top=ft, force_bot=fb,
                              return_logits=True, use_graph=graph)
        assert bool(torch.isfinite(lf).all())
        gate(f'prefill_mfma.fast_logits(T={T},graph={graph})', np.abs(np_(lf) - want[2]).max(), 0.15)


def test_ragged_batches_and_b1_vs_oracle(tiny_cls):
    """B = 1 (which the reference cannot run, hierarchical_ar.py:719) and an odd batch, against the oracle."""
    fx, spec, weights, eng = tiny_cls
    orc = O.OracleStage2(spec, weights)
    for B, n in ((1, 6), (5, 5), (8, 4)):
        noise = synth.exp_noise(100 + B, n, B, spec.vocab_top)
        cond = np.arange(B) % spec.n_classes
        want = orc.sample(cond, B, n, noise, (50, 20), (None, 0.9), (1.0, 0.8), return_logits=True)
        got = eng.sample(B, torch.from_numpy(cond), n, precision=PRECISION_EXACT, top_k=(50, 20), top_p=(None, 0.9),
                         temperature=(1.0, 0.8), noise=torch.from_numpy(noise), return_logits=True)
        assert np.abs(np_(got[2]) - want[2]).max() <= LOGIT_TOL
        assert (np_(got[0]) == want[0]).all() and (np_(got[1]) == want[1]).all()


def test_philox_noise_is_shard_invariant(tiny_cls):
    """Sharding property: rows [2,6) of a global batch of 8 drawn with sample_offset=2 equal
    the same rows of the unsharded run (size-independent property, no oracle needed)."""
    fx, spec, weights, eng = tiny_cls
    cond = torch.arange(8) % spec.n_classes
    full = eng.sample(8, cond, 10, precision=PRECISION_EXACT, seed=1234, top_k=(100, 100))
    part = eng.sample(4, cond[2:6], 10, precision=PRECISION_EXACT, seed=1234, sample_offset=2, top_k=(100, 100))
    assert (np_(full[0])[2:6] == np_(part[0])).all() and (np_(full[1])[2:6] == np_(part[1])).all()
    again = eng.sample(8, cond, 10, precision=PRECISION_EXACT, seed=1234, top_k=(100, 100))
    assert (np_(full[0]) == np_(again[0])).all()                      # deterministic
    other = eng.sample(8, cond, 10, precision=PRECISION_EXACT, seed=99, top_k=(100, 100))
    assert (np_(full[0]) != np_(other[0])).any()
    assert np_(full[0]).min() >= 0 and np_(full[0]).max() < spec.vocab_top


def test_philox_draws_follow_the_softmax(tiny_cls):
    """Statistical property of the in-kernel Exp(1) noise: with B identical rows the empirical top-code
    histogram of step 0 must match softmax(logits) (chi-square-ish bound)."""
    fx, spec, weights, eng = tiny_cls
    B = 8
    counts = np.zeros(spec.vocab_top)
    logits = None
    for rep in range(64):
        ct, cb, lg = eng.sample(B, torch.full((B,), 2), 1, precision=PRECISION_EXACT, seed=rep, return_logits=True)
        counts += np.bincount(np_(ct)[:, 0], minlength=spec.vocab_top)
        logits = np_(lg)[0, 0, 0]
    p = O.softmax(logits[None])[0]
    n = counts.sum()
    top = np.argsort(-p)[:8]
    assert np.abs(counts[top] / n - p[top]).max() < 5 * np.sqrt(p[top].max() / n) + 0.01


def test_fast_precision_teacher_forced(tiny_cls):
    """FAST (bf16) arithmetic cannot be bit-exact against an fp32 reference; it is gated by teacher-forced
    logits (|diff| <= 0.15 on logits of std ~3, i.e. a few bf16 ulps through 8 blocks; measured 0.090) and by agreement of
    the drawn codes under identical noise (>= 96 %; measured 98.2 %).  Every FAST gate of this suite sits at about twice the
    figure measured on the MI355X (profiles/r02_fast_gates.txt)."""
    fx, spec, weights, eng = tiny_cls
    B, n = int(fx['B']), 16
    noise = synth.exp_noise(int(fx['noise_seed']), 64, B, spec.vocab_top)[:n]
    ft = torch.from_numpy(fx['codes_top_0'][:, :n].copy())
    fb = torch.from_numpy(fx['codes_bot_0'][:, :n].copy())
    for graph in (False, True):
        ct, cb, lg = eng.sample(B, torch.full((B,), 7), n, precision=PRECISION_FAST, noise=torch.from_numpy(noise),
                                force_top=ft, force_bot=fb, return_logits=True, use_graph=graph)
        ex = eng.sample(B, torch.full((B,), 7), n, precision=PRECISION_EXACT, noise=torch.from_numpy(noise),
                        force_top=ft, force_bot=fb, return_logits=True, use_graph=False)
        err = (lg - ex[2]).abs().max().item()
        gate(f'tiny_cls.fast_logits(graph={graph})', err, 0.15)
        agree = ((ct == ex[0]).float().mean().item() + (cb == ex[1]).float().mean().item()) / 2
        gate(f'tiny_cls.fast_code_agreement(graph={graph})', agree, 0.96, '>=')


def test_fast_wide_passes_up_to_4096_rows(tiny_cls):
    """Merged passes reach the streaming GEMMs with 512 .. 4096 activation rows (depth sub-step 1 runs 4 rows per sample): the
    packed-operand layout with 16 .. 128 row blocks and the pipelined 64-row-tile variants.  Same gates as the 2-row-block case."""
    fx, spec, weights, _ = tiny_cls
    for B in (160, 320, 1000):                               # 640 / 1280 / 4000 rows in depth sub-step 1; 1000 is ragged (not a multiple of 32)
        eng = engine_s2(spec, weights, B)
        n = 3
        noise = torch.from_numpy(synth.exp_noise(11, n, B, spec.vocab_top))
        cond = torch.from_numpy(synth.class_ids(6, B, spec.n_classes))
        ct, cb, lg_e = eng.sample(B, cond, n, precision=PRECISION_EXACT, noise=noise, return_logits=True, use_graph=False)
        for graph in (False, True):
            ft, fb, lg_f = eng.sample(B, cond, n, precision=PRECISION_FAST, noise=noise, force_top=ct, force_bot=cb, return_logits=True, use_graph=graph)
            gate(f'tiny_cls.wide{B}.fast_logits(graph={graph})', (lg_f - lg_e).abs().max().item(), 0.15)
            agree = ((ft == ct).float().mean().item() + (fb == cb).float().mean().item()) / 2
            gate(f'tiny_cls.wide{B}.fast_code_agreement(graph={graph})', agree, 0.96, '>=')
        del eng


def test_fast_single_key_shortcut_is_bit_identical(tiny_cls):
    """Depth sub-step 0 attends to exactly one key, so its attention output is the value row: the FAST path skips the
    query third of the fused GEMM and the attention launch.  softmax of one score is exactly 1.0, hence logits and
    codes must be BIT-identical to the long way round (hqt_set_switch(HQT_SWITCH_SINGLE_KEY, 0))."""
    import os
    fx, spec, weights, eng = tiny_cls
    B, n = int(fx['B']), 8
    noise = torch.from_numpy(synth.exp_noise(int(fx['noise_seed']), 64, B, spec.vocab_top)[:n])
    eng.set_persist(False)                       # the launch chain's claim (the persistent chain of round 5 has no long way round: tests/test_gpu_persist.py)
    try:
        a = eng.sample(B, torch.full((B,), 3), n, precision=PRECISION_FAST, noise=noise, return_logits=True, use_graph=False)
        eng.set_single_key(False)
        try:
            b = eng.sample(B, torch.full((B,), 3), n, precision=PRECISION_FAST, noise=noise, return_logits=True, use_graph=False)
        finally:
            eng.set_single_key(True)
    finally:
        eng.set_persist(True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert torch.equal(a[2], b[2])


@pytest.mark.parametrize('B', [2, 64])
def test_imagenet_head_geometry_vs_oracle(B):
    """One body + one depth layer at the ImageNet width (D = 1536, 24 heads of 64; the block arithmetic at this shape is
    pinned to the reference by fixture G2): EXACT codes bit-exact and logits <= 2e-4 against the oracle; FAST
    teacher-forced logits within the bf16 budget.  B = 64 runs the streaming-GEMM variants the benchmark uses
    (64- and 256-row tiles, single-key shortcut, deferred LayerNorm) on non-trivial data."""
    spec = Stage2Spec(embed_dim=1536, n_layers=1, n_heads=24, n_layers_depth=1, vocab_top=512, vocab_bot=512, vocab_txt=64,
                      ctx_len_img=64, ctx_len_txt=16, n_classes=10, cond=1, embedding=0)
    weights = synth.stage2_weights(spec, 91, 'fixture')
    n = 4 if B > 8 else 6
    noise = synth.exp_noise(92, n, B, spec.vocab_top)
    cond = np.arange(B) % spec.n_classes
    want = O.OracleStage2(spec, weights).sample(cond, B, n, noise, (None, 64), (None, 0.9), (1.0, 0.9), return_logits=True)
    eng = engine_s2(spec, weights, B)
    ct, cb, lg = eng.sample(B, torch.from_numpy(cond), n, precision=PRECISION_EXACT, top_k=(None, 64), top_p=(None, 0.9),
                            temperature=(1.0, 0.9), noise=torch.from_numpy(noise), return_logits=True, use_graph=False)
    assert np.abs(np_(lg) - want[2]).max() <= LOGIT_TOL
    assert (np_(ct) == want[0]).all() and (np_(cb) == want[1]).all()
    ft, fb = torch.from_numpy(want[0]), torch.from_numpy(want[1])
    for graph in (False, True):
        _, _, lf = eng.sample(B, torch.from_numpy(cond), n, precision=PRECISION_FAST, top_k=(None, 64), top_p=(None, 0.9),
                              temperature=(1.0, 0.9), noise=torch.from_numpy(noise), force_top=ft, force_bot=fb,
                              return_logits=True, use_graph=graph)
        err = np.abs(np_(lf) - want[2]).max()
        gate(f'imagenet_head_geometry.fast_logits(B={B},graph={graph})', err, 0.1)
    # throughput-oriented tile shapes (hqt_set_policy; what several lanes in flight run): same bar
    eng.set_policy(1)
    try:
        for graph in (False, True):
            _, _, lt = eng.sample(B, torch.from_numpy(cond), n, precision=PRECISION_FAST, top_k=(None, 64), top_p=(None, 0.9),
                                  temperature=(1.0, 0.9), noise=torch.from_numpy(noise), force_top=ft, force_bot=fb,
                                  return_logits=True, use_graph=graph)
            err = np.abs(np_(lt) - want[2]).max()
            gate(f'imagenet_head_geometry.fast_logits_throughput_policy(B={B},graph={graph})', err, 0.1)
            gate(f'imagenet_head_geometry.policy_difference(B={B},graph={graph})', np.abs(np_(lt) - np_(lf)).max(), 0.05)   # fp32 summation order only
    finally:
        eng.set_policy(0)


@pytest.mark.parametrize('opts', [((None, None), (None, None), (1.0, 1.0)), ((2048, 100), (1.0, 0.9), (0.95, 0.8))])
def test_full_vocabulary_sampler_vs_oracle(opts):
    """V = 8192 (the ImageNet vocabulary) takes the 1024-thread sampler: temperature, radix-select top-k, sorted top-p with
    the double prefix, argmax(p / q) -- codes bit-exact and logits <= 2e-4 against the oracle on a small model."""
    tk, tp, T = opts
    spec = Stage2Spec(embed_dim=64, n_layers=1, n_heads=2, n_layers_depth=1, vocab_top=8192, vocab_bot=8192, vocab_txt=64,
                      ctx_len_img=64, ctx_len_txt=16, n_classes=10, cond=1, embedding=0)
    weights = synth.stage2_weights(spec, 51, 'fixture')
    B, n = 3, 3
    noise = synth.exp_noise(52, n, B, spec.vocab_top)
    cond = np.array([1, 5, 9])
    want = O.OracleStage2(spec, weights).sample(cond, B, n, noise, tk, tp, T, return_logits=True)
    eng = engine_s2(spec, weights, B)
    for graph in (False, True):
        ct, cb, lg = eng.sample(B, torch.from_numpy(cond), n, precision=PRECISION_EXACT, top_k=tk, top_p=tp, temperature=T,
                                noise=torch.from_numpy(noise), return_logits=True, use_graph=graph)
        assert np.abs(np_(lg) - want[2]).max() <= LOGIT_TOL
        assert (np_(ct) == want[0]).all() and (np_(cb) == want[1]).all()


# ----------------------------------------------------------------------------------------- stage 1
def test_decode_64_exact_vs_reference_fixture():
    fx = load('g5_decode_64.npz')
    spec, weights = stage1_from_fixture(fx)
    eng = engine_s1(spec, weights, 2)
    ct, cb = torch.from_numpy(fx['code_t']), torch.from_numpy(fx['code_b'])
    px = np_(eng.decode(ct, cb, precision=PRECISION_EXACT))
    assert np.abs(px - fx['pixels']).max() <= PIXEL_TOL
    assert np.abs(np_(eng.decode(ct[:1], None, precision=PRECISION_EXACT)) - fx['pixels_top_only']).max() <= PIXEL_TOL
    assert np.abs(np_(eng.decode(None, cb[:1], precision=PRECISION_EXACT)) - fx['pixels_bot_only']).max() <= PIXEL_TOL
    clamped = np_(eng.decode(ct, cb, precision=PRECISION_EXACT, clamp01=True))
    np.testing.assert_allclose(clamped, O.postprocess(fx['pixels']), atol=PIXEL_TOL)
    # sampler layout + folded rearranges (sampling_hqmodel.py:119-120)
    seq_t = ct.reshape(2, 64)
    seq_b = cb.reshape(2, 8, 2, 8, 2).permute(0, 1, 3, 2, 4).reshape(2, 64, 4)
    px_seq = np_(eng.decode(seq_t, seq_b, precision=PRECISION_EXACT, seq_layout=True))
    assert (px_seq == px).all()


def test_decode_256_exact_vs_reference_fixture():
    fx = load('g5_decode_256.npz')
    spec, weights = stage1_from_fixture(fx)
    eng = engine_s1(spec, weights, 1)
    px = np_(eng.decode(torch.from_numpy(fx['code_t']), torch.from_numpy(fx['code_b']), precision=PRECISION_EXACT))
    assert np.abs(px - fx['pixels']).max() <= PIXEL_TOL


def test_decode_fast_tolerance():
    """FAST decode (bf16 activations and filters, fp32 accumulation and GroupNorm statistics): pixels of
    range ~[-5, 5] within 0.1 of the fp32 reference (2 % of the range), mean abs error within 1e-2."""
    for name in ('g5_decode_64.npz', 'g5_decode_256.npz'):
        fx = load(name)
        spec, weights = stage1_from_fixture(fx)
        eng = engine_s1(spec, weights, 2)
        px = np_(eng.decode(torch.from_numpy(fx['code_t']), torch.from_numpy(fx['code_b']), precision=PRECISION_FAST))
        d = np.abs(px - fx['pixels'])
        gate(f'decode_fast.{name}.max', d.max(), 0.06 if '64' in name else 0.12)
        gate(f'decode_fast.{name}.mean', d.mean(), 1e-2)


def test_decode_fast_big_tile_kernels_vs_oracle():
    """FAST decode on a config wide enough (Cin % 64 == 0, Cout >= 128) for the LDS-DMA 128x128x64 MFMA conv
    kernel; HQT_FORCE_TILE128 makes the dispatcher pick it even though the grid is small.  Same tolerance as
    test_decode_fast_tolerance, against the CPU oracle."""
    import os
    spec = Stage1Spec(ch=64, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[16], resolution=64, z_channels=64,
                      embed_dim=32, n_embed=256)
    weights = synth.stage1_weights(spec, 31, 'fixture')
    r = np.random.default_rng(32)
    ct, cb = r.integers(0, 256, (3, 8, 8)), r.integers(0, 256, (3, 16, 16))
    want = O.OracleStage1(spec, weights).decode_code(ct, cb)
    os.environ['HQT_FORCE_TILE128'] = '1'
    try:
        eng = engine_s1(spec, weights, 3)
        got = np_(eng.decode(torch.from_numpy(ct), torch.from_numpy(cb), precision=PRECISION_FAST))
        exact = np_(eng.decode(torch.from_numpy(ct), torch.from_numpy(cb), precision=PRECISION_EXACT))
    finally:
        del os.environ['HQT_FORCE_TILE128']
    assert np.abs(exact - want).max() <= PIXEL_TOL
    d = np.abs(got - want)
    gate('decode_fast_big_tile.max', d.max(), 0.06)
    gate('decode_fast_big_tile.mean', d.mean(), 1e-2)


def test_decode_fast_halo_conv_matches_generic_implicit_gemm():
    """The 3x3 'halo tile' conv kernel (input patch resident in LDS, chunk-major K order) against the generic LDS-DMA
    implicit-GEMM kernel (HQT_NO_HALO=1) on the same bf16 inputs: they differ only in fp32 summation order (then bf16
    rounding per layer), so they agree to about one bf16 ulp per pixel and are equally far from the fp32 oracle.  Covers upsampling convs, the
    residual epilogue, image borders (zero page) and the NCHW conv_out store."""
    import os
    spec = Stage1Spec(ch=64, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[16], resolution=64, z_channels=64,
                      embed_dim=32, n_embed=256)
    weights = synth.stage1_weights(spec, 41, 'fixture')
    r = np.random.default_rng(42)
    ct, cb = r.integers(0, 256, (5, 8, 8)), r.integers(0, 256, (5, 16, 16))
    os.environ['HQT_FORCE_TILE128'] = '1'
    halo = {}
    try:
        eng = engine_s1(spec, weights, 5)
        for ty in ('8', '16'):                      # 8 x 16 and 16 x 16 pixel tiles (the latter: single patch buffer, two-half epilogue)
            os.environ['HQT_HALO_TY'] = ty
            halo[ty] = np_(eng.decode(torch.from_numpy(ct), torch.from_numpy(cb), precision=PRECISION_FAST))
            os.environ['HQT_NO_NARROW_OUT'] = '1'   # conv_out through the 128-channel tiles instead of the 32-channel variant
            wide_out = np_(eng.decode(torch.from_numpy(ct), torch.from_numpy(cb), precision=PRECISION_FAST))
            del os.environ['HQT_NO_NARROW_OUT']
            assert np.array_equal(wide_out, halo[ty]), ty   # same products in the same order: bit-identical
        os.environ['HQT_NO_FUSED_GN'] = '1'         # statistics by the separate pass instead of the conv epilogue
        unfused = np_(eng.decode(torch.from_numpy(ct), torch.from_numpy(cb), precision=PRECISION_FAST))
        os.environ['HQT_NO_HALO'] = '1'
        generic = np_(eng.decode(torch.from_numpy(ct), torch.from_numpy(cb), precision=PRECISION_FAST))
    finally:
        for k in ('HQT_NO_HALO', 'HQT_HALO_TY', 'HQT_NO_FUSED_GN', 'HQT_FORCE_TILE128', 'HQT_NO_NARROW_OUT'):
            os.environ.pop(k, None)
    want = O.OracleStage1(spec, weights).decode_code(ct, cb)
    eg = np.abs(generic - want)
    for ty, px in list(halo.items()) + [('16, separate GroupNorm statistics', unfused)]:
        eh = np.abs(px - want)
        gate(f'halo_conv.fast_vs_oracle.max({ty})', eh.max(), 0.1)
        assert eh.mean() <= 1.25 * eg.mean() + 1e-3, (ty, eh.max(), eh.mean(), eg.mean())
        d = np.abs(px - generic)                    # ~1 bf16 ulp of an O(1) pixel on average, no outliers (a wrong tap or border would be O(1))
        gate(f'halo_conv.vs_generic_kernel.max({ty})', d.max(), 0.06)
        gate(f'halo_conv.vs_generic_kernel.mean({ty})', d.mean(), 8e-3)
    if not np.array_equal(halo['8'], halo['16']):
        gate('halo_conv.tile_heights_8_vs_16.max', np.abs(halo['8'] - halo['16']).max(), 0.06)


def test_decode_batch_chunking_and_ragged():
    """More images than one decode chunk, decoded in one call, equal the per-image decodes."""
    fx = load('g5_decode_64.npz')
    spec, weights = stage1_from_fixture(fx)
    eng = engine_s1(spec, weights, 3)
    r = np.random.default_rng(5)
    ct = torch.from_numpy(r.integers(0, spec.n_embed, (7, 8, 8)))
    cb = torch.from_numpy(r.integers(0, spec.n_embed, (7, 16, 16)))
    allpx = np_(eng.decode(ct, cb, precision=PRECISION_EXACT))
    for i in (0, 3, 6):
        one = np_(eng.decode(ct[i:i + 1], cb[i:i + 1], precision=PRECISION_EXACT))
        assert (one[0] == allpx[i]).all()
    orc = O.OracleStage1(spec, weights)
    assert np.abs(orc.decode_code(ct[:2].numpy(), cb[:2].numpy()) - allpx[:2]).max() <= PIXEL_TOL


def test_error_reporting(tiny_cls):
    from hqtransformer_amd._lib import HqtError
    fx, spec, weights, eng = tiny_cls
    with pytest.raises(HqtError):
        eng.sample(9, torch.zeros(9, dtype=torch.int64), 4)           # B > max_batch
    with pytest.raises(HqtError):
        eng.sample(2, torch.zeros(2, dtype=torch.int64), 65)          # n_steps > ctx_len_img
    e2 = Engine(spec, None, dev(), 2)
    with pytest.raises(HqtError):
        e2.finalize()                                                 # missing weights


# ----------------------------------------------------------------------------------------- three code levels
def test_l3_sampling_vs_reference_fixture_and_oracle():
    """Three-level HQTransformer ('parallel-add', 1 + 4 + 16 codes per position) through hqt_sample_l3: EXACT codes bit-exact
    and logits <= 2e-4 against fixture G7 generated from the reference, eager and graph; FAST teacher-forced logits
    within the bf16 budget."""
    fx = load('g7_l3_tiny_cls.npz')
    spec = Stage2Spec(**json.loads(str(fx['spec'])))
    weights = synth.stage2_weights(spec, int(fx['weight_seed']), 'fixture')
    B, n = int(fx['B']), int(fx['n_steps'])
    noise = np.maximum(np.random.default_rng([int(fx['noise_seed']), 0x9e3779b9]).standard_exponential((n, 21, B, spec.vocab_top), dtype=np.float32),
                       np.float32(1e-30))
    eng = engine_s2(spec, weights, 4)
    for si, (tk, tp, T) in enumerate(json.loads(str(fx['settings']))):
        for graph in (False, True):
            c0, c1, c2, lg = eng.sample3(B, torch.full((B,), 7), n, precision=PRECISION_EXACT, top_k=tk, top_p=tp, temperature=T,
                                         noise=torch.from_numpy(noise), return_logits=True, use_graph=graph)
            assert np.abs(np_(lg)[fx['keep_steps']] - fx[f'logits_{si}']).max() <= LOGIT_TOL
            assert (np_(c0) == fx[f'codes0_{si}']).all() and (np_(c1) == fx[f'codes1_{si}']).all() and (np_(c2) == fx[f'codes2_{si}']).all()
        # SPLIT (fp32-accurate AR loop on the matrix cores): the same bar as EXACT
        from hqtransformer_amd._lib import PRECISION_SPLIT
        c0, c1, c2, lg = eng.sample3(B, torch.full((B,), 7), n, precision=PRECISION_SPLIT, top_k=tk, top_p=tp, temperature=T,
                                     noise=torch.from_numpy(noise), return_logits=True, use_graph=True)
        assert np.abs(np_(lg)[fx['keep_steps']] - fx[f'logits_{si}']).max() <= LOGIT_TOL
        assert (np_(c0) == fx[f'codes0_{si}']).all() and (np_(c1) == fx[f'codes1_{si}']).all() and (np_(c2) == fx[f'codes2_{si}']).all()
    eng.range_check()
    force = [torch.from_numpy(fx[f'codes{i}_0'].copy()) for i in range(3)]
    ex = eng.sample3(B, torch.full((B,), 7), 16, precision=PRECISION_EXACT, noise=torch.from_numpy(noise[:16]),
                     force=[f[:, :16] for f in force], return_logits=True, use_graph=False)
    for graph in (False, True):
        fa = eng.sample3(B, torch.full((B,), 7), 16, precision=PRECISION_FAST, noise=torch.from_numpy(noise[:16]),
                         force=[f[:, :16] for f in force], return_logits=True, use_graph=graph)
        err = (fa[3] - ex[3]).abs().max().item()
        gate(f'l3_tiny.fast_logits(graph={graph})', err, 0.12)


@pytest.mark.parametrize('name', ['g7_l3_tiny_cls_parallel.npz', 'g7_l3_tiny_cls_parallel_reduce.npz', 'g7_l3_tiny_cls_top2mid2bot.npz'])
def test_l3_other_decoding_types_vs_reference_fixture(name):
    """hqt_config.depth_decoding 1 / 2 / 3: HQTransformer 'parallel', 'parallel-reduce' (hqtransformer.py:105-157,526-551) and the causal
    21-sub-step head 'top2mid2bot' (:700-800) through
    hqt_sample_l3 against fixtures G7b generated from the reference: EXACT codes bit-exact and logits <= 2e-4, eager and graph; FAST
    teacher-forced logits inside the gate of the 'parallel-add' test."""
    fx = load(name)
    spec = Stage2Spec(**json.loads(str(fx['spec'])))
    weights = synth.stage2_weights(spec, int(fx['weight_seed']), 'fixture')
    B, n, cls = int(fx['B']), int(fx['n_steps']), int(fx['cond'])
    noise = np.maximum(np.random.default_rng([int(fx['noise_seed']), 0x9e3779b9]).standard_exponential((n, 21, B, spec.vocab_top), dtype=np.float32),
                       np.float32(1e-30))
    tk, tp, T = json.loads(str(fx['settings']))[0]
    eng = engine_s2(spec, weights, 4)
    for graph in (False, True):
        c0, c1, c2, lg = eng.sample3(B, torch.full((B,), cls), n, precision=PRECISION_EXACT, top_k=tk, top_p=tp, temperature=T,
                                     noise=torch.from_numpy(noise), return_logits=True, use_graph=graph)
        assert np.abs(np_(lg)[fx['keep_steps']] - fx['logits_0']).max() <= LOGIT_TOL
        assert (np_(c0) == fx['codes0_0']).all() and (np_(c1) == fx['codes1_0']).all() and (np_(c2) == fx['codes2_0']).all()
    force = [torch.from_numpy(fx[f'codes{i}_0'].copy()) for i in range(3)]
    ex = eng.sample3(B, torch.full((B,), cls), 16, precision=PRECISION_EXACT, noise=torch.from_numpy(noise[:16]),
                     force=[f[:, :16] for f in force], return_logits=True, use_graph=False)
    fa = eng.sample3(B, torch.full((B,), cls), 16, precision=PRECISION_FAST, noise=torch.from_numpy(noise[:16]),
                     force=[f[:, :16] for f in force], return_logits=True, use_graph=True)
    gate(f'l3_tiny.{spec.depth_decoding}.fast_logits', (fa[3] - ex[3]).abs().max().item(), 0.12)


def test_l3_wide_batch_vs_oracle():
    """B = 64 at D = 256: the third level runs 1024-row GEMMs (tiled MFMA path in FAST), levels 0 / 1 the streaming GEMMs."""
    spec = Stage2Spec(embed_dim=256, n_layers=1, n_heads=4, n_layers_depth=1, vocab_top=512, vocab_bot=512, vocab_txt=64,
                      ctx_len_img=64, ctx_len_txt=16, n_classes=10, cond=1, embedding=0, levels=3)
    weights = synth.stage2_weights(spec, 71, 'fixture')
    B, n = 64, 3
    noise = np.maximum(np.random.default_rng([72, 1]).standard_exponential((n, 21, B, 512), dtype=np.float32), np.float32(1e-30))
    cond = np.arange(B) % 10
    want = O.OracleStage2L3(spec, weights).sample(cond, B, n, noise, (None, 64, 32), (None, 0.9, None), (1.0, 0.9, 0.8), return_logits=True)
    eng = engine_s2(spec, weights, B)
    got = eng.sample3(B, torch.from_numpy(cond), n, precision=PRECISION_EXACT, top_k=(None, 64, 32), top_p=(None, 0.9, None),
                      temperature=(1.0, 0.9, 0.8), noise=torch.from_numpy(noise), return_logits=True, use_graph=False)
    assert np.abs(np_(got[3]) - want[3]).max() <= LOGIT_TOL
    assert all((np_(got[i]) == want[i]).all() for i in range(3))
    force = [torch.from_numpy(w) for w in want[:3]]
    for graph in (False, True):
        fa = eng.sample3(B, torch.from_numpy(cond), n, precision=PRECISION_FAST, top_k=(None, 64, 32), top_p=(None, 0.9, None),
                         temperature=(1.0, 0.9, 0.8), noise=torch.from_numpy(noise), force=force, return_logits=True, use_graph=graph)
        err = np.abs(np_(fa[3]) - want[3]).max()
        gate(f'l3_wide.fast_logits(graph={graph})', err, 0.12)


def test_l3_decode_vs_reference_fixture():
    fx = load('g8_l3_decode.npz')
    spec = Stage1Spec(**json.loads(str(fx['spec'])))
    weights = synth.stage1_weights(spec, int(fx['weight_seed']), 'fixture')
    eng = engine_s1(spec, weights, 2)
    ct, cm, cb = (torch.from_numpy(fx[k]) for k in ('code_t', 'code_m', 'code_b'))
    px = np_(eng.decode3([ct, cm, cb], precision=PRECISION_EXACT))
    assert np.abs(px - fx['pixels']).max() <= PIXEL_TOL
    assert np.abs(np_(eng.decode3([ct[:1], None, None])) - fx['pixels_top_only']).max() <= PIXEL_TOL
    assert np.abs(np_(eng.decode3([None, None, cb[:1]])) - fx['pixels_bot_only']).max() <= PIXEL_TOL
    # sampler layout: [B, n], [B, n, 4], [B, n, 16] with the rearranges folded into the lookup
    K = ct.shape[1]
    s0 = ct.reshape(2, K * K)
    s1 = cm.reshape(2, K, 2, K, 2).permute(0, 1, 3, 2, 4).reshape(2, K * K, 4)
    s2 = cb.reshape(2, K, 4, K, 4).permute(0, 1, 3, 2, 4).reshape(2, K * K, 16)
    assert (np_(eng.decode3([s0, s1, s2], seq_layout=True)) == px).all()
    fast = np_(eng.decode3([ct, cm, cb], precision=PRECISION_FAST))
    d = np.abs(fast - fx['pixels'])
    gate('level3_decode.fast_pixels.max', d.max(), 0.1)
    gate('level3_decode.fast_pixels.mean', d.mean(), 1e-2)


# --------------------------------------------------------------------------------------------- HQ-VAE encode side (hqt_encode)
def _synth_images(seed, B, R):
    r = np.random.default_rng([seed, 78])
    yy, xx = np.meshgrid(np.linspace(0, 1, R, dtype=np.float32), np.linspace(0, 1, R, dtype=np.float32), indexing='ij')
    img = np.zeros((B, 3, R, R), np.float32)
    for b in range(B):
        for c in range(3):
            for _ in range(4):
                fx_, fy_, ph = r.uniform(0.5, 6.0), r.uniform(0.5, 6.0), r.uniform(0, 2 * np.pi)
                img[b, c] += r.uniform(0.1, 0.4) * np.sin(2 * np.pi * (fx_ * xx + fy_ * yy) + ph).astype(np.float32)
    return np.clip(img + 0.05 * r.standard_normal(img.shape).astype(np.float32), -1.0, 1.0).astype(np.float32)


def _excess_distance(resid, emb, codes):
    """float64 d(z, e[code]) - min_n d(z, e[n]) per row: 0 where the code is the nearest one."""
    z = np.ascontiguousarray(resid.transpose(0, 2, 3, 1)).reshape(-1, resid.shape[1]).astype(np.float64)
    e = emb.astype(np.float64)
    d = (z ** 2).sum(1, keepdims=True) + (e ** 2).sum(1)[None] - 2 * z @ e.T
    return d[np.arange(len(z)), codes.reshape(-1)] - d.min(1)


def _codebooks(spec, weights):
    if spec.code_levels == 3:
        return [weights[f'quantizers.{l}.embedding'] for l in range(3)]
    return [weights['quantize_t.embedding'], weights['quantize_b.embedding']]


@pytest.mark.parametrize('name', ['g9_encode_64.npz', 'g9_encode_64_noinit.npz', 'g9_encode_64_l3.npz'])
def test_encode_exact_vs_reference_fixture(name):
    """hqt_encode in EXACT precision against the reference's own encode (tools/gen_golden_enc.py): bit-identical codes at every
    level (the fixtures sit >= 4e-4 from any argmin tie), quantiser inputs / straight-through outputs / reconstruction within
    1e-4, commitment terms within 1e-4 relative.  Covers the 4x4 stride-2 and the 3x3 conv_in, Downsample's one-sided padding,
    attention inside a `down` level, two and three code levels."""
    fx = load(name)
    spec = Stage1Spec(**json.loads(str(fx['spec'])))
    weights = synth.stage1_weights(spec, int(fx['weight_seed']), 'fixture', encoder=True)
    eng = engine_s1(spec, weights, int(fx['B']))
    assert eng.has_encoder
    L = 3 if spec.code_levels == 3 else 2
    o = eng.encode(torch.from_numpy(fx['pixels']), precision=PRECISION_EXACT, want_quant=True, want_resid=True, want_recon=True, want_diff=True)
    for l in range(L):
        assert np.abs(np_(o['resid'][l]) - fx[f'resid_{l}']).max() <= PIXEL_TOL, l
        assert np.array_equal(np_(o['codes'][l]).reshape(-1), fx[f'code_{l}'].reshape(-1)), l
        assert abs(float(o['diff'][l]) - float(fx[f'diff_{l}'])) <= 1e-4 * float(fx[f'diff_{l}']), l
        if f'quant_{l}' in fx:
            assert np.abs(np_(o['quant'][l]) - fx[f'quant_{l}']).max() <= PIXEL_TOL, l
    if 'recon' in fx:
        assert np.abs(np_(o['recon']) - fx['recon']).max() <= PIXEL_TOL
    # decode(encode(x)) = the reference's forward() in eval mode
    if L == 3:
        rec = eng.decode3(o['codes'], precision=PRECISION_EXACT)
    else:
        rec = eng.decode(o['codes'][0], o['codes'][1], precision=PRECISION_EXACT)
    assert np.abs(np_(rec) - fx['reconstruction']).max() <= PIXEL_TOL
    # a decode-only handle refuses to encode, loudly
    dec_only = engine_s1(spec, synth.stage1_weights(spec, int(fx['weight_seed']), 'fixture'), 1)
    assert not dec_only.has_encoder
    with pytest.raises(Exception, match='encoder'):
        dec_only.encode(torch.from_numpy(fx['pixels'][:1]))


def test_encode_wide_config_exact_and_fast_vs_oracle():
    """A config wide enough for the MFMA kernels (64..128 channels; HQT_FORCE_TILE128 makes the dispatcher pick the LDS-DMA kernel
    with its stride-2 taps for Downsample).  EXACT: codes equal the oracle's except where the oracle's own float64 distance gap is
    within 1e-4 (argmin ties under a different fp32 summation order), feature map within 1e-4.  FAST (bf16 convolutions,
    fp32 distance GEMM): every chosen code is the nearest one for the device's own quantiser input (up to 1e-4 ties), the
    feature map within 0.15 (2 % of its range) of the fp32 one, >= 60 % of the codes identical to the fp32 choice."""
    import os
    spec = Stage1Spec(ch=64, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[16], resolution=64, z_channels=64,
                      embed_dim=32, n_embed=256)
    weights = synth.stage1_weights(spec, 81, 'fixture', encoder=True)
    x = _synth_images(82, 5, 64)
    want = O.OracleStage1(spec, weights).encode(x)
    cbs = _codebooks(spec, weights)
    os.environ['HQT_FORCE_TILE128'] = '1'
    try:
        eng = engine_s1(spec, weights, 5)
        ex = eng.encode(torch.from_numpy(x), precision=PRECISION_EXACT, want_resid=True)
        fa = eng.encode(torch.from_numpy(x), precision=PRECISION_FAST, want_resid=True)
    finally:
        del os.environ['HQT_FORCE_TILE128']
    assert np.abs(np_(ex['resid'][0]) - want['resid'][0]).max() <= PIXEL_TOL
    for l in range(2):
        got = np_(ex['codes'][l])
        bad = got != want['codes'][l]
        if bad.any():                           # only allowed at ties of the oracle's own distances
            assert l == 0 and _excess_distance(want['resid'][l], cbs[l], got)[bad.reshape(-1)].max() <= 1e-4
    for l in range(2):
        resid, codes = np_(fa['resid'][l]), np_(fa['codes'][l])
        assert _excess_distance(resid, cbs[l], codes).max() <= 1e-4, l
    gate('encode_wide.fast_feature_map', np.abs(np_(fa['resid'][0]) - want['resid'][0]).max(), 0.08)
    agree = np.mean(np_(fa['codes'][0]) == want['codes'][0])
    gate('encode_wide.fast_code_agreement', agree, 0.95, '>=')


def test_encode_batch_chunking_is_batch_invariant():
    """More images than one conv-stack chunk (64): codes of image i do not depend on its position in the batch, EXACT and FAST."""
    fx = load('g9_encode_64.npz')
    spec = Stage1Spec(**json.loads(str(fx['spec'])))
    weights = synth.stage1_weights(spec, int(fx['weight_seed']), 'fixture', encoder=True)
    x = np.concatenate([_synth_images(91, 67, 64), fx['pixels']])
    eng = engine_s1(spec, weights, 70)
    small = engine_s1(spec, weights, 3)
    for prec in (PRECISION_EXACT, PRECISION_FAST):
        big = eng.encode(torch.from_numpy(x), precision=prec)
        ref = small.encode(torch.from_numpy(x[67:]), precision=prec)
        for l in range(2):
            assert np.array_equal(np_(big['codes'][l])[67:], np_(ref['codes'][l])), (prec, l)
    assert np.array_equal(np_(big['codes'][0])[67:].reshape(-1), np_(ref['codes'][0]).reshape(-1))
    exact = eng.encode(torch.from_numpy(x), precision=PRECISION_EXACT)
    assert np.array_equal(np_(exact['codes'][1])[67:].reshape(-1), fx['code_1'].reshape(-1))


def test_encode_imagenet_geometry_vs_oracle():
    """The released ImageNet stage-1 shape (256 x 256, ch 128, ch_mult [1, 2, 4, 4], 4x4 stride-2 conv_in, 8192 codes of 1024 / 256
    dims) on 2 images: EXACT feature map within 2e-4 of the oracle and codes equal up to fp32 argmin ties (oracle's own float64
    gap <= 1e-4); FAST (the halo / LDS-DMA MFMA kernels incl. the stride-2 Downsample taps on their real shapes): feature map
    within 3 % of its range, codes the exact nearest ones for the device's own feature map, decode(encode(x)) finite."""
    import os
    from hqtransformer_amd.config import load_config
    from hqtransformer_amd.spec import stage1_spec_from_config
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = stage1_spec_from_config(load_config(os.path.join(root, 'configs', 'imagenet-12l.yaml')))
    assert spec.resolution == 256 and spec.n_embed == 8192
    weights = synth.stage1_weights(spec, 95, 'fixture', encoder=True)
    x = _synth_images(96, 2, 256)
    want = O.OracleStage1(spec, weights).encode(x)
    cbs = _codebooks(spec, weights)
    eng = engine_s1(spec, weights, 2)
    ex = eng.encode(torch.from_numpy(x), precision=PRECISION_EXACT, want_resid=True)
    assert np.abs(np_(ex['resid'][0]) - want['resid'][0]).max() <= 2e-4
    top_same = np.array_equal(np_(ex['codes'][0]), want['codes'][0])
    for l in range(2):
        got = np_(ex['codes'][l])
        bad = (got != want['codes'][l]).reshape(-1)
        if bad.any() and (l == 0 or top_same):
            assert _excess_distance(want['resid'][l], cbs[l], got)[bad].max() <= 1e-4, l
    fa = eng.encode(torch.from_numpy(x), precision=PRECISION_FAST, want_resid=True)
    rng = float(np.abs(want['resid'][0]).max())
    assert np.abs(np_(fa['resid'][0]) - want['resid'][0]).max() <= 0.03 * rng, (np.abs(np_(fa['resid'][0]) - want['resid'][0]).max(), rng)
    for l in range(2):
        assert _excess_distance(np_(fa['resid'][l]), cbs[l], np_(fa['codes'][l])).max() <= 1e-3, l
    rec = eng.decode(fa['codes'][0], fa['codes'][1], precision=PRECISION_FAST)
    assert bool(torch.isfinite(rec).all())


def test_decode_five_level_geometry_vs_oracle():
    """BASELINE configs[3] geometry (the reference's 5-level pattern ch_mult [1, 2, 4, 4, 4], attention at 32 x 32 = 1024 tokens,
    bottom grid 16 x 16 here instead of 32 x 32, 32 base channels so the CPU oracle finishes in seconds): EXACT pixels within
    1e-4 of the oracle, FAST within the bf16 budget of test_decode_fast_tolerance.  tools/bench_decoder.py runs the full
    1024 x 1024 / 128-channel size (oracle-free: determinism + finiteness + throughput)."""
    spec = Stage1Spec(ch=32, ch_mult=[1, 2, 4, 4, 4], num_res_blocks=1, attn_resolutions=[32], resolution=512, z_channels=64,
                      embed_dim=32, n_embed=256)
    assert spec.z_res == 16
    weights = synth.stage1_weights(spec, 101, 'fixture')
    r = np.random.default_rng(102)
    ct, cb = r.integers(0, 256, (1, 8, 8)), r.integers(0, 256, (1, 16, 16))
    want = O.OracleStage1(spec, weights).decode_code(ct, cb)
    eng = engine_s1(spec, weights, 1)
    exact = np_(eng.decode(torch.from_numpy(ct), torch.from_numpy(cb), precision=PRECISION_EXACT))
    assert exact.shape == (1, 3, 512, 512)
    assert np.abs(exact - want).max() <= PIXEL_TOL, np.abs(exact - want).max()
    fast = np_(eng.decode(torch.from_numpy(ct), torch.from_numpy(cb), precision=PRECISION_FAST))
    d = np.abs(fast - want)
    gate('five_level_decode.fast_pixels.max', d.max(), 0.1 * max(1.0, float(np.abs(want).max()) / 5.0))
    gate('five_level_decode.fast_pixels.mean', d.mean(), 1e-2)


def test_fast_vs_exact_at_the_benchmark_model_size():
    """The full ImageNet 12-layer / D = 1536 stage 2 (random-init 'bench' weights, as the harness uses): FAST teacher-forced on
    EXACT's codes.  Logits (std ~0.8) within 0.1, KL(exact || fast) <= 1e-3 nats per draw, >= 98 % identical draws under the
    same noise (measured: 0.0275 / 1.4e-5 / 99.7 %, tools/fast_ar_error.py).  Runs the real streaming-GEMM variants, deferred
    LayerNorm, single-key shortcut and fused embedding at their production shapes."""
    import os
    from hqtransformer_amd.config import load_config
    from hqtransformer_amd.spec import stage2_spec_from_config
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s2 = stage2_spec_from_config(load_config(os.path.join(root, 'configs', 'imagenet-12l.yaml')))
    eng = engine_s2(s2, synth.stage2_weights(s2, 0, 'bench'), 8)
    B, n, V = 8, 4, s2.vocab_top
    noise = torch.from_numpy(synth.exp_noise(4, n, B, V))
    cond = torch.from_numpy(synth.class_ids(5, B, s2.n_classes))
    ct, cb, lg_e = eng.sample(B, cond, n, precision=PRECISION_EXACT, noise=noise, return_logits=True, use_graph=False)
    _, _, lg_f = eng.sample(B, cond, n, precision=PRECISION_FAST, noise=noise, force_top=ct, force_bot=cb, return_logits=True, use_graph=True)
    le, lf = lg_e.double(), lg_f.double()
    gate('benchmark_model.fast_logits', (le - lf).abs().max(), 0.06)             # measured 0.0275 (tools/fast_ar_error.py)
    pe, pf = torch.softmax(le, -1), torch.softmax(lf, -1)
    kl = (pe * (torch.log(pe.clamp_min(1e-300)) - torch.log(pf.clamp_min(1e-300)))).sum(-1)
    gate('benchmark_model.fast_kl_max', kl.max(), 1e-4)                           # measured 1.4e-5 nats per draw (mean)
    q = noise.to(le.device).double()
    gate('benchmark_model.fast_identical_draws', (torch.argmax(pe / q, -1) == torch.argmax(pf / q, -1)).double().mean(), 0.99, '>=')   # measured 0.997


# ----------------------------------------------------------------------------------------- the benchmark's own shapes
def test_full_benchmark_model_exact_vs_oracle():
    """The whole ImageNet model of the benchmark (12 body + 4 depth layers, D = 1536, 24 heads, V = 8192, 530.8 M parameters,
    random-init 'bench' weights as the harness uses) at the benchmark's batch, B = 64, two top positions, against the CPU oracle
    (about 2 s per position on the GPU box's host): EXACT sampled codes bit-exact under the same noise and fp32 logits within
    2e-4; FAST (what bench.py times: streaming GEMMs, deferred LayerNorm, single-key shortcut, fused embedding, one hipGraph)
    teacher-forced on the oracle's codes within the gate of test_fast_vs_exact_at_the_benchmark_model_size."""
    import os
    from hqtransformer_amd.config import load_config
    from hqtransformer_amd.spec import stage2_spec_from_config
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s2 = stage2_spec_from_config(load_config(os.path.join(root, 'configs', 'imagenet-12l.yaml')))
    assert (s2.n_layers, s2.n_layers_depth, s2.embed_dim, s2.vocab_top) == (12, 4, 1536, 8192)
    weights = synth.stage2_weights(s2, 0, 'bench')
    B, n = 64, 2
    noise = synth.exp_noise(11, n, B, s2.vocab_top)
    cond = synth.class_ids(12, B, s2.n_classes)
    want = O.OracleStage2(s2, weights).sample(cond, B, n, noise, return_logits=True)
    eng = engine_s2(s2, weights, B, 8)
    ct, cb, lg = eng.sample(B, torch.from_numpy(cond), n, precision=PRECISION_EXACT, noise=torch.from_numpy(noise), return_logits=True,
                            use_graph=False)
    err = np.abs(np_(lg) - want[2]).max()
    assert err <= LOGIT_TOL, f'EXACT logits of the full model differ from the oracle by {err}'
    assert (np_(ct) == want[0]).all() and (np_(cb) == want[1]).all(), 'EXACT codes of the full model differ from the oracle'
    _, _, lf = eng.sample(B, torch.from_numpy(cond), n, precision=PRECISION_FAST, noise=torch.from_numpy(noise),
                          force_top=torch.from_numpy(want[0]), force_bot=torch.from_numpy(want[1]), return_logits=True, use_graph=True)
    gate('benchmark_model_B64.fast_logits_vs_oracle', np.abs(np_(lf) - want[2]).max(), 0.06)
    # SPLIT (round 4): fp32-accurate AR loop on the matrix cores -- the same bar as EXACT: codes bit-identical, logits <= 2e-4
    from hqtransformer_amd._lib import PRECISION_SPLIT
    for graph in (False, True):
        st, sb, ls = eng.sample(B, torch.from_numpy(cond), n, precision=PRECISION_SPLIT, noise=torch.from_numpy(noise), return_logits=True, use_graph=graph)
        serr = np.abs(np_(ls) - want[2]).max()
        assert serr <= LOGIT_TOL, f'SPLIT logits of the full model differ from the oracle by {serr}'
        assert (np_(st) == want[0]).all() and (np_(sb) == want[1]).all(), 'SPLIT codes of the full model differ from the oracle'
    eng.range_check()
    print(f'full 12+4-layer model, B = 64: EXACT logits vs oracle {err:.2e} (std {want[2].std():.3f}), SPLIT {serr:.2e}, FAST {np.abs(np_(lf) - want[2]).max():.4f}')


def test_text_prefill_at_the_cc15m_shape_vs_oracle():
    """BASELINE configs[4]: the 64-token causal prompt prefill at its real shape -- D = 1536, 24 heads, ctx_len_txt = 64, B = 64, i.e.
    4096 rows through the tiled MFMA kernels (FAST) / the fp32 tile kernel (EXACT) with the fused QKV split + KV-cache append and
    the causal attention over 64 queries -- one body + one depth layer (the oracle's cost), then two cached positions.  EXACT:
    codes bit-exact, logits within 2e-4; FAST: teacher-forced logits within the bf16 gate, eager and graph."""
    spec = Stage2Spec(embed_dim=1536, n_layers=1, n_heads=24, n_layers_depth=1, vocab_top=1024, vocab_bot=1024, vocab_txt=16384,
                      ctx_len_img=64, ctx_len_txt=64, n_classes=0, cond=2, embedding=0)
    weights = synth.stage2_weights(spec, 301, 'fixture')
    B, n = 64, 3
    noise = synth.exp_noise(302, n, B, spec.vocab_top)
    txt = synth.text_ids(303, B, spec.ctx_len_txt, spec.vocab_txt)
    want = O.OracleStage2(spec, weights).sample(txt, B, n, noise, return_logits=True)
    eng = engine_s2(spec, weights, B, 8)
    ct, cb, lg = eng.sample(B, torch.from_numpy(txt), n, precision=PRECISION_EXACT, noise=torch.from_numpy(noise), return_logits=True,
                            use_graph=False)
    assert np.abs(np_(lg) - want[2]).max() <= LOGIT_TOL
    assert (np_(ct) == want[0]).all() and (np_(cb) == want[1]).all()
    from hqtransformer_amd._lib import PRECISION_SPLIT           # the 4096-row prefill + cached positions, fp32-accurate on the matrix cores
    st, sb, ls = eng.sample(B, torch.from_numpy(txt), n, precision=PRECISION_SPLIT, noise=torch.from_numpy(noise), return_logits=True, use_graph=True)
    assert np.abs(np_(ls) - want[2]).max() <= LOGIT_TOL
    assert (np_(st) == want[0]).all() and (np_(sb) == want[1]).all()
    eng.range_check()
    ft, fb = torch.from_numpy(want[0]), torch.from_numpy(want[1])
    for graph in (False, True):
        _, _, lf = eng.sample(B, torch.from_numpy(txt), n, precision=PRECISION_FAST, noise=torch.from_numpy(noise), force_top=ft, force_bot=fb,
                              return_logits=True, use_graph=graph)
        gate(f'cc15m_prefill_shape.fast_logits(graph={graph})', np.abs(np_(lf) - want[2]).max(), 0.1)


@pytest.mark.parametrize('name', ['imagenet-24l', 'imagenet-42l', 'ffhq-24l'])
def test_other_released_shapes_head_geometry_vs_oracle(name):
    """The released configs beyond the benchmark's (configs/*.yaml, state-dict shapes pinned to the reference by fixture G11):
    24 and 42 layers with the 6-layer depth head (hparams_dec), FFHQ's D = 1024 / 16 heads / 'reduce' embedding / unconditional
    sos.  Shape-true in everything but the number of body layers (2, the oracle's cost): real width, heads, depth-head length,
    vocabulary 8192, B = 64.  EXACT codes bit-exact and logits <= 2e-4; FAST teacher-forced within the bf16 gate."""
    import dataclasses
    import os
    from hqtransformer_amd.config import load_config
    from hqtransformer_amd.spec import stage2_spec_from_config
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    full = stage2_spec_from_config(load_config(os.path.join(root, 'configs', name + '.yaml')))
    spec = dataclasses.replace(full, n_layers=2, ctx_len_img=64)
    assert spec.n_layers_depth == (6 if name == 'imagenet-42l' else 4) and spec.vocab_top == 8192
    weights = synth.stage2_weights(spec, 401, 'fixture')
    B, n = 64, 2
    noise = synth.exp_noise(402, n, B, spec.vocab_top)
    cond = synth.class_ids(403, B, spec.n_classes) if spec.cond == 1 else None
    want = O.OracleStage2(spec, weights).sample(cond, B, n, noise, (2048, None), (1.0, None), (0.95, 1.0), return_logits=True)
    eng = engine_s2(spec, weights, B, 8)
    tc = None if cond is None else torch.from_numpy(cond)
    ct, cb, lg = eng.sample(B, tc, n, precision=PRECISION_EXACT, top_k=(2048, None), top_p=(1.0, None), temperature=(0.95, 1.0),
                            noise=torch.from_numpy(noise), return_logits=True, use_graph=False)
    assert np.abs(np_(lg) - want[2]).max() <= LOGIT_TOL
    assert (np_(ct) == want[0]).all() and (np_(cb) == want[1]).all()
    for graph in (False, True):
        _, _, lf = eng.sample(B, tc, n, precision=PRECISION_FAST, top_k=(2048, None), top_p=(1.0, None), temperature=(0.95, 1.0),
                              noise=torch.from_numpy(noise), force_top=torch.from_numpy(want[0]), force_bot=torch.from_numpy(want[1]),
                              return_logits=True, use_graph=graph)
        gate(f'{name}.head_geometry.fast_logits(graph={graph})', np.abs(np_(lf) - want[2]).max(), 0.12)
