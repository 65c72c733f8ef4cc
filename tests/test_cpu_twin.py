"""The hqt_cpu_* twins (include/hqt_cpu.h, oracle/cpu/hqt_cpu.cpp: the C++ / OpenMP restatement of the reference's CPU path that bench.py
times as `cpu_baseline`) against the fixtures generated from the reference itself (tests/golden, tools/gen_golden.py) -- the same bars as
the numpy oracle: sampled code sequences bit-exact, fp32 logits <= 2e-4, pixels <= 1e-4 -- and against the numpy oracle on other seeded
shapes.  CPU only; the twin is test infrastructure and a baseline, never a fallback of the product."""
import json
import os
import re

import numpy as np
import pytest

from hqtransformer_amd import synth
from hqtransformer_amd.spec import Stage1Spec, Stage2Spec
from oracle import hqt_cpu
from oracle import hqt_oracle as O
from tests.helpers import load, philox_exp_noise, stage1_from_fixture, stage2_from_fixture

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOGIT_TOL = 2e-4
PIXEL_TOL = 1e-4


@pytest.fixture(scope='module', autouse=True)
def built():
    hqt_cpu.build()


def test_library_exports_every_symbol_the_header_declares():
    hdr = open(os.path.join(ROOT, 'include', 'hqt_cpu.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(hqt_cpu_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(hqt_cpu.SYMBOLS), declared ^ set(hqt_cpu.SYMBOLS)
    lib = hqt_cpu.load()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.hqt_cpu_isa().decode() in ('avx512', 'avx2')
    assert lib.hqt_cpu_create(None, 0, None) == -1 and b'null' in lib.hqt_cpu_last_error()


@pytest.mark.parametrize('si', [0, 1, 2])
def test_tiny_cls_sampling_bit_exact_vs_reference_fixture(si):
    fx = load('g4_tiny_cls.npz')
    spec, weights = stage2_from_fixture(fx)
    twin = hqt_cpu.CpuTwin(spec, None, weights)
    tk, tp, T = json.loads(str(fx['settings']))[si]
    B, n = int(fx['B']), int(fx['n_steps'])
    noise = synth.exp_noise(int(fx['noise_seed']), n, B, spec.vocab_top)
    ct, cb, lg = twin.sample(np.full(B, 7), B, n, noise, tk, tp, T, return_logits=True)
    assert (ct == fx[f'codes_top_{si}']).all() and (cb == fx[f'codes_bot_{si}']).all()
    scale = np.array([T[0]] + [T[1]] * 4, np.float32)[None, :, None, None]
    # the twin reports raw (pre-temperature) logits like hqt_sample; the fixture holds them divided by T
    np.testing.assert_allclose(lg[fx['keep_steps']] / scale, fx[f'logits_{si}'], atol=LOGIT_TOL, rtol=0)


def test_given_top_code_vs_reference_fixture():
    fx = load('g4_tiny_cls.npz')
    spec, weights = stage2_from_fixture(fx)
    twin = hqt_cpu.CpuTwin(spec, None, weights)
    B = int(fx['B'])
    noise = synth.exp_noise(int(fx['noise_seed']), 64, B, spec.vocab_top)[:8]
    ct, cb, lg = twin.sample(np.full(B, 3), B, 8, noise, force_top=fx['given_top'][:, :8], return_logits=True)
    assert (cb == fx['given_codes_bot']).all()
    np.testing.assert_allclose(lg, fx['given_logits'], atol=LOGIT_TOL, rtol=0)


def test_reduce_embedding_unconditional_vs_reference_fixture():
    fx = load('g3_tiny_reduce_uncond.npz')
    spec, weights = stage2_from_fixture(fx)
    twin = hqt_cpu.CpuTwin(spec, None, weights)
    B, n = int(fx['B']), int(fx['n_steps'])
    noise = synth.exp_noise(int(fx['noise_seed']), n, B, spec.vocab_top)
    k, p, T = int(fx['top_k']), float(fx['top_p']), [float(t) for t in fx['temps']]
    ct, cb, lg = twin.sample(None, B, n, noise, (k, k), (p, p), T, return_logits=True)
    assert (ct == fx['codes_top']).all() and (cb == fx['codes_bot']).all()
    scale = np.array([T[0]] + [T[1]] * 4, np.float32)[None, :, None, None]
    np.testing.assert_allclose(lg[fx['keep_steps']] / scale, fx['logits'], atol=LOGIT_TOL, rtol=0)


def test_text_prefill_vs_reference_fixture():
    fx = load('g3_tiny_txt.npz')
    spec, weights = stage2_from_fixture(fx)
    twin = hqt_cpu.CpuTwin(spec, None, weights)
    B, n = int(fx['B']), int(fx['n_steps'])
    noise = synth.exp_noise(int(fx['noise_seed']), n, B, spec.vocab_top)
    txt = synth.text_ids(int(fx['text_seed']), B, spec.ctx_len_txt, spec.vocab_txt)
    ct, cb, lg = twin.sample(txt, B, n, noise, return_logits=True)
    assert (ct == fx['codes_top']).all() and (cb == fx['codes_bot']).all()
    np.testing.assert_allclose(lg[fx['keep_steps']], fx['logits'], atol=LOGIT_TOL, rtol=0)


@pytest.mark.parametrize('name', ['g5_decode_64.npz', 'g5_decode_256.npz'])
def test_decode_pixels_vs_reference_fixture(name):
    fx = load(name)
    spec, weights = stage1_from_fixture(fx)
    twin = hqt_cpu.CpuTwin(None, spec, None, weights)
    px = twin.decode_code(fx['code_t'], fx['code_b'])
    np.testing.assert_allclose(px, fx['pixels'], atol=PIXEL_TOL, rtol=0)
    if 'pixels_top_only' in fx.files:                     # a missing level contributes a zero quant (generator.py:328-358)
        np.testing.assert_allclose(twin.decode_code(fx['code_t'][:1], None), fx['pixels_top_only'], atol=PIXEL_TOL, rtol=0)
        np.testing.assert_allclose(twin.decode_code(None, fx['code_b'][:1]), fx['pixels_bot_only'], atol=PIXEL_TOL, rtol=0)


def test_sampler_layout_decode_and_clamp_vs_oracle():
    """hqt_cpu_decode_seq folds 'B (H W) -> B H W' / 'B (H W) (kh kw) -> B (H kh) (W kw)' (sampling_hqmodel.py:119-120) into the lookup;
    clamp01 = clamp(0.5 x + 0.5, 0, 1) (measure_throughput/__main__.py:113)."""
    fx = load('g5_decode_64.npz')
    spec, weights = stage1_from_fixture(fx)
    twin = hqt_cpu.CpuTwin(None, spec, None, weights)
    rng = np.random.default_rng(5)
    r = spec.z_res
    ct = rng.integers(0, spec.n_embed, (3, (r // 2) ** 2))
    cb = rng.integers(0, spec.n_embed, (3, (r // 2) ** 2, 4))
    gt, gb = O.rearrange_codes(ct, cb, r // 2)
    want = O.postprocess(O.OracleStage1(spec, weights).decode_code(gt, gb))
    got = twin.decode_code(ct, cb, clamp01=True, seq_layout=True)
    np.testing.assert_allclose(got, want, atol=PIXEL_TOL, rtol=0)


def test_wider_model_and_philox_noise_vs_oracle():
    """A shape no fixture holds (D = 256, 8 heads, 2 + 2 layers, V = 1024, B = 6, ragged against every slice size), quality-mode sampler
    settings, and the in-library Philox stream: the twin with noise = NULL must draw what the oracle draws from the restated stream."""
    spec = Stage2Spec(embed_dim=256, n_layers=2, n_heads=8, n_layers_depth=2, vocab_top=1024, vocab_bot=1024, vocab_txt=64,
                      ctx_len_img=64, ctx_len_txt=16, n_classes=100, cond=1, embedding=0)
    weights = synth.stage2_weights(spec, 77, 'fixture')
    B, n = 6, 5
    cond = (np.arange(B) * 13) % spec.n_classes
    noise = philox_exp_noise([991] * B, [40 + i for i in range(B)], n, spec.vocab_top)
    want = O.OracleStage2(spec, weights).sample(cond, B, n, noise, (200, 300), (0.95, None), (0.9, 1.0), return_logits=True)
    for threads in (0, 3):
        twin = hqt_cpu.CpuTwin(spec, None, weights, threads=threads)
        ct, cb, lg = twin.sample(cond, B, n, None, (200, 300), (0.95, None), (0.9, 1.0), return_logits=True, seed=991, sample_offset=40)
        assert (ct == want[0]).all() and (cb == want[1]).all()
        assert np.abs(lg - want[2]).max() <= LOGIT_TOL
        assert twin.last_seconds > 0 and twin.threads >= 1


def test_three_levels_are_refused():
    spec = Stage1Spec(ch=32, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[16], resolution=64, z_channels=32, embed_dim=16, n_embed=64, code_levels=3)
    with pytest.raises(RuntimeError, match='two-level'):
        hqt_cpu.CpuTwin(None, spec, None, {})
