"""Pins the CPU oracle against outputs of the reference itself (tests/golden/*.npz, written by
tools/gen_golden.py in the build container).  Bars: sampled code sequences bit-exact; fp32 logits and
pixels within the tolerances written below (both sides are fp32, summation order differs)."""
import json

import numpy as np
import pytest

from hqtransformer_amd import synth
from hqtransformer_amd.spec import stage1_param_shapes, stage2_param_shapes
from oracle import hqt_oracle as O
from tests.helpers import load, oracle_stage1, oracle_stage2

LOGIT_TOL = 2e-4      # abs, fp32 logits of std ~3 after up to 8 blocks
PIXEL_TOL = 1e-4      # abs, north_star's pixel tolerance


def test_sampler_kats():
    fx = load('g1_sampler.npz')
    cases = json.loads(str(fx['cases']))
    for ci, (k, p, T) in enumerate(cases):
        idx, probs = O.sample_filtered(fx['logits'], fx['noise'], T, k, p)
        ref = fx[f'probs_{ci}']
        differs = (probs == 0) != (ref == 0)
        # the kept set is decided by `cum >= p` on fp32 prefix sums: softmax results that differ from
        # torch's by one ulp can move the cut by a few extreme-tail tokens (SURVEY.md §7 "Top-p numerics")
        assert np.where(differs, np.maximum(probs, ref), 0).sum(-1).max() < 2e-6, f'case {ci}: kept set differs'
        np.testing.assert_allclose(np.where(differs, 0, probs), np.where(differs, 0, ref), rtol=4e-6, atol=1e-9)
        assert (idx == fx[f'index_{ci}']).all(), f'case {ci}'


def test_multinomial_is_argmax_p_over_q():
    fx = load('g1_sampler.npz')
    assert (np.argmax(fx['mn_probs'] / fx['mn_noise'], -1) == fx['mn_index']).all()


def test_param_shapes_match_reference_state_dict():
    for name in ('g4_tiny_cls.npz', 'g3_tiny_reduce_uncond.npz', 'g3_tiny_txt.npz'):
        fx = load(name)
        spec, _, _ = oracle_stage2(fx)
        ref = {k: tuple(v) for k, v in json.loads(str(fx['param_shapes'])).items()}
        assert dict(stage2_param_shapes(spec)) == ref
    for name in ('g5_decode_64.npz', 'g5_decode_256.npz'):
        fx = load(name)
        spec, _, _ = oracle_stage1(fx)
        ref = {k: tuple(v) for k, v in json.loads(str(fx['param_shapes'])).items()}
        assert dict(stage1_param_shapes(spec)) == ref


@pytest.mark.parametrize('si', [0, 1, 2])
def test_tiny_cls_sampling_bit_exact(si):
    fx = load('g4_tiny_cls.npz')
    spec, _, orc = oracle_stage2(fx)
    tk, tp, T = json.loads(str(fx['settings']))[si]
    B, n = int(fx['B']), int(fx['n_steps'])
    noise = synth.exp_noise(int(fx['noise_seed']), n, B, spec.vocab_top)
    ct, cb, lg = orc.sample(np.full(B, 7), B, n, noise, tk, tp, T, return_logits=True)
    assert float(fx[f'margin_{si}']) > 1.00005          # fixture is well-conditioned for bit-exactness
    assert (ct == fx[f'codes_top_{si}']).all()
    assert (cb == fx[f'codes_bot_{si}']).all()
    scale = np.array([T[0]] + [T[1]] * 4, np.float32)[None, :, None, None]
    np.testing.assert_allclose(lg[fx['keep_steps']] / scale, fx[f'logits_{si}'], atol=LOGIT_TOL, rtol=0)


def test_tiny_cls_given_top_code():
    fx = load('g4_tiny_cls.npz')
    spec, _, orc = oracle_stage2(fx)
    B = int(fx['B'])
    noise = synth.exp_noise(int(fx['noise_seed']), 64, B, spec.vocab_top)[:8]
    ct, cb, lg = orc.sample(np.full(B, 3), B, 8, noise, force_top=fx['given_top'], return_logits=True)
    assert (cb == fx['given_codes_bot']).all()
    np.testing.assert_allclose(lg, fx['given_logits'], atol=LOGIT_TOL, rtol=0)


def test_tiny_reduce_uncond_bit_exact():
    fx = load('g3_tiny_reduce_uncond.npz')
    spec, _, orc = oracle_stage2(fx)
    B, n = int(fx['B']), int(fx['n_steps'])
    noise = synth.exp_noise(int(fx['noise_seed']), n, B, spec.vocab_top)
    k, p, T = int(fx['top_k']), float(fx['top_p']), [float(t) for t in fx['temps']]
    ct, cb, lg = orc.sample(None, B, n, noise, (k, k), (p, p), T, return_logits=True)
    assert (ct == fx['codes_top']).all() and (cb == fx['codes_bot']).all()
    scale = np.array([T[0]] + [T[1]] * 4, np.float32)[None, :, None, None]
    np.testing.assert_allclose(lg[fx['keep_steps']] / scale, fx['logits'], atol=LOGIT_TOL, rtol=0)


def test_tiny_txt_prefill_bit_exact():
    fx = load('g3_tiny_txt.npz')
    spec, _, orc = oracle_stage2(fx)
    B, n = int(fx['B']), int(fx['n_steps'])
    noise = synth.exp_noise(int(fx['noise_seed']), n, B, spec.vocab_top)
    txt = synth.text_ids(int(fx['text_seed']), B, spec.ctx_len_txt, spec.vocab_txt)
    ct, cb, lg = orc.sample(txt, B, n, noise, return_logits=True)
    assert (ct == fx['codes_top']).all() and (cb == fx['codes_bot']).all()
    np.testing.assert_allclose(lg[fx['keep_steps']], fx['logits'], atol=LOGIT_TOL, rtol=0)


def test_decode_64_pixels_and_intermediates():
    fx = load('g5_decode_64.npz')
    spec, _, orc = oracle_stage1(fx)
    px = orc.decode_code(fx['code_t'], fx['code_b'])
    np.testing.assert_allclose(px, fx['pixels'], atol=PIXEL_TOL, rtol=0)
    np.testing.assert_allclose(orc.decode_code(fx['code_t'][:1], None), fx['pixels_top_only'], atol=PIXEL_TOL, rtol=0)
    np.testing.assert_allclose(orc.decode_code(None, fx['code_b'][:1]), fx['pixels_bot_only'], atol=PIXEL_TOL, rtol=0)
    w = orc.w
    qt = w['quantize_t.embedding'][fx['code_t'][:1]].transpose(0, 3, 1, 2)
    qb = w['quantize_b.embedding'][fx['code_b'][:1]].transpose(0, 3, 1, 2)
    z = orc._conv('post_quant_conv_b', np.concatenate([O.pixel_shuffle2(qt), qb], 1))
    np.testing.assert_allclose(z, fx['z'], atol=1e-5, rtol=0)
    h0 = orc._conv('decoder.conv_in', z)
    np.testing.assert_allclose(h0, fx['conv_in'], atol=2e-5, rtol=0)
    h1 = orc._resblock('decoder.mid.block_1', h0)
    np.testing.assert_allclose(h1, fx['mid_block_1'], atol=5e-5, rtol=0)
    np.testing.assert_allclose(orc._attnblock('decoder.mid.attn_1', h1), fx['mid_attn_1'], atol=5e-5, rtol=0)


def test_decode_256_pixels():
    fx = load('g5_decode_256.npz')
    spec, _, orc = oracle_stage1(fx)
    px = orc.decode_code(fx['code_t'], fx['code_b'])
    np.testing.assert_allclose(px, fx['pixels'], atol=PIXEL_TOL, rtol=0)


def test_index_maps():
    fx = load('g6_index_maps.npz')
    gt, gb = O.rearrange_codes(fx['codes_top'], fx['codes_bot'], 8)
    assert (gt == fx['grid_top']).all() and (gb == fx['grid_bot']).all()
    assert (O.pixel_shuffle2(fx['pixel_shuffle_in']) == fx['pixel_shuffle_out']).all()


@pytest.mark.parametrize('name', ['tiny', 'true'])
def test_block_steps_match_reference(name):
    """G2: one body Block (causal prefix, then a cached decode step) and one depth ParallelBlock (no-past token, then four
    tokens over one past key) of the reference, at a tiny shape and at the ImageNet head geometry (D = 1536, 24 x 64)."""
    import json
    from hqtransformer_amd import synth
    from hqtransformer_amd.spec import Stage2Spec
    from oracle.hqt_oracle import OracleStage2
    fx = load('g2_block_step.npz')
    spec = Stage2Spec(**json.loads(str(fx[f'{name}_spec'])))
    orc = OracleStage2(spec, synth.stage2_weights(spec, int(fx['weight_seed']), 'fixture'))
    B, nh, hs = 2, spec.n_heads, spec.head_dim
    cache = {}
    yp = orc._block('blocks.0', fx[f'{name}_xp'], cache, True)
    yn = orc._block('blocks.0', fx[f'{name}_xn'], cache, True)
    assert np.abs(yp - fx[f'{name}_yp']).max() <= 2e-4 and np.abs(yn - fx[f'{name}_yn']).max() <= 2e-4
    k, v = cache['blocks.0']                                           # [B, nh, 4, hs]; the reference's `present` = the new token only
    assert k.shape == (B, nh, 4, hs)
    assert np.abs(k[:, :, 3:].reshape(B * nh, 1, hs) - fx[f'{name}_k']).max() <= 1e-4
    assert np.abs(v[:, :, 3:].reshape(B * nh, 1, hs) - fx[f'{name}_v']).max() <= 1e-4
    dcache = {}
    y0 = orc._block('depths.0', fx[f'{name}_xd0'], dcache, False)
    y1 = orc._block('depths.0', fx[f'{name}_xd1'], dcache, False)
    assert np.abs(y0 - fx[f'{name}_yd0']).max() <= 2e-4 and np.abs(y1 - fx[f'{name}_yd1']).max() <= 2e-4
    dk, dv = dcache['depths.0']                                        # 1 past + 4 new keys
    assert np.abs(dk[:, :, 1:].reshape(B * nh, 4, hs) - fx[f'{name}_dk']).max() <= 1e-4
    assert np.abs(dv[:, :, 1:].reshape(B * nh, 4, hs) - fx[f'{name}_dv']).max() <= 1e-4


@pytest.mark.parametrize('si', [0, 1])
def test_l3_sampling_bit_exact(si):
    """G7: three-level HQTransformer 'parallel-add' sampling (1 + 4 + 16 codes per position), 64 positions, B = 3."""
    from hqtransformer_amd.spec import Stage2Spec
    from oracle.hqt_oracle import OracleStage2L3
    fx = load('g7_l3_tiny_cls.npz')
    spec = Stage2Spec(**json.loads(str(fx['spec'])))
    assert {k: tuple(v) for k, v in json.loads(str(fx['ref_shapes'])).items()} == dict(stage2_param_shapes(spec))
    orc = OracleStage2L3(spec, synth.stage2_weights(spec, int(fx['weight_seed']), 'fixture'))
    B, n = int(fx['B']), int(fx['n_steps'])
    noise = np.maximum(np.random.default_rng([int(fx['noise_seed']), 0x9e3779b9]).standard_exponential((n, 21, B, spec.vocab_top), dtype=np.float32),
                       np.float32(1e-30))
    tk, tp, T = json.loads(str(fx['settings']))[si]
    c0, c1, c2, lg = orc.sample(np.full(B, 7), B, n, noise, tk, tp, T, return_logits=True)
    assert np.abs(lg[fx['keep_steps']] - fx[f'logits_{si}']).max() <= 2e-4
    assert (c0 == fx[f'codes0_{si}']).all() and (c1 == fx[f'codes1_{si}']).all() and (c2 == fx[f'codes2_{si}']).all()
    assert float(fx[f'margin_{si}']) > 1.00005         # every draw is decided by a margin an fp32-accurate implementation resolves


@pytest.mark.parametrize('name', ['g7_l3_tiny_cls_parallel.npz', 'g7_l3_tiny_cls_parallel_reduce.npz', 'g7_l3_tiny_cls_top2mid2bot.npz'])
def test_l3_other_decoding_types_bit_exact(name):
    """G7b: the three other HQTransformer.decoding_type values whose three-level sampling runs in the reference -- 'parallel' (level-2
    tokens without the top code's embedding), 'parallel-reduce' ([V, 4 D] depth tables, one D-slice per child position) and
    'top2mid2bot' (a causal head of 21 one-token sub-steps fed through the spatial tables, hqtransformer.py:700-800) --,
    24 positions, B = 3, top-k / top-p / temperature per level (hqtransformer.py:105-157,526-551)."""
    from hqtransformer_amd.spec import DEPTH_DECODINGS, Stage2Spec
    from oracle.hqt_oracle import OracleStage2L3
    fx = load(name)
    spec = Stage2Spec(**json.loads(str(fx['spec'])))
    assert spec.depth_decoding in DEPTH_DECODINGS and spec.depth_decoding != 'parallel-add'
    assert {k: tuple(v) for k, v in json.loads(str(fx['ref_shapes'])).items()} == dict(stage2_param_shapes(spec))
    orc = OracleStage2L3(spec, synth.stage2_weights(spec, int(fx['weight_seed']), 'fixture'))
    B, n = int(fx['B']), int(fx['n_steps'])
    noise = np.maximum(np.random.default_rng([int(fx['noise_seed']), 0x9e3779b9]).standard_exponential((n, 21, B, spec.vocab_top), dtype=np.float32),
                       np.float32(1e-30))
    tk, tp, T = json.loads(str(fx['settings']))[0]
    c0, c1, c2, lg = orc.sample(np.full(B, int(fx['cond'])), B, n, noise, tk, tp, T, return_logits=True)
    assert np.abs(lg[fx['keep_steps']] - fx['logits_0']).max() <= 2e-4
    assert (c0 == fx['codes0_0']).all() and (c1 == fx['codes1_0']).all() and (c2 == fx['codes2_0']).all()
    assert float(fx['margin_0']) > 1.00005


def test_l3_decode_pixels():
    """G8: HQVAEGenerator.decode_code([top, mid, bottom]) -- additive pixel-shuffle pyramid + decoder."""
    from hqtransformer_amd.spec import Stage1Spec
    from oracle.hqt_oracle import OracleStage1
    fx = load('g8_l3_decode.npz')
    spec = Stage1Spec(**json.loads(str(fx['spec'])))
    assert {k: tuple(v) for k, v in json.loads(str(fx['ref_shapes'])).items()} == dict(stage1_param_shapes(spec))
    orc = OracleStage1(spec, synth.stage1_weights(spec, int(fx['weight_seed']), 'fixture'))
    assert np.abs(orc.decode_codes3([fx['code_t'], fx['code_m'], fx['code_b']]) - fx['pixels']).max() <= 1e-4
    assert np.abs(orc.decode_codes3([fx['code_t'][:1], None, None]) - fx['pixels_top_only']).max() <= 1e-4
    assert np.abs(orc.decode_codes3([None, None, fx['code_b'][:1]]) - fx['pixels_bot_only']).max() <= 1e-4


ENCODE_FIXTURES = ('g9_encode_64.npz', 'g9_encode_64_noinit.npz', 'g9_encode_64_l3.npz')


@pytest.mark.parametrize('name', ENCODE_FIXTURES)
def test_encode_side_vs_reference(name):
    """G9: SimRQGAN2Generator.encode / HQVAEGenerator.encode run by the reference itself (tools/gen_golden_enc.py): the encoder's
    feature map within 1e-4, then -- the fixtures sit >= 4e-4 away from any argmin tie -- bit-identical codes at every level,
    quantiser inputs and straight-through outputs within 1e-4, the commitment terms within 1e-5 relative."""
    from hqtransformer_amd.spec import Stage1Spec, stage1_encoder_param_shapes
    from oracle.hqt_oracle import OracleStage1
    fx = load(name)
    spec = Stage1Spec(**json.loads(str(fx['spec'])))
    want_shapes = dict(stage1_encoder_param_shapes(spec))
    want_shapes.update(stage1_param_shapes(spec))
    assert {k: tuple(v) for k, v in json.loads(str(fx['param_shapes'])).items()} == want_shapes
    orc = OracleStage1(spec, synth.stage1_weights(spec, int(fx['weight_seed']), 'fixture', encoder=True))
    x = fx['pixels']
    assert np.abs(orc.encoder(x[:1])[0] - fx['encoder_out'][0]).max() <= 1e-4
    out = orc.encode(x)
    assert np.abs(out['h'] - fx['h']).max() <= 1e-4
    L = 3 if spec.code_levels == 3 else 2
    assert float(fx['margins'].min()) > 1e-4
    for l in range(L):
        assert np.array_equal(out['codes'][l].reshape(-1), fx[f'code_{l}'].reshape(-1)), l
        assert np.abs(out['resid'][l] - fx[f'resid_{l}']).max() <= 1e-4
        assert abs(float(out['diff'][l]) - float(fx[f'diff_{l}'])) <= 1e-5 * abs(float(fx[f'diff_{l}'])) + 1e-7
        if f'quant_{l}' in fx:
            assert np.abs(out['quant'][l] - fx[f'quant_{l}']).max() <= 1e-4
    if 'recon' in fx:
        assert np.abs(out['recon'] - fx['recon']).max() <= 1e-4
    # decode(encode(x)) closes the loop through the decode-side oracle
    rec = orc.decode_codes3(out['codes']) if L == 3 else orc.decode_code(out['codes'][0], out['codes'][1])
    assert np.abs(rec - fx['reconstruction']).max() <= 1e-4


# ----------------------------------------------------------------------------------------- G12: trained-like statistics
def _g12():
    from hqtransformer_amd.spec import Stage1Spec, Stage2Spec
    fx = load('g12_trained.npz')
    s2 = Stage2Spec(**json.loads(str(fx['spec2'])))
    s1 = Stage1Spec(**json.loads(str(fx['spec1'])))
    return fx, s2, synth.stage2_weights(s2, int(fx['weight_seed2']), 'trained'), s1, synth.stage1_weights(s1, int(fx['weight_seed1']), 'trained')


def test_trained_profile_oracle_vs_reference():
    """Fixture G12 (tools/gen_golden_trained.py): the REFERENCE on weights with trained-like statistics -- LayerNorm / GroupNorm gains with
    outlier channels, a residual stream ~20x its input (largest block output 20.8), logits of std 3.6, decoder activations up to 77 -- pins
    the oracle at the scale of real checkpoints: codes bit-exact, logits within the 2e-4 bar of the unit-scale fixtures (measured 2.0e-5),
    pixels within north_star's 1e-4 (measured 2.8e-6)."""
    fx, s2, w2, s1, w1 = _g12()
    assert float(fx['stream_max']) > 15 and float(fx['act_max']) > 50 and float(fx['margin']) > 1.00005
    B, n = int(fx['B']), int(fx['n_steps'])
    noise = synth.exp_noise(int(fx['noise_seed']), n, B, s2.vocab_top)
    ct, cb, lg = O.OracleStage2(s2, w2).sample(np.full(B, 7), B, n, noise, return_logits=True)
    assert (ct == fx['codes_top']).all() and (cb == fx['codes_bot']).all()
    assert np.abs(lg - fx['logits']).max() <= LOGIT_TOL, np.abs(lg - fx['logits']).max()
    px = O.OracleStage1(s1, w1).decode_code(fx['code_t'], fx['code_b'])
    assert np.abs(px - fx['pixels']).max() <= PIXEL_TOL, np.abs(px - fx['pixels']).max()
