"""SPLIT precision (fp32-accurate convolutions on the matrix cores: fp16 hi/lo operands, fp32 accumulation) against the
CPU oracle and the reference-generated fixtures.  The bar is the EXACT one: decoded pixels within 1e-4 (north_star)."""
import json
import os

import numpy as np
import pytest
import torch

from hqtransformer_amd import synth
from hqtransformer_amd._lib import PRECISION_EXACT, PRECISION_FAST, PRECISION_SPLIT
from hqtransformer_amd.engine import Engine
from hqtransformer_amd.spec import Stage1Spec
from oracle import hqt_oracle as O
from tests.helpers import gate, load, stage1_from_fixture

pytestmark = pytest.mark.gpu
PIXEL_TOL = 1e-4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def engine_s1(spec, weights, max_batch):
    e = Engine(None, spec, dev(), max_batch)
    e.load(stage1=weights)
    e.finalize()
    return e


def np_(t):
    return t.detach().cpu().numpy()


def kernels_run(eng, fn):
    eng.timing(True)
    eng.timing_reset()
    out = fn()
    torch.cuda.synchronize()
    rep = eng.timing_report()
    eng.timing(False)
    return out, rep


def test_split_decode_wide_config_vs_oracle():
    """A config every layer of which the split kernels take (channel counts multiples of 64): 3x3 halo-tile convs incl. the
    upsampling one and conv_out's NCHW store, the 1x1 GEMM kernel (post_quant, nin_shortcut, attention q/k/v/proj with the V^T
    store), fused GroupNorm statistics -- pixels within 1e-4 of the oracle, on a NaN-poisoned workspace, and batch-invariant."""
    spec = Stage1Spec(ch=64, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[16], resolution=64, z_channels=64,
                      embed_dim=32, n_embed=256)
    weights = synth.stage1_weights(spec, 31, 'fixture')
    r = np.random.default_rng(32)
    ct, cb = r.integers(0, 256, (3, 8, 8)), r.integers(0, 256, (3, 16, 16))
    want = O.OracleStage1(spec, weights).decode_code(ct, cb)
    os.environ['HQT_POISON_WORKSPACE'] = '1'
    try:
        eng = engine_s1(spec, weights, 3)
    finally:
        del os.environ['HQT_POISON_WORKSPACE']
    tct, tcb = torch.from_numpy(ct), torch.from_numpy(cb)
    got, rep = kernels_run(eng, lambda: np_(eng.decode(tct, tcb, precision=PRECISION_SPLIT)))
    assert rep.get('split_pack', (0, 0))[0] >= 10, rep            # the matrix-core path ran (not the EXACT fallback)
    err = np.abs(got - want).max()
    assert err <= PIXEL_TOL, err
    exact = np_(eng.decode(tct, tcb, precision=PRECISION_EXACT))
    assert np.abs(exact - want).max() <= PIXEL_TOL
    one = np_(eng.decode(tct[1:2], tcb[1:2], precision=PRECISION_SPLIT))
    assert np.array_equal(one[0], got[1])                          # same bits whatever the batch
    again = np_(eng.decode(tct, tcb, precision=PRECISION_SPLIT))
    assert np.array_equal(again, got)                              # run-to-run deterministic
    cl = np_(eng.decode(tct, tcb, precision=PRECISION_SPLIT, clamp01=True))
    np.testing.assert_allclose(cl, O.postprocess(want), atol=PIXEL_TOL)
    top_only = np_(eng.decode(tct[:1], None, precision=PRECISION_SPLIT))
    assert np.abs(top_only - O.OracleStage1(spec, weights).decode_code(ct[:1], None)).max() <= PIXEL_TOL


def test_split_decode_reference_fixtures():
    """The reference's own decode_code outputs (fixtures G5): narrow layers fall back to the fp32 vector kernels inside SPLIT
    mode; every mix of the two must stay within 1e-4."""
    for name in ('g5_decode_64.npz', 'g5_decode_256.npz'):
        fx = load(name)
        spec, weights = stage1_from_fixture(fx)
        eng = engine_s1(spec, weights, 2)
        px = np_(eng.decode(torch.from_numpy(fx['code_t']), torch.from_numpy(fx['code_b']), precision=PRECISION_SPLIT))
        assert np.abs(px - fx['pixels']).max() <= PIXEL_TOL, name


def test_imagenet_size_decoder_all_precisions_vs_oracle():
    """The benchmark's own decoder (configs/imagenet-12l.yaml: ch 128, ch_mult [1, 2, 4, 4], 53.95 M parameters, 256 x 256)
    on 2 images of random codes against the CPU oracle: EXACT and SPLIT within 1e-4 (north_star's pixel bar), FAST (bf16)
    within max 0.08 / mean 0.01 (measured 0.043 / 0.0058)."""
    from hqtransformer_amd.config import load_config
    from hqtransformer_amd.spec import stage1_spec_from_config
    spec = stage1_spec_from_config(load_config(os.path.join(ROOT, 'configs', 'imagenet-12l.yaml')))
    weights = synth.stage1_weights(spec, 1, 'bench')
    r = np.random.default_rng(7)
    ct, cb = r.integers(0, spec.n_embed, (2, 8, 8)), r.integers(0, spec.n_embed, (2, 16, 16))
    want = O.OracleStage1(spec, weights).decode_code(ct, cb)
    eng = engine_s1(spec, weights, 2)
    tct, tcb = torch.from_numpy(ct), torch.from_numpy(cb)
    split, rep = kernels_run(eng, lambda: np_(eng.decode(tct, tcb, precision=PRECISION_SPLIT)))
    assert 'conv3x3' in rep and rep.get('split_pack', (0, 0))[0] >= 30, rep
    assert rep.get('variant:conv_out_direct:conv_out', (0, 0))[0] == 1, rep    # norm_out + swish + conv_out: one fp32 kernel
    e_split = np.abs(split - want).max()
    assert e_split <= PIXEL_TOL, e_split
    exact = np_(eng.decode(tct, tcb, precision=PRECISION_EXACT))
    e_exact = np.abs(exact - want).max()
    assert e_exact <= PIXEL_TOL, e_exact
    fast = np_(eng.decode(tct, tcb, precision=PRECISION_FAST))
    d = np.abs(fast - want)
    gate('imagenet_decoder.fast_pixels.max', d.max(), 0.08)                # measured 0.043
    gate('imagenet_decoder.fast_pixels.mean', d.mean(), 1e-2)              # measured 0.0058
    print(f'imagenet-size decoder vs oracle: exact {e_exact:.2e}, split {e_split:.2e}, fast max {d.max():.3f} mean {d.mean():.4f}, '
          f'output std {want.std():.3f}')


def test_split_five_level_geometry_vs_oracle():
    """BASELINE configs[3] geometry in SPLIT precision: the reference's 5-level pattern ch_mult [1, 2, 4, 4, 4] with attention at
    32 x 32 = 1024 tokens, 64 base channels (so every conv of the decoder is one the matrix-core kernels take) at 512 x 512 on one
    image against the CPU oracle: pixels within 1e-4, and the SPLIT kernels -- not the fp32 fallback -- did the work."""
    spec = Stage1Spec(ch=64, ch_mult=[1, 2, 4, 4, 4], num_res_blocks=1, attn_resolutions=[32], resolution=512, z_channels=64,
                      embed_dim=32, n_embed=256)
    assert spec.z_res == 16
    weights = synth.stage1_weights(spec, 101, 'fixture')
    r = np.random.default_rng(102)
    ct, cb = r.integers(0, 256, (1, 8, 8)), r.integers(0, 256, (1, 16, 16))
    want = O.OracleStage1(spec, weights).decode_code(ct, cb)
    eng = engine_s1(spec, weights, 1)
    tct, tcb = torch.from_numpy(ct), torch.from_numpy(cb)
    got, rep = kernels_run(eng, lambda: np_(eng.decode(tct, tcb, precision=PRECISION_SPLIT)))
    assert got.shape == (1, 3, 512, 512)
    assert rep.get('split_pack', (0, 0))[0] >= 15 and 'conv3x3' in rep, rep
    err = np.abs(got - want).max()
    assert err <= PIXEL_TOL, err


def test_full_size_1024_decoder_properties():
    """BASELINE configs[3] at FULL size (Decoder(resolution = 1024, ch = 128, ch_mult [1, 2, 4, 4, 4], 2 ResnetBlocks per
    level, attention at 32 x 32, codes 16 x 16 + 32 x 32: 2.8 TFLOP per image) -- far beyond what the CPU oracle finishes in a test,
    so the size-independent properties: SPLIT output finite, run-to-run bit-identical, the same bits for an image whatever batch it is
    decoded in and wherever it sits in it, clamp01 = clamp(0.5 x + 0.5) of the unclamped pixels, different codes -> different
    pixels; and SPLIT against FAST (an independent set of kernels: bf16 halo convs) inside the bf16 budget of the ImageNet-size
    test.  The 512 x 512 geometry above pins the same layer pattern to the oracle."""
    spec = Stage1Spec(ch=128, ch_mult=[1, 2, 4, 4, 4], num_res_blocks=2, attn_resolutions=[32], resolution=1024, z_channels=256,
                      embed_dim=256, n_embed=8192, use_init_downsample=True)
    assert spec.z_res == 32
    eng = engine_s1(spec, synth.stage1_weights(spec, 1, 'bench'), 3)
    r = np.random.default_rng(5)
    ct = torch.from_numpy(r.integers(0, spec.n_embed, (3, 16, 16)))
    cb = torch.from_numpy(r.integers(0, spec.n_embed, (3, 32, 32)))
    px = eng.decode(ct, cb, precision=PRECISION_SPLIT)
    assert tuple(px.shape) == (3, 3, 1024, 1024) and bool(torch.isfinite(px).all())
    assert bool(torch.equal(px, eng.decode(ct, cb, precision=PRECISION_SPLIT))), 'not deterministic'
    one = eng.decode(ct[2:3], cb[2:3], precision=PRECISION_SPLIT)
    assert bool(torch.equal(one[0], px[2])), 'pixels of an image depend on the batch it was decoded in'
    swapped = eng.decode(ct.flip(0), cb.flip(0), precision=PRECISION_SPLIT)
    assert bool(torch.equal(swapped.flip(0), px))
    cl = eng.decode(ct, cb, precision=PRECISION_SPLIT, clamp01=True)
    assert float((cl - torch.clamp(0.5 * px + 0.5, 0.0, 1.0)).abs().max()) <= 1e-6
    assert float((px[0] - px[1]).abs().max()) > 1e-2
    fast = eng.decode(ct, cb, precision=PRECISION_FAST)
    d = (fast - px).abs()
    gate('decoder_1024.fast_vs_split_pixels.max', float(d.max()), 0.1 * max(1.0, float(px.abs().max()) / 5.0))
    gate('decoder_1024.fast_vs_split_pixels.mean', float(d.mean()), 1e-2)


def test_split_range_check_large_and_tiny_activations():
    """SPLIT carries activations as fp16 hi / lo planes (|x| < 65504).  A codebook scaled so that the quantised vectors reach ~4e5 makes
    the very first operand pass (post_quant_conv_b: no GroupNorm in front) meet values fp16 cannot hold: the call must not hand out NaNs
    silently -- hqt_range_check raises HQT_ERR_RANGE, the reference-shaped surface raises with it, EXACT on the same weights still
    matches the oracle, and the flag is cleared for the next call.  Scaled to ~4e-6 instead (hi planes are fp16 subnormals) the SPLIT
    pixels stay within 1e-4 of the oracle."""
    from hqtransformer_amd._lib import HqtError
    spec = Stage1Spec(ch=64, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[16], resolution=64, z_channels=64,
                      embed_dim=32, n_embed=256)
    base = synth.stage1_weights(spec, 33, 'fixture')
    r = np.random.default_rng(34)
    ct, cb = r.integers(0, 256, (2, 8, 8)), r.integers(0, 256, (2, 16, 16))
    tct, tcb = torch.from_numpy(ct), torch.from_numpy(cb)
    for scale, overflow in ((1e5, True), (1e-6, False)):
        w = dict(base)
        w['quantize_t.embedding'] = (base['quantize_t.embedding'] * np.float32(scale)).astype(np.float32)
        w['quantize_b.embedding'] = (base['quantize_b.embedding'] * np.float32(scale)).astype(np.float32)
        want = O.OracleStage1(spec, w).decode_code(ct, cb)
        eng = engine_s1(spec, w, 2)
        got = eng.decode(tct, tcb, precision=PRECISION_SPLIT)
        if overflow:
            with pytest.raises(HqtError) as ei:
                eng.range_check()
            assert ei.value.code == -7 and 'fp16 range' in str(ei.value)
            assert bool(torch.isfinite(got).all()), 'saturation keeps the (invalid) output finite'
            eng.range_check()                                              # cleared: nothing pending, nothing raised
            exact = np_(eng.decode(tct, tcb, precision=PRECISION_EXACT))
            eng.range_check()
            assert np.abs(exact - want).max() <= PIXEL_TOL * max(1.0, float(np.abs(want).max()))
            again = eng.decode(tct, tcb, precision=PRECISION_SPLIT)        # and a second offending call is reported again
            with pytest.raises(HqtError):
                eng.range_check()
            del again
        else:
            eng.range_check()
            assert np.abs(np_(got) - want).max() <= PIXEL_TOL, np.abs(np_(got) - want).max()


def test_upsampling_conv_as_four_phase_convs_matches_the_nine_tap_form():
    """The nearest-x2 upsampling convs run as four 2x2 convolutions on the low-resolution image with pre-summed taps
    (conv2x2_split_up16_kernel: the same sums as layers.py:42-53, regrouped).  HQT_SPLIT_UP=0 switches back to nine taps on the
    upsampled image (conv3x3_split_ring16_kernel); both must sit within 1e-4 of the oracle, and within 2e-5 of each other, on a
    decoder whose three upsampling convs all take the phase kernel -- borders, the 16-pixel-wide single-tile case and GroupNorm
    statistics from the phase partials included."""
    import subprocess
    import sys
    import tempfile
    code = """
import sys, numpy as np, torch
sys.path.insert(0, %r)
from hqtransformer_amd import synth
from hqtransformer_amd._lib import PRECISION_SPLIT
from hqtransformer_amd.engine import Engine
from hqtransformer_amd.spec import Stage1Spec
spec = Stage1Spec(ch=128, ch_mult=[1, 1, 2], num_res_blocks=1, attn_resolutions=[16], resolution=128, z_channels=64, embed_dim=32, n_embed=256, use_init_downsample=True)
w = synth.stage1_weights(spec, 41, 'fixture')
r = np.random.default_rng(42)
ct, cb = r.integers(0, 256, (3, 8, 8)), r.integers(0, 256, (3, 16, 16))
e = Engine(None, spec, torch.device('cuda:0'), 3); e.load(stage1=w); e.finalize()
e.timing(True); e.timing_reset()
px = e.decode(torch.from_numpy(ct), torch.from_numpy(cb), precision=PRECISION_SPLIT)
torch.cuda.synchronize()
np.save(sys.argv[1], px.cpu().numpy())
""" % ROOT
    spec = Stage1Spec(ch=128, ch_mult=[1, 1, 2], num_res_blocks=1, attn_resolutions=[16], resolution=128, z_channels=64,
                      embed_dim=32, n_embed=256, use_init_downsample=True)
    assert spec.z_res == 16
    weights = synth.stage1_weights(spec, 41, 'fixture')
    r = np.random.default_rng(42)
    ct, cb = r.integers(0, 256, (3, 8, 8)), r.integers(0, 256, (3, 16, 16))
    want = O.OracleStage1(spec, weights).decode_code(ct, cb)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, env in (('phase', {}), ('nine', {'HQT_SPLIT_UP': '0'})):
            path = os.path.join(tmp, name + '.npy')
            rr = subprocess.run([sys.executable, '-c', code, path], cwd=ROOT, env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
            assert rr.returncode == 0, (name, rr.stdout[-1500:], rr.stderr[-1500:])
            out[name] = np.load(path)
    for name, px in out.items():
        err = np.abs(px - want).max()
        assert err <= PIXEL_TOL, (name, err)
    assert np.abs(out['phase'] - out['nine']).max() <= 2e-5
    assert not np.array_equal(out['phase'], out['nine']), 'HQT_SPLIT_UP=0 did not change the kernel'


def test_conv_out_in_one_fp32_kernel_matches_the_matrix_core_form():
    """SPLIT's last stage -- norm_out, swish, conv_out, clamp -- runs as ONE fp32 kernel on the fp32 tensor (conv_out_direct_kernel: no
    operand pass over the decoder's largest tensor, no 3 -> 16 channel padding).  HQT_CONV_OUT_DIRECT=0 switches back to the operand pass
    + conv3x3_split_out16_kernel; both must sit within 1e-4 of the oracle and within 2e-5 of each other (image borders, clamp and a
    batch that is not a multiple of anything included), and the timing report must name the kernel that ran."""
    import subprocess
    import sys
    import tempfile
    code = """
import sys, json, numpy as np, torch
sys.path.insert(0, %r)
from hqtransformer_amd import synth
from hqtransformer_amd._lib import PRECISION_SPLIT
from hqtransformer_amd.engine import Engine
from hqtransformer_amd.spec import Stage1Spec
spec = Stage1Spec(ch=64, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[16], resolution=64, z_channels=64, embed_dim=32, n_embed=256)
w = synth.stage1_weights(spec, 51, 'fixture')
r = np.random.default_rng(52)
ct, cb = r.integers(0, 256, (5, 8, 8)), r.integers(0, 256, (5, 16, 16))
e = Engine(None, spec, torch.device('cuda:0'), 3); e.load(stage1=w); e.finalize()
e.timing(True); e.timing_reset()
px = e.decode(torch.from_numpy(ct), torch.from_numpy(cb), precision=PRECISION_SPLIT)
torch.cuda.synchronize()
rep = {k: v[0] for k, v in e.timing_report().items()}
e.timing(False)
cl = e.decode(torch.from_numpy(ct), torch.from_numpy(cb), precision=PRECISION_SPLIT, clamp01=True)
np.savez(sys.argv[1], px=px.cpu().numpy(), cl=cl.cpu().numpy(), rep=json.dumps(rep))
""" % ROOT
    spec = Stage1Spec(ch=64, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[16], resolution=64, z_channels=64, embed_dim=32, n_embed=256)
    weights = synth.stage1_weights(spec, 51, 'fixture')
    r = np.random.default_rng(52)
    ct, cb = r.integers(0, 256, (5, 8, 8)), r.integers(0, 256, (5, 16, 16))
    want = O.OracleStage1(spec, weights).decode_code(ct, cb)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, env in (('direct', {}), ('planes', {'HQT_CONV_OUT_DIRECT': '0'})):
            path = os.path.join(tmp, name + '.npz')
            rr = subprocess.run([sys.executable, '-c', code, path], cwd=ROOT, env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
            assert rr.returncode == 0, (name, rr.stdout[-1500:], rr.stderr[-1500:])
            z = np.load(path)
            out[name] = (z['px'], z['cl'], json.loads(str(z['rep'])))
    for name, (px, cl, rep) in out.items():
        assert np.abs(px - want).max() <= PIXEL_TOL, (name, np.abs(px - want).max())
        np.testing.assert_allclose(cl, O.postprocess(want), atol=PIXEL_TOL)
    # chunks of 3 + 2 images: one launch per chunk
    assert out['direct'][2].get('variant:conv_out_direct:conv_out') == 2, out['direct'][2]
    assert 'variant:conv_out_direct:conv_out' not in out['planes'][2], out['planes'][2]
    assert out['planes'][2]['split_pack'] == out['direct'][2]['split_pack'] + 2        # the operand pass the fused kernel does not need
    assert np.abs(out['direct'][0] - out['planes'][0]).max() <= 2e-5


def test_resblock_epilogue_emits_the_upsampling_convs_operand_planes():
    """A ResnetBlock whose only consumer is an upsampling conv (no GroupNorm in between) writes fp16 hi / lo operand planes from its second conv's epilogue
    (GemmArgs::out_split) instead of the fp32 tensor + a separate operand pass.  Same split of the same fp32 values: HQT_SPLIT_PLANES_OUT=0 (the operand pass) must give
    the SAME pixels bit for bit; the timing report must show the passes that disappeared (the block in front of an attention layer is not eligible)."""
    import subprocess
    import sys
    import tempfile
    code = """
import sys, json, numpy as np, torch
sys.path.insert(0, %r)
from hqtransformer_amd import synth
from hqtransformer_amd._lib import PRECISION_SPLIT
from hqtransformer_amd.engine import Engine
from hqtransformer_amd.spec import Stage1Spec
spec = Stage1Spec(ch=128, ch_mult=[1, 1, 2], num_res_blocks=1, attn_resolutions=[16], resolution=128, z_channels=64, embed_dim=32, n_embed=256, use_init_downsample=True)
w = synth.stage1_weights(spec, 71, 'fixture')
r = np.random.default_rng(72)
ct, cb = r.integers(0, 256, (3, 8, 8)), r.integers(0, 256, (3, 16, 16))
e = Engine(None, spec, torch.device('cuda:0'), 3); e.load(stage1=w); e.finalize()
e.timing(True); e.timing_reset()
px = e.decode(torch.from_numpy(ct), torch.from_numpy(cb), precision=PRECISION_SPLIT)
torch.cuda.synchronize()
e.range_check()
rep = {k: v[0] for k, v in e.timing_report().items()}
np.savez(sys.argv[1], px=px.cpu().numpy(), rep=json.dumps(rep))
""" % ROOT
    spec = Stage1Spec(ch=128, ch_mult=[1, 1, 2], num_res_blocks=1, attn_resolutions=[16], resolution=128, z_channels=64, embed_dim=32, n_embed=256,
                      use_init_downsample=True)
    weights = synth.stage1_weights(spec, 71, 'fixture')
    r = np.random.default_rng(72)
    ct, cb = r.integers(0, 256, (3, 8, 8)), r.integers(0, 256, (3, 16, 16))
    want = O.OracleStage1(spec, weights).decode_code(ct, cb)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, env in (('planes', {}), ('pass', {'HQT_SPLIT_PLANES_OUT': '0'})):
            path = os.path.join(tmp, name + '.npz')
            rr = subprocess.run([sys.executable, '-c', code, path], cwd=ROOT, env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
            assert rr.returncode == 0, (name, rr.stdout[-1500:], rr.stderr[-1500:])
            z = np.load(path)
            out[name] = (z['px'], json.loads(str(z['rep'])))
    assert np.abs(out['planes'][0] - want).max() <= PIXEL_TOL
    assert np.array_equal(out['planes'][0], out['pass'][0])
    n = out['planes'][1].get('variant:conv3x3_planes_out:conv3x3', 0)
    assert n == 2, out['planes'][1]                                # the 32 x 32 and 64 x 64 levels; the 16 x 16 level ends in an attention block
    assert 'variant:conv3x3_planes_out:conv3x3' not in out['pass'][1]
    assert out['pass'][1]['split_pack'] == out['planes'][1]['split_pack'] + n


def test_split_sampler_tiny_fixture_bit_exact():
    """HQT_PRECISION_SPLIT on the stage-2 entry point (round 4): the EXACT launch sequence with every nn.Linear on the matrix cores
    (fp32 activation rows split into fp16 hi / lo while staged, fp16 hi / lo weight planes, three MFMAs per term) -- the reference
    fixture G4 (codes of 64 positions under three sampler settings) must come out bit for bit, logits within 2e-4."""
    import json
    from tests.helpers import stage2_from_fixture
    fx = load('g4_tiny_cls.npz')
    spec, weights = stage2_from_fixture(fx)
    B, n = int(fx['B']), int(fx['n_steps'])
    e = Engine(spec, None, dev(), B, n)
    e.load(stage2=weights)
    e.finalize()
    noise = torch.from_numpy(synth.exp_noise(int(fx['noise_seed']), n, B, spec.vocab_top))
    e.timing(True)
    e.timing_reset()
    for si, (tk, tp, T) in enumerate(json.loads(str(fx['settings']))):
        for graph in (False, True):
            ct, cb, lg = e.sample(B, torch.full((B,), 7), n, precision=PRECISION_SPLIT, top_k=tk, top_p=tp, temperature=T, noise=noise,
                                  return_logits=True, use_graph=graph)
            assert (ct.cpu().numpy() == fx[f'codes_top_{si}']).all() and (cb.cpu().numpy() == fx[f'codes_bot_{si}']).all(), (si, graph)
            scale = np.array([T[0]] + [T[1]] * 4, np.float32)[None, :, None, None]
            np.testing.assert_allclose(lg.cpu().numpy()[fx['keep_steps']] / scale, fx[f'logits_{si}'], atol=2e-4, rtol=0)
    v = {k: c[0] for k, c in e.timing_report().items() if k.startswith('variant:')}
    e.timing(False)
    # 4 / 16-row launches: up to 256 rows SPLIT takes the fp32 matrix instructions (exact_gemm.hip), above that the fp16 hi / lo GEMM
    # (covered at 512 / 640 / 2048 rows by tests/test_gpu_timed_schedule.py); never the vector-ALU kernel
    assert any(k.startswith(('variant:split_gemm:', 'variant:exact_mfma:')) for k in v), v
    assert not any(k.startswith('variant:gemm_generic_f32:') for k in v), f'fp32 vector-ALU GEMMs ran inside a SPLIT call: {v}'
    e.range_check()


def test_split_sampler_k_sliced_gemms_on_the_tiny_model():
    """SPLIT AR above 256 rows on fixture G4's tiny model (D = 128: K = 128 -> two K slices, K = 512 -> four): the K-sliced GEMMs + their combine launch
    (split_rows_combine_kernel: slices summed in index order, bias, GELU, residual) against EXACT on the same noise -- identical codes, logits <= 2e-4 -- graph replay == eager, bitwise; the variant counters prove both slice counts ran."""
    from tests.helpers import stage2_from_fixture
    fx = load('g4_tiny_cls.npz')
    spec, weights = stage2_from_fixture(fx)
    B, n = 320, 3
    e = Engine(spec, None, dev(), B, 8)
    e.load(stage2=weights)
    e.finalize()
    noise = torch.from_numpy(synth.exp_noise(21, n, B, spec.vocab_top))
    cond = torch.from_numpy((np.arange(B) * 3) % spec.n_classes)
    ct, cb, lg = e.sample(B, cond, n, precision=0, noise=noise, return_logits=True, use_graph=False)
    e.timing(True)
    e.timing_reset()
    st, sb, ls = e.sample(B, cond, n, precision=PRECISION_SPLIT, noise=noise, return_logits=True, use_graph=False)
    v = {k: c[0] for k, c in e.timing_report().items() if k.startswith('variant:')}
    e.timing(False)
    assert any(k.startswith('variant:split_gemm_kslices2:') for k in v) and any(k.startswith('variant:split_gemm_kslices4:') for k in v), v
    assert (ls - lg).abs().max().item() <= 2e-4
    assert torch.equal(st, ct) and torch.equal(sb, cb)
    sg, sbg, lsg = e.sample(B, cond, n, precision=PRECISION_SPLIT, noise=noise, return_logits=True, use_graph=True)
    assert torch.equal(lsg, ls) and torch.equal(sg, st) and torch.equal(sbg, sb)
    e.range_check()
