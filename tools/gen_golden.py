#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (CPU, fp32) in the build container.

Container-only tool: it imports /root/reference (SURVEY.md Appendix C recipe) and therefore cannot run
on the GPU box.  Only its outputs -- inputs and expected outputs, never reference source -- are
committed.  Weights and noise are not stored: both sides regenerate them from
hqtransformer_amd.synth (numpy default_rng keyed by state-dict name).

    python tools/gen_golden.py            # rewrites every fixture
"""
import copy
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = '/root/reference'
sys.path.insert(0, REF)

# --- import shims: omegaconf stub, bare hqvae.models package, Tensor.cuda no-op
_om = types.ModuleType('omegaconf')
_om.OmegaConf = object
sys.modules['omegaconf'] = _om
# the reference's `hqvae` is a namespace package; this repository's own `hqvae/` import shim (a regular package) would win
# the import regardless of sys.path order, so the reference tree is bound explicitly
_hq = types.ModuleType('hqvae')
_hq.__path__ = [os.path.join(REF, 'hqvae')]
sys.modules['hqvae'] = _hq
_pkg = types.ModuleType('hqvae.models')
_pkg.__path__ = [os.path.join(REF, 'hqvae/models')]
sys.modules['hqvae.models'] = _pkg

import torch  # noqa: E402
from einops import rearrange  # noqa: E402
from hqvae.models.stage2 import hierarchical_ar as ref_har  # noqa: E402
from hqvae.models.stage2.hierarchical_ar import iHQGPT  # noqa: E402
from hqvae.models.stage1.generator import SimRQGAN2Generator  # noqa: E402
from hqvae.utils import sampling as ref_sampling  # noqa: E402

from hqtransformer_amd import synth  # noqa: E402
from hqtransformer_amd.spec import (Stage1Spec, Stage2Spec, stage1_is_ignored as _s1_ign, stage1_is_encoder_key)  # noqa: E402


def stage1_is_ignored(k):          # decode-side fixtures: the encoder tensors keep the reference's own initialisation
    return _s1_ign(k) or stage1_is_encoder_key(k)


torch.Tensor.cuda = lambda self, *a, **k: self
torch.set_grad_enabled(False)
torch.set_num_threads(8)
OUT = os.path.join(ROOT, 'tests', 'golden')


class AD(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return AD(copy.deepcopy(dict(self), memo))


def build_stage2(spec: Stage2Spec, seed: int, profile: str = 'fixture'):
    hp = AD(embed_dim=spec.embed_dim, n_layers=spec.n_layers, n_heads=spec.n_heads, n_dense_layers=spec.n_layers,
            ctx_len=None, ctx_len_img=spec.ctx_len_img, ctx_len_txt=spec.ctx_len_txt, embd_pdrop=0.0,
            resid_pdrop=0.1, attn_pdrop=0.0, mlp_bias=True, attn_bias=True, gelu_use_approx=spec.gelu_approx,
            use_head_txt=True, n_classes=spec.n_classes, causal_attn=None,
            embedding_type='reduce' if spec.embedding == 1 else 'transformer1', position_embedding='1d',
            bottom_head_type='linear', use_random_order=False, rate_random_order=1.0)
    hp_dec = None
    if spec.n_layers_depth != 4:
        hp_dec = copy.deepcopy(hp)
        hp_dec.n_layers = spec.n_layers_depth
    m = iHQGPT(spec.vocab_top, spec.vocab_bot, spec.vocab_txt, 4, spec.cond == 1, spec.cond == 2, 'parallel', hp, hp_dec)
    sd = {k: torch.from_numpy(v) for k, v in synth.stage2_weights(spec, seed, profile).items()}
    ref_shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert ref_shapes == {k: tuple(v.shape) for k, v in sd.items()}, 'spec.stage2_param_shapes != reference state_dict'
    m.load_state_dict(sd, strict=True)
    return m.eval(), ref_shapes


def build_stage1(spec: Stage1Spec, seed: int, profile: str = 'fixture'):
    hp = AD(double_z=False, z_channels=spec.z_channels, resolution=spec.resolution, in_channels=3, out_ch=spec.out_ch,
            ch=spec.ch, ch_mult=list(spec.ch_mult), num_res_blocks=spec.num_res_blocks,
            attn_resolutions=list(spec.attn_resolutions), pdrop=0.0, use_init_downsample=spec.use_init_downsample,
            use_mid_block=spec.use_mid_block, use_attn=spec.use_attn)
    aux = AD(upsample='pixelshuffle', shared_codebook=False, bottom_start=10 ** 11, decoding_type='concat',
             restart_unused_codes=None, code_levels=None)
    g = SimRQGAN2Generator(spec.n_embed, spec.embed_dim, True, hp, aux)
    sd = {k: torch.from_numpy(v) for k, v in synth.stage1_weights(spec, seed, profile).items()}
    ref_shapes = {k: tuple(v.shape) for k, v in g.state_dict().items() if not stage1_is_ignored(k)}
    assert ref_shapes == {k: tuple(v.shape) for k, v in sd.items()}, 'spec.stage1_param_shapes != reference state_dict'
    missing, unexpected = g.load_state_dict(sd, strict=False)
    assert not unexpected and all(stage1_is_ignored(k) for k in missing)
    return g.eval(), ref_shapes


def run_sampling(model, spec: Stage2Spec, cond, B, n_steps, noise, top_k, top_p, temps, given_top=None):
    """Runs the reference's sampling_ihqgpt with torch.multinomial replaced by argmax(p / q_next)
    (identity established in SURVEY.md §0 item 5 and re-checked by fixture 'multinomial' below) and
    records the post-temperature logits of every draw plus the winner/runner-up margin."""
    if given_top is not None:          # the reference skips the top multinomial when the top code is given
        noise = noise[:, 1:]
    draws = iter(noise.reshape(-1, B, noise.shape[-1]))
    logits_log, margins = [], []
    real_topk = ref_sampling.cutoff_topk_logits

    def topk_spy(logits, k):
        logits_log.append(logits.detach().clone().numpy())
        return real_topk(logits, k)

    def fake_multinomial(probs, num_samples=1, **kw):
        q = torch.from_numpy(next(draws))
        r = probs / q
        top2 = torch.topk(r, 2, dim=-1).values
        margins.append(float((top2[:, 0] / top2[:, 1]).min()))
        return torch.argmax(r, dim=-1, keepdim=True)

    real_mn = torch.multinomial
    ref_har.cutoff_topk_logits = topk_spy
    torch.multinomial = fake_multinomial
    try:
        ct, cb = ref_sampling.sampling_ihqgpt(model, num_candidates=B, cond=cond, top_k_top=top_k[0], top_p_top=top_p[0],
                                              top_k_bot=top_k[1], top_p_bot=top_p[1], softmax_temperature=list(temps),
                                              is_tqdm=False, use_fp16=True, max_seq_len=n_steps, model_stage1=None,
                                              given_top_code=given_top)
    finally:
        torch.multinomial = real_mn
        ref_har.cutoff_topk_logits = real_topk
    lg = np.stack(logits_log).reshape(n_steps, 5, B, -1)
    return ct.numpy(), cb.numpy(), lg.astype(np.float32), float(min(margins))


def spec_json(spec):
    return json.dumps(spec.__dict__)


def main():
    os.makedirs(OUT, exist_ok=True)

    # ---------------------------------------------------------------- G1: sampler known-answer tests
    rng = np.random.default_rng(11)
    V = 512
    logits = (3.0 * rng.standard_normal((6, V))).astype(np.float32)
    logits[4, :8] = logits[4, 8]            # tie row: nine equal logits straddling a top-k cut
    logits[5] = np.sort(logits[5])[::-1]    # monotone row
    q = np.maximum(rng.standard_exponential((6, V), dtype=np.float32), 1e-30)
    cases = [(None, None, 1.0), (64, 0.9, 1.0), (8, 1.0, 0.95), (512, 1.0, 1.0), (3, 0.5, 0.7), (None, 0.95, 1.3)]
    g1 = dict(logits=logits, noise=q, cases=json.dumps(cases))
    for ci, (k, p, T) in enumerate(cases):
        lg = torch.from_numpy(logits.copy())
        lg /= T
        lg = ref_sampling.cutoff_topk_logits(lg, k)
        pr = torch.softmax(lg, dim=-1)
        pr = ref_sampling.cutoff_topp_probs(pr, p)
        g1[f'probs_{ci}'] = pr.numpy()
        g1[f'index_{ci}'] = torch.argmax(pr / torch.from_numpy(q), dim=-1).numpy()
    # real torch.multinomial: under one seed it draws exactly the Exp(1) tensor that
    # empty_like(p).exponential_(1) draws under the same seed, and returns argmax(p / q)
    pr = torch.softmax(torch.from_numpy(logits), dim=-1)
    torch.manual_seed(1234)
    idx = torch.multinomial(pr, num_samples=1)
    torch.manual_seed(1234)
    captured = [torch.empty_like(pr).exponential_(1).numpy()]
    g1['mn_probs'] = pr.numpy()
    g1['mn_noise'] = captured[0]
    g1['mn_index'] = idx[:, 0].numpy()
    assert (np.argmax(g1['mn_probs'] / g1['mn_noise'], -1) == g1['mn_index']).all()
    np.savez_compressed(os.path.join(OUT, 'g1_sampler.npz'), **g1)
    print('g1_sampler ok')

    # ---------------------------------------------------------------- G3/G4: full sampling, tiny configs
    keep_steps = [0, 1, 2, 31, 63]
    tiny_cls = Stage2Spec(embed_dim=128, n_layers=4, n_heads=4, n_layers_depth=4, vocab_top=512, vocab_bot=512,
                          vocab_txt=64, ctx_len_img=64, ctx_len_txt=16, n_classes=10, cond=1, embedding=0)
    m, shapes = build_stage2(tiny_cls, seed=3)
    out = dict(spec=spec_json(tiny_cls), weight_seed=3, noise_seed=5, B=4, n_steps=64,
               param_shapes=json.dumps({k: list(v) for k, v in shapes.items()}), keep_steps=np.array(keep_steps))
    noise = synth.exp_noise(5, 64, 4, 512)
    settings = [((None, None), (None, None), (1.0, 1.0)), ((64, 32), (0.9, 0.8), (0.95, 0.9)), ((512, 512), (1.0, 1.0), (1.0, 1.0))]
    out['settings'] = json.dumps(settings)
    for si, (tk, tp, T) in enumerate(settings):
        ct, cb, lg, margin = run_sampling(m, tiny_cls, 7, 4, 64, noise, tk, tp, T)
        out[f'codes_top_{si}'], out[f'codes_bot_{si}'] = ct, cb
        out[f'logits_{si}'] = lg[keep_steps]
        out[f'margin_{si}'] = margin
        print(f'tiny_cls setting {si}: margin {margin:.6f}')
    # teacher-forced top code (the reference's given_top_code path) -- only the bottom draws are free
    given = torch.from_numpy(np.random.default_rng(9).integers(0, 512, (4, 8)))
    # the reference indexes given_top_code[:, cnt] -> [B]; keep [B,1] columns so its cat works
    class _G:
        def __init__(self, t): self.t = t
        def __getitem__(self, key): return self.t[:, key[1]:key[1] + 1]
    ct, cb, lg, margin = run_sampling(m, tiny_cls, 3, 4, 8, noise[:8], (None, None), (None, None), (1.0, 1.0), given_top=_G(given))
    assert (ct == given.numpy()).all()
    out['given_top'], out['given_codes_bot'], out['given_logits'] = given.numpy(), cb, lg
    np.savez_compressed(os.path.join(OUT, 'g4_tiny_cls.npz'), **out)
    print('g4_tiny_cls ok')

    tiny_red = Stage2Spec(embed_dim=128, n_layers=2, n_heads=4, n_layers_depth=2, vocab_top=512, vocab_bot=512,
                          vocab_txt=64, ctx_len_img=64, ctx_len_txt=16, n_classes=0, cond=0, embedding=1)
    m, shapes = build_stage2(tiny_red, seed=4)
    noise = synth.exp_noise(6, 16, 3, 512)
    ct, cb, lg, margin = run_sampling(m, tiny_red, None, 3, 16, noise, (100, 100), (0.95, 0.95), (1.0, 0.9))
    np.savez_compressed(os.path.join(OUT, 'g3_tiny_reduce_uncond.npz'), spec=spec_json(tiny_red), weight_seed=4,
                        noise_seed=6, B=3, n_steps=16, codes_top=ct, codes_bot=cb, logits=lg[[0, 1, 15]],
                        keep_steps=np.array([0, 1, 15]), margin=margin, top_k=100, top_p=0.95, temps=np.array([1.0, 0.9]),
                        param_shapes=json.dumps({k: list(v) for k, v in shapes.items()}))
    print(f'g3_tiny_reduce_uncond ok, margin {margin:.6f}')

    tiny_txt = Stage2Spec(embed_dim=128, n_layers=2, n_heads=4, n_layers_depth=4, vocab_top=512, vocab_bot=512,
                          vocab_txt=64, ctx_len_img=64, ctx_len_txt=16, n_classes=0, cond=2, embedding=0)
    m, shapes = build_stage2(tiny_txt, seed=5)
    noise = synth.exp_noise(7, 12, 3, 512)
    txt = synth.text_ids(8, 3, 16, 64)
    ct, cb, lg, margin = run_sampling(m, tiny_txt, torch.from_numpy(txt), 3, 12, noise, (None, None), (None, None), (1.0, 1.0))
    np.savez_compressed(os.path.join(OUT, 'g3_tiny_txt.npz'), spec=spec_json(tiny_txt), weight_seed=5, noise_seed=7,
                        text_seed=8, B=3, n_steps=12, codes_top=ct, codes_bot=cb, logits=lg[[0, 1, 11]],
                        keep_steps=np.array([0, 1, 11]), margin=margin,
                        param_shapes=json.dumps({k: list(v) for k, v in shapes.items()}))
    print(f'g3_tiny_txt ok, margin {margin:.6f}')

    # ---------------------------------------------------------------- G5: HQ-VAE decode_code
    s1_64 = Stage1Spec(ch=32, ch_mult=[1, 2], num_res_blocks=2, attn_resolutions=[16], resolution=64, z_channels=32,
                       embed_dim=16, n_embed=64)
    g, shapes = build_stage1(s1_64, seed=21)
    r = np.random.default_rng(22)
    code_t = r.integers(0, 64, (2, 8, 8))
    code_b = r.integers(0, 64, (2, 16, 16))
    px = g.decode_code(torch.from_numpy(code_t), torch.from_numpy(code_b)).numpy()
    px_t = g.decode_code(torch.from_numpy(code_t[:1]), None).numpy()
    px_b = g.decode_code(None, torch.from_numpy(code_b[:1])).numpy()
    # a few intermediate tensors of sample 0 to localise failures
    qt = g.quantize_t.get_codebook_entry(torch.from_numpy(code_t[:1])).permute(0, 3, 1, 2)
    qb = g.quantize_b.get_codebook_entry(torch.from_numpy(code_b[:1])).permute(0, 3, 1, 2)
    z = g.post_quant_conv_b(torch.cat([g.upsample_t(qt), qb], dim=1))
    h0 = g.decoder.conv_in(z)
    h1 = g.decoder.mid.block_1(h0)
    h2 = g.decoder.mid.attn_1(h1)
    np.savez_compressed(os.path.join(OUT, 'g5_decode_64.npz'), spec=spec_json(s1_64), weight_seed=21, code_t=code_t,
                        code_b=code_b, pixels=px, pixels_top_only=px_t, pixels_bot_only=px_b, z=z.numpy(),
                        conv_in=h0.numpy(), mid_block_1=h1.numpy(), mid_attn_1=h2.numpy(),
                        param_shapes=json.dumps({k: list(v) for k, v in shapes.items()}))
    print('g5_decode_64 ok', px.shape, float(np.abs(px).max()))

    s1_256 = Stage1Spec(ch=32, ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[16], resolution=256,
                        z_channels=64, embed_dim=32, n_embed=128)
    g, shapes = build_stage1(s1_256, seed=23)
    r = np.random.default_rng(24)
    code_t = r.integers(0, 128, (1, 8, 8))
    code_b = r.integers(0, 128, (1, 16, 16))
    px = g.decode_code(torch.from_numpy(code_t), torch.from_numpy(code_b)).numpy()
    np.savez_compressed(os.path.join(OUT, 'g5_decode_256.npz'), spec=spec_json(s1_256), weight_seed=23, code_t=code_t,
                        code_b=code_b, pixels=px.astype(np.float32),
                        param_shapes=json.dumps({k: list(v) for k, v in shapes.items()}))
    print('g5_decode_256 ok', px.shape, float(np.abs(px).max()))

    # ---------------------------------------------------------------- G6: index maps
    ct = torch.arange(2 * 64).reshape(2, 64)
    cb = torch.arange(2 * 64 * 4).reshape(2, 64, 4)
    np.savez_compressed(os.path.join(OUT, 'g6_index_maps.npz'),
                        codes_top=ct.numpy(), codes_bot=cb.numpy(),
                        grid_top=rearrange(ct, 'B (H W) -> B H W', H=8).numpy(),
                        grid_bot=rearrange(cb, 'B (H W) (kerH kerW) -> B (H kerH) (W kerW)', H=8, kerH=2).numpy(),
                        pixel_shuffle_in=np.arange(2 * 8 * 3 * 3, dtype=np.float32).reshape(2, 8, 3, 3),
                        pixel_shuffle_out=torch.nn.PixelShuffle(2)(
                            torch.arange(2 * 8 * 3 * 3, dtype=torch.float32).reshape(2, 8, 3, 3)).numpy())
    print('g6_index_maps ok')


if __name__ == '__main__':
    main()
