#!/usr/bin/env python3
"""Fixtures G7/G8: the three-level path of the reference.

Container-only, like tools/gen_golden.py (whose import shims it reuses): builds the reference's ``HQTransformer``
('parallel-add' -- and, for G7b, 'parallel', 'parallel-reduce' and 'top2mid2bot' --, transformer1 embedding) and ``HQVAEGenerator`` (code_levels = 3) with weights derived from
numpy.default_rng by state-dict name (hqtransformer_amd/synth.py, 'fixture' profile), runs ``sampling_hqtransformer`` with
torch.multinomial replaced by argmax(p / q) on external Exp(1) noise, and ``decode_code([t, m, b])``; stores only inputs
and outputs:
  tests/golden/g7_l3_tiny_cls.npz   codes of all three levels for 64 positions, B = 3, two sampler settings, logits of 4 positions
  tests/golden/g7_l3_tiny_cls_parallel.npz, g7_l3_tiny_cls_parallel_reduce.npz, g7_l3_tiny_cls_top2mid2bot.npz   the same for the three other decoding types, 24 positions
  tests/golden/g8_l3_decode.npz     3-level decode_code pixels (all levels; top only; bottom only) on a 64-pixel decoder
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (installs the import shims)
import torch  # noqa: E402
from hqvae.models.stage2 import hqtransformer as ref_hqt  # noqa: E402
from hqvae.models.stage2.hqtransformer import HQTransformer  # noqa: E402
from hqvae.models.stage1.generator import HQVAEGenerator  # noqa: E402
from hqvae.utils import sampling as ref_sampling  # noqa: E402

from hqtransformer_amd import synth  # noqa: E402
from hqtransformer_amd.spec import Stage1Spec, Stage2Spec, stage1_is_encoder_key, stage1_is_ignored as _s1_ign  # noqa: E402


def stage1_is_ignored(k):          # decode-side fixtures: the encoder tensors keep the reference's own initialisation
    return _s1_ign(k) or stage1_is_encoder_key(k)



def build_stage2_l3(spec: Stage2Spec, seed: int):
    hp = G.AD(embed_dim=spec.embed_dim, n_layers=spec.n_layers, n_heads=spec.n_heads, n_dense_layers=spec.n_layers,
              ctx_len=None, ctx_len_img=spec.ctx_len_img, ctx_len_txt=spec.ctx_len_txt, embd_pdrop=0.0,
              resid_pdrop=0.1, attn_pdrop=0.0, mlp_bias=True, attn_bias=True, gelu_use_approx=spec.gelu_approx,
              use_head_txt=True, n_classes=spec.n_classes, causal_attn=None, embedding_type='transformer1',
              position_embedding='1d', bottom_head_type='linear', use_random_order=False, rate_random_order=1.0)
    hp_dec = None
    if spec.n_layers_depth != 4:
        import copy
        hp_dec = copy.deepcopy(hp)
        hp_dec.n_layers = spec.n_layers_depth
    m = HQTransformer([spec.vocab_top] * 3, spec.vocab_txt, spec.depth_decoding, spec.cond == 1, spec.cond == 2, hp, hp_dec)
    sd = {k: torch.from_numpy(v) for k, v in synth.stage2_weights(spec, seed, 'fixture').items()}
    ref_shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    mine = {k: tuple(v.shape) for k, v in sd.items()}
    assert ref_shapes == mine, (sorted(set(ref_shapes) ^ set(mine)), [k for k in ref_shapes if k in mine and ref_shapes[k] != mine[k]])
    m.load_state_dict(sd, strict=True)
    return m.eval(), ref_shapes


def build_stage1_l3(spec: Stage1Spec, seed: int):
    hp = G.AD(double_z=False, z_channels=spec.z_channels, resolution=spec.resolution, in_channels=3, out_ch=spec.out_ch,
              ch=spec.ch, ch_mult=list(spec.ch_mult), num_res_blocks=spec.num_res_blocks,
              attn_resolutions=list(spec.attn_resolutions), pdrop=0.0, use_init_downsample=spec.use_init_downsample,
              use_mid_block=spec.use_mid_block, use_attn=spec.use_attn)
    aux = G.AD(upsample='pixelshuffle', shared_codebook=False, bottom_start=10 ** 11, decoding_type='concat',
               restart_unused_codes=None, code_levels=3)
    g = HQVAEGenerator([spec.n_embed] * 3, spec.embed_dim, True, hp, aux)
    sd = {k: torch.from_numpy(v) for k, v in synth.stage1_weights(spec, seed, 'fixture').items()}
    ref_shapes = {k: tuple(v.shape) for k, v in g.state_dict().items() if not stage1_is_ignored(k)}
    mine = {k: tuple(v.shape) for k, v in sd.items()}
    assert ref_shapes == mine, (sorted(set(ref_shapes) ^ set(mine)), [k for k in ref_shapes if k in mine and ref_shapes[k] != mine[k]])
    missing, unexpected = g.load_state_dict(sd, strict=False)
    assert not unexpected and all(stage1_is_ignored(k) for k in missing)
    return g.eval(), ref_shapes


def run_sampling_l3(model, B, cond, n_steps, noise, top_k, top_p, temps):
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
    ref_hqt.cutoff_topk_logits = topk_spy
    torch.multinomial = fake_multinomial
    try:
        codes = ref_sampling.sampling_hqtransformer(model, num_candidates=B, cond=cond, top_k=list(top_k), top_p=list(top_p),
                                                    softmax_temperature=list(temps), is_tqdm=False, use_fp16=True,
                                                    max_seq_len=n_steps, model_stage1=None)
    finally:
        torch.multinomial = real_mn
        ref_hqt.cutoff_topk_logits = real_topk
    lg = np.stack(logits_log).reshape(n_steps, 21, B, -1)
    return [c.numpy() for c in codes], lg.astype(np.float32), float(min(margins))


def main():
    os.makedirs(G.OUT, exist_ok=True)
    # ---------------------------------------------------------------- G7: full 3-level sampling, tiny config
    spec = Stage2Spec(embed_dim=128, n_layers=3, n_heads=4, n_layers_depth=2, vocab_top=512, vocab_bot=512, vocab_txt=64,
                      ctx_len_img=64, ctx_len_txt=16, n_classes=10, cond=1, embedding=0, levels=3)
    model, shapes = build_stage2_l3(spec, 61)
    B, n = 3, 64
    noise = np.maximum(np.random.default_rng([62, 0x9e3779b9]).standard_exponential((n, 21, B, spec.vocab_top), dtype=np.float32), np.float32(1e-30))
    settings = [((None, None, None), (None, None, None), (1.0, 1.0, 1.0)), ((100, 50, 20), (1.0, 0.9, None), (1.0, 0.9, 0.8))]
    keep = [0, 1, 31, 63]
    out = dict(spec=G.spec_json(spec), weight_seed=np.int64(61), noise_seed=np.int64(62), B=np.int64(B), n_steps=np.int64(n),
               settings=json.dumps(settings), keep_steps=np.array(keep), ref_shapes=json.dumps({k: list(v) for k, v in shapes.items()}))
    for si, (tk, tp, T) in enumerate(settings):
        codes, lg, margin = run_sampling_l3(model, B, 7, n, noise, tk, tp, T)
        scale = np.array([T[0]] + [T[1]] * 4 + [T[2]] * 16, np.float32)[None, :, None, None]
        out[f'codes0_{si}'], out[f'codes1_{si}'], out[f'codes2_{si}'] = codes[0], codes[1], codes[2]
        out[f'logits_{si}'] = (lg * scale)[keep]                       # raw (pre-temperature) logits
        out[f'margin_{si}'] = np.float64(margin)
        print('G7 setting', si, 'shapes', [c.shape for c in codes], 'min winner/runner-up ratio', margin)
    np.savez_compressed(os.path.join(G.OUT, 'g7_l3_tiny_cls.npz'), **out)
    print('g7_l3_tiny_cls ok', os.path.getsize(os.path.join(G.OUT, 'g7_l3_tiny_cls.npz')), 'bytes')

    # ---------------------------------------------------------------- G7b: the other decoding types whose three-level sampling runs in the reference
    for dd in ('parallel', 'parallel-reduce', 'top2mid2bot'):
        vspec = Stage2Spec(embed_dim=128, n_layers=2, n_heads=4, n_layers_depth=2, vocab_top=512, vocab_bot=512, vocab_txt=64,
                           ctx_len_img=64, ctx_len_txt=16, n_classes=10, cond=1, embedding=0, levels=3, depth_decoding=dd)
        vmodel, vshapes = build_stage2_l3(vspec, 71)
        vB, vn = 3, 24
        vnoise = np.maximum(np.random.default_rng([72, 0x9e3779b9]).standard_exponential((vn, 21, vB, vspec.vocab_top), dtype=np.float32), np.float32(1e-30))
        vset = ((60, 40, 30), (1.0, 0.95, None), (1.0, 0.9, 0.8))
        vkeep = [0, 1, 23]
        codes, lg, margin = run_sampling_l3(vmodel, vB, 4, vn, vnoise, *vset)
        scale = np.array([vset[2][0]] + [vset[2][1]] * 4 + [vset[2][2]] * 16, np.float32)[None, :, None, None]
        name = 'g7_l3_tiny_cls_' + dd.replace('-', '_') + '.npz'
        np.savez_compressed(os.path.join(G.OUT, name), spec=G.spec_json(vspec), weight_seed=np.int64(71), noise_seed=np.int64(72), B=np.int64(vB),
                            n_steps=np.int64(vn), cond=np.int64(4), settings=json.dumps([vset]), keep_steps=np.array(vkeep), codes0_0=codes[0], codes1_0=codes[1],
                            codes2_0=codes[2], logits_0=(lg * scale)[vkeep], margin_0=np.float64(margin),
                            ref_shapes=json.dumps({k: list(v) for k, v in vshapes.items()}))
        print(name, 'ok', [c.shape for c in codes], 'min winner/runner-up ratio', margin, os.path.getsize(os.path.join(G.OUT, name)), 'bytes')

    # ---------------------------------------------------------------- G8: 3-level decode_code
    s1 = Stage1Spec(ch=32, ch_mult=[1, 2, 4], num_res_blocks=1, attn_resolutions=[8], resolution=64, z_channels=32, embed_dim=16,
                    n_embed=256, code_levels=3)
    gen, shapes1 = build_stage1_l3(s1, 63)
    r = np.random.default_rng(64)
    zr = s1.z_res                                                        # 8 -> top 2x2, mid 4x4, bottom 8x8
    ct, cm, cb = r.integers(0, 256, (2, zr // 4, zr // 4)), r.integers(0, 256, (2, zr // 2, zr // 2)), r.integers(0, 256, (2, zr, zr))
    with torch.no_grad():
        px = gen.decode_code([torch.from_numpy(ct), torch.from_numpy(cm), torch.from_numpy(cb)]).numpy()
        px_top = gen.decode_code([torch.from_numpy(ct[:1]), None, None]).numpy()
        px_bot = gen.decode_code([None, None, torch.from_numpy(cb[:1])]).numpy()
    np.savez_compressed(os.path.join(G.OUT, 'g8_l3_decode.npz'), spec=G.spec_json(s1), weight_seed=np.int64(63), code_t=ct, code_m=cm,
                        code_b=cb, pixels=px, pixels_top_only=px_top, pixels_bot_only=px_bot,
                        ref_shapes=json.dumps({k: list(v) for k, v in shapes1.items()}))
    print('g8_l3_decode ok', px.shape, float(np.abs(px).max()), os.path.getsize(os.path.join(G.OUT, 'g8_l3_decode.npz')), 'bytes')


if __name__ == '__main__':
    main()
