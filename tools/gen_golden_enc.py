#!/usr/bin/env python3
"""Generate the encode-side fixtures (tests/golden/g9_encode_*.npz) by running the REFERENCE's own
``SimRQGAN2Generator.encode`` / ``HQVAEGenerator.encode`` (CPU, fp32) in the build container.

Container-only, like tools/gen_golden.py: it imports /root/reference.  Only inputs and expected outputs are committed;
weights come from hqtransformer_amd.synth on both sides (keyed by state-dict name), images from ``synth_images``.

    python tools/gen_golden_enc.py
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402  (installs the import shims)
import torch  # noqa: E402
from hqvae.models.stage1.generator import HQVAEGenerator, SimRQGAN2Generator  # noqa: E402

from hqtransformer_amd import synth  # noqa: E402
from hqtransformer_amd.spec import Stage1Spec, stage1_is_ignored  # noqa: E402


def synth_images(seed: int, B: int, R: int) -> np.ndarray:
    """Smooth random images in [-1, 1] (a few low-frequency waves plus a little noise): fp32 [B, 3, R, R]."""
    r = np.random.default_rng([seed, 77])
    yy, xx = np.meshgrid(np.linspace(0, 1, R, dtype=np.float32), np.linspace(0, 1, R, dtype=np.float32), indexing='ij')
    img = np.zeros((B, 3, R, R), np.float32)
    for b in range(B):
        for c in range(3):
            for _ in range(4):
                fx, fy, ph = r.uniform(0.5, 6.0), r.uniform(0.5, 6.0), r.uniform(0, 2 * np.pi)
                img[b, c] += r.uniform(0.1, 0.4) * np.sin(2 * np.pi * (fx * xx + fy * yy) + ph).astype(np.float32)
    img += 0.05 * r.standard_normal(img.shape).astype(np.float32)
    return np.clip(img, -1.0, 1.0).astype(np.float32)


def build(spec: Stage1Spec, seed: int):
    hp = G.AD(double_z=False, z_channels=spec.z_channels, resolution=spec.resolution, in_channels=3, out_ch=spec.out_ch,
              ch=spec.ch, ch_mult=list(spec.ch_mult), num_res_blocks=spec.num_res_blocks,
              attn_resolutions=list(spec.attn_resolutions), pdrop=0.0, use_init_downsample=spec.use_init_downsample,
              use_mid_block=spec.use_mid_block, use_attn=spec.use_attn)
    aux = G.AD(upsample='pixelshuffle', shared_codebook=False, bottom_start=10 ** 11, decoding_type='concat',
               restart_unused_codes=None, code_levels=3 if spec.code_levels == 3 else None)
    if spec.code_levels == 3:
        g = HQVAEGenerator([spec.n_embed] * 3, spec.embed_dim, True, hp, aux)
    else:
        g = SimRQGAN2Generator(spec.n_embed, spec.embed_dim, True, hp, aux)
    sd = {k: torch.from_numpy(v) for k, v in synth.stage1_weights(spec, seed, 'fixture', encoder=True).items()}
    ref_shapes = {k: tuple(v.shape) for k, v in g.state_dict().items() if not stage1_is_ignored(k)}
    mine = {k: tuple(v.shape) for k, v in sd.items()}
    assert ref_shapes == mine, (sorted(set(ref_shapes) ^ set(mine)), [k for k in ref_shapes if k in mine and ref_shapes[k] != mine[k]])
    missing, unexpected = g.load_state_dict(sd, strict=False)
    assert not unexpected and all(stage1_is_ignored(k) for k in missing), (missing, unexpected)
    return g.eval(), ref_shapes


def code_margin(resid: torch.Tensor, emb: torch.Tensor) -> float:
    """Smallest gap between the best and the second-best squared distance over all rows (float64): how far the fixture is
    from an argmin tie."""
    z = resid.permute(0, 2, 3, 1).reshape(-1, resid.shape[1]).double()
    d = (z ** 2).sum(1, keepdim=True) + (emb.double() ** 2).sum(1)[None] - 2 * z @ emb.double().T
    top2 = torch.topk(d, 2, dim=1, largest=False).values
    return float((top2[:, 1] - top2[:, 0]).min())


def main():
    os.makedirs(G.OUT, exist_ok=True)
    torch.set_grad_enabled(False)
    cases = [
        # the 4x4 stride-2 conv_in + two Downsample levels + mid attention; attention never fires inside `down` (reference quirk)
        ('g9_encode_64', Stage1Spec(ch=32, ch_mult=[1, 2], num_res_blocks=2, attn_resolutions=[16], resolution=64, z_channels=32,
                                    embed_dim=16, n_embed=64), 71, 3),
        # 3x3 conv_in (no initial downsample): attention inside the last `down` level (label 16 == real 16)
        ('g9_encode_64_noinit', Stage1Spec(ch=32, ch_mult=[1, 1, 2], num_res_blocks=1, attn_resolutions=[16], resolution=64, z_channels=32,
                                           embed_dim=16, n_embed=128, use_init_downsample=False), 73, 2),
        # three code levels (HQVAEGenerator)
        ('g9_encode_64_l3', Stage1Spec(ch=32, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[16], resolution=64, z_channels=32,
                                       embed_dim=16, n_embed=96, code_levels=3), 75, 2),
    ]
    for name, spec, seed, B in cases:
        g, shapes = build(spec, seed)
        x = synth_images(seed + 1, B, spec.resolution)
        xt = torch.from_numpy(x)
        h_enc = g.encoder(xt)
        h = g.quant_conv_b(h_enc)
        out = dict(spec=G.spec_json(spec), weight_seed=seed, image_seed=seed + 1, B=B, pixels=x, h=h.numpy(),
                   conv_in=g.encoder.conv_in(xt).numpy()[:1], encoder_out=h_enc.numpy(),
                   param_shapes=json.dumps({k: list(v) for k, v in shapes.items()}))
        if spec.code_levels == 3:
            quant, diffs, codes, resids = g.encode(xt)
            embs = [q.embedding for q in g.quantizers]
            out['recon'] = quant.numpy()
            h_maps = [h]
            for d in g.downsamples:
                h_maps.insert(0, d(h_maps[0]))
            all_resids = [h_maps[0]] + list(resids)
            for l in range(3):
                out[f'code_{l}'] = codes[l].numpy()
                out[f'diff_{l}'] = np.float32(diffs[l])
                out[f'resid_{l}'] = all_resids[l].numpy()
            margins = [code_margin(all_resids[l], embs[l]) for l in range(3)]
            dec = g.decode(quant).numpy()
        else:
            quant_t, quant_b, diff_t, diff_b, (code_t, code_b, h_b) = g.encode(xt)
            out.update(code_0=code_t.numpy(), code_1=code_b.numpy(), quant_0=quant_t.numpy(), quant_1=quant_b.numpy(),
                       diff_0=np.float32(diff_t), diff_1=np.float32(diff_b), resid_0=g.down_t(h).numpy(), resid_1=h_b.numpy())
            ct, cb = g.get_codes(xt)
            assert (ct == code_t).all() and (cb == code_b).all()
            margins = [code_margin(g.down_t(h), g.quantize_t.embedding), code_margin(h_b, g.quantize_b.embedding)]
            dec = g.decode(quant_t, quant_b).numpy()
        out['margins'] = np.array(margins)
        out['reconstruction'] = dec.astype(np.float32)          # decode(encode(x)): the `dec` of forward() in eval mode
        np.savez_compressed(os.path.join(G.OUT, name + '.npz'), **out)
        print(name, 'ok: h', tuple(h.shape), 'margins', ['%.3g' % m for m in margins], 'codes', [int(out[f'code_{l}'].size) for l in range(len(margins))])


if __name__ == '__main__':
    main()
