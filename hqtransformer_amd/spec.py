"""Flat description of the two networks on the sampling path and of every tensor they read.

Parameter names are the reference's state-dict keys (stage-2 keys are
created at ``hqvae/models/stage2/hierarchical_ar.py:64-78,101-103,120,134-144,156-167,205-209``,
stage-1 keys at ``hqvae/models/stage1/generator.py:190-254`` and
``hqvae/models/stage1/modules/layers.py:300-383``) so a reference checkpoint loads by name.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field, asdict
from typing import Dict, List, Tuple

COND_NONE, COND_CLS, COND_TXT = 0, 1, 2
EMB_TRANSFORMER1, EMB_REDUCE = 0, 1
# HQTransformer.decoding_type values whose three-level SAMPLING works in the reference and is built here (index = hqt_config.depth_decoding)
DEPTH_DECODINGS = ('parallel-add', 'parallel', 'parallel-reduce', 'top2mid2bot')


@dataclass
class Stage2Spec:
    embed_dim: int
    n_layers: int
    n_heads: int
    n_layers_depth: int
    vocab_top: int
    vocab_bot: int
    vocab_txt: int
    ctx_len_img: int
    ctx_len_txt: int
    n_classes: int
    cond: int                 # COND_*
    embedding: int            # EMB_*
    gelu_approx: bool = False
    ratio_bot2top: int = 4
    levels: int = 2           # code levels: 2 = iHQGPT (1 + 4 codes per position), 3 = HQTransformer 'parallel*' (1 + 4 + 16)
    depth_decoding: str = 'parallel-add'      # three levels: HQTransformer.decoding_type, one of DEPTH_DECODINGS (hqtransformer.py:105-157,526-551)

    @property
    def codes_per_pos(self) -> int:  # hqtransformer.py:187-194
        return 1 + 4 + (16 if self.levels == 3 else 0)

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.n_heads

    @property
    def idx_pred(self) -> int:  # hierarchical_ar.py:66,74,78
        return self.ctx_len_txt if self.cond == COND_TXT else 0


@dataclass
class Stage1Spec:
    ch: int
    ch_mult: List[int]
    num_res_blocks: int
    attn_resolutions: List[int]
    resolution: int
    z_channels: int
    embed_dim: int            # bottom codebook dim; top codebook dim = 4 * embed_dim (pixelshuffle 2)
    n_embed: int
    out_ch: int = 3
    use_init_downsample: bool = True
    use_mid_block: bool = True
    use_attn: bool = True
    code_levels: int = 2      # 2 = SimRQGAN2Generator (concat), 3 = HQVAEGenerator (additive pixel-shuffle pyramid)

    @property
    def z_res(self) -> int:   # layers.py:330-331
        n = len(self.ch_mult)
        return self.resolution // 2 ** (n if self.use_init_downsample else n - 1)


def stage2_spec_from_config(cfg) -> Stage2Spec:
    """Mirrors the argument plumbing of ``ImageGPT2.__init__`` (``hqvae/models/__init__.py:123-137``)."""
    s2 = cfg.stage2
    levels = 2
    if 'multilevel-hq' in s2.type:          # HQTransformer (hqvae/models/__init__.py:138-145)
        vs = list(s2.vocab_sizes_img)
        if len(vs) != 3 or len(set(vs)) != 1:
            raise NotImplementedError('multilevel-hq: three levels with one vocabulary size are built')
        if s2.decoding_type not in DEPTH_DECODINGS:
            raise NotImplementedError(f"decoding_type '{s2.decoding_type}': {', '.join(DEPTH_DECODINGS)} are built ('parallel-add' is the released "
                                      "level-3 config; 'tree', 'old-parallel' and 'parallel-add-reduce' cannot sample three levels in the "
                                      "reference either: hqtransformer.py:537-549)")
        levels = 3
    elif 'hq-transformer' not in s2.type:
        raise ValueError(f"stage2.type '{s2.type}' is not on the HQ-Transformer sampling path")
    else:
        model_type = s2.type.split('/')[-1] if '/' in s2.type else 'top2bot'
        if model_type != 'parallel':
            raise NotImplementedError(f"model_type '{model_type}': only 'parallel' is built (SURVEY.md §0 item 2)")
    hp = s2.hparams
    hp_dec = s2.hparams_dec
    if hp_dec is None:  # hierarchical_ar.py:150-153
        n_layers_depth, dec_dim, dec_heads = 4, hp.embed_dim, hp.n_heads
    else:
        n_layers_depth, dec_dim, dec_heads = hp_dec.n_layers, hp_dec.embed_dim, hp_dec.n_heads
    if dec_dim != hp.embed_dim or dec_heads != hp.n_heads:
        raise NotImplementedError('depth head with its own width is not used by any released config')
    if hp.embedding_type == 'reduce' and levels == 2:
        emb = EMB_REDUCE
    elif hp.embedding_type == 'transformer1':
        emb = EMB_TRANSFORMER1
    else:
        raise NotImplementedError(f"embedding_type '{hp.embedding_type}'")
    if hp.position_embedding != '1d' or hp.use_random_order:
        raise NotImplementedError('position_embedding 2d / use_random_order are not used by any released config')
    if levels == 2 and s2.ratio_bot2top != 4:
        raise NotImplementedError('ratio_bot2top != 4')
    if not (hp.mlp_bias and hp.attn_bias):
        raise NotImplementedError('bias-free blocks')
    cond = COND_CLS if s2.use_cls_cond else (COND_TXT if s2.use_txt_cond else COND_NONE)
    vocab = list(s2.vocab_sizes_img)[0] if levels == 3 else s2.vocab_size_img
    return Stage2Spec(levels=levels, embed_dim=hp.embed_dim, n_layers=hp.n_layers, n_heads=hp.n_heads,
                      n_layers_depth=n_layers_depth, vocab_top=vocab, vocab_bot=vocab,
                      vocab_txt=s2.vocab_size_txt, ctx_len_img=hp.ctx_len_img, ctx_len_txt=hp.ctx_len_txt,
                      n_classes=(hp.n_classes or 0), cond=cond, embedding=emb,
                      gelu_approx=bool(hp.gelu_use_approx), ratio_bot2top=(s2.ratio_bot2top or 4),
                      depth_decoding=(s2.decoding_type if levels == 3 else 'parallel-add'))


def stage1_spec_from_config(cfg) -> Stage1Spec:
    s1 = cfg.stage1
    if s1.type not in ('simrqgan2', 'hqvae'):
        raise NotImplementedError(f"stage1.type '{s1.type}': 'simrqgan2' and 'hqvae' are built")
    aux = s1.hparams_aux
    if aux is None or aux.upsample != 'pixelshuffle' or aux.decoding_type != 'concat':
        raise NotImplementedError('only upsample=pixelshuffle (kernel 2), decoding_type=concat is built')
    code_levels = 2
    if s1.type == 'hqvae':                  # HQVAEGenerator (generator.py:451-515): additive pyramid, decoding_type is not read by decode
        code_levels = int(aux.code_levels or 2)
        if code_levels != 3:
            raise NotImplementedError('hqvae: code_levels = 3 is built')
    hp = s1.hparams
    return Stage1Spec(ch=hp.ch, ch_mult=list(hp.ch_mult), num_res_blocks=hp.num_res_blocks,
                      attn_resolutions=list(hp.attn_resolutions), resolution=hp.resolution,
                      z_channels=hp.z_channels, embed_dim=s1.embed_dim, n_embed=s1.n_embed, out_ch=hp.out_ch,
                      use_init_downsample=bool(hp.use_init_downsample), use_mid_block=bool(hp.use_mid_block),
                      use_attn=bool(hp.use_attn), code_levels=code_levels)


def _block_shapes(prefix: str, D: int, out: Dict[str, Tuple[int, ...]]) -> None:
    for ln in ('ln1', 'ln2'):
        out[f'{prefix}.{ln}.weight'] = (D,)
        out[f'{prefix}.{ln}.bias'] = (D,)
    for lin in ('key', 'query', 'value', 'proj'):
        out[f'{prefix}.attn.{lin}.weight'] = (D, D)
        out[f'{prefix}.attn.{lin}.bias'] = (D,)
    out[f'{prefix}.mlp.0.weight'] = (4 * D, D)
    out[f'{prefix}.mlp.0.bias'] = (4 * D,)
    out[f'{prefix}.mlp.2.weight'] = (D, 4 * D)
    out[f'{prefix}.mlp.2.bias'] = (D,)


def stage2_param_shapes(s: Stage2Spec) -> 'OrderedDict[str, Tuple[int, ...]]':
    D = s.embed_dim
    out: 'OrderedDict[str, Tuple[int, ...]]' = OrderedDict()
    out['sos_depth'] = (1, 1, D)
    if s.cond == COND_CLS:
        out['sos.weight'] = (s.n_classes, D)
    elif s.cond == COND_TXT:
        out['tok_emb_txt.weight'] = (s.vocab_txt, D)
        out['pos_emb_txt.weight'] = (s.ctx_len_txt, D)
        out['head_txt.weight'] = (s.vocab_txt, D)
        out['ln_txt.weight'] = (D,)
        out['ln_txt.bias'] = (D,)
    else:
        out['sos'] = (1, 1, D)
    if s.levels == 3:                       # HQTransformer (hqtransformer.py:24-205); 'transformer1' has no emb_blocks
        for li in range(3):
            out[f'tok_emb_levels.{li}.weight'] = (s.vocab_top, D)
        out['pos_emb_emb.weight'] = (21, D)
        out['pos_emb_top.weight'] = (s.ctx_len_img, D)
        for i in range(s.n_layers):
            _block_shapes(f'blocks.{i}', D, out)
        out['ln_f.weight'] = (D,)
        out['ln_f.bias'] = (D,)
        for li in range(3):                 # 'reduce': one D-slice of a wider row per child position (hqtransformer.py:108-116)
            mult = (16 if li == 2 else 4) if 'reduce' in s.depth_decoding else 1
            out[f'tok_emb_depth_levels.{li}.weight'] = (s.vocab_top, mult * D)
        if s.depth_decoding == 'top2mid2bot':   # one table over the 21-token causal sequence (hqtransformer.py:131-135)
            out['pos_emb_depths.0.weight'] = (1 + 4 + 16, D)
        else:
            out['pos_emb_depths.0.weight'] = (4, D)
            out['pos_emb_depths.1.weight'] = (16, D)
        for j in range(s.n_layers_depth):
            _block_shapes(f'depths.{j}', D, out)
        for li in range(3):
            out[f'ln_levels.{li}.weight'] = (D,)
            out[f'ln_levels.{li}.bias'] = (D,)
        for li in range(3):
            out[f'head_levels.{li}.weight'] = (s.vocab_top, D)
        return out
    out['tok_emb_top.weight'] = (s.vocab_top, D)
    if s.embedding == EMB_REDUCE:
        out['tok_emb_bot.weight'] = (s.vocab_bot, D // s.ratio_bot2top)
    else:
        out['tok_emb_bot.weight'] = (s.vocab_bot, D)
        out['pos_emb_emb.weight'] = (s.ratio_bot2top + 1, D)
    out['pos_emb_top.weight'] = (s.ctx_len_img, D)
    for i in range(s.n_layers):
        _block_shapes(f'blocks.{i}', D, out)
    out['ln_f.weight'] = (D,)
    out['ln_f.bias'] = (D,)
    out['tok_emb_top_depth.weight'] = (s.vocab_top, D)
    out['tok_emb_bot_depth.weight'] = (s.vocab_bot, D)       # present in checkpoints, unused by 'parallel'
    out['pos_emb_depth.weight'] = (max(1 + 1, 5), D)          # hierarchical_ar.py:167 (len_seq_depth = 2)
    for j in range(s.n_layers_depth):
        _block_shapes(f'depths.{j}', D, out)
    out['ln_top.weight'] = (D,)
    out['ln_top.bias'] = (D,)
    out['head_top.weight'] = (s.vocab_top, D)
    out['ln_bot.weight'] = (D,)
    out['ln_bot.bias'] = (D,)
    out['head_bot.weight'] = (s.vocab_bot, D)
    return out


# Stage-2 tensors the sampling path never reads (kept out of device memory, accepted on load).
STAGE2_UNUSED = ('tok_emb_bot_depth.weight', 'tok_emb_depth_levels.2.weight', 'head_txt.weight', 'ln_txt.weight', 'ln_txt.bias')


def stage2_unused(s) -> tuple:
    """Checkpoint keys of THIS model the sampling path never reads: accepted by load_state_dict, never uploaded.  'top2mid2bot' feeds its
    21 causal sub-steps from tok_emb_levels (hqtransformer.py:718-725), so none of its tok_emb_depth_levels tables is read
    ([V, D] each; under 'parallel-reduce' level 2 alone is [V, 16 D] = 805 MB at V = 8192, D = 1536)."""
    extra = tuple(f'tok_emb_depth_levels.{li}.weight' for li in range(3)) if getattr(s, 'levels', 2) == 3 and getattr(s, 'depth_decoding', '') == 'top2mid2bot' else ()
    return STAGE2_UNUSED + extra


@dataclass
class DecoderLayer:
    """One step of ``Decoder.forward`` (layers.py:385-410) in execution order."""
    kind: str                  # 'conv3' | 'res' | 'attn' | 'upconv' | 'out'
    name: str                  # state-dict prefix
    cin: int
    cout: int
    res: int                   # input spatial size


def decoder_plan(s: Stage1Spec) -> List[DecoderLayer]:
    n = len(s.ch_mult)
    block_in = s.ch * s.ch_mult[n - 1]
    res = s.z_res
    plan = [DecoderLayer('conv3', 'decoder.conv_in', s.z_channels, block_in, res)]
    if s.use_mid_block:
        plan.append(DecoderLayer('res', 'decoder.mid.block_1', block_in, block_in, res))
        if s.use_attn:
            plan.append(DecoderLayer('attn', 'decoder.mid.attn_1', block_in, block_in, res))
        plan.append(DecoderLayer('res', 'decoder.mid.block_2', block_in, block_in, res))
    for lvl in reversed(range(n)):
        block_out = s.ch * s.ch_mult[lvl]
        for b in range(s.num_res_blocks + 1):
            plan.append(DecoderLayer('res', f'decoder.up.{lvl}.block.{b}', block_in, block_out, res))
            block_in = block_out
            if res in s.attn_resolutions and s.use_attn:
                plan.append(DecoderLayer('attn', f'decoder.up.{lvl}.attn.{b}', block_in, block_in, res))
        if lvl != 0 or s.use_init_downsample:
            plan.append(DecoderLayer('upconv', f'decoder.up.{lvl}.upsample.conv', block_in, block_in, res))
            res *= 2
    plan.append(DecoderLayer('out', 'decoder', block_in, s.out_ch, res))
    return plan


def stage1_param_shapes(s: Stage1Spec) -> 'OrderedDict[str, Tuple[int, ...]]':
    """Tensors ``decode_code`` reads (generator.py:312-367): two codebooks, the 1x1 post-quant conv, the decoder."""
    out: 'OrderedDict[str, Tuple[int, ...]]' = OrderedDict()
    if s.code_levels == 3:                  # HQVAEGenerator (generator.py:478-506): dims E*16, E*4, E; 1x1 conv from E channels
        for ci in range(3):
            out[f'quantizers.{ci}.embedding'] = (s.n_embed, s.embed_dim * 4 ** (2 - ci))
        out['post_quant_conv_b.weight'] = (s.z_channels, s.embed_dim, 1, 1)
    else:
        out['quantize_t.embedding'] = (s.n_embed, s.embed_dim * 4)
        out['quantize_b.embedding'] = (s.n_embed, s.embed_dim)
        out['post_quant_conv_b.weight'] = (s.z_channels, 2 * s.embed_dim, 1, 1)
    out['post_quant_conv_b.bias'] = (s.z_channels,)
    for l in decoder_plan(s):
        if l.kind in ('conv3', 'upconv'):
            out[f'{l.name}.weight'] = (l.cout, l.cin, 3, 3)
            out[f'{l.name}.bias'] = (l.cout,)
        elif l.kind == 'res':
            out[f'{l.name}.norm1.weight'] = (l.cin,)
            out[f'{l.name}.norm1.bias'] = (l.cin,)
            out[f'{l.name}.conv1.weight'] = (l.cout, l.cin, 3, 3)
            out[f'{l.name}.conv1.bias'] = (l.cout,)
            out[f'{l.name}.norm2.weight'] = (l.cout,)
            out[f'{l.name}.norm2.bias'] = (l.cout,)
            out[f'{l.name}.conv2.weight'] = (l.cout, l.cout, 3, 3)
            out[f'{l.name}.conv2.bias'] = (l.cout,)
            if l.cin != l.cout:
                out[f'{l.name}.nin_shortcut.weight'] = (l.cout, l.cin, 1, 1)
                out[f'{l.name}.nin_shortcut.bias'] = (l.cout,)
        elif l.kind == 'attn':
            out[f'{l.name}.norm.weight'] = (l.cin,)
            out[f'{l.name}.norm.bias'] = (l.cin,)
            for c in ('q', 'k', 'v', 'proj_out'):
                out[f'{l.name}.{c}.weight'] = (l.cin, l.cin, 1, 1)
                out[f'{l.name}.{c}.bias'] = (l.cin,)
        elif l.kind == 'out':
            out['decoder.norm_out.weight'] = (l.cin,)
            out['decoder.norm_out.bias'] = (l.cin,)
            out['decoder.conv_out.weight'] = (l.cout, l.cin, 3, 3)
            out['decoder.conv_out.bias'] = (l.cout,)
    return out


def encoder_plan(s: Stage1Spec) -> List[DecoderLayer]:
    """``Encoder.forward`` (stage1/modules/layers.py:270-297) in execution order; kinds 'in', 'res', 'attn', 'down', 'out'.
    The reference starts its ``curr_res`` bookkeeping at ``resolution`` even when conv_in already halves the image
    (layers.py:221), so with use_init_downsample the attention test sees twice the real feature-map size."""
    n = len(s.ch_mult)
    plan = [DecoderLayer('in', 'encoder.conv_in', 3, s.ch, s.resolution)]
    label = s.resolution
    res = s.resolution // 2 if s.use_init_downsample else s.resolution
    block_in = s.ch
    for lvl in range(n):
        block_out = s.ch * s.ch_mult[lvl]
        for b in range(s.num_res_blocks):
            plan.append(DecoderLayer('res', f'encoder.down.{lvl}.block.{b}', block_in, block_out, res))
            block_in = block_out
            if label in s.attn_resolutions and s.use_attn:
                plan.append(DecoderLayer('attn', f'encoder.down.{lvl}.attn.{b}', block_in, block_in, res))
        if lvl != n - 1:
            plan.append(DecoderLayer('down', f'encoder.down.{lvl}.downsample.conv', block_in, block_in, res))
            res //= 2
            label //= 2
    if s.use_mid_block:
        plan.append(DecoderLayer('res', 'encoder.mid.block_1', block_in, block_in, res))
        if s.use_attn:
            plan.append(DecoderLayer('attn', 'encoder.mid.attn_1', block_in, block_in, res))
        plan.append(DecoderLayer('res', 'encoder.mid.block_2', block_in, block_in, res))
    plan.append(DecoderLayer('out', 'encoder', block_in, s.z_channels, res))
    return plan


def _layer_shapes(l: DecoderLayer, out: 'OrderedDict[str, Tuple[int, ...]]') -> None:
    if l.kind == 'res':
        out[f'{l.name}.norm1.weight'] = (l.cin,)
        out[f'{l.name}.norm1.bias'] = (l.cin,)
        out[f'{l.name}.conv1.weight'] = (l.cout, l.cin, 3, 3)
        out[f'{l.name}.conv1.bias'] = (l.cout,)
        out[f'{l.name}.norm2.weight'] = (l.cout,)
        out[f'{l.name}.norm2.bias'] = (l.cout,)
        out[f'{l.name}.conv2.weight'] = (l.cout, l.cout, 3, 3)
        out[f'{l.name}.conv2.bias'] = (l.cout,)
        if l.cin != l.cout:
            out[f'{l.name}.nin_shortcut.weight'] = (l.cout, l.cin, 1, 1)
            out[f'{l.name}.nin_shortcut.bias'] = (l.cout,)
    elif l.kind == 'attn':
        out[f'{l.name}.norm.weight'] = (l.cin,)
        out[f'{l.name}.norm.bias'] = (l.cin,)
        for c in ('q', 'k', 'v', 'proj_out'):
            out[f'{l.name}.{c}.weight'] = (l.cin, l.cin, 1, 1)
            out[f'{l.name}.{c}.bias'] = (l.cin,)


def stage1_encoder_param_shapes(s: Stage1Spec) -> 'OrderedDict[str, Tuple[int, ...]]':
    """Tensors the encode side adds (generator.py:298-310): the Encoder and quant_conv_b (the codebooks are shared with
    decode_code).  In a reference state_dict these keys precede the decoder's; order here is execution order."""
    out: 'OrderedDict[str, Tuple[int, ...]]' = OrderedDict()
    for l in encoder_plan(s):
        if l.kind == 'in':
            ks = 4 if s.use_init_downsample else 3           # layers.py:212-216
            out['encoder.conv_in.weight'] = (l.cout, 3, ks, ks)
            out['encoder.conv_in.bias'] = (l.cout,)
        elif l.kind == 'down':
            out[f'{l.name}.weight'] = (l.cout, l.cin, 3, 3)
            out[f'{l.name}.bias'] = (l.cout,)
        elif l.kind == 'out':
            out['encoder.norm_out.weight'] = (l.cin,)
            out['encoder.norm_out.bias'] = (l.cin,)
            out['encoder.conv_out.weight'] = (l.cout, l.cin, 3, 3)
            out['encoder.conv_out.bias'] = (l.cout,)
        else:
            _layer_shapes(l, out)
    out['quant_conv_b.weight'] = (s.embed_dim, s.z_channels, 1, 1)
    out['quant_conv_b.bias'] = (s.embed_dim,)
    return out


def stage1_is_encoder_key(key: str) -> bool:
    return key.startswith('encoder.') or key.startswith('quant_conv_b.')


def stage1_is_ignored(key: str) -> bool:
    """Checkpoint keys of the training side (EMA statistics of the quantisers) that ``load_state_dict`` tolerates."""
    return key.endswith('.cluster_size') or key.endswith('.embedding_avg')


def work_per_image(s2: Stage2Spec, s1: Stage1Spec, n_pos: int) -> Dict[str, float]:
    """Algorithmic work used by bench.py's roofline: AR weight bytes per top position
    (bf16) and decoder FLOPs per image."""
    D, V = s2.embed_dim, s2.vocab_top
    blk = 12 * D * D + 13 * D
    sub, toks = (3, 21) if s2.levels == 3 else (2, 5)          # depth sub-steps (each streams the depth weights once), depth tokens
    ar_weight_bytes_per_pos = 2 * ((s2.n_layers + sub * s2.n_layers_depth) * blk + sub * D * V)
    ar_flops = 2 * n_pos * (s2.n_layers * 12 * D * D + s2.n_layers_depth * 12 * D * D * toks + toks * D * V)
    mac = s1.z_res ** 2 * (1 if s1.code_levels == 3 else 2) * s1.embed_dim * s1.z_channels
    for l in decoder_plan(s1):
        px = l.res * l.res
        if l.kind == 'conv3':
            mac += px * 9 * l.cin * l.cout
        elif l.kind == 'upconv':
            mac += 4 * px * 9 * l.cin * l.cout
        elif l.kind == 'res':
            mac += px * 9 * l.cin * l.cout + px * 9 * l.cout * l.cout
            if l.cin != l.cout:
                mac += px * l.cin * l.cout
        elif l.kind == 'attn':
            mac += 4 * px * l.cin * l.cin + 2 * px * px * l.cin
        elif l.kind == 'out':
            mac += px * 9 * l.cin * l.cout
    return dict(ar_weight_bytes_per_pos=float(ar_weight_bytes_per_pos), ar_flops=float(ar_flops),
                dec_flops=float(2 * mac))
