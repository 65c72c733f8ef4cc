"""Host-side logic of the drop-in surface (no GPU): config loading and defaults, spec derivation, state-dict
contract, harness accounting, code-grid maps, and the refusal to compute off-GPU."""
import os

import numpy as np
import pytest
import torch

from hqtransformer_amd import _lib
from hqtransformer_amd.config import get_base_config, load_config, merge, parse_dotlist
from hqtransformer_amd.measure_throughput import EXPERIMENT_DEFAULTS, iterations_per_loop
from hqtransformer_amd.models import ImageGPT2
from hqtransformer_amd.sampling import rearrange_codes, sampling_ihqgpt
from hqtransformer_amd.sampling_hqmodel import build_parser, remap_legacy_keys
from hqtransformer_amd.spec import (decoder_plan, stage1_param_shapes, stage1_spec_from_config, stage2_param_shapes,
                                    stage2_spec_from_config, work_per_image)
from tests.helpers import load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY = os.path.join(ROOT, 'configs', 'tiny-cls.yaml')
IMAGENET = os.path.join(ROOT, 'configs', 'imagenet-12l.yaml')


def test_defaults_the_yaml_files_omit():
    cfg = load_config(TINY)
    assert cfg.stage2.ratio_bot2top == 4 and cfg.stage2.vocab_size_txt == 16384          # config2.py:88,92
    assert cfg.stage2.hparams.position_embedding == '1d' and cfg.stage2.hparams.use_random_order is False
    assert cfg.stage2.hparams.ctx_len_txt == 64 and cfg.stage2.hparams_dec is None
    assert cfg.stage1.hparams.num_res_blocks == 2 and cfg.stage1.hparams.out_ch == 3
    assert cfg.stage1.hparams_aux.bottom_start == 100000000000
    base = get_base_config(False)
    assert base.stage2.hparams is None and base.stage1.hparams.ch_mult == [1, 1, 2, 2, 4]
    over = merge(base, {'stage2': {'hparams': {'embed_dim': 64}}})
    assert over.stage2.hparams.embed_dim == 64 and over.stage2.hparams.n_heads == 24     # Stage2Hparams defaults fill in


def test_imagenet_spec_matches_the_survey_numbers():
    cfg = load_config(IMAGENET)
    s2, s1 = stage2_spec_from_config(cfg), stage1_spec_from_config(cfg)
    assert (s2.embed_dim, s2.n_layers, s2.n_layers_depth, s2.n_heads, s2.vocab_top) == (1536, 12, 4, 24, 8192)
    n2 = sum(int(np.prod(s)) for s in stage2_param_shapes(s2).values())
    assert abs(n2 / 1e6 - 530.76) < 0.01
    dec = sum(int(np.prod(s)) for k, s in stage1_param_shapes(s1).items() if k.startswith('decoder'))
    assert abs(dec / 1e6 - 53.95) < 0.01
    w = work_per_image(s2, s1, 64)
    assert abs(w['ar_weight_bytes_per_pos'] / 1e9 - 1.184) < 0.002 and abs(w['dec_flops'] / 1e9 - 177.5) < 0.1
    plan = decoder_plan(s1)
    assert [l.kind for l in plan].count('attn') == 4 and plan[-1].res == 256 and plan[0].res == 16


def test_unsupported_configs_are_refused():
    cfg = load_config(TINY)
    cfg.stage2.type = 'hq-transformer/top2bot'
    with pytest.raises(NotImplementedError):
        stage2_spec_from_config(cfg)
    cfg = load_config(TINY)
    cfg.stage1.type = 'hqvae'
    with pytest.raises(NotImplementedError):
        stage1_spec_from_config(cfg)


def test_dotlist_like_omegaconf_from_cli():
    a = parse_dotlist(['model_path=x.yaml', 'batch_size=64', 'n_loop=3'], EXPERIMENT_DEFAULTS)
    assert a.batch_size == 64 and a.n_loop == 3 and a.warmup == 1 and a.top_resolution == 8 and a.model_path == 'x.yaml'
    with pytest.raises(KeyError):
        parse_dotlist(['nope=1'], EXPERIMENT_DEFAULTS)
    assert iterations_per_loop(50) == 20 and iterations_per_loop(64) == 16 and iterations_per_loop(1000) == 1


def test_sampling_hqmodel_arguments_and_legacy_keys():
    a = build_parser().parse_args(['-r', 'out', '-m', 'cfg.yaml'])
    assert (a.top_k, a.top_p, a.temperature, a.temperature_decay, a.batch_size, a.seed, a.num_classes) == (2048, 1.0, 1.0, 1.0, 50, 0, 1000)
    sd = {'generator.stage1XXXX.decoder.conv_in.weight': 1, 'stage2.ln_f.weight': 2}
    out = remap_legacy_keys(sd)
    assert 'stage2.ln_f.weight' in out and any(k.startswith('stage1.') for k in out)


def test_model_surface_and_state_dict_contract():
    model = ImageGPT2(load_config(TINY), seed=0).eval()
    s2 = model.stage2
    assert s2.use_cls_cond and not s2.use_txt_cond and s2.idx_pred == 0
    assert tuple(s2.sos.weight.shape) == (1000, 128)
    fx = load('g4_tiny_cls.npz')
    assert sum(p.numel() for p in s2.parameters()) > 0
    sd = model.state_dict()
    assert all(k.startswith(('stage1.', 'stage2.')) for k in sd)
    model.load_state_dict(sd, strict=True)
    extra = dict(sd)
    assert 'stage1.encoder.conv_in.weight' in sd and 'stage1.quant_conv_b.weight' in sd   # the encode side is part of the model
    extra['stage1.quantize_t.cluster_size'] = torch.zeros(1)
    model.load_state_dict(extra, strict=True)
    bad = dict(sd)
    bad.pop('stage2.ln_f.weight')
    with pytest.raises(RuntimeError):
        model.load_state_dict(bad, strict=True)
    bad = dict(sd)
    bad['stage2.ln_f.weight'] = torch.zeros(3)
    with pytest.raises(RuntimeError):
        model.load_state_dict(bad, strict=False)
    bad = dict(sd)
    bad['stage2.bogus'] = torch.zeros(1)
    with pytest.raises(RuntimeError):
        model.load_state_dict(bad, strict=True)


def test_no_cpu_compute_path():
    model = ImageGPT2(load_config(TINY), seed=0)
    with pytest.raises(_lib.HqtLibraryError):
        sampling_ihqgpt(model.stage2, 2, 3, max_seq_len=4)
    with pytest.raises(_lib.HqtLibraryError):
        model.stage1.decode_code(torch.zeros(1, 8, 8, dtype=torch.long), torch.zeros(1, 16, 16, dtype=torch.long))


def test_rearrange_codes_matches_einops_fixture():
    fx = load('g6_index_maps.npz')
    gt, gb = rearrange_codes(torch.from_numpy(fx['codes_top']), torch.from_numpy(fx['codes_bot']), 8)
    assert (gt.numpy() == fx['grid_top']).all() and (gb.numpy() == fx['grid_bot']).all()


def test_text_front_end(tmp_path):
    """Tokenizer front-end of the text-conditional path (hqvae/datasets/__init__.py:145-151): lower-casing char-BPE,
    [PAD] special token, padding and truncation to context_length -- on a tiny own vocabulary (the reference's 16k
    vocabulary is not copied into this repository)."""
    import json
    from hqtransformer_amd import text as T
    vocab = {'[UNK]': 0, 'a</w>': 1, 'c': 2, 'a': 3, 't</w>': 4, 'ca': 5, 'cat</w>': 6, 'd': 7, 'o': 8, 'g</w>': 9, 'do': 10,
             'dog</w>': 11, 'n': 12}
    (tmp_path / 'v.json').write_text(json.dumps(vocab))
    (tmp_path / 'm.txt').write_text('#version: 0.2\nc a\nca t</w>\nd o\ndo g</w>\n')
    tok = T.build_tokenizer(str(tmp_path / 'v.json'), str(tmp_path / 'm.txt'), context_length=6)
    pad = tok.token_to_id('[PAD]')
    assert pad == 13                                              # appended behind the vocabulary
    ids = T.encode(tok, ['A cat', 'a dog a cat a dog a cat a dog', 'zzz'])
    assert ids.dtype == torch.int64 and ids.tolist() == [[1, 6, pad, pad, pad, pad], [1, 11, 1, 6, 1, 11], [0, 0, 0, pad, pad, pad]]
    (tmp_path / 'val_list.txt').write_text('img/0.jpg\ta cat\nimg/1.jpg\ta dog\n\n')
    (tmp_path / 'plain.txt').write_text('a cat\na dog\n')
    assert T.read_captions(str(tmp_path / 'val_list.txt')) == ['a cat', 'a dog'] == T.read_captions(str(tmp_path / 'plain.txt'))
    assert T.find_reference_vocab(str(tmp_path)) is None


@pytest.mark.parametrize('name', ['imagenet-12l', 'imagenet-24l', 'imagenet-42l', 'ffhq-24l', 'cc15m-12l-txt'])
def test_released_configs_match_the_reference_state_dict(name):
    """Every released two-level stage-2 config (12 / 24 / 42 layers with the 6-layer depth head, FFHQ D = 1024 / 16 heads / 'reduce',
    CC-15M text): the YAML of this repository yields exactly the state-dict keys and shapes of the reference's own iHQGPT built
    from ITS YAML (fixture G11, tools/gen_released_shapes.py: the reference module on the meta device)."""
    import json
    want = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'g11_released_shapes.json')))[name]
    spec = stage2_spec_from_config(load_config(os.path.join(ROOT, 'configs', name + '.yaml')))
    got = {k: list(v) for k, v in stage2_param_shapes(spec).items()}
    ignore = {k for k in want['shapes'] if k not in got}
    from hqtransformer_amd.spec import STAGE2_UNUSED
    assert ignore <= set(STAGE2_UNUSED) | {k for k in ignore if k.endswith(('.mask', 'mask'))}, sorted(ignore)[:5]
    assert {k: want['shapes'][k] for k in got} == got
    n = sum(int(np.prod(v)) for v in got.values())
    unused = sum(int(np.prod(want['shapes'][k])) for k in ignore)
    assert n + unused == want['n_params']


def test_measure_throughput_txt_host_logic(tmp_path):
    """measure_throughput_txt counterpart, host side (no GPU): the reference's loop count (measure_throughput_txt/__main__.py:103), its
    Experiment keys, synthetic prompt batches inside the text vocabulary, and captions cycled through the BPE front-end."""
    import json
    from types import SimpleNamespace
    from hqtransformer_amd import measure_throughput_txt as mtt
    from hqtransformer_amd.config import parse_dotlist
    assert mtt.iterations_per_loop(50) == 20 and mtt.iterations_per_loop(64) == 16 and mtt.iterations_per_loop(1000) == 1
    args = parse_dotlist(['batch_size=8', 'top_resolution=8', 'dataset=cc3m'], mtt.EXPERIMENT_DEFAULTS)
    assert (args.batch_size, args.n_loop, args.warmup, args.top_k, args.top_p) == (8, 6, 1, 2048, 1.0)       # the reference's defaults
    spec = SimpleNamespace(ctx_len_txt=16, vocab_txt=300)
    gen = mtt.prompt_batches(args, spec)
    a, b = next(gen), next(gen)
    assert a.dtype.is_floating_point is False and tuple(a.shape) == (8, 16) and int(a.min()) >= 0 and int(a.max()) < 300 and not (a == b).all()
    vocab = {'[UNK]': 0, 'a</w>': 1, 'c': 2, 'a': 3, 't</w>': 4, 'ca': 5, 'cat</w>': 6, 'd': 7, 'o': 8, 'g</w>': 9, 'do': 10, 'dog</w>': 11}
    (tmp_path / 'v.json').write_text(json.dumps(vocab))
    (tmp_path / 'm.txt').write_text('#version: 0.2\nc a\nca t</w>\nd o\ndo g</w>\n')
    (tmp_path / 'caps.txt').write_text('x.jpg\ta cat\ny.jpg\ta dog\nz.jpg\ta cat a dog\n')
    args = parse_dotlist(['batch_size=4', f'captions={tmp_path / "caps.txt"}', f'tokenizer_vocab={tmp_path / "v.json"}',
                          f'tokenizer_merges={tmp_path / "m.txt"}'], mtt.EXPERIMENT_DEFAULTS)
    gen = mtt.prompt_batches(args, spec)
    c, d = next(gen), next(gen)
    assert tuple(c.shape) == (4, 16) and (c[0] == c[3]).all() and (d[0] == c[1]).all()        # 3 captions cycled over batches of 4
    import pytest
    with pytest.raises(ValueError):
        next(mtt.prompt_batches(parse_dotlist([f'captions={tmp_path / "caps.txt"}'], mtt.EXPERIMENT_DEFAULTS), spec))
