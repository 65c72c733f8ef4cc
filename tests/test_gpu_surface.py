"""The reference-shaped Python surface on the GPU: ImageGPT2 / sampling_ihqgpt / decode_code and the two harness
counterparts, checked against the CPU oracle built from the same state dict."""
import os
import pickle

import numpy as np
import pytest
import torch

from hqtransformer_amd import synth
from hqtransformer_amd.config import load_config, parse_dotlist
from hqtransformer_amd.models import ImageGPT2
from hqtransformer_amd.sampling import rearrange_codes, sampling_ihqgpt
from hqtransformer_amd.utils import set_seed
from oracle import hqt_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY = os.path.join(ROOT, 'configs', 'tiny-cls.yaml')


@pytest.fixture(scope='module')
def model():
    return ImageGPT2(load_config(TINY), seed=5).to('cuda').eval()


def test_sampling_ihqgpt_and_decode_code_match_the_oracle(model):
    s2, s1 = model.stage2.spec, model.stage1.spec
    w2 = {k: v.numpy() for k, v in model.stage2.state_dict().items()}
    w1 = {k: v.numpy() for k, v in model.stage1.state_dict().items()}
    B, n = 3, 64
    noise = synth.exp_noise(11, n, B, s2.vocab_top)
    ct, cb = sampling_ihqgpt(model.stage2, num_candidates=B, cond=417, top_k_top=100, top_p_top=0.9, top_k_bot=50,
                             top_p_bot=None, softmax_temperature=[1.0, 0.9], use_fp16=False, is_tqdm=False, max_seq_len=n,
                             noise=torch.from_numpy(noise))
    assert ct.dtype == torch.int64 and tuple(ct.shape) == (B, n) and tuple(cb.shape) == (B, n, 4) and ct.is_cuda
    want = O.OracleStage2(s2, w2).sample(np.full(B, 417), B, n, noise, (100, 50), (0.9, None), (1.0, 0.9))
    assert (ct.cpu().numpy() == want[0]).all() and (cb.cpu().numpy() == want[1]).all()
    gt, gb = rearrange_codes(ct, cb, 8)
    px = model.stage1.decode_code(gt, gb)                      # default precision = the reference's fp32 decode
    ref = O.OracleStage1(s1, w1).decode_code(*O.rearrange_codes(want[0], want[1], 8))
    assert tuple(px.shape) == (B, 3, 64, 64) and np.abs(px.cpu().numpy() - ref).max() <= 1e-4
    px_seq = model.stage1.decode_sequences(ct, cb, clamp01=True)
    assert np.abs(px_seq.cpu().numpy() - O.postprocess(ref)).max() <= 1e-4
    half = model.stage1.decode_code(gt[:1], None)
    assert np.abs(half.cpu().numpy() - O.OracleStage1(s1, w1).decode_code(gt[:1].cpu().numpy(), None)).max() <= 1e-4


def test_set_seed_controls_the_in_kernel_noise(model):
    def run():
        return sampling_ihqgpt(model.stage2, 4, 3, use_fp16=True, is_tqdm=False, max_seq_len=16)
    set_seed(0)
    a = run()
    set_seed(0)
    b = run()
    set_seed(1)
    c = run()
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and (a[0] != c[0]).any()
    with pytest.raises(IndexError):
        sampling_ihqgpt(model.stage2, 2, 1000, max_seq_len=4)          # class id out of range, as nn.Embedding raises


def test_measure_throughput_counterpart(capsys):
    from hqtransformer_amd import measure_throughput as mt
    args = parse_dotlist([f'model_path={TINY}', 'batch_size=500', 'n_loop=2', 'warmup=1'], mt.EXPERIMENT_DEFAULTS)
    out = mt.main(args)
    text = capsys.readouterr().out
    assert 'bs500, sampling loops 2-2' in text and 'transformer size:' in text and 'ms/sample (ar:' in text
    assert out['ms_per_sample'] > 0 and abs(out['ms_ar'] + out['ms_decode'] - out['ms_per_sample']) / out['ms_per_sample'] < 0.2
    # several iterations in flight: same loop accounting, phases are lane times
    args = parse_dotlist([f'model_path={TINY}', 'batch_size=250', 'n_loop=2', 'warmup=1', 'inflight=3'], mt.EXPERIMENT_DEFAULTS)
    out3 = mt.main(args)
    assert 'bs250, sampling loops 2-2' in capsys.readouterr().out and out3['ms_per_sample'] > 0
    # ... and merged into passes of 4 x 50 rows on 2 lanes (20 iterations per loop: 5 passes), phases measured per pass
    args = parse_dotlist([f'model_path={TINY}', 'batch_size=50', 'n_loop=2', 'warmup=1', 'inflight=2', 'merge=4'], mt.EXPERIMENT_DEFAULTS)
    outm = mt.main(args)
    assert 'bs50, sampling loops 2-2' in capsys.readouterr().out and outm['ms_per_sample'] > 0 and outm['ms_ar'] > 0 and outm['ms_decode'] > 0


def test_sampling_hqmodel_counterpart_writes_reference_formats(tmp_path):
    from hqtransformer_amd import sampling_hqmodel as sh
    sh.main(['-r', str(tmp_path), '-m', TINY, '--batch-size', '2', '--num-classes', '2', '--samples-per-class', '2',
             '--top-k', '100', '--top-resolution', '8'])
    for cls in (1, 2):
        px = pickle.load(open(tmp_path / f'samples_({cls}_0).pkl', 'rb'))
        tg = np.load(tmp_path / f'targets_({cls}_0).npz')['targets']
        assert px.dtype == np.float32 and px.shape == (2, 3, 64, 64) and px.min() >= 0 and px.max() <= 1
        assert tg.dtype == np.int64 and (tg == cls - 1).all()


def test_lanes_share_weights_and_are_bit_identical(model):
    """hqt_clone lanes (hqtransformer_amd.pipeline): a lane has its own KV cache / activations / graph cache over the
    parent's weights.  Codes (exact AND fast arithmetic) and pixels must not depend on the lane, whether the batches
    run one after another or interleaved on separate streams; the parent cannot be destroyed under a live clone."""
    from hqtransformer_amd.pipeline import InflightSampler
    B, n = 3, 16
    s2 = model.stage2.spec
    noise = torch.from_numpy(synth.exp_noise(21, n, B, s2.vocab_top))
    ref = {}
    for fp16 in (False, True):
        ct, cb = sampling_ihqgpt(model.stage2, B, 5, use_fp16=fp16, is_tqdm=False, max_seq_len=n, noise=noise)
        px = model.stage1.decode_sequences(torch.cat([ct] * 4, 1), torch.cat([cb] * 4, 1), precision='fast' if fp16 else 'exact')
        ref[fp16] = (ct.clone(), cb.clone(), px.clone())
        for lane in (1, 2):
            ct2, cb2 = sampling_ihqgpt(model.stage2, B, 5, use_fp16=fp16, is_tqdm=False, max_seq_len=n, noise=noise, lane=lane)
            px2 = model.stage1.decode_sequences(torch.cat([ct2] * 4, 1), torch.cat([cb2] * 4, 1), precision='fast' if fp16 else 'exact',
                                                lane=lane)
            assert torch.equal(ct2, ct) and torch.equal(cb2, cb) and torch.equal(px2, px)
    # interleaved: six batches with different seeds over three lanes == the same six batches one at a time on lane 0
    serial = [sampling_ihqgpt(model.stage2, B, 7, use_fp16=True, is_tqdm=False, max_seq_len=n, seed=100 + k) for k in range(6)]
    serial = [(a.clone(), b.clone()) for a, b in serial]
    pipe = InflightSampler(model, lanes=3)
    got = [pipe.submit(B, 7, seed=100 + k, max_seq_len=n, use_fp16=True, decode=False) for k in range(6)]
    pipe.drain()
    eq = [bool(ev.query() and px is None and torch.equal(a, ct) and torch.equal(b, cb)) for (a, b), (ct, cb, px, ev) in zip(serial, got)]
    assert all(eq), eq
    # ownership: the parent handle refuses to go away while a clone lives
    eng = model.stage2.engine(B, n)
    assert eng.lib.hqt_destroy(eng.h) != 0 and b'clone' in eng.lib.hqt_last_error()


def test_txt2img_counterpart_writes_reference_format(tmp_path):
    """sampling_hqmodel_txt2img counterpart on the tiny text-conditional config: synthetic prompt ids, and real captions
    through the tokenizer front-end (tiny own vocabulary); one pickle of float32 [B, 3, H, W] in [0, 1] per batch."""
    import json
    from hqtransformer_amd import sampling_hqmodel_txt2img as st
    tiny_txt = os.path.join(ROOT, 'configs', 'tiny-txt.yaml')
    out1 = tmp_path / 'syn'
    st.main(['-r', str(out1), '-m', tiny_txt, '--batch_size', '2', '--synthetic-prompts', '3', '--top-k', '50'])
    a = pickle.load(open(out1 / 'samples_(1_2).pkl', 'rb'))
    b = pickle.load(open(out1 / 'samples_(2_2).pkl', 'rb'))
    assert a.dtype == np.float32 and a.shape == (2, 3, 64, 64) and b.shape == (1, 3, 64, 64) and a.min() >= 0 and a.max() <= 1
    vocab = {'[UNK]': 0, 'a</w>': 1, 'c': 2, 'a': 3, 't</w>': 4, 'ca': 5, 'cat</w>': 6, 'd': 7, 'o': 8, 'g</w>': 9, 'do': 10, 'dog</w>': 11}
    (tmp_path / 'v.json').write_text(json.dumps(vocab))
    (tmp_path / 'm.txt').write_text('#version: 0.2\nc a\nca t</w>\nd o\ndo g</w>\n')
    (tmp_path / 'caps.txt').write_text('x.jpg\ta cat\ny.jpg\ta dog\n')
    out2 = tmp_path / 'cap'
    st.main(['-r', str(out2), '-m', tiny_txt, '--batch_size', '2', '--captions', str(tmp_path / 'caps.txt'),
             '--tokenizer-vocab', str(tmp_path / 'v.json'), '--tokenizer-merges', str(tmp_path / 'm.txt'), '--top-k', '50'])
    c = pickle.load(open(out2 / 'samples_(1_2).pkl', 'rb'))
    assert c.shape == (2, 3, 64, 64) and np.isfinite(c).all() and np.abs(c[0] - c[1]).max() > 0      # different prompts, different images


def test_three_level_surface_matches_the_oracle(tmp_path, capsys):
    """ImageGPT2 from a multilevel-hq / hqvae config: sampling_hqtransformer + stage1.decode_code([t, m, b]) against the
    CPU oracle, and the harness / driver counterparts in their code_levels = 3 modes."""
    from hqtransformer_amd import measure_throughput as mt, sampling_hqmodel as sh
    from hqtransformer_amd.sampling import rearrange_codes3, sampling_hqtransformer
    l3 = os.path.join(ROOT, 'configs', 'tiny-l3.yaml')
    m = ImageGPT2(load_config(l3), seed=9).to('cuda').eval()
    s2, s1 = m.stage2.spec, m.stage1.spec
    assert s2.levels == 3 and s1.code_levels == 3 and m.stage2.code_level == 3
    B, n = 3, 16
    noise = np.maximum(np.random.default_rng(5).standard_exponential((n, 21, B, s2.vocab_top), dtype=np.float32), np.float32(1e-30))
    codes = sampling_hqtransformer(m.stage2, num_candidates=B, cond=123, top_k=[100, 50, None], top_p=[0.9, None, None],
                                   softmax_temperature=[1.0, 0.9, 0.8], use_fp16=False, is_tqdm=False, max_seq_len=n,
                                   noise=torch.from_numpy(noise))
    assert [tuple(c.shape) for c in codes] == [(B, n), (B, n, 4), (B, n, 16)] and all(c.dtype == torch.int64 and c.is_cuda for c in codes)
    w2 = {k: v.numpy() for k, v in m.stage2.state_dict().items()}
    w1 = {k: v.numpy() for k, v in m.stage1.state_dict().items()}
    want = O.OracleStage2L3(s2, w2).sample(np.full(B, 123), B, n, noise, (100, 50, None), (0.9, None, None), (1.0, 0.9, 0.8))
    assert all((c.cpu().numpy() == w).all() for c, w in zip(codes, want))
    grids = rearrange_codes3(codes, 4)
    px = m.stage1.decode_code(list(grids))
    ref = O.OracleStage1(s1, w1).decode_codes3(list(O.rearrange_codes3(*want, 4)))
    assert tuple(px.shape) == (B, 3, 64, 64) and np.abs(px.cpu().numpy() - ref).max() <= 1e-4
    assert np.abs(m.stage1.decode_sequences(codes, clamp01=True).cpu().numpy() - O.postprocess(ref)).max() <= 1e-4
    out = mt.main(parse_dotlist([f'model_path={l3}', 'batch_size=500', 'n_loop=2', 'warmup=1', 'code_levels=3', 'top_resolution=4'],
                                mt.EXPERIMENT_DEFAULTS))
    assert 'ms/sample (ar:' in capsys.readouterr().out and out['ms_per_sample'] > 0
    sh.main(['-r', str(tmp_path), '-m', l3, '--batch-size', '2', '--num-classes', '1', '--samples-per-class', '2', '--code-level', '3',
             '--top-k', '100', '--top-resolution', '4'])
    pxs = pickle.load(open(tmp_path / 'samples_(1_0).pkl', 'rb'))
    assert pxs.dtype == np.float32 and pxs.shape == (2, 3, 64, 64) and pxs.min() >= 0 and pxs.max() <= 1


def test_encode_surface_matches_the_oracle(model):
    """``stage1.encode`` / ``get_codes`` / ``forward`` with the reference's return shapes (generator.py:298-310, 369-370, 262-280),
    against the CPU oracle built from the same state dict.  tiny-cls uses 'bench'-profile weights, so a code may sit on an fp32
    argmin tie: mismatches are accepted only where the oracle's own float64 distance gap is below 1e-4."""
    s1 = model.stage1.spec
    w1 = {k: v.numpy() for k, v in model.stage1.state_dict().items()}
    r = np.random.default_rng(3)
    x = np.clip(0.5 * r.standard_normal((2, 3, s1.resolution, s1.resolution)), -1, 1).astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    quant_t, quant_b, diff_t, diff_b, (code_t, code_b, h_b) = model.stage1.encode(xt)
    rz = s1.z_res
    assert tuple(quant_t.shape) == (2, 4 * s1.embed_dim, rz // 2, rz // 2) and tuple(quant_b.shape) == (2, s1.embed_dim, rz, rz)
    assert tuple(code_t.shape) == (2 * (rz // 2) ** 2,) and tuple(code_b.shape) == (2 * rz * rz,) and code_t.dtype == torch.int64
    assert tuple(h_b.shape) == (2, s1.embed_dim, rz, rz) and diff_t.ndim == 0
    want = O.OracleStage1(s1, w1).encode(x)
    assert np.abs(h_b.cpu().numpy() - want['resid'][1]).max() <= 1e-4 or not np.array_equal(code_t.cpu().numpy(), want['codes'][0].reshape(-1))
    for got, l in ((code_t, 0), (code_b, 1)):
        g = got.cpu().numpy()
        bad = g != want['codes'][l].reshape(-1)
        if l == 0 and bad.any():
            z = want['resid'][0].transpose(0, 2, 3, 1).reshape(-1, want['resid'][0].shape[1]).astype(np.float64)
            e = w1['quantize_t.embedding'].astype(np.float64)
            d = (z ** 2).sum(1, keepdims=True) + (e ** 2).sum(1)[None] - 2 * z @ e.T
            assert (d[np.arange(len(z)), g] - d.min(1))[bad].max() <= 1e-4
        elif l == 1 and np.array_equal(code_t.cpu().numpy(), want['codes'][0].reshape(-1)):
            assert bad.mean() <= 0.01
    assert abs(float(diff_t) - float(want['diff'][0])) <= 1e-3 * float(want['diff'][0])
    ct, cb = model.stage1.get_codes(xt)
    assert torch.equal(ct, code_t) and torch.equal(cb, code_b)
    rec = model.stage1(xt)                                     # decode(encode(x))
    grids = model.stage1.code_grids(xt)
    assert torch.equal(rec, model.stage1.decode_code(grids[0], grids[1]))
    assert tuple(rec.shape) == (2, 3, s1.resolution, s1.resolution)
    fast = model.stage1.get_codes(xt, precision='fast')
    assert (fast[0] == ct).float().mean() >= 0.5


def test_three_level_encode_surface(tmp_path):
    """``HQVAEGenerator.encode`` through the model mirror (generator.py:530-568): ``(quant, diffs, codes, resids[1:])`` with the
    reference's shapes; the summed reconstruction equals what ``decode_code`` rebuilds from the codes (checked through the
    pixels), and the round trip ``stage1(x)`` equals ``decode_code(code_grids(x))``."""
    l3 = os.path.join(ROOT, 'configs', 'tiny-l3.yaml')
    m = ImageGPT2(load_config(l3), seed=9).to('cuda').eval()
    s1 = m.stage1.spec
    w1 = {k: v.numpy() for k, v in m.stage1.state_dict().items()}
    r = np.random.default_rng(4)
    x = np.clip(0.5 * r.standard_normal((2, 3, s1.resolution, s1.resolution)), -1, 1).astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    quant, diffs, codes, resids = m.stage1.encode(xt)
    rz, E = s1.z_res, s1.embed_dim
    assert tuple(quant.shape) == (2, E, rz, rz) and len(diffs) == 3 and len(codes) == 3 and len(resids) == 2
    assert [tuple(c.shape) for c in codes] == [(2 * (rz // 4) ** 2,), (2 * (rz // 2) ** 2,), (2 * rz * rz,)]
    assert [tuple(z.shape) for z in resids] == [(2, 4 * E, rz // 2, rz // 2), (2, E, rz, rz)]
    want = O.OracleStage1(s1, w1).encode(x)
    assert np.abs(resids[0].cpu().numpy() - want['resid'][1]).max() <= 1e-4 or not np.array_equal(codes[0].cpu().numpy(), want['codes'][0].reshape(-1))
    if all(np.array_equal(c.cpu().numpy(), w.reshape(-1)) for c, w in zip(codes, want['codes'])):
        assert np.abs(quant.cpu().numpy() - want['recon']).max() <= 1e-4
        assert all(abs(float(d) - float(wd)) <= 1e-3 * float(wd) for d, wd in zip(diffs, want['diff']))
    grids = m.stage1.code_grids(xt)
    assert torch.equal(m.stage1(xt), m.stage1.decode_code(list(grids)))
    assert [c.reshape(-1).tolist() for c in grids] == [c.tolist() for c in m.stage1.get_codes(xt)]


def test_merged_steps_draw_what_the_separate_calls_draw(model):
    """`InflightSampler(merge=k)`: k independent steps (own class id, own seed, own global offset) executed as one pass of k x B
    rows.  In EXACT arithmetic every step's codes and pixels must be bit-identical to the separate call; in FAST arithmetic the
    GEMM tiles depend on the row count, so the gate is agreement of the drawn codes."""
    from hqtransformer_amd.pipeline import InflightSampler
    from hqtransformer_amd.sampling import sampling_ihqgpt
    B, n = 3, 64
    steps = [(5, 11, 0), (2, 12, 64), (9, 13, 7)]          # (class id, seed, sample_offset)
    for fast in (False, True):
        sep = []
        for cls, seed, off in steps:
            ct, cb = sampling_ihqgpt(model.stage2, num_candidates=B, cond=cls, top_k_top=100, top_p_top=0.95, top_k_bot=None, top_p_bot=None,
                                     softmax_temperature=[1.0, 0.9], use_fp16=fast, is_tqdm=False, max_seq_len=n, seed=seed, sample_offset=off)
            px = model.stage1.decode_sequences(ct, cb, precision='exact', clamp01=True)
            sep.append((ct.clone(), cb.clone(), px.clone()))
        pipe = InflightSampler(model, lanes=1, merge=3)
        pend = [pipe.submit(B, cls, seed=seed, max_seq_len=n, use_fp16=fast, precision='exact', sample_offset=off, top_k_top=100, top_p_top=0.95,
                            top_k_bot=None, top_p_bot=None, softmax_temperature=[1.0, 0.9]) for cls, seed, off in steps]
        pipe.drain()
        torch.cuda.synchronize()
        for p, (ct, cb, px) in zip(pend, sep):
            mct, mcb, mpx, _ = p.get()
            if fast:
                agree = ((mct == ct).float().mean().item() + (mcb == cb).float().mean().item()) / 2
                assert agree >= 0.95, agree
            else:
                assert torch.equal(mct, ct) and torch.equal(mcb, cb), 'merged EXACT codes differ from the separate call'
                assert torch.equal(mpx, px)
    # a partial group is launched by drain()
    pipe = InflightSampler(model, lanes=1, merge=4)
    p = pipe.submit(B, 1, seed=3, max_seq_len=n, use_fp16=True)
    with pytest.raises(RuntimeError):
        p.get()
    pipe.drain()
    assert p.get()[0].shape == (B, n)


@pytest.mark.gpu
@pytest.mark.parametrize('cfg_name', ['tiny-txt.yaml', 'tiny-l3.yaml'])
def test_merged_steps_text_and_three_levels(cfg_name):
    """Merged steps for the text-conditional model (one prompt row per sample, 16-token prefill of k x B rows) and the three-level
    HQTransformer: EXACT codes and pixels of every step bit-identical to the separate calls."""
    from hqtransformer_amd.config import load_config
    from hqtransformer_amd.models import ImageGPT2
    from hqtransformer_amd.pipeline import InflightSampler
    from hqtransformer_amd.sampling import sampling_hqtransformer, sampling_ihqgpt
    m = ImageGPT2(load_config(os.path.join(ROOT, 'configs', cfg_name)), seed=3).to('cuda').eval()
    s2 = m.stage2.spec
    three = getattr(s2, 'levels', 2) == 3
    txt = m.stage2.use_txt_cond
    B, n = 2, (16 if three else 64)                          # tiny-l3 has a 4 x 4 top grid
    r = np.random.default_rng(5)
    steps = []
    for k in range(3):
        cond = torch.from_numpy(r.integers(0, s2.vocab_txt, (B, s2.ctx_len_txt))) if txt else int(r.integers(0, max(1, s2.n_classes)))
        steps.append((cond, 20 + k, 5 * k))
    kw = dict(top_k=[50, 50, 50], top_p=[None, 0.9, None], softmax_temperature=[1.0, 0.9, 0.8]) if three else \
        dict(top_k_top=50, top_p_top=0.9, top_k_bot=None, top_p_bot=None, softmax_temperature=[1.0, 0.9])
    sep = []
    for cond, seed, off in steps:
        if three:
            c = sampling_hqtransformer(m.stage2, num_candidates=B, cond=cond, use_fp16=False, is_tqdm=False, max_seq_len=n, seed=seed, sample_offset=off, **kw)
            px = m.stage1.decode_sequences(list(c), None, precision='exact', clamp01=True)
            sep.append(([x.clone() for x in c], px.clone()))
        else:
            ct, cb = sampling_ihqgpt(m.stage2, num_candidates=B, cond=cond, use_fp16=False, is_tqdm=False, max_seq_len=n, seed=seed, sample_offset=off, **kw)
            px = m.stage1.decode_sequences(ct, cb, precision='exact', clamp01=True)
            sep.append(([ct.clone(), cb.clone()], px.clone()))
    pipe = InflightSampler(m, lanes=1, merge=3)
    pend = [pipe.submit(B, cond, seed=seed, max_seq_len=n, use_fp16=False, precision='exact', sample_offset=off, **kw) for cond, seed, off in steps]
    pipe.drain()
    torch.cuda.synchronize()
    for p, (codes, px) in zip(pend, sep):
        mct, mcb, mpx, _ = p.get()
        got = [mct] + (list(mcb) if isinstance(mcb, (list, tuple)) else [mcb])
        assert len(got) == len(codes) and all(torch.equal(a, b) for a, b in zip(got, codes)), 'merged EXACT codes differ from the separate call'
        assert torch.equal(mpx, px)


def test_merged_steps_reject_a_mismatched_step_without_losing_the_queue(model):
    """Steps of one merged pass must share their settings.  The check runs when a step is SUBMITTED: the offending step raises and is
    not queued, the steps already queued keep their Pending objects and run (a ValueError inside flush() used to orphan them), and
    settings that hold tensors (given codes) are compared by value."""
    from hqtransformer_amd.pipeline import InflightSampler
    pipe = InflightSampler(model, lanes=2, merge=3)
    a = pipe.submit(2, 3, seed=1, max_seq_len=64, use_fp16=False, precision='exact')
    with pytest.raises(ValueError):
        pipe.submit(2, 4, seed=2, max_seq_len=32, use_fp16=False, precision='exact')      # another max_seq_len
    with pytest.raises(ValueError):
        pipe.submit(2, 4, seed=2, max_seq_len=64, use_fp16=False, precision='exact', top_k_top=10)
    given = torch.arange(64).reshape(1, 64) % model.stage2.spec.vocab_top
    with pytest.raises(ValueError):
        pipe.submit(2, 4, seed=2, max_seq_len=64, use_fp16=False, precision='exact', given_top_code=given)
    b = pipe.submit(2, 5, seed=3, max_seq_len=64, use_fp16=False, precision='exact')
    pipe.drain()
    torch.cuda.synchronize()
    ct_a, ct_b = a.get()[0], b.get()[0]
    assert tuple(ct_a.shape) == (2, 64) and tuple(ct_b.shape) == (2, 64)
    want = sampling_ihqgpt(model.stage2, 2, 3, use_fp16=False, is_tqdm=False, max_seq_len=64, seed=1)
    assert (ct_a == want[0]).all()
    # tensors inside the settings compare by value
    pipe2 = InflightSampler(model, lanes=1, merge=2)
    x = pipe2.submit(2, 3, seed=1, max_seq_len=64, use_fp16=False, precision='exact', given_top_code=given.clone())
    y = pipe2.submit(2, 3, seed=2, max_seq_len=64, use_fp16=False, precision='exact', given_top_code=given.clone())
    pipe2.drain()
    torch.cuda.synchronize()
    alone = sampling_ihqgpt(model.stage2, 2, 3, use_fp16=False, is_tqdm=False, max_seq_len=64, seed=1, given_top_code=given.clone())
    assert (x.get()[0] == alone[0]).all() and (x.get()[1] == alone[1]).all() and tuple(y.get()[0].shape) == (2, 64)


def test_measure_throughput_txt_counterpart(capsys, tmp_path):
    """measure_throughput_txt counterpart (measure_throughput_txt/__main__.py:83-188) on the tiny text-conditional config: synthetic
    prompts, the harness's quality-mode sampler (top-k clipped to the vocabulary, top-p 1.0), H1's loop accounting and printed lines;
    then real captions through the BPE front-end, several iterations in flight, and the refusal of a class-conditional config."""
    import json
    from hqtransformer_amd import measure_throughput_txt as mtt
    tiny_txt = os.path.join(ROOT, 'configs', 'tiny-txt.yaml')
    out = mtt.main(parse_dotlist([f'model_path={tiny_txt}', 'batch_size=500', 'n_loop=2', 'warmup=1', 'top_k=100'], mtt.EXPERIMENT_DEFAULTS))
    text = capsys.readouterr().out
    assert 'bs500, sampling loops 2-2' in text and 'transformer size:' in text and 'ms/sample (ar:' in text
    assert out['ms_per_sample'] > 0 and abs(out['ms_ar'] + out['ms_decode'] - out['ms_per_sample']) / out['ms_per_sample'] < 0.2
    vocab = {'[UNK]': 0, 'a</w>': 1, 'c': 2, 'a': 3, 't</w>': 4, 'ca': 5, 'cat</w>': 6, 'd': 7, 'o': 8, 'g</w>': 9, 'do': 10, 'dog</w>': 11}
    (tmp_path / 'v.json').write_text(json.dumps(vocab))
    (tmp_path / 'm.txt').write_text('#version: 0.2\nc a\nca t</w>\nd o\ndo g</w>\n')
    (tmp_path / 'caps.txt').write_text('x.jpg\ta cat\ny.jpg\ta dog\n')
    out2 = mtt.main(parse_dotlist([f'model_path={tiny_txt}', 'batch_size=250', 'n_loop=2', 'warmup=1', 'top_k=100', 'inflight=2',
                                   f'captions={tmp_path / "caps.txt"}', f'tokenizer_vocab={tmp_path / "v.json"}', f'tokenizer_merges={tmp_path / "m.txt"}'],
                                  mtt.EXPERIMENT_DEFAULTS))
    assert 'bs250, sampling loops 2-2' in capsys.readouterr().out and out2['ms_per_sample'] > 0
    with pytest.raises(ValueError):
        mtt.main(parse_dotlist([f'model_path={TINY}', 'batch_size=50', 'n_loop=1', 'warmup=0'], mtt.EXPERIMENT_DEFAULTS))


def test_reference_token_ids_through_the_text_conditional_sampler():
    """Fixture G10 holds the ids the REFERENCE's tokenizer + dataset padding / truncation produced for 20 captions (64-token form,
    16k vocabulary, incl. an empty and an over-long caption).  Those ids -- not a vocabulary file -- go through sampling_ihqgpt on a tiny
    model with the released text front-end's geometry (vocab_size_txt 16384, ctx_len_txt 64): the 64-row causal prefill and 8 decode
    steps in EXACT arithmetic draw bit-identical codes to the CPU oracle on the same ids, and FAST stays within its logits gate."""
    from tests.helpers import gate, load
    fx = load('g10_tokenizer.npz')
    ids = fx['ids_64'].astype(np.int64)
    assert ids.shape == (20, 64) and ids.max() < 16384
    m = ImageGPT2(load_config(os.path.join(ROOT, 'configs', 'tiny-txt-bpe16k.yaml')), seed=7).to('cuda').eval()
    s2 = m.stage2.spec
    assert (s2.vocab_txt, s2.ctx_len_txt) == (16384, 64)
    w2 = {k: v.numpy() for k, v in m.stage2.state_dict().items()}
    B, n = ids.shape[0], 8
    noise = synth.exp_noise(5, n, B, s2.vocab_top)
    ct, cb = sampling_ihqgpt(m.stage2, num_candidates=1, cond=torch.from_numpy(ids), use_fp16=False, is_tqdm=False, max_seq_len=n,
                             noise=torch.from_numpy(noise))
    want = O.OracleStage2(s2, w2).sample(ids, B, n, noise, return_logits=True)
    assert (ct.cpu().numpy() == want[0]).all() and (cb.cpu().numpy() == want[1]).all()
    assert (want[0][0] != want[0][1]).any()                      # two different captions do not draw the same codes
    eng = m.stage2.engine(B, n)
    _, _, lf = eng.sample(B, torch.from_numpy(ids), n, precision=1, noise=torch.from_numpy(noise), force_top=torch.from_numpy(want[0]),
                          force_bot=torch.from_numpy(want[1]), return_logits=True, use_graph=True)
    gate('g10_ids.fast_logits_vs_oracle', np.abs(lf.cpu().numpy() - want[2]).max(), 0.15)
