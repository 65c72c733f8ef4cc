"""CPU oracle for the HQ-Transformer sampling path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A numpy (fp32) restatement of what the reference computes on its CPU path, where autocast is off and
every activation is fp32.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module; nothing under ``hqtransformer_amd/`` does,
and the product path raises when its HIP library is missing instead of falling back to this.

Parity pin: the reference ships no tests or golden vectors, so this oracle is pinned
against outputs of the reference itself, generated in the build container by ``tools/gen_golden.py``
(which imports ``/root/reference``) and committed under ``tests/golden/`` --
``tests/test_oracle_golden.py`` checks every fixture.

Each function cites the reference lines it restates (paths relative to the reference root).
Third-party arithmetic the reference relies on is PyTorch's (torch==1.10.0, requirements.txt:11):
Linear, LayerNorm, GELU(erf), softmax, topk, sort, cumsum, multinomial, Conv2d, GroupNorm, nearest
interpolate, PixelShuffle, embedding -- restated here from their published definitions.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
from scipy.special import erf

F32 = np.float32


# ----------------------------------------------------------------------------- primitives
def linear(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray] = None) -> np.ndarray:
    """nn.Linear: y = x W^T + b, W stored [out, in]."""
    y = x.astype(F32, copy=False) @ w.T
    if b is not None:
        y = y + b
    return y.astype(F32, copy=False)


def layer_norm(x: np.ndarray, g: np.ndarray, b: np.ndarray, eps: float = 1e-5) -> np.ndarray:
    """nn.LayerNorm over the last axis, eps 1e-5 (stage2/layers.py:302-303, hierarchical_ar.py:144,205,208)."""
    x64 = x.astype(np.float64)
    mu = x64.mean(-1, keepdims=True)
    var = ((x64 - mu) ** 2).mean(-1, keepdims=True)
    return (((x64 - mu) / np.sqrt(var + eps)).astype(F32) * g + b).astype(F32)


def gelu(x: np.ndarray, approx: bool) -> np.ndarray:
    """stage2/layers.py:14-23: exact erf GELU, or x*sigmoid(1.702x) when gelu_use_approx."""
    if approx:
        return (x / (1.0 + np.exp(-1.702 * x))).astype(F32)
    return (x * 0.5 * (1.0 + erf(x / np.sqrt(F32(2.0))))).astype(F32)


def softmax(x: np.ndarray) -> np.ndarray:
    m = x.max(-1, keepdims=True)
    e = np.exp(x - m)
    return (e / e.sum(-1, keepdims=True)).astype(F32)


# ----------------------------------------------------------------------------- sampler (A7)
# Tests that demand bit-identical free-running code sequences set this to a list: every draw appends its smallest winner / runner-up
# ratio of p / q, so the test can state that the sequence it compares is well-conditioned (a margin of 1 + 2e-6 is decided by the
# summation order of ANY fp32 implementation, the reference's included).
MARGIN_SINK = None


def cutoff_topk_logits(logits: np.ndarray, k: Optional[int]) -> np.ndarray:
    """hqvae/utils/sampling.py:12-19 -- keep logits >= k-th largest (ties kept), others -> -inf."""
    if k is None:
        return logits
    kth = np.partition(logits, logits.shape[-1] - int(k), axis=-1)[..., logits.shape[-1] - int(k)][..., None]
    out = logits.copy()
    out[out < kth] = -np.inf
    return out


def cutoff_topp_probs(probs: np.ndarray, p: Optional[float]) -> np.ndarray:
    """hqvae/utils/sampling.py:22-37 -- sort desc, running sum, drop where cum >= p shifted right by one
    (first crossing token kept, head always kept), scatter back, renormalise.  torch.cumsum on the CPU
    accumulates an fp32 row in a double and rounds every prefix to fp32 (acc_type<float> is double on
    CPU; checked against torch 2.10 here: bit-identical on 64x8192 rows), which is what decides the
    ``cum >= p`` cut at p = 1.0."""
    if p is None:
        return probs
    order = np.argsort(-probs, axis=-1, kind='stable')
    sp = np.take_along_axis(probs, order, axis=-1)
    cum = np.cumsum(sp.astype(np.float64), axis=-1).astype(F32)
    remove_sorted = cum >= F32(p)
    remove_sorted[..., 1:] = remove_sorted[..., :-1].copy()
    remove_sorted[..., 0] = False
    remove = np.zeros_like(remove_sorted)
    np.put_along_axis(remove, order, remove_sorted, axis=-1)
    out = np.where(remove, F32(0.0), probs)
    return (out / out.sum(-1, keepdims=True)).astype(F32)


def sample_filtered(logits: np.ndarray, q: np.ndarray, temperature: float, top_k: Optional[int],
                    top_p: Optional[float]) -> Tuple[np.ndarray, np.ndarray]:
    """hierarchical_ar.py:762-769 / 778-784: logits /= T; top-k; softmax; top-p; multinomial.
    ``torch.multinomial(p, 1)`` draws ``argmax(p / q)`` with q ~ Exp(1); q is an
    explicit input here.  Returns (index [R], probs [R, V])."""
    lg = (logits / F32(temperature)).astype(F32)
    lg = cutoff_topk_logits(lg, top_k)
    pr = softmax(lg)
    pr = cutoff_topp_probs(pr, top_p)
    ratio = pr / q
    idx = np.argmax(ratio, axis=-1).astype(np.int64)
    if MARGIN_SINK is not None:                      # conditioning of the draw: winner / runner-up of p / q per row (tests only)
        part = np.partition(ratio, -2, axis=-1)
        MARGIN_SINK.append(float((part[..., -1] / np.maximum(part[..., -2], F32(1e-38))).min()))
    return idx, pr


# ----------------------------------------------------------------------------- stage 2 (A1-A6, A14)
class OracleStage2:
    """iHQGPT 'parallel' sampling (hierarchical_ar.py:428-563, 667-789; sampling.py:164-237)."""

    def __init__(self, spec, weights: Dict[str, np.ndarray]):
        self.s = spec
        self.w = {k: np.ascontiguousarray(v, dtype=F32) for k, v in weights.items()}

    # one transformer block on T new tokens, appending K/V to the cache (stage2/layers.py:61-195,324-328,371-375)
    def _block(self, prefix: str, x: np.ndarray, cache: Dict[str, np.ndarray], causal_new: bool) -> np.ndarray:
        w, s = self.w, self.s
        B, T, D = x.shape
        nh, hs = s.n_heads, s.head_dim
        h = layer_norm(x, w[f'{prefix}.ln1.weight'], w[f'{prefix}.ln1.bias'])
        q = linear(h, w[f'{prefix}.attn.query.weight'], w[f'{prefix}.attn.query.bias'])
        k = linear(h, w[f'{prefix}.attn.key.weight'], w[f'{prefix}.attn.key.bias'])
        v = linear(h, w[f'{prefix}.attn.value.weight'], w[f'{prefix}.attn.value.bias'])
        split = lambda t: t.reshape(B, T, nh, hs).transpose(0, 2, 1, 3)          # [B, nh, T, hs]
        q, k, v = split(q), split(k), split(v)
        if prefix in cache:
            pk, pv = cache[prefix]
            k = np.concatenate([pk, k], axis=2)
            v = np.concatenate([pv, v], axis=2)
        cache[prefix] = (k, v)
        t_past = k.shape[2] - T
        att = q @ (k.transpose(0, 1, 3, 2) * F32(1.0 / np.sqrt(hs)))                # scale on K (layers.py:102)
        if causal_new and T > 1:                                                    # layers.py:107-111,118-123
            mask = np.concatenate([np.ones((T, t_past), bool), np.tril(np.ones((T, T), bool))], axis=1)
            att = np.where(mask[None, None], att, F32(-np.inf))
        att = softmax(att.astype(F32))
        y = (att @ v).transpose(0, 2, 1, 3).reshape(B, T, D)
        x = x + linear(y, w[f'{prefix}.attn.proj.weight'], w[f'{prefix}.attn.proj.bias'])
        h = layer_norm(x, w[f'{prefix}.ln2.weight'], w[f'{prefix}.ln2.bias'])
        h = gelu(linear(h, w[f'{prefix}.mlp.0.weight'], w[f'{prefix}.mlp.0.bias']), s.gelu_approx)
        return (x + linear(h, w[f'{prefix}.mlp.2.weight'], w[f'{prefix}.mlp.2.bias'])).astype(F32)

    def _embed(self, code_t: np.ndarray, code_b: np.ndarray, pos: int) -> np.ndarray:
        """hierarchical_ar.py:505-544: input embedding of one top position from its 1+4 codes."""
        w, s = self.w, self.s
        top = w['tok_emb_top.weight'][code_t] + w['pos_emb_top.weight'][pos]          # [B, D]
        if s.embedding == 1:                                                         # 'reduce' :522-526
            eb = w['tok_emb_bot.weight'][code_b]                                     # [B, 4, D/4]
            return (top + eb.transpose(0, 2, 1).reshape(top.shape[0], -1))[:, None, :].astype(F32)   # ch = k*4+slot
        eb = w['tok_emb_bot.weight'][code_b]                                         # [B, 4, D]   :535-544
        h = np.concatenate([top[:, None, :], eb], axis=1) + w['pos_emb_emb.weight'][None]
        return h.mean(axis=1, dtype=F32)[:, None, :].astype(F32)

    def sample(self, cond, batch: int, n_steps: int, noise: np.ndarray,
               top_k: Sequence[Optional[int]] = (None, None), top_p: Sequence[Optional[float]] = (None, None),
               temperature: Sequence[float] = (1.0, 1.0), force_top: Optional[np.ndarray] = None,
               force_bot: Optional[np.ndarray] = None, return_logits: bool = False):
        """sampling_ihqgpt (sampling.py:164-237).  ``cond``: int64 [B] class ids, or [B, ctx_len_txt]
        text ids, or None.  ``noise`` [n_steps, 5, B, V].  ``force_top`` [B, n_steps] / ``force_bot``
        [B, n_steps, 4] teacher-force the codes fed back (force_top = the reference's given_top_code,
        hierarchical_ar.py:768-774); the draws are still computed and returned."""
        w, s = self.w, self.s
        B = batch
        if s.cond == 1:
            sos = w['sos.weight'][np.asarray(cond, np.int64).reshape(-1)][:, None, :]      # sampling.py:183-186
        elif s.cond == 2:
            sos = w['tok_emb_txt.weight'][np.asarray(cond, np.int64)] + w['pos_emb_txt.weight'][None, :s.ctx_len_txt]
        else:
            sos = np.repeat(w['sos'], B, axis=0)                                         # sampling.py:191-192
        cache: Dict[str, Tuple[np.ndarray, np.ndarray]] = {}
        codes_top = np.zeros((B, n_steps), np.int64)
        codes_bot = np.zeros((B, n_steps, 4), np.int64)
        logits_out = np.zeros((n_steps, 5, B, s.vocab_top), F32) if return_logits else None
        for cnt in range(n_steps):
            if cnt == 0:
                xs = sos.astype(F32)
            else:
                ct = force_top[:, cnt - 1] if force_top is not None else codes_top[:, cnt - 1]
                cb = force_bot[:, cnt - 1] if force_bot is not None else codes_bot[:, cnt - 1]
                xs = self._embed(ct, cb, cnt - 1)
            for i in range(s.n_layers):                                                  # :501-503,557-559
                xs = self._block(f'blocks.{i}', xs, cache, causal_new=True)
            hs = layer_norm(xs, w['ln_f.weight'], w['ln_f.bias'])
            if hs.shape[1] > 1:                                                          # :684-685
                hs = hs[:, s.idx_pred - 1:s.idx_pred, :]
            # depth sub-step 0: top code (hierarchical_ar.py:682-695, 762-775)
            dcache: Dict[str, Tuple[np.ndarray, np.ndarray]] = {}
            xd = (hs + w['sos_depth']).astype(F32)
            for j in range(s.n_layers_depth):
                xd = self._block(f'depths.{j}', xd, dcache, causal_new=False)
            lt = linear(layer_norm(xd, w['ln_top.weight'], w['ln_top.bias']), w['head_top.weight'])[:, 0]
            top, _ = sample_filtered(lt, noise[cnt, 0], temperature[0], top_k[0], top_p[0])
            codes_top[:, cnt] = top
            if return_logits:
                logits_out[cnt, 0] = lt
            fed_top = force_top[:, cnt] if force_top is not None else top
            # depth sub-step 1: four bottom codes in one pass (hierarchical_ar.py:696-719, 776-785)
            xd = (w['tok_emb_top_depth.weight'][fed_top][:, None, :] + w['pos_emb_depth.weight'][None, :4]).astype(F32)
            for j in range(s.n_layers_depth):
                xd = self._block(f'depths.{j}', xd, dcache, causal_new=False)            # all-ones mask, layers.py:147-152
            lb = linear(layer_norm(xd, w['ln_bot.weight'], w['ln_bot.bias']), w['head_bot.weight'])  # [B,4,V]
            for slot in range(4):
                idx, _ = sample_filtered(lb[:, slot], noise[cnt, 1 + slot], temperature[1], top_k[1], top_p[1])
                codes_bot[:, cnt, slot] = idx
                if return_logits:
                    logits_out[cnt, 1 + slot] = lb[:, slot]
        if return_logits:
            return codes_top, codes_bot, logits_out
        return codes_top, codes_bot


class OracleStage2L3(OracleStage2):
    """HQTransformer 'parallel-add' / 'parallel' / 'parallel-reduce' sampling, three code levels: hqtransformer.py:409-635 and
    sampling.py:240-307.  Per top position 1 + 4 + 16 codes; the depth transformer runs three sub-steps over 1, 4 and 16
    tokens, each seeing every earlier and current token (the 'parallel' mask of layers.py:154-178 is all-ones on the
    rows a sampling sub-step evaluates); 'transformer1' has no embedding blocks (hqtransformer.py:36-54), so the body
    input is the mean of the 21 code embeddings."""

    # raster order of the 16 level-2 tokens: index (H1 H2 W1 W2) -> parent cell (H1 W1), child (H2 W2)
    _L2_PARENT = np.array([(i // 8) * 2 + (i % 4) // 2 for i in range(16)])
    _L2_CHILD = np.array([((i // 4) % 2) * 2 + i % 2 for i in range(16)])

    def _embed3(self, c0: np.ndarray, c1: np.ndarray, c2: np.ndarray, pos: int) -> np.ndarray:
        """hqtransformer.py:466-488: mean over the 21 embedded codes (+ position of the embedding slot; + top position)."""
        w = self.w
        e0 = w['tok_emb_levels.0.weight'][c0] + w['pos_emb_top.weight'][pos]         # [B, D]
        e1 = w['tok_emb_levels.1.weight'][c1]                                        # [B, 4, D]
        e2 = w['tok_emb_levels.2.weight'][c2]                                        # [B, 16, D]
        h = np.concatenate([e0[:, None, :], e1, e2], axis=1) + w['pos_emb_emb.weight'][None]
        return h.mean(axis=1, dtype=F32)[:, None, :].astype(F32)

    def sample(self, cond, batch: int, n_steps: int, noise: np.ndarray,
               top_k: Sequence[Optional[int]] = (None, None, None), top_p: Sequence[Optional[float]] = (None, None, None),
               temperature: Sequence[float] = (1.0, 1.0, 1.0), force: Optional[Sequence[np.ndarray]] = None,
               return_logits: bool = False):
        """sampling_hqtransformer (sampling.py:240-307).  ``noise`` [n_steps, 21, B, V], draw order level 0, the four
        level-1 slots, the sixteen level-2 tokens in (H1 H2 W1 W2) raster order (hqtransformer.py:553-560,616-627).
        ``force`` = (codes0 [B, n], codes1 [B, n, 4], codes2 [B, n, 16]) teacher-forces what is fed back.
        Returns (codes0, codes1, codes2[, logits [n_steps, 21, B, V]])."""
        w, s = self.w, self.s
        B = batch
        if s.cond == 1:
            sos = w['sos.weight'][np.asarray(cond, np.int64).reshape(-1)][:, None, :]
        elif s.cond == 2:
            sos = w['tok_emb_txt.weight'][np.asarray(cond, np.int64)] + w['pos_emb_txt.weight'][None, :s.ctx_len_txt]
        else:
            sos = np.repeat(w['sos'], B, axis=0)
        cache: Dict[str, Tuple[np.ndarray, np.ndarray]] = {}
        c0 = np.zeros((B, n_steps), np.int64)
        c1 = np.zeros((B, n_steps, 4), np.int64)
        c2 = np.zeros((B, n_steps, 16), np.int64)
        logits_out = np.zeros((n_steps, 21, B, s.vocab_top), F32) if return_logits else None
        for cnt in range(n_steps):
            if cnt == 0:
                xs = sos.astype(F32)
            else:
                f = force if force is not None else (c0, c1, c2)
                xs = self._embed3(f[0][:, cnt - 1], f[1][:, cnt - 1], f[2][:, cnt - 1], cnt - 1)
            for i in range(s.n_layers):
                xs = self._block(f'blocks.{i}', xs, cache, causal_new=True)
            hs = layer_norm(xs, w['ln_f.weight'], w['ln_f.bias'])
            if hs.shape[1] > 1:                                                          # hqtransformer.py:522-523
                hs = hs[:, s.idx_pred - 1:s.idx_pred, :]
            dcache: Dict[str, Tuple[np.ndarray, np.ndarray]] = {}

            def depth(xd, level):
                for j in range(s.n_layers_depth):
                    xd = self._block(f'depths.{j}', xd, dcache, causal_new=False)
                return linear(layer_norm(xd, w[f'ln_levels.{level}.weight'], w[f'ln_levels.{level}.bias']), w[f'head_levels.{level}.weight'])

            if getattr(s, 'depth_decoding', 'parallel-add') == 'top2mid2bot':
                # sampling_depth_causal (hqtransformer.py:700-800): 21 causal sub-steps of one token.  Sub-step cnt >= 1 is fed the code of
                # cnt - 1 through tok_emb_levels[0 if cnt == 1 else 1 if cnt < 5 else 2] (:718-723 -- the table follows the sub-step being
                # computed) + pos_emb_depths.0[cnt - 1]; head / sampler settings of level 0 (cnt 0), 1 (cnt 1..4), 2 (cnt 5..20)
                seq = np.zeros((B, 21), np.int64)
                fseq = (np.concatenate([force[0][:, cnt, None], force[1][:, cnt], force[2][:, cnt]], axis=1) if force is not None else None)
                for sub in range(21):
                    lv = 0 if sub == 0 else (1 if sub < 5 else 2)
                    if sub == 0:
                        xd = (hs + w['sos_depth']).astype(F32)
                    else:
                        tbl = 0 if sub == 1 else (1 if sub < 5 else 2)
                        prev = (fseq if fseq is not None else seq)[:, sub - 1]
                        xd = (w[f'tok_emb_levels.{tbl}.weight'][prev] + w['pos_emb_depths.0.weight'][sub - 1])[:, None, :].astype(F32)
                    lg = depth(xd, lv)[:, 0]
                    seq[:, sub], _ = sample_filtered(lg, noise[cnt, sub], temperature[lv], top_k[lv], top_p[lv])
                    if return_logits:
                        logits_out[cnt, sub] = lg
                c0[:, cnt], c1[:, cnt], c2[:, cnt] = seq[:, 0], seq[:, 1:5], seq[:, 5:21]
                continue
            # level 0 (hqtransformer.py:518-524)
            l0 = depth((hs + w['sos_depth']).astype(F32), 0)[:, 0]
            d0, _ = sample_filtered(l0, noise[cnt, 0], temperature[0], top_k[0], top_p[0])
            c0[:, cnt] = d0
            fed0 = force[0][:, cnt] if force is not None else d0
            # level 1: four tokens = emb(top code) + positions 0..3 (:526-536 with cnt == 1)
            dd = getattr(s, 'depth_decoding', 'parallel-add')
            D = s.embed_dim
            e0 = w['tok_emb_depth_levels.0.weight'][fed0]                                # [B, D]; 'reduce': [B, 4 D], one D-slice per slot (:532-533)
            e0 = e0.reshape(B, 4, D) if 'reduce' in dd else e0[:, None, :]
            x1 = (e0 + w['pos_emb_depths.0.weight'][None, :4]).astype(F32)
            l1 = depth(x1, 1)                                                            # [B, 4, V]
            d1 = np.zeros((B, 4), np.int64)
            for k in range(4):
                d1[:, k], _ = sample_filtered(l1[:, k], noise[cnt, 1 + k], temperature[1], top_k[1], top_p[1])
            c1[:, cnt] = d1
            fed1 = force[1][:, cnt] if force is not None else d1
            # level 2: sixteen tokens in (H1 H2 W1 W2) raster order; token i carries its parent's level-1 embedding,
            # position i of pos_emb_depths[1], and the top code's embedding ('add', :537-551)
            e1 = w['tok_emb_depth_levels.1.weight'][fed1]                                # [B, 4, D], parents in (H1 W1) order
            if 'reduce' in dd:                                                           # [B, 4, 4 D]: child (H2 W2) of parent p takes slice 2 H2 + W2 (:538-539)
                e1 = e1.reshape(B, 4, 4, D)[:, self._L2_PARENT, self._L2_CHILD, :]
            else:
                e1 = e1[:, self._L2_PARENT, :]
            x2 = e1 + w['pos_emb_depths.1.weight'][None, :16]
            if 'add' in dd:                                                              # + the top code's embedding (:549-551)
                x2 = x2 + w['tok_emb_depth_levels.0.weight'][fed0][:, None, :]
            x2 = x2.astype(F32)
            l2 = depth(x2, 2)                                                            # [B, 16, V]
            for k in range(16):
                c2[:, cnt, k], _ = sample_filtered(l2[:, k], noise[cnt, 5 + k], temperature[2], top_k[2], top_p[2])
            if return_logits:
                logits_out[cnt, 0] = l0
                logits_out[cnt, 1:5] = l1.transpose(1, 0, 2)
                logits_out[cnt, 5:21] = l2.transpose(1, 0, 2)
        if return_logits:
            return c0, c1, c2, logits_out
        return c0, c1, c2


# ----------------------------------------------------------------------------- stage 1 (A9-A12)
def conv2d(x: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    """nn.Conv2d stride 1, 'same' zero padding, kernel 1 or 3, NCHW (stage1/modules/layers.py:40-44,88-98)."""
    B, C, H, W = x.shape
    O, _, kh, kw = w.shape
    if kh == 1:
        y = np.einsum('oc,bchw->bohw', w[:, :, 0, 0], x, optimize=True)
    else:
        xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
        cols = np.empty((B, C, 9, H, W), F32)
        for dy in range(3):
            for dx in range(3):
                cols[:, :, dy * 3 + dx] = xp[:, :, dy:dy + H, dx:dx + W]
        y = (w.reshape(O, C * 9) @ cols.reshape(B, C * 9, H * W)).reshape(B, O, H, W)
    return (y + b[None, :, None, None]).astype(F32)


def conv2d_strided(x: np.ndarray, w: np.ndarray, b: np.ndarray, stride: int, pad: Tuple[int, int, int, int]) -> np.ndarray:
    """nn.Conv2d with a stride and explicit zero padding (top, bottom, left, right): the encoder's conv_in (4x4, stride 2,
    padding 1 -- stage1/modules/layers.py:212-216) and Downsample (pad right / bottom by one, 3x3, stride 2, padding 0 --
    stage1/modules/layers.py:56-72)."""
    B, C, H, W = x.shape
    O, _, kh, kw = w.shape
    xp = np.pad(x, ((0, 0), (0, 0), (pad[0], pad[1]), (pad[2], pad[3])))
    Ho = (xp.shape[2] - kh) // stride + 1
    Wo = (xp.shape[3] - kw) // stride + 1
    cols = np.empty((B, C, kh * kw, Ho, Wo), F32)
    for dy in range(kh):
        for dx in range(kw):
            cols[:, :, dy * kw + dx] = xp[:, :, dy:dy + stride * Ho:stride, dx:dx + stride * Wo:stride]
    y = (w.reshape(O, C * kh * kw) @ cols.reshape(B, C * kh * kw, Ho * Wo)).reshape(B, O, Ho, Wo)
    return (y + b[None, :, None, None]).astype(F32)


def pixel_unshuffle2(x: np.ndarray) -> np.ndarray:
    """nn.PixelUnshuffle(2): out[4c + 2i + j, h, w] = in[c, 2h + i, 2w + j] (generator.py:228)."""
    B, C, H, W = x.shape
    return x.reshape(B, C, H // 2, 2, W // 2, 2).transpose(0, 1, 3, 5, 2, 4).reshape(B, C * 4, H // 2, W // 2)


def nearest_code(z: np.ndarray, emb: np.ndarray):
    """EMAVectorQuantizer.forward in eval mode (quantizer.py:91-133): z [B, C, H, W] -> (z + (e - z) [B, C, H, W], diff, codes [B*H*W]).
    d = |z|^2 + |e|^2 - 2 z.e in fp32 with the reference's association; argmin takes the first minimum."""
    B, C, H, W = z.shape
    zf = np.ascontiguousarray(z.transpose(0, 2, 3, 1)).reshape(-1, C).astype(F32)
    d = ((zf ** 2).sum(1, keepdims=True, dtype=F32) + (emb ** 2).sum(1, dtype=F32)[None, :]).astype(F32) - F32(2.0) * (zf @ emb.T).astype(F32)
    codes = np.argmin(d, axis=1)
    zq = emb[codes]
    diff = F32(0.25) * np.mean((zq - zf) ** 2, dtype=F32)
    st = (zf + (zq - zf)).astype(F32)                                       # quantizer.py:131
    return np.ascontiguousarray(st.reshape(B, H, W, C).transpose(0, 3, 1, 2)), F32(diff), codes.astype(np.int64)


def group_norm(x: np.ndarray, g: np.ndarray, b: np.ndarray, groups: int = 32, eps: float = 1e-6) -> np.ndarray:
    """GroupNorm(32, eps=1e-6, affine) -- stage1/modules/layers.py:17-21; stats per (sample, group)."""
    B, C, H, W = x.shape
    xg = x.reshape(B, groups, -1).astype(np.float64)
    mu = xg.mean(-1, keepdims=True)
    var = ((xg - mu) ** 2).mean(-1, keepdims=True)
    xn = ((xg - mu) / np.sqrt(var + eps)).astype(F32).reshape(B, C, H, W)
    return (xn * g[None, :, None, None] + b[None, :, None, None]).astype(F32)


def swish(x: np.ndarray) -> np.ndarray:
    """stage1/modules/layers.py:12-14."""
    return (x / (1.0 + np.exp(-x))).astype(F32)


def upsample_nearest2(x: np.ndarray) -> np.ndarray:
    """F.interpolate(scale_factor=2, mode='nearest') -- stage1/modules/layers.py:50."""
    return x.repeat(2, axis=2).repeat(2, axis=3)


def pixel_shuffle2(x: np.ndarray) -> np.ndarray:
    """nn.PixelShuffle(2): out[c, 2h+i, 2w+j] = in[4c+2i+j, h, w] (generator.py:229,316)."""
    B, C4, H, W = x.shape
    return x.reshape(B, C4 // 4, 2, 2, H, W).transpose(0, 1, 4, 2, 5, 3).reshape(B, C4 // 4, 2 * H, 2 * W)


class OracleStage1:
    """SimRQGAN2Generator.decode_code (generator.py:312-367) + Decoder.forward (stage1/modules/layers.py:385-410)."""

    def __init__(self, spec, weights: Dict[str, np.ndarray]):
        self.s = spec
        self.w = {k: np.ascontiguousarray(v, dtype=F32) for k, v in weights.items()}

    def _conv(self, name: str, x: np.ndarray) -> np.ndarray:
        return conv2d(x, self.w[f'{name}.weight'], self.w[f'{name}.bias'])

    def _gn(self, name: str, x: np.ndarray) -> np.ndarray:
        return group_norm(x, self.w[f'{name}.weight'], self.w[f'{name}.bias'])

    def _resblock(self, name: str, x: np.ndarray) -> np.ndarray:
        """ResnetBlock.forward (stage1/modules/layers.py:115-133)."""
        h = self._conv(f'{name}.conv1', swish(self._gn(f'{name}.norm1', x)))
        h = self._conv(f'{name}.conv2', swish(self._gn(f'{name}.norm2', h)))
        if f'{name}.nin_shortcut.weight' in self.w:
            x = self._conv(f'{name}.nin_shortcut', x)
        return (x + h).astype(F32)

    def _attnblock(self, name: str, x: np.ndarray) -> np.ndarray:
        """AttnBlock.forward (stage1/modules/layers.py:163-186): single head, scale C^-0.5 after q.k."""
        B, C, H, W = x.shape
        h = self._gn(f'{name}.norm', x)
        q = self._conv(f'{name}.q', h).reshape(B, C, H * W).transpose(0, 2, 1)
        k = self._conv(f'{name}.k', h).reshape(B, C, H * W)
        v = self._conv(f'{name}.v', h).reshape(B, C, H * W)
        wgt = softmax(((q @ k) * F32(int(C) ** (-0.5))).astype(F32))               # [B, i(query), j(key)]
        o = (v @ wgt.transpose(0, 2, 1)).reshape(B, C, H, W)                       # o[c, i] = sum_j v[c, j] w[i, j]
        return (x + self._conv(f'{name}.proj_out', o.astype(F32))).astype(F32)

    def decoder(self, z: np.ndarray) -> np.ndarray:
        s = self.s
        n = len(s.ch_mult)
        res = s.z_res
        h = self._conv('decoder.conv_in', z)
        if s.use_mid_block:
            h = self._resblock('decoder.mid.block_1', h)
            if s.use_attn:
                h = self._attnblock('decoder.mid.attn_1', h)
            h = self._resblock('decoder.mid.block_2', h)
        for lvl in reversed(range(n)):
            for blk in range(s.num_res_blocks + 1):
                h = self._resblock(f'decoder.up.{lvl}.block.{blk}', h)
                if res in s.attn_resolutions and s.use_attn:
                    h = self._attnblock(f'decoder.up.{lvl}.attn.{blk}', h)
            if lvl != 0 or s.use_init_downsample:
                h = self._conv(f'decoder.up.{lvl}.upsample.conv', upsample_nearest2(h))
                res *= 2
        h = swish(self._gn('decoder.norm_out', h))
        return self._conv('decoder.conv_out', h)

    def encoder(self, x: np.ndarray) -> np.ndarray:
        """Encoder.forward (stage1/modules/layers.py:270-297).  The attention test uses the reference's own `curr_res`, which
        starts at `resolution` even when conv_in halves the image (layers.py:221)."""
        s = self.s
        n = len(s.ch_mult)
        w = self.w
        if s.use_init_downsample:
            h = conv2d_strided(x, w['encoder.conv_in.weight'], w['encoder.conv_in.bias'], 2, (1, 1, 1, 1))
        else:
            h = self._conv('encoder.conv_in', x)
        label = s.resolution
        for lvl in range(n):
            for blk in range(s.num_res_blocks):
                h = self._resblock(f'encoder.down.{lvl}.block.{blk}', h)
                if label in s.attn_resolutions and s.use_attn:
                    h = self._attnblock(f'encoder.down.{lvl}.attn.{blk}', h)
            if lvl != n - 1:
                name = f'encoder.down.{lvl}.downsample.conv'
                h = conv2d_strided(h, w[f'{name}.weight'], w[f'{name}.bias'], 2, (0, 1, 0, 1))
                label //= 2
        if s.use_mid_block:
            h = self._resblock('encoder.mid.block_1', h)
            if s.use_attn:
                h = self._attnblock('encoder.mid.attn_1', h)
            h = self._resblock('encoder.mid.block_2', h)
        h = swish(self._gn('encoder.norm_out', h))
        return self._conv('encoder.conv_out', h)

    def encode(self, x: np.ndarray) -> Dict[str, object]:
        """SimRQGAN2Generator.encode (generator.py:298-310) or, with three code levels, HQVAEGenerator.encode
        (generator.py:530-568).  Returns codes / quant / resid / diff per level (coarse -> fine), `h` = quant_conv_b(encoder(x))
        and `recon` = the summed reconstruction at the bottom resolution."""
        w, s = self.w, self.s
        h = self._conv('quant_conv_b', self.encoder(x))
        out: Dict[str, object] = {'h': h, 'codes': [], 'quant': [], 'resid': [], 'diff': []}
        if s.code_levels == 3:
            h_map = [h]
            for _ in range(2):
                h_map.insert(0, pixel_unshuffle2(h_map[0]))
            recon = None
            for qi in range(3):
                resid = h_map[qi] if recon is None else (h_map[qi] - recon).astype(F32)
                q, diff, code = nearest_code(resid, w[f'quantizers.{qi}.embedding'])
                recon = q if recon is None else (q + recon).astype(F32)
                if qi < 2:
                    recon = pixel_shuffle2(recon)
                B, _, H, W = resid.shape
                out['codes'].append(code.reshape(B, H, W)); out['quant'].append(q); out['resid'].append(resid); out['diff'].append(diff)
            out['recon'] = recon
            return out
        h_t = pixel_unshuffle2(h)
        q_t, diff_t, code_t = nearest_code(h_t, w['quantize_t.embedding'])
        h_b = (h - pixel_shuffle2(q_t)).astype(F32)
        q_b, diff_b, code_b = nearest_code(h_b, w['quantize_b.embedding'])
        B = x.shape[0]
        out['codes'] = [code_t.reshape(B, *h_t.shape[2:]), code_b.reshape(B, *h_b.shape[2:])]
        out['quant'] = [q_t, q_b]
        out['resid'] = [h_t, h_b]
        out['diff'] = [diff_t, diff_b]
        out['recon'] = (q_b + pixel_shuffle2(q_t)).astype(F32)
        return out

    def decode_codes3(self, codes: Sequence[Optional[np.ndarray]]) -> np.ndarray:
        """HQVAEGenerator.decode_code (generator.py:577-599): codes = [top [B, r/4, r/4], mid [B, r/2, r/2], bottom [B, r, r]]
        (None = zero quant of that level); quant = PS(PS(q0) + q1) + q2, then post_quant_conv_b and the decoder."""
        w, s = self.w, self.s
        ref = next(c for c in codes if c is not None)
        B = ref.shape[0]
        quant = None
        for hi, code in enumerate(codes):
            r = s.z_res // 2 ** (2 - hi)
            dim = s.embed_dim * 4 ** (2 - hi)
            q = (w[f'quantizers.{hi}.embedding'][code].transpose(0, 3, 1, 2) if code is not None
                 else np.zeros((B, dim, r, r), F32))
            quant = q if quant is None else (quant + q).astype(F32)
            if hi < 2:
                quant = pixel_shuffle2(quant)
        z = self._conv('post_quant_conv_b', quant.astype(F32))
        return self.decoder(z)

    def decode_code(self, code_t: Optional[np.ndarray], code_b: Optional[np.ndarray]) -> np.ndarray:
        """code_t int64 [B, r/2, r/2] or None, code_b int64 [B, r, r] or None -> fp32 [B, 3, H, W], unclamped.
        A missing level contributes a zero quant (generator.py:328-358)."""
        assert code_t is not None or code_b is not None
        w, s = self.w, self.s
        if code_t is not None:
            qt = w['quantize_t.embedding'][code_t].transpose(0, 3, 1, 2)              # quantizer.py:179-186
        if code_b is not None:
            qb = w['quantize_b.embedding'][code_b].transpose(0, 3, 1, 2)
        if code_t is None:
            qt = np.zeros((qb.shape[0], qb.shape[1] * 4, qb.shape[2] // 2, qb.shape[3] // 2), F32)
        if code_b is None:
            qb = np.zeros((qt.shape[0], qt.shape[1] // 4, qt.shape[2] * 2, qt.shape[3] * 2), F32)
        quant = np.concatenate([pixel_shuffle2(qt), qb], axis=1).astype(F32)          # generator.py:316-318
        z = self._conv('post_quant_conv_b', quant)
        return self.decoder(z)


def postprocess(pixels: np.ndarray) -> np.ndarray:
    """clamp(0.5 x + 0.5, 0, 1) -- measure_throughput/__main__.py:113, sampling_hqmodel.py:198-199."""
    return np.clip(F32(0.5) * pixels + F32(0.5), 0.0, 1.0).astype(F32)


def rearrange_codes3(c0: np.ndarray, c1: np.ndarray, c2: np.ndarray, top_res: int):
    """sampling_hqmodel.py:150-153: 'B (H W) -> B H W' and 'B (H W) (kh kw) -> B (H kh) (W kw)' with kh = 2 and 4."""
    B, K = c0.shape[0], top_res
    return (c0.reshape(B, K, K),
            c1.reshape(B, K, K, 2, 2).transpose(0, 1, 3, 2, 4).reshape(B, 2 * K, 2 * K),
            c2.reshape(B, K, K, 4, 4).transpose(0, 1, 3, 2, 4).reshape(B, 4 * K, 4 * K))


def rearrange_codes(codes_top: np.ndarray, codes_bot: np.ndarray, top_res: int) -> Tuple[np.ndarray, np.ndarray]:
    """sampling_hqmodel.py:119-120: 'B (H W) -> B H W' and 'B (H W) (kh kw) -> B (H kh) (W kw)', kh = kw = 2."""
    B = codes_top.shape[0]
    ct = codes_top.reshape(B, top_res, top_res)
    cb = codes_bot.reshape(B, top_res, top_res, 2, 2).transpose(0, 1, 3, 2, 4).reshape(B, 2 * top_res, 2 * top_res)
    return ct, cb
