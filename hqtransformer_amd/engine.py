"""Thin torch-tensor wrapper over one libhqt handle.

PyTorch is plumbing here (device memory, streams); all arithmetic happens inside libhqt.so.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import PRECISION_EXACT, PRECISION_FAST, hqt_config, hqt_encode_out, hqt_sample_opts, hqt_sample_opts_l3
from .spec import DEPTH_DECODINGS, Stage1Spec, Stage2Spec


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def make_config(s2: Optional[Stage2Spec], s1: Optional[Stage1Spec], max_batch: int, max_steps: int, ar_layouts: int = 0) -> hqt_config:
    c = hqt_config()
    c.abi_version = _lib.ABI_VERSION
    c.max_batch = int(max_batch)
    c.max_steps = int(max_steps)
    c.ar_layouts = int(ar_layouts)
    if s2 is not None:
        c.has_stage2 = 1
        c.embed_dim, c.n_layers, c.n_heads, c.n_layers_depth = s2.embed_dim, s2.n_layers, s2.n_heads, s2.n_layers_depth
        c.vocab_top, c.vocab_bot, c.vocab_txt = s2.vocab_top, s2.vocab_bot, s2.vocab_txt
        c.ctx_len_img, c.ctx_len_txt, c.n_classes = s2.ctx_len_img, s2.ctx_len_txt, s2.n_classes
        c.cond_type, c.embedding_type, c.gelu_approx = s2.cond, s2.embedding, int(s2.gelu_approx)
        c.depth_decoding = DEPTH_DECODINGS.index(getattr(s2, 'depth_decoding', 'parallel-add'))
    if (s2 is not None and getattr(s2, 'levels', 2) == 3) or (s1 is not None and getattr(s1, 'code_levels', 2) == 3):
        c.code_levels = 3
    if s1 is not None:
        c.has_stage1 = 1
        c.s1_ch, c.s1_n_mult = s1.ch, len(s1.ch_mult)
        for i, m in enumerate(s1.ch_mult):
            c.s1_ch_mult[i] = m
        c.s1_num_res_blocks = s1.num_res_blocks
        c.s1_n_attn_res = len(s1.attn_resolutions)
        for i, r in enumerate(s1.attn_resolutions):
            c.s1_attn_res[i] = r
        c.s1_resolution, c.s1_z_channels, c.s1_embed_dim = s1.resolution, s1.z_channels, s1.embed_dim
        c.s1_n_embed, c.s1_out_ch = s1.n_embed, s1.out_ch
        c.s1_use_init_downsample, c.s1_use_mid_block, c.s1_use_attn = (int(s1.use_init_downsample), int(s1.use_mid_block),
                                                                        int(s1.use_attn))
    return c


# (data_ptr, numel) -> tensor version of device tensors whose indices are known to be in range: produced by a sampler of this
# process (any engine: the stage-2 engine's codes go to the stage-1 engine's decode) or already validated once
_TRUSTED: Dict[tuple, tuple] = {}       # (data_ptr, shape, stride, dtype) -> (tensor version, weakref to its storage, largest valid id + 1)


class Engine:
    """One libhqt handle on one GPU.  Not thread-safe; asynchronous on torch's current stream."""

    def __init__(self, s2: Optional[Stage2Spec], s1: Optional[Stage1Spec], device: torch.device, max_batch: int,
                 max_steps: Optional[int] = None, ar_layouts: int = 0):
        """``ar_layouts``: bit mask of ``_lib.LAYOUT_*`` -- which derived layouts of the AR loop's weights ``finalize`` builds
        (``hqt_config.ar_layouts``; 0 = all).  A FAST-only replica passes ``_lib.LAYOUT_FAST`` and holds 5.2 instead of 9.1 GB for the
        ImageNet-12L model; a call in a precision the engine was built without raises HqtError (HQT_ERR_STATE)."""
        self.lib = _lib.load()                      # raises HqtLibraryError when the HIP library is absent
        self.s2, self.s1 = s2, s1
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.HqtLibraryError(f'libhqt runs on an MI355X only; got device {self.device} (no CPU fallback)')
        self.max_batch = int(max_batch)
        self.max_steps = int(max_steps if max_steps is not None else (s2.ctx_len_img if s2 else 1))
        self.cfg = make_config(s2, s1, self.max_batch, self.max_steps, ar_layouts)
        h = C.c_void_p()
        _lib.check(self.lib.hqt_create(C.byref(self.cfg), self.device.index or 0, C.byref(h)))
        self.h = h
        self.finalized = False
        self.policy = _lib.POLICY_LATENCY

    def clone(self) -> 'Engine':
        """A further lane over the same weights (``hqt_clone``): own KV cache / activations / graph cache, results
        bit-identical to this engine's.  Lanes on different streams keep several batches in flight on one GPU."""
        if not self.finalized:
            raise _lib.HqtError(-1, 'clone() needs a finalized engine')
        e = Engine.__new__(Engine)
        e.lib, e.s2, e.s1, e.device = self.lib, self.s2, self.s1, self.device
        e.max_batch, e.max_steps, e.cfg = self.max_batch, self.max_steps, self.cfg
        h = C.c_void_p()
        _lib.check(self.lib.hqt_clone(self.h, C.byref(h)))
        e.h, e.finalized = h, True
        e.policy = _lib.POLICY_LATENCY               # hqt_clone resets a lane to the default policy
        e._parent = self                             # keeps the weights' owner alive
        self._clones = getattr(self, '_clones', [])
        self._clones.append(e)
        return e

    def set_policy(self, policy: int) -> None:
        """``hqt_set_policy``: 0 = latency-oriented kernel choice (one batch at a time), 1 = throughput-oriented (several
        lanes in flight).  Part of the graph key: the next sample() re-captures if it changed."""
        _lib.check(self.lib.hqt_set_policy(self.h, int(policy)))
        self.policy = int(policy)

    def set_persist(self, on: bool) -> None:
        """``hqt_set_switch(HQT_SWITCH_PERSIST)``: False = the launch chain instead of the persistent AR launch (FAST sampling of up to 64
        rows); True also re-arms a handle that fell back to the chain after a launch gave up.  Part of the graph key."""
        _lib.check(self.lib.hqt_set_switch(self.h, _lib.SWITCH_PERSIST, int(bool(on))))

    def set_persist_fault(self, cu_plus_one: int) -> None:
        """Test hook ``hqt_set_switch(HQT_SWITCH_PERSIST_FAULT)``: compute unit ``cu_plus_one - 1`` withholds its first grid-barrier signal in
        every later persistent launch (0 = none), cached graphs included."""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.hqt_set_switch(self.h, _lib.SWITCH_PERSIST_FAULT, int(cu_plus_one)))

    def set_split_kslices(self, on: bool) -> None:
        """``hqt_set_switch(HQT_SWITCH_SPLIT_KSLICES)``: False = the SPLIT AR GEMMs are never K-sliced (one fp32 summation order at every row count)."""
        _lib.check(self.lib.hqt_set_switch(self.h, _lib.SWITCH_SPLIT_KSLICES, int(bool(on))))

    def set_single_key(self, on: bool) -> None:
        """``hqt_set_switch(HQT_SWITCH_SINGLE_KEY)``: False = depth sub-step 0 the long way round (A/B runs and tests; bit-identical results)."""
        _lib.check(self.lib.hqt_set_switch(self.h, _lib.SWITCH_SINGLE_KEY, int(bool(on))))

    def _note_split(self, precision: int, stream: int, ar_rows: int = 0) -> None:
        # calls whose validity the device reports after the fact: SPLIT (an activation outside the fp16 range) and FAST sampling of up to
        # 64 rows (the persistent AR chain: a launch that could not get the whole GPU gives up after 1 s instead of hanging)
        if int(precision) == _lib.PRECISION_SPLIT or (int(precision) == _lib.PRECISION_FAST and 0 < ar_rows <= 64):
            self._split_streams = getattr(self, '_split_streams', set()) | {int(stream or 0)}

    def range_check(self) -> None:
        """hqt_range_check for every stream SPLIT-precision calls (and FAST sampling calls of up to 64 rows) of this engine were enqueued on
        since the last check: waits for them and raises HqtError if an activation left the fp16 range (HQT_ERR_RANGE) or a persistent AR
        launch gave up on its grid barrier (HQT_ERR_STATE) -- the output of such a call is invalid.  No-op -- and no synchronisation --
        when no such call is pending."""
        streams, self._split_streams = getattr(self, '_split_streams', set()), set()
        err = None
        with torch.cuda.device(self.device):
            for st in sorted(streams):
                try:
                    _lib.check(self.lib.hqt_range_check(self.h, C.c_void_p(st)))
                except _lib.HqtError as e:       # keep draining the other streams: the flag is per handle and already cleared
                    err = e
        if err is not None:
            raise err

    def close(self) -> None:
        for c in getattr(self, '_clones', []):       # clones go first: the parent owns the weights
            c.close()
        self._clones = []
        if getattr(self, 'h', None) is not None and self.h.value:
            self.lib.hqt_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def set_weight(self, name: str, t) -> None:
        if isinstance(t, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32))
        t = t.detach().to(dtype=torch.float32).contiguous()
        shape = (C.c_int64 * t.dim())(*t.shape)
        _lib.check(self.lib.hqt_set_weight(self.h, name.encode(), C.c_void_p(t.data_ptr()), 0, shape, t.dim()))

    def load(self, stage2: Optional[Dict[str, object]] = None, stage1: Optional[Dict[str, object]] = None) -> None:
        for prefix, sd in (('stage2.', stage2), ('stage1.', stage1)):
            for k, v in (sd or {}).items():
                self.set_weight(prefix + k, v)

    def finalize(self) -> None:
        _lib.check(self.lib.hqt_finalize_weights(self.h))
        self.finalized = True

    # ------------------------------------------------------------------ input validation
    # The reference indexes nn.Embedding / F.embedding tables with these tensors and raises IndexError for an id outside the
    # table.  Host tensors are checked on the host (free).  Device tensors need one reduction + a host read, i.e. a stream
    # synchronisation: they are checked once per (storage, version) and remembered, and tensors this engine produced itself
    # (sampled codes fed to decode) are trusted -- so a pipelined run (bench.py: lanes, sampled codes straight into decode)
    # never synchronises.  The kernels clamp every such index into its table regardless (csrc/common.h: clamp_idx).
    @staticmethod
    def _ident(t: torch.Tensor) -> tuple:
        return (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype)

    def _trust(self, *tensors, bound: int) -> None:
        """Remember that every id in these device tensors lies in [0, bound): a later check against a table of n >= bound rows needs
        no reduction; against a SMALLER table the tensor is checked again."""
        for t in tensors:
            if t is not None:
                # (address, size) alone is not an identity: the caching allocator hands a freed block to the next tensor of that size.
                # The storage object is: torch keeps one Python wrapper per live storage, and a weak reference to it dies with the storage.
                _TRUSTED[self._ident(t)] = (t._version, weakref.ref(t.untyped_storage()), int(bound))
                if len(_TRUSTED) > 256:
                    _TRUSTED.pop(next(iter(_TRUSTED)))

    def _check_index(self, t: Optional[torch.Tensor], n: int, what: str) -> None:
        if t is None or t.numel() == 0:
            return
        seen = _TRUSTED.get(self._ident(t)) if t.is_cuda else None
        if seen is not None and seen[0] == t._version and seen[1]() is t.untyped_storage() and seen[2] <= n:
            return
        lo, hi = (int(v) for v in torch.stack([t.min(), t.max()]).tolist())
        if lo < 0 or hi >= n:
            raise IndexError(f'{what}: index out of range (values span [{lo}, {hi}], table has {n} rows)')
        if t.is_cuda:
            self._trust(t, bound=hi + 1)

    @staticmethod
    def _check_out(t: torch.Tensor, shape, dtype, dev, what: str) -> torch.Tensor:
        if not isinstance(t, torch.Tensor) or tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != dev or not t.is_contiguous():
            raise ValueError(f'{what}: expected a contiguous {dtype} tensor of shape {tuple(shape)} on {dev}, got '
                             f'{getattr(t, "dtype", type(t))} {tuple(getattr(t, "shape", ()))} on {getattr(t, "device", "?")}')
        return t

    @staticmethod
    def _row_keys(o, B: int, row_seeds, row_offsets):
        """Fills ``o.row_seeds`` / ``o.row_offsets`` (host arrays of B entries) and returns them so that they outlive the call."""
        if row_seeds is None and row_offsets is None:
            return None
        if row_seeds is None or row_offsets is None or len(row_seeds) != B or len(row_offsets) != B:
            raise ValueError(f'row_seeds and row_offsets come together, {B} entries each')
        rs = (C.c_uint64 * B)(*[int(v) & (2 ** 64 - 1) for v in row_seeds])
        ro = (C.c_int64 * B)(*[int(v) for v in row_offsets])
        o.row_seeds, o.row_offsets = C.cast(rs, C.c_void_p), C.cast(ro, C.c_void_p)
        return rs, ro

    # ------------------------------------------------------------------ stage 2
    def sample(self, batch: int, cond: Optional[torch.Tensor], n_steps: int, *, precision: int = PRECISION_FAST,
               top_k: Sequence[Optional[int]] = (None, None), top_p: Sequence[Optional[float]] = (None, None),
               temperature: Sequence[float] = (1.0, 1.0), noise: Optional[torch.Tensor] = None, seed: int = 0,
               sample_offset: int = 0, force_top: Optional[torch.Tensor] = None, force_bot: Optional[torch.Tensor] = None,
               return_logits: bool = False, use_graph: bool = True,
               out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
               row_seeds: Optional[Sequence[int]] = None, row_offsets: Optional[Sequence[int]] = None):
        """``row_seeds`` / ``row_offsets`` (both or neither, ``batch`` entries): merged steps -- row b draws what the row with
        global index ``row_offsets[b]`` of a call seeded ``row_seeds[b]`` draws (``hqt_sample_opts.row_seeds``)."""
        dev = self.device
        B, V = int(batch), self.s2.vocab_top
        o = hqt_sample_opts()
        o.precision, o.n_steps = int(precision), int(n_steps)
        o.top_k_top = int(top_k[0]) if top_k[0] else 0
        o.top_k_bot = int(top_k[1]) if top_k[1] else 0
        o.top_p_top = float(top_p[0]) if top_p[0] else 0.0
        o.top_p_bot = float(top_p[1]) if top_p[1] else 0.0
        o.temperature_top, o.temperature_bot = float(temperature[0]), float(temperature[1])
        o.seed, o.sample_offset, o.use_graph = int(seed) & (2 ** 64 - 1), int(sample_offset), int(bool(use_graph))
        rows = self._row_keys(o, B, row_seeds, row_offsets)

        def prep(t, shape, dtype, what, table=0):
            if t is None:
                return None
            t = torch.as_tensor(t)
            if tuple(t.shape) != tuple(shape):
                raise ValueError(f'{what}: expected shape {tuple(shape)}, got {tuple(t.shape)}')
            if table:
                self._check_index(t, table, what)
            return t.to(device=dev, dtype=dtype).contiguous()
        if self.s2.cond == 1:
            cond = prep(cond, (B,), torch.int64, 'cond (class ids)', self.s2.n_classes)
        elif self.s2.cond == 2:
            cond = prep(cond, (B, self.s2.ctx_len_txt), torch.int64, 'cond (text token ids)', self.s2.vocab_txt)
        else:
            cond = None
        noise = prep(noise, (n_steps, 5, B, V), torch.float32, 'noise')
        force_top = prep(force_top, (B, n_steps), torch.int64, 'force_top', V)
        force_bot = prep(force_bot, (B, n_steps, 4), torch.int64, 'force_bot', V)
        if out is None:
            out_top = torch.empty((B, n_steps), dtype=torch.int64, device=dev)
            out_bot = torch.empty((B, n_steps, 4), dtype=torch.int64, device=dev)
        else:
            if not isinstance(out, (tuple, list)) or len(out) != 2:
                raise ValueError('out: expected a pair (codes_top [B, n_steps], codes_bot [B, n_steps, 4])')
            out_top = self._check_out(out[0], (B, n_steps), torch.int64, dev, 'out[0]')
            out_bot = self._check_out(out[1], (B, n_steps, 4), torch.int64, dev, 'out[1]')
        logits = torch.empty((n_steps, 5, B, V), dtype=torch.float32, device=dev) if return_logits else None
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _lib.check(self.lib.hqt_sample(self.h, B, _ptr(cond), C.byref(o), _ptr(noise), _ptr(force_top), _ptr(force_bot),
                                           _ptr(logits), _ptr(out_top), _ptr(out_bot), C.c_void_p(stream)))
            self._note_split(precision, stream, ar_rows=B)
        # inputs must outlive the asynchronous launches
        self._keep = (cond, noise, force_top, force_bot, rows)
        self._trust(out_top, out_bot, bound=max(self.s2.vocab_top, self.s2.vocab_bot))     # the sampler only writes ids inside the vocabulary
        if return_logits:
            return out_top, out_bot, logits
        return out_top, out_bot

    # ------------------------------------------------------------------ three code levels (hqt_sample_l3 / hqt_decode_l3)
    def sample3(self, batch: int, cond: Optional[torch.Tensor], n_steps: int, *, precision: int = PRECISION_FAST,
                top_k: Sequence[Optional[int]] = (None, None, None), top_p: Sequence[Optional[float]] = (None, None, None),
                temperature: Sequence[float] = (1.0, 1.0, 1.0), noise: Optional[torch.Tensor] = None, seed: int = 0,
                sample_offset: int = 0, force: Optional[Sequence[torch.Tensor]] = None, return_logits: bool = False,
                use_graph: bool = True, row_seeds: Optional[Sequence[int]] = None, row_offsets: Optional[Sequence[int]] = None):
        """Three-level sampling: returns (codes0 [B, n], codes1 [B, n, 4], codes2 [B, n, 16][, logits [n, 21, B, V]])."""
        dev = self.device
        B, V = int(batch), self.s2.vocab_top
        o = hqt_sample_opts_l3()
        o.precision, o.n_steps = int(precision), int(n_steps)
        for i in range(3):
            o.top_k[i] = int(top_k[i]) if top_k[i] else 0
            o.top_p[i] = float(top_p[i]) if top_p[i] else 0.0
            o.temperature[i] = float(temperature[i])
        o.seed, o.sample_offset, o.use_graph = int(seed) & (2 ** 64 - 1), int(sample_offset), int(bool(use_graph))
        rows = self._row_keys(o, B, row_seeds, row_offsets)

        def prep(t, shape, dtype, what, table=0):
            if t is None:
                return None
            t = torch.as_tensor(t)
            if tuple(t.shape) != tuple(shape):
                raise ValueError(f'{what}: expected shape {tuple(shape)}, got {tuple(t.shape)}')
            if table:
                self._check_index(t, table, what)
            return t.to(device=dev, dtype=dtype).contiguous()
        if self.s2.cond == 1:
            cond = prep(cond, (B,), torch.int64, 'cond (class ids)', self.s2.n_classes)
        elif self.s2.cond == 2:
            cond = prep(cond, (B, self.s2.ctx_len_txt), torch.int64, 'cond (text token ids)', self.s2.vocab_txt)
        else:
            cond = None
        noise = prep(noise, (n_steps, 21, B, V), torch.float32, 'noise')
        f = [None, None, None]
        if force is not None:
            f = [prep(force[0], (B, n_steps), torch.int64, 'force[0]', V), prep(force[1], (B, n_steps, 4), torch.int64, 'force[1]', V),
                 prep(force[2], (B, n_steps, 16), torch.int64, 'force[2]', V)]
        outs = [torch.empty(shp, dtype=torch.int64, device=dev) for shp in ((B, n_steps), (B, n_steps, 4), (B, n_steps, 16))]
        logits = torch.empty((n_steps, 21, B, V), dtype=torch.float32, device=dev) if return_logits else None
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _lib.check(self.lib.hqt_sample_l3(self.h, B, _ptr(cond), C.byref(o), _ptr(noise), _ptr(f[0]), _ptr(f[1]), _ptr(f[2]),
                                              _ptr(logits), _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]), C.c_void_p(stream)))
            self._note_split(precision, stream, ar_rows=B)        # the three-level body runs persistently too (run_position_l3)
        self._keep = (cond, noise, f, rows)
        self._trust(*outs, bound=max(self.s2.vocab_top, self.s2.vocab_bot))
        return (outs[0], outs[1], outs[2], logits) if return_logits else tuple(outs)

    def decode3(self, codes: Sequence[Optional[torch.Tensor]], *, precision: int = PRECISION_EXACT, clamp01: bool = False,
                seq_layout: bool = False) -> torch.Tensor:
        """``HQVAEGenerator.decode_code([t, m, b])``; ``seq_layout``: the sampler's [B, n], [B, n, 4], [B, n, 16]."""
        dev = self.device
        ref = next((c for c in codes if c is not None), None)
        if ref is None or len(codes) != 3:
            raise ValueError('decode3 takes three code tensors, at least one not None')
        B = int(ref.shape[0])
        r = self.s1.z_res
        n = (r // 4) ** 2
        want = ((B, n), (B, n, 4), (B, n, 16)) if seq_layout else ((B, r // 4, r // 4), (B, r // 2, r // 2), (B, r, r))
        cs = []
        for c, w in zip(codes, want):
            if c is not None:
                if tuple(c.shape) != w:
                    raise ValueError(f'code grid: expected {w}, got {tuple(c.shape)}')
                self._check_index(c, self.s1.n_embed, 'code grid')
                c = c.to(device=dev, dtype=torch.int64).contiguous()
            cs.append(c)
        H = self.s1.resolution
        out = torch.empty((B, self.s1.out_ch, H, H), dtype=torch.float32, device=dev)
        fn = self.lib.hqt_decode_seq_l3 if seq_layout else self.lib.hqt_decode_l3
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _lib.check(fn(self.h, B, _ptr(cs[0]), _ptr(cs[1]), _ptr(cs[2]), _ptr(out), int(clamp01), int(precision), C.c_void_p(stream)))
            self._note_split(precision, stream)
        self._keep_dec = cs
        return out

    # ------------------------------------------------------------------ stage 1, encode side
    @property
    def has_encoder(self) -> bool:
        return self.lib.hqt_has_encoder(self.h) == 1

    def encode(self, pixels: torch.Tensor, *, precision: int = PRECISION_EXACT, want_quant: bool = False,
               want_resid: bool = False, want_recon: bool = False, want_diff: bool = False) -> Dict[str, object]:
        """``SimRQGAN2Generator.encode`` / ``HQVAEGenerator.encode`` (generator.py:298-310, 530-568) through ``hqt_encode``:
        fp32 [B, 3, R, R] -> ``codes`` (list, coarse -> fine, int64 [B, r_l, r_l]) and on request ``quant`` / ``resid``
        (lists of fp32 [B, dim_l, r_l, r_l]), ``recon`` (fp32 [B, E, r, r]) and ``diff`` (fp32 [levels])."""
        dev = self.device
        s1 = self.s1
        x = pixels.to(device=dev, dtype=torch.float32).contiguous()
        B = int(x.shape[0])
        if tuple(x.shape) != (B, 3, s1.resolution, s1.resolution):
            raise ValueError(f'pixels: expected {(B, 3, s1.resolution, s1.resolution)}, got {tuple(x.shape)}')
        L = 3 if s1.code_levels == 3 else 2
        r, E = s1.z_res, s1.embed_dim
        o = hqt_encode_out()
        res: Dict[str, object] = {'codes': [], 'quant': [], 'resid': []}
        for l in range(L):
            k = L - 1 - l
            rq, dim = r >> k, E * 4 ** k
            c = torch.empty((B, rq, rq), dtype=torch.int64, device=dev)
            res['codes'].append(c)
            o.codes[l] = _ptr(c)
            if want_quant:
                q = torch.empty((B, dim, rq, rq), dtype=torch.float32, device=dev)
                res['quant'].append(q)
                o.quant[l] = _ptr(q)
            if want_resid:
                z = torch.empty((B, dim, rq, rq), dtype=torch.float32, device=dev)
                res['resid'].append(z)
                o.resid[l] = _ptr(z)
        if want_recon:
            res['recon'] = torch.empty((B, E, r, r), dtype=torch.float32, device=dev)
            o.recon = _ptr(res['recon'])
        if want_diff:
            res['diff'] = torch.empty((L,), dtype=torch.float32, device=dev)
            o.diff = _ptr(res['diff'])
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _lib.check(self.lib.hqt_encode(self.h, B, _ptr(x), int(precision), C.byref(o), C.c_void_p(stream)))
            self._note_split(precision, stream)
        self._keep_enc = x
        return res

    # ------------------------------------------------------------------ stage 1
    def decode(self, code_t: Optional[torch.Tensor], code_b: Optional[torch.Tensor], *, precision: int = PRECISION_EXACT,
               clamp01: bool = False, seq_layout: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        dev = self.device
        ref = code_t if code_t is not None else code_b
        if ref is None:
            raise ValueError('code_t and code_b are both None')
        B = int(ref.shape[0])
        r = self.s1.z_res
        if seq_layout:
            want_t, want_b = (B, (r // 2) ** 2), (B, (r // 2) ** 2, 4)
        else:
            want_t, want_b = (B, r // 2, r // 2), (B, r, r)
        if code_t is not None and tuple(code_t.shape) != want_t:
            raise ValueError(f'code_t: expected {want_t}, got {tuple(code_t.shape)}')
        if code_b is not None and tuple(code_b.shape) != want_b:
            raise ValueError(f'code_b: expected {want_b}, got {tuple(code_b.shape)}')
        self._check_index(code_t, self.s1.n_embed, 'code_t')
        self._check_index(code_b, self.s1.n_embed, 'code_b')
        code_t = None if code_t is None else code_t.to(device=dev, dtype=torch.int64).contiguous()
        code_b = None if code_b is None else code_b.to(device=dev, dtype=torch.int64).contiguous()
        H = self.s1.resolution
        if out is None:
            out = torch.empty((B, self.s1.out_ch, H, H), dtype=torch.float32, device=dev)
        else:
            self._check_out(out, (B, self.s1.out_ch, H, H), torch.float32, dev, 'out')
        fn = self.lib.hqt_decode_seq if seq_layout else self.lib.hqt_decode
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _lib.check(fn(self.h, B, _ptr(code_t), _ptr(code_b), _ptr(out), int(clamp01), int(precision), C.c_void_p(stream)))
            self._note_split(precision, stream)
        self._keep_dec = (code_t, code_b)
        return out

    # ------------------------------------------------------------------ timing (bench.py roofline numerator)
    def timing(self, on: bool) -> None:
        _lib.check(self.lib.hqt_timing_enable(self.h, int(on)))

    def timing_reset(self) -> None:
        _lib.check(self.lib.hqt_timing_reset(self.h))

    def timing_report(self) -> Dict[str, Tuple[int, float]]:
        out = {}
        for i in range(self.lib.hqt_timing_slots(self.h)):
            name = C.create_string_buffer(64)
            n, ms = C.c_int64(), C.c_double()
            _lib.check(self.lib.hqt_timing_get(self.h, i, name, 64, C.byref(n), C.byref(ms)))
            out[name.value.decode()] = (int(n.value), float(ms.value))
        return out

    def workspace_bytes(self) -> int:
        return int(self.lib.hqt_workspace_bytes(self.h))
