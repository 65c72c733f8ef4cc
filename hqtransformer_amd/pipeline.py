"""Several batches in flight on one GPU.

The reference's harness (``measure_throughput/__main__.py:84-116``) samples and decodes one batch at a time.  On an
MI355X the 64-row AR loop is a dependent chain of ~100 small kernels per position, each of which keeps well under half
of the 256 CUs busy and is bound by per-CU latency; independent chains interleave almost for free.  ``InflightSampler``
therefore round-robins consecutive batches over N *lanes*: every lane has its own stream, KV cache and activations
(``hqt_clone``) and shares the weights, every batch is still one complete ``sampling_ihqgpt`` + ``decode_code`` pass
of the configured batch size, and results do not depend on the lane (bit-identical, tests/test_gpu_surface.py).

``merge=k`` additionally executes k queued steps as ONE pass of k x B rows (class-conditional, unconditional, text-conditional; two or
three code levels): the steps stay independent -- every row keeps the class id / prompt, the Philox seed and the global row index of its own step (``hqt_sample_opts.row_seeds`` / ``row_offsets``), so in EXACT
arithmetic each step's codes are bit-identical to the unmerged call (tests/test_gpu_surface.py) -- but the weights are streamed
once for all of them instead of once per step.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from ._lib import POLICY_LATENCY, POLICY_THROUGHPUT
from .sampling import sampling_hqtransformer, sampling_ihqgpt


def _same(a, b) -> bool:
    """Equality of sampler settings that may hold tensors (given codes): tensors compare by value, never through ``==``."""
    if torch.is_tensor(a) or torch.is_tensor(b):
        return torch.is_tensor(a) and torch.is_tensor(b) and a.shape == b.shape and bool(torch.equal(a.cpu(), b.cpu()))
    if isinstance(a, (list, tuple)) and isinstance(b, (list, tuple)):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, dict) and isinstance(b, dict):
        return a.keys() == b.keys() and all(_same(a[k], b[k]) for k in a)
    return a == b


def _settings(entry) -> tuple:
    """What the steps of one merged pass must share: max_seq_len, use_fp16, precision, clamp01, use_graph, sampler settings."""
    return (entry[4], entry[5], entry[6], entry[7], entry[8], {k: v for k, v in entry[11].items() if k != 'sample_offset'})


class Pending:
    """Result of a step queued on a merging sampler: filled when its group is launched (``InflightSampler.flush`` / ``drain``)."""
    __slots__ = ('value',)

    def __init__(self):
        self.value = None

    def get(self):
        if self.value is None:
            raise RuntimeError('the step has not been launched yet: call flush() or drain() first')
        return self.value


class InflightSampler:
    def __init__(self, model, lanes: int = 3, device: Optional[torch.device] = None, merge: int = 1, record_phases: bool = False,
                 ar_high_priority: bool = False):
        if lanes < 1 or merge < 1:
            raise ValueError('lanes and merge must be >= 1')
        self.model = model
        self.merge = int(merge)
        self.record_phases = bool(record_phases)     # merged passes: (AR start, AR end, decode end) events per pass, appended to phase_log
        self.phase_log: list = []
        self._queue: list = []
        self.n = int(lanes)
        self.device = device if device is not None else model.stage2._device
        self.streams: List[torch.cuda.Stream] = [torch.cuda.Stream(device=self.device) for _ in range(self.n)]
        # ar_high_priority: a lane's AR loop runs on a stream of the highest priority, its decode on the lane's own stream behind an event.
        # The AR kernels of a pass are thousands of short dependent launches; on equal terms each of them queues behind the other lane's
        # convolution workgroups for compute units (tools/diag_overlap.py: 4 % overlap).  Priority decides who gets a freed slot first.
        self.ar_streams: List[torch.cuda.Stream] = []
        if ar_high_priority:
            hi = -1
            if hasattr(torch.cuda.Stream, 'priority_range'):
                hi = torch.cuda.Stream.priority_range()[1]
            self.ar_streams = [torch.cuda.Stream(device=self.device, priority=hi) for _ in range(self.n)]
        self.k = 0

    def submit(self, num_candidates: int, cond, *, seed: Optional[int] = None, max_seq_len: int = 64, use_fp16: bool = True,
               decode: bool = True, precision: Optional[str] = None, clamp01: bool = True, use_graph: bool = True,
               after=None, phase_events=None, order_after_current: bool = True, **sample_kw) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor], torch.cuda.Event]:
        """Queue one batch on the next lane; returns (codes_top, codes_bot, pixels or None, done_event) immediately.
        The tensors are valid once ``done_event`` has completed (or after ``drain()``).  ``phase_events``: three timing
        events recorded on the lane's stream at AR start / AR end / decode end (lane time: phases of different lanes overlap)."""
        if self.merge > 1:
            if decode is False or phase_events is not None or sample_kw.get('noise') is not None:
                raise ValueError('merged steps support the plain sample + decode step only (no explicit noise, no phase events)')
            p = Pending()
            entry = (p, num_candidates, cond, seed, max_seq_len, use_fp16, precision, clamp01, use_graph, after, order_after_current, sample_kw)
            # checked HERE, before the step is queued: a mismatch raises without touching the queue (every Pending already handed out stays valid)
            if self._queue and not _same(_settings(entry), _settings(self._queue[0])):
                raise ValueError('steps merged into one pass must share max_seq_len, precision and sampler settings (flush() first to start a new pass)')
            self._queue.append(entry)
            if len(self._queue) >= self.merge:
                self.flush()
            return p
        return self._launch(num_candidates, cond, seed=seed, max_seq_len=max_seq_len, use_fp16=use_fp16, decode=decode, precision=precision,
                            clamp01=clamp01, use_graph=use_graph, after=after, phase_events=phase_events, order_after_current=order_after_current,
                            **sample_kw)

    def flush(self) -> None:
        """Launch the queued steps (merge > 1) as one pass; their Pending objects receive (codes_top, codes_bot, pixels, done_event)."""
        q = self._queue
        if not q:
            return
        ref = q[0]

        self._queue = []
        sizes = [e[1] for e in q]
        kw = dict(ref[11])
        offs = [int(e[11].get('sample_offset', 0)) for e in q]
        kw.pop('sample_offset', None)
        cls = getattr(self.model.stage2, 'use_cls_cond', False)
        cond = None
        if getattr(self.model.stage2, 'use_txt_cond', False):           # [n, ctx_len_txt] token ids per step
            cond = torch.cat([torch.as_tensor(e[2]).reshape(n, -1).to('cpu', torch.int64) for e, n in zip(q, sizes)])
        elif cls:
            parts = []
            for e, n in zip(q, sizes):
                c = torch.as_tensor(e[2]).reshape(-1).to('cpu', torch.int64)
                parts.append(c.expand(n) if c.numel() == 1 else c)
            cond = torch.cat(parts)
        seeds = [int(e[3]) if e[3] is not None else int(torch.randint(0, 2 ** 62, (1,)).item()) for e in q]
        row_seeds = [s for s, n in zip(seeds, sizes) for _ in range(n)]
        row_offsets = [o + i for o, n in zip(offs, sizes) for i in range(n)]
        afters = [e[9] for e in q]

        def split_after(ct, cb, px):
            lo = 0
            for n, a in zip(sizes, afters):
                if a is not None:
                    a(ct[lo:lo + n], [c[lo:lo + n] for c in cb] if isinstance(cb, (list, tuple)) else cb[lo:lo + n], None if px is None else px[lo:lo + n])
                lo += n
        phases = None
        if self.record_phases:
            phases = tuple(torch.cuda.Event(enable_timing=True) for _ in range(3))
            self.phase_log.append((phases, sum(sizes)))
        ct, cb, px, ev = self._launch(sum(sizes), cond, seed=seeds[0], max_seq_len=ref[4], use_fp16=ref[5], decode=True, precision=ref[6], clamp01=ref[7],
                                      use_graph=ref[8], after=split_after if any(a is not None for a in afters) else None, phase_events=phases,
                                      order_after_current=any(e[10] for e in q), row_seeds=row_seeds, row_offsets=row_offsets, **kw)
        lo = 0
        for e, n in zip(q, sizes):
            e[0].value = (ct[lo:lo + n], [c[lo:lo + n] for c in cb] if isinstance(cb, (list, tuple)) else cb[lo:lo + n],
                          None if px is None else px[lo:lo + n], ev)
            lo += n

    def _launch(self, num_candidates: int, cond, *, seed: Optional[int] = None, max_seq_len: int = 64, use_fp16: bool = True,
                decode: bool = True, precision: Optional[str] = None, clamp01: bool = True, use_graph: bool = True,
                after=None, phase_events=None, order_after_current: bool = True, **sample_kw):
        lane = self.k % self.n
        self.k += 1
        # `ar_precision` ('exact' | 'fast' | 'split', optional, travels with the sampler settings): arithmetic of the AR loop, overriding
        # use_fp16 (sampling_ihqgpt's `precision`; the `precision` of this method is the DECODE arithmetic)
        ar_precision = sample_kw.pop('ar_precision', None)
        st = self.streams[lane]
        caller = torch.cuda.current_stream(self.device)
        # order the lane after whatever the caller's stream has queued (inputs; earlier direct use of lane 0's engine):
        # a lane's workspace must never be touched from two streams at once
        if order_after_current:                      # False: the caller has nothing queued that this batch depends on (keeps the null stream's queue idle)
            st.wait_stream(torch.cuda.current_stream(self.device))
        if self.n > 1:
            # several batches in flight: kernels that cost the fewest CU-microseconds (hqt_set_policy).  The policy lives on the
            # engine object, so a lane rebuilt for a larger batch (or after the model dropped its engines) gets it again.
            eng = self.model.stage2.engine(num_candidates, max_seq_len, lane)
            if eng.policy != POLICY_THROUGHPUT:
                eng.set_policy(POLICY_THROUGHPUT)
        ast = self.ar_streams[lane] if self.ar_streams else st
        three = getattr(self.model.stage2.spec, 'levels', 2) == 3
        if ast is not st:
            ast.wait_stream(st)                      # the lane stays one in-order sequence: AR of this pass behind the lane's previous decode
        with torch.cuda.stream(ast):
            if phase_events is not None:
                phase_events[0].record(ast)
            if three:                                # HQTransformer: (codes0, [codes1, codes2]) keeps the 4-tuple shape of the result
                codes = sampling_hqtransformer(self.model.stage2, num_candidates=num_candidates, cond=cond, seed=seed, max_seq_len=max_seq_len,
                                               use_fp16=use_fp16, is_tqdm=False, use_graph=use_graph, lane=lane, precision=ar_precision, **sample_kw)
                ct, cb = codes[0], codes[1:]
            else:
                ct, cb = sampling_ihqgpt(self.model.stage2, num_candidates=num_candidates, cond=cond, seed=seed, max_seq_len=max_seq_len,
                                         use_fp16=use_fp16, is_tqdm=False, use_graph=use_graph, lane=lane, precision=ar_precision, **sample_kw)
            if phase_events is not None:
                phase_events[1].record(ast)
        if ast is not st:
            st.wait_stream(ast)
            for t in (ct, *(cb if isinstance(cb, (list, tuple)) else (cb,))):
                t.record_stream(st)
        with torch.cuda.stream(st):
            px = None
            if decode:
                px = self.model.stage1.decode_sequences([ct] + list(cb) if three else ct, None if three else cb,
                                                        precision=precision or ('fast' if use_fp16 else 'exact'), clamp01=clamp01, lane=lane)
            if phase_events is not None:
                phase_events[2].record(st)
            if after is not None:
                after(ct, cb, px)                    # e.g. a gather of the finished pixels, queued on the lane's stream
            ev = torch.cuda.Event()
            ev.record(st)
        # the results were allocated on the lane's stream and will be read (and eventually freed) on the caller's: tell the
        # caching allocator, or it may hand the memory to the lane again while the caller's stream still reads it
        for t in (ct, px, *(cb if isinstance(cb, (list, tuple)) else (cb,))):
            if t is not None:
                t.record_stream(caller)
        return ct, cb, px, ev

    def release(self, batch: int, max_seq_len: int) -> None:
        """Back to the latency-oriented kernels on lane 0 (the engine direct ``sampling_ihqgpt`` calls use)."""
        self.drain()
        if self.n > 1:
            self.model.stage2.engine(batch, max_seq_len, 0).set_policy(POLICY_LATENCY)

    def drain(self) -> None:
        """Launch what is still queued (merge > 1), wait for every lane; also orders the caller's stream after the lanes."""
        self.flush()
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams:
            cur.wait_stream(st)
            st.synchronize()
        self.model.stage1.range_check()          # SPLIT decodes: an activation outside the fp16 range invalidates the pass (raises)
        self.model.stage2.range_check()          # ... and SPLIT AR passes (ar_precision='split')
