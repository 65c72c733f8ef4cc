"""ctypes binding of the CPU twin (oracle/cpu/hqt_cpu.cpp, include/hqt_cpu.h) -- TEST INFRASTRUCTURE and bench.py's cpu_baseline leg.

Not product code: nothing under ``hqtransformer_amd/`` imports this module, and the product has no CPU compute path.  The twin restates
the reference's CPU path (fp32) in C++ / OpenMP; ``tests/test_cpu_twin.py`` pins it to the reference-generated fixtures.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional, Sequence

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, '_build', 'libhqt_cpu.so')
_lib = None

SYMBOLS = ('hqt_cpu_create', 'hqt_cpu_set_weight', 'hqt_cpu_finalize_weights', 'hqt_cpu_sample', 'hqt_cpu_decode', 'hqt_cpu_decode_seq',
           'hqt_cpu_last_seconds', 'hqt_cpu_threads', 'hqt_cpu_isa', 'hqt_cpu_destroy', 'hqt_cpu_last_error')


def build(verbose: bool = False) -> str:
    """`make -C oracle` (g++ -O3 -fopenmp; the GEMM micro-kernel twice: AVX2 and AVX-512, chosen at run time)."""
    proc = subprocess.run(['make', '-C', HERE], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose and proc.stdout.strip():
        print(proc.stdout)
    if proc.returncode != 0:
        raise RuntimeError('building the CPU twin failed:\n' + proc.stdout)
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    from hqtransformer_amd._lib import hqt_config, hqt_sample_opts
    lib = C.CDLL(LIB_PATH)
    VP = C.c_void_p
    sig = {
        'hqt_cpu_create': (C.c_int, [C.POINTER(hqt_config), C.c_int, C.POINTER(VP)]),
        'hqt_cpu_set_weight': (C.c_int, [VP, C.c_char_p, VP, C.POINTER(C.c_int64), C.c_int]),
        'hqt_cpu_finalize_weights': (C.c_int, [VP]),
        'hqt_cpu_sample': (C.c_int, [VP, C.c_int, VP, C.POINTER(hqt_sample_opts), VP, VP, VP, VP, VP, VP]),
        'hqt_cpu_decode': (C.c_int, [VP, C.c_int, VP, VP, VP, C.c_int]),
        'hqt_cpu_decode_seq': (C.c_int, [VP, C.c_int, VP, VP, VP, C.c_int]),
        'hqt_cpu_last_seconds': (C.c_double, [VP]),
        'hqt_cpu_threads': (C.c_int, [VP]),
        'hqt_cpu_isa': (C.c_char_p, []),
        'hqt_cpu_destroy': (C.c_int, [VP]),
        'hqt_cpu_last_error': (C.c_char_p, []),
    }
    assert set(sig) == set(SYMBOLS)
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def _check(lib, code: int) -> None:
    if code != 0:
        raise RuntimeError(f'hqt_cpu error {code}: {lib.hqt_cpu_last_error().decode()}')


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class CpuTwin:
    """One hqt_cpu handle: stage-2 and / or stage-1 weights by reference state-dict name, then ``sample`` / ``decode_code``."""

    def __init__(self, s2=None, s1=None, weights2: Optional[Dict[str, np.ndarray]] = None, weights1: Optional[Dict[str, np.ndarray]] = None,
                 threads: int = 0):
        from hqtransformer_amd.engine import make_config
        self.lib = load()
        self.s2, self.s1 = s2, s1
        cfg = make_config(s2, s1, 1, s2.ctx_len_img if s2 is not None else 1)
        h = C.c_void_p()
        _check(self.lib, self.lib.hqt_cpu_create(C.byref(cfg), int(threads), C.byref(h)))
        self.h = h
        for prefix, sd in (('stage2.', weights2), ('stage1.', weights1)):
            for k, v in (sd or {}).items():
                a = np.ascontiguousarray(v, dtype=np.float32)
                shape = (C.c_int64 * a.ndim)(*a.shape)
                _check(self.lib, self.lib.hqt_cpu_set_weight(self.h, (prefix + k).encode(), _p(a), shape, a.ndim))
        _check(self.lib, self.lib.hqt_cpu_finalize_weights(self.h))

    @property
    def threads(self) -> int:
        return int(self.lib.hqt_cpu_threads(self.h))

    @property
    def isa(self) -> str:
        return self.lib.hqt_cpu_isa().decode()

    @property
    def last_seconds(self) -> float:
        return float(self.lib.hqt_cpu_last_seconds(self.h))

    def close(self) -> None:
        if getattr(self, 'h', None) is not None and self.h.value:
            self.lib.hqt_cpu_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sample(self, cond, batch: int, n_steps: int, noise: Optional[np.ndarray] = None, top_k: Sequence[Optional[int]] = (None, None),
               top_p: Sequence[Optional[float]] = (None, None), temperature: Sequence[float] = (1.0, 1.0),
               force_top: Optional[np.ndarray] = None, force_bot: Optional[np.ndarray] = None, return_logits: bool = False,
               seed: int = 0, sample_offset: int = 0):
        """Same arguments and results as ``OracleStage2.sample``; ``noise=None`` draws libhqt's Philox stream for (seed, sample_offset)."""
        from hqtransformer_amd._lib import hqt_sample_opts
        B, V = int(batch), self.s2.vocab_top
        o = hqt_sample_opts()
        o.n_steps = int(n_steps)
        o.top_k_top, o.top_k_bot = int(top_k[0] or 0), int(top_k[1] or 0)
        o.top_p_top, o.top_p_bot = float(top_p[0] or 0.0), float(top_p[1] or 0.0)
        o.temperature_top, o.temperature_bot = float(temperature[0]), float(temperature[1])
        o.seed, o.sample_offset = int(seed) & (2 ** 64 - 1), int(sample_offset)
        cond = None if cond is None or self.s2.cond == 0 else np.ascontiguousarray(cond, dtype=np.int64)
        noise = None if noise is None else np.ascontiguousarray(noise, dtype=np.float32)
        if noise is not None and noise.shape != (n_steps, 5, B, V):
            raise ValueError(f'noise: expected {(n_steps, 5, B, V)}, got {noise.shape}')
        ft = None if force_top is None else np.ascontiguousarray(force_top, dtype=np.int64)
        fb = None if force_bot is None else np.ascontiguousarray(force_bot, dtype=np.int64)
        ct = np.zeros((B, n_steps), np.int64)
        cb = np.zeros((B, n_steps, 4), np.int64)
        lg = np.zeros((n_steps, 5, B, V), np.float32) if return_logits else None
        _check(self.lib, self.lib.hqt_cpu_sample(self.h, B, _p(cond), C.byref(o), _p(noise), _p(ft), _p(fb), _p(lg), _p(ct), _p(cb)))
        return (ct, cb, lg) if return_logits else (ct, cb)

    def decode_code(self, code_t: Optional[np.ndarray], code_b: Optional[np.ndarray], clamp01: bool = False, seq_layout: bool = False) -> np.ndarray:
        ref = code_t if code_t is not None else code_b
        B = int(ref.shape[0])
        ct = None if code_t is None else np.ascontiguousarray(code_t, dtype=np.int64)
        cb = None if code_b is None else np.ascontiguousarray(code_b, dtype=np.int64)
        H = self.s1.resolution
        out = np.zeros((B, self.s1.out_ch, H, H), np.float32)
        fn = self.lib.hqt_cpu_decode_seq if seq_layout else self.lib.hqt_cpu_decode
        _check(self.lib, fn(self.h, B, _p(ct), _p(cb), _p(out), int(clamp01)))
        return out
