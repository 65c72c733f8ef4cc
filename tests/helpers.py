"""Shared test helpers: fixture loading and oracle construction (the oracle is the checker, never the product)."""
import json
import os

import numpy as np

from hqtransformer_amd import synth
from hqtransformer_amd.spec import Stage1Spec, Stage2Spec
from oracle.hqt_oracle import OracleStage1, OracleStage2

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def stage2_from_fixture(fx):
    spec = Stage2Spec(**json.loads(str(fx['spec'])))
    weights = synth.stage2_weights(spec, int(fx['weight_seed']), 'fixture')
    return spec, weights


def stage1_from_fixture(fx):
    spec = Stage1Spec(**json.loads(str(fx['spec'])))
    weights = synth.stage1_weights(spec, int(fx['weight_seed']), 'fixture')
    return spec, weights


def oracle_stage2(fx):
    spec, weights = stage2_from_fixture(fx)
    return spec, weights, OracleStage2(spec, weights)


def oracle_stage1(fx):
    spec, weights = stage1_from_fixture(fx)
    return spec, weights, OracleStage1(spec, weights)


def gate(name: str, value: float, limit: float, op: str = '<=') -> None:
    """Assert ``value op limit`` for a FAST-precision tolerance gate.  With HQT_RECORD_GATES=<file> the measured value is
    appended to that JSON-lines file too (profiles/r02_fast_gates.txt is kept from such a run): the gates are set at about twice the
    measured figure, so a regression of the bf16 path shows up long before it reaches the old blanket 0.1 / 0.15 bounds."""
    value, limit = float(value), float(limit)
    path = os.environ.get('HQT_RECORD_GATES')
    if path:
        with open(path, 'a') as fp:
            fp.write(json.dumps({'gate': name, 'measured': value, 'limit': limit, 'op': op}) + '\n')
    ok = value <= limit if op == '<=' else value >= limit
    assert ok, f'{name}: measured {value} violates {op} {limit}'


def philox_exp_noise(row_seeds, global_rows, n_steps: int, V: int, draws: int = 5) -> np.ndarray:
    """The Exp(1) noise libhqt's sampler generates in-kernel (csrc/kernels.hip, sampler_kernel): Philox4x32-10 with key = the
    row's 64-bit seed, counter = (vocabulary index // 4, step * draws + draw, global row lo, global row hi), lane v % 4 of the
    output, q = -log((float(r >> 9) + 0.5) * 2^-23) in fp32 (23 random bits: the sum is exact and u < 1, so q > 0 always).  The RNG is this build's own design (the reference draws from
    torch's global generator), so this restatement is test infrastructure: it lets the oracle replay what a Philox-driven call
    drew.  Returns [n_steps, draws, B, V] fp32."""
    seeds = np.asarray(row_seeds, np.uint64)
    rows = np.asarray(global_rows, np.int64).astype(np.uint64)
    B = len(seeds)
    V4 = (V + 3) // 4
    M32 = np.uint64(0xFFFFFFFF)
    out = np.empty((n_steps, draws, B, V4 * 4), np.float32)
    c0_init = np.broadcast_to(np.arange(V4, dtype=np.uint64)[None, :], (B, V4))
    for step in range(n_steps):
        for d in range(draws):
            c0 = c0_init.copy()
            c1 = np.full((B, V4), step * draws + d, np.uint64)
            c2 = np.broadcast_to((rows & M32)[:, None], (B, V4)).copy()
            c3 = np.broadcast_to((rows >> np.uint64(32))[:, None], (B, V4)).copy()
            k0 = np.broadcast_to((seeds & M32)[:, None], (B, V4)).copy()
            k1 = np.broadcast_to((seeds >> np.uint64(32))[:, None], (B, V4)).copy()
            for _ in range(10):
                p0 = np.uint64(0xD2511F53) * c0
                p1 = np.uint64(0xCD9E8D57) * c2
                hi0, lo0 = p0 >> np.uint64(32), p0 & M32
                hi1, lo1 = p1 >> np.uint64(32), p1 & M32
                c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
                k0 = (k0 + np.uint64(0x9E3779B9)) & M32
                k1 = (k1 + np.uint64(0xBB67AE85)) & M32
            r = np.stack([c0, c1, c2, c3], axis=-1).reshape(B, V4 * 4)
            u = ((r >> np.uint64(9)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 8388608.0)
            out[step, d] = -np.log(u, dtype=np.float32)
    return out[..., :V]
