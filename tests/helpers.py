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
