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
