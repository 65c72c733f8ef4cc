"""The C-ABI library loads and exports every symbol include/hqt.h declares (no compute, no GPU)."""
import os
import re

import pytest

from hqtransformer_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    _lib.build()
    return _lib.load()


def test_header_symbols_are_exported_and_bound(lib):
    hdr = open(os.path.join(ROOT, 'include', 'hqt.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(hqt_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no declarations parsed'
    assert declared == set(_lib.exported_symbols()), declared ^ set(_lib.exported_symbols())
    for name in declared:
        assert getattr(lib, name) is not None


def test_abi_version_and_error_paths_without_gpu(lib):
    assert lib.hqt_abi_version() == _lib.ABI_VERSION
    assert lib.hqt_create(None, 0, None) == -1
    assert b'null' in lib.hqt_last_error()
    assert lib.hqt_param_count(None, 2) == -1
    assert lib.hqt_destroy(None) == 0


def test_struct_layout_matches_header():
    import ctypes as C
    assert C.sizeof(_lib.hqt_config) == 4 * (1 + 1 + 4 + 3 + 3 + 3 + 1 + 2 + 8 + 1 + 1 + 4 + 5 + 3 + 2 + 1)
    assert C.sizeof(_lib.hqt_sample_opts) == 56
    assert C.sizeof(_lib.hqt_sample_opts_l3) == 72 and _lib.hqt_sample_opts_l3.seed.offset == 48


def test_no_cpu_fallback():
    """The product refuses anything but the GPU instead of computing on the host."""
    import torch
    from hqtransformer_amd.engine import Engine
    from hqtransformer_amd.spec import Stage2Spec
    s2 = Stage2Spec(128, 1, 4, 1, 64, 64, 64, 64, 16, 10, 1, 0)
    with pytest.raises(_lib.HqtLibraryError):
        Engine(s2, None, torch.device('cpu'), 2)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'hqtransformer_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in src.replace('no CPU fallback', ''), f'{f} mentions the oracle'
