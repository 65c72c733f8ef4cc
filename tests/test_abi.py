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
    """sizeof / offsetof of every ABI struct as the C compiler sees include/hqt.h against the ctypes mirror in _lib.py."""
    import ctypes as C
    import subprocess
    import tempfile
    checks = {'hqt_config': ['abi_version', 'has_stage1', 's1_ch_mult', 's1_attn_res', 'max_batch', 'code_levels', 'depth_decoding', 'ar_layouts'],
              'hqt_sample_opts': ['precision', 'top_p_top', 'seed', 'sample_offset', 'use_graph', 'row_seeds', 'row_offsets'],
              'hqt_sample_opts_l3': ['top_k', 'top_p', 'temperature', 'seed', 'use_graph', 'row_seeds', 'row_offsets'],
              'hqt_encode_out': ['codes', 'quant', 'resid', 'recon', 'diff']}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "hqt.h")}"', 'int main(void) {']
    for st, fields in checks.items():
        lines.append(f'  printf("{st} %zu\\n", sizeof({st}));')
        for f in fields:
            lines.append(f'  printf("{st}.{f} %zu\\n", offsetof({st}, {f}));')
    lines += ['  return 0;', '}']
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, 'l.c'), os.path.join(d, 'l')
        open(src, 'w').write('\n'.join(lines))
        subprocess.run(['gcc', '-o', exe, src], check=True)
        out = subprocess.run([exe], check=True, stdout=subprocess.PIPE, text=True).stdout
    want = dict(l.split() for l in out.strip().splitlines())
    for st, fields in checks.items():
        cls = getattr(_lib, st)
        assert C.sizeof(cls) == int(want[st]), (st, C.sizeof(cls), want[st])
        for f in fields:
            assert getattr(cls, f).offset == int(want[f'{st}.{f}']), (st, f)
    assert C.sizeof(_lib.hqt_sample_opts) == 72 and C.sizeof(_lib.hqt_sample_opts_l3) == 88


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
